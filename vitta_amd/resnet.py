"""ResNet-50 v1.5 trunk (stride on the 3x3 conv) with torchvision's state_dict key names.

The reference builds its TANet on `torchvision.models.resnet50` (models/tanet_models/tanet.py:129,
torchvision==0.8.2, NOT vendored in the reference).  This is a from-scratch restatement of that
published architecture so that `tanet_ucf.pth.tar` style checkpoints load key-for-key:
    conv1, bn1, layer{1..4}.{i}.{conv1,bn1,conv2,bn2,conv3,bn3,downsample.{0,1}}, fc
Parity of the trunk itself is unpinned by the reference; registration ORDER of
sub-modules is load-bearing because the source statistics are positional (corpus/basics.py:490-498).

Difference from torchvision kept on purpose: no in-place ReLU / `out += identity`.  The hooked BN
outputs must survive until the backward pass (the stat-loss gradient a_c + b_c (x - mu_c) reads
them), so nothing downstream may overwrite them.
"""
import torch
import torch.nn as nn

from .fused_bn import bn_act, identity_source


def downsample_forward(downsample, x):
    """`downsample(x)` for the standard Sequential(conv1x1, BatchNorm2d) with the BN on the fused path."""
    if isinstance(downsample, nn.Sequential) and len(downsample) == 2 and isinstance(downsample[1], nn.BatchNorm2d) \
            and not downsample._forward_hooks:
        return bn_act(downsample[1], downsample[0](x), relu=False)
    return downsample(x)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=False)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        src = identity_source(x)
        identity = src if self.downsample is None else downsample_forward(self.downsample, src)
        out = bn_act(self.bn1, self.conv1(x), relu=True, act=self.relu)
        out = bn_act(self.bn2, self.conv2(out), relu=True, act=self.relu)
        return bn_act(self.bn3, self.conv3(out), residual=identity, relu=True, act=self.relu, fork=True)


class ResNet(nn.Module):
    def __init__(self, block=Bottleneck, layers=(3, 4, 6, 3), num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=False)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0.0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion))
        stage = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        stage += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*stage)

    def _stem_fusable(self, y):
        """bn1 -> relu -> maxpool as one pass (vitta_stem_bn_relu_pool_*): eval-mode affine BN without hooks, the stock
        3/2/1 max-pool, and a convolution output that needs no gradient (frozen stem under update_only_bn_affine,
        or no_grad) -- the fused backward only produces d gamma / d beta."""
        from . import fused_bn
        bn, mp = self.bn1, self.maxpool
        return (fused_bn.ENABLED and y.is_cuda and y.dtype == torch.float32 and not y.requires_grad
                and isinstance(bn, nn.BatchNorm2d) and not bn.training and bn.affine
                and not bn._forward_hooks and not bn._forward_pre_hooks and not mp._forward_hooks and not self.relu._forward_hooks
                and isinstance(mp, nn.MaxPool2d) and (mp.kernel_size, mp.stride, mp.padding, mp.dilation, mp.ceil_mode) ==
                (3, 2, 1, 1, False) and y.shape[0] * y.shape[1] <= 65535)

    def forward(self, x):
        # the hand-written trunk (vitta_amd/trunk.py: every convolution is vitta_conv_f32) whenever the configuration
        # allows it; otherwise module by module (library convolutions + the fused BN passes)
        from . import trunk
        feat = trunk.run(self, x)
        if feat is not None:
            return self.fc(feat)
        y = self.conv1(x)
        if self._stem_fusable(y):
            from . import ops
            bn = self.bn1
            x = ops.FusedStemPool.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
        else:
            x = self.maxpool(bn_act(self.bn1, y, relu=True, act=self.relu))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = torch.flatten(self.avgpool(x), 1)
        return self.fc(x)


def resnet50(pretrained=False, **kw):
    """`pretrained` is accepted for call-site compatibility (tanet.py:129 passes True); ImageNet
    weights are never downloaded -- TTA always loads a full checkpoint afterwards."""
    return ResNet(Bottleneck, (3, 4, 6, 3), **kw)
