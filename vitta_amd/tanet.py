"""TANet (TSN wrapper + Temporal Adaptive Module on ResNet-50), MI355X-native restatement.

Mirrors the interface and the state_dict / named_modules layout of
    models/tanet_models/tanet.py:16-333        (class TSN; RGB, resnet50, consensus 'avg', tam=True)
    models/tanet_models/temporal_module.py:12-134 (TAM, TemporalBottleneck, make_temporal_modeling)
    models/tanet_models/basic_ops.py:38-51     (segment average consensus)
so reference checkpoints (`module.base_model.layer1.0.net.conv1.weight`, `...tam.G.0.weight`,
`module.new_fc.weight`) load unchanged and choose_layers() sees the same 85 BatchNorm layers in the
same order (net.* before tam.*, blocks before downsample).

What is different (MI355X-first): the memory-bound TAM tail -- two permute+contiguous copies, a
broadcast multiply and a grouped conv in the reference (temporal_module.py:47-63, ~40 B/element) --
is two HIP launches on the native [N*T, C, H*W] layout (vitta_tam_pool / vitta_tam_agg, 4 + 8
B/element) with analytic backward kernels.  CPU tensors (BASELINE config 0, source-only evaluation
on the host) take a plain torch formulation of the same arithmetic.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import resnet as _resnet
from .fused_bn import bn_act, identity_source


FUSED_TAM_BRANCHES = True  # tests flip this to compare the fused G/L kernels with the torch modules


def _noop_hooks_only(module):
    """True if every forward hook on a BatchNorm1d of a TAM is a ViTTA statistics hook, which contributes
    exactly 0 there (norm_stats_utils.py:158-162) -- the only hooks the fused branch kernel may skip."""
    for fn in module._forward_hooks.values():
        owner = getattr(fn, "__self__", None)
        if owner is None or getattr(owner, "kind", None) != "bn1d":
            return False
    return not module._forward_pre_hooks


def _tam_aggregate_torch(x, gate, kern, t):
    """out[n,t] = K0 g[t-1] x[t-1] + K1 g[t] x[t] + K2 g[t+1] x[t+1] with zero padding in T."""
    nt, c, h, w = x.shape
    n = nt // t
    y = x.view(n, t, c, h * w) * gate.permute(0, 2, 1).unsqueeze(-1)  # (n,t,c,hw)
    k = kern.view(n, 1, c, 3, 1)
    yp = F.pad(y, (0, 0, 0, 0, 1, 1))  # pad T by one on both sides
    out = k[..., 0, :] * yp[:, 0:t] + k[..., 1, :] * yp[:, 1:t + 1] + k[..., 2, :] * yp[:, 2:t + 2]
    return out.reshape(nt, c, h, w)


class TAM(nn.Module):
    """Temporal adaptive module (temporal_module.py:12-65): G = global branch producing a per-(n,c)
    3-tap temporal kernel, L = local branch producing a per-(n,c,t) sigmoid gate."""

    def __init__(self, in_channels, n_segment, kernel_size=3, stride=1, padding=1):
        super().__init__()
        if kernel_size != 3 or stride != 1 or padding != 1:
            raise NotImplementedError("TAM is built with kernel 3 / stride 1 / padding 1 (tanet.py:134-138)")
        self.in_channels, self.n_segment = in_channels, n_segment
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.G = nn.Sequential(
            nn.Linear(n_segment, n_segment * 2, bias=False), nn.BatchNorm1d(n_segment * 2), nn.ReLU(inplace=True),
            nn.Linear(n_segment * 2, kernel_size, bias=False), nn.Softmax(-1))
        self.L = nn.Sequential(
            nn.Conv1d(in_channels, in_channels // 4, kernel_size, stride=1, padding=kernel_size // 2, bias=False),
            nn.BatchNorm1d(in_channels // 4), nn.ReLU(inplace=True),
            nn.Conv1d(in_channels // 4, in_channels, 1, bias=False), nn.Sigmoid())

    def forward(self, x):
        nt, c, h, w = x.shape
        t = self.n_segment
        n = nt // t
        bg, bl = self.G[1], self.L[1]
        if x.is_cuda:
            from . import ops
            if (FUSED_TAM_BRANCHES and not bg.training and not bl.training and _noop_hooks_only(bg)
                    and _noop_hooks_only(bl) and ops.tam_branch_supported(c, t)):
                return ops.TamFused.apply(x, t, self.G[0].weight, bg.weight, bg.bias, self.G[3].weight, self.L[0].weight,
                                          bl.weight, bl.bias, self.L[3].weight, bg.running_mean, bg.running_var, bg.eps,
                                          bl.running_mean, bl.running_var, bl.eps)
            pooled = ops.TamPool.apply(x, t)  # (n, c, t)
        else:
            pooled = x.view(n, t, c, h * w).mean(-1).permute(0, 2, 1).contiguous()
        kern = self.G(pooled.reshape(n * c, t))  # (n*c, 3)
        gate = self.L(pooled)  # (n, c, t)
        if x.is_cuda:
            from . import ops
            return ops.TamAggregate.apply(x, gate, kern, t)
        return _tam_aggregate_torch(x, gate, kern, t)


class TemporalBottleneck(nn.Module):
    """Bottleneck with a TAM after conv1/bn1/relu (temporal_module.py:68-106)."""

    def __init__(self, net, n_segment=8, t_kernel_size=3, t_stride=1, t_padding=1):
        super().__init__()
        if not isinstance(net, _resnet.Bottleneck):
            raise TypeError("TemporalBottleneck wraps a ResNet Bottleneck")
        self.net = net
        self.n_segment = n_segment
        self.tam = TAM(net.conv1.out_channels, n_segment, t_kernel_size, t_stride, t_padding)

    def forward(self, x):
        net = self.net
        src = identity_source(x)
        identity = src if net.downsample is None else _resnet.downsample_forward(net.downsample, src)
        out = bn_act(net.bn1, net.conv1(x), relu=True, act=net.relu)
        out = self.tam(out)
        out = bn_act(net.bn2, net.conv2(out), relu=True, act=net.relu)
        return bn_act(net.bn3, net.conv3(out), residual=identity, relu=True, act=net.relu, fork=True)


def make_temporal_modeling(net, n_segment=8, t_kernel_size=3, t_stride=1, t_padding=1):
    """Wrap every Bottleneck of layer1..4 (temporal_module.py:109-140; n_round == 1)."""
    if not isinstance(net, _resnet.ResNet):
        raise TypeError("make_temporal_modeling expects the ResNet trunk")
    for name in ("layer1", "layer2", "layer3", "layer4"):
        blocks = [TemporalBottleneck(b, n_segment, t_kernel_size, t_stride, t_padding)
                  for b in getattr(net, name).children()]
        setattr(net, name, nn.Sequential(*blocks))


class ConsensusModule(nn.Module):
    """Segment consensus (basic_ops.py:69-86); 'avg' = mean over the T frame predictions."""

    def __init__(self, consensus_type="avg", dim=1):
        super().__init__()
        if consensus_type not in ("avg", "identity"):
            raise NotImplementedError(f"consensus {consensus_type} is outside the ViTTA path")
        self.consensus_type, self.dim = consensus_type, dim

    def forward(self, x):
        return x.mean(dim=self.dim, keepdim=True) if self.consensus_type == "avg" else x


class TSN(nn.Module):
    """TSN(num_class, num_segments, 'RGB', base_model='resnet50', consensus_type='avg', tam=True)
    as built by get_model (corpus/basics.py:1463-1474)."""

    def __init__(self, num_class, num_segments, modality="RGB", base_model="resnet50", new_length=None,
                 consensus_type="avg", before_softmax=True, dropout=0.8, img_feature_dim=256, crop_num=1,
                 partial_bn=True, print_spec=False, pretrain="imagenet", tam=False, fc_lr5=False, non_local=False):
        super().__init__()
        if modality != "RGB" or base_model != "resnet50" or non_local or not before_softmax:
            raise NotImplementedError("only the RGB / resnet50 / before-softmax TANet of the ViTTA path is built")
        self.modality, self.num_segments = modality, num_segments
        self.reshape, self.before_softmax, self.dropout = True, before_softmax, dropout
        self.crop_num, self.consensus_type, self.img_feature_dim = crop_num, consensus_type, img_feature_dim
        self.pretrain, self.tam, self.base_model_name = pretrain, tam, base_model
        self.fc_lr5, self.non_local = fc_lr5, non_local
        self.new_length = 1 if new_length is None else new_length
        self.input_size, self.input_mean, self.input_std = 224, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]

        self.base_model = _resnet.resnet50(True)
        if tam:
            make_temporal_modeling(self.base_model, num_segments)
        self.base_model.last_layer_name = "fc"
        self.base_model.avgpool = nn.AdaptiveAvgPool2d(1)
        feature_dim = self.base_model.fc.in_features
        if dropout == 0:
            self.base_model.fc = nn.Linear(feature_dim, num_class)
            self.new_fc = None
            head = self.base_model.fc
        else:
            self.base_model.fc = nn.Dropout(p=dropout)
            self.new_fc = nn.Linear(feature_dim, num_class)
            head = self.new_fc
        nn.init.normal_(head.weight, 0, 0.001)
        nn.init.constant_(head.bias, 0)
        self.consensus = ConsensusModule(consensus_type)
        self._enable_pbn = partial_bn

    def partialBN(self, enable):
        self._enable_pbn = enable

    def train(self, mode=True):
        """Like tanet.py:182-198 (freeze every BatchNorm2d but the first when partial_bn is on), but
        returns self so `.eval()` chains (the reference returns None)."""
        super().train(mode)
        if self._enable_pbn and mode:
            count = 0
            for m in self.base_model.modules():
                if isinstance(m, nn.BatchNorm2d):
                    count += 1
                    if count >= 2:
                        m.eval()
                        m.weight.requires_grad = False
                        m.bias.requires_grad = False
        return self

    def _head(self, base_out):
        if self.dropout > 0:
            from . import fused_bn, ops
            fc = self.new_fc
            if fused_bn.ENABLED and ops.head_linear_supported(base_out, fc):
                base_out = ops.HeadLinear.apply(base_out, fc.weight, fc.bias)  # head.hip
            else:
                base_out = fc(base_out)
        base_out = base_out.view((-1, self.num_segments) + base_out.size()[1:])
        return self.consensus(base_out).squeeze(1)

    def fused_head_ok(self):
        """The adaptation head can run as ops.TanetHead (dropout -> new_fc -> consensus -> view consistency -> view mean in two
        launches): the stock modules, no hooks on them, 0 < p < 1 (p = 1: nn.Dropout gives zeros and a zero gradient where the fused
        node's 1 / (1 - p) scale would give inf * 0 -- the module chain keeps that case)."""
        fc, lin = self.base_model.fc, self.new_fc
        return (self.tam and 0.0 < self.dropout < 1.0 and type(fc) is nn.Dropout and type(lin) is nn.Linear and self.consensus_type == "avg"
                and self.before_softmax and not fc._forward_hooks and not fc._forward_pre_hooks and not lin._forward_hooks
                and not lin._forward_pre_hooks and not self.consensus._forward_hooks and not self._forward_hooks
                and not self._forward_pre_hooks and not self.base_model._forward_hooks)

    def trunk_features(self, input, no_reshape=False):
        """Pooled per-frame features [frames, 2048] of the hand-written trunk BEFORE the dropout (`base_model.fc`), or None when the
        configuration needs the module path (vitta_amd.trunk.run)."""
        from . import trunk
        if not no_reshape:
            input = input.view((-1, 3 * self.new_length) + input.size()[-2:])
        return trunk.run(self.base_model, input)

    def forward(self, input, no_reshape=False):
        if not no_reshape:
            sample_len = 3 * self.new_length
            input = input.view((-1, sample_len) + input.size()[-2:])
        return self._head(self.base_model(input))

    @property
    def crop_size(self):
        return self.input_size

    @property
    def scale_size(self):
        return self.input_size * 256 // 224
