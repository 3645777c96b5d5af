"""Layer selection, affine-parameter collection and the BN-statistics hook for ViTTA.

Interface mirror of utils/BNS_utils.py: BNFeatureHook :19-77 (`stat_reg='BNS'`), choose_layers :245-259,
freeze_except_bn :262-276, collect_bn_params :278-288.  The order of `choose_layers` (== model.named_modules() order) is
load-bearing: source statistics are matched to layers by POSITION (corpus/basics.py:490-498).
"""


def choose_layers(model, candidate_layers):
    """Ordered [(name, module)] of every sub-module that is an instance of a candidate class."""
    kinds = tuple(candidate_layers)
    return [(name, m) for name, m in model.named_modules() if isinstance(m, kinds)]


def freeze_except_bn(model, bn_condidiate_layers):
    """train() mode, every parameter frozen except those owned by the candidate norm layers."""
    kinds = tuple(bn_condidiate_layers)
    model.train()
    model.requires_grad_(False)
    for m in model.modules():
        if isinstance(m, kinds):
            m.requires_grad_(True)
    return model


def collect_bn_params(model, bn_candidate_layers):
    """(params, names) of the affine scale (`weight`) and shift (`bias`) of the candidate norm layers."""
    kinds = tuple(bn_candidate_layers)
    params, names = [], []
    for layer_name, m in model.named_modules():
        if not isinstance(m, kinds):
            continue
        for pname, p in m.named_parameters():
            if pname in ("weight", "bias"):
                params.append(p)
                names.append(f"{layer_name}.{pname}")
    return params, names


class BNFeatureHook:
    """`stat_reg='BNS'`: align the statistics of a BatchNorm layer's INPUT with the layer's own running
    statistics (the source model's BN statistics); zero-initialised EMA when `running_manner`.
    The reductions run on the HIP moment kernels (BN2d: over (N*T, H, W); BN3d: over (N, T, H, W);
    BatchNorm1d: over the rows of (N*C, T) or over (N, T) of (N, C, T)); backward is analytic.

    Deviation from the reference, on purpose: with `use_src_stat_in_reg` the reference keeps
    `module.running_mean.data` -- a live ALIAS (BNS_utils.py:33-34) -- so under --fix_BNS False its "source" statistics
    drift with the train-mode BN layer.  Here the source statistics are a snapshot taken at construction (what the
    name says); with frozen BN buffers (--fix_BNS True, the shipped default) the two are identical."""

    def __init__(self, module, reg_type="l2norm", running_manner=False, use_src_stat_in_reg=True, momentum=0.1,
                 backend=None):
        import torch
        from .norm_stats import HipBackend
        self.backend = backend or HipBackend()
        self.hook = module.register_forward_hook(self.hook_fn)
        self.reg_type, self.running_manner, self.use_src_stat_in_reg = reg_type, running_manner, use_src_stat_in_reg
        if use_src_stat_in_reg:
            self.source_mean = module.running_mean.data.clone()
            self.source_var = module.running_var.data.clone()
        if running_manner:
            self.mean = torch.zeros_like(module.running_mean)
            self.var = torch.zeros_like(module.running_var)
        self.momentum = momentum

    def hook_fn(self, module, input, output):
        import torch.nn as nn
        from .norm_stats import compute_regularization
        x = input[0]
        if isinstance(module, nn.BatchNorm1d):
            kind = "rows" if x.dim() == 2 else "nct"
        elif isinstance(module, nn.BatchNorm2d):
            kind = "bn2d"
        elif isinstance(module, nn.BatchNorm3d):
            kind = "bn3d"
        else:
            raise Exception(f"undefined module {module}")
        batch_mean, batch_var = self.backend.feature_moments(x, kind)
        if self.running_manner:
            self.mean = self.momentum * batch_mean + (1.0 - self.momentum) * self.mean.detach()
            self.var = self.momentum * batch_var + (1.0 - self.momentum) * self.var.detach()
        else:
            self.mean, self.var = batch_mean, batch_var
        self.mean_true = self.source_mean if self.use_src_stat_in_reg else module.running_mean.data
        self.var_true = self.source_var if self.use_src_stat_in_reg else module.running_var.data
        self.r_feature = compute_regularization(mean_true=self.mean_true, mean_pred=self.mean, var_true=self.var_true,
                                                var_pred=self.var, reg_type=self.reg_type)

    def add_hook_back(self, module):
        self.hook = module.register_forward_hook(self.hook_fn)

    def close(self):
        self.hook.remove()
