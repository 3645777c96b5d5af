"""Layer selection and affine-parameter collection for ViTTA.

Interface mirror of utils/BNS_utils.py: choose_layers :245-259, freeze_except_bn :262-276,
collect_bn_params :278-288.  The order of `choose_layers` (== model.named_modules() order) is
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
