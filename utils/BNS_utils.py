"""utils/BNS_utils.py of the reference -> vitta_amd.bns_utils."""
from vitta_amd.bns_utils import BNFeatureHook, choose_layers, collect_bn_params, freeze_except_bn  # noqa: F401
