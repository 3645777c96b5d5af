"""utils/norm_stats_utils.py of the reference -> vitta_amd.norm_stats (HIP kernels behind the hooks)."""
from vitta_amd.norm_stats import (CombineNormStatsRegHook_onereg, ComputeNormStatsHook, StatAlignEngine,  # noqa: F401
                                  compute_kld, compute_regularization)
