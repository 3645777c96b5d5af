"""utils/pred_consistency_utils.py of the reference -> vitta_amd.pred_consistency."""
from vitta_amd.pred_consistency import compute_pred_consis  # noqa: F401
