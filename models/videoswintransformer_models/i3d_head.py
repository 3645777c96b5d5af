"""models/videoswintransformer_models/i3d_head.py of the reference (:10-73) -> vitta_amd.swin.I3DHead."""
from vitta_amd.swin import I3DHead  # noqa: F401
