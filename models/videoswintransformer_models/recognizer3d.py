"""models/videoswintransformer_models/recognizer3d.py of the reference (:45-115) -> vitta_amd.swin.Recognizer3D."""
from vitta_amd.swin import Recognizer3D  # noqa: F401
