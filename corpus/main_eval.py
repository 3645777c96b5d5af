"""corpus/main_eval.py of the reference -> vitta_amd.main_eval.eval(args, model=None)."""
from vitta_amd.main_eval import eval, load_checkpoint_into, pick_device  # noqa: F401
