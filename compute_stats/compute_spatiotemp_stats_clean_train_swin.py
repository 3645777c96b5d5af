"""Source statistics of Video Swin-B on the clean training videos -> list_spatiotemp_{mean,var}_<time>.npy."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitta_amd.main_eval import eval as run_eval  # noqa: E402
from vitta_amd.scripts import compute_stats, swin_ucf101_args  # noqa: E402

if __name__ == "__main__":
    args = compute_stats(swin_ucf101_args())
    run_eval(args=args)
