"""ViTTA on Video Swin-B / UCF101-C, all 12 corruptions (entry point of the reference kept by name)."""
from vitta_amd.scripts import run_over_corruptions, swin_ucf101_args

if __name__ == "__main__":
    args = swin_ucf101_args()
    run_over_corruptions(args)
