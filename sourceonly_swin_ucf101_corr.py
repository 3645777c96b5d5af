"""Source-only evaluation of Video Swin-B on UCF101-C (no adaptation)."""
from vitta_amd.scripts import run_over_corruptions, source_only, swin_ucf101_args

if __name__ == "__main__":
    run_over_corruptions(source_only(swin_ucf101_args()))
