"""Source-only evaluation of TANet-R50 on UCF101-C (no adaptation; runs on the host CPU when no GPU
is visible -- BASELINE config 0)."""
from vitta_amd.scripts import run_over_corruptions, source_only, tanet_ucf101_args

if __name__ == "__main__":
    run_over_corruptions(source_only(tanet_ucf101_args()))
