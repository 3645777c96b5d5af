"""corpus/basics.py of the reference -> vitta_amd.tta (the functions on the ViTTA path)."""
from vitta_amd.tta import (ViTTAAdapter, compute_statistics, get_dataset_tanet, get_dataset_videoswin,  # noqa: F401
                           get_model, test_time_adapt, tta_standard, validate, validate_brief)
