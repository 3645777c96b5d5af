"""models/videoswintransformer_models/swin_transformer.py of the reference -> vitta_amd.swin."""
from vitta_amd.swin import (BasicLayer, Mlp, PatchEmbed3D, PatchMerging, SwinTransformer3D,  # noqa: F401
                            SwinTransformerBlock3D, WindowAttention3D, compute_mask, get_window_size,
                            window_partition, window_reverse)
