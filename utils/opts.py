"""utils/opts.py of the reference -> vitta_amd.opts (same flags, same defaults, get_opts())."""
from vitta_amd.opts import *  # noqa: F401,F403
from vitta_amd.opts import build_parser, get_opts, img_norm_cfg, input_mean, input_std, parser  # noqa: F401
