"""models/tanet_models/temporal_module.py of the reference -> vitta_amd.tanet."""
from vitta_amd.tanet import TAM, TemporalBottleneck, make_temporal_modeling  # noqa: F401
