"""models/tanet_models/tanet.py of the reference -> vitta_amd.tanet.TSN."""
from vitta_amd.tanet import TSN, ConsensusModule  # noqa: F401
