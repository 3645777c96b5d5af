"""models/tanet_models/basic_ops.py of the reference -> vitta_amd.tanet.ConsensusModule."""
from vitta_amd.tanet import ConsensusModule  # noqa: F401
