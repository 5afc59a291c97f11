"""muzero_amd -- MI355X-native batched reversi env + batched MCTS behind the reference's
Python-facing API (ZiyuanMa/MuZero).  HIP kernels + C ABI: muzero_amd/csrc, include/muzero_amd.h."""
from . import config  # noqa: F401
