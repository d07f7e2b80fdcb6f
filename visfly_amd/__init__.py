"""visfly_amd -- MI355X-native batched quadrotor dynamics + RL-step engine.

Drop-in for the hot path of SJTU-ViSYS-team/VisFly (``Dynamics``, ``HoverEnv`` /
``NavigationEnv`` / ``RacingEnv`` with ``visual=False``, the PPO inner loop) backed by
hand-written HIP kernels for gfx950 behind a C-ABI (include/visfly_amd.h).
"""
from .constants import derive_constants  # noqa: F401
from .dynamics import Dynamics  # noqa: F401

__version__ = "0.1.0"
