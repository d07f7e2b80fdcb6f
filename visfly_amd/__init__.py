"""visfly_amd -- MI355X-native batched quadrotor dynamics + RL-step engine.

Drop-in for the hot path of SJTU-ViSYS-team/VisFly (``Dynamics``, ``HoverEnv`` /
``NavigationEnv`` / ``RacingEnv`` with ``visual=False``, the PPO inner loop) backed by
hand-written HIP kernels for gfx950 behind a C-ABI (include/visfly_amd.h).
"""
import os as _os

# Kernel arguments in device memory: the step kernels take ~1.5 KB of constants by value and every wave reads them with scalar
# loads at its start -- from host memory that costs +6.7 us per launch on MI355X (18.3 vs 11.6 us at 65 536 agents,
# DESIGN.md 4).  ROCm 7 defaults to device kernargs on this GPU; set it explicitly in case the process environment says
# otherwise.  Only effective if it happens before the HIP runtime initialises (i.e. before the first `import torch`).
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from .constants import derive_constants  # noqa: F401
from .dynamics import Dynamics  # noqa: F401

__version__ = "0.1.0"
