"""observation / action space descriptors: gymnasium's if it is installed, else a minimal stand-in
with the attributes the algorithms read (shape, dtype, low, high, spaces)."""
import numpy as np

try:  # pragma: no cover - not installed in the build image
    from gymnasium.spaces import Box, Dict  # type: ignore
except Exception:
    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

        def __repr__(self):
            return f"Box({self.low}, {self.high}, {self.shape}, {np.dtype(self.dtype).name})"

    class Dict:
        def __init__(self, spaces):
            self.spaces = dict(spaces)

        def __getitem__(self, k):
            return self.spaces[k]

        def __setitem__(self, k, v):
            self.spaces[k] = v

        def keys(self):
            return self.spaces.keys()

        def items(self):
            return self.spaces.items()

        def __repr__(self):
            return f"Dict({self.spaces})"
