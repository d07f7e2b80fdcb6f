"""Small value types of the env surface (reference: utils/type.py)."""
from typing import Any

import numpy as np
import torch as th


class TensorDict(dict):
    """dict of tensors with the helpers the algorithms call on observations
    (reference: utils/type.py:101-193)."""

    def detach(self):
        return TensorDict({k: v.detach() if hasattr(v, "detach") else v for k, v in self.items()})

    def clone(self):
        for k in self.keys():
            self[k] = self[k].clone()
        return self

    def __getitem__(self, key: Any) -> Any:
        if isinstance(key, str):
            return super().__getitem__(key)
        if isinstance(key, (int, slice)) or hasattr(key, "__iter__"):
            return TensorDict({k: th.atleast_2d(v[key]) for k, v in self.items()})
        raise TypeError("Invalid key type. Must be either str or int.")

    def __setitem__(self, key: Any, value: Any) -> None:
        """field assignment by name, or row assignment into every field (value: mapping field -> rows)"""
        if isinstance(key, str):
            dict.__setitem__(self, key, value)
            return
        if not isinstance(key, (int, slice, list, np.ndarray, th.Tensor)):
            raise TypeError(f"TensorDict index must be a field name or a row index, not {type(key).__name__}")
        for name, column in self.items():
            column[key] = value[name]

    def append(self, data):
        for k, v in data.items():
            self[k] = th.cat([self[k], v])

    def cpu(self):
        for k in self.keys():
            self[k] = self[k].cpu()
        return self

    def to(self, device):
        for k in self.keys():
            self[k] = self[k].to(device)
        return self

    def as_tensor(self, device=th.device("cpu")):
        return TensorDict({k: th.as_tensor(v, device=device) for k, v in self.items()})

    def numpy(self):
        for k in self.keys():
            self[k] = self[k].detach().cpu().numpy()
        return self

    def reshape(self, shape):
        for k in self.keys():
            self[k] = self[k].reshape(shape)
        return self

    @staticmethod
    def stack(items):
        return TensorDict({k: th.stack([x[k] for x in items]) for k in items[0].keys()})

    def __len__(self):
        lens = {len(v) for v in self.values()}
        assert len(lens) == 1
        return lens.pop()

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]
