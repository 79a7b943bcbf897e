"""Mirror of libreasr/lib/utils.py (inference subset)."""
import numpy as np
import torch


def tensorize(x):
    """bytes of little-endian float32 PCM -> float32 tensor [1, N]  (libreasr/lib/utils.py:149-153)."""
    arr = np.frombuffer(x, dtype=np.float32)
    arr = np.copy(arr)
    return torch.from_numpy(arr)[None]
