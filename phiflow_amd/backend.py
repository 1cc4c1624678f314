"""
The `HipBackend`: device, HIP stream and libphihip context used by every operator of this package.

Mirrors the way PhiFlow selects a compute backend (reference: `phi/torch/flow.py:31-32` sets the global default,
`with backend:` overrides it per block -- `tests/commit/physics/test_fluid.py:21-22`,
`backend.set_default_device('GPU')` -- `demos/Top_Opt/Top_Opt3D.py:190`).

PyTorch is used for device memory and streams only; all arithmetic of the hot path runs in libphihip's HIP kernels.
"""
import os
from typing import List, Optional

import torch

from . import _capi


class HipBackend:
    """ One libphihip context bound to one torch device. Context manager like `phiml.backend.Backend`. """

    name = "hip"

    def __init__(self, library: Optional[_capi.Library] = None, device: Optional[str] = None):
        self.library = library if library is not None else _capi.load_default_library()
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("phiflow_amd needs a HIP device (torch.cuda.is_available() is False) -- there is no CPU fallback")
            device = f"cuda:{int(os.environ.get('LOCAL_RANK', '0')) % max(1, torch.cuda.device_count())}"
        self.device = torch.device(device)
        index = self.device.index if self.device.type == "cuda" and self.device.index is not None else 0
        self.ctx = _capi.Context(self.library, index)

    # --- phiml Backend protocol subset used by PhiFlow user code ---
    def __enter__(self):
        _STACK.append(self)
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        _STACK.pop()
        return False

    def set_default_device(self, device) -> bool:
        """ 'GPU' or a torch device string; returns True on success (phiml semantics). """
        if isinstance(device, str) and device.upper() == "GPU":
            return self.device.type == "cuda"
        return torch.device(device) == self.device

    def supports(self, feature) -> bool:
        name = feature if isinstance(feature, str) else getattr(feature, "__name__", str(feature))
        return name in ("grid_sample", "linear_solve", "conjugate_gradient")

    # --- plumbing ---
    def stream(self) -> int:
        """ raw hipStream_t of torch's current stream on this device (0 = default stream) """
        if self.device.type != "cuda":
            return 0
        return int(torch.cuda.current_stream(self.device).cuda_stream)

    def zeros(self, shape, dtype) -> torch.Tensor:
        return torch.zeros(shape, dtype=dtype, device=self.device)

    def empty(self, shape, dtype) -> torch.Tensor:
        return torch.empty(shape, dtype=dtype, device=self.device)

    def as_tensor(self, array, dtype) -> torch.Tensor:
        return torch.as_tensor(array, dtype=dtype).to(self.device).contiguous()

    def __repr__(self):
        return f"hip[{self.device}]"


_STACK: List[HipBackend] = []
_GLOBAL: Optional[HipBackend] = None


def default_backend() -> HipBackend:
    """ innermost `with backend:` or the lazily created global HIP backend """
    global _GLOBAL
    if _STACK:
        return _STACK[-1]
    if _GLOBAL is None:
        _GLOBAL = HipBackend()
    return _GLOBAL


def set_global_default_backend(backend: HipBackend):
    global _GLOBAL
    _GLOBAL = backend


PRECISION = [32]


def set_global_precision(bits: int):
    """ phiml.math.set_global_precision: 32 or 64 (Taylor_Green.ipynb cell 3) """
    assert bits in (32, 64), "only fp32 / fp64 are supported"
    PRECISION[0] = bits


def get_precision() -> int:
    return PRECISION[0]


class precision:
    """ `with math.precision(64):` (tests/commit/test_poisson_solver.py:147) """

    def __init__(self, bits: int):
        assert bits in (32, 64)
        self.bits = bits

    def __enter__(self):
        PRECISION.append(self.bits)

    def __exit__(self, *a):
        PRECISION.pop()
        return False


def float_dtype() -> torch.dtype:
    return torch.float64 if PRECISION[-1] == 64 else torch.float32
