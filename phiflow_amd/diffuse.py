"""
Explicit diffusion on the HIP backend (reference: phi/physics/diffuse.py:13-60).
"""
import warnings

import torch

from . import autodiff
from .field import Field, _ptrs


def explicit(u: Field, diffusivity: float, dt: float, substeps: int = 1, order: int = 2) -> Field:
    """ Simulate a finite-time diffusion process of the form dF/dt = α · ΔF on a StaggeredGrid with explicit Euler steps
    (`diffuse.explicit`, order 2). The velocity's own extrapolation pads the stencil (tangential wall values matter). """
    if order != 2 or not u.is_staggered:
        raise NotImplementedError("HIP backend: diffuse.explicit implements StaggeredGrid, order=2 only")
    amount = diffusivity * dt
    # CFL warning of the reference (diffuse.py:49-54)
    ratio = max(amount / substeps / (h * h) for h in u.dx)
    if ratio > 0.5:
        warnings.warn(f"CFL condition violated in diffuse.explicit: dt*diffusivity/dx^2 = {ratio:.3f} > 0.5, consider more substeps",
                      RuntimeWarning)
    be = u.backend
    cur = [t.contiguous() for t in u.values]
    tracked = autodiff.needs_grad(*cur)
    if tracked:
        cur = list(autodiff.NotDifferentiable.apply("diffuse.explicit", *cur))
        anchor = sum(t.reshape(-1)[0] * 0 for t in cur)
    for _ in range(substeps):
        out = [torch.empty_like(t) for t in cur]
        be.ctx.diffuse_explicit(u.grid_struct(), _ptrs(cur), _ptrs(out), amount / substeps, be.stream())
        cur = out
    if tracked:
        cur = [t + anchor for t in cur]
    return u.with_values(cur)
