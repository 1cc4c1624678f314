"""
Explicit diffusion on the HIP backend (reference: phi/physics/diffuse.py:13-60).
"""
import warnings

import torch

from . import autodiff
from .extrapolation import ConstantExtrapolation, resolve
from .field import Field, _ptrs


def explicit(u: Field, diffusivity: float, dt: float, substeps: int = 1, order: int = 2) -> Field:
    """ Simulate a finite-time diffusion process of the form dF/dt = α · ΔF with explicit Euler steps (`diffuse.explicit`,
    order 2) on a StaggeredGrid or a CenteredGrid. The field's own extrapolation pads the stencil (tangential wall values of a
    velocity matter). Differentiable (adjoint stencil kernels). """
    from .field import require_plain
    require_plain(u, 'diffuse.explicit')
    if order != 2:
        raise NotImplementedError("HIP backend: diffuse.explicit implements order=2 only")
    amount = diffusivity * dt
    # CFL warning of the reference (diffuse.py:49-54)
    ratio = max(amount / substeps / (h * h) for h in u.dx)
    if ratio > 0.5:
        warnings.warn(f"CFL condition violated in diffuse.explicit: dt*diffusivity/dx^2 = {ratio:.3f} > 0.5, consider more substeps",
                      RuntimeWarning)
    be = u.backend
    kdt = amount / substeps
    if u.is_staggered:
        cur = [t.contiguous() for t in u.values]
        tracked = autodiff.needs_grad(*cur)
        for _ in range(substeps):
            if tracked:
                cur = list(autodiff.DiffuseStaggered.apply(dict(be=be, grid=u.grid_struct(), kdt=kdt, dtype=u.dtype), *cur))
            else:
                out = [torch.empty_like(t) for t in cur]
                be.ctx.diffuse_explicit(u.grid_struct(), _ptrs(cur), _ptrs(out), kdt, be.stream())
                cur = out
        return u.with_values(cur)
    s_codes, s_vals = resolve(u.boundary, u.dims)
    s_val = [[s_vals[a][s][0] if isinstance(u.boundary.side(d, bool(s)), ConstantExtrapolation) else 0.0 for s in range(2)]
             for a, d in enumerate(u.dims)]
    # the grid descriptor only carries the cell grid here; periodicity must match the scalar's
    from . import _capi
    from .field import _torch_dtype_code
    grid = _capi.make_grid(u.spatial_rank, _torch_dtype_code(u.dtype), u.batch_size, list(u.resolution.values()), u.bounds.lower, u.bounds.upper,
                           [[0 if c == 0 else 2 for c in pair] for pair in s_codes])
    cur = u.values.contiguous()
    tracked = autodiff.needs_grad(cur)
    for _ in range(substeps):
        if tracked:
            cur = autodiff.DiffuseCentered.apply(dict(be=be, grid=grid, kdt=kdt, s_codes=s_codes, s_val=s_val), cur)
        else:
            out = torch.empty_like(cur)
            be.ctx.diffuse_explicit_centered(grid, cur.data_ptr(), s_codes, s_val, out.data_ptr(), kdt, False, be.stream())
            cur = out
    return u.with_values(cur)
