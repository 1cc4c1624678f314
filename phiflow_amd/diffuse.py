"""
Explicit and implicit diffusion on the HIP backend (reference: phi/physics/diffuse.py:13-92).
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


def implicit(field: Field, diffusivity: float, dt: float, solve=None, order: int = 2) -> Field:
    """ Implicit Euler diffusion (`diffuse.implicit`, phi/physics/diffuse.py:63-92; Heat_Flow.ipynb, Burgers.ipynb): solves
    `(1 - diffusivity * dt * laplace) u = field` with CG from `x0 = field` -- `solve_linear(sharpen, y=field, solve)` with
    `sharpen(x) = explicit(x, diffusivity, -dt)`. The solver is the matrix-free CG of the pressure path (same kernels, operator
    I - k dt L) on the field's own lattice and extrapolation; every component of a StaggeredGrid is solved separately (the
    operator does not couple them). `solve`: `Solve('CG' | 'CG-adaptive', rel_tol, abs_tol, max_iterations)`; raises
    `NotConverged` / `Diverged` like `solve_linear` unless suppressed. Differentiable w.r.t. the field: the operator is symmetric, the
    backward pass is one more solve (with `solve.gradient_solve` if given) of the same system with homogeneous boundary constants. """
    from .field import require_plain, _torch_dtype_code
    from .solve import Solve, SolveInfo
    from .jit import is_tracing
    traced = is_tracing()       # inside a jit_compile'd function: info = NULL, no host read-back, nothing raised (jit.py)
    from .fluid import _raise_if_failed
    from . import _capi
    require_plain(field, 'diffuse.implicit')
    if order != 2:
        raise NotImplementedError("HIP backend: diffuse.implicit implements order=2 only")
    solve = Solve('CG') if solve is None else solve
    if solve.x0 is not None:
        raise NotImplementedError("HIP backend: diffuse.implicit starts from x0 = field (the reference's default); pass solve.x0=None")
    vals = field.values if field.is_staggered else [field.values]
    tracked = autodiff.needs_grad(*vals)
    be = field.backend
    fp64 = field.dtype == torch.float64
    csolve = solve.to_c(fp64)
    if traced:
        csolve.check_every = 0
        if tracked:
            raise NotImplementedError("HIP backend: gradients through a jit_compile'd function are not implemented")
    csolve_bwd = (solve.gradient_solve or solve).to_c(fp64)
    kdt = float(diffusivity) * float(dt)
    if field.is_staggered:
        cur = [t.contiguous() for t in field.values]
        if tracked:
            meta = dict(be=be, grid=field.grid_struct(), kdt=kdt, dtype=field.dtype, csolve=csolve, csolve_bwd=csolve_bwd)
            out = list(autodiff.DiffuseImplicitStaggered.apply(meta, *cur))
            infos = meta['infos']
        else:
            out = [torch.empty_like(t) for t in cur]
            infos = be.ctx.diffuse_implicit(field.grid_struct(), _ptrs(cur), _ptrs(out), kdt, csolve, be.stream(), want_info=not traced)
        result = field.with_values(out)
    else:
        s_codes, s_vals = resolve(field.boundary, field.dims)
        s_val = [[s_vals[a][s][0] if isinstance(field.boundary.side(d, bool(s)), ConstantExtrapolation) else 0.0 for s in range(2)]
                 for a, d in enumerate(field.dims)]
        grid = _capi.make_grid(field.spatial_rank, _torch_dtype_code(field.dtype), field.batch_size, list(field.resolution.values()), field.bounds.lower,
                               field.bounds.upper, [[0 if c == 0 else 2 for c in pair] for pair in s_codes])
        cur = field.values.contiguous()
        if tracked:
            meta = dict(be=be, grid=grid, kdt=kdt, s_codes=s_codes, s_val=s_val, csolve=csolve, csolve_bwd=csolve_bwd)
            out = autodiff.DiffuseImplicitCentered.apply(meta, cur)
            infos = meta['infos']
        else:
            out = torch.empty_like(cur)
            infos = be.ctx.diffuse_implicit_centered(grid, cur.data_ptr(), s_codes, s_val, out.data_ptr(), kdt, csolve, be.stream(), want_info=not traced)
        result = field.with_values(out)
    if infos is not None:
        info = SolveInfo(solve, [i.iterations for i in infos], [i.residual_sq for i in infos], [i.rhs_sq for i in infos],
                         [bool(i.converged) for i in infos], [bool(i.diverged) for i in infos])
        _raise_if_failed(info)
        result.solve_info = info
    return result
