"""
Differentiable fluid step. PhiFlow obtains gradients from its backend's autodiff
(reference: tests/commit/physics/test_fluid.py:55-73 `math.jacobian(sim)`, tests/commit/test_colab_fluids_tutorial.py:11-34
`field.functional_gradient(simulate, wrt=[0])`; the linear solve is differentiated implicitly by phiml's `solve_linear`).
Here every HIP forward kernel has a hand-written adjoint kernel in libphihip (csrc/adjoint.hip); this module only connects
them to `torch.autograd` so that the elementwise glue between the kernels (field arithmetic on device tensors) and user losses
differentiate too. torch is the tape, not the arithmetic.

Differentiable: `advect.semi_lagrangian` (staggered + centred), `resample` centred -> staggered (buoyancy),
`advect.mac_cormack` (staggered + centred; a clamped sample passes its gradient to the extremal tap), `diffuse.explicit`
(staggered + centred), `fluid.make_incompressible` (incl. obstacles, pressure output), `fluid.apply_boundary_conditions`, field
arithmetic.
"""
from typing import Callable, List, Sequence

import torch

from . import _capi


def needs_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _ptrs(ts):
    return [t.data_ptr() for t in ts]


def _contig(g, like):
    return torch.zeros_like(like) if g is None else g.contiguous()


class SemiLagrangianStaggered(torch.autograd.Function):
    """ out = semi_lagrangian(field, velocity, dt); inputs: D field components followed by D velocity components """

    @staticmethod
    def forward(ctx, meta, *tensors):
        D = len(tensors) // 2
        f, v = [t.contiguous() for t in tensors[:D]], [t.contiguous() for t in tensors[D:]]
        out = [torch.empty_like(t) for t in f]
        meta['be'].ctx.advect_staggered(meta['grid'], _ptrs(f), _ptrs(v), _ptrs(out), meta['dt'], meta['be'].stream())
        ctx.meta = meta
        ctx.save_for_backward(*f, *v)
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        meta = ctx.meta
        saved = ctx.saved_tensors
        D = len(saved) // 2
        f, v = list(saved[:D]), list(saved[D:])
        g = [_contig(gi, fi) for gi, fi in zip(grads, f)]
        gf = [torch.zeros_like(t) for t in f]
        gv = [torch.zeros_like(t) for t in v]
        meta['be'].ctx.advect_staggered_backward(meta['grid'], _ptrs(f), _ptrs(v), _ptrs(g), meta['dt'], _ptrs(gf), _ptrs(gv), meta['be'].stream())
        return (None, *gf, *gv)


class SemiLagrangianCentered(torch.autograd.Function):
    """ out = semi_lagrangian(scalar, velocity, dt); inputs: scalar, D velocity components """

    @staticmethod
    def forward(ctx, meta, s, *vel):
        s = s.contiguous()
        v = [t.contiguous() for t in vel]
        out = torch.empty_like(s)
        be = meta['be']
        be.ctx.advect_centered(meta['grid'], s.data_ptr(), meta['s_codes'], meta['s_val'], _ptrs(v), out.data_ptr(), meta['dt'], be.stream())
        ctx.meta = meta
        ctx.save_for_backward(s, *v)
        return out

    @staticmethod
    def backward(ctx, grad):
        meta = ctx.meta
        s, *v = ctx.saved_tensors
        be = meta['be']
        g = _contig(grad, s)
        gs = torch.zeros_like(s)
        gv = [torch.zeros_like(t) for t in v]
        be.ctx.advect_centered_backward(meta['grid'], s.data_ptr(), meta['s_codes'], meta['s_val'], _ptrs(v), g.data_ptr(), meta['dt'],
                                        gs.data_ptr(), _ptrs(gv), be.stream())
        return (None, gs, *gv)


class MacCormackStaggered(torch.autograd.Function):
    """ out = mac_cormack(field, velocity, dt, strength); inputs like SemiLagrangianStaggered """

    @staticmethod
    def forward(ctx, meta, *tensors):
        D = len(tensors) // 2
        f, v = [t.contiguous() for t in tensors[:D]], [t.contiguous() for t in tensors[D:]]
        out = [torch.empty_like(t) for t in f]
        meta['be'].ctx.mac_cormack_staggered(meta['grid'], _ptrs(f), _ptrs(v), _ptrs(out), meta['dt'], meta['strength'], meta['be'].stream())
        ctx.meta = meta
        ctx.save_for_backward(*f, *v)
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        meta = ctx.meta
        saved = ctx.saved_tensors
        D = len(saved) // 2
        f, v = list(saved[:D]), list(saved[D:])
        g = [_contig(gi, fi) for gi, fi in zip(grads, f)]
        gf = [torch.zeros_like(t) for t in f]
        gv = [torch.zeros_like(t) for t in v]
        meta['be'].ctx.mac_cormack_staggered_backward(meta['grid'], _ptrs(f), _ptrs(v), _ptrs(g), meta['dt'], meta['strength'], _ptrs(gf), _ptrs(gv),
                                                      meta['be'].stream())
        return (None, *gf, *gv)


class MacCormackCentered(torch.autograd.Function):
    @staticmethod
    def forward(ctx, meta, s, *vel):
        s = s.contiguous()
        v = [t.contiguous() for t in vel]
        out = torch.empty_like(s)
        be = meta['be']
        be.ctx.mac_cormack_centered(meta['grid'], s.data_ptr(), meta['s_codes'], meta['s_val'], _ptrs(v), out.data_ptr(), meta['dt'],
                                    meta['strength'], be.stream())
        ctx.meta = meta
        ctx.save_for_backward(s, *v)
        return out

    @staticmethod
    def backward(ctx, grad):
        meta = ctx.meta
        s, *v = ctx.saved_tensors
        be = meta['be']
        g = _contig(grad, s)
        gs = torch.zeros_like(s)
        gv = [torch.zeros_like(t) for t in v]
        be.ctx.mac_cormack_centered_backward(meta['grid'], s.data_ptr(), meta['s_codes'], meta['s_val'], _ptrs(v), g.data_ptr(), meta['dt'],
                                             meta['strength'], gs.data_ptr(), _ptrs(gv), be.stream())
        return (None, gs, *gv)


class DiffuseStaggered(torch.autograd.Function):
    """ one explicit diffusion sub-step of a staggered field: out = v + k dt laplace(v) """

    @staticmethod
    def forward(ctx, meta, *vel):
        v = [t.contiguous() for t in vel]
        out = [torch.empty_like(t) for t in v]
        meta['be'].ctx.diffuse_explicit(meta['grid'], _ptrs(v), _ptrs(out), meta['kdt'], meta['be'].stream())
        ctx.meta = meta
        ctx.shapes = [tuple(t.shape) for t in v]
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        meta = ctx.meta
        be = meta['be']
        g = [gi.contiguous() if gi is not None else be.zeros(shape, meta['dtype']) for gi, shape in zip(grads, ctx.shapes)]
        gin = [torch.zeros_like(t) for t in g]
        be.ctx.diffuse_explicit_backward(meta['grid'], _ptrs(g), _ptrs(gin), meta['kdt'], be.stream())
        return (None, *gin)


def _homogeneous(grid):
    """ copy of a phihip_grid with zero wall values: the constants of an extrapolation do not depend on the inputs """
    import ctypes
    from . import _capi
    g = _capi.Grid.from_buffer_copy(grid)
    ctypes.memset(ctypes.addressof(g.bc_val), 0, ctypes.sizeof(g.bc_val))
    return g


class DiffuseImplicitStaggered(torch.autograd.Function):
    """ diffuse.implicit of a staggered field: out = (I - k dt L)^-1 v. The operator is symmetric, so the vector-Jacobian product is one
    more solve of the same system with the upstream gradient as right-hand side and homogeneous wall values (implicit-function gradient,
    like phiml's solve_linear backward); the gradient solve uses `solve.gradient_solve` (default: the same solve). """

    @staticmethod
    def forward(ctx, meta, *vel):
        v = [t.contiguous() for t in vel]
        out = [torch.empty_like(t) for t in v]
        meta['infos'] = meta['be'].ctx.diffuse_implicit(meta['grid'], _ptrs(v), _ptrs(out), meta['kdt'], meta['csolve'], meta['be'].stream())
        ctx.meta = meta
        ctx.shapes = [tuple(t.shape) for t in v]
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        meta = ctx.meta
        be = meta['be']
        g = [gi.contiguous() if gi is not None else be.zeros(shape, meta['dtype']) for gi, shape in zip(grads, ctx.shapes)]
        gin = [torch.empty_like(t) for t in g]
        be.ctx.diffuse_implicit(_homogeneous(meta['grid']), _ptrs(g), _ptrs(gin), meta['kdt'], meta['csolve_bwd'], be.stream())
        return (None, *gin)


class DiffuseImplicitCentered(torch.autograd.Function):
    @staticmethod
    def forward(ctx, meta, s):
        s = s.contiguous()
        out = torch.empty_like(s)
        be = meta['be']
        meta['infos'] = be.ctx.diffuse_implicit_centered(meta['grid'], s.data_ptr(), meta['s_codes'], meta['s_val'], out.data_ptr(), meta['kdt'],
                                                         meta['csolve'], be.stream())
        ctx.meta = meta
        return out

    @staticmethod
    def backward(ctx, grad):
        meta = ctx.meta
        be = meta['be']
        g = grad.contiguous()
        gin = torch.empty_like(g)
        zero = [[0.0, 0.0] for _ in meta['s_val']]
        be.ctx.diffuse_implicit_centered(meta['grid'], g.data_ptr(), meta['s_codes'], zero, gin.data_ptr(), meta['kdt'], meta['csolve_bwd'], be.stream())
        return None, gin


class DiffuseCentered(torch.autograd.Function):
    @staticmethod
    def forward(ctx, meta, s):
        s = s.contiguous()
        out = torch.empty_like(s)
        be = meta['be']
        be.ctx.diffuse_explicit_centered(meta['grid'], s.data_ptr(), meta['s_codes'], meta['s_val'], out.data_ptr(), meta['kdt'], False, be.stream())
        ctx.meta = meta
        return out

    @staticmethod
    def backward(ctx, grad):
        meta = ctx.meta
        be = meta['be']
        g = grad.contiguous()
        gin = torch.zeros_like(g)
        be.ctx.diffuse_explicit_centered(meta['grid'], g.data_ptr(), meta['s_codes'], meta['s_val'], gin.data_ptr(), meta['kdt'], True, be.stream())
        return None, gin


class CenteredToStaggered(torch.autograd.Function):
    """ resample(s * vector, to=velocity) """

    @staticmethod
    def forward(ctx, meta, s):
        s = s.contiguous()
        be = meta['be']
        comps = [be.empty(shape, s.dtype) for shape in meta['shapes']]
        be.ctx.centered_to_staggered(meta['grid'], s.data_ptr(), meta['s_codes'], meta['s_val'], meta['vector'], False, _ptrs(comps), be.stream())
        ctx.meta = meta
        ctx.s_shape = s.shape
        return tuple(comps)

    @staticmethod
    def backward(ctx, *grads):
        meta = ctx.meta
        be = meta['be']
        g = [_contig(gi, be.empty(shape, meta['dtype'])) if gi is not None else be.zeros(shape, meta['dtype'])
             for gi, shape in zip(grads, meta['shapes'])]
        gs = be.zeros(tuple(ctx.s_shape), meta['dtype'])
        be.ctx.centered_to_staggered_backward(meta['grid'], meta['s_codes'], meta['vector'], _ptrs(g), gs.data_ptr(), be.stream())
        return None, gs


class ApplyObstacles(torch.autograd.Function):
    """ fluid.apply_boundary_conditions: v = keep * v + mask * obstacle_velocity; backward: keep * g """

    @staticmethod
    def forward(ctx, meta, *vel):
        out = [t.clone() for t in vel]
        be = meta['be']
        be.ctx.apply_obstacles(meta['grid'], meta['obstacles'], meta['count'], _ptrs(out), be.stream())
        ctx.meta = meta
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        meta = ctx.meta
        be = meta['be']
        g = [gi.clone() if gi is not None else None for gi in grads]
        if any(gi is None for gi in g):
            g = [gi if gi is not None else be.zeros(shape, meta['dtype']) for gi, shape in zip(g, meta['shapes'])]
        be.ctx.apply_obstacles(meta['grid'], meta['obstacles_still'], meta['count'], _ptrs(g), be.stream())   # linear part only
        return (None, *g)


class MakeIncompressible(torch.autograd.Function):
    """ (v_out..., p) = project(v'...; x0); backward = implicit-function adjoint (one CG solve with the same operator) """

    @staticmethod
    def forward(ctx, meta, x0, *vel):
        be = meta['be']
        new_v = [t.clone() for t in vel]
        pressure = x0.clone()
        infos = be.ctx.make_incompressible(meta['grid'], _ptrs(new_v), None, meta['flags_ptr'], 1, meta['balance'], pressure.data_ptr(), 0,
                                           meta['csolve'], True, be.stream())
        meta['infos'] = infos
        ctx.meta = meta
        return (*new_v, pressure)

    @staticmethod
    def backward(ctx, *grads):
        meta = ctx.meta
        be = meta['be']
        *g_v, g_p = grads
        g_v = [gi.clone().contiguous() if gi is not None else be.zeros(shape, meta['dtype']) for gi, shape in zip(g_v, meta['shapes'])]
        gp_ptr = 0
        if g_p is not None:
            g_p = g_p.contiguous()
            gp_ptr = g_p.data_ptr()
        infos = be.ctx.make_incompressible_backward(meta['grid'], meta['flags_ptr'], 1, meta['balance'], _ptrs(g_v), gp_ptr, meta['csolve_bwd'],
                                                    True, be.stream())
        meta['infos_backward'] = infos
        return (None, None, *g_v)


class NotDifferentiable(torch.autograd.Function):
    """ marks ops whose adjoint kernel does not exist yet: forward passes through, backward raises """

    @staticmethod
    def forward(ctx, name, *tensors):
        ctx.name = name
        return tuple(t.view_as(t) for t in tensors)

    @staticmethod
    def backward(ctx, *grads):
        raise NotImplementedError(f"HIP backend: {ctx.name} has no backward kernel yet (SURVEY §8 f5)")


# ---------------------------------------------------------------------------------------------------------------------
# user-facing helpers mirroring phi.field / phiml.math
# ---------------------------------------------------------------------------------------------------------------------
def l2_loss(field) -> torch.Tensor:
    """ `field.l2_loss`: 0.5 * sum(values ** 2) per batch entry (summed to a scalar when not batched) """
    from .field import require_plain
    require_plain(field, 'l2_loss')
    vals = field.values if field.is_staggered else [field.values]
    per = sum((t.reshape(t.shape[0], -1) ** 2).sum(dim=1) for t in vals) * 0.5
    return per if field.batched else per[0]


def stop_gradient(field):
    vals = [t.detach() for t in field.values] if field.is_staggered else field.values.detach()
    return field.with_values(vals)


def functional_gradient(f: Callable, wrt: Sequence[int] = (0,), get_output: bool = True) -> Callable:
    """ `field.functional_gradient` / `math.jacobian`: returns a function computing the gradient of the FIRST output of `f`
    (a scalar or per-batch loss; summed over the batch like phiml does) w.r.t. the `Field` arguments listed in `wrt`.
    With `get_output=True` the function returns (outputs of f, gradients). """
    wrt = [wrt] if isinstance(wrt, int) else list(wrt)

    def grad_fn(*args):
        from .field import Field
        args = list(args)
        leaves = []
        for i in wrt:
            a = args[i]
            assert isinstance(a, Field), f"functional_gradient: argument {i} must be a Field"
            vals = [t.detach().clone().requires_grad_(True) for t in a.values] if a.is_staggered else a.values.detach().clone().requires_grad_(True)
            args[i] = a.with_values(vals)
            leaves.append(vals if a.is_staggered else [vals])
        with torch.enable_grad():
            out = f(*args)
            loss = out[0] if isinstance(out, (tuple, list)) else out
            flat = [t for group in leaves for t in group]
            grads = torch.autograd.grad(loss.sum(), flat, allow_unused=True)
        result, k = [], 0
        for i, group in zip(wrt, leaves):
            gs = [g if g is not None else torch.zeros_like(t) for g, t in zip(grads[k:k + len(group)], group)]
            k += len(group)
            a = args[i]
            result.append(a.with_values(gs if a.is_staggered else gs[0]))
        return (out, tuple(result)) if get_output else tuple(result)

    return grad_fn


jacobian = functional_gradient
gradient = functional_gradient
