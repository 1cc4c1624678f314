"""
`-m "not gpu"`: the PhiML-side glue of SURVEY §8b EXECUTES -- `phiflow_amd/phiml_plugin.py` (phi.field.Field <-> phiflow_amd Field, the
drop-ins `install()` patches into phi.physics.fluid / phi.physics.advect, reference: phi/torch/flow.py:15-35, fluid.py:94-162,
advect.py:156-215) and `phiflow_amd.linear.make_phiml_backend()` (a registered PhiML `Backend` whose `linear_solve` / `grid_sample` reach
libphihip; phi/__init__.py:41-63). Real PhiML is used when importable; otherwise the in-tree test double tests/fake_phiml (it exposes
exactly the API these modules touch). Kernels: the emulation build (test infrastructure).
"""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
try:                                             # real PhiML / PhiFlow first
    import phiml  # noqa: F401
    import phi.field  # noqa: F401
    REAL = True
except Exception:
    for name in [m for m in sys.modules if m == 'phi' or m.startswith('phi.') or m == 'phiml' or m.startswith('phiml.')]:
        del sys.modules[name]
    sys.path.insert(0, os.path.join(HERE, "fake_phiml"))
    REAL = False

from oracle import phi_oracle as O                     # noqa: E402
from phiflow_amd import linear, phiml_plugin as plug   # noqa: E402
import phiflow_amd.flow as hip                         # noqa: E402

pytestmark = pytest.mark.skipif(REAL, reason="written against the in-tree test double; tests/test_reference_crosscheck.py covers real PhiML")


def _phi_fields(batch=(), seed=0):
    """ a staggered velocity + centred scalar as phi Fields (test double), closed box 12 x 10 on Box(x=24, y=30) """
    from phiml import math
    from phiml.math import extrapolation as e
    from phi.field import Field
    from phi.geom import Box, UniformGrid
    rng = np.random.default_rng(seed)
    res = math.spatial(x=12, y=10)
    geo = UniformGrid(res, Box(x=24.0, y=(5.0, 35.0)))
    bshape = math.batch(**{n: s for n, s in batch})
    bsz = tuple(s for _, s in batch)
    vx = math.tensor(torch.as_tensor(0.1 * rng.standard_normal(bsz + (11, 10)).astype(np.float32)), bshape & math.spatial(x=11, y=10))
    vy = math.tensor(torch.as_tensor(0.1 * rng.standard_normal(bsz + (12, 9)).astype(np.float32)), bshape & math.spatial(x=12, y=9))
    v = Field(geo, math.stack({'x': vx, 'y': vy}, math.dual(vector='x,y')), e.ZERO)
    s = Field(geo, math.tensor(torch.as_tensor(rng.random(bsz + (12, 10)).astype(np.float32)), bshape & res), e.BOUNDARY)
    return v, s


def test_field_conversion_round_trip(emu_backend):
    with emu_backend:
        for batch in ((), (('sim', 3),), (('a', 2), ('b', 2))):
            v, s = _phi_fields(batch)
            hv, b1 = plug.to_hip(v)
            hs, b2 = plug.to_hip(s)
            nb = int(np.prod([n for _, n in batch])) if batch else 1
            assert hv.is_staggered and hv.batch_size == nb and hv.dims == ('x', 'y') and hv.boundary == hip.ZERO
            assert tuple(hv.bounds.lower) == (0.0, 5.0) and tuple(hv.bounds.upper) == (24.0, 35.0)
            assert [tuple(t.shape[1:]) for t in hv.values] == [(11, 10), (12, 9)] and hs.boundary == hip.BOUNDARY
            back = plug.from_hip(hv, v, b1)
            for d in ('x', 'y'):
                a, b = v.values[{'~vector': d}], back.values[{'~vector': d}]
                assert a.shape.names == b.shape.names and torch.equal(a._native, b._native)
            sb = plug.from_hip(hs, s, b2)
            assert sb.values.shape.names == s.values.shape.names and torch.equal(sb.values._native, s.values._native)
            assert back.extrapolation is v.extrapolation and back.geometry is v.geometry


def test_extrapolation_and_obstacle_conversion(emu_backend):
    from phiml import math
    from phiml.math import extrapolation as e
    from phi.geom import Box, Sphere
    from phi.physics.fluid import Obstacle
    dims = ('x', 'y')
    assert plug._to_hip_extrapolation(e.PERIODIC, dims) == hip.PERIODIC and plug._to_hip_extrapolation(e.ZERO_GRADIENT, dims) == hip.BOUNDARY
    assert plug._to_hip_extrapolation(e.ConstantExtrapolation(0.5), dims) == hip.ConstantExtrapolation(0.5)
    lid = e.ConstantExtrapolation(math.tensor([1.0, 0.0], math.channel(vector='x,y')))
    mixed = plug._to_hip_extrapolation(e.combine_sides(x=e.BOUNDARY, y=(e.ZERO, lid)), dims)
    assert mixed == hip.combine_sides(x=hip.BOUNDARY, y=(hip.ZERO, hip.ConstantExtrapolation({'x': 1.0, 'y': 0.0})))
    ob = plug._to_hip_obstacle(Obstacle(Sphere(radius=2.5, x=10.0, y=12.0), velocity=math.tensor([0.5, 0.0], math.channel(vector='x,y')), angular_velocity=0.3))
    assert isinstance(ob.geometry, hip.Sphere) and ob.geometry.radius == 2.5 and tuple(ob.velocity) == (0.5, 0.0) and tuple(ob.angular_velocity)[0] == 0.3
    box = plug._to_hip_geometry(Box(x=(1.0, 4.0), y=(2.0, 3.0)))
    assert tuple(box.lower) == (1.0, 2.0) and tuple(box.upper) == (4.0, 3.0)


def test_installed_drop_ins_run_the_hip_path_and_fall_back(emu_backend):
    from phiml import math
    from phiml.math import Solve
    from phi.geom import Sphere
    from phi.physics import advect as ref_advect, fluid as ref_fluid
    assert plug.install() is True and ref_fluid.make_incompressible is plug.make_incompressible
    try:
        with emu_backend:
            v, s = _phi_fields((('sim', 2),), seed=3)
            v2, p = ref_fluid.make_incompressible(v, [Sphere(radius=3.0, x=12.0, y=20.0)], Solve('CG', 1e-5, 0))
            assert not ref_fluid.CALLS                                                   # the HIP path ran, not the "reference"
            hv, b = plug.to_hip(v)
            hv2, hp = hip.fluid.make_incompressible(hv, hip.Obstacle(hip.Sphere(3.0, x=12.0, y=20.0)), hip.Solve('CG', 1e-5, 0))
            for d, t in zip(('x', 'y'), hv2.values):
                assert torch.equal(v2.values[{'~vector': d}].native(['sim', 'x', 'y']), t)
            assert torch.equal(p.values.native(['sim', 'x', 'y']), hp.values) and p.values.shape.names == ('sim', 'x', 'y')
            assert type(p.extrapolation).__name__ == '_Boundary'                         # closed box -> Neumann pressure (fluid.py:264-274)
            dom = O.Domain((12, 10), (0, 5), (24, 35), ((O.CLOSED, O.CLOSED),) * 2)
            vo, po, _, _ = O.make_incompressible([t.numpy() for t in hv.values], dom, [O.SphereObstacle((12.0, 20.0), 3.0)], rtol=1e-5, atol=0.0)
            for t, ref in zip(hv2.values, vo):
                assert np.abs(t.numpy() - ref).max() <= 2e-4 * max(np.abs(ref).max(), 1e-30) + 1e-5
            # everything off the fast path goes to the original function
            ref_fluid.make_incompressible(v, (), Solve(), order=4)
            ref_fluid.make_incompressible(v, (), Solve('biCG-stab(2)'))
            assert ref_fluid.CALLS == [('make_incompressible', 4), ('make_incompressible', 2)]
            # advection drop-ins
            s2 = ref_advect.semi_lagrangian(s, v, 0.7)
            hs, bs = plug.to_hip(s)
            assert torch.equal(s2.values.native(['sim', 'x', 'y']), hip.advect.semi_lagrangian(hs, hv, 0.7).values)
            s3 = ref_advect.mac_cormack(s, v, 0.7, correction_strength=0.5)
            assert torch.equal(s3.values.native(['sim', 'x', 'y']), hip.advect.mac_cormack(hs, hv, 0.7, correction_strength=0.5).values)
            v3 = ref_advect.advect(v, v, 0.7)
            assert torch.equal(v3.values[{'~vector': 'y'}].native(['sim', 'x', 'y']), hip.advect.semi_lagrangian(hv, hv, 0.7).values[1])
            assert not ref_advect.CALLS
            ref_advect.semi_lagrangian(s, v, 0.7, integrator=ref_advect.rk4)              # other integrators: reference
            assert ref_advect.CALLS == ['semi_lagrangian']
    finally:
        plug.uninstall()
        ref_fluid.CALLS.clear(); ref_advect.CALLS.clear()
    assert ref_fluid.make_incompressible is not plug.make_incompressible and ref_advect.advect is not plug.advect


def test_registered_phiml_backend_solves_and_samples(emu_backend):
    """ phi/__init__.py:41-63: a backend object in phiml.backend.BACKENDS; its linear_solve receives PhiML's assembled matrix """
    from phiml import backend as pb
    with emu_backend:
        be = linear.make_phiml_backend()
        assert be in pb.BACKENDS and be.name == 'hip' and isinstance(be, type(pb.torch.TORCH))
        assert linear.make_phiml_backend() is not None and sum(b.name == 'hip' for b in pb.BACKENDS) == 1      # registered once
        dom = O.Domain((10, 12), (0, 0), (10, 12), ((O.CLOSED, O.CLOSED), (O.PERIODIC, O.PERIODIC)))
        A = O.laplace_csr(dom, np.float64)
        rng = np.random.default_rng(1)
        y = rng.standard_normal((2, 10, 12)).astype(np.float32)
        y -= y.mean(axis=(1, 2), keepdims=True)
        be.set_grid_resolution((10, 12))
        result = be.linear_solve('CG', A, torch.as_tensor(y.reshape(2, -1)), torch.zeros(2, 120), 1e-5, 0.0, 1000)
        assert bool(result.converged.all()) and result.method == 'HIP CG'
        xo, _ = O.cg(lambda q: O.masked_laplace(q, dom, None, None), y, np.zeros_like(y), 1e-5, 0.0, 1000)
        a, b = result.x.numpy().reshape(y.shape), xo
        a, b = a - a.mean(axis=(1, 2), keepdims=True), b - b.mean(axis=(1, 2), keepdims=True)
        assert np.linalg.norm(a - b) / np.linalg.norm(b) <= 2e-3
        x2 = be.conjugate_gradient(A, torch.as_tensor(y.reshape(2, -1)), torch.zeros(2, 120), 1e-5, 0.0, 1000)
        assert torch.equal(x2.x, result.x)
        import scipy.sparse as sp
        before = be.hip_stats['fallbacks']                                               # not a Laplacian -> the base backend's solver
        M = sp.random(120, 120, 0.1, format='csr', random_state=0)
        generic = be.linear_solve('CG', M @ M.T + sp.identity(120), torch.as_tensor(y.reshape(2, -1)), torch.zeros(2, 120), 1e-5, 0.0, 10)
        assert be.hip_stats['fallbacks'] == before + 1 and generic.method.startswith('generic')
        with pytest.raises(NotImplementedError):
            be.linear_solve('biCG-stab(2)', A, torch.as_tensor(y.reshape(2, -1)), torch.zeros(2, 120), 1e-5, 0.0, 10)
        grid = torch.as_tensor(rng.standard_normal((1, 6, 7, 2)).astype(np.float32))
        coords = torch.as_tensor((rng.random((1, 9, 2)) * 6).astype(np.float32))
        out = be.grid_sample(grid, coords, 'boundary')
        ref = O.grid_sample(grid.numpy()[..., 1], [coords.numpy()[..., 0], coords.numpy()[..., 1]], ((O.OPEN, O.OPEN),) * 2, ((0.0, 0.0),) * 2)
        np.testing.assert_allclose(out.numpy()[..., 1], ref, atol=2e-5)
        with pytest.raises(NotImplementedError):
            be.grid_sample(grid, coords, 'symmetric')
        pb.BACKENDS.remove(be)


def _level_b_case(res, bc, obstacles=()):
    dom = O.Domain(res, (0.0,) * len(res), tuple(float(n) * 0.5 for n in res), bc)
    hard = active = None
    if obstacles:
        import scipy.sparse as sp
        active, hard, _ = O.obstacle_masks(obstacles, dom, np.float64)
        N = int(np.prod(res))
        cols = []
        for k in range(N):
            e = np.zeros((1,) + tuple(res)); e.reshape(-1)[k] = 1.0
            cols.append(O.masked_laplace(e, dom, hard, active).reshape(-1))
        A = sp.csr_matrix(np.stack(cols, axis=1))
    else:
        A = O.laplace_csr(dom, np.float64)
    return dom, A, hard, active


def test_level_b_rank_deficient_solves_reach_the_hip_kernels(emu_backend):
    """ `with HIP: fluid.make_incompressible(v)` on a periodic / closed box (BASELINE configs[1]-[3]): phi/physics/fluid.py:145-148 adds
    `preprocess_y=_balance_divergence` and `rank_deficiency=1`, PhiML hands the backend the assembled matrix plus `matrix_offset`. The
    override must (i) solve on libphihip -- launch counters of the marching CG kernels, no fall-back --, (ii) agree with the generic CG on
    A + offset 1 1^T, which is what every other PhiML backend iterates on, (iii) recognise the matrix once per grid although PhiFlow
    re-traces it every step (`forget_traces=True`, fluid.py:165), (iv) leave unbalanced right-hand sides and open boxes to `super()`. """
    from phiml import backend as pb, math
    ctx = emu_backend.ctx
    ctx.set_small_grid_solver(False)                     # the marching kernels also at test sizes
    try:
        with emu_backend:
            be = linear.make_phiml_backend()
            rng = np.random.default_rng(11)
            cases = [((12, 16), ((O.PERIODIC, O.PERIODIC),) * 2, ()), ((12, 16), ((O.CLOSED, O.CLOSED),) * 2, ()),
                     ((6, 8, 12), ((O.CLOSED, O.CLOSED), (O.PERIODIC, O.PERIODIC), (O.CLOSED, O.CLOSED)), ()),
                     ((11, 9), ((O.CLOSED, O.CLOSED),) * 2, [O.BoxObstacle((1.5, 1.0), (3.0, 2.5))])]
            for res, bc, obstacles in cases:
                dom, A, hard, active = _level_b_case(res, bc, obstacles)
                N = A.shape[0]
                act = np.ones(res) if active is None else active[0]
                div = rng.standard_normal((2,) + tuple(res)) * act                      # unbalanced, zero inside the obstacle (fluid.py:139-140)
                balance = lambda d, a: d - a * (d.sum(axis=tuple(range(1, d.ndim)), keepdims=True) / a.sum())     # fluid._balance_divergence
                solve = math.Solve('CG', 1e-6, 0, max_iterations=2000).with_preprocessing(balance, act)
                solve = math.copy_with(solve, rank_deficiency=1)
                stats0 = dict(be.hip_stats)
                ctx.profile_enable(True); ctx.profile_read(reset=True)
                with be:
                    for step in range(3):                                                # three "time steps": a NEW matrix object each, like a re-trace
                        traced = A.copy()
                        if step == 0 and not obstacles:                                   # PhiML's torch backend hands over a torch sparse tensor
                            traced = torch.sparse_csr_tensor(torch.as_tensor(A.indptr), torch.as_tensor(A.indices), torch.as_tensor(A.data), size=A.shape)
                        result = math.solve_linear(traced, div.astype(np.float32), solve)
                prof = ctx.profile_read(reset=True); ctx.profile_enable(False)
                assert result.method == 'HIP CG' and bool(torch.as_tensor(result.converged).all())
                assert prof['cg_matvec_dot'][0] > 0 and prof['cg_update'][0] + prof['cg_update_r'][0] > 0, prof        # march_kernel launches
                assert be.hip_stats['fallbacks'] == stats0['fallbacks'] and be.hip_stats['hip_solves'] == stats0['hip_solves'] + 3
                assert be.hip_stats['offsets_dropped'] == stats0['offsets_dropped'] + 3
                # one recognition per distinct matrix STORAGE (torch CSR, SciPy CSR), hits afterwards
                assert be.hip_stats['cache_misses'] - stats0['cache_misses'] <= 2 and be.hip_stats['cache_hits'] - stats0['cache_hits'] >= 1
                y = balance(div, act)
                generic = pb.Backend.linear_solve(be, 'CG', A, y.reshape(2, N), np.zeros((2, N)), 1e-6, 0.0, 2000, None, -1.0 / N)
                a, b = result.x.numpy().reshape(2, N).astype(np.float64), generic.x
                assert np.linalg.norm(a - b) / np.linalg.norm(b) <= 2e-3, (res, bc)
                assert np.abs((a.reshape(y.shape) * act).sum(axis=tuple(range(1, y.ndim)))).max() <= 1e-3 * np.abs(a).sum()   # no null-space component
                # a start vector with a mean: the generic solver removes it, so does the override
                x0 = (rng.standard_normal((2, N)) + 3.0)
                r2 = be.linear_solve('CG', A, torch.as_tensor(y.reshape(2, N), dtype=torch.float64), torch.as_tensor(x0), 1e-9, 0.0, 4000, None, -1.0 / N)
                g2 = pb.Backend.linear_solve(be, 'CG', A, y.reshape(2, N), x0, 1e-9, 0.0, 4000, None, -1.0 / N)
                assert np.linalg.norm(r2.x.numpy() - g2.x) / np.linalg.norm(g2.x) <= 1e-6
                # an unbalanced right-hand side is NOT in the range: the offset matters -> generic path
                f0 = be.hip_stats['fallbacks']
                r3 = be.linear_solve('CG', A, torch.as_tensor(div.reshape(2, N), dtype=torch.float64) + 1.0, torch.zeros(2, N, dtype=torch.float64), 1e-6, 0.0, 50, None, -1.0 / N)
                assert be.hip_stats['fallbacks'] == f0 + 1 and r3.method.startswith('generic')
            # an open side makes the operator non-singular: an offset would change the system -> generic path
            dom, A, _, _ = _level_b_case((10, 12), ((O.CLOSED, O.OPEN), (O.CLOSED, O.CLOSED)))
            f0 = be.hip_stats['fallbacks']
            y = rng.standard_normal((1, 120)); y -= y.mean()
            r4 = be.linear_solve('CG', A, torch.as_tensor(y), torch.zeros(1, 120, dtype=torch.float64), 1e-6, 0.0, 500, None, -1.0 / 120)
            assert be.hip_stats['fallbacks'] == f0 + 1 and r4.method.startswith('generic')
            r5 = be.linear_solve('CG', A, torch.as_tensor(y), torch.zeros(1, 120, dtype=torch.float64), 1e-6, 0.0, 500)       # without offset: HIP
            assert r5.method == 'HIP CG' and be.hip_stats['fallbacks'] == f0 + 1
            pb.BACKENDS.remove(be)
    finally:
        ctx.set_small_grid_solver(True)


def test_level_b_implicit_diffusion_matrix_reaches_the_hip_kernels(emu_backend):
    """ diffuse.implicit under `with HIP:` (phi/physics/diffuse.py:86-92): solve_linear hands the backend the matrix of
    sharpen(x) = x - k dt laplace(x); recognised as identity * I + scale * L it runs on phihip_cg_solve_shifted (march kernels) """
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    from phiml import backend as pb
    with emu_backend:
        be = linear.make_phiml_backend()
        for res, codes in (((12, 10), ((O.CLOSED, O.OPEN), (O.PERIODIC, O.PERIODIC))), ((6, 8, 12), ((O.OPEN, O.OPEN), (O.CLOSED, O.CLOSED), (O.PERIODIC, O.PERIODIC)))):
            # field extrapolation: CLOSED code = constant 0 (zero ghost), OPEN code = zero-gradient; the matrix of sharpen from the oracle's stencil
            D, N = len(res), int(np.prod(res))
            dom = O.Domain(res, (0.0,) * D, tuple(0.5 * n for n in res), ((O.PERIODIC, O.PERIODIC),) * D)
            kdt = 0.9
            cols = []
            for k in range(N):
                e = np.zeros((1,) + tuple(res)); e.reshape(-1)[k] = 1.0
                cols.append(O.diffuse_explicit_centered(e, kdt, -1.0, dom, codes, [(0.0, 0.0)] * D).reshape(-1))
            M = sp.csr_matrix(np.stack(cols, axis=1))
            d = linear.recognise_shifted_laplace(M, res)
            assert abs(d['identity'] - 1.0) <= 1e-12 and d['scale'] == -1.0 and d['flags'] is None
            rng = np.random.default_rng(3)
            y = rng.standard_normal((2, N))
            ctx = be._hip_backend().ctx
            ctx.profile_enable(True); ctx.profile_read(reset=True)
            r = be.linear_solve('CG', M, torch.as_tensor(y), torch.as_tensor(y.copy()), 1e-10, 0.0, 500)
            prof = ctx.profile_read(reset=True); ctx.profile_enable(False)
            assert r.method == 'HIP CG' and bool(r.converged.all())
            launches = sum(prof[k][0] for k in ('cg_matvec_dot', 'cg_update', 'cg_update_r', 'cg_residual') if k in prof)      # (small grids: the one-launch form)
            assert launches >= 3 and be.hip_stats.get('shifted_solves', 0) >= 1, prof
            exact = np.stack([spla.spsolve(M.tocsc(), y[b]) for b in range(2)])
            assert np.linalg.norm(r.x.numpy() - exact) / np.linalg.norm(exact) <= 1e-8
        # a rank-deficiency offset makes no sense for a definite system: generic path
        before = be.hip_stats['fallbacks']
        be.linear_solve('CG', M, torch.as_tensor(y), torch.zeros(2, N, dtype=torch.float64), 1e-6, 0.0, 50, None, -1.0 / N)
        assert be.hip_stats['fallbacks'] == before + 1
        pb.BACKENDS.remove(be)


def test_resolution_is_read_off_the_matrix():
    for res in ((64, 64), (4096 // 64, 64), (16, 16, 16), (5, 7, 9), (30, 4), (3, 50, 3)):
        for bc in (((O.PERIODIC, O.PERIODIC),) * len(res), ((O.CLOSED, O.OPEN),) * len(res)):
            dom = O.Domain(res, (0.0,) * len(res), tuple(float(n) for n in res), bc)
            assert linear.infer_resolution(O.laplace_csr(dom, np.float64)) == tuple(res), (res, bc)
    import scipy.sparse as sp
    with pytest.raises(linear.NotALaplaceStencil):
        linear.infer_resolution(sp.identity(50, format='csr'))
    # N = 4096: the old cube / square-root guess said 16^3 for a 64 x 64 grid
    dom = O.Domain((64, 64), (0, 0), (64, 64), ((O.CLOSED, O.CLOSED),) * 2)
    d = linear.recognise_laplace_stencil(O.laplace_csr(dom, np.float64), linear.infer_resolution(O.laplace_csr(dom, np.float64)))
    assert d['bc'] == [(1, 1), (1, 1)]


def test_surface_checker_agrees_with_the_test_double():
    """ tools/check_phiml_surface.py lists the PhiML names / signatures the Level-B code relies on; the double must offer the same surface
    (so that a mismatch reported against a real PhiML is a statement about PhiML, not about the checker) """
    import subprocess
    root = os.path.dirname(HERE)
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "check_phiml_surface.py"), "--fake"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
