"""
`-m gpu`: the real gfx950 library on a MI355X, driven through the C ABI (ctypes) and compared with the NumPy oracle on
the same seeded inputs, plus size-independent properties at BASELINE.json's full sizes.
"""
import math

import numpy as np
import pytest

import parity_cases as pc
from parity_cases import CLO, OPN, PER

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(gpu_backend):
    return gpu_backend.ctx


@pytest.fixture(scope="module")
def mem(gpu_backend):
    return pc.TorchMem(str(gpu_backend.device))


GRIDS = [
    ((16, 20), ((CLO, CLO), (CLO, CLO))),
    ((16, 20), ((OPN, OPN), (OPN, OPN))),
    ((16, 20), ((PER, PER), (PER, PER))),
    ((16, 20), ((OPN, OPN), (CLO, OPN))),
    ((7, 13), ((CLO, OPN), (PER, PER))),
    ((64, 128), ((CLO, CLO), (CLO, CLO))),
    ((8, 12, 16), ((PER, PER), (PER, PER), (PER, PER))),
    ((9, 7, 10), ((CLO, CLO), (OPN, OPN), (CLO, OPN))),
    ((6, 20, 72), ((CLO, OPN), (PER, PER), (CLO, CLO))),
    ((48, 40, 136), ((PER, PER), (CLO, CLO), (OPN, OPN))),
    ((33, 31, 29), ((CLO, CLO), (CLO, CLO), (CLO, CLO))),
    ((3, 5, 264), ((PER, PER), (CLO, CLO), (OPN, OPN))),
    ((4, 5, 24), ((OPN, OPN), (OPN, CLO), (OPN, CLO))),
    ((12, 10, 250), ((PER, PER), (CLO, OPN), (PER, PER))),  # fp32 rows of even length: the 8-byte-vector instantiation (V = 2)
    ((10, 9, 261), ((CLO, OPN), (PER, PER), (PER, PER))),   # r4: odd rows on the UNAL vector kernels (one full 256-cell tile + five cells; (33, 31, 29) above: all-closed odd rows)
    ((40, 36, 384), ((CLO, CLO), (CLO, CLO), (CLO, CLO))),   # config-5 rows: three (1,64) fp64 tiles per row, vector gradient kernel with n2 - 1 faces
]


@pytest.mark.parametrize("res,bc", GRIDS)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_stencils_match_oracle(ctx, mem, res, bc, dtype):
    rng = np.random.default_rng(1)
    dom, grid = pc.make_case(res, bc, dtype, batch=2)
    pc.check_component_shapes(ctx, dom, grid)
    pc.check_laplace(ctx, mem, dom, grid, dtype, rng)
    pc.check_divergence(ctx, mem, dom, grid, dtype, rng, balance=False)
    pc.check_divergence(ctx, mem, dom, grid, dtype, rng, balance=True)
    pc.check_divergence_flags(ctx, mem, dom, grid, dtype, rng)
    pc.check_grad_subtract(ctx, mem, dom, grid, dtype, rng)
    pc.check_grad_subtract_flags(ctx, mem, dom, grid, dtype, rng)
    pc.check_diffuse(ctx, mem, dom, grid, dtype, rng)


@pytest.mark.parametrize("res,bc", GRIDS[:10])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_implicit_diffusion_matches_oracle(ctx, mem, res, bc, dtype):
    """ diffuse.implicit (phi/physics/diffuse.py:63-92): the CG kernels of the pressure path with the operator I - k dt L on the field's
    lattice (staggered components with wall values, centred scalar with a constant side) vs the oracle's CG on `sharpen` """
    rng = np.random.default_rng(21)
    D = len(res)
    dom, grid = pc.make_case(res, bc, dtype, batch=2, bc_val=rng.uniform(-0.5, 0.5, (D, 2, D)))
    s_codes = tuple((PER, PER) if lo == PER else (OPN, CLO) for lo, hi in bc)
    pc.check_diffuse_implicit(ctx, mem, dom, grid, dtype, rng, s_codes, [(0.0, 0.25)] * D)


@pytest.mark.parametrize("res,bc", GRIDS)
def test_advection_matches_oracle(ctx, mem, res, bc):
    rng = np.random.default_rng(2)
    for dtype in (np.float32, np.float64):
        dom, grid = pc.make_case(res, bc, dtype, batch=2)
        pc.check_advect_staggered(ctx, mem, dom, grid, dtype, rng, dt=0.7)
        pc.check_advect_staggered(ctx, mem, dom, grid, dtype, rng, dt=2.9)
        s_codes = tuple((PER, PER) if lo == PER else (OPN, CLO) for lo, hi in bc)
        pc.check_advect_centered(ctx, mem, dom, grid, dtype, rng, s_codes, [(0.0, 0.25)] * len(res))


@pytest.mark.parametrize("res,bc", GRIDS)
def test_mac_cormack_and_resample_match_oracle(ctx, mem, res, bc):
    """ advect.mac_cormack (centred + staggered) and the centred -> staggered resample used for buoyancy """
    rng = np.random.default_rng(12)
    for dtype in (np.float32, np.float64):
        dom, grid = pc.make_case(res, bc, dtype, batch=2)
        s_codes = tuple((PER, PER) if lo == PER else (OPN, CLO) for lo, hi in bc)
        s_consts = [(0.0, 0.25)] * len(res)
        pc.check_mac_cormack_centered(ctx, mem, dom, grid, dtype, rng, s_codes, s_consts)
        pc.check_mac_cormack_centered(ctx, mem, dom, grid, dtype, rng, s_codes, s_consts, dt=2.3, strength=0.6)
        pc.check_mac_cormack_staggered(ctx, mem, dom, grid, dtype, rng)
        pc.check_centered_to_staggered(ctx, mem, dom, grid, dtype, rng, s_codes, s_consts)
        swapped = tuple((PER, PER) if lo == PER else (CLO, OPN) for lo, hi in bc)      # constant below, zero-gradient above
        pc.check_centered_to_staggered(ctx, mem, dom, grid, dtype, rng, swapped, [(0.4, 0.0)] * len(res))


def test_obstacle_rasterisation_and_moving_obstacles(ctx, mem):
    """ hard cell mask + apply_boundary_conditions with moving / rotating Box and Sphere obstacles on the device """
    rng = np.random.default_rng(13)
    for dtype in (np.float32, np.float64):
        dom, grid = pc.make_case((96, 80), ((CLO, CLO), (OPN, OPN)), dtype, batch=2)
        obstacles = [pc.O.BoxObstacle((14.0, 13.0), (40.0, 39.5), velocity=(0.5, -0.25), angular_velocity=0.3,
                                       rotation=[[np.cos(0.4), -np.sin(0.4)], [np.sin(0.4), np.cos(0.4)]]),
                     pc.O.SphereObstacle((39.0, 39.0), 13.0),
                     pc.O.SphereObstacle((67.0, 52.0), 12.5, angular_velocity=-1.0)]
        pc.check_obstacle_kernels(ctx, mem, dom, grid, dtype, rng, obstacles)
        dom, grid = pc.make_case((48, 40, 64), ((CLO, CLO), (PER, PER), (CLO, OPN)), dtype, batch=1)
        obstacles = [pc.O.BoxObstacle((14.0, 13.0, 15.0), (28.0, 27.0, 41.0), angular_velocity=(0.1, -0.2, 0.3)),
                     pc.O.SphereObstacle((26.0, 26.0, 39.0), 13.0, velocity=(1.0, 0.0, -1.0))]
        pc.check_obstacle_kernels(ctx, mem, dom, grid, dtype, rng, obstacles)
    dom, grid = pc.make_case((32, 32), ((CLO, CLO), (CLO, CLO)), np.float32, batch=1)
    many = [pc.O.SphereObstacle((2.0 + 1.5 * k, 3.0 + 1.2 * k), 1.0, velocity=(0.1 * k, 0.0)) for k in range(20)]
    pc.check_obstacle_kernels(ctx, mem, dom, grid, np.float32, rng, many)


@pytest.mark.parametrize("res,bc", GRIDS[:5] + GRIDS[6:9])
def test_adjoint_kernels_match_oracle_derivatives(ctx, mem, res, bc):
    """ backward kernels (atomics on the real GPU) vs finite differences / linear responses of the oracle (fp64) """
    rng = np.random.default_rng(14)
    dom, grid = pc.make_case(res, bc, np.float64, batch=2)
    s_codes = tuple((PER, PER) if lo == PER else (OPN, CLO) for lo, hi in bc)
    pc.check_advect_backward(ctx, mem, dom, grid, rng, s_codes, [(0.0, 0.25)] * len(res))
    pc.check_advect_backward(ctx, mem, dom, grid, rng, s_codes, [(0.0, 0.25)] * len(res), dt=0.2)      # CFL < 1 everywhere: every scatter goes through the LDS windows
    pc.check_project_backward(ctx, mem, dom, grid, rng)
    pc.check_mac_cormack_and_diffuse_backward(ctx, mem, dom, grid, rng, s_codes, [(0.0, 0.25)] * len(res))


def test_adjoint_next_to_a_lookup_kink(ctx, mem):
    """ round-3 fuzz seed 40062 (adjoint vs finite differences 3e-4 apart on the GPU) as a constructed case with the kink distance asserted """
    pc.check_adjoint_next_to_a_lookup_kink(ctx, mem)


def test_adjoint_projection_with_obstacles(ctx, mem):
    rng = np.random.default_rng(15)
    dom, grid = pc.make_case((12, 10, 16), ((CLO, CLO),) * 3, np.float64, batch=1)
    pc.check_project_backward(ctx, mem, dom, grid, rng, obstacles=[pc.O.BoxObstacle((4.0, 3.0, 5.0), (8.0, 7.0, 11.0))])
    dom, grid = pc.make_case((16, 20), ((PER, PER), (CLO, OPN)), np.float64, batch=2)
    pc.check_project_backward(ctx, mem, dom, grid, rng, obstacles=[pc.O.SphereObstacle((8.0, 9.0), 3.5)])


@pytest.mark.parametrize("res,bc", GRIDS[6:])
def test_slab_halo_planes(ctx, mem, res, bc):
    rng = np.random.default_rng(16)
    for dtype in (np.float32, np.float64):
        dom, _ = pc.make_case(res, bc, dtype, batch=1)
        pc.check_slab_halo_planes(ctx, mem, dom, dtype, rng, parts=2 if res[0] < 9 else 3)


def test_reference_known_answer_self_advection(ctx, mem):
    """ /root/reference tests/commit/physics/test_advect.py:41-45 -- the only stored known answer on the path """
    dom, grid = pc.make_case((4, 3), ((CLO, CLO), (CLO, CLO)), np.float32)
    vx = np.zeros((1, 3, 3), np.float32)
    vy = np.zeros((1, 4, 2), np.float32)
    vy[0, 1:3, :] = 1
    dv = [mem.to_dev(vx), mem.to_dev(vy)]
    out = [mem.empty(vx.shape, np.float32), mem.empty(vy.shape, np.float32)]
    ctx.advect_staggered(grid, [mem.ptr(a) for a in dv], [mem.ptr(a) for a in dv], [mem.ptr(a) for a in out], 1.0)
    mem.sync()
    np.testing.assert_allclose(mem.to_host(out[0]), 0, atol=1e-6)
    np.testing.assert_allclose(mem.to_host(out[1])[0].T, [[0, 0, 0, 0], [0, 1, 1, 0]], rtol=1e-5, atol=1e-6)


def test_identity_advection(ctx, mem):
    """ test_advect.py:12-18: adv(v, v, 0) == adv(v, 0, 1) == v """
    rng = np.random.default_rng(3)
    for res, bc in GRIDS[:4] + GRIDS[6:8]:
        dom, grid = pc.make_case(res, bc, np.float32)
        v = pc.random_velocity(dom, 1, np.float32, rng)
        dv = [mem.to_dev(a) for a in v]
        zero = [mem.to_dev(np.zeros_like(a)) for a in v]
        out = [mem.empty(a.shape, np.float32) for a in v]
        ctx.advect_staggered(grid, [mem.ptr(a) for a in dv], [mem.ptr(a) for a in dv], [mem.ptr(a) for a in out], 0.0)
        mem.sync()
        for a, b in zip(out, v):
            np.testing.assert_allclose(mem.to_host(a), b, atol=1e-5)
        ctx.advect_staggered(grid, [mem.ptr(a) for a in dv], [mem.ptr(a) for a in zero], [mem.ptr(a) for a in out], 1.0)
        mem.sync()
        for a, b in zip(out, v):
            np.testing.assert_allclose(mem.to_host(a), b, atol=1e-5)


def test_advection_with_wall_velocity(ctx, mem):
    rng = np.random.default_rng(3)
    bcv = np.zeros((3, 2, 3)); bcv[2, 1, 0] = 1.0; bcv[0, 0, 0] = 0.3
    for dtype in (np.float32, np.float64):
        dom, grid = pc.make_case((8, 8, 8), ((CLO, CLO),) * 3, dtype, batch=1, bc_val=bcv)
        pc.check_advect_staggered(ctx, mem, dom, grid, dtype, rng, dt=1.3)
        pc.check_divergence(ctx, mem, dom, grid, dtype, rng, balance=False)
        pc.check_diffuse(ctx, mem, dom, grid, dtype, rng)


@pytest.mark.parametrize("res,bc", GRIDS[:9])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_cg_matches_oracle(ctx, mem, res, bc, dtype):
    rng = np.random.default_rng(4)
    dom, grid = pc.make_case(res, bc, dtype, batch=2)
    try:
        for small in (True, False):     # single-kernel solver for small grids (cg_small.hip) and the marching kernels
            ctx.set_small_grid_solver(small)
            pc.check_cg(ctx, mem, dom, grid, dtype, np.random.default_rng(4))
    finally:
        ctx.set_small_grid_solver(True)


@pytest.mark.parametrize("res,bc", GRIDS[:9:2])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_cg_adaptive_matches_oracle(ctx, mem, res, bc, dtype):
    """ Solve('CG-adaptive') (Fluid_Logo.ipynb) on both solvers: tolerance mode, and fixed iterations with
    the true-residual refresh (stored-q branch of the marching path) """
    dom, grid = pc.make_case(res, bc, dtype, batch=2)
    try:
        for small in (True, False):
            ctx.set_small_grid_solver(small)
            pc.check_cg(ctx, mem, dom, grid, dtype, np.random.default_rng(4), refresh=20, adaptive=True)
            pc.check_cg(ctx, mem, dom, grid, dtype, np.random.default_rng(6), max_iter=12, refresh=5, fixed_iterations=True, adaptive=True)
    finally:
        ctx.set_small_grid_solver(True)


def test_cg_adaptive_64_cubed_fixed_iterations(ctx, mem):
    rng = np.random.default_rng(5)
    dom, grid = pc.make_case((64, 64, 64), ((PER, PER),) * 3, np.float32, upper=(2 * math.pi,) * 3)
    pc.check_cg(ctx, mem, dom, grid, np.float32, rng, max_iter=100, refresh=20, fixed_iterations=True, adaptive=True)


@pytest.mark.parametrize("res,bc", pc.DEGENERATE_GRIDS)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_degenerate_grids(ctx, mem, res, bc, dtype):
    """ one / two cells along an axis, 3-D grids with a single plane (whose a0 boundary rule still applies) """
    pc.check_degenerate_grid(ctx, mem, res, bc, dtype)


@pytest.mark.parametrize("shape,codes", [((7, 5), ((PER, PER), (CLO, OPN))), ((1, 9), ((CLO, CLO), (OPN, OPN))), ((4, 6, 5), ((OPN, CLO), (PER, PER), (CLO, CLO))),
                                         ((3, 1, 8), ((PER, PER), (OPN, OPN), (CLO, OPN)))])
def test_grid_sample_matches_oracle(ctx, mem, shape, codes):
    """ math.grid_sample (phi/field/_resample.py:257-259) at arbitrary coordinates, far outside the array included """
    rng = np.random.default_rng(31)
    consts = [(0.3, -1.2)] * len(shape)
    for dtype in (np.float32, np.float64):
        pc.check_grid_sample(ctx, mem, shape, codes, consts, dtype, rng)
        pc.check_grid_sample(ctx, mem, shape, codes, consts, dtype, rng, batch=3, points=70, shared_values=True, spread=0.6)


def test_grid_sample_wild_coordinates(ctx, mem):
    for dtype in (np.float32, np.float64):
        pc.check_grid_sample_wild_coordinates(ctx, mem, dtype)


def test_embedded_obstacles(ctx, mem):
    """ geom.infinite_cylinder / embed (examples/grids/Wake_Flow.ipynb): the obstacle ignores the embedding axis; also as a union member """
    rng = np.random.default_rng(22)
    O = pc.O
    for dtype in (np.float32, np.float64):
        dom, grid = pc.make_case((12, 10, 8), ((CLO, OPN), (PER, PER), (PER, PER)), dtype, batch=2)
        cylinder = O.EmbeddedObstacle(O.SphereObstacle((5.0, 4.5), 2.6), (0, 1))                        # infinite along z
        slab = O.EmbeddedObstacle(O.BoxObstacle((2.0,), (4.5,)), (2,), velocity=(0.0, 0.3, 0.0))        # infinite along x and y
        pc.check_obstacle_kernels(ctx, mem, dom, grid, dtype, rng, [cylinder, slab])
        pc.check_obstacle_kernels(ctx, mem, dom, grid, dtype, rng,
                                  [O.UnionObstacle((O.EmbeddedObstacle(O.SphereObstacle((8.0, 3.0), 2.0), (0, 2)), O.BoxObstacle((1.0, 1.0, 1.0), (4.0, 3.0, 5.0))))])
        dom, grid = pc.make_case((16, 12), ((CLO, CLO), (OPN, OPN)), dtype, batch=1)
        pc.check_obstacle_kernels(ctx, mem, dom, grid, dtype, rng, [O.EmbeddedObstacle(O.BoxObstacle((5.0,), (9.0,)), (0,))])
    bad = pc.C.make_obstacles([dict(kind=pc.C.OBSTACLE_SPHERE, center=(1, 1), half_size=(1, 1), embed_mask=3)])
    with pytest.raises(pc.C.PhiHipError):
        ctx.apply_obstacles(grid, bad, 1, [0, 0])


def test_union_obstacles(ctx, mem):
    """ Obstacle(union(geometries)) (examples/grids/Fluid_Logo.ipynb): inside = any member, soft mask = max over the members --
    not the product the same members give as separate obstacles; groups next to plain obstacles, across the 16-per-launch split """
    rng = np.random.default_rng(21)
    O = pc.O
    for dtype in (np.float32, np.float64):
        dom, grid = pc.make_case((24, 20), ((CLO, CLO), (OPN, OPN)), dtype, batch=2)
        logo = O.UnionObstacle((O.BoxObstacle((4.0, 3.0), (7.0, 12.0)), O.BoxObstacle((7.0, 9.0), (13.0, 12.0)), O.SphereObstacle((13.5, 10.0), 2.2)))
        drifting = O.UnionObstacle((O.BoxObstacle((15.0, 2.0), (18.0, 5.0)), O.BoxObstacle((17.5, 4.0), (21.0, 7.5))), velocity=(0.3, -0.2))
        obstacles = [O.SphereObstacle((9.0, 16.0), 2.0, angular_velocity=0.5), logo, drifting, O.BoxObstacle((1.0, 15.0), (4.0, 18.5))]
        pc.check_obstacle_kernels(ctx, mem, dom, grid, dtype, rng, obstacles)
        separate = list(logo.members)
        a = O.apply_boundary_conditions(pc.random_velocity(dom, 1, dtype, np.random.default_rng(3)), [logo], dom)
        b = O.apply_boundary_conditions(pc.random_velocity(dom, 1, dtype, np.random.default_rng(3)), separate, dom)
        assert max(np.abs(x - y).max() for x, y in zip(a, b)) > 1e-3          # the union is a different obstacle than its members
    dom, grid = pc.make_case((12, 10, 16), ((CLO, CLO), (PER, PER), (CLO, OPN)), np.float32, batch=1)
    pc.check_obstacle_kernels(ctx, mem, dom, grid, np.float32, rng,
                              [O.UnionObstacle((O.BoxObstacle((4.0, 3.0, 5.0), (8.0, 7.0, 9.0)), O.SphereObstacle((7.0, 6.0, 10.0), 2.5)))])
    dom, grid = pc.make_case((32, 32), ((CLO, CLO), (CLO, CLO)), np.float32, batch=1)
    many = [O.SphereObstacle((2.0 + 1.5 * k, 3.0 + 1.2 * k), 1.0) for k in range(12)]
    many.append(O.UnionObstacle(tuple(O.SphereObstacle((28.0 - 1.4 * k, 4.0 + 1.3 * k), 1.2) for k in range(9))))   # straddles entry 16
    pc.check_obstacle_kernels(ctx, mem, dom, grid, np.float32, rng, many)
    too_many = [O.UnionObstacle(tuple(O.SphereObstacle((2.0 + k, 3.0 + k), 1.0) for k in range(17)))]
    with pytest.raises(pc.C.PhiHipError) as e:
        pc.check_obstacle_kernels(ctx, mem, dom, grid, np.float32, rng, too_many)
    assert e.value.status == -3


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_cg_deferred_x_update(ctx, mem, dtype):
    """ the marching 'CG' updates x every other iteration (UPDATE_R / UPDATE_X2, phihip_set_deferred_x_update): same solution as the
    plain update for odd and even iteration counts, an odd refresh period, early exits of single batch entries (flush of the pending
    step) -- each against the oracle and against the plain path """
    dom, grid = pc.make_case((32, 24, 64), ((CLO, OPN), (PER, PER), (CLO, CLO)), dtype, batch=3)
    try:
        ctx.set_small_grid_solver(False)
        for kwargs in (dict(max_iter=7, fixed_iterations=True), dict(max_iter=10, refresh=3, fixed_iterations=True),
                       dict(max_iter=9, refresh=4, fixed_iterations=True), dict(rtol=1e-3), dict()):
            xs = []
            for defer in (True, False):
                ctx.set_deferred_x_update(defer)
                x, info = pc.check_cg(ctx, mem, dom, grid, dtype, np.random.default_rng(8), **kwargs)
                xs.append((x, [i.iterations for i in info]))
            assert xs[0][1] == xs[1][1]                                  # the recurrence does not see x
            assert pc.rel_l2(xs[0][0], xs[1][0]) <= (5e-6 if dtype == np.float32 else 1e-11)    # recovering d_k costs a few digits of x
    finally:
        ctx.set_small_grid_solver(True)
        ctx.set_deferred_x_update(True)


def test_single_kernel_solver_16384_cells(ctx, mem):
    """ the 1024-thread x 16-cell form of cg_small (8193 ... 16384 cells, fp32): chosen for batches of >= 8 entries, here forced """
    try:
        ctx.set_small_grid_solver(16384)
        for res, bc in (((96, 100), ((CLO, OPN), (PER, PER))), ((24, 20, 25), ((CLO, CLO), (OPN, OPN), (PER, PER)))):
            dom, grid = pc.make_case(res, bc, np.float32, batch=2)
            pc.check_cg(ctx, mem, dom, grid, np.float32, np.random.default_rng(4), rtol=1e-4)
            pc.check_cg(ctx, mem, dom, grid, np.float32, np.random.default_rng(5), max_iter=9, refresh=4, fixed_iterations=True, adaptive=True)
    finally:
        ctx.set_small_grid_solver(True)


def test_cg_fixed_100_iterations_matches_oracle(ctx, mem):
    """ the benchmark mode (tolerances 0, exactly 100 iterations, refresh at 50) at 64^3 periodic fp32:
    pressure within 1e-4 rel-L2 of the NumPy oracle (north-star tolerance) """
    rng = np.random.default_rng(5)
    dom, grid = pc.make_case((64, 64, 64), ((PER, PER),) * 3, np.float32, upper=(2 * math.pi,) * 3)
    pc.check_cg(ctx, mem, dom, grid, np.float32, rng, max_iter=100, refresh=50, fixed_iterations=True)


def test_batch_entries_converge_independently(ctx, mem):
    dtype = np.float32
    dom, grid = pc.make_case((16, 16), ((CLO, CLO), (CLO, CLO)), dtype, batch=2)
    rng = np.random.default_rng(6)
    rhs = pc.O.balance_divergence(rng.standard_normal((2, 16, 16)).astype(dtype), None)
    rhs[0] = 0
    drhs, dx = mem.to_dev(rhs), mem.to_dev(np.zeros_like(rhs))
    info = ctx.cg_solve(grid, 0, 1, mem.ptr(drhs), mem.ptr(dx), pc.solve_params(dtype))
    assert info[0].iterations == 0 and info[0].converged == 1
    assert info[1].iterations > 5 and info[1].converged == 1
    assert np.all(mem.to_host(dx)[0] == 0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_obstacles(ctx, mem, dtype):
    rng = np.random.default_rng(8)
    dom, grid = pc.make_case((12, 10, 16), ((CLO, CLO),) * 3, dtype, batch=1)
    obstacles = [pc.O.BoxObstacle((4.0, 3.0, 5.0), (8.0, 7.0, 11.0))]
    active, hard, soft = pc.O.obstacle_masks(obstacles, dom, dtype)
    acc = (active[0] > 0).astype(np.uint8)
    dacc, dflags = mem.to_dev(acc), mem.empty(dom.res, np.uint8)
    g1 = pc.C.make_grid(3, grid.dtype, 1, dom.res, dom.lower, dom.upper, dom.bc, dom.bc_val)
    ctx.build_cellflags(g1, mem.ptr(dacc), 0, 1, mem.ptr(dflags))
    mem.sync()
    flags = mem.to_host(dflags)
    assert np.array_equal((flags >> 6) & 1, acc)
    pc.check_laplace(ctx, mem, dom, grid, dtype, rng, flags_np=flags, hard=hard, active=active)
    pc.check_cg(ctx, mem, dom, grid, dtype, rng, flags_np=flags, hard=hard, active=active)
    pc.check_make_incompressible(ctx, mem, dom, grid, dtype, rng, obstacles=obstacles)


@pytest.mark.parametrize("res,bc", GRIDS[:4] + GRIDS[6:9])
def test_make_incompressible_matches_oracle_and_is_divergence_free(ctx, mem, res, bc):
    """ tests/commit/physics/test_fluid.py:19-53: closed / open / periodic / mixed, batched; max|div| <= 5e-5 """
    rng = np.random.default_rng(9)
    dom, grid = pc.make_case(res, bc, np.float32, batch=3, upper=tuple(100.0 for _ in res))
    pc.check_make_incompressible(ctx, mem, dom, grid, np.float32, rng)


def test_launch_plan_regimes(ctx, mem):
    """ the launch plan switches regime with the grid: single-plane chunks for small 3-D grids (latency-bound), large tiles for large 2-D
    grids (cost of re-reading the partial sums), long chunks for large 3-D grids -- every regime against the oracle """
    cases = [((48, 40, 64), ((CLO, OPN), (PER, PER), (CLO, CLO)), 12),          # small 3-D: chunks of 1-2 planes
             ((1024, 1536), ((CLO, CLO), (OPN, CLO)), 6),                       # large 2-D: >= 1024 tiles per entry with small tiles
             ((160, 96, 128), ((PER, PER), (CLO, CLO), (OPN, OPN)), 6)]         # mid 3-D
    for res, bc, iters in cases:
        dom, grid = pc.make_case(res, bc, np.float32, batch=1)
        plans = [ctx.query_plan(grid, False, f) for f in (1, 2, 3)]
        assert all(p["nblk"] >= 1 and p["chunk"] >= 1 for p in plans)
        if res == (48, 40, 64):
            assert plans[0]["chunk"] <= 4
        if len(res) == 2:
            assert plans[0]["nblk"] <= 2048
        try:
            ctx.set_small_grid_solver(False)
            pc.check_laplace(ctx, mem, dom, grid, np.float32, np.random.default_rng(2))
            pc.check_cg(ctx, mem, dom, grid, np.float32, np.random.default_rng(3), max_iter=iters, fixed_iterations=True)
        finally:
            ctx.set_small_grid_solver(True)


def test_tile_configurations_agree(ctx, mem):
    rng = np.random.default_rng(10)
    dtype = np.float32
    dom, grid = pc.make_case((20, 72, 264), ((CLO, OPN), (PER, PER), (CLO, CLO)), dtype, batch=1)
    try:
        for rows, tpr in [(1, 16), (2, 16), (2, 32), (4, 32), (4, 64), (1, 64), (2, 64), (1, 32), (1, 128), (2, 128), (4, 128)]:      # (., 128): the ROW tiles (66 lanes per row here)
            for chunk in (3, 20):
                ctx.set_tuning(rows, tpr, chunk)
                if tpr == 128:
                    plan = ctx.query_plan(grid, False, 1)
                    assert plan["tpr"] == 128 and plan["rows"] == rows, plan       # the row tile really is what runs (not a silent substitute)
                pc.check_laplace(ctx, mem, dom, grid, dtype, np.random.default_rng(11))
        ctx.set_tuning(4, 32, 7)
        pc.check_cg(ctx, mem, dom, grid, dtype, rng, max_iter=8, fixed_iterations=True)
    finally:
        ctx.set_tuning(0, 0, 0)


@pytest.mark.parametrize("res,bc,dma32,dma64", [
    ((40, 36, 256), ((PER, PER), (PER, PER), (PER, PER)), True, True),      # the benchmark configuration's shape class: four tiles per row, 4.5 tile rows
    ((24, 20, 136), ((OPN, OPN), (PER, PER), (PER, PER)), True, True),      # clamped planes, a partial last tile along the fast axis
    ((19, 27, 64), ((PER, PER), (OPN, OPN), (PER, PER)), True, True),       # clamped halo rows, n1 + 1 faces of the a1 component
    ((12, 16, 70), ((PER, PER), (PER, PER), (PER, PER)), False, True),      # periodic rows of 70 cells: not whole fp32 vectors -> the register-staged kernel
    # the GEN instantiation (closed / open sides: constants from the table, patch elements, face offsets)
    ((12, 16, 72), ((PER, PER), (PER, PER), (OPN, OPN)), True, True),       # open fast axis: 73 faces per row
    ((40, 36, 256), ((CLO, CLO), (CLO, CLO), (CLO, CLO)), True, True),      # the closed box at the benchmark's row length: 255 faces per row of the a2 component
    ((24, 44, 200), ((CLO, OPN), (OPN, CLO), (CLO, OPN)), True, True),      # mixed sides, partial tiles on both tiled axes
    ((17, 30, 131), ((OPN, CLO), (CLO, CLO), (OPN, OPN)), True, True),      # ragged cell rows: every component straddles
    ((20, 24, 192), ((CLO, CLO), (PER, PER), (CLO, CLO)), True, True),      # periodic rows wrap onto the last row (the chunk that is not transferred)
])
def test_self_advection_lds_dma_fill(ctx, mem, res, bc, dma32, dma64):
    """ r5: ring of the tiled self-advection filled by global_load_lds_dwordx4 (inline assembly, counted vmcnt waits, raw s_barrier) -- regular grids and, second
    step, the GEN instantiation for closed / open boxes: oracle parity, the SAME bits as the register-staged kernel, the path asserted; many workgroups, several
    chunks, CFL below and above 1 (fix-up list), moving walls """
    rng = np.random.default_rng(41)
    for dtype, expect in ((np.float32, dma32), (np.float64, dma64)):
        bcv = None
        if any(side == CLO for pair in bc for side in pair):
            bcv = [[[float(rng.normal()) * 0.05 if bc[a][s] == CLO else 0.0 for c in range(3)] for s in range(2)] for a in range(3)]
        dom, grid = pc.make_case(res, bc, dtype, batch=2, bc_val=bcv)
        pc.check_advect_self_dma(ctx, mem, dom, grid, dtype, rng, dt=0.7, expect_dma=expect)
        pc.check_advect_self_dma(ctx, mem, dom, grid, dtype, rng, dt=2.1, expect_dma=expect)


@pytest.mark.parametrize("res,bc", [
    ((16, 20), ((CLO, CLO), (CLO, CLO))),
    ((64, 128), ((PER, PER), (OPN, OPN))),
    ((8, 12, 16), ((PER, PER), (PER, PER), (PER, PER))),
    ((24, 40, 128), ((PER, PER), (PER, PER), (PER, PER))),                  # LDS-DMA ring, several tiles and chunks
    ((40, 36, 256), ((CLO, CLO), (CLO, CLO), (CLO, CLO))),                  # the closed box at the benchmark's row length (GEN LDS-DMA instantiation)
    ((24, 44, 200), ((CLO, OPN), (OPN, CLO), (CLO, OPN))),                  # mixed sides, partial tiles on both tiled axes
    ((20, 24, 192), ((CLO, CLO), (PER, PER), (CLO, CLO))),
])
def test_advection_paths_give_the_same_bits(ctx, mem, res, bc):
    """ r6 (VERDICT r5 weak 1 ii): tile (register-staged and LDS-DMA fill, reach 1 and 2), fix-up work list, LDS windows and gather kernels evaluate
    ONE arithmetic per sample (advect_common.hpp) -- the same bits whatever the path, also for displacements of 1e-9 cells of either sign. This is what makes
    the adaptive reach, a captured graph's fixed reach and a slab's window passes interchangeable in the last bit. """
    rng = np.random.default_rng(66)
    s_codes = tuple((PER, PER) if lo == PER else (OPN, CLO) for lo, hi in bc)
    s_consts = [(0.0, 0.25)] * len(res)
    for dtype in (np.float32, np.float64):
        bcv = None
        if len(res) == 3 and any(side == CLO for pair in bc for side in pair):
            bcv = [[[float(rng.normal()) * 0.05 if bc[a][sd] == CLO else 0.0 for c in range(3)] for sd in range(2)] for a in range(3)]
        dom, grid = pc.make_case(res, bc, dtype, batch=2, bc_val=bcv)
        for dt in (0.7, 2.3):
            pc.check_advect_paths_same_bits(ctx, mem, dom, grid, dtype, rng, s_codes, s_consts, dt=dt)


@pytest.mark.parametrize("res,bc,dma", [
    ((8, 12, 16), ((PER, PER), (PER, PER), (PER, PER)), True),
    ((24, 40, 128), ((PER, PER), (PER, PER), (PER, PER)), True),      # several tiles and chunks
    ((64, 64, 64), ((PER, PER), (PER, PER), (PER, PER)), True),
    ((21, 30, 136), ((OPN, OPN), (OPN, OPN), (PER, PER)), True),      # open slow axes (clamped planes and rows), partial tiles on both tiled axes
    ((12, 16, 72), ((PER, PER), (PER, PER), (OPN, OPN)), False),      # open fast axis
    ((20, 24, 192), ((CLO, CLO), (PER, PER), (PER, PER)), False),     # a closed side
])
def test_mac_cormack_windows_lds_dma_fill(ctx, mem, res, bc, dma):
    """ r6: the staggered MacCormack correction pass with its six LDS windows filled by global_load_lds_dwordx4 (inline assembly, counted
    vmcnt waits, raw s_barrier) on regular grids: oracle parity, the SAME bits as the register-staged windows, the path asserted; CFL below and above 1.
    Reference: phi/physics/advect.py:182-215. """
    rng = np.random.default_rng(43)
    for dtype in (np.float32, np.float64):
        dom, grid = pc.make_case(res, bc, dtype, batch=2)
        pc.check_mac_cormack_staggered_dma(ctx, mem, dom, grid, dtype, rng, dt=0.7, expect_dma=dma)
        pc.check_mac_cormack_staggered_dma(ctx, mem, dom, grid, dtype, rng, dt=2.1, expect_dma=dma)


def test_cellflags_byte_parallel_kernel(ctx, mem):
    """ r5: phihip_build_cellflags -- byte-parallel kernel (16 / 4 cells per thread) and the scalar kernel -- on random masks with arbitrary non-zero
    bytes against the NumPy restatement of fluid.py:130-137,277-288; sizes that span many workgroups, every boundary kind, per-batch masks """
    rng = np.random.default_rng(31)
    for res, bc in (((40, 36, 256), ((PER, PER), (CLO, OPN), (OPN, CLO))), ((9, 70, 80), ((CLO, CLO), (PER, PER), (PER, PER))), ((300, 512), ((OPN, OPN), (CLO, CLO))),
                    ((16, 15, 20), ((OPN, CLO), (PER, PER), (CLO, OPN))), ((33, 31, 29), ((CLO, OPN), (OPN, OPN), (PER, PER))), ((1, 1, 16), ((PER, PER), (OPN, OPN), (CLO, CLO)))):
        for masks, with_active in ((1, True), (1, False), (2, True)):
            pc.check_cellflags(ctx, mem, res, bc, rng, masks, with_active)


@pytest.mark.parametrize("res,bc,dt", [
    ((20, 72, 264), ((CLO, OPN), (PER, PER), (CLO, CLO)), np.float32),      # 66 vectors per row, mixed boundaries at the row's ends
    ((12, 40, 288), ((PER, PER), (CLO, CLO), (PER, PER)), np.float32),      # 72 lanes, ragged last tile (40 rows in tiles of 3)
    ((9, 21, 384), ((OPN, OPN), (OPN, CLO), (OPN, OPN)), np.float32),       # 96 lanes: two thread rows per workgroup
    ((40, 36, 384), ((CLO, CLO),) * 3, np.float32),                         # the size class the first-call autotune picks row tiles for
    ((8, 19, 132), ((PER, PER), (CLO, OPN), (CLO, OPN)), np.float64),       # fp64: rows of 66 vectors of two
    ((6, 30, 192), ((CLO, CLO),) * 3, np.float64),                          # fp64, 96 lanes
])
def test_row_tiles_on_the_gpu(ctx, mem, res, bc, dt):
    """ the ROWT instantiation of march_kernel (tile ids 8-10: whole rows of 65 ... 128 vectors, run-time lanes per row,
    no halo columns -- the r4 kernels that carry 288^3 ... 448^3) pinned DETERMINISTICALLY on the hardware: every row-tile shape with the plan
    asserted, the operator, fixed-iteration CG in both forms incl. a true-residual refresh, cell flags (obstacle) and fp64 -- the GPU mirror of
    tests/test_emu_kernels.py::test_tile_configurations_agree. Reference: phi/physics/fluid.py:165-202 (masked_laplace), :156 (solve_linear). """
    dom, grid = pc.make_case(res, bc, dt, batch=2)
    try:
        ctx.set_small_grid_solver(False)
        for rows in (1, 2, 4):
            for chunk in (2, 7):
                ctx.set_tuning(rows, 128, chunk)
                for flags in (False, True):
                    plan = ctx.query_plan(grid, flags, 1)
                    assert plan["tpr"] == 128 and plan["rows"] == rows, plan
                pc.check_laplace(ctx, mem, dom, grid, dt, np.random.default_rng(11))
            for mode in ((0, 2) if rows == 1 else (0,)):
                ctx.set_single_reduction_cg(mode)
                pc.check_cg(ctx, mem, dom, grid, dt, np.random.default_rng(12), max_iter=9, refresh=4, fixed_iterations=True)
            ctx.set_single_reduction_cg(0)
        # cell flags on a row tile: projection around a box obstacle (closed boxes only: the obstacle must not touch an open side's ghost)
        if all(lo == CLO and hi == CLO for lo, hi in bc):
            ctx.set_tuning(2, 128, 5)
            n = res
            ob = pc.O.BoxObstacle(tuple(0.3 * x for x in n), tuple(0.55 * x for x in n))
            dom1, grid1 = pc.make_case(res, bc, dt, batch=1, upper=tuple(float(x) for x in n))
            pc.check_make_incompressible(ctx, mem, dom1, grid1, dt, np.random.default_rng(13), obstacles=[ob])
    finally:
        ctx.set_tuning(0, 0, 0)
        ctx.set_single_reduction_cg(1)
        ctx.set_small_grid_solver(True)


@pytest.mark.parametrize("res,bc", [((8, 19, 264), ((PER, PER), (CLO, OPN), (CLO, OPN))), ((40, 36, 384), ((CLO, CLO),) * 3), ((6, 30, 512), ((OPN, OPN), (PER, PER), (PER, PER))),
                                    ((70, 300), ((CLO, OPN), (OPN, CLO)))])
def test_wide_row_tiles_on_the_gpu(ctx, mem, res, bc):
    """ r6: the WIDE row tiles -- fp64 rows of 129 ... 256 vectors as ONE tile row (ids 11 / 12; 384-cell rows of BASELINE configs[4] = 192
    lanes), no halo columns, both halo vectors of a column fetched by its lane -- pinned with the plan asserted: operator, both CG forms incl. a refresh, flags,
    a projection around an obstacle. Reference: phi/physics/fluid.py:156-161,165-202. """
    dt = np.float64
    dom, grid = pc.make_case(res, bc, dt, batch=2)
    try:
        ctx.set_small_grid_solver(False)
        for rows in (2, 4):
            for chunk in (2, 7):
                ctx.set_tuning(rows, 256, chunk)
                for flags in (False, True):
                    plan = ctx.query_plan(grid, flags, 1)
                    assert plan["tpr"] == 256 and plan["rows"] == rows, plan
                pc.check_laplace(ctx, mem, dom, grid, dt, np.random.default_rng(11))
            for mode in (0, 2):
                ctx.set_single_reduction_cg(mode)
                pc.check_cg(ctx, mem, dom, grid, dt, np.random.default_rng(12), max_iter=9, refresh=4, fixed_iterations=True)
            ctx.set_single_reduction_cg(0)
        if all(lo == CLO and hi == CLO for lo, hi in bc):
            ctx.set_tuning(2, 256, 5)
            ob = pc.O.BoxObstacle(tuple(0.3 * x for x in res), tuple(0.55 * x for x in res))
            dom1, grid1 = pc.make_case(res, bc, dt, batch=1, upper=tuple(float(x) for x in res))
            pc.check_make_incompressible(ctx, mem, dom1, grid1, dt, np.random.default_rng(13), obstacles=[ob])
    finally:
        ctx.set_tuning(0, 0, 0)
        ctx.set_single_reduction_cg(1)
        ctx.set_small_grid_solver(True)


def test_workspace_placement_same_bits_on_the_gpu(gpu_backend):
    """ r6 placement of the CG workspace (cg.hip place_workspace, include/phihip.h phihip_workspace_placement): a fresh context chooses between five candidate
    (r, d0, d1) allocations by timing the iteration loop; the solve on the kept workspace has the bits of a context that keeps its first allocation and runs
    the same launch plans (where a vector lives is no part of the arithmetic), and the record says what was chosen. 288^3 fp32: 96 MB per vector, beyond
    the Infinity Cache regime (smaller vectors are not placed). Reference: phi/physics/fluid.py:156-161 (the solve whose workspace this is). """
    import torch
    C = pc.C
    lib, dev = gpu_backend.ctx.lib, gpu_backend.device
    n = 288
    L = 2 * math.pi
    grid = C.make_grid(3, C.PHIHIP_F32, 1, (n, n, n), (0, 0, 0), (L, L, L), ((0, 0),) * 3)
    rhs = torch.randn(1, n, n, n, generator=torch.Generator(device=dev).manual_seed(5), device=dev)
    rhs -= rhs.mean()
    placed, plain = C.Context(lib, 0), C.Context(lib, 0)
    try:
        assert placed.workspace_placement(5)["candidates"] == 0
        xs = []
        for _ in range(2):                    # the second solve runs on the workspace the first one kept
            x = torch.zeros_like(rhs)
            placed.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, 30, 0, 0, 0), want_info=False)
            torch.cuda.synchronize()
            xs.append(x)
        rec = placed.workspace_placement()
        assert rec["candidates"] == 5 and 0 < rec["us_kept"] <= rec["us_first"], rec
        print("workspace placement at 288^3:", rec)
        plain.set_autotune(False)
        plain.workspace_placement(1)
        for f in (0, 1, 2, 3):
            q = placed.query_plan(grid, False, f)
            plain.set_tuning_kernel(f, int(q["rows"]), int(q["tpr"]), int(q["chunk"]))
        x = torch.zeros_like(rhs)
        plain.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, 30, 0, 0, 0), want_info=False)
        torch.cuda.synchronize()
        assert plain.workspace_placement()["candidates"] == 0
        assert torch.equal(xs[0], xs[1]) and torch.equal(xs[0], x)
        assert float(x.abs().max()) > 0 and bool(torch.isfinite(x).all())
    finally:
        placed.close()
        plain.close()


# ---- full-size properties (BASELINE.json sizes; the oracle is too slow there) ----------------------------------------
def _eigen_rhs(n, dtype):
    """ rhs = lambda_h sin x sin y sin z at cell centres: exact discrete solution p = sin x sin y sin z """
    h = 2 * math.pi / n
    c = (np.arange(n) + 0.5) * h
    s = np.sin(c)
    lam = 3 * (2 * math.cos(h) - 2) / h ** 2
    p = (s[:, None, None] * s[None, :, None] * s[None, None, :])
    return (lam * p).astype(dtype)[None], p.astype(dtype)[None]


@pytest.mark.parametrize("n", [256, 512])
def test_full_size_eigenfunction_solve(ctx, mem, n):
    """ 256^3 / 512^3 fp32 periodic: CG on the discrete eigenfunction rhs converges in one iteration to the exact solution """
    dom, grid = pc.make_case((n, n, n), ((PER, PER),) * 3, np.float32, upper=(2 * math.pi,) * 3)
    rhs, p_exact = _eigen_rhs(n, np.float32)
    drhs, dx = mem.to_dev(rhs), mem.to_dev(np.zeros_like(rhs))
    info = ctx.cg_solve(grid, 0, 1, mem.ptr(drhs), mem.ptr(dx), pc.solve_params(np.float32, max_iter=20, rtol=1e-4))
    x = mem.to_host(dx)
    # exact arithmetic: 1 iteration; fp32 round-off of the 134 M-term dot products adds a few clean-up iterations at 512^3
    assert info[0].converged and info[0].iterations <= 6, (info[0].iterations, info[0].residual_sq, info[0].rhs_sq)
    assert pc.rel_l2(x, p_exact) < 1e-4


def test_full_size_projection_is_divergence_free_and_linear(ctx, mem):
    """ 256^3 fp32 periodic: after projection to rel_tol 1e-5 the discrete divergence vanishes (<= 5e-5, the reference's own
    criterion) and the operator is linear: A(a p + q) == a A p + A q """
    import torch
    n = 256
    dom, grid = pc.make_case((n, n, n), ((PER, PER),) * 3, np.float32, upper=(2 * math.pi,) * 3)
    g = torch.Generator(device='cpu').manual_seed(0)
    v = [torch.randn(1, n, n, n, generator=g).mul_(0.05) for _ in range(3)]
    dv = [t.to(mem.device) for t in v]
    p = torch.zeros(1, n, n, n, device=mem.device)
    div = torch.empty_like(p)
    info = ctx.make_incompressible(grid, [t.data_ptr() for t in dv], None, 0, 1, True, p.data_ptr(), div.data_ptr(),
                                   pc.solve_params(np.float32, max_iter=2000, rtol=1e-5))
    assert info[0].converged
    ctx.divergence(grid, [t.data_ptr() for t in dv], 0, 1, False, div.data_ptr())
    mem.sync()
    assert float(div.abs().max()) <= 5e-5 * (n / (2 * math.pi)) * 0.05 * 10, float(div.abs().max())
    a = 0.37
    q = torch.randn(1, n, n, n, generator=g).to(mem.device)
    out1, out2, out3 = torch.empty_like(p), torch.empty_like(p), torch.empty_like(p)
    comb = (a * p + q).contiguous()
    ctx.laplace_apply(grid, 0, 1, comb.data_ptr(), out1.data_ptr())
    ctx.laplace_apply(grid, 0, 1, p.data_ptr(), out2.data_ptr())
    ctx.laplace_apply(grid, 0, 1, q.data_ptr(), out3.data_ptr())
    mem.sync()
    ref = a * out2 + out3
    assert float((out1 - ref).abs().max() / ref.abs().max()) < 1e-5
    # self-adjointness: <q, A p> == <A q, p>
    lhs = float((q.double() * out2.double()).sum())
    rhs = float((out3.double() * p.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), abs(rhs), 1.0)


def test_fp64_cavity_with_obstacle_small(ctx, mem):
    """ config-5 flavour at a size the oracle can check: closed box, lid velocity on z+, solid box obstacle, fp64 """
    rng = np.random.default_rng(12)
    bcv = np.zeros((3, 2, 3)); bcv[2, 1, 0] = 1.0
    dom, grid = pc.make_case((24, 24, 24), ((CLO, CLO),) * 3, np.float64, batch=1, bc_val=bcv, upper=(1.0, 1.0, 1.0))
    obstacles = [pc.O.BoxObstacle((0.4, 0.4, 0.4), (0.6, 0.6, 0.6))]
    pc.check_make_incompressible(ctx, mem, dom, grid, np.float64, rng, obstacles=obstacles, max_div=1e-7)


def test_full_size_advection_is_exact_translation(ctx, mem):
    """ 256^3 fp32 periodic, unit cells, constant velocity of whole cells per step: the semi-Lagrangian lookup lands on grid
    points, so both advection schemes must reproduce a cyclic shift of the field bit for bit (size-independent property) """
    import torch
    n = 256
    dom, grid = pc.make_case((n, n, n), ((PER, PER),) * 3, np.float32)          # bounds [0, n]^3 -> dx = 1
    g = torch.Generator(device='cpu').manual_seed(1)
    field = [torch.randn(1, n, n, n, generator=g).to(mem.device) for _ in range(3)]
    shift = (2.0, -1.0, 3.0)
    vel = [torch.full((1, n, n, n), s, device=mem.device) for s in shift]
    out = [torch.empty_like(t) for t in field]
    P = lambda ts: [t.data_ptr() for t in ts]
    ctx.advect_staggered(grid, P(field), P(vel), P(out), 1.0)
    mem.sync()
    for f, o in zip(field, out):
        assert torch.equal(o, torch.roll(f, shifts=(2, -1, 3), dims=(1, 2, 3)))
    ctx.mac_cormack_staggered(grid, P(field), P(vel), P(out), 1.0, 1.0)
    mem.sync()
    for f, o in zip(field, out):
        assert torch.equal(o, torch.roll(f, shifts=(2, -1, 3), dims=(1, 2, 3)))
    s = torch.randn(1, n, n, n, generator=g).to(mem.device)
    so = torch.empty_like(s)
    ctx.mac_cormack_centered(grid, s.data_ptr(), ((0, 0),) * 3, None, P(vel), so.data_ptr(), 1.0, 1.0)
    mem.sync()
    assert torch.equal(so, torch.roll(s, shifts=(2, -1, 3), dims=(1, 2, 3)))


def test_full_size_cavity_fp64_with_obstacle(ctx, mem):
    """ BASELINE configs[4] at full size: 384^3 fp64 closed box with a moving lid and a solid box. After the projection the
    discrete divergence vanishes on every fluid cell, no face of the solid carries flow, and applying the projection twice
    changes nothing (idempotence). """
    import torch
    n = 384
    bcv = np.zeros((3, 2, 3)); bcv[2, 1, 0] = 1.0
    dom, grid = pc.make_case((n, n, n), ((CLO, CLO),) * 3, np.float64, bc_val=bcv, upper=(1.0, 1.0, 1.0))
    obstacle = pc.C.make_obstacles([dict(kind=pc.C.OBSTACLE_BOX, center=(0.5, 0.5, 0.5), half_size=(0.125, 0.125, 0.125))])
    g1 = pc.C.make_grid(3, grid.dtype, 1, dom.res, dom.lower, dom.upper, dom.bc, dom.bc_val)
    acc = torch.empty(n, n, n, dtype=torch.uint8, device=mem.device)
    flags = torch.empty_like(acc)
    ctx.obstacle_accessible(g1, obstacle, 1, acc.data_ptr())
    ctx.build_cellflags(g1, acc.data_ptr(), 0, 1, flags.data_ptr())
    lo, hi = int(0.375 * n), int(0.625 * n)
    assert int((acc == 0).sum()) == (hi - lo) ** 3
    gen = torch.Generator(device='cpu').manual_seed(2)
    v = [(0.05 * torch.randn((1,) + dom.comp_shape(d), generator=gen, dtype=torch.float64)).to(mem.device) for d in range(3)]
    P = lambda ts: [t.data_ptr() for t in ts]
    ctx.apply_obstacles(grid, obstacle, 1, P(v))
    p = torch.zeros(1, n, n, n, dtype=torch.float64, device=mem.device)
    div = torch.empty_like(p)
    solve = pc.C.Solve(1e-9, 0.0, 4000, 50, 50, 0)
    info = ctx.make_incompressible(grid, P(v), None, flags.data_ptr(), 1, True, p.data_ptr(), div.data_ptr(), solve)
    assert info[0].converged, (info[0].iterations, info[0].residual_sq, info[0].rhs_sq)
    rhs_scale = float(div.abs().max())
    ctx.divergence(grid, P(v), flags.data_ptr(), 1, False, div.data_ptr())
    mem.sync()
    assert float(div.abs().max()) <= 1e-6 * rhs_scale, (float(div.abs().max()), rhs_scale)
    assert float(v[0][0, lo:hi - 1, lo:hi, lo:hi].abs().max()) == 0.0          # x-faces strictly inside the solid
    before = [t.clone() for t in v]
    p2 = p.clone()
    info2 = ctx.make_incompressible(grid, P(v), None, flags.data_ptr(), 1, True, p2.data_ptr(), 0, solve)
    mem.sync()
    for a, b in zip(before, v):
        assert float((a - b).abs().max()) <= 1e-7 * 0.05


def test_full_size_taylor_green_run_is_stable(ctx, mem):
    """ BASELINE configs[1] at full size, the benchmark's own loop: 12 steps of {semi-Lagrangian self-advection, projection with
    exactly 100 CG iterations from the previous pressure}. Size-independent sanity properties: all fields stay finite, the
    kinetic energy never grows (semi-Lagrangian advection and the projection are both dissipative), the relative residual of
    every solve stays small, and the z-component of the extruded 2-D vortex stays zero. """
    import torch
    n = 256
    L = 2 * math.pi
    dom, grid = pc.make_case((n, n, n), ((PER, PER),) * 3, np.float32, upper=(L,) * 3)
    h = L / n
    idx = torch.arange(n, dtype=torch.float64)
    face, cent = idx * h, (idx + 0.5) * h
    u = (torch.cos(face)[:, None] * torch.sin(cent)[None, :])[:, :, None].expand(n, n, n)
    w = (-torch.sin(cent)[:, None] * torch.cos(face)[None, :])[:, :, None].expand(n, n, n)
    v = [t.to(torch.float32).unsqueeze(0).contiguous().to(mem.device) for t in (u, w, torch.zeros(n, n, n, dtype=torch.float64))]
    v2 = [torch.empty_like(t) for t in v]
    p = torch.zeros(1, n, n, n, device=mem.device)
    res = torch.zeros(1, 2, dtype=torch.float64, device=mem.device)
    solve = pc.C.Solve(0.0, 0.0, 100, 50, 0, 0)
    P = lambda ts: [t.data_ptr() for t in ts]
    energy = [float(sum((t.double() ** 2).sum() for t in v))]
    for _ in range(12):
        ctx.advect_staggered(grid, P(v), P(v), P(v2), 0.5 * h)
        ctx.make_incompressible(grid, P(v2), None, 0, 1, True, p.data_ptr(), 0, solve, want_info=False)
        ctx.solve_residuals(1, res.data_ptr())
        v, v2 = v2, v
        mem.sync()
        energy.append(float(sum((t.double() ** 2).sum() for t in v)))
        assert all(bool(torch.isfinite(t).all()) for t in v) and bool(torch.isfinite(p).all())
        assert float(torch.sqrt(res[0, 0] / res[0, 1])) < 5e-2
    assert all(b <= a * (1 + 1e-6) for a, b in zip(energy, energy[1:])), energy
    assert energy[-1] > 0.9 * energy[0]                        # ... and the vortex is still there
    assert float(v[2].abs().max()) < 1e-3


def test_rccl_allreduce_residual_through_the_c_abi(ctx, mem):
    """ `phihip_allreduce_residual` over an RCCL communicator a C caller created itself (here: one rank, through ctypes on
    the librccl torch already loaded) -- the collective is resolved at run time from that library, in place, on the solve stream """
    import ctypes
    import os
    import torch
    rccl = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        dom, grid = pc.make_case((32, 24, 64), ((PER, PER),) * 3, np.float32, batch=3)
        ctx.set_small_grid_solver(False)
        pc.check_cg(ctx, mem, dom, grid, np.float32, np.random.default_rng(1), max_iter=5, fixed_iterations=True)
        res = torch.zeros(3, 2, dtype=torch.float64, device=mem.device)
        ctx.solve_residuals(3, res.data_ptr())
        before = res.clone()
        ctx.allreduce_residual(comm, res.data_ptr(), 6, op=2)                 # max over the (single) rank: unchanged
        ctx.allreduce_residual(comm, res.data_ptr(), 6, op=0)                 # sum over one rank: unchanged
        mem.sync()
        assert torch.equal(res, before) and float(before[:, 1].min()) > 0
        rel = torch.zeros(1, dtype=torch.float64, device=mem.device)          # the same result reduced to the all-reduce's operand
        ctx.solve_relative_residual(3, rel.data_ptr())
        ctx.allreduce_residual(comm, rel.data_ptr(), 1, op=2)
        mem.sync()
        assert abs(float(rel) - float(torch.sqrt(before[:, 0] / before[:, 1]).max())) <= 1e-12 * float(rel)
        with pytest.raises(pc.C.PhiHipError):
            ctx.allreduce_residual(comm, res.data_ptr(), 6, op=1)
        with pytest.raises(pc.C.PhiHipError):
            ctx.allreduce_residual(None, res.data_ptr(), 6)
    finally:
        ctx.set_small_grid_solver(True)
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)


@pytest.mark.parametrize("res,bc,batch,obstacles", [
    ((40, 216), ((CLO, CLO), (CLO, CLO)), 2, [pc.O.BoxObstacle((10.0, 60.0), (22.0, 130.0))]),
    ((33, 264), ((OPN, OPN), (CLO, OPN)), 1, [pc.O.SphereObstacle((16.0, 100.0), 9.5), pc.O.BoxObstacle((0.0, 200.0), (12.0, 230.0))]),
    ((72, 128), ((PER, PER), (CLO, CLO)), 1, [pc.O.SphereObstacle((36.0, 60.0), 14.0)]),
    ((130, 512), ((CLO, CLO), (CLO, OPN)), 4, [pc.O.BoxObstacle((40.0, 100.0), (70.0, 300.0)), pc.O.SphereObstacle((100.0, 400.0), 20.0)]),      # 9 workgroups per entry, rows of 512 cells
    ((200, 256), ((CLO, CLO), (CLO, CLO)), 8, [pc.O.SphereObstacle((100.0, 128.0), 40.0)]),                                                    # 104 workgroups
    ((400, 512), ((CLO, CLO), (CLO, CLO)), 11, [pc.O.BoxObstacle((100.0, 200.0), (180.0, 300.0))])])                                           # 25 workgroups per entry: sub-batches of 10 + 1
def test_resident_cg_with_cell_flags(ctx, mem, res, bc, batch, obstacles):
    """ r6: the resident solver takes solves WITH cell flags (obstacles): fixed iterations across a refresh, tolerance mode and the projection
    against the oracle; the launch counters assert one resident launch per solve """
    pc.check_resident_with_flags(ctx, mem, res, bc, batch, obstacles)


def test_resident_cg_cooperative_launch_and_two_streams(gpu_backend, mem):
    """ r6: (i) the OPTIONAL cooperative launch of the resident solver (PHIHIP_RESIDENT_COOP=1, read when a context is created; the default is the plain launch since the
    cooperative one was measured to cost 0.04-0.5 ms per solve, profiles/r06_resident_coop_cost.txt) solves like the plain one; (ii) resident solves of two contexts on two
    streams are chained by an event (they never overlap): both finish, both agree with a solve done alone """
    import os
    import torch
    lib = gpu_backend.ctx.lib
    old = os.environ.get("PHIHIP_RESIDENT_COOP")
    try:
        os.environ["PHIHIP_RESIDENT_COOP"] = "1"
        coop = pc.C.Context(lib, 0)
    finally:
        if old is None:
            os.environ.pop("PHIHIP_RESIDENT_COOP", None)
        else:
            os.environ["PHIHIP_RESIDENT_COOP"] = old
    plain = pc.C.Context(lib, 0)
    for c in (coop, plain):
        c.set_resident_cg(2)
    dom, grid = pc.make_case((200, 264), ((PER, PER), (CLO, OPN)), np.float32, batch=3)
    for c in (coop, plain):
        c.profile_enable(True)
        c.profile_read(True)
        pc.check_cg(c, mem, dom, grid, np.float32, np.random.default_rng(3), max_iter=60, refresh=50, fixed_iterations=True)
        prof = c.profile_read(True)
        c.profile_enable(False)
        assert prof["cg_matvec_dot"][0] == 0 and prof["cg_update"][0] == 1, prof
    # two streams, two contexts, 8 x 512^2 each (256 workgroups: ONE of them fills the chip), enqueued back to back without a host wait in between
    dom, grid = pc.make_case((512, 512), ((CLO, CLO), (CLO, CLO)), np.float32, batch=8)
    rng = np.random.default_rng(5)
    y = rng.standard_normal((8, 512, 512)).astype(np.float32)
    y -= y.mean(axis=(1, 2), keepdims=True)
    dy = mem.to_dev(y)
    xs = [mem.to_dev(np.zeros_like(y)) for _ in range(3)]
    solve = pc.C.Solve(0.0, 0.0, 200, 50, 0, 0)
    plain.cg_solve(grid, 0, 1, mem.ptr(dy), mem.ptr(xs[2]), solve)           # alone
    mem.sync()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    other = pc.C.Context(lib, 0)
    other.set_resident_cg(2)
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
    for c, st, x in ((plain, streams[0], xs[0]), (other, streams[1], xs[1])):
        c.cg_solve(grid, 0, 1, mem.ptr(dy), mem.ptr(x), solve, want_info=False, stream=st.cuda_stream)
    for st in streams:
        st.synchronize()
    ref = mem.to_host(xs[2])
    for x in xs[:2]:
        assert np.array_equal(mem.to_host(x), ref), "a resident solve that shared the device with another one differs from the solve done alone"


def test_resident_cg(ctx, mem):
    """ phihip_set_resident_cg (cg_resident.hip): the whole 2-D fp32 solve in ONE launch of resident workgroups -- vectors in registers, one
    barrier per iteration among the workgroups of a batch entry, control logic on the device. Same iterates as the launch forms / the
    oracle: fixed iteration counts incl. true-residual refreshes, tolerance mode (BASELINE configs[3]'s 8 x 512^2 among the shapes), every
    boundary kind on both axes, a ragged last workgroup (n_y % 16 != 0), the balanced projection (shift folded into the first pass) """
    try:
        ctx.set_resident_cg(2)
        for res, bc, batch in (((512, 512), ((CLO, CLO), (CLO, CLO)), 8), ((200, 264), ((PER, PER), (CLO, OPN)), 3), ((100, 96), ((OPN, CLO), (PER, PER)), 2),
                               ((512, 512), ((PER, PER), (PER, PER)), 1),
                               ((512, 512), ((CLO, OPN), (CLO, CLO)), 12)):      # (r6: 12 entries x 32 workgroups: sub-batches of 8 + 4, the second one with another layout of the exchange buffer)
            dom, grid = pc.make_case(res, bc, np.float32, batch=batch)
            pc.check_cg(ctx, mem, dom, grid, np.float32, np.random.default_rng(3), max_iter=120, refresh=50, fixed_iterations=True)
            if res[0] < 512:      # (tolerance mode on a white-noise right-hand side: thousands of iterations at 512^2)
                pc.check_cg(ctx, mem, dom, grid, np.float32, np.random.default_rng(4))
        dom, grid = pc.make_case((256, 384), ((CLO, CLO), (CLO, CLO)), np.float32, batch=4)
        pc.check_make_incompressible(ctx, mem, dom, grid, np.float32, np.random.default_rng(6))
        # the resident launch against the launch-per-iteration default on the same right-hand side: same iteration count (+- 5 %), same solution
        n = 512
        dom, grid = pc.make_case((n, n), ((CLO, CLO), (CLO, CLO)), np.float32, batch=1, upper=(100.0, 100.0))
        y = np.zeros((1, n, n), np.float32)
        y[0, n // 2 - 5:n // 2 + 5, 5:15], y[0, n // 2 - 5:n // 2 + 5, 15:25] = 0.1, -0.1
        y -= y.mean(dtype=np.float64).astype(np.float32)
        out = {}
        for mode in (0, 2):
            ctx.set_resident_cg(mode)
            dy, dx = mem.to_dev(y), mem.to_dev(np.zeros_like(y))
            info = ctx.cg_solve(grid, 0, 1, mem.ptr(dy), mem.ptr(dx), pc.C.Solve(1e-4, 0.0, 4000, 50, 10, 0))
            mem.sync()
            x = mem.to_host(dx).astype(np.float64)
            true_res = np.linalg.norm(y.astype(np.float64) - pc.O.masked_laplace(x, dom, None, None)) / np.linalg.norm(y.astype(np.float64))
            out[mode] = (info[0].iterations, bool(info[0].converged), true_res, x - x.mean())
            print(f"closed 512^2 fp32 rtol 1e-4, resident mode {mode}: {info[0].iterations} iterations, converged {bool(info[0].converged)}, true relative residual {true_res:.3e}")
        assert out[0][1] and out[2][1]
        assert abs(out[2][0] - out[0][0]) <= 0.05 * out[0][0], (out[0][0], out[2][0])
        assert out[2][2] <= 2e-4
        assert np.linalg.norm(out[2][3] - out[0][3]) <= 2e-3 * np.linalg.norm(out[0][3])
    finally:
        ctx.set_resident_cg(1)          # the library's default since r6


def test_single_reduction_cg_opt_in(ctx, mem):
    """ phihip_set_single_reduction_cg: ONE fused launch per iteration (Chronopoulos-Gear recurrences). Same iterates as PhiML's cg while
    far from the rounding floor (fixed iteration counts, fp32), converged solutions in fp64; off by default (fp32 accuracy floor) """
    try:
        ctx.set_small_grid_solver(False)
        ctx.set_single_reduction_cg(2)
        dom, grid = pc.make_case((64, 64, 64), ((PER, PER), (CLO, OPN), (CLO, CLO)), np.float32, batch=2)
        pc.check_cg(ctx, mem, dom, grid, np.float32, np.random.default_rng(3), max_iter=40, refresh=16, fixed_iterations=True)
        dom, grid = pc.make_case((512, 512), ((CLO, CLO), (CLO, CLO)), np.float32, batch=2)
        pc.check_cg(ctx, mem, dom, grid, np.float32, np.random.default_rng(4), max_iter=60, fixed_iterations=True)
        dom, grid = pc.make_case((96, 128), ((CLO, OPN), (PER, PER)), np.float64, batch=3)
        pc.check_cg(ctx, mem, dom, grid, np.float64, np.random.default_rng(5))
        pc.check_make_incompressible(ctx, mem, dom, grid, np.float64, np.random.default_rng(6), max_div=1e-7)
        # r3, the five-sum closure of alpha: TOLERANCE mode at the size where the textbook closure stalled (closed 512^2 fp32, localized
        # dipole: relative residual floor 9e-4 -- it never reached 1e-4; tools/cg1_accuracy.py). The single-reduction form now has to
        # converge to 1e-4 in as many iterations (+- 5 %) as the two-launch form, and both solutions must agree.
        n = 512
        dom, grid = pc.make_case((n, n), ((CLO, CLO), (CLO, CLO)), np.float32, batch=1, upper=(100.0, 100.0))
        y = np.zeros((1, n, n), np.float32)
        y[0, n // 2 - 5:n // 2 + 5, 5:15], y[0, n // 2 - 5:n // 2 + 5, 15:25] = 0.1, -0.1
        y -= y.mean(dtype=np.float64).astype(np.float32)
        out = {}
        for mode in (0, 2):
            ctx.set_single_reduction_cg(mode)
            dy, dx = mem.to_dev(y), mem.to_dev(np.zeros_like(y))
            info = ctx.cg_solve(grid, 0, 1, mem.ptr(dy), mem.ptr(dx), pc.C.Solve(1e-4, 0.0, 4000, 50, 10, 0))
            mem.sync()
            x = mem.to_host(dx).astype(np.float64)
            true_res = np.linalg.norm(y.astype(np.float64) - pc.O.masked_laplace(x, dom, None, None)) / np.linalg.norm(y.astype(np.float64))
            out[mode] = (info[0].iterations, bool(info[0].converged), true_res, x - x.mean())
            print(f"closed 512^2 fp32 rtol 1e-4, single-reduction mode {mode}: {info[0].iterations} iterations, converged {bool(info[0].converged)}, true relative residual {true_res:.3e}")
        assert out[0][1] and out[2][1]
        assert abs(out[2][0] - out[0][0]) <= 0.05 * out[0][0], (out[0][0], out[2][0])
        assert out[2][2] <= 2e-4
        assert np.linalg.norm(out[2][3] - out[0][3]) <= 2e-3 * np.linalg.norm(out[0][3])
    finally:
        ctx.set_small_grid_solver(True)
        ctx.set_single_reduction_cg(1)
