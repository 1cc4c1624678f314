"""
Pins the NumPy oracle (oracle/phi_oracle.py) against every known-answer / property test the REFERENCE holds for the hot
path (/root/reference tests/commit/...) and against two independent solvers (discrete-FFT Poisson,
SciPy sparse direct solve of the assembled operator). The reference itself cannot be imported (phiml absent).
"""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle import phi_oracle as O

PER, CLO, OPN = O.PERIODIC, O.CLOSED, O.OPEN


def test_reference_known_answer_self_advect_staggered():
    """ tests/commit/physics/test_advect.py:41-45
    v0 = StaggeredGrid(Box(x=(.9,2.6), y=(.9,2)), 0, x=4, y=3) * (0, 1); v = semi_lagrangian(v0, v0, 1)
    => v['x'] == 0 and v['y'] == [[0,0,0,0],[0,1,1,0]] (dims y,x) """
    dom = O.Domain((4, 3), (0, 0), (4, 3), ((CLO, CLO), (CLO, CLO)))
    vx = np.zeros((1,) + dom.comp_shape(0), np.float32)
    vy = np.zeros((1,) + dom.comp_shape(1), np.float32)
    # the Box mask sampled at the y-faces (x centres .5,1.5,2.5,3.5; y faces 1,2): inside for x in {1.5,2.5}
    vy[0, 1:3, :] = 1
    out = O.semi_lagrangian_staggered([vx, vy], [vx, vy], 1.0, dom)
    np.testing.assert_allclose(out[0], 0, atol=1e-7)
    np.testing.assert_allclose(out[1][0].T, [[0, 0, 0, 0], [0, 1, 1, 0]], rtol=1e-5)


@pytest.mark.parametrize("bc", [((CLO, CLO), (CLO, CLO)), ((OPN, OPN), (OPN, OPN)), ((PER, PER), (PER, PER))])
def test_reference_identity_advection(bc):
    """ tests/commit/physics/test_advect.py:12-18: adv(s, v, 0) == adv(s, v*0, 1) == s for centred and staggered fields """
    rng = np.random.default_rng(0)
    dom = O.Domain((4, 3), (0, 0), (4, 3), bc)
    v = [rng.standard_normal((1,) + dom.comp_shape(d)).astype(np.float32) for d in range(2)]
    zero = [np.zeros_like(a) for a in v]
    s = rng.standard_normal((1, 4, 3)).astype(np.float32)
    for a, b in zip(O.semi_lagrangian_staggered(v, v, 0.0, dom), v):
        np.testing.assert_allclose(a, b, atol=1e-5)
    for a, b in zip(O.semi_lagrangian_staggered(v, zero, 1.0, dom), v):
        np.testing.assert_allclose(a, b, atol=1e-5)
    np.testing.assert_allclose(O.semi_lagrangian_centered(s, v, 0.0, dom, bc), s, atol=1e-5)
    np.testing.assert_allclose(O.semi_lagrangian_centered(s, zero, 1.0, dom, bc), s, atol=1e-5)


def test_reference_staggered_storage_sizes():
    """ tests/commit/field/test__grid.py:25-36: x-component has x=19 (ZERO), 20 (PERIODIC), 21 (BOUNDARY) for x=20, y=10 """
    for code, expect in ((CLO, 19), (PER, 20), (OPN, 21)):
        dom = O.Domain((20, 10), (0, 0), (20, 10), ((code, code), (code, code)))
        assert dom.comp_shape(0) == (expect, 10)
        assert dom.comp_shape(1) == (20, expect - 10)


def test_reference_laplace_stencil_values():
    """ tests/commit/physics/test_diffuse.py:68-72: diffuse.explicit(impulse, 1, 1) on a ZERO-padded grid with dx = 1 gives
    [[0,1,0],[1,-3,1],[0,1,0]] (impulse + 5-point Laplacian). Same stencil applied here to a velocity component. """
    dom = O.Domain((4, 4), (0, 0), (4, 4), ((CLO, CLO), (CLO, CLO)))
    vx = np.zeros((1, 3, 4), np.float32)     # x-component faces: 3 x 4
    vx[0, 1, 1] = 1
    vy = np.zeros((1, 4, 3), np.float32)
    out = O.diffuse_explicit([vx, vy], 1.0, 1.0, dom)
    np.testing.assert_allclose(out[0][0, 0:3, 0:3], [[0, 1, 0], [1, -3, 1], [0, 1, 0]])
    np.testing.assert_allclose(out[1], 0)


def _buoyancy_faces(smoke, dom):
    sp_ = O.pad_scalar(smoke, [(0, 0), (1, 1)], dom.bc)
    faces = 0.5 * (sp_[:, :, 1:] + sp_[:, :, :-1])
    off, n = dom.face_offset(1), dom.comp_shape(1)[1]
    return faces[:, :, off:off + n]


@pytest.mark.parametrize("name,bc", [("closed", ((CLO, CLO), (CLO, CLO))), ("open", ((OPN, OPN), (OPN, OPN))),
                                     ("periodic", ((PER, PER), (PER, PER))), ("mixed", ((OPN, OPN), (CLO, OPN)))])
@pytest.mark.parametrize("batch", [1, 3])
def test_reference_make_incompressible_divergence_free(name, bc, batch):
    """ tests/commit/physics/test_fluid.py:19-53: 16x20 cells on Box[0:100, 0:100], smoke sphere source, two rounds of
    (buoyancy, make_incompressible with default Solve()); asserts |div| <= 5e-5 (test_fluid.py:28). Batched variants use
    different source positions per batch entry. """
    rng = np.random.default_rng(0)
    for dtype in (np.float32, np.float64):
        dom = O.Domain((16, 20), (0, 0), (100, 100), bc)
        cp = O.cell_positions(dom, np.float64)
        xs = rng.uniform(0, 100, size=batch)
        smoke = np.stack([(((cp[0] - x0) ** 2 + (cp[1] - 10) ** 2) <= 25).astype(dtype) for x0 in xs])
        v = [np.zeros((batch,) + dom.comp_shape(d), dtype) for d in range(2)]
        for _ in range(2):
            v[1] = v[1] + dtype(0.1) * _buoyancy_faces(smoke, dom).astype(dtype)
            v, p, info, _ = O.make_incompressible(v, dom, rtol=1e-5, atol=0.0)
        assert info.converged.all()
        assert np.abs(O.divergence(v, dom)).max() <= 5e-5


def test_cg_agrees_with_fft_poisson_solve_periodic():
    """ independent check: periodic discrete Poisson problem solved by FFT with the DISCRETE eigenvalues
    (cf. the FFT solver of tests/commit/test_poisson_solver.py:9-19, which uses the continuous spectrum) """
    rng = np.random.default_rng(1)
    n = (16, 12, 20)
    L = (2.0, 1.5, 3.0)
    dom = O.Domain(n, (0, 0, 0), L, ((PER, PER),) * 3)
    rhs = rng.standard_normal((1,) + n)
    rhs -= rhs.mean()
    x, info = O.cg(lambda p: O.masked_laplace(p, dom), rhs, np.zeros_like(rhs), 1e-12, 0, 5000)
    lam = 0
    for a in range(3):
        k = np.fft.fftfreq(n[a]) * 2 * np.pi
        shape = [1, 1, 1]; shape[a] = n[a]
        lam = lam + ((2 * np.cos(k) - 2) / dom.dx[a] ** 2).reshape(shape)
    lam[0, 0, 0] = np.inf
    x_fft = np.real(np.fft.ifftn(np.fft.fftn(rhs[0]) / lam))
    assert info.converged.all()
    np.testing.assert_allclose(x[0] - x[0].mean(), x_fft - x_fft.mean(), atol=1e-9)


def _assemble(dom, hard=None, active=None):
    """ dense probing of the oracle operator -> sparse matrix """
    N = int(np.prod(dom.res))
    cols = []
    for j in range(N):
        e = np.zeros((1, N)); e[0, j] = 1
        cols.append(O.masked_laplace(e.reshape((1,) + dom.res), dom, hard, active).reshape(N))
    return sp.csc_matrix(np.stack(cols, axis=1))


@pytest.mark.parametrize("bc", [((OPN, OPN), (OPN, OPN)), ((OPN, OPN), (CLO, OPN)), ((CLO, OPN), (PER, PER))])
def test_cg_agrees_with_sparse_direct_solve(bc):
    """ independent check on non-singular systems: assemble the operator by probing and solve with SciPy's direct solver """
    rng = np.random.default_rng(2)
    dom = O.Domain((7, 9), (0, 0), (3.0, 2.0), bc)
    A = _assemble(dom)
    assert abs(A - A.T).max() < 1e-12            # symmetric
    rhs = rng.standard_normal((1, 7, 9))
    x, info = O.cg(lambda p: O.masked_laplace(p, dom), rhs, np.zeros_like(rhs), 1e-13, 0, 5000)
    x_direct = spla.spsolve(A, rhs.reshape(-1)).reshape(7, 9)
    assert info.converged.all()
    np.testing.assert_allclose(x[0], x_direct, rtol=1e-8, atol=1e-10)


def test_obstacle_operator_is_symmetric_and_negative_semidefinite():
    dom = O.Domain((8, 8), (0, 0), (8, 8), ((CLO, CLO), (CLO, CLO)))
    active, hard, soft = O.obstacle_masks([O.BoxObstacle((3, 3), (5, 6))], dom, np.float64)
    A = _assemble(dom, hard, active).toarray()
    assert np.abs(A - A.T).max() < 1e-12
    act = active.reshape(-1) > 0
    w = np.linalg.eigvalsh(A[np.ix_(act, act)])
    assert w.max() < 1e-10
    assert np.allclose(A[~act][:, ~act], np.eye((~act).sum()))          # identity rows on inactive cells (fluid.py:202)
    # soft mask: 1 deep inside, 0 far away
    assert soft[0].max() == 1 and soft[0].min() == 0


def test_pressure_boundary_rules():
    """ fluid._pressure_extrapolation (fluid.py:264-274): closed wall -> zero normal gradient, open -> zero ghost """
    dom = O.Domain((4,  4), (0, 0), (4, 4), ((CLO, CLO), (OPN, OPN)))
    p = np.ones((1, 4, 4))
    g = O.pressure_gradient(p, dom)
    assert g[0].shape == (1, 3, 4) and np.all(g[0] == 0)
    assert g[1].shape == (1, 4, 5) and np.all(g[1][:, :, 0] == 1) and np.all(g[1][:, :, -1] == -1) and np.all(g[1][:, :, 1:-1] == 0)


def test_mixed_boundary_corner_padding_order():
    """ sequential padding (x, then y): outside both an x-constant and a y-constant side the LAST axis wins """
    bcv = np.zeros((2, 2, 2)); bcv[0, 1, 0] = 7.0; bcv[1, 1, 0] = 3.0
    dom = O.Domain((3, 3), (0, 0), (3, 3), ((CLO, CLO), (CLO, CLO)), bcv)
    a = np.zeros((1,) + dom.comp_shape(0))
    codes, consts = O._comp_codes(dom, 0)
    val = O.grid_sample(a, [np.array([[5.0]]), np.array([[5.0]])], codes, consts)
    assert val[0, 0] == 3.0
    padded = O.pad_component(a, 0, [(0, 2), (0, 2)], dom)
    assert padded[0, -1, -1] == 3.0 and padded[0, -1, 0] == 7.0


# ---- pins of the widened rows --------------------------------------------------------------------------
def test_reference_explicit_centered_diffusion_known_answer():
    """ tests/commit/physics/test_diffuse.py:68-72 exactly: a unit impulse on a 3 x 3 CenteredGrid with ZERO extrapolation,
    `diffuse.explicit(grid, 1, 1).values == [[0,1,0],[1,-3,1],[0,1,0]]` (dx = 1) """
    dom = O.Domain((3, 3), (0, 0), (3, 3), ((CLO, CLO), (CLO, CLO)))
    s = np.zeros((1, 3, 3), np.float32)
    s[0, 1, 1] = 1
    out = O.diffuse_explicit_centered(s, 1.0, 1.0, dom, ((CLO, CLO), (CLO, CLO)), [(0.0, 0.0)] * 2)
    np.testing.assert_allclose(out[0], [[0, 1, 0], [1, -3, 1], [0, 1, 0]], atol=1e-6)


@pytest.mark.parametrize("bc", [((CLO, CLO), (CLO, CLO)), ((OPN, OPN), (OPN, OPN)), ((PER, PER), (PER, PER))])
def test_reference_mac_cormack_identity(bc):
    """ tests/commit/physics/test_advect.py:29-30 (`_test_advection(advect.mac_cormack)`): adv(f, v, 0) == adv(f, v*0, 1) == f
    for centred and staggered fields """
    rng = np.random.default_rng(0)
    dom = O.Domain((4, 3), (0, 0), (4, 3), bc)
    v = [rng.standard_normal((1,) + dom.comp_shape(d)).astype(np.float32) for d in range(2)]
    zero = [np.zeros_like(a) for a in v]
    s = rng.standard_normal((1, 4, 3)).astype(np.float32)
    for a, b in zip(O.mac_cormack_staggered(v, v, 0.0, dom), v):
        np.testing.assert_allclose(a, b, atol=1e-5)
    for a, b in zip(O.mac_cormack_staggered(v, zero, 1.0, dom), v):
        np.testing.assert_allclose(a, b, atol=1e-5)
    np.testing.assert_allclose(O.mac_cormack_centered(s, v, 0.0, dom, bc), s, atol=1e-5)
    np.testing.assert_allclose(O.mac_cormack_centered(s, zero, 1.0, dom, bc), s, atol=1e-5)


def test_mac_cormack_is_second_order_and_bounded():
    """ properties the scheme is built for (advect.py:188-190): on a smooth periodic field advected by a constant velocity the
    error is well below semi-Lagrangian's, and the result never leaves the range of the field (limiter) """
    n = 64
    dom = O.Domain((n, n), (0, 0), (1, 1), ((PER, PER), (PER, PER)))
    x = (np.arange(n) + 0.5) / n
    s = (np.sin(2 * np.pi * x)[:, None] * np.cos(2 * np.pi * x)[None, :]).astype(np.float64)[None]
    u, w, dt = 0.37, -0.21, 0.4 / n * 3
    v = [np.full((1,) + dom.comp_shape(0), u), np.full((1,) + dom.comp_shape(1), w)]
    exact = (np.sin(2 * np.pi * (x - u * dt))[:, None] * np.cos(2 * np.pi * (x - w * dt))[None, :])[None]
    err_sl = np.abs(O.semi_lagrangian_centered(s, v, dt, dom, dom.bc) - exact).mean()
    mc = O.mac_cormack_centered(s, v, dt, dom, dom.bc)
    assert np.abs(mc - exact).mean() < 0.25 * err_sl          # (at the extrema the limiter falls back to first order)
    assert mc.max() <= s.max() + 1e-12 and mc.min() >= s.min() - 1e-12


def test_centered_to_staggered_resample_values():
    """ `smoke * (0, 0.1) @ velocity` (tests/commit/physics/test_fluid.py:26): face values are means of the adjacent cells, wall
    faces of a closed domain are not stored, open domains store N + 1 faces with the edge value copied outside """
    s = np.arange(12, dtype=np.float32).reshape(1, 4, 3)
    closed = O.Domain((4, 3), (0, 0), (4, 3), ((CLO, CLO), (CLO, CLO)))
    out = O.centered_to_staggered(s, closed, ((OPN, OPN), (OPN, OPN)), None, (0.0, 0.1))
    assert out[0].shape == (1, 3, 3) and out[1].shape == (1, 4, 2) and np.all(out[0] == 0)
    np.testing.assert_allclose(out[1][0], 0.1 * 0.5 * (s[0, :, 1:] + s[0, :, :-1]), rtol=1e-6)
    opened = O.Domain((4, 3), (0, 0), (4, 3), ((OPN, OPN), (OPN, OPN)))
    out = O.centered_to_staggered(s, opened, ((OPN, OPN), (OPN, OPN)), None, (1.0, 1.0))
    assert out[0].shape == (1, 5, 3)
    np.testing.assert_allclose(out[0][0, 0], s[0, 0])            # zero-gradient ghost cell: mean of a value with itself
    np.testing.assert_allclose(out[0][0, 2], 0.5 * (s[0, 1] + s[0, 2]))


def test_obstacle_boundary_conditions_known_values():
    """ fluid.apply_boundary_conditions (fluid.py:212-240): deep inside a moving obstacle the fluid takes the obstacle's
    velocity (linear + angular x r), far away it is untouched, the transition is one cell wide """
    dom = O.Domain((32, 32), (0, 0), (32, 32), ((CLO, CLO), (CLO, CLO)))
    ob = O.BoxObstacle((10, 10), (22, 22), velocity=(1.5, -0.5), angular_velocity=0.25)
    v = [np.full((1,) + dom.comp_shape(d), 7.0) for d in range(2)]
    out = O.apply_boundary_conditions(v, [ob], dom)
    fx = O.face_positions(0, dom, np.float64)
    inside = (np.abs(fx[0] - 16) < 4) & (np.abs(fx[1] - 16) < 4)
    np.testing.assert_allclose(out[0][0][inside], (1.5 - 0.25 * (fx[1] - 16))[inside])          # u_x = U_x - w (y - c_y)
    far = (np.abs(fx[0] - 16) > 8) | (np.abs(fx[1] - 16) > 8)
    assert np.all(out[0][0][far] == 7.0)
    active, hard, soft = O.obstacle_masks([ob], dom)
    assert active[0].sum() == 32 * 32 - 12 * 12


def test_assembled_csr_operator_equals_the_matrix_free_stencil():
    """ oracle.laplace_csr (the sparse matrix PhiML would trace from masked_laplace, fluid.py:165) == oracle.masked_laplace for every
    boundary kind, incl. one-cell axes; and CG on either form gives the same pressure """
    rng = np.random.default_rng(41)
    cases = [((6, 5), ((O.PERIODIC, O.PERIODIC), (O.CLOSED, O.OPEN))), ((4, 5, 3), ((O.OPEN, O.CLOSED), (O.PERIODIC, O.PERIODIC), (O.CLOSED, O.CLOSED))),
             ((1, 4), ((O.OPEN, O.OPEN), (O.PERIODIC, O.PERIODIC))), ((3, 3, 3), ((O.OPEN, O.OPEN),) * 3)]
    for res, bc in cases:
        dom = O.Domain(res, (0.0,) * len(res), tuple(0.7 * r for r in res), bc)
        A = O.laplace_csr(dom, np.float64)
        p = rng.standard_normal((2,) + res)
        ref = O.masked_laplace(p, dom, None, None)
        got = np.stack([(A @ q.ravel()).reshape(res) for q in p])
        assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max()
        assert A.nnz <= (2 * len(res) + 1) * int(np.prod(res)) and abs(A - A.T).max() < 1e-15          # symmetric 5/7-point matrix
    dom = O.Domain((8, 6, 5), (0, 0, 0), (1, 1, 1), ((O.OPEN, O.OPEN), (O.CLOSED, O.OPEN), (O.PERIODIC, O.PERIODIC)))
    A = O.laplace_csr(dom, np.float64)
    y = rng.standard_normal((1, 8, 6, 5))
    x1, i1 = O.cg(lambda q: O.masked_laplace(q, dom, None, None), y, np.zeros_like(y), 1e-10, 0, 500)
    x2, i2 = O.cg(lambda q: (A @ q.ravel()).reshape(q.shape), y, np.zeros_like(y), 1e-10, 0, 500)
    assert int(i1.iterations[0]) == int(i2.iterations[0]) and np.abs(x1 - x2).max() <= 1e-9 * np.abs(x1).max()
