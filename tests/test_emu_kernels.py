"""
Kernel LOGIC tests without a GPU: the HIP sources of phiflow_amd/csrc are compiled with g++ against a fiber emulation of
the HIP execution model (tests/hipemu) and driven through the same C ABI + ctypes binding as on the GPU; results are
compared with the NumPy oracle. This guards indexing / halo / boundary-condition logic; the `-m gpu` tests in
test_gpu_parity.py repeat the same cases on the real gfx950 library.
"""
import numpy as np
import pytest

import parity_cases as pc
from parity_cases import CLO, OPN, PER

MEM = pc.NumpyMem()

GRIDS_2D = [
    ((16, 20), ((CLO, CLO), (CLO, CLO))),
    ((16, 20), ((OPN, OPN), (OPN, OPN))),
    ((16, 20), ((PER, PER), (PER, PER))),
    ((16, 20), ((OPN, OPN), (CLO, OPN))),      # combine_sides(x=BOUNDARY, y=(ZERO, BOUNDARY)), test_fluid.py:51
    ((7, 13), ((CLO, OPN), (PER, PER))),       # ragged: scalar fallback path (n2 % 4 != 0)
    ((10, 18), ((OPN, CLO), (PER, PER))),      # fp32 rows of even length: the 8-byte-vector instantiation (V = 2)
]
GRIDS_3D = [
    ((8, 12, 16), ((PER, PER), (PER, PER), (PER, PER))),
    ((9, 7, 10), ((CLO, CLO), (OPN, OPN), (CLO, OPN))),
    ((6, 20, 72), ((CLO, OPN), (PER, PER), (CLO, CLO))),   # more than one tile along a2, partial tiles
    ((3, 5, 264), ((PER, PER), (CLO, CLO), (OPN, OPN))),   # open fast axis: rows of n2 + 1 faces, two patches of the vector kernels
    ((4, 5, 24), ((OPN, OPN), (OPN, CLO), (OPN, CLO))),    # lower face stored, upper wall face not: rows of n2 faces without wrap
    ((5, 6, 134), ((PER, PER), (CLO, OPN), (CLO, CLO))),   # fp32 V = 2 (134 % 4 = 2): two 128-cell tiles per row, the second nearly empty
]


@pytest.mark.parametrize("res,bc", GRIDS_2D + GRIDS_3D)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_stencils_match_oracle(emu_ctx, res, bc, dtype):
    rng = np.random.default_rng(1)
    dom, grid = pc.make_case(res, bc, dtype, batch=2)
    pc.check_component_shapes(emu_ctx, dom, grid)
    pc.check_laplace(emu_ctx, MEM, dom, grid, dtype, rng)
    pc.check_divergence(emu_ctx, MEM, dom, grid, dtype, rng, balance=False)
    pc.check_divergence(emu_ctx, MEM, dom, grid, dtype, rng, balance=True)
    pc.check_divergence_flags(emu_ctx, MEM, dom, grid, dtype, rng)
    pc.check_grad_subtract(emu_ctx, MEM, dom, grid, dtype, rng)
    pc.check_grad_subtract_flags(emu_ctx, MEM, dom, grid, dtype, rng)
    pc.check_diffuse(emu_ctx, MEM, dom, grid, dtype, rng)


@pytest.mark.parametrize("res,bc", GRIDS_2D + GRIDS_3D)
def test_advection_matches_oracle(emu_ctx, res, bc):
    rng = np.random.default_rng(2)
    for dtype in (np.float32, np.float64):
        dom, grid = pc.make_case(res, bc, dtype, batch=2)
        pc.check_advect_staggered(emu_ctx, MEM, dom, grid, dtype, rng, dt=0.7)
        pc.check_advect_staggered(emu_ctx, MEM, dom, grid, dtype, rng, dt=2.9)   # CFL ~ 3: taps several cells away
        s_codes = tuple((PER, PER) if lo == PER else (OPN, CLO) for lo, hi in bc)
        s_consts = [(0.0, 0.25)] * len(res)
        pc.check_advect_centered(emu_ctx, MEM, dom, grid, dtype, rng, s_codes, s_consts)


@pytest.mark.parametrize("res,bc", [GRIDS_2D[0], GRIDS_2D[2], GRIDS_2D[3], GRIDS_3D[0], GRIDS_3D[1], GRIDS_3D[2], ((12, 16, 72), ((PER, PER), (PER, PER), (OPN, OPN))),
                                    ((20, 24, 64), ((CLO, CLO), (PER, PER), (PER, PER)))])
def test_advection_paths_give_the_same_bits(emu_ctx, res, bc):
    """ r6: tile (both fills, both reaches), fix-up list, windows and gather kernels evaluate ONE arithmetic per sample -- bit for bit, also where a
    displacement is a tiny negative number (v_fract vs x - floor(x)) """
    rng = np.random.default_rng(66)
    s_codes = tuple((PER, PER) if lo == PER else (OPN, CLO) for lo, hi in bc)
    s_consts = [(0.0, 0.25)] * len(res)
    for dtype in (np.float32, np.float64):
        bcv = None
        if any(side == CLO for pair in bc for side in pair):
            pad = 3 - len(res)
            bcv = [[[float(rng.normal()) * 0.05 if a >= pad and bc[a - pad][sd] == CLO else 0.0 for c in range(3)] for sd in range(2)] for a in range(3)]
        dom, grid = pc.make_case(res, bc, dtype, batch=2, bc_val=bcv if len(res) == 3 else None)
        for dt in (0.7, 2.3):
            pc.check_advect_paths_same_bits(emu_ctx, MEM, dom, grid, dtype, rng, s_codes, s_consts, dt=dt)


def test_adaptive_reach_is_deterministic(emu_library):
    """ r5: the reach of an LDS-staged advection pass follows the fallback fraction of ONE named earlier pass, read a fixed number of passes after it (capi.hip
    adv_choose) -- not of whatever pass had completed when the host looked. Pinned here: with a quarter of the field moving 1.6 cells per step the
    self-advection leaves the narrow window at the FOURTH pass (pass 1 observed, resolved before pass 4), on every fresh context alike; the passes before it
    carry the bits of the fixed narrow reach, the ones after it those of the fixed wide reach; a gentle field never switches. """
    from phiflow_amd import _capi as C
    rng = np.random.default_rng(17)
    dom, grid = pc.make_case((16, 24, 64), ((PER, PER),) * 3, np.float32, batch=1)
    v = pc.random_velocity(dom, 1, np.float32, rng, 1.0)
    vmax = max(float(np.abs(a).max()) for a in v)
    gentle = [a * np.float32(0.8 / vmax) for a in v]
    fast = [a.copy() for a in gentle]
    for a in fast:
        a[:, :4] *= np.float32(2.0)                  # the first four planes of every component move up to 1.6 cells: ~ a quarter of the (tile, plane) units
    def run(ctx, field, halo, passes):
        """ -> per pass: (result, units the fix-up list redid) -- the second tells WHICH reach ran: 1.6 cells leave the narrow window and stay inside the wide one """
        ctx.set_advect_halo(halo)
        dv = [MEM.to_dev(a) for a in field]
        outs = []
        for _ in range(passes):
            dout = [MEM.empty(a.shape, np.float32) for a in field]
            ctx.advect_staggered(grid, [MEM.ptr(a) for a in dv], [MEM.ptr(a) for a in dv], [MEM.ptr(a) for a in dout], 1.0)
            MEM.sync()
            outs.append((np.concatenate([MEM.to_host(a).ravel() for a in dout]).copy(), ctx.advect_fallback_stats()[0]))
        return outs
    narrow, redone_narrow = run(C.Context(emu_library, 0), fast, 1, 1)[0]
    wide, redone_wide = run(C.Context(emu_library, 0), fast, 2, 1)[0]
    assert redone_narrow > 0 and redone_wide == 0
    # r6: the two reaches (and the fix-up list) evaluate ONE arithmetic -- until r5 they agreed to rounding only, and the reach of a pass showed in its last bits
    assert np.array_equal(narrow, wide)
    runs = [run(C.Context(emu_library, 0), fast, -1, 9) for _ in range(2)]
    for outs in runs:
        assert all(r > 0 for _, r in outs[:3]), "passes 1-3 run with the narrow reach"
        assert all(r == 0 for _, r in outs[3:]), "from pass 4 on the wide reach"
        assert all(np.array_equal(o, narrow) for o, _ in outs), "the same bits whatever the reach"
    calm = run(C.Context(emu_library, 0), gentle, -1, 9)
    assert all(np.array_equal(o, calm[0][0]) and r == 0 for o, r in calm)
    # r6: one policy PER GRID -- two grids that alternate on one context (SlabFluid's whole-slab and window passes, two simulations) each keep
    # their history; until r5 every change of grid restarted the policy and the reach never left narrow
    dom2, grid2 = pc.make_case((16, 24, 32), ((PER, PER),) * 3, np.float32, batch=1)
    v2 = pc.random_velocity(dom2, 1, np.float32, rng, 1.0)
    gentle2 = [a * np.float32(0.8 / max(float(np.abs(b).max()) for b in v2)) for a in v2]
    ctx = C.Context(emu_library, 0)
    ctx.set_advect_halo(-1)
    dv, dv2 = [MEM.to_dev(a) for a in fast], [MEM.to_dev(a) for a in gentle2]
    redone = []
    for _ in range(9):
        for g, d, field in ((grid, dv, fast), (grid2, dv2, gentle2)):
            dout = [MEM.empty(a.shape, np.float32) for a in field]
            ctx.advect_staggered(g, [MEM.ptr(a) for a in d], [MEM.ptr(a) for a in d], [MEM.ptr(a) for a in dout], 1.0)
            MEM.sync()
            if g is grid:
                redone.append(ctx.advect_fallback_stats()[0])
    assert all(r > 0 for r in redone[:3]) and all(r == 0 for r in redone[3:]), redone


@pytest.mark.parametrize("res,bc,dma32,dma64", [
    ((8, 12, 16), ((PER, PER), (PER, PER), (PER, PER)), True, True),
    ((6, 10, 64), ((PER, PER), (PER, PER), (PER, PER)), True, True),
    ((5, 9, 136), ((OPN, OPN), (PER, PER), (PER, PER)), True, True),      # open slow axis (clamped planes), three tiles along the fast axis, the last one partial
    ((9, 11, 24), ((PER, PER), (OPN, OPN), (PER, PER)), True, True),      # open rows: n1 + 1 faces of the a1 component, clamped halo rows, ragged last tile row
    ((7, 8, 18), ((PER, PER), (PER, PER), (PER, PER)), False, True),      # periodic rows of 18 cells: not whole fp32 vectors -> the register-staged kernel (fp64: regular)
    # r5, second step -- the GEN instantiation: constants from the table, patch elements, face offsets
    ((6, 8, 16), ((PER, PER), (PER, PER), (OPN, OPN)), True, True),       # open fast axis: rows of 17 faces (a straddling chunk per row), halo columns = copies of the edge
    ((6, 8, 16), ((CLO, CLO), (PER, PER), (PER, PER)), True, True),       # closed slow axis: constant planes
    ((9, 7, 20), ((CLO, CLO), (CLO, CLO), (CLO, CLO)), True, True),       # the closed box: 19 faces per row of the a2 component, constant rows / planes / chunks
    ((5, 13, 72), ((CLO, OPN), (OPN, CLO), (CLO, OPN)), True, True),      # mixed sides, two tiles along the fast axis
    ((4, 9, 67), ((OPN, CLO), (CLO, CLO), (OPN, OPN)), True, True),       # ragged cell rows (67): every component straddles; 68 faces of the a2 component
    ((6, 5, 130), ((CLO, CLO), (PER, PER), (CLO, CLO)), True, True),      # 129 faces: the straddling chunk in the third tile; periodic rows wrap onto the last row
])
def test_self_advection_lds_dma_fill(emu_ctx, res, bc, dma32, dma64):
    rng = np.random.default_rng(41)
    for dtype, expect in ((np.float32, dma32), (np.float64, dma64)):
        bcv = None
        if any(side == CLO for pair in bc for side in pair):      # walls that move (tangentially and, for the constants' sake, normally too)
            bcv = [[[float(rng.normal()) * 0.05 if bc[a][s] == CLO else 0.0 for c in range(3)] for s in range(2)] for a in range(3)]      # (small: the "gentle" fields must stay below one cell per step at the walls too)
        dom, grid = pc.make_case(res, bc, dtype, batch=2, bc_val=bcv)
        pc.check_advect_self_dma(emu_ctx, MEM, dom, grid, dtype, rng, dt=0.7, expect_dma=expect)
        pc.check_advect_self_dma(emu_ctx, MEM, dom, grid, dtype, rng, dt=2.1, expect_dma=expect)


@pytest.mark.parametrize("res,bc,dma", [
    ((8, 12, 16), ((PER, PER), (PER, PER), (PER, PER)), True),
    ((6, 20, 72), ((PER, PER), (PER, PER), (PER, PER)), True),        # two tiles along the fast axis (the second partial), three along a1
    ((13, 11, 24), ((OPN, OPN), (PER, PER), (PER, PER)), True),       # open slow axis: clamped planes, n0 + 1 faces of the a0 component, two chunks
    ((9, 19, 64), ((PER, PER), (OPN, OPN), (PER, PER)), True),        # open rows: n1 + 1 faces of the a1 component, clamped halo rows, ragged last tile row
    ((7, 8, 18), ((PER, PER), (PER, PER), (PER, PER)), None),         # rows of 18 cells: fp32 not whole vectors -> register-staged windows; fp64 regular
    ((6, 10, 64), ((PER, PER), (PER, PER), (OPN, OPN)), False),       # open fast axis: not regular
    ((6, 10, 64), ((CLO, CLO), (PER, PER), (PER, PER)), False),       # a closed side: constants
])
def test_mac_cormack_windows_lds_dma_fill(emu_ctx, res, bc, dma):
    rng = np.random.default_rng(43)
    for dtype in (np.float32, np.float64):
        expect = (dtype == np.float64) if dma is None else dma
        dom, grid = pc.make_case(res, bc, dtype, batch=2)
        emu_ctx.set_advect_chunk(5 if res[0] > 8 else 0)
        try:
            pc.check_mac_cormack_staggered_dma(emu_ctx, MEM, dom, grid, dtype, rng, dt=0.7, expect_dma=expect)
            pc.check_mac_cormack_staggered_dma(emu_ctx, MEM, dom, grid, dtype, rng, dt=2.1, expect_dma=expect)
        finally:
            emu_ctx.set_advect_chunk(0)


@pytest.mark.parametrize("res,bc", GRIDS_2D + GRIDS_3D)
def test_mac_cormack_and_resample_match_oracle(emu_ctx, res, bc):
    """ advect.mac_cormack (centred + staggered) and the centred -> staggered resample used for buoyancy """
    rng = np.random.default_rng(12)
    for dtype in (np.float32, np.float64):
        dom, grid = pc.make_case(res, bc, dtype, batch=2)
        s_codes = tuple((PER, PER) if lo == PER else (OPN, CLO) for lo, hi in bc)
        s_consts = [(0.0, 0.25)] * len(res)
        pc.check_mac_cormack_centered(emu_ctx, MEM, dom, grid, dtype, rng, s_codes, s_consts)
        pc.check_mac_cormack_centered(emu_ctx, MEM, dom, grid, dtype, rng, s_codes, s_consts, dt=2.3, strength=0.6)
        pc.check_mac_cormack_staggered(emu_ctx, MEM, dom, grid, dtype, rng)
        pc.check_centered_to_staggered(emu_ctx, MEM, dom, grid, dtype, rng, s_codes, s_consts)
        swapped = tuple((PER, PER) if lo == PER else (CLO, OPN) for lo, hi in bc)      # constant below, zero-gradient above
        pc.check_centered_to_staggered(emu_ctx, MEM, dom, grid, dtype, rng, swapped, [(0.4, 0.0)] * len(res))


@pytest.mark.parametrize("res,bc,dtype", [(r, b, np.float32) for r, b in GRIDS_2D + GRIDS_3D[:2]] + [(r, b, np.float64) for r, b in (GRIDS_2D[3], GRIDS_2D[4], GRIDS_3D[1])])
def test_implicit_diffusion_matches_oracle(emu_ctx, res, bc, dtype):
    """ diffuse.implicit (phi/physics/diffuse.py:63-92): the CG kernels of the pressure path with the operator I - k dt L on the field's lattice """
    rng = np.random.default_rng(21)
    D = len(res)
    bc_val = rng.uniform(-0.5, 0.5, (D, 2, D))           # wall values (tangential ones matter; a lid): the affine part of the operator
    dom, grid = pc.make_case(res, bc, dtype, batch=2, bc_val=bc_val)
    s_codes = tuple((PER, PER) if lo == PER else (OPN, CLO) for lo, hi in bc)
    pc.check_diffuse_implicit(emu_ctx, MEM, dom, grid, dtype, rng, s_codes, [(0.0, 0.25)] * D)


@pytest.mark.parametrize("res,bc", GRIDS_2D + GRIDS_3D[:3])
def test_adjoint_kernels_match_oracle_derivatives(emu_ctx, res, bc):
    """ backward kernels vs finite differences / linear responses of the oracle's forward functions (fp64) """
    rng = np.random.default_rng(14)
    dom, grid = pc.make_case(res, bc, np.float64, batch=2)
    s_codes = tuple((PER, PER) if lo == PER else (OPN, CLO) for lo, hi in bc)
    s_consts = [(0.0, 0.25)] * len(res)
    pc.check_advect_backward(emu_ctx, MEM, dom, grid, rng, s_codes, s_consts)
    pc.check_advect_backward(emu_ctx, MEM, dom, grid, rng, s_codes, s_consts, dt=0.2)      # CFL < 1 everywhere: every scatter goes through the LDS windows
    pc.check_project_backward(emu_ctx, MEM, dom, grid, rng)
    pc.check_mac_cormack_and_diffuse_backward(emu_ctx, MEM, dom, grid, rng, s_codes, s_consts)


def test_adjoint_next_to_a_lookup_kink(emu_ctx):
    """ the round-3 GPU observation (fuzz seed 40062) as a constructed, asserted case """
    pc.check_adjoint_next_to_a_lookup_kink(emu_ctx, MEM)


def test_adjoint_projection_with_obstacles(emu_ctx):
    rng = np.random.default_rng(15)
    dom, grid = pc.make_case((12, 10, 16), ((CLO, CLO),) * 3, np.float64, batch=1)
    pc.check_project_backward(emu_ctx, MEM, dom, grid, rng, obstacles=[pc.O.BoxObstacle((4.0, 3.0, 5.0), (8.0, 7.0, 11.0))])
    dom, grid = pc.make_case((16, 20), ((PER, PER), (CLO, OPN)), np.float64, batch=2)
    pc.check_project_backward(emu_ctx, MEM, dom, grid, rng, obstacles=[pc.O.SphereObstacle((8.0, 9.0), 3.5)])


@pytest.mark.parametrize("res,bc", GRIDS_3D)
def test_slab_halo_planes(emu_ctx, res, bc):
    rng = np.random.default_rng(16)
    for dtype in (np.float32, np.float64):
        dom, _ = pc.make_case(res, bc, dtype, batch=1)
        pc.check_slab_halo_planes(emu_ctx, MEM, dom, dtype, rng, parts=2 if res[0] < 9 else 3)


def test_advection_with_wall_velocity(emu_ctx):
    """ lid-driven cavity boundary: tangential wall velocity on one side (Lid_Driven_Cavity.ipynb cell 5) """
    rng = np.random.default_rng(3)
    bcv = np.zeros((3, 2, 3)); bcv[2, 1, 0] = 1.0; bcv[0, 0, 0] = 0.3   # z+ lid moves in x; x- inflow
    for dtype in (np.float32, np.float64):
        dom, grid = pc.make_case((8, 8, 8), ((CLO, CLO), (CLO, CLO), (CLO, CLO)), dtype, batch=1, bc_val=bcv)
        pc.check_advect_staggered(emu_ctx, MEM, dom, grid, dtype, rng, dt=1.3)
        pc.check_divergence(emu_ctx, MEM, dom, grid, dtype, rng, balance=False)
        pc.check_diffuse(emu_ctx, MEM, dom, grid, dtype, rng)


@pytest.mark.parametrize("res,bc", GRIDS_2D + GRIDS_3D[:2])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_cg_matches_oracle(emu_ctx, res, bc, dtype):
    rng = np.random.default_rng(4)
    dom, grid = pc.make_case(res, bc, dtype, batch=2)
    try:
        for small in (True, False):     # single-kernel solver for small grids (cg_small.hip) and the marching kernels
            emu_ctx.set_small_grid_solver(small)
            pc.check_cg(emu_ctx, MEM, dom, grid, dtype, np.random.default_rng(4))
    finally:
        emu_ctx.set_small_grid_solver(True)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_cg_deferred_x_update(emu_ctx, dtype):
    """ the marching 'CG' updates x every other iteration (UPDATE_R / UPDATE_X2, phihip_set_deferred_x_update): same solution as the
    plain update for odd and even iteration counts, an odd refresh period, early exits of single batch entries (flush of the pending
    step) -- each against the oracle and against the plain path """
    dom, grid = pc.make_case((8, 12, 16), ((CLO, OPN), (PER, PER), (CLO, CLO)), dtype, batch=3)
    try:
        emu_ctx.set_small_grid_solver(False)
        for kwargs in (dict(max_iter=7, fixed_iterations=True), dict(max_iter=10, refresh=3, fixed_iterations=True),
                       dict(max_iter=9, refresh=4, fixed_iterations=True), dict(rtol=1e-3)):
            xs = []
            for defer in (True, False):
                emu_ctx.set_deferred_x_update(defer)
                x, info = pc.check_cg(emu_ctx, MEM, dom, grid, dtype, np.random.default_rng(8), **kwargs)
                xs.append((x, [i.iterations for i in info]))
            assert xs[0][1] == xs[1][1]                                  # the recurrence does not see x
            assert pc.rel_l2(xs[0][0], xs[1][0]) <= (5e-6 if dtype == np.float32 else 1e-11)    # recovering d_k costs a few digits of x
    finally:
        emu_ctx.set_small_grid_solver(True)
        emu_ctx.set_deferred_x_update(True)


def test_single_kernel_solver_16384_cells(emu_ctx):
    """ the 1024-thread x 16-cell form of cg_small (8193 ... 16384 cells, fp32): chosen for batches of >= 8 entries, here forced """
    try:
        emu_ctx.set_small_grid_solver(16384)
        for res, bc in (((96, 100), ((CLO, OPN), (PER, PER))), ((24, 20, 25), ((CLO, CLO), (OPN, OPN), (PER, PER)))):
            dom, grid = pc.make_case(res, bc, np.float32, batch=2)
            pc.check_cg(emu_ctx, MEM, dom, grid, np.float32, np.random.default_rng(4), rtol=1e-4)
            pc.check_cg(emu_ctx, MEM, dom, grid, np.float32, np.random.default_rng(5), max_iter=9, refresh=4, fixed_iterations=True, adaptive=True)
    finally:
        emu_ctx.set_small_grid_solver(True)


def test_cg_fixed_iterations_and_refresh(emu_ctx):
    """ benchmark mode: tolerances 0, exactly max_iterations; refresh every 7 exercises the true-residual branch """
    rng = np.random.default_rng(5)
    dom, grid = pc.make_case((8, 8, 16), ((PER, PER),) * 3, np.float32, batch=2)
    pc.check_cg(emu_ctx, MEM, dom, grid, np.float32, rng, max_iter=20, refresh=7, fixed_iterations=True)


@pytest.mark.parametrize("res,bc,dtype", [GRIDS_2D[0] + (np.float32,), GRIDS_2D[-1] + (np.float64,), GRIDS_3D[0] + (np.float64,), GRIDS_3D[1] + (np.float32,)])
def test_cg_adaptive_matches_oracle(emu_ctx, res, bc, dtype):
    """ Solve('CG-adaptive') (Fluid_Logo.ipynb): both solvers, tolerance mode and fixed iterations + refresh """
    dom, grid = pc.make_case(res, bc, dtype, batch=2)
    try:
        for small in (True, False):
            emu_ctx.set_small_grid_solver(small)
            pc.check_cg(emu_ctx, MEM, dom, grid, dtype, np.random.default_rng(4), refresh=20, adaptive=True)
            pc.check_cg(emu_ctx, MEM, dom, grid, dtype, np.random.default_rng(6), max_iter=12, refresh=5, fixed_iterations=True, adaptive=True)
    finally:
        emu_ctx.set_small_grid_solver(True)


@pytest.mark.parametrize("res,bc", pc.DEGENERATE_GRIDS[::2] + [pc.DEGENERATE_GRIDS[5]])
def test_degenerate_grids(emu_ctx, res, bc):
    """ one / two cells along an axis, 3-D grids with a single plane (whose a0 boundary rule still applies) """
    pc.check_degenerate_grid(emu_ctx, MEM, res, bc, np.float32, projection=False)


def test_batch_entries_converge_independently(emu_ctx):
    """ per-batch alpha / beta / stop (PhiML batch dims): a zero rhs entry stops at iteration 0, the other runs on """
    dtype = np.float32
    dom, grid = pc.make_case((16, 16), ((CLO, CLO), (CLO, CLO)), dtype, batch=2)
    rng = np.random.default_rng(6)
    rhs = pc.O.balance_divergence(rng.standard_normal((2, 16, 16)).astype(dtype), None)
    rhs[0] = 0
    x = np.zeros_like(rhs)
    info = emu_ctx.cg_solve(grid, 0, 1, rhs.ctypes.data, x.ctypes.data, pc.solve_params(dtype))
    assert info[0].iterations == 0 and info[0].converged == 1
    assert info[1].iterations > 5 and info[1].converged == 1
    assert np.all(x[0] == 0)


def test_cg_reports_not_converged_and_diverged(emu_ctx):
    dtype = np.float32
    dom, grid = pc.make_case((16, 16), ((PER, PER), (PER, PER)), dtype, batch=1)
    rng = np.random.default_rng(7)
    rhs = pc.O.balance_divergence(rng.standard_normal((1, 16, 16)).astype(dtype), None)
    x = np.zeros_like(rhs)
    info = emu_ctx.cg_solve(grid, 0, 1, rhs.ctypes.data, x.ctypes.data, pc.solve_params(dtype, max_iter=3))
    assert info[0].iterations == 3 and not info[0].converged and not info[0].diverged
    rhs2 = rhs + 1.0                      # inconsistent rhs on a singular (periodic) system
    x = np.zeros_like(rhs)
    info = emu_ctx.cg_solve(grid, 0, 1, rhs2.ctypes.data, x.ctypes.data, pc.solve_params(dtype, max_iter=300, rtol=1e-9))
    assert not info[0].converged


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_obstacles(emu_ctx, dtype):
    """ config-5 style: closed box with a solid box obstacle; flags kernel, masked Laplacian, CG and the projection """
    rng = np.random.default_rng(8)
    dom, grid = pc.make_case((12, 10, 16), ((CLO, CLO),) * 3, dtype, batch=1)
    obstacles = [pc.O.BoxObstacle((4.0, 3.0, 5.0), (8.0, 7.0, 11.0))]
    active, hard, soft = pc.O.obstacle_masks(obstacles, dom, dtype)
    acc = (active[0] > 0).astype(np.uint8)
    flags = np.zeros(dom.res, np.uint8)
    g1 = pc.C.make_grid(3, grid.dtype, 1, dom.res, dom.lower, dom.upper, dom.bc, dom.bc_val)
    emu_ctx.build_cellflags(g1, acc.ctypes.data, 0, 1, flags.ctypes.data)
    # flags agree with the oracle's hard_bcs / active
    assert np.array_equal((flags >> 6) & 1, acc)
    for d in range(3):
        lo_bit = (flags >> (2 * d)) & 1
        h = hard[d][0]      # faces 1..N-1 (closed)
        sl = [slice(None)] * 3; sl[d] = slice(1, None)
        assert np.array_equal(lo_bit[tuple(sl)], h.astype(np.uint8))
    pc.check_laplace(emu_ctx, MEM, dom, grid, dtype, rng, flags_np=flags, hard=hard, active=active)
    pc.check_cg(emu_ctx, MEM, dom, grid, dtype, rng, flags_np=flags, hard=hard, active=active)
    pc.check_make_incompressible(emu_ctx, MEM, dom, grid, dtype, rng, obstacles=obstacles)


def test_cellflags_byte_parallel_kernel(emu_ctx):
    """ r5: the byte-parallel flag kernel (16 / 4 cells per thread) and the scalar one on random masks with arbitrary non-zero bytes, every
    boundary kind per side, 2-D and 3-D, per-batch masks, with and without a user `active` mask """
    rng = np.random.default_rng(31)
    for res, bc in (((5, 9, 32), ((PER, PER), (CLO, OPN), (OPN, CLO))), ((4, 6, 80), ((CLO, CLO), (PER, PER), (PER, PER))), ((7, 64), ((OPN, OPN), (CLO, CLO))),
                    ((6, 5, 20), ((OPN, CLO), (PER, PER), (CLO, OPN))), ((9, 12), ((PER, PER), (PER, PER))), ((3, 4, 18), ((CLO, OPN), (OPN, OPN), (PER, PER))),
                    ((1, 1, 16), ((PER, PER), (OPN, OPN), (CLO, CLO))), ((5, 7), ((CLO, CLO), (OPN, OPN)))):
        for masks, with_active in ((1, True), (1, False), (3, True)):
            pc.check_cellflags(emu_ctx, MEM, res, bc, rng, masks, with_active)


def test_obstacle_rasterisation_and_moving_obstacles(emu_ctx):
    """ overlapping box + sphere, linear and angular obstacle velocities, 2-D and 3-D, more than one launch worth """
    rng = np.random.default_rng(13)
    for dtype in (np.float32, np.float64):
        dom, grid = pc.make_case((24, 20), ((CLO, CLO), (OPN, OPN)), dtype, batch=2)
        obstacles = [pc.O.BoxObstacle((4.0, 3.0), (10.0, 9.5), velocity=(0.5, -0.25), angular_velocity=0.3,
                                       rotation=[[np.cos(0.4), -np.sin(0.4)], [np.sin(0.4), np.cos(0.4)]]),
                     pc.O.SphereObstacle((9.0, 9.0), 3.0),
                     pc.O.SphereObstacle((17.0, 12.0), 2.5, angular_velocity=-1.0)]
        pc.check_obstacle_kernels(emu_ctx, MEM, dom, grid, dtype, rng, obstacles)
        dom, grid = pc.make_case((12, 10, 16), ((CLO, CLO), (PER, PER), (CLO, OPN)), dtype, batch=1)
        obstacles = [pc.O.BoxObstacle((4.0, 3.0, 5.0), (8.0, 7.0, 11.0), angular_velocity=(0.1, -0.2, 0.3)),
                     pc.O.SphereObstacle((6.0, 6.0, 9.0), 3.0, velocity=(1.0, 0.0, -1.0))]
        pc.check_obstacle_kernels(emu_ctx, MEM, dom, grid, dtype, rng, obstacles)
    dom, grid = pc.make_case((32, 32), ((CLO, CLO), (CLO, CLO)), np.float32, batch=1)
    many = [pc.O.SphereObstacle((2.0 + 1.5 * k, 3.0 + 1.2 * k), 1.0, velocity=(0.1 * k, 0.0)) for k in range(20)]   # > 16: two launches
    pc.check_obstacle_kernels(emu_ctx, MEM, dom, grid, np.float32, rng, many)


@pytest.mark.parametrize("shape,codes", [((7, 5), ((PER, PER), (CLO, OPN))), ((1, 9), ((CLO, CLO), (OPN, OPN))), ((4, 6, 5), ((OPN, CLO), (PER, PER), (CLO, CLO))),
                                         ((3, 1, 8), ((PER, PER), (OPN, OPN), (CLO, OPN)))])
def test_grid_sample_matches_oracle(emu_ctx, shape, codes):
    """ math.grid_sample (phi/field/_resample.py:257-259) at arbitrary coordinates, far outside the array included """
    rng = np.random.default_rng(31)
    consts = [(0.3, -1.2)] * len(shape)
    for dtype in (np.float32, np.float64):
        pc.check_grid_sample(emu_ctx, MEM, shape, codes, consts, dtype, rng)
        pc.check_grid_sample(emu_ctx, MEM, shape, codes, consts, dtype, rng, batch=3, points=70, shared_values=True, spread=0.6)


def test_grid_sample_wild_coordinates(emu_ctx):
    for dtype in (np.float32, np.float64):
        pc.check_grid_sample_wild_coordinates(emu_ctx, MEM, dtype)


def test_embedded_obstacles(emu_ctx):
    """ geom.infinite_cylinder / embed (examples/grids/Wake_Flow.ipynb): the obstacle ignores the embedding axis; also as a union member """
    rng = np.random.default_rng(22)
    O = pc.O
    for dtype in (np.float32, np.float64):
        dom, grid = pc.make_case((12, 10, 8), ((CLO, OPN), (PER, PER), (PER, PER)), dtype, batch=2)
        cylinder = O.EmbeddedObstacle(O.SphereObstacle((5.0, 4.5), 2.6), (0, 1))                        # infinite along z
        slab = O.EmbeddedObstacle(O.BoxObstacle((2.0,), (4.5,)), (2,), velocity=(0.0, 0.3, 0.0))        # infinite along x and y
        pc.check_obstacle_kernels(emu_ctx, MEM, dom, grid, dtype, rng, [cylinder, slab])
        pc.check_obstacle_kernels(emu_ctx, MEM, dom, grid, dtype, rng,
                                  [O.UnionObstacle((O.EmbeddedObstacle(O.SphereObstacle((8.0, 3.0), 2.0), (0, 2)), O.BoxObstacle((1.0, 1.0, 1.0), (4.0, 3.0, 5.0))))])
        dom, grid = pc.make_case((16, 12), ((CLO, CLO), (OPN, OPN)), dtype, batch=1)
        pc.check_obstacle_kernels(emu_ctx, MEM, dom, grid, dtype, rng, [O.EmbeddedObstacle(O.BoxObstacle((5.0,), (9.0,)), (0,))])
    bad = pc.C.make_obstacles([dict(kind=pc.C.OBSTACLE_SPHERE, center=(1, 1), half_size=(1, 1), embed_mask=3)])
    with pytest.raises(pc.C.PhiHipError):
        emu_ctx.apply_obstacles(grid, bad, 1, [0, 0])


def test_union_obstacles(emu_ctx):
    """ Obstacle(union(geometries)) (examples/grids/Fluid_Logo.ipynb): inside = any member, soft mask = max over the members --
    not the product the same members give as separate obstacles; groups next to plain obstacles, across the 16-per-launch split """
    rng = np.random.default_rng(21)
    O = pc.O
    for dtype in (np.float32, np.float64):
        dom, grid = pc.make_case((24, 20), ((CLO, CLO), (OPN, OPN)), dtype, batch=2)
        logo = O.UnionObstacle((O.BoxObstacle((4.0, 3.0), (7.0, 12.0)), O.BoxObstacle((7.0, 9.0), (13.0, 12.0)), O.SphereObstacle((13.5, 10.0), 2.2)))
        drifting = O.UnionObstacle((O.BoxObstacle((15.0, 2.0), (18.0, 5.0)), O.BoxObstacle((17.5, 4.0), (21.0, 7.5))), velocity=(0.3, -0.2))
        obstacles = [O.SphereObstacle((9.0, 16.0), 2.0, angular_velocity=0.5), logo, drifting, O.BoxObstacle((1.0, 15.0), (4.0, 18.5))]
        pc.check_obstacle_kernels(emu_ctx, MEM, dom, grid, dtype, rng, obstacles)
        separate = list(logo.members)
        a = O.apply_boundary_conditions(pc.random_velocity(dom, 1, dtype, np.random.default_rng(3)), [logo], dom)
        b = O.apply_boundary_conditions(pc.random_velocity(dom, 1, dtype, np.random.default_rng(3)), separate, dom)
        assert max(np.abs(x - y).max() for x, y in zip(a, b)) > 1e-3          # the union is a different obstacle than its members
    dom, grid = pc.make_case((12, 10, 16), ((CLO, CLO), (PER, PER), (CLO, OPN)), np.float32, batch=1)
    pc.check_obstacle_kernels(emu_ctx, MEM, dom, grid, np.float32, rng,
                              [O.UnionObstacle((O.BoxObstacle((4.0, 3.0, 5.0), (8.0, 7.0, 9.0)), O.SphereObstacle((7.0, 6.0, 10.0), 2.5)))])
    dom, grid = pc.make_case((32, 32), ((CLO, CLO), (CLO, CLO)), np.float32, batch=1)
    many = [O.SphereObstacle((2.0 + 1.5 * k, 3.0 + 1.2 * k), 1.0) for k in range(12)]
    many.append(O.UnionObstacle(tuple(O.SphereObstacle((28.0 - 1.4 * k, 4.0 + 1.3 * k), 1.2) for k in range(9))))   # straddles entry 16
    pc.check_obstacle_kernels(emu_ctx, MEM, dom, grid, np.float32, rng, many)
    too_many = [O.UnionObstacle(tuple(O.SphereObstacle((2.0 + k, 3.0 + k), 1.0) for k in range(17)))]
    with pytest.raises(pc.C.PhiHipError) as e:
        pc.check_obstacle_kernels(emu_ctx, MEM, dom, grid, np.float32, rng, too_many)
    assert e.value.status == -3


@pytest.mark.parametrize("res,bc", GRIDS_2D[:4] + GRIDS_3D[:2])
def test_make_incompressible_matches_oracle_and_is_divergence_free(emu_ctx, res, bc):
    rng = np.random.default_rng(9)
    dtype = np.float32
    dom, grid = pc.make_case(res, bc, dtype, batch=2, upper=tuple(100.0 for _ in res))
    pc.check_make_incompressible(emu_ctx, MEM, dom, grid, dtype, rng)


@pytest.mark.parametrize("res,bc", [((4, 9, 264), ((PER, PER), (CLO, OPN), (CLO, OPN))), ((3, 7, 384), ((CLO, CLO), (CLO, CLO), (CLO, CLO))), ((5, 10, 512), ((OPN, OPN), (PER, PER), (PER, PER))),
                                    ((11, 300), ((CLO, OPN), (OPN, CLO)))])
def test_wide_row_tiles(emu_ctx, res, bc):
    """ r6: the WIDE row tiles (ids 11 / 12: fp64 rows of 129 ... 256 vectors -- 132, 192 (BASELINE configs[4]'s 384-cell rows), 256, 150 lanes -- ONE thread
    row, each lane fetches both halo vectors of its column): the operator, both CG forms across a refresh, cell flags; every boundary rule at the row's ends """
    dt = np.float64
    dom, grid = pc.make_case(res, bc, dt, batch=2)
    try:
        emu_ctx.set_small_grid_solver(False)
        for rows in (2, 4):
            for chunk in (2, 3):
                emu_ctx.set_tuning(rows, 256, chunk)
                for flags in (False, True):
                    plan = emu_ctx.query_plan(grid, flags, 1)
                    assert plan["tpr"] == 256 and plan["rows"] == rows, plan
                pc.check_laplace(emu_ctx, MEM, dom, grid, dt, np.random.default_rng(11))
            for mode in (0, 2):
                emu_ctx.set_single_reduction_cg(mode)
                pc.check_cg(emu_ctx, MEM, dom, grid, dt, np.random.default_rng(12), max_iter=9, refresh=4, fixed_iterations=True)
            emu_ctx.set_single_reduction_cg(0)
        if all(lo == CLO and hi == CLO for lo, hi in bc):
            emu_ctx.set_tuning(2, 256, 2)
            ob = pc.O.BoxObstacle(tuple(0.3 * x for x in res), tuple(0.55 * x for x in res))
            dom1, grid1 = pc.make_case(res, bc, dt, batch=1, upper=tuple(float(x) for x in res))
            pc.check_make_incompressible(emu_ctx, MEM, dom1, grid1, dt, np.random.default_rng(13), obstacles=[ob], max_div=1e-4)      # (a 3 x 7 x 384 sliver: the solve stops at its relative tolerance)
        # fp32 has no wide row tile: a pinned (., 256) falls back to the full-row tile silently
        dom32, grid32 = pc.make_case(res, bc, np.float32, batch=1)
        emu_ctx.set_tuning(2, 256, 2)
        assert emu_ctx.query_plan(grid32, False, 1)["tpr"] != 256
        pc.check_laplace(emu_ctx, MEM, dom32, grid32, np.float32, np.random.default_rng(11))
    finally:
        emu_ctx.set_tuning(0, 0, 0)
        emu_ctx.set_single_reduction_cg(1)
        emu_ctx.set_small_grid_solver(True)


def test_tile_configurations_agree(emu_ctx):
    """ every tile configuration of the marching kernel computes the same operator """
    rng = np.random.default_rng(10)
    dtype = np.float32
    dom, grid = pc.make_case((5, 36, 264), ((CLO, OPN), (PER, PER), (CLO, CLO)), dtype, batch=1)
    try:
        for rows, tpr in [(1, 16), (2, 16), (2, 32), (4, 32), (4, 64), (1, 64), (2, 64), (1, 32), (1, 128), (2, 128), (4, 128)]:      # (., 128): the row tiles (66 lanes per row here)
            for chunk in (2, 5):
                emu_ctx.set_tuning(rows, tpr, chunk)
                pc.check_laplace(emu_ctx, MEM, dom, grid, dtype, np.random.default_rng(11))
        emu_ctx.set_tuning(4, 32, 3)
        pc.check_cg(emu_ctx, MEM, dom, grid, dtype, rng, max_iter=8, fixed_iterations=True)
        # the row tile (r4): whole rows of 66 / 72 lanes, three thread rows per workgroup, no halo columns -- every boundary rule at the row's ends,
        # a ragged last tile (36 and 10 rows in tiles of 3), both CG forms, flags, fp64 (rows of 66 vectors of two)
        for res, bc, dt in (((5, 36, 264), ((CLO, OPN), (PER, PER), (CLO, CLO)), np.float32), ((4, 10, 288), ((PER, PER), (CLO, CLO), (PER, PER)), np.float32),
                            ((3, 7, 264), ((OPN, OPN), (OPN, CLO), (OPN, OPN)), np.float32), ((4, 9, 132), ((PER, PER), (CLO, OPN), (CLO, OPN)), np.float64)):
            dom, grid = pc.make_case(res, bc, dt, batch=2)
            emu_ctx.set_small_grid_solver(False)
            for rows in (1, 2, 4):
                emu_ctx.set_tuning(rows, 128, 2)
                plan = emu_ctx.query_plan(grid, False, 1)
                assert plan["tpr"] == 128 and plan["rows"] == rows
                for mode in ((0, 2) if rows == 1 else (0,)):
                    emu_ctx.set_single_reduction_cg(mode)
                    pc.check_laplace(emu_ctx, MEM, dom, grid, dt, np.random.default_rng(11))
                    pc.check_cg(emu_ctx, MEM, dom, grid, dt, np.random.default_rng(12), max_iter=9, refresh=4, fixed_iterations=True)
            emu_ctx.set_tuning(1, 128, 2)
        dom1, grid1 = pc.make_case((6, 7, 264), ((CLO, CLO),) * 3, np.float32, batch=1)
        pc.check_make_incompressible(emu_ctx, MEM, dom1, grid1, np.float32, np.random.default_rng(13), obstacles=[pc.O.BoxObstacle((2.0, 2.0, 80.0), (4.0, 5.0, 180.0))])
    finally:
        emu_ctx.set_tuning(0, 0, 0)
        emu_ctx.set_small_grid_solver(True)
        emu_ctx.set_single_reduction_cg(1)


def test_bad_arguments_are_reported(emu_ctx, emu_library):
    dom, grid = pc.make_case((8, 8), ((PER, PER), (PER, PER)), np.float32)
    with pytest.raises(pc.C.PhiHipError) as e:
        emu_ctx.laplace_apply(grid, 0, 1, 0, 0)
    assert e.value.status == -1 and "NULL" in str(e.value)
    bad = pc.C.make_grid(2, 0, 1, (8, 8), (0, 0), (8, 8), ((PER, CLO), (PER, PER)))
    with pytest.raises(pc.C.PhiHipError):
        emu_ctx.component_shape(bad, 0)
    for batch in (0, 65536):                                      # empty and oversized batches are rejected, not launched
        with pytest.raises(pc.C.PhiHipError) as e:
            emu_ctx.component_shape(pc.C.make_grid(2, 0, batch, (8, 8), (0, 0), (8, 8), ((PER, PER), (PER, PER))), 0)
        assert e.value.status == -1 and "batch" in str(e.value)
    with pytest.raises(pc.C.PhiHipError) as e:                    # unknown solve method
        emu_ctx.cg_solve(grid, 0, 1, 8, 16, pc.C.Solve(1e-5, 0.0, 10, 50, 10, 7))
    assert e.value.status == -1 and "method" in str(e.value)


def test_bad_arguments_of_the_widened_entry_points(emu_ctx):
    """ error behaviour of the f2-f5 entry points: negative status + message, never a crash """
    C = pc.C
    dom, grid = pc.make_case((8, 8, 8), ((PER, PER),) * 3, np.float32)
    v = [np.zeros((1, 8, 8, 8), np.float32) for _ in range(3)]
    P = lambda arrs: [a.ctypes.data for a in arrs]
    with pytest.raises(C.PhiHipError) as e:                       # output aliases the input
        emu_ctx.mac_cormack_staggered(grid, P(v), P(v), P(v), 0.1, 1.0)
    assert e.value.status == -1 and "alias" in str(e.value)
    s = np.zeros((1, 8, 8, 8), np.float32)
    with pytest.raises(C.PhiHipError) as e:                       # scalar periodicity must match the grid
        emu_ctx.mac_cormack_centered(grid, s.ctypes.data, ((CLO, CLO),) * 3, None, P(v), np.zeros_like(s).ctypes.data, 0.1, 1.0)
    assert "periodicity" in str(e.value)
    bad = C.make_obstacles([dict(kind=7, center=(1, 1, 1), half_size=(1, 1, 1))])
    with pytest.raises(C.PhiHipError) as e:
        emu_ctx.apply_obstacles(grid, bad, 1, P(v))
    assert "unknown kind" in str(e.value)
    with pytest.raises(C.PhiHipError) as e:                       # halo announced but no plane given
        emu_ctx.slab_residual(grid, (True, False), 0, s.ctypes.data, (0, 0), s.ctypes.data, np.zeros_like(s).ctypes.data,
                              np.zeros(2).ctypes.data)
    assert "halo plane missing" in str(e.value)
    g2, grid2 = pc.make_case((8, 8), ((PER, PER),) * 2, np.float32)
    with pytest.raises(C.PhiHipError) as e:                       # slabs are 3-D only
        emu_ctx.slab_residual(grid2, (False, False), 0, s.ctypes.data, (0, 0), s.ctypes.data, s.ctypes.data, np.zeros(2).ctypes.data)
    assert "3-D" in str(e.value)
    with pytest.raises(C.PhiHipError):                            # gradient buffers are mandatory for the MacCormack adjoint
        emu_ctx.mac_cormack_centered_backward(grid, s.ctypes.data, ((PER, PER),) * 3, None, P(v), s.ctypes.data, 0.1, 1.0, 0, None)
    plan = emu_ctx.query_plan(grid, False, 1)
    assert plan["rows"] in (1, 2, 4) and plan["nblk"] >= 1 and plan["occupancy"] >= 1
    with pytest.raises(C.PhiHipError):
        emu_ctx.set_tuning_kernel(5, 1, 16, 8)


def test_baseline_config_cases_at_toy_sizes(emu_ctx):
    """ tests/baseline_cases.py (the oracle comparisons the GPU suite runs at the BASELINE sizes) at sizes the emulation finishes """
    import baseline_cases as bc
    try:
        emu_ctx.set_small_grid_solver(False)
        bc.config2_step(emu_ctx, MEM, 16, 12)
        bc.config3_solve(emu_ctx, MEM, 16, 8)
        bc.config5_cavity(emu_ctx, MEM, 16, 8)
        bc.config4_batched_smoke(emu_ctx, MEM, 32, 3, 2, 10)
        bc.max_size_step(emu_ctx, MEM, 24, 8)            # (the 1024^3 case's own logic: x-planes bit-identical, plane 0 = the 2-D oracle)
    finally:
        emu_ctx.set_small_grid_solver(True)


def test_unaligned_buffers_take_the_scalar_path(emu_ctx):
    """ field pointers that are not 16-byte aligned (offset views) must not reach the ALIGNED 16-byte vector loads: the plan falls
    back to the element-aligned instantiation (r4: UNAL vectors; rows shorter than two vectors: scalar) and the results stay those of the oracle """
    dtype = np.float32
    dom, grid = pc.make_case((6, 20, 72), ((CLO, OPN), (PER, PER), (CLO, CLO)), dtype, batch=1)
    rng = np.random.default_rng(3)
    p = rng.standard_normal((1,) + dom.res).astype(dtype)
    ref = pc.O.masked_laplace(p, dom, None, None)
    cells = p.size
    raw_in, raw_out = np.zeros(cells + 8, dtype), np.zeros(cells + 8, dtype)
    for shift in (0, 1, 3):                                       # 0: aligned (numpy allocations are 16-byte aligned here), 1 / 3: +4 / +12 bytes
        a, o = raw_in[shift:shift + cells], raw_out[shift:shift + cells]
        a[:] = p.ravel()
        o[:] = np.nan
        assert (a.ctypes.data % 16 != 0) == (shift != 0) or raw_in.ctypes.data % 16 != 0
        emu_ctx.laplace_apply(grid, 0, 1, a.ctypes.data, o.ctypes.data)
        assert pc.rel_err(o.reshape(p.shape), ref) <= pc.tol(dtype)['stencil']
    rhs = pc.O.balance_divergence(rng.standard_normal((1,) + dom.res).astype(dtype), None) if not dom.flexible() else rng.standard_normal((1,) + dom.res).astype(dtype)
    xs = []
    try:
        emu_ctx.set_small_grid_solver(False)
        for shift in (0, 1):
            a, o = raw_in[shift:shift + cells], raw_out[shift:shift + cells]
            a[:] = rhs.ravel()
            o[:] = 0
            emu_ctx.cg_solve(grid, 0, 1, a.ctypes.data, o.ctypes.data, pc.solve_params(dtype, max_iter=6, rtol=0.0, check=0))
            xs.append(o.copy())
    finally:
        emu_ctx.set_small_grid_solver(True)
    assert pc.rel_l2(xs[1], xs[0]) <= 1e-5


@pytest.mark.parametrize("res,bc", [((20, 24, 72), ((OPN, CLO), (CLO, CLO), (PER, PER))), ((19, 35, 70), ((PER, PER), (OPN, OPN), (CLO, OPN))),
                                    ((40, 136), ((CLO, OPN), (PER, PER)))])
def test_tiled_advection_across_tiles_and_chunks(emu_ctx, res, bc):
    """ advect_tile.hip: several tiles along both fast axes and several chunks of planes (ring refill, chunk prologue), every boundary
    kind on the tile edges, displacements below and beyond the halo (LDS taps / global fallback) """
    rng = np.random.default_rng(31)
    for dtype in (np.float32, np.float64):
        dom, grid = pc.make_case(res, bc, dtype, batch=2)
        for dt in (0.45, 1.4, 3.3):
            pc.check_advect_staggered(emu_ctx, MEM, dom, grid, dtype, rng, dt=dt)


@pytest.mark.parametrize("res,bc,batch", [((33, 264), ((OPN, OPN), (CLO, OPN)), 2), ((72, 128), ((PER, PER), (PER, PER)), 1), ((40, 216), ((CLO, CLO), (CLO, CLO)), 1),
                                          ((32, 264), ((CLO, OPN), (PER, PER)), 9)])      # (r6: 9 entries x 2 workgroups > the 16 a launch holds here: sub-batches of 8 + 1)
def test_resident_cg_matches_oracle(emu_ctx, res, bc, batch):
    """ cg_resident.hip under the emulation's RESIDENT launch (tests/hipemu: every workgroup of the grid alive at once, a polling fiber
    yields): rows split over 3-5 workgroups per entry incl. a ragged last one, one and two vectors per thread, wrap / clamp / zero rows and
    columns, true-residual refreshes inside the launch; the closed box also through make_incompressible (balance shift in the first pass) """
    try:
        emu_ctx.set_resident_cg(2)
        dom, grid = pc.make_case(res, bc, np.float32, batch=batch)
        pc.check_cg(emu_ctx, MEM, dom, grid, np.float32, np.random.default_rng(3), max_iter=12, refresh=5, fixed_iterations=True)
        if bc[0][0] == CLO and bc[1][0] == CLO:
            # the projection of a closed box hands the solver the UNBALANCED divergence + its mean (`shift`): the resident kernel subtracts it
            # in its first pass and writes the balanced right-hand side back, like MODE_RESID_BAL -- ten iterations against the launch forms
            v = pc.random_velocity(dom, batch, np.float32, np.random.default_rng(6), 0.1)
            out = {}
            for mode in (0, 2):
                emu_ctx.set_resident_cg(mode)
                dv = [MEM.to_dev(a) for a in v]
                dp, ddiv = MEM.to_dev(np.zeros((batch,) + dom.res, np.float32)), MEM.empty((batch,) + dom.res, np.float32)
                info = emu_ctx.make_incompressible(grid, [MEM.ptr(a) for a in dv], None, 0, 1, True, MEM.ptr(dp), MEM.ptr(ddiv),
                                                   pc.solve_params(np.float32, 10, 0.0, 0.0, 50, 0, 0))
                out[mode] = (MEM.to_host(dp), MEM.to_host(ddiv), [MEM.to_host(a) for a in dv], [i.iterations for i in info])
            assert out[0][3] == out[2][3] == [10] * batch
            assert pc.rel_l2(out[2][1], out[0][1]) <= 1e-6 and abs(float(out[2][1].mean())) <= 1e-6 * float(np.abs(out[2][1]).max())      # the balanced rhs
            assert pc.rel_l2(pc.demean(out[2][0]), pc.demean(out[0][0])) <= 2e-5
            assert all(pc.rel_l2(a, b_) <= 2e-5 for a, b_ in zip(out[2][2], out[0][2]))
    finally:
        emu_ctx.set_resident_cg(1)          # the library's default since r6


@pytest.mark.parametrize("res,bc,batch,obstacles", [
    ((40, 216), ((CLO, CLO), (CLO, CLO)), 2, [pc.O.BoxObstacle((10.0, 60.0), (22.0, 130.0))]),                                 # three workgroups per entry, the box spans a cut
    ((33, 264), ((OPN, OPN), (CLO, OPN)), 1, [pc.O.SphereObstacle((16.0, 100.0), 9.5), pc.O.BoxObstacle((0.0, 200.0), (12.0, 230.0))])])   # two vectors per thread, ragged last workgroup
# (both > 8192 cells: below that the one-workgroup solver takes a solve; three more grids, up to 104 workgroups, run on the GPU: tests/test_gpu_parity.py)
def test_resident_cg_with_cell_flags(emu_ctx, res, bc, batch, obstacles):
    """ r6: cg_resident.hip with obstacles (cell flags held in registers) against the oracle: fixed iterations across a refresh, tolerance mode, the projection """
    pc.check_resident_with_flags(emu_ctx, MEM, res, bc, batch, obstacles)


@pytest.mark.parametrize("res,bc", [((9, 13), ((CLO, OPN), (PER, PER))), ((5, 6, 11), ((PER, PER), (CLO, OPN), (OPN, CLO))), ((3, 5, 261), ((CLO, CLO), (PER, PER), (PER, PER))), ((2, 4, 257), ((PER, PER), (CLO, OPN), (PER, PER))),
                                    ((4, 7, 15), ((OPN, OPN), (CLO, CLO), (CLO, CLO)))])
def test_ragged_rows_on_the_vector_kernels(emu_ctx, res, bc):
    """ r4: rows that are not whole 16-byte vectors take the UNAL instantiation of the marching kernels (16-byte vectors at element alignment,
    the last vector of a row OVERLAPS its neighbour instead of being partial; the plan reports a negative vector width): every CG form on
    the marching kernels -- two launches, single reduction, 'CG-adaptive', with obstacle flags, the balanced projection -- both dtypes;
    261 cells = one full 256-cell tile + a second one that holds 5 cells; 257 = a second tile of ONE cell: the overlapping vector would reach
    into the first tile, whose cells this workgroup does not stage -- such row lengths keep the scalar kernels in fp32 (fp64: 129 of its
    128-cell tiles, so 257 = one cell after two tiles likewise); 11 / 13 / 15 = 1 / 3 / 1 cells of overlap in fp32 """
    try:
        emu_ctx.set_small_grid_solver(False)
        for dtype in (np.float32, np.float64):
            dom, grid = pc.make_case(res, bc, dtype, batch=2)
            vmax = 4 if dtype == np.float32 else 2
            assert emu_ctx.query_plan(grid, False, 1)["vec"] == (1 if 0 < res[-1] % (64 * vmax) < vmax else -vmax)
            for mode in (0, 2):
                emu_ctx.set_single_reduction_cg(mode)
                pc.check_laplace(emu_ctx, MEM, dom, grid, dtype, np.random.default_rng(1))
                pc.check_cg(emu_ctx, MEM, dom, grid, dtype, np.random.default_rng(7), max_iter=9, refresh=4, fixed_iterations=True)
                if dtype == np.float32 and len(res) == 2:       # (tolerance mode: the emulation's minutes go here)
                    pc.check_cg(emu_ctx, MEM, dom, grid, dtype, np.random.default_rng(9))
                    pc.check_make_incompressible(emu_ctx, MEM, dom, grid, dtype, np.random.default_rng(10))
            emu_ctx.set_single_reduction_cg(0)
            if len(res) == 2 or dtype == np.float32:
                pc.check_cg(emu_ctx, MEM, dom, grid, dtype, np.random.default_rng(12), max_iter=9, refresh=4, fixed_iterations=True, adaptive=True)
            if res == (4, 7, 15) and dtype == np.float32:
                obstacles = [pc.O.BoxObstacle(tuple(0.3 * n for n in res), tuple(0.7 * n for n in res))]
                dom1, grid1 = pc.make_case(res, bc, dtype, batch=1)
                pc.check_make_incompressible(emu_ctx, MEM, dom1, grid1, dtype, np.random.default_rng(11), obstacles=obstacles)
    finally:
        emu_ctx.set_small_grid_solver(True)
        emu_ctx.set_single_reduction_cg(1)


def test_autotuned_launch_plans_stay_correct(emu_library):
    """ first-call autotune of the CG marching kernels (cg.hip autotune_cg): whatever (tile, chunk) the timings pick -- noise under the
    emulation -- the solve must equal the oracle's, explicit tunings still win, and disabling it restores the analytic plan """
    ctx = pc.C.Context(emu_library, 0)
    try:
        ctx.set_small_grid_solver(False)
        ctx.set_autotune(True)
        for res, bc in (((8, 12, 16), ((PER, PER), (CLO, OPN), (PER, PER))), ((20, 24), ((CLO, CLO), (PER, PER)))):
            dom, grid = pc.make_case(res, bc, np.float32, batch=2)
            pc.check_cg(ctx, MEM, dom, grid, np.float32, np.random.default_rng(7), max_iter=9, refresh=4, fixed_iterations=True)
            tuned = [ctx.query_plan(grid, False, f) for f in (1, 2, 3)]
            assert all(p["nblk"] >= 1 and p["chunk"] >= 1 for p in tuned)
            pc.check_cg(ctx, MEM, dom, grid, np.float32, np.random.default_rng(8))          # cached plans, tolerance mode
        dom, grid = pc.make_case((8, 12, 16), ((PER, PER),) * 3, np.float32, batch=1)
        ctx.set_tuning(2, 16, 3)
        assert ctx.query_plan(grid, False, 1)["chunk"] == 3 and ctx.query_plan(grid, False, 1)["rows"] == 2
        ctx.set_tuning(0, 0, 0)
        ctx.set_autotune(False)
        model = [ctx.query_plan(grid, False, f) for f in (1, 2, 3)]
        ctx2 = pc.C.Context(emu_library, 0)                                                  # PHIHIP_AUTOTUNE=0 (conftest): analytic plan
        assert model == [ctx2.query_plan(grid, False, f) for f in (1, 2, 3)]
    finally:
        ctx.close()


def test_workspace_placement_keeps_the_solve(emu_library, monkeypatch):
    """ r6 placement of the CG workspace (cg.hip place_workspace): the first solve on a freshly grown workspace allocates candidate (r, d0, d1) triples, times
    the iteration loop on each and keeps one -- whichever it keeps (timing noise under the emulation), the solve equals the oracle's, the record says how many
    candidates there were, a second solve on the same workspace does not choose again, and without free "device" memory the first allocation simply stays """
    monkeypatch.setenv("HIPEMU_FREE_BYTES", str(1 << 28))
    monkeypatch.setenv("PHIHIP_WS_PLACE_MIN_BYTES", "0")          # (the library places vectors beyond the Infinity Cache regime only: > 72 MB)
    ctx = pc.C.Context(emu_library, 0)
    try:
        ctx.set_small_grid_solver(False)
        ctx.set_single_reduction_cg(0)
        ctx.set_autotune(True)
        assert ctx.workspace_placement(3)["candidates"] == 0
        dom, grid = pc.make_case((8, 12, 16), ((PER, PER), (CLO, OPN), (PER, PER)), np.float32, batch=2)
        pc.check_cg(ctx, MEM, dom, grid, np.float32, np.random.default_rng(7), max_iter=9, refresh=4, fixed_iterations=True)
        rec = ctx.workspace_placement()
        assert rec["candidates"] == 3 and rec["us_first"] > 0 and 0 < rec["us_kept"] <= rec["us_first"]
        pc.check_cg(ctx, MEM, dom, grid, np.float32, np.random.default_rng(8))                  # the kept workspace, tolerance mode
        assert ctx.workspace_placement() == rec
        dom, grid = pc.make_case((20, 24), ((CLO, CLO), (PER, PER)), np.float32, batch=2)       # a smaller grid: nothing grows, nothing is chosen
        pc.check_cg(ctx, MEM, dom, grid, np.float32, np.random.default_rng(9), max_iter=9, refresh=4, fixed_iterations=True)
        assert ctx.workspace_placement() == rec
        monkeypatch.setenv("HIPEMU_FREE_BYTES", "0")
        dom, grid = pc.make_case((8, 12, 16), ((CLO, CLO),) * 3, np.float64, batch=3)           # grows (fp64, batch 3) with no memory for candidates
        pc.check_cg(ctx, MEM, dom, grid, np.float64, np.random.default_rng(10), max_iter=9, refresh=4, fixed_iterations=True)
        assert ctx.workspace_placement()["candidates"] == 1
    finally:
        ctx.close()


@pytest.mark.parametrize("res,bc", [((20, 24), ((CLO, CLO), (PER, PER))), ((16, 20), ((OPN, OPN), (CLO, OPN))), ((8, 12, 16), ((PER, PER), (CLO, OPN), (PER, PER))),
                                    ((4, 12, 72), ((CLO, OPN), (PER, PER), (CLO, CLO)))])
def test_single_reduction_cg_matches_oracle(emu_ctx, res, bc):
    """ the one-launch-per-iteration (Chronopoulos-Gear) form of 'CG' (stencil_march.hpp MODE_CG1): same iterates as PhiML's cg in exact
    arithmetic -- fixed iteration counts incl. refreshes, tolerance mode with per-entry freezing, obstacles, both dtypes """
    try:
        emu_ctx.set_small_grid_solver(False)
        emu_ctx.set_single_reduction_cg(2)
        for dtype in (np.float32, np.float64):
            dom, grid = pc.make_case(res, bc, dtype, batch=2)
            pc.check_cg(emu_ctx, MEM, dom, grid, dtype, np.random.default_rng(7), max_iter=7, fixed_iterations=True)
            pc.check_cg(emu_ctx, MEM, dom, grid, dtype, np.random.default_rng(8), max_iter=11, refresh=4, fixed_iterations=True)
            if dtype == np.float32 or len(res) == 2:          # (tolerance mode to 1e-10 in fp64 costs the emulation a minute on the 3-D grids)
                pc.check_cg(emu_ctx, MEM, dom, grid, dtype, np.random.default_rng(9))
                pc.check_make_incompressible(emu_ctx, MEM, dom, grid, dtype, np.random.default_rng(10))
        if res == (8, 12, 16):
            dom, grid = pc.make_case((12, 10, 16), ((CLO, CLO),) * 3, np.float32, batch=1)
            pc.check_make_incompressible(emu_ctx, MEM, dom, grid, np.float32, np.random.default_rng(11), obstacles=[pc.O.BoxObstacle((4.0, 3.0, 5.0), (8.0, 7.0, 11.0))])
    finally:
        emu_ctx.set_small_grid_solver(True)
        emu_ctx.set_single_reduction_cg(1)
