"""
`-m gpu`: ORACLE parity at the sizes BASELINE.json quotes numbers for (VERDICT r1 "What's weak" 1) -- not properties: the HIP path
through the C ABI against the NumPy oracle on the same inputs, with the launch plans that only large grids select.
Each case prints its measured errors (pytest -s / captured in the log) so the margins are visible.
"""
import numpy as np
import pytest

import baseline_cases as bc
import parity_cases as pc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(gpu_backend):
    return gpu_backend.ctx


@pytest.fixture(scope="module")
def mem(gpu_backend):
    return pc.TorchMem(str(gpu_backend.device))


def test_config2_taylor_green_256_100_iterations_vs_oracle(ctx, mem):
    """ BASELINE configs[1] = the benchmark workload itself: 256^3 fp32, advect + 100 fixed CG iterations, one step """
    rep = {}
    bc.config2_step(ctx, mem, 256, 100, rep)
    print("config2 parity:", rep)


def test_config3_pressure_solve_512_vs_oracle(ctx, mem):
    """ BASELINE configs[2]: 512^3 fp32 pressure solve, 20 fixed iterations on the seeded random rhs (the (4,64) / chunk-64 plans) """
    rep = {}
    bc.config3_solve(ctx, mem, 512, 20, rep)
    print("config3 parity:", rep)
    for fam in (1, 2, 3):
        dom, grid = pc.make_case((512,) * 3, ((pc.PER, pc.PER),) * 3, np.float32)
        print("  plan family", fam, ctx.query_plan(grid, False, fam))


def test_config5_cavity_fp64_obstacle_256_vs_oracle(ctx, mem):
    """ BASELINE configs[4] size class: 256^3 fp64 closed cavity + lid + solid box (flags path), 20 fixed iterations """
    rep = {}
    bc.config5_cavity(ctx, mem, 256, 20, rep)
    print("config5 parity:", rep)


def test_config5_cavity_fp64_obstacle_384_vs_oracle(ctx, mem):
    """ BASELINE configs[4] AT ITS OWN SIZE: 384^3 fp64 closed cavity + lid + solid box, 20 fixed iterations. Only this size selects its
    launch plans -- (4,64) fp64 tiles with the 1.5-round rule, three 128-cell tiles per row, the 16-plane advection chunk, the vector
    gradient kernel on rows of n - 1 faces -- so only this size pins them. The oracle needs ~3.5 min of NumPy and
    ~18 GB of host memory at this size. """
    rep = {}
    bc.config5_cavity(ctx, mem, 384, 20, rep)
    print("config5 parity at 384^3:", rep)
    dom, grid = pc.make_case((384,) * 3, ((pc.CLO, pc.CLO),) * 3, np.float64)
    for fam in (1, 2, 3):
        print("  plan family", fam, ctx.query_plan(grid, True, fam))


def test_config4_batched_smoke_8x512_vs_oracle(ctx, mem):
    """ BASELINE configs[3]: 8 x 512^2 batched smoke plumes, 3 steps, 60 fixed CG iterations per projection """
    rep = {}
    bc.config4_batched_smoke(ctx, mem, 512, 8, 3, 60, rep)
    print("config4 parity:", rep)


def test_config4_batched_smoke_8x512_resident_solver_vs_oracle(ctx, mem):
    """ BASELINE configs[3] with the opt-in resident solver INSIDE the smoke step (until r5 only a bare cg_solve and one
    projection ran resident on the GPU): the same 3 steps / 60 iterations vs the oracle, projection from the previous pressure (x0 != 0), the
    balance shift folded into the resident kernel's first pass; the launch counters assert that the resident kernel is what solved """
    rep = {}
    try:
        ctx.set_resident_cg(2)
        ctx.profile_enable(True)
        ctx.profile_read(True)
        bc.config4_batched_smoke(ctx, mem, 512, 8, 3, 60, rep)
        prof = ctx.profile_read(True)
        assert prof["cg_matvec_dot"][0] == 0 and prof["cg_update"][0] == 3, prof       # one resident launch per projection, no launch-per-iteration kernels
    finally:
        ctx.profile_enable(False)
        ctx.set_resident_cg(1)          # the library's default since r6
    print("config4 parity (resident solver):", rep)



def test_max_size_1024_cubed_plane_invariance_and_2d_oracle(ctx, mem):
    """ "maximum sizes": 1024^3 fp32 (2^30 cells, 4.3 GB per array -- byte offsets beyond 2^32, element offsets up to 2^30) through properties that do not
    need a 3-D oracle at that size: bit-identical x-planes of an x-invariant flow and plane 0 against the 2-D oracle (tests/baseline_cases.py max_size_step) """
    import torch
    free, total = torch.cuda.mem_get_info()
    if free < 80 * 2 ** 30:
        pytest.skip(f"needs ~60 GB of device memory, {free / 2 ** 30:.0f} GB free")
    rep = {}
    bc.max_size_step(ctx, mem, 1024, 20, rep)
    print("max size:", rep)
    torch.cuda.empty_cache()
