"""
`-m gpu`: the phi-level API (`phiflow_amd.flow`) on the MI355X -- golden fixtures and the reference-style scenarios of
tests/test_host_api.py with the real library.
"""
import numpy as np
import pytest

import golden_cases
import test_host_api as host

pytestmark = pytest.mark.gpu


def test_golden_smoke_plume(gpu_backend):
    golden_cases.run_smoke_plume(gpu_backend)


def test_golden_smoke_plume_mac_cormack_50_steps(gpu_backend):
    """ BASELINE configs[0]: the reference's CPU-runnable case, all 50 steps against the oracle's trajectory """
    report = {}
    golden_cases.run_smoke_plume_mac_cormack(gpu_backend, report=report)
    print("config-1 rel-L2 errors vs oracle:", report)


def test_golden_taylor_green(gpu_backend):
    golden_cases.run_taylor_green(gpu_backend)


def test_golden_cavity_obstacle(gpu_backend):
    golden_cases.run_cavity_obstacle(gpu_backend)


def test_reference_style_scenarios(gpu_backend):
    host.test_staggered_storage_sizes(gpu_backend)
    host.test_with_extrapolation_restores_wall_faces(gpu_backend)
    host.test_self_advect_staggered_known_answer(gpu_backend)
    from phiflow_amd.flow import BOUNDARY, PERIODIC, ZERO, combine_sides
    for ext in (ZERO, BOUNDARY, PERIODIC):
        host.test_identity_advection(gpu_backend, ext)
    for name, ext in (("closed", ZERO), ("open", BOUNDARY), ("periodic", PERIODIC), ("mixed", combine_sides(x=BOUNDARY, y=(ZERO, BOUNDARY)))):
        host.test_make_incompressible_staggered(gpu_backend, name, ext)
    for method in ('CG', 'CG-adaptive', 'auto'):
        host.test_make_incompressible_matches_oracle(gpu_backend, method)
    host.test_obstacles_and_x0(gpu_backend)
    host.test_moving_and_rotating_obstacles(gpu_backend)
    host.test_fluid_logo_union_obstacle_and_cg_adaptive(gpu_backend)
    host.test_wake_flow_inflow_boundary_and_infinite_cylinder(gpu_backend)
    host.test_batched_smoke_with_batched_obstacle_and_inflow(gpu_backend)
    for dtype_name in ('float32', 'float64'):
        host.test_fields_on_different_grids(gpu_backend, dtype_name)
        host.test_rk4_integrator(gpu_backend, dtype_name)
    host.test_user_active_mask_plain_and_batched(gpu_backend)
    host.test_user_active_mask_with_nan_velocity(gpu_backend)
    host.test_convergence_exceptions(gpu_backend)
    host.test_lid_driven_cavity_boundaries_and_diffusion(gpu_backend)
    host.test_implicit_diffusion(gpu_backend)
    host.test_spatial_gradient_at_faces(gpu_backend)
    host.test_fp64_precision_context(gpu_backend)


def test_gradients(gpu_backend):
    """ adjoint kernels behind torch.autograd on the GPU vs finite differences of the forward path """
    host.test_make_incompressible_gradient(gpu_backend)
    host.test_implicit_diffusion_gradient(gpu_backend)
    host.test_functional_gradient_through_a_fluid_step(gpu_backend)
    host.test_gradient_through_sampling_between_grids(gpu_backend)
    host.test_colab_tutorial_functional_gradient(gpu_backend, full=True)


def test_default_backend_is_the_gpu(gpu_backend):
    """ the product path: no explicit backend -> libphihip.so + cuda device; tensors live on the GPU """
    from phiflow_amd.flow import PERIODIC, Solve, StaggeredGrid, advect, default_backend, fluid
    be = default_backend()
    assert be.device.type == "cuda"
    v = StaggeredGrid(lambda x, y: (np.sin(x), np.cos(y)), PERIODIC, x=32, y=32)
    assert v.values[0].is_cuda
    v = advect.semi_lagrangian(v, v, 0.1)
    v, p = fluid.make_incompressible(v, (), Solve('CG', 1e-5, 0))
    assert p.values.is_cuda and p.solve_info.converged == [True]


def test_full_size_batched_smoke_8x512(gpu_backend):
    """ BASELINE configs[3] at full size on one GPU: 8 x 512^2 smoke plumes (closed box, per-entry inflow position), 3 steps of
    Smoke_Plume.ipynb cell 5. Properties: the remaining divergence equals each entry's own CG residual (<= 1e-3 |rhs|), and batch
    entries are independent simulations -- entry 5 run alone gives the same fields (to solver tolerance). """
    from phiflow_amd.flow import Box, CenteredGrid, Solve, Sphere, StaggeredGrid, ZERO_GRADIENT, advect, divergence, fluid, resample
    n, B = 512, 8
    bounds = Box(x=100, y=100)
    xs = np.linspace(30, 70, B)

    def run(positions):
        b = len(positions)
        inflow_np = np.stack([0.2 * CenteredGrid(Sphere(x=float(x0), y=9.5, radius=5), ZERO_GRADIENT, bounds, x=n, y=n, backend=gpu_backend).numpy()
                              for x0 in positions])
        inflow = CenteredGrid(inflow_np, ZERO_GRADIENT, bounds, x=n, y=n, backend=gpu_backend)
        v = StaggeredGrid(0, 0, bounds, x=n, y=n, batch=b, backend=gpu_backend)
        smoke = CenteredGrid(np.zeros((b, n, n), np.float32), ZERO_GRADIENT, bounds, x=n, y=n, backend=gpu_backend)
        p = None
        for _ in range(3):
            smoke = advect.mac_cormack(smoke, v, 1.0) + inflow
            v = advect.semi_lagrangian(v, v, 1.0) + resample(smoke * (0, 0.1), to=v)
            v, p = fluid.make_incompressible(v, (), Solve('CG', 1e-3, 0, x0=p, max_iterations=6000))   # the notebook's tolerance
        return v, smoke, p

    v, smoke, p = run(xs)
    assert p.solve_info.converged == [True] * B
    # the divergence left after the projection IS the solver's residual rhs - A p (per entry, CG stops at 1e-3 |rhs|)
    div_sq = (divergence(v).values.double() ** 2).sum(dim=(1, 2)).cpu().numpy()
    assert np.all(div_sq <= 1.0e-6 * np.asarray(p.solve_info.rhs_sq) * 1.5)
    np.testing.assert_allclose(div_sq, p.solve_info.residual_sq, rtol=0.3)
    v1, smoke1, p1 = run(xs[5:6])
    # (the launch plan, hence the summation order of the dot products, depends on the batch size: equal to solver tolerance,
    # not bit for bit -- the bit-exact variant of this check runs at a small size in tests/test_parallel_gloo.py)
    rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
    assert rel(smoke1.numpy()[0], smoke.numpy()[5]) <= 1e-3
    for a, b in zip(v1.numpy(), v.numpy()):
        assert rel(a[0], b[5]) <= 2e-2
    other = rel(smoke.numpy()[2], smoke.numpy()[5])
    assert other > 0.5                                        # ... while different entries are different simulations


def test_solve_linear_and_the_backend_level_boundary(gpu_backend):
    """ SURVEY §8b on the real library: flow.solve_linear(masked_laplace, ...), the matrix-level linear_solve and grid_sample """
    import test_linear_boundary as lb
    lb.test_solve_linear_matches_make_incompressible_and_the_oracle(gpu_backend)
    lb.test_solve_linear_with_an_active_mask(gpu_backend)
    lb.test_backend_linear_solve_from_an_assembled_matrix(gpu_backend)
    lb.test_backend_grid_sample_matches_the_oracle(gpu_backend)


def test_level_b_rank_deficient_matrix_solve_on_the_device(gpu_backend):
    """ SURVEY §8b Level B on the GPU: what `with HIP: fluid.make_incompressible(v)` hands a PhiML backend for a periodic / closed box
    (phi/physics/fluid.py:145-156) -- a device-resident torch sparse matrix of `masked_laplace` + `matrix_offset` (rank_deficiency=1) + a
    balanced right-hand side -- has to run the marching CG kernels (launch counters), be recognised ONCE although every step brings a new
    matrix object, and give the same pressure as the phi-level solve. 96^3 = 885 k rows: the marching path, not the single-workgroup solver. """
    import torch
    from oracle import phi_oracle as O
    from phiflow_amd import _capi as C, linear

    class Probe(linear.HipLinearSolveMixin):
        def _hip_backend(self):
            return gpu_backend
    be, dev, ctx = Probe(), gpu_backend.device, gpu_backend.ctx
    rng = np.random.default_rng(21)
    for res, bc in (((96, 96, 96), ((O.PERIODIC, O.PERIODIC),) * 3), ((640, 512), ((O.CLOSED, O.CLOSED),) * 2)):
        dom = O.Domain(res, (0.0,) * len(res), tuple(float(n) for n in res), bc)
        A = O.laplace_csr(dom, np.float32)
        N = A.shape[0]
        y = rng.standard_normal((1, N)).astype(np.float32)
        y -= y.mean()
        yt = torch.as_tensor(y).to(dev)
        stats0 = dict(be.hip_stats)
        ctx.profile_enable(True); ctx.profile_read(reset=True)
        for step in range(3):        # a NEW device matrix per "time step" (forget_traces=True, fluid.py:165)
            lin = torch.sparse_csr_tensor(torch.as_tensor(A.indptr).to(dev), torch.as_tensor(A.indices).to(dev), torch.as_tensor(A.data).to(dev), size=A.shape)
            x, its, rsq, conv, div = be.hip_linear_solve('CG', lin, yt, torch.zeros_like(yt), 1e-4, 0.0, 2000, matrix_offset=-1.0 / N)
        prof = ctx.profile_read(reset=True); ctx.profile_enable(False)
        assert all(conv) and not any(div) and x.is_cuda
        # (these sizes take the single-reduction form: ONE fused launch per iteration, counted as cg_update; the refresh iterations add passes)
        launches = prof['cg_matvec_dot'][0] + prof['cg_update'][0] + prof['cg_update_r'][0]
        assert launches >= 3 * (its[0] - its[0] // 50 - 2) and prof['cg_residual'][0] >= 3, (prof, its)
        assert be.hip_stats['cache_misses'] == stats0['cache_misses'] + 1 and be.hip_stats['cache_hits'] == stats0['cache_hits'] + 2
        assert be.hip_stats['offsets_dropped'] == stats0['offsets_dropped'] + 3
        # same system through the phi-level C ABI call
        grid = C.make_grid(len(res), C.PHIHIP_F32, 1, res, (0.0,) * len(res), tuple(float(n) for n in res),
                           [tuple({O.PERIODIC: C.BC_PERIODIC, O.CLOSED: C.BC_CLOSED}[c] for c in pair) for pair in bc])
        rhs = yt.reshape(1, *res).contiguous()
        p = torch.zeros_like(rhs)
        info = ctx.cg_solve(grid, 0, 1, rhs.data_ptr(), p.data_ptr(), C.Solve(1e-4, 0.0, 2000, 50, 10, 0))
        assert abs(info[0].iterations - its[0]) <= 1
        a, b = x.reshape(-1).double(), p.reshape(-1).double()
        assert float((a - b).norm() / b.norm()) <= 1e-3
        assert abs(float(a.mean())) <= 1e-4 * float(a.abs().mean())


def test_scene_files_on_the_device(gpu_backend, tmp_path):
    """ SURVEY §8 f6 with the real library: the reference's scene-file window -> GPU fields -> one step vs the oracle -> round trip """
    golden_cases.run_scene_files(gpu_backend, tmp_path)


def test_slab_fluid_single_rank_on_the_device(gpu_backend):
    """ SURVEY §8 f4 on the GPU with ONE rank (no process group): the slab step is the ordinary step -- exercises SlabFluid / SlabSolver
    with device tensors; the two-rank form runs under gloo in tests/test_parallel_gloo.py (the boxes have one GPU) """
    import torch
    from phiflow_amd import _capi as C
    from phiflow_amd.slab import SlabFluid
    res, bc, batch = (48, 32, 64), ((1, 2), (1, 1), (0, 0)), 2
    ctx = gpu_backend.ctx
    grid = C.make_grid(3, C.PHIHIP_F32, batch, res, (0, 0, 0), tuple(float(r) for r in res), bc)
    gen = torch.Generator().manual_seed(5)
    v = [(0.4 * torch.randn((batch,) + tuple(ctx.component_shape(grid, c)), generator=gen)).to(gpu_backend.device) for c in range(3)]
    fluid = SlabFluid(gpu_backend, res, (0.0, 0.0, 0.0), tuple(float(r) for r in res), bc, torch.float32, batch=batch)
    assert fluid.world == 1 and [tuple(s) for s in fluid.own_shape] == [tuple(t.shape) for t in v]
    p = torch.zeros(fluid.cell_shape, device=gpu_backend.device)
    out, infos = fluid.step([t.clone() for t in v], p, 0.5, rel_tol=1e-5, max_iterations=500)
    P = lambda ts: [t.data_ptr() for t in ts]
    adv = [torch.empty_like(t) for t in v]
    ctx.advect_staggered(grid, P(v), P(v), P(adv), 0.5)
    div = torch.empty((batch,) + res, device=gpu_backend.device)
    ctx.divergence(grid, P(adv), 0, 1, False, div.data_ptr())
    p_ref = torch.zeros_like(div)
    info = ctx.cg_solve(grid, 0, 1, div.data_ptr(), p_ref.data_ptr(), C.Solve(1e-5, 0.0, 500, 50, 10, 0))
    ctx.grad_subtract(grid, 0, 1, p_ref.data_ptr(), P(adv))
    torch.cuda.synchronize()
    assert all(i.converged for i in infos) and all(abs(a.iterations - b.iterations) <= 2 for a, b in zip(infos, info))
    assert float((p - p_ref).abs().max()) <= 2e-4 * float(p_ref.abs().max())
    for c in range(3):
        assert float((out[c] - adv[c]).abs().max()) <= 1e-4


def test_example_scripts_run(gpu_backend):
    """ examples/*.py are the reference notebooks with the import line changed: they have to keep running """
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for script, args in (("smoke_plume.py", ["--size", "64", "--steps", "3"]), ("smoke_plume.py", ["--size", "64", "--steps", "4", "--jit"]), ("taylor_green_3d.py", ["--size", "32", "--steps", "2"]),
                         ("viscous_taylor_green.py", ["--size", "64", "--steps", "4"])):
        r = subprocess.run([sys.executable, os.path.join(root, "examples", script)] + args, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "ms per step" in r.stdout, r.stderr[-2000:]
