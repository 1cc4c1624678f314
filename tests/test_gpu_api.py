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
    host.test_make_incompressible_matches_oracle(gpu_backend)
    host.test_obstacles_and_x0(gpu_backend)
    host.test_moving_and_rotating_obstacles(gpu_backend)
    host.test_convergence_exceptions(gpu_backend)
    host.test_lid_driven_cavity_boundaries_and_diffusion(gpu_backend)
    host.test_spatial_gradient_at_faces(gpu_backend)
    host.test_fp64_precision_context(gpu_backend)


def test_gradients(gpu_backend):
    """ adjoint kernels behind torch.autograd on the GPU vs finite differences of the forward path """
    host.test_make_incompressible_gradient(gpu_backend)
    host.test_functional_gradient_through_a_fluid_step(gpu_backend)
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
