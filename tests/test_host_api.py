"""
The phi-level mirror (`phiflow_amd.flow`) exercised the way the reference's own tests exercise PhiFlow
(/root/reference tests/commit/physics/test_fluid.py, test_advect.py, tests/commit/field/test__grid.py). Runs on CPU with the
kernel sources under the fiber emulation (test infrastructure); tests/test_gpu_api.py repeats the core of it on the MI355X.
"""
import numpy as np
import pytest

from phiflow_amd import _capi
from phiflow_amd.flow import (ZERO_GRADIENT, BOUNDARY, PERIODIC, ZERO, Box, CenteredGrid, Diverged, NotConverged, Obstacle, Solve, Sphere,
                              StaggeredGrid, advect, combine_sides, diffuse, divergence, fluid, spatial_gradient, vec)


def test_staggered_storage_sizes(emu_backend):
    """ tests/commit/field/test__grid.py:25-36 """
    for ext, nx in ((ZERO, 19), (PERIODIC, 20), (BOUNDARY, 21)):
        v = StaggeredGrid(0, ext, x=20, y=10, backend=emu_backend)
        assert v['x'].values.shape[1:] == (nx, 10)
        assert v['y'].values.shape[1:] == (20, nx - 10)
        assert v.resolution == {'x': 20, 'y': 10}


def test_with_extrapolation_restores_wall_faces(emu_backend):
    """ tests/commit/field/test__grid.py:85-94: BOUNDARY -> ZERO -> BOUNDARY leaves zeros on the wall faces """
    rng = np.random.default_rng(0)
    vals = [rng.standard_normal((21, 10)).astype(np.float32), rng.standard_normal((20, 11)).astype(np.float32)]
    grid = StaggeredGrid(vals, BOUNDARY, x=20, y=10, backend=emu_backend)
    grid_0 = grid.with_extrapolation(ZERO)
    assert grid_0['x'].values.shape[1:] == (19, 10)
    grid_ = grid_0.with_extrapolation(BOUNDARY)
    assert grid_.resolution == grid.resolution
    vx, vy = grid_.numpy()
    assert np.all(vx[0] == 0) and np.all(vx[-1] == 0) and np.all(vy[:, 0] == 0) and np.all(vy[:, -1] == 0)
    np.testing.assert_array_equal(vx[1:-1], vals[0][1:-1])


def test_self_advect_staggered_known_answer(emu_backend):
    """ tests/commit/physics/test_advect.py:41-45 """
    v0 = StaggeredGrid(Box(x=(.9, 2.6), y=(.9, 2)), 0, x=4, y=3, backend=emu_backend) * (0, 1)
    v = advect.semi_lagrangian(v0, v0, 1)
    np.testing.assert_allclose(v['x'].numpy(), 0, atol=1e-6)
    np.testing.assert_allclose(v['y'].numpy().T, [[0, 0, 0, 0], [0, 1, 1, 0]], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("ext", [ZERO, BOUNDARY, PERIODIC])
def test_identity_advection(emu_backend, ext):
    """ tests/commit/physics/test_advect.py:12-18 with the three schemes of :23-30 (advect, semi_lagrangian, mac_cormack) """
    rng = np.random.default_rng(1)
    shapes = StaggeredGrid(0, ext, x=4, y=3, backend=emu_backend).component_shapes
    sv = StaggeredGrid([rng.standard_normal(s).astype(np.float32) for s in shapes], ext, x=4, y=3, backend=emu_backend)
    s = CenteredGrid(rng.standard_normal((4, 3)).astype(np.float32), ext, x=4, y=3, backend=emu_backend)
    for adv in (advect.advect, advect.semi_lagrangian, advect.mac_cormack):
        for a, b in zip(adv(sv, sv, 0).numpy(), sv.numpy()):
            np.testing.assert_allclose(a, b, atol=1e-5)
        for a, b in zip(adv(sv, sv * 0, 1).numpy(), sv.numpy()):
            np.testing.assert_allclose(a, b, atol=1e-5)
        np.testing.assert_allclose(adv(s, sv, 0).numpy(), s.numpy(), atol=1e-5)
        np.testing.assert_allclose(adv(s, sv * 0, 1).numpy(), s.numpy(), atol=1e-5)


def _test_make_incompressible(backend, extrapolation, batch=None):
    """ tests/commit/physics/test_fluid.py:19-32 """
    rng = np.random.default_rng(2)
    bounds = Box['x,y', 0:100, 0:100]
    xs = rng.uniform(0, 100, size=batch or 1)
    smoke_vals = np.stack([CenteredGrid(Sphere(x=x0, y=10, radius=5), extrapolation, bounds, x=16, y=20, backend=backend).numpy() for x0 in xs])
    smoke = CenteredGrid(smoke_vals if batch else smoke_vals[0], extrapolation, bounds, x=16, y=20, backend=backend)
    velocity = StaggeredGrid(0, extrapolation, bounds, x=16, y=20, batch=batch, backend=backend)
    for _ in range(2):
        velocity += smoke * (0, 0.1) @ velocity
        velocity, pressure = fluid.make_incompressible(velocity)
    assert np.abs(divergence(velocity).numpy()).max() <= 5e-5
    assert pressure.is_centered and pressure.resolution == velocity.resolution
    return velocity


@pytest.mark.parametrize("name,ext", [("closed", ZERO), ("open", BOUNDARY), ("periodic", PERIODIC),
                                      ("mixed", combine_sides(x=BOUNDARY, y=(ZERO, BOUNDARY)))])
def test_make_incompressible_staggered(emu_backend, name, ext):
    """ tests/commit/physics/test_fluid.py:38-53 (unbatched and with a batch of 3) """
    _test_make_incompressible(emu_backend, ext)
    _test_make_incompressible(emu_backend, ext, batch=3)


@pytest.mark.parametrize("method", ['CG', 'CG-adaptive', 'auto'])
def test_make_incompressible_matches_oracle(emu_backend, method):
    """ 'CG-adaptive': Solve('CG-adaptive', 1e-5, x0=pressure) of examples/grids/Fluid_Logo.ipynb; 'auto' runs 'CG' """
    from oracle import phi_oracle as O
    rng = np.random.default_rng(3)
    bounds = Box['x,y', 0:100, 0:100]
    ext = combine_sides(x=BOUNDARY, y=(ZERO, BOUNDARY))
    shapes = StaggeredGrid(0, ext, bounds, x=16, y=20, backend=emu_backend).component_shapes
    vals = [rng.standard_normal(s).astype(np.float32) * 0.1 for s in shapes]
    v, p = fluid.make_incompressible(StaggeredGrid(vals, ext, bounds, x=16, y=20, backend=emu_backend), (), Solve(method, 1e-5, 0))
    dom = O.Domain((16, 20), (0, 0), (100, 100), ((O.OPEN, O.OPEN), (O.CLOSED, O.OPEN)))
    vo, po, info, _ = O.make_incompressible([a[None] for a in vals], dom, rtol=1e-5, atol=0, method='CG-adaptive' if method == 'CG-adaptive' else 'CG')
    assert abs(p.solve_info.iterations[0] - int(info.iterations[0])) <= (0 if method == 'CG' else 1)
    np.testing.assert_allclose(p.numpy(), po[0], atol=2e-4 * np.abs(po).max())
    for a, b in zip(v.numpy(), vo):
        np.testing.assert_allclose(a, b[0], atol=1e-5)


def test_obstacles_and_x0(emu_backend):
    """ Batched_Smoke / Lid_Driven_Cavity style: box obstacle in a closed domain, warm start from the previous pressure """
    rng = np.random.default_rng(4)
    bounds = Box(x=32, y=32)
    shapes = StaggeredGrid(0, ZERO, bounds, x=32, y=32, backend=emu_backend).component_shapes
    v = StaggeredGrid([rng.standard_normal(s).astype(np.float32) * 0.1 for s in shapes], ZERO, bounds, x=32, y=32, backend=emu_backend)
    obstacle = Obstacle(Box(x=(12, 20), y=(10, 16)))
    v1, p1 = fluid.make_incompressible(v, obstacle, Solve('CG', 1e-5, 0))
    vx, vy = v1.numpy()
    # faces inside / on the obstacle carry no flow
    assert np.abs(vx[12:20, 10:16]).max() == 0
    div = divergence(v1).numpy()
    active = np.ones((32, 32), bool); active[12:20, 10:16] = False
    assert np.abs(div[active]).max() <= 5e-5
    it_cold = p1.solve_info.iterations[0]
    v2, p2 = fluid.make_incompressible(v, [obstacle], Solve('CG', 1e-5, 0, x0=p1))
    assert p2.solve_info.iterations[0] <= max(2, it_cold // 4)       # warm start converges almost immediately
    np.testing.assert_allclose(p2.numpy(), p1.numpy(), atol=1e-3 * np.abs(p1.numpy()).max())


def test_moving_and_rotating_obstacles(emu_backend):
    """ Moving_Obstacles.ipynb cells 3, 7 and Rotating_Bar.ipynb cells 3, 5 (two steps each) against the oracle:
    obstacles move / rotate between steps, mac_cormack self-advection, projection with Solve(x0=p) """
    from oracle import phi_oracle as O
    from phiflow_amd.flow import Cuboid
    n = 32
    # --- moving box + sphere in a periodic domain ---
    bounds = Box(x=100, y=100)
    obstacles = [Obstacle(Cuboid(vec(x=20, y=80), x=20, y=20), velocity=vec(x=5., y=0)),
                 Obstacle(Sphere(x=20, y=20, radius=10), velocity=vec(x=1, y=4))]
    dom = O.Domain((n, n), (0, 0), (100, 100), ((O.PERIODIC, O.PERIODIC),) * 2)
    o_obs = [O.BoxObstacle((10, 70), (30, 90), velocity=(5., 0.)), O.SphereObstacle((20, 20), 10, velocity=(1., 4.))]
    v = StaggeredGrid(0, PERIODIC, bounds, x=n, y=n, backend=emu_backend)
    vo = [np.zeros((1,) + dom.comp_shape(d), np.float32) for d in range(2)]
    p, po, dt = None, None, 0.5
    for _ in range(2):
        obstacles = [ob.at([(c + u * dt) % 100 for c, u in zip(ob.geometry.center, ob.velocity)]) for ob in obstacles]
        o_obs = [O.BoxObstacle(tuple(l + u * dt for l, u in zip(ob.lower, ob.velocity)), tuple(h + u * dt for h, u in zip(ob.upper, ob.velocity)),
                               velocity=ob.velocity) if isinstance(ob, O.BoxObstacle) else
                 O.SphereObstacle(tuple(c + u * dt for c, u in zip(ob.center, ob.velocity)), ob.radius, velocity=ob.velocity) for ob in o_obs]
        v = advect.mac_cormack(v, v, dt)
        v, p = fluid.make_incompressible(v, obstacles, Solve('CG', 1e-4, 0, x0=p))
        vo = O.mac_cormack_staggered(vo, vo, dt, dom)
        vo, po, info, _ = O.make_incompressible(vo, dom, o_obs, x0=po, rtol=1e-4, atol=0)
        assert abs(p.solve_info.iterations[0] - int(info.iterations[0])) <= max(3, 0.1 * int(info.iterations[0]))   # warm-started
    for a, b in zip(v.numpy(), vo):
        np.testing.assert_allclose(a, b[0], atol=1e-3 * np.abs(b).max())
    assert np.abs(vo[0]).max() > 1.0        # the moving obstacles did stir the fluid
    # --- rotating bar, open domain ---
    bar = Obstacle(Cuboid(vec(x=50, y=50), x=6, y=60), angular_velocity=0.05)
    dom = O.Domain((n, n), (0, 0), (100, 100), ((O.OPEN, O.OPEN),) * 2)
    v = StaggeredGrid(0, ZERO_GRADIENT, bounds, x=n, y=n, backend=emu_backend)
    vo = [np.zeros((1,) + dom.comp_shape(d), np.float32) for d in range(2)]
    p, po, angle = None, None, 0.0
    for _ in range(2):
        bar = bar.rotated(bar.angular_velocity[0] * 1.0)
        angle += 0.05
        R = [[np.cos(angle), -np.sin(angle)], [np.sin(angle), np.cos(angle)]]
        o_bar = [O.BoxObstacle((47, 20), (53, 80), angular_velocity=0.05, rotation=R)]
        v = advect.mac_cormack(v, v, 1.0)
        v, p = fluid.make_incompressible(v, bar, Solve('CG', 1e-4, 0, x0=p))
        vo = O.mac_cormack_staggered(vo, vo, 1.0, dom)
        vo, po, info, _ = O.make_incompressible(vo, dom, o_bar, x0=po, rtol=1e-4, atol=0)
    for a, b in zip(v.numpy(), vo):
        np.testing.assert_allclose(a, b[0], atol=1e-3 * np.abs(b).max())
    assert np.abs(vo[1]).max() > 0.5


def test_fluid_logo_union_obstacle_and_cg_adaptive(emu_backend):
    """ examples/grids/Fluid_Logo.ipynb (cells 2-5) at 32^2: obstacle = union(boxes), inflow = CenteredGrid(Box), smoke and velocity
    advected semi-Lagrangian, buoyancy resampled to the faces, Solve('CG-adaptive', 1e-5, x0=pressure) starting from pressure None """
    from oracle import phi_oracle as O
    from phiflow_amd.flow import resample, union
    n = 32
    domain = dict(x=n, y=n, bounds=Box(x=100, y=100))
    geometries = [Box(x=(15 + x * 7, 15 + (x + 1) * 7), y=(41, 83)) for x in range(1, 10, 2)] + [Box['x,y', 43:50, 41:48], Box['x,y', 15:43, 83:90], Box['x,y', 50:85, 83:90]]
    geometry = union(geometries)
    assert union(geometry, geometries[0]).geometries[-1] is geometries[0] and union([geometries[0]]) is geometries[0]
    inflow = CenteredGrid(Box(x=(14, 21), y=(6, 10)), ZERO_GRADIENT, backend=emu_backend, **domain) + \
        CenteredGrid(Box(x=(81, 88), y=(6, 10)), ZERO_GRADIENT, backend=emu_backend, **domain) * 0.9
    v = StaggeredGrid(0, 0, backend=emu_backend, **domain)
    smoke = CenteredGrid(0, ZERO_GRADIENT, backend=emu_backend, **domain)
    pressure = None
    dom = O.Domain((n, n), (0, 0), (100, 100), ((O.CLOSED, O.CLOSED),) * 2)
    o_geo = [O.UnionObstacle(tuple(O.BoxObstacle(g.lower, g.upper) for g in geometries))]
    s_codes = ((O.OPEN, O.OPEN),) * 2
    vo = [np.zeros((1,) + dom.comp_shape(d), np.float32) for d in range(2)]
    so, po = np.zeros((1, n, n), np.float32), None
    inflow_o = inflow.numpy()[None].astype(np.float32)
    for _ in range(3):
        smoke = advect.semi_lagrangian(smoke, v, 1) + inflow
        buoyancy_force = resample(smoke * (0, 0.1), to=v)
        v = advect.semi_lagrangian(v, v, 1) + buoyancy_force
        v, pressure = fluid.make_incompressible(v, geometry, Solve('CG-adaptive', 1e-5, x0=pressure))
        so = O.semi_lagrangian_centered(so, vo, 1.0, dom, s_codes) + inflow_o
        bo = O.centered_to_staggered(so, dom, s_codes, vector=(0.0, 0.1))
        vo = [a + b for a, b in zip(O.semi_lagrangian_staggered(vo, vo, 1.0, dom), bo)]
        vo, po, info, _ = O.make_incompressible(vo, dom, o_geo, x0=po, rtol=1e-5, atol=1e-5, method='CG-adaptive')   # abs_tol defaults to 1e-5 (fp32)
        assert abs(pressure.solve_info.iterations[0] - int(info.iterations[0])) <= max(3, 0.1 * int(info.iterations[0]))
    np.testing.assert_allclose(smoke.numpy(), so[0], atol=1e-5)
    for a, b in zip(v.numpy(), vo):
        np.testing.assert_allclose(a, b[0], atol=2e-4 * np.abs(np.concatenate([c.ravel() for c in vo])).max())
    assert np.abs(vo[1]).max() > 0.02                                                   # the plumes rise
    inside = geometry.lies_inside(np.meshgrid(*[(np.arange(n) + 0.5) * 100 / n] * 2, indexing='ij'))
    assert inside.sum() > 50 and np.abs(pressure.numpy()[inside]).max() == 0.0         # inactive cells keep x0 = 0 (fluid.py:202)


def test_wake_flow_inflow_boundary_and_infinite_cylinder(emu_backend):
    """ examples/grids/Wake_Flow.ipynb at 32 x 16 x 4: boundary dict with a constant inflow at x-, open x+, periodic y / z; obstacle =
    geom.infinite_cylinder; step = semi-Lagrangian self-advection + make_incompressible(v, cylinder, Solve(x0=p)) """
    from oracle import phi_oracle as O
    from phiflow_amd.flow import geom
    cylinder = geom.infinite_cylinder(x=20, y=50, radius=10, inf_dim='z')
    assert cylinder.dims == ('x', 'y', 'z') and geom.embed(Sphere(x=1, y=2, radius=1), 'x,y') .dims == ('x', 'y')
    boundary = {'x-': vec(x=2, y=0, z=0), 'x+': ZERO_GRADIENT, 'y': PERIODIC, 'z': PERIODIC}
    v = StaggeredGrid((8., 0, 0), boundary, x=32, y=16, z=4, bounds=Box(x=200, y=100, z=5), backend=emu_backend)
    dom = O.Domain((32, 16, 4), (0, 0, 0), (200, 100, 5), ((O.CLOSED, O.OPEN), (O.PERIODIC, O.PERIODIC), (O.PERIODIC, O.PERIODIC)),
                   [[[2.0, 0.0, 0.0], [0.0] * 3], [[0.0] * 3] * 2, [[0.0] * 3] * 2])
    o_cyl = [O.EmbeddedObstacle(O.SphereObstacle((20.0, 50.0), 10.0), (0, 1))]
    vo = [np.full((1,) + dom.comp_shape(d), 8.0 if d == 0 else 0.0, np.float32) for d in range(3)]
    v, p = fluid.make_incompressible(v, cylinder, Solve())
    vo, po, info, _ = O.make_incompressible(vo, dom, o_cyl, rtol=1e-5, atol=1e-5)
    for _ in range(2):
        v = advect.semi_lagrangian(v, v, 1.)
        v, p = fluid.make_incompressible(v, cylinder, Solve(x0=p))
        vo = O.semi_lagrangian_staggered(vo, vo, 1.0, dom)
        vo, po, info, _ = O.make_incompressible(vo, dom, o_cyl, x0=po, rtol=1e-5, atol=1e-5)
        assert abs(p.solve_info.iterations[0] - int(info.iterations[0])) <= max(3, 0.1 * int(info.iterations[0]))
    for a, b in zip(v.numpy(), vo):
        np.testing.assert_allclose(a, b[0], atol=2e-4 * 8.0)
    assert np.abs(vo[1]).max() > 1.0 and np.abs(vo[2]).max() == 0.0        # the flow goes around the cylinder, nothing along z
    np.testing.assert_allclose(p.numpy(), po[0], atol=2e-3 * np.abs(po).max())


def test_batched_smoke_with_batched_obstacle_and_inflow(emu_backend):
    """ examples/grids/Batched_Smoke.ipynb at 32^2 (smoke on the velocity's grid): three settings that differ in the obstacle position
    ("this affects the pressure matrix"), the inflow position and the inflow rate; every batch entry equals its own oracle run """
    from oracle import phi_oracle as O
    from phiflow_amd.flow import Cuboid, resample
    n, B = 32, 3
    domain = Box(x=100, y=100)
    inflow_rate = np.array([.1, .2, .3])
    inflow_x, obstacle_x = [40, 50, 60], [15, 50, 70]
    obstacle = Cuboid(vec(x=obstacle_x, y=60), half_size=vec(x=15, y=10))
    inflow = Sphere(x=inflow_x, y=9.5, radius=5)
    assert obstacle.batch_size == 3 and inflow.batch_size == 3 and repr(obstacle.entry(2)) == repr(Cuboid(vec(x=70, y=60), x=30, y=20))
    v = StaggeredGrid(0, 0, domain, x=n, y=n, backend=emu_backend)
    s = CenteredGrid(0, ZERO_GRADIENT, domain, x=n, y=n, backend=emu_backend)
    p = None
    for _ in range(3):
        s = advect.mac_cormack(s, v, 1.) + inflow_rate * resample(inflow, to=s, soft=True)
        buoyancy = resample(s * (0, 0.1), to=v)
        v = advect.semi_lagrangian(v, v, 1.) + buoyancy * 1.
        v, p = fluid.make_incompressible(v, obstacle, Solve(x0=p))
    assert v.batch_size == B and s.batch_size == B and p.batch_size == B
    dom = O.Domain((n, n), (0, 0), (100, 100), ((O.CLOSED, O.CLOSED),) * 2)
    s_codes = ((O.OPEN, O.OPEN),) * 2
    radius = float(np.hypot(100 / n / 2, 100 / n / 2))
    for b in range(B):
        o_obs = [O.BoxObstacle((obstacle_x[b] - 15, 50), (obstacle_x[b] + 15, 70))]
        pts = O.cell_positions(dom, np.float64)
        mask = np.clip(0.5 - O.SphereObstacle((inflow_x[b], 9.5), 5).sdf(pts) / radius, 0, 1).astype(np.float32)[None]
        vo = [np.zeros((1,) + dom.comp_shape(d), np.float32) for d in range(2)]
        so, po = np.zeros((1, n, n), np.float32), None
        for _ in range(3):
            so = O.mac_cormack_centered(so, vo, 1.0, dom, s_codes) + np.float32(inflow_rate[b]) * mask
            bo = O.centered_to_staggered(so, dom, s_codes, vector=(0.0, 0.1))
            vo = [a + c for a, c in zip(O.semi_lagrangian_staggered(vo, vo, 1.0, dom), bo)]
            vo, po, info, _ = O.make_incompressible(vo, dom, o_obs, x0=po, rtol=1e-5, atol=1e-5)
        assert abs(p.solve_info.iterations[b] - int(info.iterations[0])) <= max(3, 0.1 * int(info.iterations[0]))
        np.testing.assert_allclose(s.numpy()[b], so[0], atol=1e-5)
        scale = max(np.abs(c).max() for c in vo)
        for a, c in zip(v.numpy(), vo):
            np.testing.assert_allclose(a[b], c[0], atol=3e-4 * scale)
    assert not np.allclose(v.numpy()[1][0], v.numpy()[1][1])                # the settings do differ


@pytest.mark.parametrize("dtype_name", ["float32", "float64"])
def test_fields_on_different_grids(emu_backend, dtype_name):
    """ examples/grids/Batched_Smoke.ipynb samples smoke (200^2) and velocity (64^2) on different grids: mac_cormack / semi_lagrangian of
    the smoke by the coarse velocity and resample(smoke * (0, 0.1), to=velocity) go through math.grid_sample at explicit points
    (phi/physics/advect.py:193, phi/field/_resample.py:66-72,241-259) -- here 40^2 / 16^2 against the oracle's restatement """
    from oracle import phi_oracle as O
    from phiflow_amd.flow import precision, resample
    dtype = np.dtype(dtype_name).type
    rng = np.random.default_rng(17)
    with precision(64 if dtype_name == "float64" else 32):
        domain = Box(x=100, y=100)
        ext = combine_sides(x=0, y=(0, BOUNDARY))
        shapes = StaggeredGrid(0, ext, domain, x=16, y=16, backend=emu_backend).component_shapes
        v_np = [rng.standard_normal((2,) + sh).astype(dtype) * 4 for sh in shapes]
        s_np = rng.standard_normal((40, 40)).astype(dtype)
        v = StaggeredGrid(v_np, ext, domain, x=16, y=16, backend=emu_backend)
        s = CenteredGrid(s_np, ZERO_GRADIENT, domain, x=40, y=40, backend=emu_backend)
        dom_v = O.Domain((16, 16), (0, 0), (100, 100), ((O.CLOSED, O.CLOSED), (O.CLOSED, O.OPEN)))
        dom_s = O.Domain((40, 40), (0, 0), (100, 100), ((O.CLOSED, O.CLOSED),) * 2)
        s_codes = ((O.OPEN, O.OPEN),) * 2
        tol = 2e-5 if dtype_name == "float32" else 1e-12
        sl = advect.semi_lagrangian(s, v, 1.5)
        ref = O.semi_lagrangian_centered_general(s_np[None], dom_s, v_np, dom_v, 1.5, s_codes)
        assert sl.batch_size == 2 and sl.resolution == s.resolution
        np.testing.assert_allclose(sl.numpy(), ref, atol=tol * np.abs(ref).max())
        mc = advect.mac_cormack(s, v, 1.5, correction_strength=0.8)
        ref = O.semi_lagrangian_centered_general(s_np[None], dom_s, v_np, dom_v, 1.5, s_codes, correction_strength=0.8)
        assert (np.abs(mc.numpy() - ref) > tol * 10 * np.abs(ref).max()).mean() < 5e-3       # clamp windows may flip at cell boundaries
        assert mc.numpy().max() <= s_np.max() + 1e-6 and mc.numpy().min() >= s_np.min() - 1e-6
        buoyancy = resample(s * (0, 0.1), to=v)
        ref = O.resample_centered_general(s_np[None], dom_s, s_codes, None, dom_v, staggered=True, vector=(0.0, 0.1))
        assert [tuple(c.shape) for c in buoyancy.values] == [(1,) + sh for sh in shapes]
        for a, b in zip(buoyancy.numpy(), ref):
            np.testing.assert_allclose(a, b[0], atol=tol)
        fine = resample(CenteredGrid(v_np[0][0, :, :15] if False else rng.standard_normal((16, 16)).astype(dtype), 1.5, domain, x=16, y=16, backend=emu_backend), to=s)
        assert fine.resolution == s.resolution and fine.boundary == s.boundary
        # velocity advected by a velocity on a finer grid, component by component
        shapes2 = StaggeredGrid(0, ext, domain, x=24, y=20, backend=emu_backend).component_shapes
        w_np = [rng.standard_normal((1,) + sh).astype(dtype) * 3 for sh in shapes2]
        w = StaggeredGrid(w_np, ext, domain, x=24, y=20, backend=emu_backend)
        adv = advect.semi_lagrangian(v, w, 0.7)
        dom_w = O.Domain((24, 20), (0, 0), (100, 100), ((O.CLOSED, O.CLOSED), (O.CLOSED, O.OPEN)))
        for d in range(2):
            pts = [np.broadcast_to(p[None], (2,) + p.shape) for p in O.face_positions(d, dom_v, dtype)]
            u = O.sample_staggered_at([np.broadcast_to(c, (2,) + c.shape[1:]) for c in w_np], dom_w, pts)
            back = [p + uc * dtype(-0.7) for p, uc in zip(pts, u)]
            codes, consts = O._comp_codes(dom_v, d)
            ref = O.grid_sample(v_np[d], O._index_coords(back, d, dom_v, dtype), codes, consts)
            np.testing.assert_allclose(adv.numpy()[d], ref, atol=tol * 10 * np.abs(ref).max())
    with pytest.raises(NotImplementedError):
        advect.mac_cormack(v, w, 0.5)


@pytest.mark.parametrize("dtype_name", ["float32", "float64"])
def test_rk4_integrator(emu_backend, dtype_name):
    """ `advect.semi_lagrangian(..., integrator=advect.rk4)` / `mac_cormack(..., integrator=advect.rk4)` (phi/physics/advect.py:27-36):
    four velocity evaluations per back-trace; staggered self-advection and a centred scalar, same grid and a coarser velocity grid """
    from oracle import phi_oracle as O
    from phiflow_amd.flow import precision
    dtype = np.dtype(dtype_name).type
    rng = np.random.default_rng(23)
    tol = 3e-5 if dtype_name == "float32" else 1e-12
    with precision(64 if dtype_name == "float64" else 32):
        domain = Box(x=100, y=80)
        ext = combine_sides(x=PERIODIC, y=(0, BOUNDARY))
        shapes = StaggeredGrid(0, ext, domain, x=20, y=16, backend=emu_backend).component_shapes
        v_np = [rng.standard_normal((1,) + sh).astype(dtype) * 5 for sh in shapes]
        v = StaggeredGrid(v_np, ext, domain, x=20, y=16, backend=emu_backend)
        dom_v = O.Domain((20, 16), (0, 0), (100, 80), ((O.PERIODIC, O.PERIODIC), (O.CLOSED, O.OPEN)))
        adv = advect.semi_lagrangian(v, v, 1.2, integrator=advect.rk4)
        ref = O.semi_lagrangian_staggered_general(v_np, dom_v, v_np, dom_v, 1.2, integrator='rk4')
        for a, b in zip(adv.numpy(), ref):
            np.testing.assert_allclose(a, b, atol=tol * np.abs(b).max())
        euler_ref = O.semi_lagrangian_staggered(v_np, v_np, 1.2, dom_v)
        assert max(np.abs(a - b).max() for a, b in zip(adv.numpy(), euler_ref)) > 1e-2             # it is not the Euler back-trace
        same = advect.semi_lagrangian(v, v, 1.2, integrator=advect.finite_rk4)
        for a, b in zip(same.numpy(), adv.numpy()):
            np.testing.assert_allclose(a, b, atol=0)
        s_np = rng.standard_normal((2, 50, 40)).astype(dtype)
        s = CenteredGrid(s_np, combine_sides(x=PERIODIC, y=ZERO_GRADIENT), domain, x=50, y=40, backend=emu_backend)
        dom_s = O.Domain((50, 40), (0, 0), (100, 80), ((O.PERIODIC, O.PERIODIC), (O.CLOSED, O.CLOSED)))
        s_codes = ((O.PERIODIC, O.PERIODIC), (O.OPEN, O.OPEN))
        mc = advect.mac_cormack(s, v, 0.9, integrator=advect.rk4)
        ref = O.semi_lagrangian_centered_general(s_np, dom_s, v_np, dom_v, 0.9, s_codes, correction_strength=1.0, integrator='rk4')
        assert (np.abs(mc.numpy() - ref) > tol * 10 * np.abs(ref).max()).mean() < 5e-3
    with pytest.raises(NotImplementedError):
        advect.semi_lagrangian(v, v, 1.0, integrator=lambda *a: None)


def test_user_active_mask_plain_and_batched(emu_backend):
    """ make_incompressible(..., active=CenteredGrid) (phi/physics/fluid.py:97,139-148,200-202): the pressure is only solved where
    active != 0 (identity rows elsewhere), the divergence is never balanced; one mask for all entries, and one mask per batch entry """
    from oracle import phi_oracle as O
    rng = np.random.default_rng(29)
    n = 16
    bounds = Box(x=1, y=1)
    shapes = StaggeredGrid(0, 0, bounds, x=n, y=n, backend=emu_backend).component_shapes
    v_np = [rng.standard_normal((2,) + sh).astype(np.float32) for sh in shapes]
    v = StaggeredGrid(v_np, 0, bounds, x=n, y=n, backend=emu_backend)
    masks = np.ones((2, n, n), np.float32)
    masks[0, 3:6, 4:9] = 0
    masks[1, 10:14, 2:5] = 0
    dom = O.Domain((n, n), (0, 0), (1, 1), ((O.CLOSED, O.CLOSED),) * 2)

    def oracle(vel, act):
        div = O.divergence(vel, dom) * act
        A = lambda q: O.masked_laplace(q, dom, None, act)
        p, info = O.cg(A, div, np.zeros_like(div), 1e-5, 1e-5, 1000, 50)
        return O.gradient_subtract(vel, p, dom, None), p, info
    for act_np in (masks[:1], masks):                          # shared mask, per-entry masks
        active = CenteredGrid(act_np if act_np.shape[0] > 1 else act_np[0], 0, bounds, x=n, y=n, backend=emu_backend)
        v_new, p = fluid.make_incompressible(v, (), Solve('CG'), active=active)
        vo, po, info = oracle(v_np, np.broadcast_to(act_np, masks.shape))
        assert [abs(a - int(b)) <= 2 for a, b in zip(p.solve_info.iterations, info.iterations)] == [True, True]
        np.testing.assert_allclose(p.numpy(), po, atol=3e-4 * np.abs(po).max())
        for a, b in zip(v_new.numpy(), vo):
            np.testing.assert_allclose(a, b, atol=3e-4 * np.abs(b).max())
        inactive = np.broadcast_to(act_np, masks.shape) == 0
        assert np.abs(p.numpy()[inactive]).max() == 0.0


def test_user_active_mask_with_nan_velocity(emu_backend):
    """ "If given [`active`], the velocity may take NaN values where it does not contribute to the pressure" (phi/physics/fluid.py:112-113):
    `div = field.where(field.is_finite(div), div, 0)` (fluid.py:143-144) zeroes the non-finite divergence on EVERY cell -- also next to an
    active one -- before the solve; against the oracle's restatement of those lines (oracle.make_incompressible(active_user=...)). """
    from oracle import phi_oracle as O
    rng = np.random.default_rng(31)
    n = 16
    bounds = Box(x=1, y=1)
    shapes = StaggeredGrid(0, 0, bounds, x=n, y=n, backend=emu_backend).component_shapes
    v_np = [rng.standard_normal((1,) + sh).astype(np.float32) for sh in shapes]
    mask = np.ones((1, n, n), np.float32)
    mask[0, 3:7, 4:9] = 0
    v_np[0][0, 4, 5] = np.nan            # x-face between two inactive cells
    v_np[1][0, 3, 8] = np.nan            # y-face between an inactive (3, 8) and an ACTIVE cell (3, 9): its divergence is NaN on an active cell
    v_np[0][0, 10, 10] = np.nan          # far from the mask: both neighbours active
    dom = O.Domain((n, n), (0, 0), (1, 1), ((O.CLOSED, O.CLOSED),) * 2)
    v = StaggeredGrid([a[0] for a in v_np], 0, bounds, x=n, y=n, backend=emu_backend)
    active = CenteredGrid(mask[0], 0, bounds, x=n, y=n, backend=emu_backend)
    v_new, p = fluid.make_incompressible(v, (), Solve('CG', 1e-5, 1e-5), active=active)
    vo, po, info, rhs = O.make_incompressible(v_np, dom, (), None, 1e-5, 1e-5, 1000, active_user=mask)
    assert np.isfinite(po).all() and np.isfinite(p.numpy()).all(), "the guard must keep NaN out of the solve"
    assert abs(p.solve_info.iterations[0] - int(info.iterations[0])) <= 2
    np.testing.assert_allclose(p.numpy(), po[0], atol=3e-4 * np.abs(po).max())
    for a, b in zip(v_new.numpy(), vo):
        assert (np.isnan(a) == np.isnan(b[0])).all()                 # the NaN samples stay NaN (v - grad p), nothing else does
        np.testing.assert_allclose(np.nan_to_num(a), np.nan_to_num(b[0]), atol=3e-4 * np.nanmax(np.abs(b)))
    # without `active` the reference offers no such guarantee: a NaN next to an active cell reaches the solver (here: reported as diverged)
    with pytest.raises((Diverged, NotConverged)):
        fluid.make_incompressible(v, (), Solve('CG', 1e-5, 1e-5, max_iterations=20))


def test_convergence_exceptions(emu_backend):
    """ phiml.math.solve_linear raises NotConverged / Diverged unless suppressed (tests/commit/physics/test_diffuse.py:60-66) """
    rng = np.random.default_rng(5)
    shapes = StaggeredGrid(0, PERIODIC, x=16, y=16, backend=emu_backend).component_shapes
    v = StaggeredGrid([rng.standard_normal(s).astype(np.float32) for s in shapes], PERIODIC, x=16, y=16, backend=emu_backend)
    with pytest.raises(NotConverged) as e:
        fluid.make_incompressible(v, (), Solve('CG', 1e-6, 0, max_iterations=2))
    assert e.value.result.iterations == [2]
    v2, p2 = fluid.make_incompressible(v, (), Solve('CG', 0, 0, max_iterations=7, suppress=[NotConverged]))   # benchmark mode
    assert p2.solve_info.iterations == [7]
    with pytest.raises(NotImplementedError):
        fluid.make_incompressible(v, (), Solve('biCG-stab(2)', 1e-5))
    with pytest.raises(NotImplementedError):
        fluid.make_incompressible(v, (), order=4)


def test_implicit_diffusion(emu_backend):
    """ diffuse.implicit the way tests/commit/physics/test_diffuse.py:30-66 uses it (2-D instead of 1-D grids: the backend has rank 2 / 3),
    Heat_Flow.ipynb (constant boundary temperature) and a staggered velocity with a lid, against the oracle """
    from oracle import phi_oracle as O
    # test_constant_diffusion: a constant stays constant (the solve starts converged)
    grid = CenteredGrid(1, PERIODIC, x=5, y=5, backend=emu_backend)
    out = diffuse.implicit(grid, 1, 1)
    np.testing.assert_allclose(out.numpy(), 1.0, atol=1e-6)
    assert out.solve_info.iterations == [0]
    # test_equality_1d_periodic / test_implicit_stability: explicit with substeps ~ implicit; maximum principle at a large diffusivity
    step = np.zeros((40, 8), np.float32); step[:20] = 1
    grid = CenteredGrid(step, PERIODIC, x=40, y=8, backend=emu_backend)
    ex, im = diffuse.explicit(grid, 0.5, 1, substeps=10), diffuse.implicit(grid, 0.5, 1)
    assert np.abs(ex.numpy() - im.numpy()).max() <= 0.1        # (one implicit Euler step vs ten explicit ones: 0.064 next to the jump, the analytic difference)
    assert abs(float(im.numpy()[19, 0]) - 0.5 * (1 + 3 ** -0.5)) <= 1e-5      # closed form of the implicit step at the jump for k dt / dx^2 = 1/2
    stiff = diffuse.implicit(grid, 10, 1, Solve('CG', 1e-6, 0)).numpy()
    assert stiff.min() >= -1e-4 and stiff.max() <= 1.0001
    # Heat_Flow.ipynb: constant temperature on one side, insulated (zero-gradient) on the others; the constant is the affine part
    rng = np.random.default_rng(31)
    t0 = rng.standard_normal((12, 10)).astype(np.float32)
    ext = combine_sides(x=(1.0, ZERO_GRADIENT), y=ZERO_GRADIENT)
    t = CenteredGrid(t0, ext, x=12, y=10, bounds=Box['x,y', 0:6, 0:5], backend=emu_backend)
    out = diffuse.implicit(t, 0.8, 0.5, Solve('CG', 1e-6, 0))
    dom = O.Domain((12, 10), (0, 0), (6, 5), ((O.CLOSED, O.OPEN), (O.OPEN, O.OPEN)))
    ref, info = O.diffuse_implicit_centered(t0[None], 0.8, 0.5, dom, ((O.CLOSED, O.OPEN), (O.OPEN, O.OPEN)), [(1.0, 0.0), (0.0, 0.0)], 1e-6, 0.0, 1000)
    np.testing.assert_allclose(out.numpy(), ref[0], atol=2e-5)
    assert abs(out.solve_info.iterations[0] - int(info.iterations[0])) <= 2
    # staggered velocity in a lid-driven cavity: every component on its own lattice, the lid value drags the top row
    boundary = {'x': 0, 'y-': 0, 'y+': vec(x=1, y=0)}
    v = StaggeredGrid(0, boundary, x=8, y=8, backend=emu_backend)
    v = diffuse.implicit(v, 0.1, 1.0, Solve('CG', 1e-6, 0))
    bcv = np.zeros((2, 2, 2)); bcv[1, 1, 0] = 1.0
    dom = O.Domain((8, 8), (0, 0), (8, 8), ((O.CLOSED, O.CLOSED),) * 2, bcv)
    ref, _ = O.diffuse_implicit([np.zeros((1, 7, 8), np.float32), np.zeros((1, 8, 7), np.float32)], 0.1, 1.0, dom, 1e-6, 0.0, 1000)
    vx, vy = v.numpy()
    np.testing.assert_allclose(vx, ref[0][0], atol=2e-6)
    np.testing.assert_allclose(vy, ref[1][0], atol=2e-6)
    assert vx[:, -1].min() > 0.05 and np.all(vy == 0)
    # solve_linear semantics: NotConverged unless suppressed; what the backend does not do is refused
    with pytest.raises(NotConverged):
        diffuse.implicit(t, 50.0, 1.0, Solve('CG', 1e-7, 0, max_iterations=2))
    diffuse.implicit(t, 50.0, 1.0, Solve('CG', 1e-7, 0, max_iterations=2, suppress=[NotConverged]))
    with pytest.raises(Exception):
        diffuse.implicit(t, 0.5, -1.0)


def test_lid_driven_cavity_boundaries_and_diffusion(emu_backend):
    """ Lid_Driven_Cavity.ipynb cell 5 boundary: {'x': 0, 'y-': 0, 'y+': vec(x=1, y=0)}; diffuse.explicit pads with it """
    from oracle import phi_oracle as O
    boundary = {'x': 0, 'y-': 0, 'y+': vec(x=1, y=0)}
    v = StaggeredGrid(0, boundary, x=8, y=8, backend=emu_backend)
    v = diffuse.explicit(v, 0.1, 1.0)
    vx, vy = v.numpy()
    bcv = np.zeros((2, 2, 2)); bcv[1, 1, 0] = 1.0
    dom = O.Domain((8, 8), (0, 0), (8, 8), ((O.CLOSED, O.CLOSED),) * 2, bcv)
    ref = O.diffuse_explicit([np.zeros((1, 7, 8), np.float32), np.zeros((1, 8, 7), np.float32)], 0.1, 1.0, dom)
    np.testing.assert_allclose(vx, ref[0][0], atol=1e-7)
    np.testing.assert_allclose(vy, ref[1][0], atol=1e-7)
    assert vx[:, -1].max() > 0 and np.all(vx[:, :-1] == 0)       # the lid drags the top row of x-faces
    v, p = fluid.make_incompressible(advect.semi_lagrangian(v, v, 0.5), (), Solve('CG', 1e-5, 0))
    assert np.abs(divergence(v).numpy()).max() <= 5e-5


def test_spatial_gradient_at_faces(emu_backend):
    from oracle import phi_oracle as O
    from phiflow_amd.extrapolation import pressure_extrapolation
    rng = np.random.default_rng(6)
    vb = combine_sides(x=PERIODIC, y=(ZERO, BOUNDARY))
    p = CenteredGrid(rng.standard_normal((8, 6)).astype(np.float32), pressure_extrapolation(vb, ('x', 'y')), Box(x=4, y=3), x=8, y=6,
                     backend=emu_backend)
    g = spatial_gradient(p, vb, at='face')
    dom = O.Domain((8, 6), (0, 0), (4, 3), ((O.PERIODIC, O.PERIODIC), (O.CLOSED, O.OPEN)))
    ref = O.pressure_gradient(p.numpy()[None], dom)
    for a, b in zip(g.numpy(), ref):
        np.testing.assert_allclose(a, b[0], rtol=1e-5, atol=1e-6)


def test_fp64_precision_context(emu_backend):
    from phiflow_amd.flow import precision
    import torch
    with precision(64):
        v = StaggeredGrid(lambda x, y: (np.sin(x), np.cos(y)), PERIODIC, Box(x=2 * np.pi, y=2 * np.pi), x=16, y=16, backend=emu_backend)
        assert v.dtype == torch.float64
        v, p = fluid.make_incompressible(v, (), Solve('CG', 1e-12, 1e-12))
        assert np.abs(divergence(v).numpy()).max() <= 1e-10


def _fd_gradient_check(loss_of_values, values, grads, rng, eps=1e-6, tol=2e-5, n_dirs=3):
    """ directional central differences of a scalar python function of a list of float64 arrays vs analytic gradients """
    for _ in range(n_dirs):
        d = [rng.standard_normal(v.shape) for v in values]
        plus = loss_of_values([v + eps * di for v, di in zip(values, d)])
        minus = loss_of_values([v - eps * di for v, di in zip(values, d)])
        fd = (plus - minus) / (2 * eps)
        an = float(sum(np.vdot(g, di) for g, di in zip(grads, d)))
        assert abs(fd - an) <= tol * max(abs(fd), abs(an), 1e-3), f"finite difference {fd} vs gradient {an}"


def test_make_incompressible_gradient(emu_backend):
    """ tests/commit/physics/test_fluid.py:55-73: gradient of l2_loss(make_incompressible(v)[0]) w.r.t. the velocity. The reference
    compares its backends with each other; here the adjoint kernels are compared with finite differences of the forward path. """
    import torch
    from phiflow_amd.flow import jacobian, l2_loss, precision
    rng = np.random.default_rng(20)
    with precision(64):
        bounds = Box['x,y', 0:100, 0:100]
        for ext in (ZERO, PERIODIC, combine_sides(x=BOUNDARY, y=(ZERO, BOUNDARY))):
            shapes = StaggeredGrid(0, ext, bounds, x=16, y=16, backend=emu_backend).component_shapes
            vals = [rng.standard_normal(s) for s in shapes]
            solve = Solve('CG', 1e-12, 0)

            def sim(velocity):
                velocity, _ = fluid.make_incompressible(velocity, (), solve)
                loss = l2_loss(velocity)
                assert bool(torch.isfinite(loss).all())
                return loss

            sim_grad = jacobian(sim, get_output=False)
            grad, = sim_grad(StaggeredGrid(vals, ext, bounds, x=16, y=16, backend=emu_backend))
            assert grad.is_staggered and all(np.isfinite(g).all() for g in grad.numpy())
            loss_np = lambda vs: float(sim(StaggeredGrid(vs, ext, bounds, x=16, y=16, backend=emu_backend)))
            _fd_gradient_check(loss_np, vals, grad.numpy(), rng, tol=1e-6)


def test_make_incompressible_gradient_with_user_active(emu_backend):
    """ with a user-supplied `active` the forward pass does not balance the divergence (fluid.py:145: all_active is False)
    but carries the is_finite guard bit; the backward pass must not apply the balancing adjoint. Gradient vs finite differences. """
    import torch
    from phiflow_amd.flow import jacobian, l2_loss, precision
    rng = np.random.default_rng(21)
    with precision(64):
        bounds = Box['x,y', 0:100, 0:100]
        for ext in (ZERO, combine_sides(x=BOUNDARY, y=(ZERO, BOUNDARY))):
            n = 16
            shapes = StaggeredGrid(0, ext, bounds, x=n, y=n, backend=emu_backend).component_shapes
            vals = [rng.standard_normal(s) for s in shapes]
            act = np.ones((n, n))
            act[4:7, 5:11] = 0
            active = CenteredGrid(act, 0, bounds, x=n, y=n, backend=emu_backend)
            solve = Solve('CG', 1e-12, 0)

            def sim(velocity):
                velocity, pressure = fluid.make_incompressible(velocity, (), solve, active=active)
                return l2_loss(velocity) + l2_loss(pressure)     # (the projected velocity alone has a zero adjoint right-hand side)

            sim_grad = jacobian(sim, get_output=False)
            grad, = sim_grad(StaggeredGrid(vals, ext, bounds, x=n, y=n, backend=emu_backend))
            loss_np = lambda vs: float(sim(StaggeredGrid(vs, ext, bounds, x=n, y=n, backend=emu_backend)))
            _fd_gradient_check(loss_np, vals, grad.numpy(), rng, tol=1e-6)


def test_implicit_diffusion_gradient(emu_backend):
    """ gradient of a loss on diffuse.implicit(field) w.r.t. the field: one more solve with the symmetric operator and homogeneous wall
    values, against finite differences of the forward path (staggered velocity with a lid, centred scalar with a constant side) """
    import torch
    from phiflow_amd.flow import jacobian, l2_loss, precision
    rng = np.random.default_rng(23)
    with precision(64):
        solve = Solve('CG', 1e-12, 0)
        boundary = {'x': 0, 'y-': 0, 'y+': vec(x=1, y=0)}
        shapes = StaggeredGrid(0, boundary, x=10, y=8, backend=emu_backend).component_shapes
        vals = [rng.standard_normal(s) for s in shapes]
        sim = lambda v: l2_loss(diffuse.implicit(v, 0.7, 1.0, solve))
        grad, = jacobian(sim, get_output=False)(StaggeredGrid(vals, boundary, x=10, y=8, backend=emu_backend))
        loss_np = lambda vs: float(sim(StaggeredGrid(vs, boundary, x=10, y=8, backend=emu_backend)))
        _fd_gradient_check(loss_np, vals, grad.numpy(), rng, tol=1e-6, n_dirs=2)
        ext = combine_sides(x=(1.0, ZERO_GRADIENT), y=PERIODIC)
        t0 = rng.standard_normal((9, 8))
        sim_c = lambda t: l2_loss(diffuse.implicit(t, 0.4, 2.0, solve))
        grad_c, = jacobian(sim_c, get_output=False)(CenteredGrid(t0, ext, x=9, y=8, backend=emu_backend))
        loss_c = lambda vs: float(sim_c(CenteredGrid(vs[0], ext, x=9, y=8, backend=emu_backend)))
        _fd_gradient_check(loss_c, [t0], [grad_c.numpy()], rng, tol=1e-6, n_dirs=2)


def test_functional_gradient_through_a_fluid_step(emu_backend):
    """ tests/commit/test_colab_fluids_tutorial.py:11-34 pattern: gradient of a loss on the smoke after several steps of
    {advect smoke, buoyancy, self-advection, projection} w.r.t. the initial velocity (semi-Lagrangian smoke advection: smooth between the
    kinks of the lookup, so finite differences are a usable check; the MacCormack variant is test_colab_tutorial_functional_gradient), with an obstacle in the way. """
    from phiflow_amd.flow import functional_gradient, l2_loss, precision, resample
    rng = np.random.default_rng(21)
    with precision(64):
        bounds = Box(x=32, y=40)
        inflow = 0.6 * CenteredGrid(Sphere(x=16, y=6, radius=4), BOUNDARY, bounds, x=16, y=20, backend=emu_backend)
        obstacle = Obstacle(Box(x=(10, 18), y=(22, 26)))
        solve = Solve('CG', 1e-12, 0)

        def simulate(velocity, smoke):
            for _ in range(2):
                smoke = advect.semi_lagrangian(smoke, velocity, 1.0) + inflow
                buoyancy = resample(smoke * (0, 0.5), to=velocity)
                velocity = advect.semi_lagrangian(velocity, velocity, 1.0) + buoyancy
                velocity, pressure = fluid.make_incompressible(velocity, obstacle, solve)
            return l2_loss(smoke) + 0.1 * l2_loss(velocity) + 0.01 * l2_loss(pressure), smoke, velocity

        shapes = StaggeredGrid(0, 0, bounds, x=16, y=20, backend=emu_backend).component_shapes
        v_vals = [0.3 * rng.standard_normal(s) for s in shapes]
        s_vals = rng.random((16, 20))
        mk = lambda vs, ss: (StaggeredGrid(vs, 0, bounds, x=16, y=20, backend=emu_backend), CenteredGrid(ss, BOUNDARY, bounds, x=16, y=20, backend=emu_backend))
        sim_grad = functional_gradient(simulate, wrt=[0, 1], get_output=False)
        g_v, g_s = sim_grad(*mk(v_vals, s_vals))
        loss_np = lambda arrs: float(simulate(*mk(arrs[:2], arrs[2]))[0])
        _fd_gradient_check(loss_np, v_vals + [s_vals], g_v.numpy() + [g_s.numpy()], rng, eps=1e-6, tol=1e-4)


def test_gradient_through_sampling_between_grids(emu_backend):
    """ gradients through the general sampling path (phihip_grid_sample_backward behind torch.autograd): smoke on a finer grid than the
    velocity (Batched_Smoke.ipynb), semi-Lagrangian with the euler and the rk4 back-trace, resample of the buoyancy to the coarse faces """
    from phiflow_amd.flow import functional_gradient, l2_loss, precision, resample
    rng = np.random.default_rng(33)
    with precision(64):
        bounds = Box(x=32, y=40)

        def simulate(velocity, smoke):
            smoke = advect.semi_lagrangian(smoke, velocity, 1.3)
            smoke = advect.semi_lagrangian(smoke, velocity, 0.7, integrator=advect.rk4)
            velocity = velocity + resample(smoke * (0.2, 0.5), to=velocity)
            return l2_loss(smoke) + l2_loss(velocity), smoke

        shapes = StaggeredGrid(0, 0, bounds, x=8, y=10, backend=emu_backend).component_shapes
        v_vals = [0.8 * rng.standard_normal(s) for s in shapes]
        s_vals = rng.random((20, 24))
        mk = lambda vs, ss: (StaggeredGrid(vs, 0, bounds, x=8, y=10, backend=emu_backend), CenteredGrid(ss, BOUNDARY, bounds, x=20, y=24, backend=emu_backend))
        g_v, g_s = functional_gradient(simulate, wrt=[0, 1], get_output=False)(*mk(v_vals, s_vals))
        loss_np = lambda arrs: float(simulate(*mk(arrs[:2], arrs[2]))[0])
        _fd_gradient_check(loss_np, v_vals + [s_vals], g_v.numpy() + [g_s.numpy()], rng, eps=1e-7, tol=2e-4)


def test_colab_tutorial_functional_gradient(emu_backend, full=False):
    """ tests/commit/test_colab_fluids_tutorial.py:11-34 (batch of 4 inflow locations, MacCormack smoke, buoyancy, self-advection,
    projection, loss = l2(diffuse.explicit(smoke - stop_gradient(target)))) with `functional_gradient(simulate, wrt=[0])`; the
    gradient w.r.t. the initial velocity is checked against finite differences in fp64. """
    from phiflow_amd.flow import functional_gradient, l2_loss, precision, resample, stop_gradient
    rng = np.random.default_rng(22)
    with precision(64):
        bounds = Box(x=32, y=40)
        n = dict(x=16, y=20)
        locs = [(4., 5), (8., 5), (12., 5), (16., 5)] if full else [(4., 5), (12., 5)]     # the emulated CPU run keeps it short
        nb, steps = len(locs), (3 if full else 2)
        inflow_vals = np.stack([0.6 * CenteredGrid(Sphere(x=x, y=y, radius=3), BOUNDARY, bounds, backend=emu_backend, **n).numpy() for x, y in locs])
        inflow = CenteredGrid(inflow_vals, BOUNDARY, bounds, backend=emu_backend, **n)
        solve = Solve('CG', 1e-12, 0)

        frozen_target = []     # the finite differences must see the same constant target as the stop_gradient'ed run

        def simulate(velocity, smoke):
            for _ in range(steps):
                smoke = advect.mac_cormack(smoke, velocity, dt=1) + inflow
                buoyancy_force = smoke * (0, 0.5) @ velocity
                velocity = advect.semi_lagrangian(velocity, velocity, dt=1) + buoyancy_force
                velocity, _ = fluid.make_incompressible(velocity, (), solve)
            if not frozen_target:
                frozen_target.append(stop_gradient(smoke).values[-1:].clone())       # smoke.inflow_loc[-1], no gradient
            diff = smoke.with_values(smoke.values - frozen_target[0])
            loss = l2_loss(diffuse.explicit(diff, 1, 1, 10))
            return loss, smoke, velocity

        shapes = StaggeredGrid(0, 0, bounds, backend=emu_backend, **n).component_shapes
        v_vals = [0.2 * rng.standard_normal((nb,) + s) for s in shapes]
        smoke0 = np.zeros((nb, 16, 20))
        mk = lambda vs: (StaggeredGrid(vs, 0, bounds, backend=emu_backend, **n), CenteredGrid(smoke0, BOUNDARY, bounds, backend=emu_backend, **n))
        sim_grad = functional_gradient(simulate, wrt=[0], get_output=False)
        velocity_grad, = sim_grad(*mk(v_vals))
        assert velocity_grad.is_staggered and velocity_grad.batch_size == nb
        loss_np = lambda arrs: float(simulate(*mk(arrs))[0].sum())
        _fd_gradient_check(loss_np, v_vals, velocity_grad.numpy(), rng, eps=1e-6, tol=5e-4, n_dirs=2)
        v1 = mk(v_vals)[0] - 0.01 * velocity_grad            # the tutorial's gradient-descent update
        assert float(simulate(v1, mk(v_vals)[1])[0].sum()) < float(simulate(*mk(v_vals))[0].sum())


def test_phiml_plugin_is_inert_without_phiml():
    """ the PhiFlow plug-in glue imports cleanly and reports that it cannot install itself when phiml is absent (it is in this
    environment: the reference's arithmetic dependency is not installable -- SURVEY fact 3) """
    from phiflow_amd import phiml_plugin
    if phiml_plugin.phiml_available():
        pytest.skip("phiml is installed: the plug-in is exercised by PhiFlow's own tests instead")
    assert phiml_plugin.install() is False
    phiml_plugin.uninstall()


def test_lazy_vector_factor_survives_arithmetic(emu_backend):
    """ `smoke * (0, 0.1) * dt @ v` -- the lazy constant vector of `scalar * vector` must survive scalar arithmetic, negation
    and boundary changes, and adding something that would silently drop it must raise """
    from phiflow_amd.flow import ZERO_GRADIENT
    rng = np.random.default_rng(0)
    bounds = Box(x=8, y=8)
    s = CenteredGrid(rng.random((8, 8)).astype(np.float32) + 0.5, ZERO_GRADIENT, bounds, x=8, y=8, backend=emu_backend)
    v = StaggeredGrid(0, 0, bounds, x=8, y=8, backend=emu_backend)
    base = (s * (0, 0.5)) @ v
    assert float(np.abs(base.numpy()[0]).max()) == 0 and float(np.abs(base.numpy()[1]).max()) > 0.1
    for expr, factor in ((s * (0, 1.0) * 0.5, 1.0), (0.5 * (s * (0, 1.0)), 1.0), (-(s * (0, 0.5)), -1.0), ((s * (0, 1.0)) / 2.0, 1.0),
                         ((s * (0, 0.25)) + (s * (0, 0.25)), 1.0), ((s * (0, 1.0)).with_boundary(ZERO_GRADIENT) * 0.5, 1.0),
                         ((s * (1, 1)) * (0, 0.5), 1.0)):
        out = expr @ v
        np.testing.assert_array_equal(out.numpy()[0], 0 * base.numpy()[0])
        np.testing.assert_allclose(out.numpy()[1], factor * base.numpy()[1], rtol=1e-6)
    with pytest.raises(NotImplementedError):
        (s * (0, 0.5)) + s
    with pytest.raises(NotImplementedError):
        (s * (0, 0.5)) + 1.0
    with pytest.raises(NotImplementedError):
        (s * (0, 0.5)) @ s


def test_resample_compares_bounds_not_only_resolution(emu_backend):
    """ fields with equal resolution on different boxes do NOT share sample points: `@` must interpolate, arithmetic refuse """
    from phiflow_amd.flow import ZERO_GRADIENT
    a = CenteredGrid(lambda x, y: x + 0 * y, ZERO_GRADIENT, Box(x=8, y=8), x=8, y=8, backend=emu_backend)
    b = CenteredGrid(0, ZERO_GRADIENT, Box(x=4, y=4), x=8, y=8, backend=emu_backend)
    out = a @ b
    assert tuple(out.bounds.upper) == (4.0, 4.0)
    expect = (np.arange(8) + 0.5) * 0.5                     # f(x) = x sampled at b's cell centres (linear interpolation is exact)
    np.testing.assert_allclose(out.numpy()[:, 0], np.maximum(expect, 0.5), atol=1e-5)      # left of a's first centre: zero-gradient
    with pytest.raises(AssertionError):
        a + b
    v = StaggeredGrid(0, 0, Box(x=4, y=4), x=8, y=8, backend=emu_backend)
    to_faces = (a * (1.0, 0.0)) @ v                         # general path: different boxes
    assert tuple(to_faces.bounds.upper) == (4.0, 4.0)
    np.testing.assert_allclose(to_faces.numpy()[0][:, 0], np.arange(1, 8) * 0.5, atol=1e-5)


def test_vector_scaled_scalar_is_refused_where_it_would_be_taken_for_the_scalar(emu_backend, tmp_path):
    """ `smoke * (0, 0.1)` stays a lazily scaled scalar until it is resampled to faces (Smoke_Plume.ipynb cell 5); every consumer that
    would read its `values` as the field itself has to refuse instead of computing with the plain scalar """
    from phiflow_amd import field_io
    from phiflow_amd.autodiff import l2_loss
    from phiflow_amd.field import mean, spatial_gradient
    from phiflow_amd.flow import Box, CenteredGrid, StaggeredGrid, ZERO_GRADIENT, advect, diffuse, resample
    bounds = Box(x=8, y=6)
    smoke = CenteredGrid(np.random.default_rng(0).random((8, 6)).astype(np.float32), ZERO_GRADIENT, bounds, x=8, y=6, backend=emu_backend)
    v = StaggeredGrid(0.1, 0, bounds, x=8, y=6, backend=emu_backend)
    buoyancy = smoke * (0, 0.1)
    assert buoyancy.vector_scale == [0.0, 0.1]
    for call in (lambda: mean(buoyancy), lambda: spatial_gradient(buoyancy, v.boundary), lambda: advect.semi_lagrangian(buoyancy, v, 1.0),
                 lambda: advect.mac_cormack(buoyancy, v, 1.0), lambda: diffuse.explicit(buoyancy, 0.1, 1.0), lambda: l2_loss(buoyancy),
                 lambda: field_io.write(buoyancy, str(tmp_path / "b.npz"))):
        with pytest.raises(NotImplementedError, match="constant vector"):
            call()
    on_faces = resample(buoyancy, to=v)                      # the supported use
    assert on_faces.is_staggered and float(on_faces.values[0].abs().max()) == 0.0 and float(on_faces.values[1].max()) > 0
    scaled = (2.0 * buoyancy) @ v                            # arithmetic carries the vector along
    assert float(scaled.values[1].max()) == pytest.approx(2.0 * float(on_faces.values[1].max()), rel=1e-6)
