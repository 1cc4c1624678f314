"""
`jit_compile` / `iterate` (phiflow_amd/jit.py; reference usage: examples/grids/Smoke_Plume.ipynb cell 5 `@jit_compile def step(...)`,
tests/commit/physics/test_higher_order.py:55-56 `math.jit_compile(fourth_ord_runge_kutta)`, SURVEY 3.1 `iterate(step, batch(time=N), ...)`).

CPU part (emulation device: nothing can be captured -- the wrapper runs the function eagerly under the rules of a captured one): argument
trees, signature keys, the no-read-back form of the solves. GPU part (`-m gpu`): capture + replay must leave the bits of the eager steps.
"""
import numpy as np
import pytest
import torch

from phiflow_amd import jit as J
from phiflow_amd.flow import (ZERO_GRADIENT, PERIODIC, Box, CenteredGrid, NotConverged, Obstacle, Solve, Sphere, StaggeredGrid, advect, diffuse,
                              fluid, iterate, jit_compile, resample)


def _plume(be, n=32):
    dom = Box(x=100, y=100)
    inflow = 0.2 * resample(Sphere(x=50, y=9.5, radius=5), to=CenteredGrid(0, ZERO_GRADIENT, dom, x=n, y=n, backend=be), soft=True)
    v0 = StaggeredGrid(0, 0, dom, x=n, y=n, backend=be)
    s0 = CenteredGrid(0, ZERO_GRADIENT, dom, x=n, y=n, backend=be)

    def step(v, s, p, dt=1.0, iters=30):
        s = advect.mac_cormack(s, v, dt) + inflow
        v = advect.semi_lagrangian(v, v, dt) + resample(s * (0, 0.1), to=v)
        v, p = fluid.make_incompressible(v, (), Solve('CG', 0, 0, x0=p, max_iterations=iters, suppress=[NotConverged]))
        return v, s, p
    return step, v0, s0


def _np(field):
    out = field.numpy()
    return out if isinstance(out, list) else [out]


def _same(a, b):
    return all(np.array_equal(x, y) for fa, fb in zip(a, b) for x, y in zip(_np(fa), _np(fb)))


def _diff(a, b):
    """ which arrays of two states differ, and by how much (assertion messages) """
    out = []
    for k, (fa, fb) in enumerate(zip(a, b)):
        for c, (x, y) in enumerate(zip(_np(fa), _np(fb))):
            if not np.array_equal(x, y):
                d = np.abs(x.astype(np.float64) - y.astype(np.float64))
                out.append(f"field {k} component {c}: {int((d > 0).sum())} of {d.size} differ, max {d.max():.3e} at {np.unravel_index(int(d.argmax()), d.shape)}")
    return "; ".join(out) or "equal"


def test_argument_trees_round_trip(emu_backend):
    v = StaggeredGrid(1.5, PERIODIC, x=8, y=6, backend=emu_backend)
    s = CenteredGrid(2.0, ZERO_GRADIENT, x=8, y=6, backend=emu_backend)
    t = torch.arange(4.0)
    tensors = []
    tree = ((v, [s, t], {"k": 3, "w": (s, None)}), {"dt": 0.5, "solve": Solve('CG', 1e-3)})
    spec = J._flatten(tree, tensors)
    assert len(tensors) == 2 + 1 + 1 + 1                    # two components, the scalar, the tensor, the scalar again
    back = J._unflatten(spec, iter(tensors))
    assert back[0][0].is_staggered and back[0][0].boundary == v.boundary and back[0][0].values[1] is v.values[1]
    assert back[0][1][1] is t and back[0][2]["k"] == 3 and back[0][2]["w"][1] is None and back[1]["dt"] == 0.5
    # the key separates what a capture depends on: auxiliary values, boundaries, resolutions -- not the tensors' contents
    k1 = J._spec_key(spec)
    tensors2 = []
    tree2 = ((v * 2.0, [s, t + 1], {"k": 3, "w": (s, None)}), {"dt": 0.5, "solve": Solve('CG', 1e-3)})
    assert J._spec_key(J._flatten(tree2, tensors2)) == k1
    for changed in (((v, [s, t], {"k": 4, "w": (s, None)}), {"dt": 0.5, "solve": Solve('CG', 1e-3)}),
                    ((v, [s, t], {"k": 3, "w": (s, None)}), {"dt": 0.25, "solve": Solve('CG', 1e-3)}),
                    ((v, [s, t], {"k": 3, "w": (s, None)}), {"dt": 0.5, "solve": Solve('CG', 1e-4)}),
                    ((v.with_extrapolation(0), [s, t], {"k": 3, "w": (s, None)}), {"dt": 0.5, "solve": Solve('CG', 1e-3)})):
        assert J._spec_key(J._flatten(changed, [])) != k1


def test_solve_arguments_are_traced_and_tensor_holding_auxiliaries_refused(emu_backend):
    """ r6: a Solve passed as an ARGUMENT -- PhiML traces Solve.x0 like any tensor: the guess becomes a graph input (one capture for every guess,
    fresh values copied in), list-valued fields need no hash; an unhashable auxiliary object that holds a tensor is refused, never keyed on its repr """
    p_a = CenteredGrid(1.0, ZERO_GRADIENT, x=8, y=6, backend=emu_backend)
    p_b = CenteredGrid(2.0, ZERO_GRADIENT, x=8, y=6, backend=emu_backend)
    v = StaggeredGrid(1.5, PERIODIC, x=8, y=6, backend=emu_backend)
    solve_a = Solve('CG', 1e-3, x0=p_a, suppress=[NotConverged])
    solve_b = Solve('CG', 1e-3, x0=p_b, suppress=[NotConverged])
    with pytest.raises(TypeError):
        hash(solve_a)                                  # (the list-valued field: this object went through repr() until r5)
    ta, tb = [], []
    spec_a, spec_b = J._flatten(((v, solve_a), {}), ta), J._flatten(((v, solve_b), {}), tb)
    assert len(ta) == len(tb) == 3 and ta[2] is p_a.values and tb[2] is p_b.values          # x0 is among the graph inputs
    assert J._spec_key(spec_a) == J._spec_key(spec_b)                                         # one capture serves both guesses
    assert J._spec_key(J._flatten(((v, Solve('CG', 1e-4, x0=p_a, suppress=[NotConverged])), {}), [])) != J._spec_key(spec_a)
    assert J._spec_key(J._flatten(((v, Solve('CG', 1e-3, x0=None, suppress=[NotConverged])), {}), [])) != J._spec_key(spec_a)
    back = J._unflatten(spec_b, iter(tb))[0][1]
    assert isinstance(back, Solve) and back.x0.values is p_b.values and back.rel_tol == 1e-3 and back.suppress == [NotConverged]

    class Bag:                      # unhashable, holds a Field, its repr shows no values
        __hash__ = None

        def __init__(self, f):
            self.guess = {"p": [f]}

        def __repr__(self):
            return "Bag()"
    with pytest.raises(TypeError, match="holds a tensor"):
        J._spec_key(J._flatten(((v, Bag(p_a)), {}), []))

    class Plain:
        __hash__ = None

        def __repr__(self):
            return "Plain(3)"
    assert J._spec_key(J._flatten(((v, Plain()), {}), [])) == J._spec_key(J._flatten(((v, Plain()), {}), []))

    # the function runs with the guess of THIS call
    rng = np.random.default_rng(5)
    w = StaggeredGrid([rng.standard_normal((16, 16)).astype(np.float32) for _ in range(2)], PERIODIC, x=16, y=16, backend=emu_backend)

    def project(u, solve):
        return fluid.make_incompressible(u, (), solve)
    jproject = jit_compile(project)
    _, p1 = project(w, Solve('CG', 0, 0, max_iterations=3, suppress=[NotConverged]))
    for guess in (p1, p1 * 0.5):
        s = Solve('CG', 0, 0, x0=guess, max_iterations=4, suppress=[NotConverged])
        assert _same(project(w, s), jproject(w, s))


def test_jit_function_on_the_emulation_device_matches_eager(emu_backend):
    """ no capture without a HIP device: the wrapper runs the function under the rules of a captured one (info = NULL, check_every = 0,
    no exceptions from the solve) -- same arithmetic, so the same bits as the eager step """
    step, v0, s0 = _plume(emu_backend)
    jstep = jit_compile(step)
    assert jstep.__name__ == "step"
    seen = []

    def probe(v, s, p):
        out = step(v, s, p)
        seen.append((J.is_tracing(), out[2].solve_info))
        return out
    state_e, state_j = (v0, s0, None), (v0, s0, None)
    for _ in range(3):
        state_e = step(*state_e)
        state_j = jstep(*state_j)
    assert _same(state_e, state_j)
    assert state_e[2].solve_info.iterations == [30] and state_j[2].solve_info is None
    jit_compile(probe)(v0, s0, None)
    probe(v0, s0, None)
    assert seen[0][0] is True and seen[0][1] is None and seen[1][0] is False and seen[1][1] is not None
    assert not J.is_tracing()
    # iterate: N steps, the final state
    v_i, s_i, p_i = iterate(jstep, 3, v0, s0, None)
    assert _same((v_i, s_i, p_i), state_e)
    (v_t, s_t, p_t), times = iterate(jstep, 2, v0, s0, None, f_kwargs=dict(dt=1.0), measure=__import__("time").perf_counter)
    assert len(times) == 2 and all(t > 0 for t in times)
    with pytest.raises(NotImplementedError):
        iterate(jstep, (3,), v0, s0, None)
    # the flag is per thread: a function captured on one thread leaves the solves of another thread alone
    import threading
    other = []
    t = threading.Thread(target=lambda: other.append(J.is_tracing()))
    with J._tracing():
        t.start(); t.join()
        assert J.is_tracing()
    assert other == [False]


def test_what_a_captured_function_may_not_do(emu_backend):
    v = StaggeredGrid(1.0, PERIODIC, x=8, y=8, backend=emu_backend)

    @jit_compile
    def bad(v):
        return diffuse.implicit(v, 0.1, 1.0, order=4)
    with pytest.raises(NotImplementedError, match="order=2"):
        bad(v)
    assert not J.is_tracing()              # the flag is restored when the function raises

    # diffuse.implicit inside a captured function: the same solve without the read-back (phihip_diffuse_implicit with info = NULL)
    rng0 = np.random.default_rng(3)
    w = StaggeredGrid([rng0.standard_normal((8, 8)).astype(np.float32) for _ in range(2)], PERIODIC, x=8, y=8, backend=emu_backend)
    c = CenteredGrid(rng0.standard_normal((8, 8)).astype(np.float32), ZERO_GRADIENT, x=8, y=8, backend=emu_backend)
    for f in (w, c):
        eager = diffuse.implicit(f, 0.1, 1.0, Solve('CG', 1e-6, max_iterations=200))
        traced = jit_compile(lambda u: diffuse.implicit(u, 0.1, 1.0, Solve('CG', 1e-6, max_iterations=200)))(f)
        assert _same((eager,), (traced,)) and eager.solve_info is not None and traced.solve_info is None

    # no autograd graph behind a replay: inputs that require gradients are refused instead of silently detached
    vg = StaggeredGrid([torch.ones(1, 8, 8, requires_grad=True), torch.ones(1, 8, 8)], PERIODIC, x=8, y=8, backend=emu_backend)
    with pytest.raises(NotImplementedError, match="gradients"):
        jit_compile(lambda u: advect.semi_lagrangian(u, u, 0.1))(vg)
    with torch.no_grad():
        jit_compile(lambda u: advect.semi_lagrangian(u, u, 0.1))(vg)

    # a tolerance solve that does not converge raises outside, stays silent inside (the host is not told)
    rng = np.random.default_rng(1)
    rough = StaggeredGrid([rng.standard_normal((16, 16)).astype(np.float32) for _ in range(2)], PERIODIC, x=16, y=16, backend=emu_backend)
    with pytest.raises(NotConverged):
        fluid.make_incompressible(rough, (), Solve('CG', 1e-7, 0, max_iterations=2))
    v2, p2 = jit_compile(lambda u: fluid.make_incompressible(u, (), Solve('CG', 1e-7, 0, max_iterations=2)))(rough)
    assert p2.solve_info is None and np.isfinite(p2.numpy()).all()


# ---- GPU: capture and replay ------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("n,iters", [(32, 30), (128, 50), (192, 20)])       # one-kernel solve (<= 16384 cells), the benchmark's plume, the marching CG
def test_captured_plume_step_replays_the_eager_bits(gpu_backend, n, iters):
    """ default settings since r6 (until r5 the reach of the LDS-staged advection passes had to be pinned: a replayed graph keeps the reach it was captured with
    while the eager loop adapts it to the plume's CFL, and window and gather kernels agreed to rounding only) """
    _captured_plume(gpu_backend, n, iters)


def _captured_plume(gpu_backend, n, iters):
    step, v0, s0 = _plume(gpu_backend, n)
    jstep = jit_compile(step)
    state_e, state_j = (v0, s0, None), (v0, s0, None)
    held = []
    for k in range(6):
        prev_e, prev_j = state_e, state_j
        state_e = step(*state_e, iters=iters)
        state_j = jstep(*state_j, iters=iters)
        held.append((state_j, [a.copy() for f in state_j for a in _np(f)]))
        if not _same(state_e, state_j):      # which side moved? both are repeated from the same inputs
            again_e, again_j = step(*prev_e, iters=iters), jstep(*prev_j, iters=iters)
            raise AssertionError(f"step {k}: eager vs replay {_diff(state_e, state_j)} || eager repeated vs eager {_diff(again_e, state_e)} || replay repeated vs replay "
                                 f"{_diff(again_j, state_j)} || fallback {gpu_backend.ctx.advect_fallback_stats()}")
    # signatures: (v, s, None) and (v, s, p) -- two captures, four replays of the second
    assert jstep.traces == 2 and jstep.replays == 6
    # results are clones: what step k returned is still what it was after later replays
    for fields, copies in held:
        assert all(np.array_equal(a, b) for a, b in zip([a for f in fields for a in _np(f)], copies))
    assert state_j[2].solve_info is None and state_e[2].solve_info.iterations == [iters]
    # an auxiliary value changes: a new capture, the same bits as eager again
    e2 = step(*state_e, dt=0.5, iters=iters)
    j2 = jstep(*state_j, dt=0.5, iters=iters)
    assert jstep.traces == 3 and _same(e2, j2)


@pytest.mark.gpu
def test_captured_plume_with_the_adaptive_reach_gives_the_eager_bits(gpu_backend):
    """ default settings, 40 steps of the 128^2 plume (its CFL passes 1 on the way): the eager loop changes the reach of its advection passes, the captured
    step keeps its own -- r6: every path evaluates one arithmetic per sample, so the two runs agree BIT FOR BIT (r5: to 2e-3). (No fused multi-tensor launch
    between the replays here: that is unsafe on this ROCm build whatever the graph holds -- profiles/r06_jit_flaky_probe.txt.) """
    step, v0, s0 = _plume(gpu_backend, 128)
    se = iterate(step, 40, v0, s0, None, f_kwargs=dict(iters=50))
    sj = iterate(jit_compile(step), 40, v0, s0, None, f_kwargs=dict(iters=50))
    assert _same(se, sj), _diff(se, sj)


@pytest.mark.gpu
def test_captured_3d_step_with_obstacle_and_iterate(gpu_backend):
    n = 40
    dom = Box(x=1, y=1, z=1)
    rng = np.random.default_rng(5)
    v0 = StaggeredGrid([rng.standard_normal(s).astype(np.float32) * 0.02 for s in ((n - 1, n, n), (n, n - 1, n), (n, n, n - 1))], 0, dom, x=n, y=n, z=n,
                       backend=gpu_backend)
    ball = Obstacle(Sphere(x=0.5, y=0.5, z=0.4, radius=0.15))

    def step(v, p, dt):
        v = advect.semi_lagrangian(v, v, dt)
        v = diffuse.explicit(v, 1e-3, dt)
        return fluid.make_incompressible(v, [ball], Solve('CG', 0, 0, x0=p, max_iterations=25, suppress=[NotConverged]))
    ve, pe = iterate(step, 4, v0, None, f_kwargs=dict(dt=0.2))
    jstep = jit_compile(step, forget_traces=True, copy_outputs=False)
    vj, pj = iterate(jstep, 4, v0, None, f_kwargs=dict(dt=0.2))
    assert _same((ve, pe), (vj, pj))
    assert len(jstep.captures) == 1 and jstep.traces == 3          # forget_traces: only the latest signature is kept -- with (copy_outputs=False) its ping-pong partner


@pytest.mark.gpu
def test_captured_viscous_step_with_implicit_diffusion(gpu_backend):
    """ Heat_Flow / Burgers-type step: advection + diffuse.implicit (a CG per component with the operator I - k dt L) + projection, captured and replayed """
    n = 64
    rng = np.random.default_rng(8)
    v0 = StaggeredGrid([rng.standard_normal((n, n)).astype(np.float32) * 0.05 for _ in range(2)], PERIODIC, x=n, y=n, backend=gpu_backend)

    def step(v, p, dt=0.5):
        v = advect.semi_lagrangian(v, v, dt)
        v = diffuse.implicit(v, 0.05, dt, Solve('CG', 0, 0, max_iterations=12, suppress=[NotConverged]))
        return fluid.make_incompressible(v, (), Solve('CG', 0, 0, x0=p, max_iterations=20, suppress=[NotConverged]))
    jstep = jit_compile(step)
    se, sj = (v0, None), (v0, None)
    for k in range(5):
        se, sj = step(*se), jstep(*sj)
        assert _same(se, sj), f"step {k}"
    assert jstep.traces == 2 and jstep.replays == 5


@pytest.mark.gpu
def test_solve_argument_with_a_new_guess_replays_one_capture(gpu_backend):
    """ r6: `step(v, solve)` with Solve(x0=p): the guess is a graph input -- ONE capture, every call's own guess, the eager bits """
    rng = np.random.default_rng(8)
    w = StaggeredGrid([rng.standard_normal((64, 64)).astype(np.float32) for _ in range(2)], PERIODIC, x=64, y=64, backend=gpu_backend)

    def project(u, solve):
        return fluid.make_incompressible(u, (), solve)
    jproject = jit_compile(project)
    _, p = project(w, Solve('CG', 0, 0, max_iterations=5, suppress=[NotConverged]))
    for k in range(4):
        s = Solve('CG', 0, 0, x0=p * (1.0 - 0.1 * k), max_iterations=7, suppress=[NotConverged])
        assert _same(project(w, s), jproject(w, s)), f"call {k}"
    assert jproject.traces == 1 and jproject.replays == 4          # (the capturing call replays its graph once, too)


@pytest.mark.gpu
def test_ping_pong_of_two_captures_without_output_copies(gpu_backend):
    """ r6: `copy_outputs=False` + the loop `state = step(*state)`: the second capture reads the first one's output buffers in place and the two alternate --
    the eager bits, no output clones, input copies every other step only """
    step, v0, s0 = _plume(gpu_backend, 64)
    n = 9
    se = iterate(step, n, v0, s0, None, f_kwargs=dict(iters=20))
    jstep = jit_compile(step, copy_outputs=False)
    state = (v0, s0, None)
    for _ in range(n):
        state = jstep(*state, iters=20)
    assert _same(se, state)
    # signatures: (v, s, None) -- one capture; (v, s, p) -- the capture and its ping-pong partner, whose inputs ARE the first one's output buffers
    assert jstep.traces == 3 and jstep.replays == n
    pair = [v for v in jstep.captures.values() if len(v) == 2]
    assert len(pair) == 1 and set(pair[0][1].input_ptrs) <= pair[0][0].output_ptrs
    # calls 2 .. n run on the (v, s, p) signature: call 2 captures (copying), call 3 captures the partner (in place), then the partner's results go back into
    # the first capture (4 tensors copied: two velocity components, smoke, pressure) every OTHER call
    assert jstep.input_copies == 4 * ((n - 3) // 2), jstep.input_copies
    # the default (copy_outputs=True) is untouched: results are clones, every replay copies its inputs in
    jdef = jit_compile(step)
    sd = iterate(jdef, n, v0, s0, None, f_kwargs=dict(iters=20))
    assert _same(se, sd) and jdef.traces == 2 and all(len(v) == 1 for v in jdef.captures.values())


@pytest.mark.gpu
def test_auxiliary_args_keep_a_field_out_of_the_graph_inputs(gpu_backend):
    n = 48
    rng = np.random.default_rng(2)
    v0 = StaggeredGrid([rng.standard_normal((n, n)).astype(np.float32) * 0.1 for _ in range(2)], PERIODIC, x=n, y=n, backend=gpu_backend)
    force = CenteredGrid(np.random.default_rng(3).standard_normal((n, n)).astype(np.float32), PERIODIC, x=n, y=n, backend=gpu_backend)

    def step(v, force, dt=0.1):
        v = advect.semi_lagrangian(v, v, dt) + resample(force * (0, 0.1), to=v)
        return fluid.make_incompressible(v, (), Solve('CG', 0, 0, max_iterations=20, suppress=[NotConverged]))[0]
    j_in = jit_compile(step)
    j_aux = jit_compile(step, auxiliary_args='force')
    e = step(v0, force)
    assert _same((e,), (j_in(v0, force),)) and _same((e,), (j_aux(v0, force),))
    # as an input the field's NEW values reach the replay; as an auxiliary argument a new object is a new capture
    force2 = force * 2.0
    e2 = step(v0, force2)
    assert _same((e2,), (j_in(v0, force2),)) and j_in.traces == 1
    assert _same((e2,), (j_aux(v0, force2),)) and j_aux.traces == 2
    # a function called with ever new auxiliary values keeps a bounded number of captures (each owns its graph's memory pool)
    j_small = jit_compile(step)
    j_small.MAX_CAPTURES = 2
    for dt in (0.1, 0.2, 0.3, 0.1):
        assert _same((step(v0, force, dt=dt),), (j_small(v0, force, dt=dt),))
    assert len(j_small.captures) == 2 and j_small.traces == 4
