"""
`jit_compile` and `iterate` -- the two PhiML functions a PhiFlow user wraps around the hot path: the step of every fluid example is
`@jit_compile def step(v, p, dt): v = advect.semi_lagrangian(v, v, dt); ...; v, p = fluid.make_incompressible(v, obstacles, Solve(x0=p))`
driven by `iterate(step, N, v0, p0, dt=...)` (reference: examples/grids/Smoke_Plume.ipynb cell 5, docs/Fluid_Simulation.ipynb cell 3,
tests/commit/test_colab_fluids_tutorial.py:43, tests/commit/physics/test_higher_order.py:55-56).

PhiML's `jit_compile` TRACES the function with its backend's compiler. Here there is no tracing compiler to hand the function to, and none
is wanted: every operator of this package is already one or a few HIP kernels behind the C ABI. What a small grid pays for is the host --
Python objects, ctypes marshalling, ~20 kernel launches of a few microseconds each (bench.py `phi_level`: +50 % at 128^2). So `jit_compile`
here CAPTURES the function's launches into a hipGraph once (torch.cuda.CUDAGraph on a side stream) and REPLAYS the graph on every later call
with the same signature:
  * tensors of the arguments (the `values` of Fields, bare tensors; nested in tuples / lists / dicts / dataclass instances -- r6: a `Solve` argument's `x0`
    is traced like PhiML traces it) are the graph's inputs: they are copied into the capture's input buffers before a replay (skipped where the caller passes
    the very buffer back);
  * everything else (numbers, strings, obstacles, boundaries, resolutions ...) is AUXILIARY like PhiML's non-tensor arguments: part of the signature, a new
    value means a new capture (`forget_traces=True` keeps only the latest, otherwise the 16 most recent); an unhashable auxiliary object that holds a tensor
    is refused (its repr would not tell two tensors apart);
  * the results are cloned out of the graph's output buffers (Fields are immutable: a result must survive the next replay); `copy_outputs=False`
    hands out the buffers themselves for callers that consume a result before the next call. Results that are not tensors (numbers, None, strings) are those of
    the capture run: a replay cannot recompute them. With `copy_outputs=False` a loop `v, p = step(v, p)` runs as a PING-PONG of two captures (r6): the
    second reads the first one's output buffers in place, so a step copies nothing out and (every other step) nothing in; a result then stays valid until
    the call after next.
Inside a captured function the host cannot see a solve's outcome: `make_incompressible` / `solve_linear` / `diffuse.implicit` run with `info = NULL` and
`check_every = 0` (the library's capture-safe form: no host read-back, no synchronisation, no allocation, no first-call autotune --
tests/test_gpu_graph.py), `pressure.solve_info` is None and NotConverged / Diverged are not raised. A tolerance solve under capture
enqueues its whole launch budget (entries that converged early freeze on the device), so give it a `max_iterations` that fits the
problem; grids of <= 16384 cells run the whole solve in ONE kernel with the convergence test on the device and need no such care.

Reproducibility: a replay gives the bits of the eager function (tests/test_jit.py), whatever reach the eager passes adapt to: every advection path computes the same
bits since r6 (tests/parity_cases.py check_advect_paths_same_bits). One caveat stays, measured and not explained: a fused `torch._foreach_*` launch directly in front
of a replay is not safe on this ROCm build (rounding-level differences in r5, a GPU memory fault in r6 on a context that had served other grids:
tools/micro/jit_flaky_probe.py); this wrapper copies its inputs tensor by tensor and launches none.

The function must be a pure function of its arguments (the capture runs it twice -- a warm-up that sizes the workspaces and tunes the
launch plans, then the capture itself -- and never again). On the CPU emulation device (tests) nothing can be captured: the wrapper then
runs the function eagerly under the same no-read-back rules, which exercises the signature / cache bookkeeping only.
"""
import dataclasses
import functools
import inspect
import threading
from contextlib import contextmanager
from typing import Any, Callable, Dict, List, Optional

import torch

from .field import Field

import os
_FUSED_COPY = os.environ.get("PHIHIP_JIT_FUSED_COPY", "0") == "1"      # see JitFunction.__call__: NOT safe in front of a replay on this ROCm build (tools/micro/jit_flaky_probe.py)
_STATE = threading.local()         # per thread: a capture on one thread must not switch another thread's solves to the no-read-back form


def is_tracing() -> bool:
    """ True while a `jit_compile`d function body runs on this thread (warm-up, capture, or the eager form on the emulation device) """
    return getattr(_STATE, "depth", 0) > 0


@contextmanager
def _tracing():
    _STATE.depth = getattr(_STATE, "depth", 0) + 1
    try:
        yield
    finally:
        _STATE.depth -= 1


# ---- argument trees: tensors out, tensors back in ---------------------------------------------------------------------------------------

class _FieldSpec:
    __slots__ = ("resolution", "bounds", "boundary", "staggered", "backend", "batched", "vector_scale", "count")

    def __init__(self, f: Field):
        self.resolution, self.bounds, self.boundary = dict(f.resolution), f.bounds, f.boundary
        self.staggered, self.backend, self.batched, self.vector_scale = f.is_staggered, f.backend, f.batched, f.vector_scale
        self.count = len(f.values) if f.is_staggered else 1

    def key(self):
        return ("Field", tuple(self.resolution.items()), repr(self.bounds), repr(self.boundary), self.staggered, id(self.backend), self.batched,
                tuple(self.vector_scale) if self.vector_scale is not None else None)

    def build(self, tensors: List[torch.Tensor]) -> Field:
        values = list(tensors) if self.staggered else tensors[0]
        return Field(self.resolution, self.bounds, self.boundary, values, self.staggered, self.backend, self.batched, self.vector_scale)


class _Aux:
    """ wrapper of an argument named in `auxiliary_args`: never descended into """
    __slots__ = ("value",)

    def __init__(self, value):
        self.value = value


def _flatten(obj, tensors: List[torch.Tensor]):
    """ -> spec; appends the tensors of `obj` to `tensors` """
    if isinstance(obj, _Aux):
        return ("A", obj.value)
    if isinstance(obj, torch.Tensor):
        tensors.append(obj)
        return ("T",)
    if isinstance(obj, Field):
        spec = _FieldSpec(obj)
        tensors.extend(obj.values if obj.is_staggered else [obj.values])
        return ("F", spec)
    if isinstance(obj, (tuple, list)):
        return ("L" if isinstance(obj, list) else "U", tuple(_flatten(o, tensors) for o in obj))
    if isinstance(obj, dict):
        return ("D", tuple((k, _flatten(v, tensors)) for k, v in obj.items()))
    if dataclasses.is_dataclass(obj) and not isinstance(obj, type):
        # r6: a Solve travels as an argument in the reference's examples (`step(v, p, solve)`), and PhiML traces `Solve.x0` like any other tensor --
        # descend into dataclass instances so that x0 becomes a graph INPUT (a new guess is copied in, not baked into / keyed on) and list-valued
        # fields (`suppress=[NotConverged]`) need no hash
        return ("C", type(obj), tuple((f.name, _flatten(getattr(obj, f.name), tensors)) for f in dataclasses.fields(obj) if f.init))
    return ("A", obj)           # auxiliary: by value


def _unflatten(spec, it):
    kind = spec[0]
    if kind == "T":
        return next(it)
    if kind == "F":
        return spec[1].build([next(it) for _ in range(spec[1].count)])
    if kind in ("L", "U"):
        items = [_unflatten(s, it) for s in spec[1]]
        return items if kind == "L" else tuple(items)
    if kind == "D":
        return {k: _unflatten(s, it) for k, s in spec[1]}
    if kind == "C":
        return spec[1](**{k: _unflatten(s, it) for k, s in spec[2]})
    return spec[1]


def _aux_key(value):
    """ hashable stand-in of an auxiliary argument: its value where it is hashable, else its repr (Solve, Obstacle, Box, extrapolations print
    their defining values) """
    if isinstance(value, (torch.Tensor, Field)) or type(value).__module__ == "numpy" and hasattr(value, "shape") and getattr(value, "ndim", 0) > 0:
        return ("object", id(value))        # by identity (the capture keeps the object alive, so the id stays its own)
    try:
        hash(value)
        return value
    except TypeError:
        # repr() of an object that HOLDS a tensor prints no values: two calls with different tensors would share one capture and the replay would reuse the
        # first call's buffer silently -- refuse instead of guessing
        if _holds_tensor(value):
            raise TypeError(f"jit_compile: the auxiliary argument {type(value).__name__} is unhashable and holds a tensor or Field; pass tensors as arguments "
                            f"of their own, inside tuples / lists / dicts / dataclasses (these are traced), or make the object hashable") from None
        return repr(value)


def _holds_tensor(value, depth: int = 0) -> bool:
    if isinstance(value, (torch.Tensor, Field)):
        return True
    if depth > 4 or isinstance(value, (str, bytes, int, float, complex, bool, type(None), type)):
        return False
    if isinstance(value, dict):
        return any(_holds_tensor(v, depth + 1) for v in value.values())
    if isinstance(value, (tuple, list, set, frozenset)):
        return any(_holds_tensor(v, depth + 1) for v in value)
    members = list(getattr(value, "__dict__", {}).values())
    for cls in type(value).__mro__:
        members += [getattr(value, name) for name in getattr(cls, "__slots__", ()) if isinstance(name, str) and hasattr(value, name)]
    return any(_holds_tensor(v, depth + 1) for v in members)


def _spec_key(spec):
    kind = spec[0]
    if kind == "T":
        return "T"
    if kind == "F":
        return spec[1].key()
    if kind in ("L", "U"):
        return (kind,) + tuple(_spec_key(s) for s in spec[1])
    if kind == "D":
        return ("D",) + tuple((k, _spec_key(s)) for k, s in spec[1])
    if kind == "C":
        return ("C", spec[1].__module__, spec[1].__qualname__) + tuple((k, _spec_key(s)) for k, s in spec[2])
    return ("A", type(spec[1]).__name__, _aux_key(spec[1]))


class _Capture:
    __slots__ = ("graph", "inputs", "outputs", "out_spec", "spec", "input_ptrs", "output_ptrs")


class JitFunction:
    """ the callable `jit_compile` returns """

    MAX_CAPTURES = 16       # a capture owns its graph's memory pool: a function called with ever new auxiliary values (an adaptive dt) must not grow without bound

    def __init__(self, f: Callable, auxiliary_args: str = "", forget_traces: Optional[bool] = None, copy_outputs: bool = True):
        self.f = f
        self.auxiliary_args = tuple(a.strip() for a in auxiliary_args.split(",") if a.strip())
        self.forget_traces = bool(forget_traces)
        self.copy_outputs = copy_outputs
        self.captures: Dict[Any, List[_Capture]] = {}      # per signature: the capture and, with copy_outputs=False, its ping-pong partner
        self.traces = 0           # captures made so far (PhiML: the number of times the function was traced)
        self.replays = 0
        self.input_copies = 0     # tensors copied into a capture's input buffers so far (a ping-pong pair copies every other step only)
        try:
            self._sig = inspect.signature(f)
        except (TypeError, ValueError):
            self._sig = None
        functools.update_wrapper(self, f)

    def _mark_auxiliary(self, args, kwargs):
        """ auxiliary_args name parameters whose tensors must NOT become graph inputs (PhiML: "not traced"): their value is part of the signature """
        if not self.auxiliary_args or self._sig is None:
            return args, kwargs
        bound = self._sig.bind(*args, **kwargs)
        for name in self.auxiliary_args:
            if name in bound.arguments:
                bound.arguments[name] = _Aux(bound.arguments[name])
        return bound.args, bound.kwargs

    def __call__(self, *args, **kwargs):
        args, kwargs = self._mark_auxiliary(args, kwargs)
        tensors: List[torch.Tensor] = []
        spec = ("U", (_flatten(tuple(args), tensors), _flatten(dict(kwargs), tensors)))

        def call(tree):
            return self.f(*tree[0], **tree[1])

        if any(t.requires_grad for t in tensors) and torch.is_grad_enabled():
            raise NotImplementedError("HIP backend: gradients through a jit_compile'd function are not implemented (a replay has no autograd graph): differentiate "
                                      "the eager function, or call the captured one under torch.no_grad() / with detached inputs")
        capturable = bool(tensors) and all(t.is_cuda for t in tensors)
        if not capturable:
            # emulation device / no tensors: the function itself, under the rules of a captured one
            with _tracing():
                return call(_unflatten(spec, iter(tensors)))
        key = (_spec_key(spec), tuple((tuple(t.shape), t.dtype, t.device.index) for t in tensors))
        variants = self.captures.get(key)
        cap = None
        if variants is None:
            if self.forget_traces:
                self.captures.clear()
            while len(self.captures) >= self.MAX_CAPTURES:
                self.captures.pop(next(iter(self.captures)))          # the oldest signature goes (dicts keep insertion order)
            cap = self._capture(spec, tensors, call, alias_inputs=False)
            self.captures[key] = [cap]
        else:
            ptrs = [t.data_ptr() for t in tensors]
            for c in variants:                                          # a capture that READS these very buffers: nothing to copy
                if c.input_ptrs == ptrs:
                    cap = c
                    break
            if cap is None and not self.copy_outputs and len(variants) < 2 and any(q in variants[0].output_ptrs for q in ptrs):
                # r6, two-graph ping-pong (copy_outputs=False, the loop `v, p = step(v, p)`): the caller hands back what the first capture RETURNED -- a second
                # capture is made that reads those output buffers in place; from now on the two alternate and, at 256^3, a step stops copying 268 MB in and
                # 268 MB out (bench.py phi_level: the captured step was 3.7 % SLOWER than the eager one in r5). The first capture still needs its inputs
                # copied in (the second one's outputs are not its input buffers): one copy every other step.
                cap = self._capture(spec, tensors, call, alias_inputs=True)
                variants.append(cap)
            if cap is None:
                cap = variants[0]
                # One copy per tensor, by `copy_`. A fused `torch._foreach_copy_` (one launch, ~3 % faster on the 128^2 plume) is NOT safe in front of a replay on this
                # ROCm build (PHIHIP_JIT_FUSED_COPY=1 switches it on for experiments). Two separate findings: (i) r5 saw the replay behind a fused launch leave the
                # eager bits by a rounding-level amount -- r6 removed the library's share of that (which kernel computed a sample of an advection pass was policy and
                # the paths did not share one arithmetic: csrc/advect_common.hpp; all nine variants of tools/micro/jit_foreach_debug.py then gave the eager bits in a
                # fresh process, profiles/r06_jit_foreach_debug.txt) and switched the fused copy on; (ii) with it on, a replay on a context that had served other
                # grids before FAULTED (GPU memory access fault at the first pure replay, tools/micro/jit_flaky_probe.py: every run with the fused copy, no run of
                # 2 x 24 configurations with per-tensor copies -- profiles/r06_jit_flaky_probe.txt) or, in the test suite, differed from the eager step in the last
                # bits. A torch-only reproducer was attempted (tools/micro/graph_after_foreach_repro.py). So: per-tensor copies ship.
                pairs = [(dst, src) for dst, src in zip(cap.inputs, tensors) if dst.data_ptr() != src.data_ptr()]
                if len(pairs) > 1 and _FUSED_COPY:
                    torch._foreach_copy_([d for d, _ in pairs], [s for _, s in pairs])
                else:
                    for d, s in pairs:
                        d.copy_(s)
                self.input_copies += len(pairs)
        cap.graph.replay()
        self.replays += 1
        outs = [t.clone() for t in cap.outputs] if self.copy_outputs else list(cap.outputs)
        return _unflatten(cap.out_spec, iter(outs))

    def _capture(self, spec, tensors, call, alias_inputs: bool) -> _Capture:
        """ alias_inputs: the graph reads the caller's tensors IN PLACE (they are output buffers of this function's other capture, which the caller promised to
        consume before the call after next: copy_outputs=False); otherwise it owns copies of them """
        device = tensors[0].device
        cap = _Capture()
        cap.spec = spec           # (keeps the auxiliary objects alive whose identity is part of the key)
        cap.inputs = [t.detach() for t in tensors] if alias_inputs else [t.detach().clone() for t in tensors]
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        saved = [t.clone() for t in cap.inputs] if alias_inputs else None
        with torch.cuda.stream(side), _tracing():
            call(_unflatten(spec, iter(cap.inputs)))       # warm-up on the capturing stream: workspaces grown, launch plans tuned, masks rasterised
        side.synchronize()
        for dst, src in zip(cap.inputs, saved if alias_inputs else tensors):      # (a function that wrote into its inputs would have spoilt them: Fields never do, bare tensors may)
            dst.copy_(src)
        torch.cuda.current_stream(device).synchronize()
        cap.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cap.graph, stream=side), _tracing():
            out = call(_unflatten(spec, iter(cap.inputs)))
        cap.outputs = []
        cap.out_spec = _flatten(out, cap.outputs)
        cap.input_ptrs = [t.data_ptr() for t in cap.inputs]
        cap.output_ptrs = {t.data_ptr() for t in cap.outputs}
        self.traces += 1
        return cap


def jit_compile(f: Callable = None, auxiliary_args: str = "", forget_traces: bool = None, copy_outputs: bool = True):
    """ `phiml.math.jit_compile(f, auxiliary_args='', forget_traces=None)`: returns a function with the signature of `f` whose GPU work is
    captured in a hipGraph at the first call with a given signature and replayed afterwards (module docstring). Usable as `@jit_compile` and as
    `@jit_compile(auxiliary_args='dt')`. """
    if f is None:
        return lambda g: JitFunction(g, auxiliary_args, forget_traces, copy_outputs)
    return f if isinstance(f, JitFunction) else JitFunction(f, auxiliary_args, forget_traces, copy_outputs)


def iterate(f: Callable, iterations: int, *x0, f_kwargs: dict = None, measure: Callable = None):
    """ `phiml.math.iterate(f, iterations, *x0, f_kwargs=...)` for an integer count: x <- f(*x, **f_kwargs) `iterations` times, returns the
    final state (a tuple if there are several state variables). `measure` (e.g. `time.perf_counter`): also returns the per-iteration
    differences like PhiML. A trajectory (`iterations=batch(time=N)`) is not built here: append inside `f_kwargs` callbacks or loop by hand. """
    if not isinstance(iterations, int):
        raise NotImplementedError("iterate: pass the number of iterations as an int (a trajectory dimension is not supported on the HIP backend)")
    f_kwargs = f_kwargs or {}
    x = x0
    times = []
    t_prev = measure() if measure else None
    for _ in range(iterations):
        out = f(*x, **f_kwargs)
        x = out if isinstance(out, tuple) else (out,)
        assert len(x) == len(x0), f"function must return the {len(x0)} state variable(s) it was given, got {len(x)}"
        if measure:
            t_now = measure()
            times.append(t_now - t_prev)
            t_prev = t_now
    result = x[0] if len(x0) == 1 else tuple(x)
    return (result, times) if measure else result
