"""
`Solve`, `SolveInfo` and the convergence exceptions of `phiml.math.solve_linear`, restricted to what
`fluid.make_incompressible` uses (reference call sites: phi/physics/fluid.py:96,145-156; usages
`Solve('CG', 1e-3, x0=p)` Smoke_Plume.ipynb cell 5, `Solve('CG', 1e-12, 1e-12, x0=p)` Taylor_Green.ipynb cell 12,
`Solve('auto', 1e-5, x0=p, max_iterations=100000)` demos/Top_Opt/Top_Opt3D.py:70).
"""
from dataclasses import dataclass, field as _f, replace
from typing import Any, Callable, List, Optional, Sequence, Tuple


class ConvergenceException(RuntimeError):
    """ Base class of `NotConverged` and `Diverged` (phiml.math) """

    def __init__(self, result: 'SolveInfo'):
        super().__init__(result.msg)
        self.result = result


class NotConverged(ConvergenceException):
    """ raised when the solver hit `max_iterations` before reaching the tolerance """


class Diverged(ConvergenceException):
    """ raised when the residual grew beyond 100x its initial value (after >= 8 iterations) or became non-finite """


@dataclass
class SolveInfo:
    solve: 'Solve'
    iterations: List[int]
    residual_sq: List[float]
    rhs_sq: List[float]
    converged: List[bool]
    diverged: List[bool]
    msg: str = ""


@dataclass(frozen=True)
class Solve:
    """ `Solve(method, rel_tol, abs_tol, x0, max_iterations, suppress, preprocess_y, rank_deficiency)`.
    Unset tolerances default to 1e-5 (fp32) / 1e-12 (fp64). Supported methods: 'CG', 'auto' (-> CG) and 'CG-adaptive'
    (examples/grids/Fluid_Logo.ipynb; SURVEY Appendix B.2: alpha = d.r / d.Ad, d' = r' - (r'.Ad / d.Ad) d). """
    method: str = 'auto'
    rel_tol: Optional[float] = None
    abs_tol: Optional[float] = None
    x0: Any = None
    max_iterations: int = 1000
    suppress: Sequence[type] = ()
    preprocess_y: Optional[Callable] = None
    preprocess_y_args: tuple = ()
    rank_deficiency: Optional[int] = None
    gradient_solve: Optional['Solve'] = None      # solve used by the backward pass (phiml: defaults to this solve)
    # backend-specific knobs (not in PhiML): how often the host polls the device-side continue flags, and the
    # true-residual refresh period (None: PhiML's value for the method -- 50 for 'CG', 20 for 'CG-adaptive'; 0 = never)
    check_every: int = 10
    refresh_every: Optional[int] = None

    def with_defaults(self, fp64: bool) -> 'Solve':
        default = 1e-12 if fp64 else 1e-5
        refresh = (20 if self.method == 'CG-adaptive' else 50) if self.refresh_every is None else int(self.refresh_every)
        return replace(self, rel_tol=default if self.rel_tol is None else float(self.rel_tol),
                       abs_tol=default if self.abs_tol is None else float(self.abs_tol), refresh_every=refresh)

    METHODS = {'auto': 0, 'CG': 0, 'CG-adaptive': 1}     # phihip_method (include/phihip.h)

    def to_c(self, fp64: bool):
        """ the `phihip_solve` struct of this solve """
        from . import _capi
        if self.method not in self.METHODS:
            raise NotImplementedError(f"HIP backend: Solve(method={self.method!r}) is not available, use one of {tuple(self.METHODS)}")
        s = self.with_defaults(fp64)
        return _capi.Solve(s.rel_tol, s.abs_tol, int(s.max_iterations), int(s.refresh_every), int(s.check_every), self.METHODS[self.method])

    def with_preprocessing(self, preprocess_y: Callable, *args) -> 'Solve':
        return replace(self, preprocess_y=preprocess_y, preprocess_y_args=args)


def copy_with(obj, **changes):
    """ `phiml.math.copy_with` for dataclasses (phi/physics/fluid.py:148-151) """
    return replace(obj, **changes)
