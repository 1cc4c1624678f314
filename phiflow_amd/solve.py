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
    Unset tolerances default to 1e-5 (fp32) / 1e-12 (fp64). Supported methods: 'CG' and 'auto' (-> CG). """
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
    # true-residual refresh period of PhiML's cg (50)
    check_every: int = 10
    refresh_every: int = 50

    def with_defaults(self, fp64: bool) -> 'Solve':
        default = 1e-12 if fp64 else 1e-5
        return replace(self, rel_tol=default if self.rel_tol is None else float(self.rel_tol),
                       abs_tol=default if self.abs_tol is None else float(self.abs_tol))

    def with_preprocessing(self, preprocess_y: Callable, *args) -> 'Solve':
        return replace(self, preprocess_y=preprocess_y, preprocess_y_args=args)


def copy_with(obj, **changes):
    """ `phiml.math.copy_with` for dataclasses (phi/physics/fluid.py:148-151) """
    return replace(obj, **changes)
