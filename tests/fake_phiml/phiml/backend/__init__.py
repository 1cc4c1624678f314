""" TEST DOUBLE of phiml.backend: the registry PhiFlow's detect_backends() fills (phi/__init__.py:41-63) and a Backend base class """
from collections import namedtuple

SolveResult = namedtuple('SolveResult', ['method', 'x', 'residual', 'iterations', 'function_evaluations', 'converged', 'diverged', 'message'])


class Backend:
    def __init__(self, name='base', devices=(), default_device=None):
        self._name = name

    @property
    def name(self): return self._name

    def grid_sample(self, grid, coordinates, extrapolation: str):
        raise NotImplementedError("generic gather (not part of the test double)")

    def linear_solve(self, method, lin, y, x0, rtol, atol, max_iter, pre=None, matrix_offset=None):
        raise NotImplementedError("generic sparse solve (not part of the test double)")

    def __enter__(self): _DEFAULT.append(self); return self
    def __exit__(self, *a): _DEFAULT.pop(); return False


BACKENDS = []
_DEFAULT = []


def default_backend(): return _DEFAULT[-1] if _DEFAULT else None
def set_global_default_backend(b): _DEFAULT[:] = [b]
