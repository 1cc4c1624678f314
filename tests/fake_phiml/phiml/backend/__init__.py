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
        """ the generic path of PhiML's backends [PHIML-RECALL: Backend.conjugate_gradient, SURVEY Appendix B.2]: CG on natives with
        `linear(lin, v) = lin @ v + matrix_offset * sum(v)` -- the reference the HIP override is compared with. fp64 NumPy. """
        import numpy as np
        if method not in ('CG', 'auto') or pre is not None:
            raise NotImplementedError("generic sparse solve: the test double implements CG only")
        if hasattr(lin, 'layout'):                                   # torch sparse tensor -> SciPy
            import scipy.sparse as sp
            t = lin.to_sparse_coo().coalesce()
            i = t.indices().numpy()
            lin = sp.csr_matrix((t.values().numpy(), (i[0], i[1])), shape=tuple(t.shape))
        Y = np.asarray(y, np.float64)
        X = np.array(np.asarray(x0, np.float64), copy=True)
        A = lambda v: (lin @ v) + (0.0 if matrix_offset is None else float(matrix_offset) * v.sum())
        its, conv, div, res = [], [], [], []
        for b in range(Y.shape[0]):
            x, yb = X[b], Y[b]
            tol_sq = max(float(rtol) ** 2 * float(yb @ yb), float(atol) ** 2)
            r = yb - A(x); d = r.copy(); rsq = float(r @ r); k = 0
            while rsq > tol_sq and k < int(max_iter):
                q = A(d); alpha = rsq / float(d @ q)
                x += alpha * d; r -= alpha * q
                rsq_new = float(r @ r); d = r + (rsq_new / rsq) * d; rsq = rsq_new; k += 1
            its.append(k); conv.append(rsq <= tol_sq); div.append(False); res.append(np.sqrt(rsq))
        return SolveResult('generic CG (test double)', X, np.asarray(res), np.asarray(its), np.asarray(its), np.asarray(conv), np.asarray(div), [""] * len(its))

    def conjugate_gradient(self, lin, y, x0, rtol, atol, max_iter, pre=None, matrix_offset=None):
        return Backend.linear_solve(self, 'CG', lin, y, x0, rtol, atol, max_iter, pre, matrix_offset)

    def __enter__(self): _DEFAULT.append(self); return self
    def __exit__(self, *a): _DEFAULT.pop(); return False


BACKENDS = []
_DEFAULT = []


def default_backend(): return _DEFAULT[-1] if _DEFAULT else None
def set_global_default_backend(b): _DEFAULT[:] = [b]
