""" TEST DOUBLE of phiml.math.extrapolation: singletons + ConstantExtrapolation + per-side mixes """


class Extrapolation:
    def __repr__(self): return type(self).__name__


class _Periodic(Extrapolation): pass
class _Boundary(Extrapolation): pass


class ConstantExtrapolation(Extrapolation):
    def __init__(self, value):
        from . import tensor
        self.value = tensor(value)

    def __repr__(self): return f"Constant({self.value._native.tolist()})"


class _MixedExtrapolation(Extrapolation):
    def __init__(self, ext): self.ext = dict(ext)


PERIODIC = _Periodic()
BOUNDARY = ZERO_GRADIENT = _Boundary()
ZERO = ConstantExtrapolation(0.0)
ONE = ConstantExtrapolation(1.0)


def combine_sides(**dims):
    return _MixedExtrapolation({d: (v if isinstance(v, tuple) else (v, v)) for d, v in dims.items()})
