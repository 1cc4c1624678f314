""" TEST DOUBLE of phi.geom: Box / Sphere with named-vector centres and bounds, UniformGrid geometry of a field """
from phiml import math


def _vec(**c):
    return math.tensor([float(v) for v in c.values()], math.channel(vector=tuple(c.keys())))


class Geometry:
    @property
    def vector(self): return self.center.shape.only(math.CHANNEL)
    @property
    def shape(self): return math.Shape()


class Box(Geometry):
    def __init__(self, **bounds):
        self.lower = _vec(**{d: (b[0] if isinstance(b, tuple) else 0.0) for d, b in bounds.items()})
        self.upper = _vec(**{d: (b[1] if isinstance(b, tuple) else b) for d, b in bounds.items()})
        self.center = math.tensor((self.lower._native + self.upper._native) / 2, self.lower.shape)


class Sphere(Geometry):
    def __init__(self, radius=1.0, **center):
        self.center, self.radius = _vec(**center), radius


class UniformGrid(Geometry):
    def __init__(self, resolution, bounds: Box):
        self.resolution, self.bounds, self.center = resolution, bounds, bounds.center
