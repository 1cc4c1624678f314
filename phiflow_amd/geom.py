"""
Geometry subset needed by the grid fluid step: `Box` (domain bounds and box obstacles) and `Sphere`
(reference: phi/geom/_box.py:46-236, phi/geom/_sphere.py). Geometries are only ever *rasterised to masks on the host*
-- exactly what PhiFlow does (`with NUMPY:` in phi/physics/fluid.py:132).
"""
from typing import Dict, Sequence, Tuple

import numpy as np


class Vector(dict):
    """ named vector `vec(x=1, y=0)` """


def vec(**components) -> Vector:
    """ `vec(x=1, y=0)`: named vector, e.g. a wall velocity for a constant extrapolation """
    return Vector(components)


class Geometry:
    dims: Tuple[str, ...]

    def lies_inside(self, points: Sequence[np.ndarray]) -> np.ndarray:
        raise NotImplementedError

    def approximate_signed_distance(self, points: Sequence[np.ndarray]) -> np.ndarray:
        raise NotImplementedError


class _BoxType(type):
    def __getitem__(cls, item):
        """ `Box['x,y', 0:100, 0:100]` (tests/commit/physics/test_fluid.py:23) """
        assert isinstance(item, tuple) and isinstance(item[0], str), "use Box['x,y', 0:1, 0:1]"
        dims = [d.strip() for d in item[0].split(',')]
        assert len(item) == len(dims) + 1
        kwargs = {}
        for d, sl in zip(dims, item[1:]):
            kwargs[d] = (0 if sl.start is None else sl.start, sl.stop)
        return cls(**kwargs)


class Box(Geometry, metaclass=_BoxType):
    """ axis-aligned box; `Box(x=100, y=(10, 20))`: a number means (0, number) """

    def __init__(self, **bounds):
        self.dims = tuple(bounds.keys())
        self.lower = tuple(float(b[0]) if isinstance(b, (tuple, list)) else 0.0 for b in bounds.values())
        self.upper = tuple(float(b[1]) if isinstance(b, (tuple, list)) else float(b) for b in bounds.values())

    @property
    def size(self):
        return tuple(u - l for l, u in zip(self.lower, self.upper))

    @property
    def center(self):
        return tuple((u + l) / 2 for l, u in zip(self.lower, self.upper))

    @property
    def half_size(self):
        return tuple((u - l) / 2 for l, u in zip(self.lower, self.upper))

    def lies_inside(self, points):
        """ |x - c| <= half, inclusive (phi/geom/_box.py:174-185) """
        ok = np.ones(points[0].shape, dtype=bool)
        for a in range(len(self.dims)):
            ok &= np.abs(points[a] - self.center[a]) <= self.half_size[a]
        return ok

    def approximate_signed_distance(self, points):
        """ L-infinity distance to the surface (phi/geom/_box.py:217-236) """
        dist = None
        for a in range(len(self.dims)):
            da = np.abs(points[a] - self.center[a]) - self.half_size[a]
            dist = da if dist is None else np.maximum(dist, da)
        return dist

    def __repr__(self):
        return "Box(" + ", ".join(f"{d}=({l}, {u})" for d, l, u in zip(self.dims, self.lower, self.upper)) + ")"


Cuboid = Box


class Sphere(Geometry):
    """ `Sphere(x=50, y=10, radius=5)` """

    def __init__(self, radius: float, **center):
        self.dims = tuple(center.keys())
        self.center = tuple(float(c) for c in center.values())
        self.radius = float(radius)

    def lies_inside(self, points):
        d2 = sum((p - c) ** 2 for p, c in zip(points, self.center))
        return d2 <= self.radius ** 2

    def approximate_signed_distance(self, points):
        d2 = sum((p - c) ** 2 for p, c in zip(points, self.center))
        return np.sqrt(d2) - self.radius

    def __repr__(self):
        return f"Sphere({dict(zip(self.dims, self.center))}, radius={self.radius})"


def union_lies_inside(geometries: Sequence[Geometry], points) -> np.ndarray:
    inside = np.zeros(points[0].shape, dtype=bool)
    for g in geometries:
        inside |= g.lies_inside(points)
    return inside
