""" TEST DOUBLE of phi.field.Field for uniform grids: geometry + values tensor + extrapolation (phi/field/_field.py:51-211) """
from phiml import math
from phi.geom import UniformGrid


class Field:
    def __init__(self, geometry: UniformGrid, values, extrapolation):
        self.geometry, self.values, self.extrapolation = geometry, values, extrapolation

    @property
    def boundary(self): return self.extrapolation
    @property
    def resolution(self): return self.geometry.resolution
    @property
    def bounds(self): return self.geometry.bounds
    @property
    def is_grid(self): return isinstance(self.geometry, UniformGrid)
    @property
    def is_staggered(self): return '~vector' in self.values.shape.names
    @property
    def is_centered(self): return not self.is_staggered
