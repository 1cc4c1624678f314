""" TEST DOUBLE of phi.physics.fluid: the entry point the plug-in patches + the helpers it borrows (fluid.py:21-91,264-288). The
"reference implementation" here only records that it was called -- the fall-back branch of the drop-in is what the test observes. """
from phiml.math import extrapolation as e

CALLS = []


class Obstacle:
    def __init__(self, geometry, velocity=0, angular_velocity=0):
        self.geometry, self.velocity, self.angular_velocity = geometry, velocity, angular_velocity


def _get_obstacles_for(obstacles, space):
    obstacles = [obstacles] if not isinstance(obstacles, (tuple, list)) else list(obstacles)
    return [o if isinstance(o, Obstacle) else Obstacle(o) for o in obstacles]


def _pressure_extrapolation(vext):
    if vext is e.PERIODIC: return e.PERIODIC
    if vext is e.BOUNDARY: return e.ZERO
    if isinstance(vext, e.ConstantExtrapolation): return e.BOUNDARY
    return e._MixedExtrapolation({d: (_pressure_extrapolation(lo), _pressure_extrapolation(up)) for d, (lo, up) in vext.ext.items()})


def make_incompressible(velocity, obstacles=(), solve=None, active=None, order=2, correct_skew=False, wide_stencil=None):
    CALLS.append(('make_incompressible', order))
    return velocity, None
