"""
`from phiflow_amd.flow import *` -- the counterpart of `from phi.torch.flow import *` (reference: phi/torch/flow.py:15-35,
phi/flow.py:13-28) for the part of PhiFlow that this backend accelerates: grid fields and the incompressible fluid step.
"""
from . import advect, diffuse, fluid
from . import extrapolation
from .autodiff import functional_gradient, gradient, jacobian, l2_loss, stop_gradient
from .backend import HipBackend, default_backend, precision, set_global_default_backend, set_global_precision
from .extrapolation import BOUNDARY, ONE, PERIODIC, ZERO, ZERO_GRADIENT, ConstantExtrapolation, combine_sides
from .field import CenteredGrid, Field, StaggeredGrid, assert_close, divergence, mean, resample, spatial_gradient
from . import geom
from .geom import Box, Cuboid, Sphere, embed, infinite_cylinder, union, vec
from .fluid import Obstacle
from .solve import ConvergenceException, Diverged, NotConverged, Solve, SolveInfo, copy_with
from .linear import solve_linear
from .jit import iterate, jit_compile

__all__ = [
    'advect', 'diffuse', 'fluid', 'extrapolation',
    'HipBackend', 'default_backend', 'precision', 'set_global_default_backend', 'set_global_precision',
    'BOUNDARY', 'ONE', 'PERIODIC', 'ZERO', 'ZERO_GRADIENT', 'ConstantExtrapolation', 'combine_sides',
    'CenteredGrid', 'Field', 'StaggeredGrid', 'assert_close', 'divergence', 'mean', 'resample', 'spatial_gradient',
    'geom', 'Box', 'Cuboid', 'Sphere', 'embed', 'infinite_cylinder', 'union', 'vec', 'Obstacle',
    'functional_gradient', 'gradient', 'jacobian', 'l2_loss', 'stop_gradient',
    'ConvergenceException', 'Diverged', 'NotConverged', 'Solve', 'SolveInfo', 'copy_with', 'solve_linear',
    'iterate', 'jit_compile',
]
