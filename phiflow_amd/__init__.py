"""
phiflow_amd -- an MI355X (gfx950) native backend for PhiFlow's incompressible-fluid time step:
`advect.semi_lagrangian` + `fluid.make_incompressible` on `StaggeredGrid`s, implemented as hand-written HIP kernels in
`libphihip.so` (C ABI: include/phihip.h) and exposed through PhiFlow's own operator names.

    from phiflow_amd.flow import *
    v = StaggeredGrid(0, PERIODIC, x=256, y=256, z=256)
    v = advect.semi_lagrangian(v, v, dt)
    v, p = fluid.make_incompressible(v, (), Solve('CG', 1e-5, x0=None))

There is no CPU fallback: importing this package is cheap, but the first operator call loads libphihip.so and creates a
HIP context, and fails loudly if either is unavailable.
"""
__version__ = "0.1.0"
