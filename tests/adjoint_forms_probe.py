"""
TEST HELPER (run as a subprocess by tests/test_adjoint_forms.py): the staggered advection adjoints on one seeded case; prints a SHA-256 of every gradient array. The launch form
(all components per launch / one component per launch) is chosen by PHIHIP_ADJOINT_ALL in the environment of the process -- the library reads it once.
    python tests/adjoint_forms_probe.py emu|gpu
"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parity_cases as pc          # noqa: E402
from parity_cases import CLO, OPN, PER   # noqa: E402
from phiflow_amd import _capi      # noqa: E402


def main():
    where = sys.argv[1]
    if where == "emu":
        os.environ["PHIHIP_AUTOTUNE"] = "0"
        lib = _capi.Library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu", "libphihip_emu.so"))
        mem = pc.NumpyMem()
    else:
        lib = _capi.load_default_library()
        mem = pc.TorchMem()
    ctx = _capi.Context(lib, 0)
    P = lambda hs: [mem.ptr(h) for h in hs]
    for res, bc, dtype in (((9, 7, 10), ((CLO, CLO), (OPN, OPN), (CLO, OPN)), np.float64), ((8, 12, 16), ((PER, PER),) * 3, np.float32), ((16, 20), ((OPN, OPN), (CLO, OPN)), np.float64)):
        rng = np.random.default_rng(77)
        dom, grid = pc.make_case(res, bc, dtype, batch=2)
        v = [0.5 * a for a in pc.random_velocity(dom, 2, dtype, rng)]
        g = pc.random_velocity(dom, 2, dtype, rng)
        dt = 0.2 * min(dom.dx)                                   # CFL < 1 everywhere: no sample takes the atomic path (whose summation order is not fixed on a GPU)
        dv, dg = [mem.to_dev(a) for a in v], [mem.to_dev(a) for a in g]
        gf, gv = [mem.to_dev(np.zeros_like(a)) for a in v], [mem.to_dev(np.zeros_like(a)) for a in v]
        ctx.advect_staggered_backward(grid, P(dv), P(dv), P(dg), dt, P(gf), P(gv))
        mf, mv = [mem.to_dev(np.zeros_like(a)) for a in v], [mem.to_dev(np.zeros_like(a)) for a in v]
        ctx.mac_cormack_staggered_backward(grid, P(dv), P(dv), P(dg), dt, 1.0, P(mf), P(mv))
        mem.sync()
        h = hashlib.sha256()
        for a in gf + gv + mf + mv:
            h.update(np.ascontiguousarray(mem.to_host(a)).tobytes())
        print(res, np.dtype(dtype).name, h.hexdigest(), flush=True)


if __name__ == "__main__":
    main()
