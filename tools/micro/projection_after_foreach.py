#!/usr/bin/env python3
"""
Round 5: does the PROJECTION itself (no capture) depend on what ran on the device just before it? The replay of a captured plume step leaves the eager bits when a
fused torch._foreach_copy_ launch precedes it (jit_foreach_debug.py). Here the eager `fluid.make_incompressible` is run after different predecessors -- nothing,
a per-tensor copy, a fused foreach copy, a large random fill -- on the same inputs, and as a captured graph after the same predecessors; every result is compared bit
for bit with the first.        python tools/micro/projection_after_foreach.py [n]
"""
import os
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from phiflow_amd import jit as J                      # noqa: E402
from phiflow_amd.backend import HipBackend            # noqa: E402
from phiflow_amd.flow import *                        # noqa: E402,F401,F403
import test_jit as T                                  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
be = HipBackend()
be.ctx.set_advect_halo(1)
step, v0, s0 = T._plume(be, n)
state = (v0, s0, None)
for _ in range(4):
    state = step(*state, iters=50)
v_pre = state[0] + resample(state[1] * (0, 0.1), to=state[0])
p0 = state[2]
dev = v_pre.values[0].device
a = [torch.randn(1, n, n, device=dev) for _ in range(4)]
b = [torch.empty_like(t) for t in a]


def project(v, p):
    return fluid.make_incompressible(v, (), Solve('CG', 0, 0, x0=p, max_iterations=50, suppress=[NotConverged]))


def crc(fields):
    return [zlib.crc32(x.tobytes()) & 0xffffff for f in fields for x in T._np(f)]


PRED = {"nothing": lambda: None,
        "per-tensor copy_": lambda: [d.copy_(s) for d, s in zip(b, a)],
        "fused foreach copy": lambda: torch._foreach_copy_(b, a),
        "fused foreach copy of ONE tensor": lambda: torch._foreach_copy_(b[:1], a[:1]),
        "large random fill": lambda: torch.randn(64, 1024, 1024, device=dev)}
ref = None
for mode in ("eager", "captured"):
    fn = project if mode == "eager" else J.jit_compile(project)
    for name, pred in PRED.items():
        for rep in range(2):
            torch.cuda.synchronize()
            pred()
            out = fn(v_pre, p0)
            torch.cuda.synchronize()
            c = crc(out)
            ref = ref or c
            print(f"{mode:9s} after {name:34s} #{rep}: {'same bits' if c == ref else 'DIFFERENT ' + str(c)}", flush=True)
