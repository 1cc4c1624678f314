#!/usr/bin/env python3
"""
Round 5: does a replayed step depend on what OTHER calls left in the library's workspaces? Reference: the captured plume step iterated alone. Test: the same, with an
eager step on an UNRELATED state (another plume, perturbed) run on the same context before every replay. A captured step that reads only what it writes gives the
same bits in both.      python tools/micro/jit_leftover_check.py [n] [steps]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from phiflow_amd import jit as J                      # noqa: E402
from phiflow_amd.backend import HipBackend            # noqa: E402
import test_jit as T                                  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
be = HipBackend()
be.ctx.set_advect_halo(1)
step, v0, s0 = T._plume(be, n)


def run(interleave, eager):
    fn = step if eager else J.jit_compile(step)
    st = (v0, s0, None)
    other = (v0, s0 + 0.37, None)
    outs = []
    for k in range(steps):
        if interleave:
            other = step(*other, iters=50)           # an unrelated eager step: different smoke, velocity, pressure in every workspace
        st = fn(*st, iters=50)
        outs.append([a.copy() for f in st for a in T._np(f)])
    return outs


ref = run(False, True)
for name, interleave, eager in (("eager alone", False, True), ("eager + unrelated eager steps", True, True), ("captured alone", False, False),
                                ("captured + unrelated eager steps", True, False)):
    res = run(interleave, eager)
    first = next((k for k in range(steps) if not all(np.array_equal(a, b) for a, b in zip(res[k], ref[k]))), None)
    print(f"n={n} {name:34s}: first step that differs from the eager reference: {first}", flush=True)
