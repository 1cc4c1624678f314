#!/usr/bin/env python3
"""
Round 5: what does a replayed graph SEE of inputs that a fused torch._foreach_copy_ wrote? The captured function returns, next to its results, two images of every
input taken inside the graph -- one by `clone()` (a memcpy node), one by `x + 0` (a kernel node) -- and the caller compares them with what it passed in.
    python tools/micro/jit_foreach_probe.py [n]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools", "micro"))
from phiflow_amd import jit as J                      # noqa: E402
from phiflow_amd.backend import HipBackend            # noqa: E402
import test_jit as T                                  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
be = HipBackend()
be.ctx.set_advect_halo(1)
for fused in (0, 1):
    step, v0, s0 = T._plume(be, n)

    def probe(v, s, p, iters=50):
        ins = list(v.values) + [s.values] + ([p.values] if p is not None else [])
        by_memcpy = [t.clone() for t in ins]
        by_kernel = [t + 0 for t in ins]
        out = step(v, s, p, iters=iters)
        return out, by_memcpy, by_kernel
    jprobe = J.jit_compile(probe)
    if fused:
        def call(self, *args, **kwargs):
            tensors = []
            spec = ("U", (J._flatten(tuple(args), tensors), J._flatten(dict(kwargs), tensors)))
            key = (J._spec_key(spec), tuple((tuple(t.shape), t.dtype, t.device.index) for t in tensors))
            cap = self.captures.get(key)
            if cap is None:
                cap = self._capture(spec, tensors, lambda tree: self.f(*tree[0], **tree[1]), False)
                self.captures[key] = cap
            else:
                torch._foreach_copy_(cap.inputs, tensors)
            cap.graph.replay()
            return J._unflatten(cap.out_spec, iter([t.clone() for t in cap.outputs]))
        J.JitFunction.__call__ = call
    se, sj = (v0, s0, None), (v0, s0, None)
    for k in range(7):
        se = step(*se, iters=50)
        ins = list(sj[0].values) + [sj[1].values] + ([sj[2].values] if sj[2] is not None else [])
        sj, by_memcpy, by_kernel = jprobe(*sj)
        torch.cuda.synchronize()
        seen_m = [bool(torch.equal(a, b)) for a, b in zip(ins, by_memcpy)]
        seen_k = [bool(torch.equal(a, b)) for a, b in zip(ins, by_kernel)]
        same = T._same(se, sj)
        print(f"fused={fused} step {k}: results equal eager: {same}; inputs as the graph's memcpy nodes saw them equal what was passed: {seen_m}; as its kernel nodes saw them: {seen_k}", flush=True)
