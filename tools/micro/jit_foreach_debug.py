#!/usr/bin/env python3
"""
Round 5 record: a captured phi-level step (phiflow_amd/jit.py) replayed after ANY fused `torch._foreach_copy_` launch wrote (some of) its inputs leaves the bits of
the eager step -- first difference per way of writing the inputs: 0 = per-tensor copy_, 1 = fused (plus an element-wise check of the copy after a device
synchronisation: exact), 2 = fused + host synchronisation, 3 = fused + an unrelated small kernel, 4 = per-tensor arithmetic kernels, 5 = fused copy of all inputs but
the pressure guess, 6 = fused copy of the pressure guess alone, 7 = per-tensor copy_ of the inputs followed by a fused copy between UNRELATED tensors, 8 = the same in
the other order. Eager results agree across all of them; the captured ones differ at the first pure replay with 1, 2,
3, 5, 6 -- by the SAME amount, in the projection's results (v, p), not in the smoke -- and never with 0 and 4.
    python tools/micro/jit_foreach_debug.py          (needs an MI355X)
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


def make_call(fin, fout):
    def call(self, *args, **kwargs):
        tensors = []
        spec = ("U", (J._flatten(tuple(args), tensors), J._flatten(dict(kwargs), tensors)))
        key = (J._spec_key(spec), tuple((tuple(t.shape), t.dtype, t.device.index) for t in tensors))
        cap = self.captures.get(key)
        if cap is None:
            cap = self._capture(spec, tensors, lambda tree: self.f(*tree[0], **tree[1]), False)
            self.captures[key] = cap
        else:
            pairs = [(d, s) for d, s in zip(cap.inputs, tensors) if d.data_ptr() != s.data_ptr()]
            if fin == 1 and pairs:
                torch._foreach_copy_([d for d, _ in pairs], [s for _, s in pairs])
                torch.cuda.synchronize()
                for i, (d, s) in enumerate(pairs):
                    if not torch.equal(d, s):
                        print("   COPY MISMATCH pair", i, tuple(d.shape), d.stride(), s.stride(), d.storage_offset(), s.storage_offset(), int((d != s).sum()), float((d - s).abs().max()),
                              "dst ptr % 512:", d.data_ptr() % 512, "src ptr % 512:", s.data_ptr() % 512, flush=True)
            elif fin == 2 and pairs:           # fused copy, then the host waits before the replay
                torch._foreach_copy_([d for d, _ in pairs], [s for _, s in pairs])
                torch.cuda.current_stream().synchronize()
            elif fin == 3 and pairs:           # fused copy, then an unrelated small kernel
                torch._foreach_copy_([d for d, _ in pairs], [s for _, s in pairs])
                torch.zeros(16, device=pairs[0][0].device).add_(1.0)
            elif fin in (5, 6) and pairs:      # 5: fused copy of everything but the pressure guess (the last input), 6: of the pressure guess alone
                fused = pairs[:-1] if fin == 5 else pairs[-1:]
                rest = pairs[-1:] if fin == 5 else pairs[:-1]
                torch._foreach_copy_([d for d, _ in fused], [s for _, s in fused])
                for d, s in rest:
                    d.copy_(s)
            elif fin == 7:                     # per-tensor copy_ of the inputs, then a fused copy between UNRELATED tensors
                for d, s in pairs:
                    d.copy_(s)
                torch._foreach_copy_(DUMMY_B, DUMMY_A)
            elif fin == 8:                     # a fused copy between unrelated tensors FIRST, then per-tensor copy_ of the inputs
                torch._foreach_copy_(DUMMY_B, DUMMY_A)
                for d, s in pairs:
                    d.copy_(s)
            elif fin == 4:                     # per-tensor copies by an arithmetic KERNEL (not the copy engine / blit path of copy_)
                for d, s in pairs:
                    torch.add(s, 0.0, out=d)
            else:
                for d, s in pairs:
                    d.copy_(s)
        cap.graph.replay()
        if fout:
            outs = [torch.empty_like(t) for t in cap.outputs]
            torch._foreach_copy_(outs, cap.outputs)
        else:
            outs = [t.clone() for t in cap.outputs]
        return J._unflatten(cap.out_spec, iter(outs))
    return call


be = HipBackend()
DUMMY_A = [torch.randn(1, 128, 128, device='cuda') for _ in range(4)]
DUMMY_B = [torch.empty_like(t) for t in DUMMY_A]
be.ctx.set_advect_halo(1)
for n, iters in ((128, 50), (192, 20)):
    for fin, fout in ((0, 0), (1, 0), (3, 0), (5, 0), (6, 0), (7, 0), (8, 0)):
        step, v0, s0 = T._plume(be, n)
        jstep = J.jit_compile(step)
        J.JitFunction.__call__ = make_call(fin, fout)
        se, sj = (v0, s0, None), (v0, s0, None)
        first = None
        for k in range(8):
            se = step(*se, iters=iters)
            sj = jstep(*sj, iters=iters)
            for name, fe, fj in zip(("v", "s", "p"), se, sj):
                for c, (a, b) in enumerate(zip(T._np(fe), T._np(fj))):
                    if first is None and not np.array_equal(a, b):
                        first = (k, name, c, float(np.abs(a - b).max()), int((a != b).sum()), a.size)
            import zlib
            crc = lambda st: [zlib.crc32(a.tobytes()) & 0xffff for f in st for a in T._np(f)]
            print(f"   n={n} fin={fin} step {k}: eager {crc(se)} captured {crc(sj)}", flush=True)
        print(f"n={n} foreach inputs={fin} results={fout}: first difference {first}")
