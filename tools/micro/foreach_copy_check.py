#!/usr/bin/env python3
"""
Is `torch._foreach_copy_` the per-tensor `copy_` on this ROCm build? (round 5: phiflow_amd/jit.py copies a captured step's inputs and results tensor by tensor
because the bit comparison with the eager steps failed with the fused form at 128^2 / 192^2 and passed at 32^2.) Shapes of the plume's state; every call is
followed by a graph replay in the real use, so the check also runs the copies between replays of a small captured graph.
"""
import torch

dev = torch.device("cuda:0")
torch.manual_seed(0)
for n in (32, 128, 192, 256):
    shapes = [(1, n - 1, n), (1, n, n - 1), (1, n, n), (1, n, n)]
    src = [torch.randn(s, device=dev) for s in shapes]
    a = [torch.empty_like(t) for t in src]
    b = [torch.empty_like(t) for t in src]
    torch._foreach_copy_(a, src)
    for d, s_ in zip(b, src):
        d.copy_(s_)
    torch.cuda.synchronize()
    plain = all(torch.equal(x, y) for x, y in zip(a, b))
    # between replays of a graph that reads the copied tensors
    static = [torch.zeros_like(t) for t in src]
    out = [torch.zeros_like(t) for t in src]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for o, t in zip(out, static):
            o.copy_(t * 2.0 + 1.0)
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for o, t in zip(out, static):
            o.copy_(t * 2.0 + 1.0)
    bad = 0
    for k in range(50):
        cur = [torch.randn(s, device=dev) for s in shapes]
        torch._foreach_copy_(static, cur)
        g.replay()
        res = [torch.empty_like(o) for o in out]
        torch._foreach_copy_(res, out)
        ref = [c * 2.0 + 1.0 for c in cur]
        torch.cuda.synchronize()
        bad += sum(int(not torch.equal(x, y)) for x, y in zip(res, ref))
    print(f"n={n}: foreach == per-tensor on fresh tensors: {plain}; wrong tensors over 50 copy / replay / copy rounds: {bad} of {50 * len(shapes)}")
