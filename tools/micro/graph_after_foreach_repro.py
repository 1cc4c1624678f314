#!/usr/bin/env python3
"""
r6: does a hipGraph replay that follows a fused multi-tensor launch (torch._foreach_copy_) misbehave WITHOUT libphihip in the process? (VERDICT r5 item 1a asked for
"a 30-line reproducer without libphihip that shows the runtime at fault" if the library could not be shown at fault.) torch only: a graph of a few hundred small
elementwise / reduction kernels, its inputs written by a fused copy (variant 1) or by per-tensor copies (variant 0), its outputs compared bit for bit with the same
ops run eagerly.            python tools/micro/graph_after_foreach_repro.py
"""
import torch

dev = torch.device("cuda:0")
torch.manual_seed(0)
n = 192


def work(v0, v1, s, p):
    for _ in range(40):
        s = s * 0.999 + torch.roll(s, 1, 0) * 0.001
        v0 = v0 + 0.1 * s[:-1, :] * 0.5 + 0.1 * s[1:, :] * 0.5
        v1 = v1 * 0.99 + 0.01 * v1.mean()
        p = p + 0.25 * (torch.roll(p, 1, 1) + torch.roll(p, -1, 1) - 2 * p) + 1e-3 * s
    return v0, v1, s, p


def main():
    # a "used" process: big allocations made and released, so the caching allocator serves later tensors from split blocks
    junk = [torch.randn(64, 256, 256, device=dev) for _ in range(6)]
    del junk
    shapes = [(n - 1, n), (n, n - 1), (n, n), (n, n)]
    for fused in (0, 1):
        ins = [torch.randn(s, device=dev) for s in shapes]
        static = [t.clone() for t in ins]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            work(*static)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            outs = work(*static)
        bad = 0
        state = ins
        for k in range(40):
            ref = work(*state)
            if fused:
                torch._foreach_copy_(static, list(state))
            else:
                for d, s_ in zip(static, state):
                    d.copy_(s_)
            g.replay()
            got = [o.clone() for o in outs]
            torch.cuda.synchronize()
            if not all(torch.equal(a, b) for a, b in zip(ref, got)):
                bad += 1
                print(f"fused={fused} step {k}: replay differs from eager, max {max(float((a - b).abs().max()) for a, b in zip(ref, got)):.3e}", flush=True)
            state = [t / (1.0 + t.abs().max()) for t in ref]
        print(f"fused={fused}: {bad} of 40 replays differ", flush=True)


if __name__ == "__main__":
    main()
