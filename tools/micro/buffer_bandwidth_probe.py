#!/usr/bin/env python3
"""
r6 (last session): is the per-allocation spread of a CG iteration (tools/micro/ws_placement_probe.py) a property of a SINGLE buffer (a plain read + write pass over it is
slower than over its neighbour) or of the COMBINATION of buffers a kernel streams at once? M buffers of the size of a 512^3 fp32 vector, each its own hipMalloc
(torch hands allocations of this size through), timed alone (b *= c: one read, one write) and in triples (torch.add(a, b, out=c): two reads, one write).
    python tools/micro/buffer_bandwidth_probe.py [buffers] [MiB per buffer]
"""
import itertools
import json
import sys

import torch


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    mib = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    dev = torch.device("cuda:0")
    n = mib * (1 << 20) // 4
    bufs = [torch.zeros(n, device=dev) for _ in range(m)]
    print(json.dumps({"buffers": m, "MiB": mib, "addresses": [hex(b.data_ptr()) for b in bufs]}), flush=True)
    single = [round(timed(lambda b=b: b.mul_(1.0)), 5) for b in bufs]
    print(json.dumps({"single_ms_read_write": single, "GBs": [round(2 * n * 4 / t / 1e6, 1) for t in single]}), flush=True)
    single2 = [round(timed(lambda b=b: b.mul_(1.0)), 5) for b in bufs]
    print(json.dumps({"single_ms_read_write_again": single2}), flush=True)
    triples = list(itertools.permutations(range(min(m, 5)), 3))[:40]
    rows = []
    for (i, j, k) in triples:
        t = timed(lambda: torch.add(bufs[i], bufs[j], out=bufs[k]), reps=10)
        rows.append([i, j, k, round(t, 5)])
    print(json.dumps({"triples_ms_two_reads_one_write": rows}), flush=True)


if __name__ == "__main__":
    main()
