#!/usr/bin/env python3
"""
r6 (last session): does the rate of a three-stream pass depend on the RELATIVE OFFSET of the streams inside one allocation? One hipMalloc (physically contiguous in
large fragments on a fresh box) holds a, b, c = 512-MiB views at base offsets 0, 512 MiB + s, 1024 MiB + 2 s; torch.add(a, b, out=c) (two reads, one write) is timed per
skew s. A periodic pattern would give a placement RULE for the CG workspace instead of the candidate search of cg.hip place_workspace; separately allocated buffers
(tools/micro/buffer_bandwidth_probe.py) differ by 3.5 % with the triple.
    python tools/micro/stream_skew_probe.py [MiB per stream]
"""
import json
import sys

import torch


def timed(fn, reps=12):
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
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    dev = torch.device("cuda:0")
    n = mib * (1 << 20) // 4
    slack = (256 << 20) // 4
    for trial in range(2):                                   # two allocations: does the pattern survive another base address?
        pool = torch.zeros(3 * n + 3 * slack, device=dev)
        rows = []
        skews = [0, 256, 512, 1024, 2048, 4096, 8192, 12288, 16384, 32768, 65536, 131072, 262144, 524288, 1 << 20, 3 << 19, 2 << 20, 3 << 20, 4 << 20, 6 << 20, 8 << 20,
                 16 << 20, 32 << 20, 48 << 20, 64 << 20, 96 << 20]
        for s in skews:
            e = s // 4
            a, b, c = pool[0:n], pool[n + e:2 * n + e], pool[2 * n + 2 * e:3 * n + 2 * e]
            t = timed(lambda: torch.add(a, b, out=c))
            rows.append([s, round(t, 5), round(3 * n * 4 / t / 1e6, 1)])
        print(json.dumps({"trial": trial, "MiB": mib, "base": hex(pool.data_ptr()), "skew_bytes_ms_GBs": rows}), flush=True)
        del pool, a, b, c
        torch.cuda.empty_cache()
        hold = torch.zeros(300 << 20, device=dev)            # perturbs where the next pool lands
    del hold


if __name__ == "__main__":
    main()
