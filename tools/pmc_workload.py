#!/usr/bin/env python3
"""
Workload for the rocprofv3 PMC passes (run once per counter set, see tools/sessions/gpu_session.sh):
  1. calibration: a plain 16 B/lane streaming copy of a known size (torch copy_ of 512 MiB fp32) -- known bytes read/written
  2. the CG loop at the sizes given on the command line (default 512^3 and 256^3) fp32, a few iterations, default launch plan
     -- one size per profiled process keeps kernels with identical template arguments and grids apart
The summary (tools/pmc_summary.py) turns FETCH_SIZE / WRITE_SIZE per dispatch into HBM bytes per launch.
"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phiflow_amd import _capi as C   # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [512, 256]
    dev = torch.device("cuda:0")
    a = torch.randn(128 * 1024 * 1024, device=dev)       # 512 MiB
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    del a, b
    lib = C.load_default_library()
    ctx = C.Context(lib, 0)
    # r5: the caller (bench.py) hands over the launch plans of ITS solve -- {"<size>": {"<family>": [rows, tpr, chunk]}} -- and the traced solve runs
    # exactly those with the autotune off. Until r4 the traced process tuned for itself: candidate launches of the winning tile with OTHER chunk
    # lengths carry the same kernel name (the chunk is a run-time argument) and were averaged in -- short chunks re-read two halo planes each, which is
    # where the bench line's traffic_over_moved of 1.096 came from while the pinned-plan table (tools/path_workload.py) said 0.998.
    import json
    pinned = json.loads(os.environ.get("PHIHIP_PMC_PLANS", "{}") or "{}")
    L = 2 * math.pi
    for n, iters in [(n, 24 if n >= 512 else 40) for n in sizes]:   # (many more launches than the autotune spends on any one candidate)
        grid = C.make_grid(3, C.PHIHIP_F32, 1, (n, n, n), (0, 0, 0), (L, L, L), ((0, 0),) * 3)
        g = torch.Generator(device="cpu").manual_seed(0)
        rhs = torch.randn(1, n, n, n, generator=g)
        rhs -= rhs.mean()
        rhs = rhs.to(dev)
        x = torch.zeros_like(rhs)
        plans = pinned.get(str(n))
        if plans:
            ctx.set_autotune(False)
            for fam, (rows, tpr, chunk) in plans.items():
                ctx.set_tuning_kernel(int(fam), int(rows), int(tpr), int(chunk))
        else:      # no plans given: tune first (untraced kernels of other names / chunks still appear in the trace: use the pinned form)
            ctx.set_autotune(True)
        ctx.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, iters, 0, 0, 0), want_info=False)
        torch.cuda.synchronize()
        del rhs, x


if __name__ == "__main__":
    main()
