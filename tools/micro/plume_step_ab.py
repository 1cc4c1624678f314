#!/usr/bin/env python3
""" r6: the 128^2 smoke-plume step of bench.py's phi_level block through the C ABI (SmokeBatchStep, one entry, 50 CG iterations), per resident-solver mode; the launch form
(cooperative / plain) comes from PHIHIP_RESIDENT_COOP in the environment.   python tools/micro/plume_step_ab.py [n] [steps] """
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as B   # noqa: E402
from phiflow_amd import _capi as C   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
lib = C.load_default_library()
for mode in (0, 1, 2):
    ctx = C.Context(lib, 0)
    ctx.set_resident_cg(mode)
    sim = B.SmokeBatchStep(ctx, n, 1, 0, 1, 50, dev)
    for _ in range(10):
        sim.step(None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sim.step(None)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    ctx.profile_enable(True); ctx.profile_read(True)
    sim.step(None); torch.cuda.synchronize()
    prof = {k: (v[0], round(v[1], 4)) for k, v in ctx.profile_read(True).items() if v[0]}
    ctx.profile_enable(False)
    print(json.dumps({"n": n, "resident_mode": mode, "coop": os.environ.get("PHIHIP_RESIDENT_COOP", "0"), "ms_per_step": round(ms, 4), "launches_ms": prof}), flush=True)
