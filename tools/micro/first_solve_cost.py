import json, math, os, sys, time
import torch
sys.path.insert(0, os.getcwd())
from phiflow_amd import _capi as C
dev = torch.device("cuda:0"); lib = C.load_default_library(); L = 2 * math.pi
for n, dt, td in ((512, C.PHIHIP_F32, torch.float32), (384, C.PHIHIP_F64, torch.float64), (320, C.PHIHIP_F32, torch.float32)):
    grid = C.make_grid(3, dt, 1, (n, n, n), (0, 0, 0), (L, L, L), ((0, 0),) * 3)
    rhs = torch.randn(1, n, n, n, device=dev, dtype=td); rhs -= rhs.mean(); x = torch.zeros_like(rhs)
    for k in (1, 12, 1, 12):
        for tune in (True, False):
            ctx = C.Context(lib, 0); ctx.workspace_placement(k); ctx.set_autotune(tune)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, 10, 0, 0, 0), want_info=False)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            ctx.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, 10, 0, 0, 0), want_info=False)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            print(json.dumps({"size": n, "dtype": str(td), "candidates": k, "autotune": tune, "first_solve_ms": round((t1 - t0) * 1e3, 1), "second_solve_ms": round((t2 - t1) * 1e3, 1), "placement": ctx.workspace_placement()}), flush=True)
            ctx.close()
