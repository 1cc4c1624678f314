import sys, os, time, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from phiflow_amd import _capi as C
ctx = C.Context(C.load_default_library(), 0)
dev = torch.device("cuda:0")
for n, B in ((128, 1), (512, 1), (512, 8)):
    grid = C.make_grid(2, C.PHIHIP_F32, B, (n, n), (0, 0), (100.0, 100.0), ((1, 1), (1, 1)))
    rhs = torch.randn(B, n, n); rhs -= rhs.mean(dim=(1, 2), keepdim=True); rhs = rhs.to(dev); x = torch.zeros_like(rhs)
    solve = C.Solve(0.0, 0.0, 400, 0, 0, 0)
    ctx.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), solve, want_info=False); torch.cuda.synchronize()
    x.zero_(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.cg_solve(grid, 0, 1, rhs.data_ptr(), x.data_ptr(), solve, want_info=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(json.dumps({"n": n, "batch": B, "enqueue_us_per_iter": round((t1 - t0) / 400 * 1e6, 2), "total_us_per_iter": round((t2 - t0) / 400 * 1e6, 2)}))
