#!/bin/bash
# same-box A/B of the plan of the residual / apply kernels (family 0): APPLY throughput per size and the RESID launches of a solve
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out; export TMPDIR=/tmp
for ROUND in 1 2; do
 for LIB in phiflow_amd/lib/libphihip_prev.so ""; do
  python - "$LIB" <<'PY'
import sys, math, json, torch
sys.path.insert(0, '.')
from phiflow_amd import _capi as C
lib = C.Library(sys.argv[1], strict=False) if sys.argv[1] else C.load_default_library()
ctx = C.Context(lib, 0)
dev = torch.device('cuda:0'); L = 2 * math.pi
for n in (192, 256, 320, 384, 512):
    grid = C.make_grid(3, C.PHIHIP_F32, 1, (n, n, n), (0, 0, 0), (L, L, L), ((0, 0),) * 3)
    a = torch.randn(1, n, n, n, device=dev); a -= a.mean(); c = torch.empty_like(a); x = torch.zeros_like(a)
    def timed(fn, reps=30):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    t = timed(lambda: ctx.laplace_apply(grid, 0, 1, a.data_ptr(), c.data_ptr()))
    ctx.cg_solve(grid, 0, 1, a.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, 3, 0, 0, 0), want_info=False); torch.cuda.synchronize(); x.zero_()
    ctx.profile_enable(True); ctx.profile_read(reset=True)
    ctx.cg_solve(grid, 0, 1, a.data_ptr(), x.data_ptr(), C.Solve(0.0, 0.0, 100, 50, 0, 0), want_info=False); torch.cuda.synchronize()
    prof = ctx.profile_read(reset=True); ctx.profile_enable(False)
    print(json.dumps({"lib": sys.argv[1][-12:] or "default", "size": n, "apply_us": round(t * 1e3, 1), "apply_GBs": round(8 * n ** 3 / t / 1e6), "resid_us": round(prof["cg_residual"][1] / prof["cg_residual"][0] * 1e3, 1),
                      "resid_launches": prof["cg_residual"][0], "plan0": list(ctx.query_plan(grid, False, 0).values())[:4]}), flush=True)
PY
 done
done
