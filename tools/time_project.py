import sys, math, json, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phiflow_amd import _capi as C
lib = sys.argv[1] if len(sys.argv) > 1 else ''
ctx = C.Context(C.Library(lib, strict=False) if lib else C.load_default_library(), 0)
n=256; dev=torch.device('cuda:0'); L=2*math.pi
grid = C.make_grid(3, C.PHIHIP_F32, 1, (n,n,n), (0,0,0), (L,L,L), ((0,0),)*3)
v=[torch.randn(1,n,n,n,device=dev) for _ in range(3)]; p=torch.randn(1,n,n,n,device=dev); div=torch.empty_like(p)
P=lambda ts:[t.data_ptr() for t in ts]
def timed(fn, reps=30):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
print(json.dumps({"lib": lib or "default", "ms_grad_subtract": round(timed(lambda: ctx.grad_subtract(grid,0,1,p.data_ptr(),P(v))),5), "ms_divergence_balance": round(timed(lambda: ctx.divergence(grid,P(v),0,1,True,div.data_ptr())),5)}))
