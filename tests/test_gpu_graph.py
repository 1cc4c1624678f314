"""
`-m gpu`: the step under hipGraph capture. The library must be capture-safe when the caller asks for no host read-back (info = NULL,
check_every = 0): no allocation, no synchronisation, no first-call autotune inside the capture (cg.hip: stream_is_capturing). A captured pair of
steps replayed N times must leave exactly the bits that 2 N eager steps leave.
"""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("res,batch,bc,dtype", [((64, 64), 2, 1, "f32"), ((96, 128), 1, 0, "f64"), ((32, 32, 32), 1, 1, "f32"), ((48, 40, 64), 2, 2, "f32")])
def test_captured_step_replays_bit_identically(gpu_backend, res, batch, bc, dtype):
    import graph_step
    rec = graph_step.run(gpu_backend.ctx, res, batch, 20, bc, dtype, 3, torch.device(str(gpu_backend.device)))
    print(rec)
    assert rec["finite"] and rec["bit_identical"], rec
