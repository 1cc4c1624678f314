#!/bin/bash
# Round-5 GPU session R: jit_compile (hipGraph capture / replay of a phi-level step) -- the GPU tests and the phi_level block of the bench line
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"
O=gpurun_out/${SESSION_TAG:-r5r}; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_jit.py tests/test_gpu_graph.py -m gpu -q -p no:cacheprovider -x > $O/pytest_jit.log 2>&1; echo "jit rc=$?"; tail -25 $O/pytest_jit.log
timeout 600 python bench.py --steps 10 --warmup 3 --pmc 0 --cpu-size 0 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -5 $O/bench.err
python -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(json.dumps(d['phi_level'])[:1500])"
echo finished
