#!/bin/bash
# Round-3 GPU session M: the step captured in a hipGraph (test + timing per size)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r3m; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_graph.py -m gpu -q -p no:cacheprovider -s > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log | cut -c1-300
for SPEC in "128,128 1" "512,512 1" "512,512 8" "64,64,64 1" "128,128,128 1" "256,256,256 1"; do
  set -- $SPEC
  timeout 300 python tools/graph_step.py --res $1 --batch $2 --iters 100 --bc 1 >> $O/graph_step.jsonl 2>> $O/err.log
done
timeout 300 python tools/graph_step.py --res 256,256,256 --batch 1 --iters 100 --bc 0 >> $O/graph_step.jsonl 2>> $O/err.log
cat $O/graph_step.jsonl; tail -5 $O/err.log
