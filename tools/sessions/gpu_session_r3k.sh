#!/bin/bash
# Round-3 GPU session K: implicit diffusion (tests + timing)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r3k; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "implicit or reference_style" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for B in 0 1 2; do timeout 300 python tools/time_diffuse_implicit.py --size 256 --bc $B >> $O/time_implicit.jsonl 2>> $O/err.log; done
timeout 300 python tools/time_diffuse_implicit.py --size 256 --bc 1 --dtype f64 >> $O/time_implicit.jsonl 2>> $O/err.log
cat $O/time_implicit.jsonl
