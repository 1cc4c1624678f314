#!/bin/bash
# rocprofv3 --hip-trace --kernel-trace of any command; prints the HIP API calls that took longest (host-side stalls): bash tools/prof_hip_api.sh OUTDIR command...
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/$1"; shift
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
( cd "$REPO" && rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d "$OUT" -o run -- "$@" ) > "$OUT.log" 2>&1
python - "$OUT" <<'PY'
import csv,glob,sys,collections
for f in glob.glob(sys.argv[1]+'/**/*hip_api_trace.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(lambda:[0,0.0,0.0])
    for r in rows:
        d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
        a=agg[r['Function']]; a[0]+=1; a[1]+=d; a[2]=max(a[2],d)
    for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:14]: print(f"{k:40s} calls {v[0]:6d} total_ms {v[1]/1e3:9.2f} max_us {v[2]:10.1f}")
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    t0=int(rows[0]['Start_Timestamp'])
    print('calls longer than 1 ms in the last 40 % of the trace:')
    tend=int(rows[-1]['End_Timestamp'])
    for r in rows:
        s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
        if e-s>1e6 and s-t0>0.6*(tend-t0): print(f"  {(s-t0)/1e6:10.2f} ms {r['Function']:34s} {(e-s)/1e6:8.2f} ms")
PY
