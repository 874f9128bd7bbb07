#!/bin/bash
# Config 3 probe + kernel trace of the windowed path (collect_kernel / window_kernel split).  Usage: bash scripts/prof_window2.sh [n_inst] [n_sets]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_win
rm -rf $OUT && mkdir -p $OUT
NI=${1:-256}; NS=${2:-64}
timeout -k 10 200 python3 scripts/proj_perf.py $NI $NS > $OUT/perf.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 scripts/proj_perf.py $NI $NS > $OUT/kt.log 2>&1 || exit 1
cat $OUT/perf.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_win/kt/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "collect" in r["Name"] or "window_kernel" in r["Name"]:
        print(r["Name"][:70], r["Calls"], "avg_us", float(r["AverageNs"]) / 1e3, "min", float(r["MinNs"]) / 1e3, "max", float(r["MaxNs"]) / 1e3)
PY
