#!/bin/bash
# per-kernel time of the windowed search (config 3 probe). Usage through gpurun: bash scripts/trace_window.sh TH
TH=${1:-3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_win_$TH
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 scripts/proj_prof.py $TH ${2:-512} > $OUT/kt.log 2>&1 || exit 1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/kt/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print("%-90s calls=%s avg_us=%.1f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
