#!/bin/bash
# kernel and copy time inside the single-instance tracking calls of scripts/track_perf.py. Usage through gpurun: bash scripts/trace_track.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_track
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/kt -- python3 scripts/track_perf.py > $OUT/kt.log 2>&1 || { tail -5 $OUT/kt.log; exit 1; }
grep "per call" $OUT/kt.log
python3 - <<PY
import csv, glob
for pat in ("*_kernel_stats.csv", "*_memory_copy_stats.csv"):
    for f in glob.glob("$OUT/kt/*/" + pat):
        for r in csv.DictReader(open(f)):
            print("%-100s calls=%s avg_us=%.1f min_us=%.1f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
