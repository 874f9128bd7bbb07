#!/bin/bash
# HIP API + kernel time of the C++ drop-in test (per-call overheads). Usage through gpurun: bash scripts/trace_class.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_class
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $OUT/t -- tests/cpp/test_orbmatcher > $OUT/t.log 2>&1 || exit 1
python3 - <<PY
import csv, glob
for pat in ("hip_api_stats", "kernel_stats"):
    for f in glob.glob("$OUT/t/*/*_%s.csv" % pat):
        print("==", pat)
        for r in csv.DictReader(open(f)):
            print("%-70s calls=%6s avg_us=%9.1f total_ms=%8.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
