#!/bin/bash
# texture-address / L1 counters of the BoW kernel (scripts/bow_perf.py). Usage through gpurun: bash scripts/prof_bow_ta.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_bow_ta
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 60 rocprofv3 -L > $OUT/list.txt 2>&1
grep -o "TA_[A-Z0-9_]*\|TCP_[A-Z0-9_]*" $OUT/list.txt | sort -u | tr '\n' ' ' | cut -c1-3000
echo
timeout -s KILL 120 rocprofv3 --pmc TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE --output-format csv -d $OUT/ta -- python3 scripts/bow_perf.py > $OUT/ta.log 2>&1 || { tail -3 $OUT/ta.log; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/ta/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "bowq" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print("%-24s mean %14.0f max %14.0f (x%d)" % (k, sum(v) / len(v), max(v), len(v)))
PY
