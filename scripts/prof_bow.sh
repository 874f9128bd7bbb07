#!/bin/bash
# SQ counters of the BoW kernel inside the bench's extras. Usage through gpurun: bash scripts/prof_bow.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_bow
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/sq.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/sq/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "bow" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print("%-42s %-24s %14.0f (x%d)" % (k[0], k[1], sum(v) / len(v), len(v)))
PY
