#!/bin/bash
# SQ counter passes over the windowed kernel (config 3 probe). Usage through gpurun: bash scripts/profile_window.sh TH TAG
set -e
TH=${1:-3}
TAG=${2:-win}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/sq1 -- python3 scripts/proj_prof.py $TH > $OUT/sq1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD --output-format csv -d $OUT/sq2 -- python3 scripts/proj_prof.py $TH > $OUT/sq2.log 2>&1
python3 - <<PY
import csv, glob, collections
for sub in ("sq1", "sq2"):
    f = glob.glob("$OUT/%s/*/*_counter_collection.csv" % sub)[0]
    for kern in ("collect_kernel", "window_kernel"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            print("%-16s %-24s %14.0f  (x%d)" % (kern, k, sum(v) / len(v), len(v)))
PY
