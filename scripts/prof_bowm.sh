#!/bin/bash
# SQ counters of the BoW kernels on scripts/bow_perf.py (8192 pairs per launch, three pair mixes).  Usage through gpurun: bash scripts/prof_bowm.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_bowm
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 scripts/bow_perf.py > $OUT/kt.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 scripts/bow_perf.py > $OUT/sq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/sq2 -- python3 scripts/bow_perf.py > $OUT/sq2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 scripts/bow_perf.py > $OUT/f.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("sq", "sq2", "fetch"):
    for f in glob.glob(f"gpurun_out/prof_bowm/{sub}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "bow" in r["Kernel_Name"]:
                agg["bowm" if "bowm_kernel" in r["Kernel_Name"] else "bowq"][r["Counter_Name"]].append(float(r["Counter_Value"]) / (int(r["Grid_Size"]) // 256))
for kn, d in agg.items():
    print(kn, "(per pair, mean over launches)", {k: round(sum(v) / len(v), 1) for k, v in sorted(d.items())})
f = glob.glob("gpurun_out/prof_bowm/kt/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "bow" in r["Name"]:
        print(r["Name"][:60], r["Calls"], "avg_us", float(r["AverageNs"]) / 1e3, "min", float(r["MinNs"]) / 1e3)
PY
