#!/bin/bash
# PMC passes over the windowed path at one th (ORBM_TH).  Usage: ORBM_TH=3 bash scripts/prof_window_pmc.sh [n_inst] [n_sets]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_winpmc
rm -rf $OUT && mkdir -p $OUT
NI=${1:-256}; NS=${2:-64}
export ORBM_TH=${ORBM_TH:-3}
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 scripts/proj_perf.py $NI $NS > $OUT/sq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM --output-format csv -d $OUT/sq2 -- python3 scripts/proj_perf.py $NI $NS > $OUT/sq2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 scripts/proj_perf.py $NI $NS > $OUT/f.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 scripts/proj_perf.py $NI $NS > $OUT/w.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("sq", "sq2", "fetch", "write"):
    for f in glob.glob(f"gpurun_out/prof_winpmc/{sub}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            kn = "collect" if "collect" in r["Kernel_Name"] else ("claims" if "window_kernel" in r["Kernel_Name"] else None)
            if kn: agg[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kn, d in agg.items():
    print(kn, {k: round(sum(v) / len(v), 1) for k, v in sorted(d.items())})
PY
