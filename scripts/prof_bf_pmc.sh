#!/bin/bash
# Counter pass over the brute-force kernel in two forms (17: +-1 operands, 22: sign-free operands): matrix-pipe busy cycles,
# instruction counts and the active-cycle count the clock estimate comes from.  Usage (through gpurun): bash scripts/prof_bf_pmc.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_bf_pmc
mkdir -p $OUT
for v in 17 22; do
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/v$v -- python3 scripts/bf_prof.py $v > $OUT/v$v.log 2>&1
done
python3 - <<'P'
import csv, glob, collections
for v in (17, 22):
    rows = []
    for f in glob.glob(f"gpurun_out/prof_bf_pmc/v{v}/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    acc = collections.defaultdict(list)
    for r in rows:
        if "mfma4r" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    # counters come one row per dispatch and counter: mean over the dispatches
    m = {k: sum(x) / len(x) for k, x in acc.items()}
    print("form", v, {k: round(x) for k, x in m.items()})
    if m:
        simd_cycles = m["GRBM_GUI_ACTIVE"] / 8 * 256 * 4      # the counter is summed over the 8 XCDs
        print("   matrix pipe busy %.3f of SIMD-cycles; VALU instr per MFMA %.2f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles, m["SQ_INSTS_VALU"] / m["SQ_INSTS_MFMA"]))
    kt = []
    for f in glob.glob(f"gpurun_out/prof_bf_pmc/v{v}/**/*kernel_trace.csv", recursive=True):
        kt += [r for r in csv.DictReader(open(f)) if "mfma4r" in r["Kernel_Name"]]
    if kt and m:
        ns = sum(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in kt) / len(kt)
        print("   kernel %.3f ms under the counters, clock estimate %.2f GHz" % (ns / 1e6, m["GRBM_GUI_ACTIVE"] / 8 / ns))
P
