#!/bin/bash
# Profiles the default bench.py run on the GPU box: kernel trace + stats, then two separate PMC passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md, rocprofv3 PMC slots).
# Usage (from the repo root, through gpurun): bash scripts/profile_bench.sh r01
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
PART=${2:-all}                  # all | a (kernel trace + traffic passes) | b (instruction counter passes): two gpurun calls fit the time limit
mkdir -p $OUT                   # (delete the local gpurun_out/prof_<tag> before a re-run: results are merged into it)
# counter passes of the secondary kernels: a smaller store (the triangle extras of the full store take minutes under counter collection)
ARGS="bench.py --steps 4 --warmup 1 --keyframes 640 --no-cpu-baseline"
# HBM traffic of the headline kernel: the store the bench times (4500 keyframes), no extras
ARGS_FULL="bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline"
( while true; do sleep 50; echo "[profile_bench] $(date +%T) still running"; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
if [ "$PART" != "b" ]; then
# the kernel trace is of the DEFAULT command (python bench.py), the one the bench line comes from
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py > $OUT/kt.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_full -- python3 $ARGS_FULL > $OUT/fetch_full.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_full -- python3 $ARGS_FULL > $OUT/write_full.log 2>&1
fi
if [ "$PART" != "a" ]; then
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 $ARGS > $OUT/sq.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq2 -- python3 $ARGS > $OUT/sq2.log 2>&1
(timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F8 --output-format csv -d $OUT/sq3 -- python3 $ARGS > $OUT/sq3.log 2>&1) || echo "sq3 pass failed (counter names)"
fi
tail -n 2 $OUT/kt.log || true
ls -R $OUT | head -40
