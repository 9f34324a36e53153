#!/bin/bash
# Hardware-counter passes over one bench step (each pass its own run: counters + kernel trace only).
#   pass 1  FETCH_SIZE            bytes the L2s fetched from the fabric (HBM + Infinity Cache side), KB
#   pass 2  WRITE_SIZE            bytes written back, KB
#   pass 3  SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F16 GRBM_GUI_ACTIVE
# Writes gpurun_out/pmc_bench/summary.{md,json}: per kernel, per launch.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_bench
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-breakdown --strong-total 0 --batch ${WMI_PMC_BATCH:-32} --tokens ${WMI_PMC_TOKENS:-16}"
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/p1 -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/p1.log 2>&1 || { tail -3 $OUT/p1.log; exit 1; }
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/p2 -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/p2.log 2>&1 || { tail -3 $OUT/p2.log; exit 1; }
timeout -s KILL 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F16 GRBM_GUI_ACTIVE --kernel-trace -d $OUT/p3 -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/p3.log 2>&1 || { tail -3 $OUT/p3.log; exit 1; }
python3 $GRAFT_REPO_ROOT/scripts/pmc_summarize.py $OUT
