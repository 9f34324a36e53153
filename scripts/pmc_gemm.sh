#!/bin/bash
# Counter passes over one GEMM variant (each pass its own run; counters only, no trace domains besides kernel-trace).
set -e
cd /tmp && export TMPDIR=/tmp
V=${1:-15}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm_v$V
mkdir -p $OUT
i=0
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F16 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $C --kernel-trace -d $OUT/p$i -o p$i --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/gemm_one.py $V 24000 5120 1280 > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if 'gemm' in r['Kernel_Name']:
            a = agg[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    for k, (s, n) in agg.items():
        print(f"{k:32s} per-launch {s / n:16.1f}  (n={n})")
PY
