#!/bin/bash
# Two SQ counter passes (8 counters each, counters + kernel trace only) over one small bench step; per-kernel sums.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-breakdown --strong-total 0 --batch ${WMI_PMC_BATCH:-128} --tokens ${WMI_PMC_TOKENS:-4}"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_COEXEC_CYCLES --kernel-trace -d $OUT/p1 -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/p1.log 2>&1 || { tail -3 $OUT/p1.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_ANY --kernel-trace -d $OUT/p2 -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/p2.log 2>&1 || { tail -3 $OUT/p2.log; exit 1; }
python3 - <<PY
import csv, glob, collections, json, re
res = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for p in ("p1", "p2"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True)[0]
    seen = set()
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "enc_attention" if "enc_attention" in n else "gemm256_" + re.search(r"NS_\d+(Epi\w+?)ELi", n).group(1) if ("gemm_nt_kernelILi256ELi256" in n or "gemm_nt8" in n) else "layer_norm" if "layer_norm" in n else "cross_attn" if "dec_cross_attention" in n else None
        if not k: continue
        res[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if p == "p1" and r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
out = {}
for k, c in res.items():
    w = c.get("SQ_WAVE_CYCLES", 0) or 1
    out[k] = {"launches": cnt[k], **{n: round(v / w, 4) for n, v in sorted(c.items())}, "SQ_WAVE_CYCLES_abs": w}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
for k, v in out.items(): print(k, json.dumps(v))
PY
