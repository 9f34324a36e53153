#!/bin/bash
# FETCH_SIZE and WRITE_SIZE of the log-mel kernel (each counter in its own rocprofv3 --pmc run, counters + kernel trace only)
# over bench.py --config mel80 / mel128 (120 x 30 s of PCM per launch); prints bytes per launch next to the algorithmic ones.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_mel
rm -rf $OUT; mkdir -p $OUT
for cfg in mel80 mel128; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $c --kernel-trace -d $OUT/${cfg}_$c -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-breakdown > $OUT/${cfg}_$c.log 2>&1 || { tail -3 $OUT/${cfg}_$c.log; exit 1; }
  done
done
# instruction-side counters of the same launches (one pass, eight SQ counters): what bounds the kernel when the bytes do not
for cfg in mel80 mel128; do
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/${cfg}_SQ -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-breakdown > $OUT/${cfg}_SQ.log 2>&1 || { tail -3 $OUT/${cfg}_SQ.log; exit 1; }
done
python3 - <<PY
import csv, glob, json
out = {}
for cfg, n_mel in (("mel80", 80), ("mel128", 128)):
    r = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob("$OUT/%s_%s/**/*counter_collection.csv" % (cfg, c), recursive=True)[0]
        tot = n = 0
        for row in csv.DictReader(open(f)):
            if "mel_raw_kernel" in row["Kernel_Name"] and row["Counter_Name"] == c:
                tot += float(row["Counter_Value"]); n += 1
        # both counters are in KB; on gfx950 FETCH_SIZE tallies 128-byte requests as 64 bytes: doubled
        r[c] = {"launches": n, "MB_per_launch": round(tot / max(n, 1) * 1024 / 1e6 * (2 if c == "FETCH_SIZE" else 1), 2)}
    sq = {}
    f = glob.glob("$OUT/%s_SQ/**/*counter_collection.csv" % cfg, recursive=True)
    if f:
        n = 0
        for row in csv.DictReader(open(f[0])):
            if "mel_raw_kernel" in row["Kernel_Name"]:
                sq[row["Counter_Name"]] = sq.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                n += row["Counter_Name"] == "SQ_WAVE_CYCLES"
        w = sq.get("SQ_WAVE_CYCLES", 0) or 1
        # SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles summed over waves; the fractions are per wave-cycle
        r["sq"] = {"launches": n, **{k: round(v / w, 4) for k, v in sorted(sq.items()) if k != "SQ_WAVE_CYCLES"}, "SQ_INSTS_VALU_per_launch": sq.get("SQ_INSTS_VALU", 0) / max(n, 1)}
    r["algorithmic_MB_per_launch"] = {"pcm_in": round(120 * 480000 * 4 / 1e6, 2), "logmel_out": round(120 * n_mel * 3000 * 4 / 1e6, 2)}
    out[cfg] = r
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
