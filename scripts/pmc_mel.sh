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
    r["algorithmic_MB_per_launch"] = {"pcm_in": round(120 * 480000 * 4 / 1e6, 2), "logmel_out": round(120 * n_mel * 3000 * 4 / 1e6, 2)}
    out[cfg] = r
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
