#!/bin/bash
# rocprofv3 kernel trace + stats of one bench step; prints the per-kernel summary sorted by total time.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${1:-x}
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-breakdown --strong-total 0 ${@:2} > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over 2 steps")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {int(r["Calls"]):6d} calls {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:110]}')
PY
tail -1 $OUT/run.log | cut -c1-200
