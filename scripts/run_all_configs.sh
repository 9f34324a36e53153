#!/bin/bash
# Every BASELINE workload through bench.py, one JSON line each (run on the GPU box): gpurun_out/configs.jsonl
set -e
OUT=gpurun_out/configs.jsonl
rm -f $OUT
for cfg in headline small32 largev3_beam5 tiny1 mel80 mel128; do
  timeout -k 10 900 python bench.py --config $cfg >> $OUT 2>> gpurun_out/configs.err
done
timeout -k 10 300 python scripts/clip_latency_probe.py tiny.en small large-v3-turbo > gpurun_out/clip_latency.log 2>&1
