"""Throughput of the resampler in front of the path (run on the GPU box): one hour of audio at 48 / 44.1 kHz in device memory
to 16 kHz frames. Prints one JSON line per rate: audio-seconds per second, the HBM roofline fraction (algorithmic bytes =
PCM in + frames out, 4 B each) and the CPU restatement (numpy, one core) timed on a 60 s sample."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handy_amd import ingest                      # noqa: E402
from oracle import ingest_oracle as io            # noqa: E402

secs = int(os.environ.get("WMI_INGEST_SECS", "3600"))
for hz in (48000, 44100):
    n = secs * hz
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    x = (torch.rand(n, device="cuda", generator=g) - 0.5) * 0.2
    rs = ingest.FrameResampler(hz)
    out = torch.empty((rs.n_frames(n), 480), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream          # torch is imported first here, so both share its HIP runtime's stream
    for _ in range(2):
        rs.process(x, out=out, stream=st)
    torch.cuda.synchronize()
    iters = 10
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        rs.process(x, out=out, stream=st)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(iters)]
    avg = float(np.mean(ms))
    alg = 4.0 * (n + out.numel())
    xs = x[: 60 * hz].cpu().numpy()
    t0 = time.perf_counter(); ref = io.resample_frames(xs, hz); cpu_s = time.perf_counter() - t0
    o60 = torch.empty(ref.shape, dtype=torch.float32, device="cuda")
    rs.process(x[: 60 * hz].contiguous(), out=o60, stream=st)
    err = float(np.abs(o60.cpu().numpy() - ref).max())
    print(json.dumps({
        "metric": f"RTFx (audio-sec/sec), resample {hz} Hz -> 16 kHz frames", "value": secs / (avg * 1e-3), "unit": "audio-sec/sec",
        "ms_per_step": avg, "ms_min": float(np.min(ms)), "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{secs} s of {hz} Hz mono f32 PCM resident in HBM -> {out.shape[0]} frames of 480 samples (includes the memset of the output)"},
        "roofline": {"bound": "hbm", "kernel": "resample_blocks_kernel", "achieved": alg / (avg * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": alg / (avg * 1e-3) / 1e9 / 8000.0, "algorithmic_work_per_launch": alg, "avg_launch_ms": avg},
        "cpu_baseline": {"value": 60.0 / cpu_s, "unit": "audio-sec/sec", "cores": 1, "kind": "port", "sample": "60 s of the same input through oracle/ingest_oracle.py (numpy float64 FFTs)"},
        "max_abs_err_vs_restatement_60s": err}), flush=True)
