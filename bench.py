#!/usr/bin/env python3
"""Headline benchmark: RTFx (audio-seconds per wall-second) of Whisper large-v3-turbo batched
transcription on MI355X, one independent engine per GPU, data-parallel by utterance.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one pass of the whole hot path (log-mel -> encoder -> cross-K/V -> greedy decode) over one batch
of synthetic 30 s / 16 kHz utterances whose PCM is already resident in HBM. Weights are a seeded synthetic
ggml file of the exact large-v3-turbo architecture (no checkpoint can be fetched here); with content-free
audio and weights the natural transcript length is meaningless, so every utterance decodes exactly
--tokens tokens (end-of-text masked before, forced at, that position).

Rank 0 prints ONE JSON line carrying, besides the contract fields,
  roofline     -- the dominant kernel class (encoder GEMMs, MFMA-bound): algorithmic flop / device time,
                  measured with HIP events on the engine's stream inside the timed region;
  cpu_baseline -- the CPU oracle (a port, not whisper.cpp) timed on this box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0   # MI355X dense fp16/bf16 MFMA peak (vendor figure without sparsity)


def model_file(arch: str, seed: int = 0) -> str:
    from handy_amd import ggml_file
    d = os.environ.get("WMI_MODEL_DIR", "/tmp/wmi_models")
    os.makedirs(d, exist_ok=True)
    real = os.path.join(d, {"large-v3-turbo": "ggml-large-v3-turbo.bin"}.get(arch, f"ggml-{arch}.bin"))
    if os.path.exists(real):
        return real
    p = os.path.join(d, f"synthetic-{arch}-s{seed}.bin")
    if not os.path.exists(p):
        tmp = p + f".{os.getpid()}.tmp"
        ggml_file.write_synthetic_model(tmp, arch, seed=seed, decisive=2.0)
        os.replace(tmp, p)
    return p


def pmc_traffic():
    """HBM-side bytes per launch of the dominant kernel from the newest committed counter summary
    (profiles/rNN_pmc_summary.json, written by scripts/pmc_bench.sh: separate rocprofv3 --pmc passes, FETCH_SIZE
    doubled for gfx950, WRITE_SIZE as read). Counters cannot be collected from inside this process."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return None, None
    rows = [r for r in json.load(open(files[-1])) if r["kernel"].startswith("gemm_nt<256x256") and r.get("fetch_MB") is not None and r.get("write_MB") is not None
            and not r["kernel"].endswith("EpiF32>")]
    n = sum(r["launches"] for r in rows)
    if not n:
        return None, None
    return sum((r["fetch_MB"] + r["write_MB"]) * 1e6 * r["launches"] for r in rows) / n, os.path.basename(files[-1])


def cpu_baseline(path: str, tokens: int, budget_s: float = 30.0):
    """The oracle (kind 'port': this repo's CPU restatement, not whisper.cpp) on one 30 s utterance:
    log-mel + encoder on all host threads it is given, then as many forced decode steps as fit the budget."""
    from handy_amd.synth import synth_pcm
    from oracle import oracle as wo
    wo.build()
    om = wo.OracleModel(path)
    st = wo.OracleState(om)
    pcm = synth_pcm(0)
    t0 = time.time()
    st.pcm_to_mel(pcm)
    t_mel = time.time() - t0
    t0 = time.time()
    st.encode(0)
    t_enc = time.time() - t0
    sot = om.special("sot")
    prompt = [sot, sot + 1, om.special("transcribe"), om.special("not")] if om.n_vocab >= 51865 else [sot, om.special("not")]
    t0 = time.time()
    st.decode(prompt, 0)
    n_done, pos = 0, len(prompt)
    while n_done < tokens and (time.time() - t0) < max(2.0, budget_s - t_enc - t_mel):
        st.decode([1000 + n_done], pos)
        pos += 1
        n_done += 1
    t_dec_measured = time.time() - t0
    t_dec = t_dec_measured * (tokens / max(n_done, 1))
    total = t_mel + t_enc + t_dec
    cores = wo.lib().wo_n_threads()
    st.close(); om.close()
    return {"value": 30.0 / total, "unit": "audio-sec/sec", "cores": cores, "kind": "port",
            "sample": f"1 utterance x 30 s: mel {t_mel:.2f}s + encoder {t_enc:.2f}s measured, decode {n_done}/{tokens} tokens measured "
                      f"({t_dec_measured:.2f}s) and scaled to {tokens}; OpenMP threads={cores}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("WMI_BENCH_BATCH", "1024")), help="utterances per GPU per step")
    ap.add_argument("--tokens", type=int, default=128, help="decoded tokens per utterance")
    ap.add_argument("--arch", default="large-v3-turbo")
    ap.add_argument("--enc-chunk", type=int, default=int(os.environ.get("WMI_ENC_CHUNK", "128")), help="utterances per encoder launch set")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("WMI_BENCH_STREAMS", "1")),
                    help="engine states per GPU, each on its own HIP stream and host thread, each transcribing --batch utterances per step")
    ap.add_argument("--pipeline-groups", type=int, default=int(os.environ.get("WMI_PIPELINE_GROUPS", "1")),
                    help="index ranges of the batch that decode one after the other while the encoder works on the later ones")
    ap.add_argument("--encoder-lanes", type=int, default=int(os.environ.get("WMI_ENCODER_LANES", "1")), help="2: alternate encoder chunks between two streams")
    ap.add_argument("--dist-backend", default=os.environ.get("WMI_DIST_BACKEND", "nccl"),
                    help="torch.distributed backend of the result gather; 'gloo' together with --all-ranks-on-device0 rehearses the multi-rank path on one GPU")
    ap.add_argument("--all-ranks-on-device0", action="store_true", help="rehearsal only: every rank uses GPU 0")
    ap.add_argument("--host-pcm", action="store_true",
                    help="hand the PCM over as host buffers every step (the PCIe-inclusive rate; never the contract's value)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true", help="skip the extra untimed step that profiles every kernel class")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU path")
    dev_index = 0 if args.all_ranks_on_device0 else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)
    gather_dev = dev if args.dist_backend == "nccl" else None

    from handy_amd import whisper_rs as wr
    from handy_amd.synth import synth_pcm

    # rank 0 writes the synthetic model file once; the others wait for it
    if rank == 0:
        path = model_file(args.arch)
    if world > 1:
        dist.barrier()
    path = model_file(args.arch)

    ctx = wr.WhisperContext(path, wr.WhisperContextParameters(dev_index))
    B = args.batch
    S = max(1, args.streams)
    states = [ctx.create_batch_state(B, 1, args.enc_chunk) for _ in range(S)]
    for s_ in states:
        s_.set_force_tokens(args.tokens)
        s_.set_pipeline_groups(args.pipeline_groups)
        s_.set_encoder_lanes(args.encoder_lanes)
    st = states[0]
    fp = wr.FullParams(wr.SamplingStrategy.Greedy(1))
    fp.set_temperature_inc(0.0)
    fp.set_no_timestamps(True)
    fp.set_language("en")

    # synthetic PCM resident in HBM: a pool of 16 distinct utterances per rank, tiled to the batch
    n_samp = 30 * 16000
    pool = np.stack([synth_pcm(rank * 1000 + i, n_samp) for i in range(min(16, B))])
    pcm_dev = torch.from_numpy(pool).to(dev)
    pcm_dev = pcm_dev.repeat((B + pool.shape[0] - 1) // pool.shape[0], 1)[:B].contiguous()
    ptrs = [pcm_dev.data_ptr() + i * n_samp * 4 for i in range(B)]
    ns = [n_samp] * B
    host_pcm = [pool[i % pool.shape[0]] for i in range(B)] if args.host_pcm else None

    from handy_amd.node import gather_token_records, pack_token_records

    import threading

    def step(gather=True):
        # S engine states run the same hot path concurrently (own stream, own host thread; ctypes drops the GIL during
        # the call): while one state's decoder streams K/V and weights from HBM, another state's encoder keeps the MFMAs busy
        if host_pcm is not None:
            st.full_batch(fp, host_pcm)
        elif S == 1:
            st.full_batch_device(fp, ptrs, ns)
        else:
            errs = []
            def run(s_):
                try:
                    s_.full_batch_device(fp, ptrs, ns)
                except Exception as e:      # surfaced after the join
                    errs.append(e)
            th = [threading.Thread(target=run, args=(s_,)) for s_ in states]
            for t in th: t.start()
            for t in th: t.join()
            if errs:
                raise errs[0]
        recs = [pack_token_records(range((rank * S + k) * B, (rank * S + k + 1) * B), [s_.batch_tokens(u) for u in range(B)], args.tokens + 2)
                for k, s_ in enumerate(states)]
        rec = np.concatenate(recs, axis=0)
        if world > 1 and gather:       # the only collective on the path: gather the per-utterance token records on rank 0 (RCCL)
            gather_token_records(rec, B * S, device=gather_dev)
        return rec[:, 2:]

    for _ in range(args.warmup):
        toks = step()
    # HIP events inside the timed region only around the dominant kernel class (the encoder GEMMs): an event pair
    # around each of the ~50 small decoder launches of every decode pass would itself slow the step down
    st.set_profiling_classes(["enc_gemm"])
    st.reset_kernel_stats()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        toks = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    stats = st.kernel_stats()
    st.set_profiling(False)
    breakdown = None
    if rank == 0 and not args.no_breakdown:      # one extra, untimed step with every class profiled
        st.set_profiling(True)
        st.reset_kernel_stats()
        t1 = time.perf_counter()
        step(gather=False)          # rank 0 alone: no collective here
        torch.cuda.synchronize()
        bd_ms = (time.perf_counter() - t1) * 1e3
        breakdown = st.kernel_stats()
        st.set_profiling(False)
    n_tok = int((toks >= 0).sum(axis=1).min())
    assert n_tok == args.tokens + 1, f"decode produced {n_tok} tokens per utterance, expected {args.tokens + 1}"

    if world > 1:
        tmax = torch.tensor([dt], device=dev if gather_dev is not None else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank == 0:
        audio_s = 30.0 * B * S * world * args.steps
        g = stats["enc_gemm"]
        achieved = g["work"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
        per_class = None
        if breakdown is not None:
            per_class = {k: {"ms": v["ms"], "launches": v["launches"], "rate": (v["work"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 else 0.0}
                         for k, v in breakdown.items()}
            per_class["_note"] = f"one extra untimed step with every kernel class under HIP events ({bd_ms:.1f} ms wall); rate in T(flop|B)/s"
        traffic, traffic_src = pmc_traffic()
        line = {
            "metric": "RTFx (audio-sec/sec) whole node, Whisper large-v3-turbo", "value": audio_s / dt, "unit": "audio-sec/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"Whisper {args.arch} fp16 (synthetic ggml weights of the exact architecture), {B * S} x 30 s utterances per GPU per step"
                                   f"{'' if S == 1 else f' ({S} concurrent engine states x {B})'}, "
                                   f"greedy, no_timestamps, {args.tokens} forced tokens per utterance, "
                                   f"{'PCM handed over as host buffers every step (PCIe-inclusive)' if args.host_pcm else 'PCM resident in HBM'}",
                       "utterances_per_gpu": B * S, "engine_states_per_gpu": S, "pipeline_groups": args.pipeline_groups, "encoder_lanes": args.encoder_lanes, "tokens_per_utterance": args.tokens, "enc_chunk": args.enc_chunk, "parallelism": f"dp{world} by utterance"},
            "roofline": {"bound": "mfma", "kernel": "gemm_nt_kernel (encoder conv/QKV/out/MLP/cross-KV GEMMs)", "achieved": achieved,
                         "peak": MFMA_F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_F16_DENSE_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_note": f"bytes per launch at the L2-fabric boundary (HBM + Infinity Cache), mean over the class, from profiles/{traffic_src}" if traffic else None,
                         "algorithmic_flop_per_launch": g["work"] / max(g["launches"], 1),
                         "avg_launch_ms": g["ms"] / max(g["launches"], 1), "launches": g["launches"]},
            "kernel_classes": per_class,
        }
        if not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(path, args.tokens)
            except Exception as e:      # the baseline is reported, never required for the GPU number
                line["cpu_baseline"] = {"value": None, "unit": "audio-sec/sec", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(line))
    for s_ in states:
        s_.free()
    ctx.free()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
