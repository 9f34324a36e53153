#!/usr/bin/env python3
"""Benchmark of the Whisper hot path on MI355X: RTFx (audio-seconds per wall-second), one independent engine per GPU,
data-parallel by utterance.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one pass of the whole hot path (log-mel -> encoder -> cross-K/V -> decode) over one batch of synthetic 30 s /
16 kHz utterances, every one different, whose PCM is already resident in HBM. Weights are a seeded synthetic ggml file of
the exact architecture (no checkpoint can be fetched here); with content-free audio and weights the natural transcript
length is meaningless, so every utterance decodes exactly --tokens tokens (end-of-text masked before, forced at, that
position).

--config picks the workload (default: the headline one, Whisper large-v3-turbo, 1024 utterances per GPU per step):
  headline        large-v3-turbo, greedy, 128 tokens                    (the 10k x 30 s job run as batches; weak scaling)
  small32         Whisper small, 32 x 30 s, greedy, 128 tokens
  largev3_beam5   Whisper large-v3, 64 x 30 s, beam size 5, 64 tokens
  turbo_beam3     large-v3-turbo, 256 x 30 s, beam size 3, 64 tokens     (the headline model under the beam width Handy's
                  transcribe-rs engine defaults to; not a BASELINE workload)
  tiny1           Whisper tiny.en, one 30 s clip, greedy, 64 tokens      (latency of the call Handy makes)
  mel80 / mel128  log-mel frontend only, 1 h of audio (120 x 30 s)

Rank 0 prints ONE JSON line carrying, besides the contract fields,
  roofline      the dominant kernel class of the workload: algorithmic flop or bytes / device time, measured with HIP events
                on the engine's stream inside the timed region;
  cpu_baseline  the CPU oracle (a port, not whisper.cpp) timed on this box's host cores on a bounded sample;
  strong_scaling (headline only) a fixed job of --strong-total utterances pulled by the ranks from a shared queue of batches,
                token records gathered once at the end: the same engine, strong scaling beside the weak-scaling `value`;
  host_pcm      (headline, rank 0) one extra step with the PCM handed over as host buffers: the PCIe-inclusive rate.
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
HBM_PEAK_GBPS = 8000.0                # MI355X HBM3E peak
N_SAMP = 30 * 16000

CONFIGS = {
    #                 arch              utterances tokens beam  dominant kernel class (what `roofline` describes)
    "headline":      ("large-v3-turbo", 1024,      128,   1,    "enc_gemm"),
    "small32":       ("small",          32,        128,   1,    "dec_gemm"),
    "largev3_beam5": ("large-v3",       64,        64,    5,    "dec_cross_attn"),
    "turbo_beam3":   ("large-v3-turbo", 256,       64,    3,    "enc_gemm"),
    "tiny1":         ("tiny.en",        1,         64,    1,    "dec_gemm"),
    "mel80":         ("micro.en",       120,       0,     1,    "mel"),
    "mel128":        ("micro-v3",       120,       0,     1,    "mel"),
}
CLASS_KERNEL = {
    "enc_gemm": ("mfma", "gemm_nt8p_kernel / gemm_nt8_kernel / gemm_nt_kernel (encoder conv, QKV, out, MLP, cross-K/V GEMMs)"),
    "enc_attn": ("mfma", "enc_attention_wide_kernel"),
    "dec_cross_attn": ("hbm", "dec_cross_attention_kernel / dec_cross_attention_mq_kernel (cached encoder K/V streamed once per utterance and layer)"),
    "dec_gemm": ("hbm", "gemm_skinny_kernel / gemm_nt_kernel on decoder rows (decoder weights streamed once per pass)"),
    "mel": ("hbm", "mel_raw_kernel + mel_normalize_batch_kernel (PCM in; clamped, scaled fp32 log-mel out)"),
}


def log(msg: str):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


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


def synth_pcm_device(seeds, n, dev, chunk: int = 64):
    """The recipe of handy_amd.synth.synth_pcm (three sinusoids, white noise, a slow envelope, clipped) evaluated on the
    device, one different clip per seed: [len(seeds), n] float32. (Benchmark input only; the parity tests use the host recipe.)"""
    import torch
    out = torch.empty((len(seeds), n), dtype=torch.float32, device=dev)
    t = torch.arange(n, device=dev, dtype=torch.float64) / 16000.0
    for c0 in range(0, len(seeds), chunk):
        sd = seeds[c0:c0 + chunk]
        par = np.zeros((len(sd), 8))
        for i, s in enumerate(sd):
            rng = np.random.default_rng(1000003 * (int(s) + 1))
            par[i, 0:3] = rng.uniform(80.0, 4000.0, 3); par[i, 3:6] = rng.uniform(0, 2 * np.pi, 3)
            par[i, 6] = rng.uniform(2.0, 5.0); par[i, 7] = rng.uniform(0, 2 * np.pi)
        p = torch.from_numpy(par).to(dev)
        x = torch.zeros((len(sd), n), dtype=torch.float64, device=dev)
        for k in range(3):
            x += 0.05 * torch.sin(2 * np.pi * p[:, k:k + 1] * t[None, :] + p[:, 3 + k:4 + k])
        g = torch.Generator(device=dev); g.manual_seed(7919 * (int(sd[0]) + 1))
        x += 0.02 * torch.randn((len(sd), n), generator=g, device=dev, dtype=torch.float32)
        env = 0.55 + 0.45 * torch.sin(2 * np.pi * p[:, 6:7] * t[None, :] + p[:, 7:8])
        out[c0:c0 + len(sd)] = torch.clamp(x * env, -1.0, 1.0).to(torch.float32)
    return out


def pmc_traffic():
    """HBM-side bytes per launch of the headline's dominant kernel from the newest committed counter summary
    (profiles/rNN_pmc_summary.json, written by scripts/pmc_bench.sh: separate rocprofv3 --pmc passes, FETCH_SIZE
    doubled for gfx950, WRITE_SIZE as read). Counters cannot be collected from inside this process."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return None, None
    rows = [r for r in json.load(open(files[-1])) if r["kernel"].startswith(("gemm_nt<256x256", "gemm_nt8")) and r.get("fetch_MB") is not None
            and r.get("write_MB") is not None and not r["kernel"].endswith("EpiF32>")]
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
    t_dec = t_dec_measured * (tokens / max(n_done, 1)) if tokens else 0.0
    total = t_mel + t_enc + t_dec
    cores = wo.lib().wo_n_threads()
    st.close(); om.close()
    return {"value": 30.0 / total, "unit": "audio-sec/sec", "cores": cores, "kind": "port",
            "sample": f"1 utterance x 30 s: mel {t_mel:.2f}s + encoder {t_enc:.2f}s measured, decode {n_done}/{tokens} tokens measured "
                      f"({t_dec_measured:.2f}s) and scaled to {tokens}; OpenMP threads={cores}"}


def cpu_baseline_mel(path: str, budget_s: float = 15.0):
    from handy_amd.synth import synth_pcm
    from oracle import oracle as wo
    wo.build()
    om = wo.OracleModel(path); st = wo.OracleState(om)
    pcm = synth_pcm(0)
    n, t0 = 0, time.time()
    while time.time() - t0 < budget_s and n < 64:
        st.pcm_to_mel(pcm); n += 1
    dt = time.time() - t0
    cores = wo.lib().wo_n_threads()
    st.close(); om.close()
    return {"value": 30.0 * n / dt, "unit": "audio-sec/sec", "cores": cores, "kind": "port", "sample": f"log-mel of {n} x 30 s clips in {dt:.2f}s; OpenMP threads={cores}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="headline", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=int(os.environ.get("WMI_BENCH_BATCH", "0")), help="utterances per GPU per step (0: the config's)")
    ap.add_argument("--tokens", type=int, default=-1, help="decoded tokens per utterance (-1: the config's)")
    ap.add_argument("--arch", default="", help="override the config's architecture")
    ap.add_argument("--enc-chunk", type=int, default=int(os.environ.get("WMI_ENC_CHUNK", "128")), help="utterances per encoder launch set")
    ap.add_argument("--pipeline-groups", type=int, default=int(os.environ.get("WMI_PIPELINE_GROUPS", "1")),
                    help="index ranges of the batch that decode one after the other while the encoder works on the later ones")
    ap.add_argument("--encoder-lanes", type=int, default=int(os.environ.get("WMI_ENCODER_LANES", "1")), help="2: alternate encoder chunks between two streams")
    ap.add_argument("--dist-backend", default=os.environ.get("WMI_DIST_BACKEND", "nccl"),
                    help="torch.distributed backend of the result gather; 'gloo' together with --all-ranks-on-device0 rehearses the multi-rank path on one GPU")
    ap.add_argument("--all-ranks-on-device0", action="store_true", help="rehearsal only: every rank uses GPU 0")
    ap.add_argument("--host-pcm", action="store_true",
                    help="hand the PCM over as host buffers every step (the PCIe-inclusive rate; never the contract's value)")
    ap.add_argument("--strong-total", type=int, default=int(os.environ.get("WMI_STRONG_TOTAL", "10240")),
                    help="headline only: utterances of the fixed strong-scaling job run after the timed steps (0: skip)")
    ap.add_argument("--strong-batch", type=int, default=256, help="utterances per queue entry of the strong-scaling job")
    ap.add_argument("--result-log", default=os.environ.get("WMI_RESULT_LOG", ""),
                    help="directory of the strong-scaling job's append-only per-batch result log (default: a directory under the temp dir)")
    ap.add_argument("--resume", action="store_true", help="strong-scaling job: skip the batches the result log already holds")
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
    from handy_amd.node import BatchQueue, ResultLog, gather_ragged_records, gather_token_records, pack_token_records

    arch, B, tokens, beam, dom = CONFIGS[args.config]
    arch = args.arch or arch
    B = args.batch or B
    tokens = tokens if args.tokens < 0 else args.tokens
    mel_only = args.config.startswith("mel")
    headline = args.config == "headline"

    # rank 0 writes the synthetic model file once; the others wait for it
    if rank == 0:
        log(f"config {args.config}: {arch}, {B} utterances per GPU per step, {tokens} tokens, beam {beam}; writing/looking up the model file")
        path = model_file(arch)
    if world > 1:
        dist.barrier()
    path = model_file(arch)

    ctx = wr.WhisperContext(path, wr.WhisperContextParameters(dev_index))
    st = ctx.create_batch_state(B, beam, min(args.enc_chunk, B))
    st.set_force_tokens(tokens)
    st.set_pipeline_groups(args.pipeline_groups)
    st.set_encoder_lanes(args.encoder_lanes)
    fp = wr.FullParams(wr.SamplingStrategy.BeamSearch(beam, -1.0) if beam > 1 else wr.SamplingStrategy.Greedy(1))
    fp.set_temperature_inc(0.0)
    fp.set_no_timestamps(True)
    fp.set_language("en")

    # synthetic PCM resident in HBM: B different utterances per rank
    pcm_dev = synth_pcm_device([rank * 100000 + i for i in range(B)], N_SAMP, dev)
    torch.cuda.synchronize()
    ptrs = [pcm_dev.data_ptr() + i * N_SAMP * 4 for i in range(B)]
    ns = [N_SAMP] * B
    host_pcm = None
    if args.host_pcm:
        host_pcm = list(pcm_dev.cpu().numpy())
    if rank == 0:
        log("engine and inputs ready")

    import ctypes as C
    mel_out = torch.empty((B, ctx.n_mels(), 3000), dtype=torch.float32, device=dev) if mel_only else None
    offs_c = (C.c_longlong * B)(*[i * N_SAMP for i in range(B)]); ns_c = (C.c_int * B)(*ns)

    def step(gather=True):
        if mel_only:        # the whole frontend: log-mel, per-utterance maximum, clamp and scale into mel_out (the contract's output)
            assert ctx.L.wmi_mel_batch_device_normalized(ctx.h, st.h, C.c_void_p(pcm_dev.data_ptr()), offs_c, ns_c, B, C.c_void_p(mel_out.data_ptr()), 3000) == 0
            return None
        if host_pcm is not None:
            st.full_batch(fp, host_pcm)
        else:
            st.full_batch_device(fp, ptrs, ns)
        rec = pack_token_records(range(rank * B, (rank + 1) * B), [st.batch_tokens(u) for u in range(B)], tokens + 2)
        if world > 1 and gather:       # the only collective on the path: gather the per-utterance token records on rank 0 (RCCL)
            gather_token_records(rec, B, device=gather_dev)
        return rec[:, 2:]

    for _ in range(args.warmup):
        toks = step()
    # HIP events inside the timed region only around the dominant kernel class: an event pair around each of the ~50 small
    # decoder launches of every decode pass would itself slow the step down
    # (the headline's dominant class is ~1300 launches of 1.5 ms per step; the decoder classes of the small configurations are
    # thousands of 5-10 us launches, whose event pairs would be a third of the step: those configurations take their roofline
    # from a repeat of the timed steps with the events on, and the timed region itself runs without any)
    events_in_timed = headline or mel_only or dom.startswith("enc")
    if events_in_timed:
        st.set_profiling_classes([dom])
    else:
        st.set_profiling(False)
    st.reset_kernel_stats()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    host_t0 = st.host_timings()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        toks = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    host_t = st.host_timings()
    roofline_from = "HIP events around the class inside the timed region"
    if not events_in_timed:
        st.set_profiling_classes([dom])
        st.reset_kernel_stats()
        for _ in range(args.steps):
            step(gather=False)
        torch.cuda.synchronize()
        roofline_from = "HIP events around the class in a repeat of the timed steps (the timed region runs without events)"
    stats = st.kernel_stats()
    st.set_profiling(False)
    if rank == 0:
        log(f"timed region done: {dt / args.steps * 1e3:.1f} ms per step")
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
    if not mel_only:
        n_tok = int((toks >= 0).sum(axis=1).min())
        assert n_tok == tokens + 1, f"decode produced {n_tok} tokens per utterance, expected {tokens + 1}"

    if world > 1:
        tmax = torch.tensor([dt], device=dev if gather_dev is not None else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- the PCIe-inclusive rate: one extra step on rank 0 with the same utterances handed over as host buffers
    host_pcm_info = None
    if headline and rank == 0 and host_pcm is None and not args.no_breakdown:
        hp = list(pcm_dev.cpu().numpy())
        st.full_batch(fp, hp)
        t1 = time.perf_counter()
        st.full_batch(fp, hp)
        torch.cuda.synchronize()
        hms = (time.perf_counter() - t1) * 1e3
        host_pcm_info = {"ms_per_step": hms, "value": 30.0 * B / (hms * 1e-3), "note": "one step, PCM passed as pageable host buffers (wmi_full_batch copies them to the device); never `value`"}
        del hp
        log(f"host-PCM step: {hms:.1f} ms")

    # ---- strong scaling: a fixed job pulled from a shared queue of batches by every rank, one ragged gather at the end
    strong = None
    if headline and args.strong_total > 0 and host_pcm is None:
        sb = min(args.strong_batch, B)
        n_batches = (args.strong_total + sb - 1) // sb
        queue = BatchQueue(n_batches, name=f"wmi_strong_{args.strong_total}")
        # every finished batch goes to an append-only log before the next one is pulled: a run that dies late resumes
        # (--resume) at the first batch the log does not hold instead of starting over
        import tempfile
        job_key = f"{arch}-n{args.strong_total}-b{sb}-t{tokens}-beam{beam}"
        log_dir = args.result_log or os.path.join(tempfile.gettempdir(), f"wmi_result_log_{job_key}")
        rlog = ResultLog(log_dir, job_key, rank, world, resume=args.resume)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        already = rlog.done()
        recs, pulled = rlog.records(), 0
        for b in queue:
            if b in already:
                continue
            u0, u1 = b * sb, min((b + 1) * sb, args.strong_total)
            sel = [(u % B) for u in range(u0, u1)]           # utterance u of the job is clip u mod B of this rank's resident pool
            st.full_batch_device(fp, [ptrs[i] for i in sel], [N_SAMP] * len(sel))
            rec = pack_token_records(range(u0, u1), [st.batch_tokens(k) for k in range(len(sel))], tokens + 2)
            rlog.append(b, rec)
            recs.append(rec)
            pulled += 1
        rlog.close()
        rec = np.concatenate(recs, axis=0) if recs else np.zeros((0, tokens + 4), np.int32)
        allrec = gather_ragged_records(rec, device=gather_dev)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        sdt = time.perf_counter() - t1
        if world > 1:
            tm = torch.tensor([sdt], device=dev if gather_dev is not None else "cpu", dtype=torch.float64)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            sdt = float(tm.item())
        if rank == 0:
            assert allrec is not None and allrec[:, 0].tolist() == list(range(args.strong_total)), "strong-scaling job: records missing or duplicated"
            assert int((allrec[:, 1] == tokens + 1).all()), "strong-scaling job: wrong token count"
            strong = {"utterances": args.strong_total, "batch": sb, "batches": n_batches, "batches_pulled_by_rank0": pulled, "batches_taken_from_log": len(already), "result_log": log_dir, "seconds": sdt,
                      # a resumed job is priced on what this run transcribed (None when the log already held everything)
                      "value": (30.0 * sum(min((b + 1) * sb, args.strong_total) - b * sb for b in range(n_batches) if b not in already) / sdt) if len(already) < n_batches else None,
                      "unit": "audio-sec/sec", "scaling": "strong",
                      "note": "ranks pull batches from a shared queue (atomic counter on the rendezvous store), token records gathered once at the end"}
            log(f"strong-scaling job: {sdt:.2f} s")

    if rank == 0:
        audio_s = 30.0 * B * world * args.steps
        g = stats[dom]
        bound, kname = CLASS_KERNEL[dom]
        rate = g["work"] / (g["ms"] * 1e-3) if g["ms"] > 0 else 0.0          # flop/s or bytes/s
        achieved = rate / 1e12 if bound == "mfma" else rate / 1e9
        peak = MFMA_F16_DENSE_PEAK_TFLOPS if bound == "mfma" else HBM_PEAK_GBPS
        per_class = None
        if breakdown is not None:
            per_class = {k: {"ms": v["ms"], "launches": v["launches"], "rate": (v["work"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 else 0.0}
                         for k, v in breakdown.items()}
            per_class["_note"] = f"one extra untimed step with every kernel class under HIP events ({bd_ms:.1f} ms wall); rate in T(flop|B)/s"
        traffic, traffic_src = pmc_traffic() if headline else (None, None)
        if mel_only:
            n_mel = ctx.n_mels()
            workload = (f"log-mel frontend only, {B} x 30 s = {B * 30 / 3600:.2f} h of 16 kHz audio to {n_mel}-bin log-mel, clamped to max - 8 per utterance and "
                        f"scaled ((x + 4) / 4), fp32 [utterance][mel][3000 frames] out, PCM resident in HBM; two kernels per step (raw log-mel + maximum, clamp + scale)")
            metric = f"RTFx (audio-sec/sec), log-mel frontend only, {n_mel} mel bins"
        else:
            workload = (f"Whisper {arch} fp16 (synthetic ggml weights of the exact architecture), {B} different 30 s utterances per GPU per step, "
                        f"{'beam size %d' % beam if beam > 1 else 'greedy'}, no_timestamps, {tokens} forced tokens per utterance, "
                        f"{'PCM handed over as host buffers every step (PCIe-inclusive)' if args.host_pcm else 'PCM resident in HBM'}")
            metric = f"RTFx (audio-sec/sec) whole node, Whisper {arch}"
        line = {
            "metric": metric, "value": audio_s / dt, "unit": "audio-sec/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if mel_only else "f16", "data": "synthetic",
            "config": {"workload": workload, "name": args.config, "utterances_per_gpu": B, "pipeline_groups": args.pipeline_groups,
                       "encoder_lanes": args.encoder_lanes, "tokens_per_utterance": tokens, "beam_size": beam, "enc_chunk": min(args.enc_chunk, B),
                       "parallelism": f"dp{world} by utterance"},
            "roofline": {"bound": bound, "kernel": kname, "achieved": achieved, "peak": peak, "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
                         "frac": achieved / peak, "traffic": traffic,
                         "traffic_note": f"bytes per launch at the L2-fabric boundary (HBM + Infinity Cache), mean over the class, from profiles/{traffic_src}" if traffic else None,
                         "algorithmic_work_per_launch": g["work"] / max(g["launches"], 1),
                         "avg_launch_ms": g["ms"] / max(g["launches"], 1), "launches": g["launches"], "measured": roofline_from},
            "kernel_classes": per_class,
            # where the host thread spent the timed steps (ms per step): launching decoder passes vs waiting for them
            "host": {k: (host_t[k] - host_t0[k]) / args.steps for k in ("mel_ms", "encode_launch_ms", "decode_launch_ms", "decode_wait_ms", "passes")},
        }
        if strong is not None:
            line["strong_scaling"] = strong
        if host_pcm_info is not None:
            line["host_pcm"] = host_pcm_info
        if not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline_mel(path) if mel_only else cpu_baseline(path, tokens)
            except Exception as e:      # the baseline is reported, never required for the GPU number
                line["cpu_baseline"] = {"value": None, "unit": "audio-sec/sec", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(line))
    st.free()
    ctx.free()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
