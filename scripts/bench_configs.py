"""Secondary workloads named beside the headline one (parity-test cases, not the contract line of bench.py):
  tiny     Whisper tiny.en, one 30 s clip, greedy with default parameters' decoding length forced to 64 tokens: latency
  small    Whisper small fp16, 32 x 30 s, greedy, 128 forced tokens
  beam     Whisper large-v3 fp16, 64 x 30 s, beam size 5, 64 forced tokens (decoder KV-cache and beam path)
Synthetic ggml weights of the exact architectures, PCM resident in HBM. Prints one JSON object per workload."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import model_file, cpu_baseline
from handy_amd import whisper_rs as wr
from handy_amd.synth import synth_pcm

which = sys.argv[1:] or ["tiny", "small", "beam"]
dev = torch.device("cuda", 0)
n = 480000
CFG = {"turbo1": ("large-v3-turbo", 1, 64, "greedy"), "turbo1beam": ("large-v3-turbo", 1, 64, "beam5"),
       "tiny": ("tiny.en", 1, 64, "greedy"), "small": ("small", 32, 128, "greedy"), "beam": ("large-v3", 64, 64, "beam5")}
for name in which:
    arch, B, tokens, mode = CFG[name]
    t0 = time.time(); path = model_file(arch); t_model = time.time() - t0
    ctx = wr.WhisperContext(path)
    beam = 5 if mode == "beam5" else 1
    st = ctx.create_batch_state(B, beam, 0)
    st.set_force_tokens(tokens)
    fp = wr.FullParams(wr.SamplingStrategy.BeamSearch(5, -1.0) if beam > 1 else wr.SamplingStrategy.Greedy(1))
    fp.set_temperature_inc(0.0); fp.set_no_timestamps(True); fp.set_language("en")
    pool = np.stack([synth_pcm(i, n) for i in range(min(8, B))])
    pcm = torch.from_numpy(pool).to(dev).repeat((B + 7) // 8, 1)[:B].contiguous()
    ptrs = [pcm.data_ptr() + i * n * 4 for i in range(B)]; ns = [n] * B
    st.full_batch_device(fp, ptrs, ns)
    torch.cuda.synchronize()
    steps = 3
    t0 = time.perf_counter()
    for _ in range(steps):
        st.full_batch_device(fp, ptrs, ns)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps              # timed without per-kernel events
    ht = st.host_timings()
    st.set_profiling(True); st.reset_kernel_stats()
    for _ in range(steps):
        st.full_batch_device(fp, ptrs, ns)
    torch.cuda.synchronize()
    ks = st.kernel_stats()
    st.set_profiling(False)
    n_tok = len(st.batch_tokens(0))
    extra = {}
    if name == "tiny":      # the single-clip comparison: the same clip and token count on the host cores (CPU restatement, not whisper.cpp)
        cb = cpu_baseline(path, tokens, budget_s=20.0)
        extra = {"cpu_port_ms": 30.0 / cb["value"] * 1e3, "cpu_port_cores": cb["cores"], "cpu_port_sample": cb["sample"]}
    print(json.dumps({**extra, "workload": name, "arch": arch, "utterances": B, "mode": mode, "tokens_per_utterance": n_tok, "ms_per_step": dt * 1e3, "rtfx": 30.0 * B / dt,
                      "kernel_classes_ms_per_step": {k: round(v["ms"] / steps, 2) for k, v in ks.items()},
                      "cross_attn_TBps": ks["dec_cross_attn"]["work"] / max(ks["dec_cross_attn"]["ms"], 1e-9) / 1e9, "model_file_s": round(t_model, 1),
                      "host_ms_per_pass": {"launch": round(ht["decode_launch_ms"] / max(ht["passes"], 1), 4), "wait": round(ht["decode_wait_ms"] / max(ht["passes"], 1), 4)}}), flush=True)
    st.free(); ctx.free()
