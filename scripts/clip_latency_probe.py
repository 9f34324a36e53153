"""One 30 s clip end to end through whisper_full (greedy, 64 forced tokens): wall time per call and where the host
spends it (launching the decoder passes, waiting for them). Usage: clip_latency_probe.py [arch ...]
WMI_PROBE_FTYPE: ggml file type of the synthetic model (1 f16, 3 Q4_1, 8 Q5_0, 7 Q8_0 ...); WMI_PROBE_QUANT=0: decode from the fp16 copies;
WMI_PROBE_GRAPHS=0: decoder passes as plain launches instead of HIP graph replays; WMI_PROBE_BATCH: clips per call (default 1)."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from handy_amd import ggml_file as gf, whisper_rs as wr
from handy_amd.synth import synth_pcm

L = wr.load_library()
L.wmi_set_quant_resident(1 if os.environ.get("WMI_PROBE_QUANT", "1") != "0" else 0)
out = {}
for arch in sys.argv[1:] or ["tiny.en", "large-v3-turbo"]:
    p = os.path.join(tempfile.mkdtemp(), arch + ".bin")
    ftype = int(os.environ.get("WMI_PROBE_FTYPE", "1"))
    gf.write_synthetic_model(p, arch, seed=0, decisive=2.0, ftype=ftype)
    nb = int(os.environ.get("WMI_PROBE_BATCH", "1"))
    ctx = wr.WhisperContext(p); st = ctx.create_batch_state(nb, 1, min(nb, 16))
    st.set_pass_graphs(os.environ.get("WMI_PROBE_GRAPHS", "1") != "0")
    st.set_force_tokens(64)
    st.set_quant_decode(os.environ.get("WMI_PROBE_QUANT", "1") != "0")
    fp = wr.FullParams(wr.SamplingStrategy.Greedy(1)); fp.set_temperature_inc(0.0); fp.set_no_timestamps(True); fp.set_language("en")
    pcm = synth_pcm(5, 30 * 16000)
    st.full_batch(fp, [pcm] * nb)
    walls = []
    h0 = st.host_timings()
    for _ in range(5):
        t = time.perf_counter(); st.full_batch(fp, [pcm] * nb); walls.append((time.perf_counter() - t) * 1e3)
    h1 = st.host_timings()
    d = {k: (h1[k] - h0[k]) / 5 for k in h0}
    out[arch] = {"wall_ms": sorted(walls)[2], "per_call": d, "us_per_pass": 1e3 * d["decode_ms"] / max(1.0, d["passes"]),
                 "few_rows": os.environ.get("WMI_FEW_ROWS", "default"), "graphs": os.environ.get("WMI_PROBE_GRAPHS", "1"), "clips": nb, "ftype": ftype, "quant_decode": os.environ.get("WMI_PROBE_QUANT", "1")}
    print(arch, json.dumps(out[arch]), flush=True)
    st.free(); ctx.free()
