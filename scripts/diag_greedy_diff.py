"""Diagnostic: where the HIP engine's greedy tokens leave the oracle's, and how close the oracle's own logits were there."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from handy_amd import ggml_file as gf, whisper_rs as wr
from handy_amd.synth import synth_pcm
from oracle import oracle as wo

arch = sys.argv[1] if len(sys.argv) > 1 else "micro.en"
secs = (30.0, 12.5, 4.0, 47.0, 1.2, 30.0, 61.0, 8.0) if arch != "mini" else (30.0, 12.5, 4.0, 33.0)
path = os.path.join(tempfile.gettempdir(), f"diag_{arch}.bin")
gf.write_synthetic_model(path, arch, seed=0, decisive=4.0, style="varied")
om = wo.OracleModel(path); ost = wo.OracleState(om)
ctx = wr.WhisperContext(path); st = ctx.create_state()
fp = wr.FullParams(wr.SamplingStrategy.Greedy(1)); fp.set_temperature_inc(0.0); fp.set_language("en")
op = wo.default_params(0); op.temperature_inc = 0.0; op.best_of = 1; op.language = b"en"
sot = om.special("sot")
for i, s in enumerate(secs):
    pcm = synth_pcm(10 + i, int(16000 * s))
    st.full(fp, pcm)
    got = st.batch_segments(0)
    ref = ost.full(op, pcm)
    g = [(x.t0, x.t1, x.tokens) for x in got]; r = [(x["t0"], x["t1"], x["tokens"]) for x in ref]
    if g == r:
        print(i, s, "same", len(r), "segments"); continue
    k = next((j for j in range(min(len(g), len(r))) if g[j] != r[j]), min(len(g), len(r)))
    print(i, s, "DIFF at segment", k, "of", len(g), len(r))
    print("   got", g[k] if k < len(g) else None)
    print("   ref", r[k] if k < len(r) else None)
    if k < len(g) and k < len(r):
        # raw-logit margin in the oracle between the two engines' choices at the first differing token of that segment
        a, b = g[k][2], r[k][2]
        j = next((q for q in range(min(len(a), len(b))) if a[q] != b[q]), None)
        print("   first differing token index in segment:", j, "got", a[j] if j is not None else None, "ref", b[j] if j is not None else None)
