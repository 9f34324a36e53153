"""Does an utterance's encoder result depend on what shares its encoder chunk? large-v3-turbo dimensions (synthetic weights):
the cross-attention K/V of one clip encoded inside a chunk of 128 utterances (persistent 256x256 GEMMs) against the same clip
encoded alone (256x128 and 64x64 tiles), compared bit for bit, per decoder layer."""
import json, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from handy_amd import ggml_file as gf, whisper_rs as wr
from handy_amd.synth import synth_pcm

p = os.path.join(tempfile.mkdtemp(), "turbo.bin")
gf.write_synthetic_model(p, "large-v3-turbo", seed=0, decisive=2.0)
ctx = wr.WhisperContext(p)
fp = wr.FullParams(wr.SamplingStrategy.Greedy(1)); fp.set_temperature_inc(0.0); fp.set_no_timestamps(True); fp.set_language("en")
clips = [synth_pcm(900 + i, 16000 * 30) for i in range(8)]
B = int(os.environ.get("WMI_DIAG_BATCH", "256"))
st = ctx.create_batch_state(B, 1, 128); st.set_force_tokens(2)
st.full_batch(fp, [clips[i % 8] for i in range(B)])
big = [st.cross_kv(l) for l in range(ctx.n_text_layer())]
st.free()
st1 = ctx.create_batch_state(1, 1, 1); st1.set_force_tokens(2)
st1.full_batch(fp, [clips[0]])
one = [st1.cross_kv(l) for l in range(ctx.n_text_layer())]
st1.free(); ctx.free()
for l, ((kb, vb), (k1, v1)) in enumerate(zip(big, one)):
    print(json.dumps({"layer": l, "k_values_different": int((kb != k1).sum()), "v_values_different": int((vb != v1).sum()),
                      "k_max_abs_diff": float(np.abs(kb - k1).max()), "v_max_abs_diff": float(np.abs(vb - v1).max()), "k_max_abs": float(np.abs(k1).max())}), flush=True)
