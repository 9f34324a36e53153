"""Does a row's result depend on how many rows share its decoder pass?  Decodes the same prefix in a pass of 250 rows and in
a pass of N rows and prints the largest logit difference (0.0 = same bits). WMI_SPLIT_ROWS picks the row count up to which the
narrow GEMMs take the split-K path."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from handy_amd import ggml_file as gf, whisper_rs as wr
from handy_amd.synth import synth_pcm

wr.load_library()
for arch in sys.argv[1:] or ["mini", "mini768"]:
    p = os.path.join(tempfile.mkdtemp(), arch + ".bin")
    gf.write_synthetic_model(p, arch, seed=0)
    ctx = wr.WhisperContext(p); st = ctx.create_state()
    st.pcm_to_mel(synth_pcm(9, 6 * 16000)); st.encode(0)
    rng = np.random.default_rng(17)
    toks = [50257] + [int(t) for t in rng.integers(0, 50000, size=249)]
    long = st.decode_logits_all(toks, 0)
    for n in (1, 8, 16, 17, 32, 33, 64, 65, 128, 200):
        short = st.decode_logits_all(toks[:n], 0)
        print(arch, "split<=", os.environ.get("WMI_SPLIT_ROWS", "default"), "rows", n, "max diff vs 250-row pass", float(np.abs(short - long[:n]).max()), flush=True)
    st.free(); ctx.free()
