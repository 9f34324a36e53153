"""Encoder output of the HIP engine against the CPU oracle on the test suite's synthetic models (plain and varied weights):
max and rms error per clip. WMI_LIB_PATH selects another build of the library, for A/B runs (run on the GPU box)."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handy_amd import ggml_file as gf, whisper_rs as wr
from handy_amd.synth import synth_pcm
from oracle import oracle as wo

wr.load_library()
tmp = tempfile.mkdtemp()
for arch, style, decisive in (("micro.en", "plain", 1.0), ("micro", "plain", 1.0), ("micro.en", "varied", 4.0), ("micro", "varied", 4.0), ("micro-v3", "varied", 4.0)):
    path = os.path.join(tmp, f"{arch}_{style}.bin")
    gf.write_synthetic_model(path, arch, seed=0, ftype=1, decisive=decisive, style=style)
    om = wo.OracleModel(path); ost = wo.OracleState(om)
    ctx = wr.WhisperContext(path); st = ctx.create_state()
    worst = rms = 0.0
    for i, secs in enumerate((30.0, 12.5, 4.0, 30.0, 30.0, 30.0, 8.0)):
        pcm = synth_pcm(10 + i, int(16000 * secs))
        ost.pcm_to_mel(pcm); st.pcm_to_mel(pcm)
        ref = ost.encode(0); st.encode(0); got = st.encoder_output()
        e = np.abs(got - ref)
        worst = max(worst, float(e.max())); rms = max(rms, float(np.sqrt((e ** 2).mean())))
    print(f"{arch:9s} {style:6s} encoder max abs err {worst:.3e}  rms {rms:.3e}  (ref std {ref.std():.3f}, max {np.abs(ref).max():.2f})", flush=True)
    st.free(); ctx.free(); ost.close(); om.close()
