"""Where the wall time of one headline step goes on the host side: log-mel, encoder launches + wait, decoder passes (building and
launching, waiting for the stream) and the state machines between passes. B utterances x 30 s, large-v3-turbo, 128 forced tokens."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import model_file
from handy_amd import whisper_rs as wr
from handy_amd.synth import synth_pcm
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0); n = 480000
ctx = wr.WhisperContext(model_file("large-v3-turbo"))
st = ctx.create_batch_state(B, 1, 128); st.set_force_tokens(128)
fp = wr.FullParams(wr.SamplingStrategy.Greedy(1)); fp.set_temperature_inc(0.0); fp.set_no_timestamps(True); fp.set_language("en")
pool = np.stack([synth_pcm(i, n) for i in range(8)])
pcm = torch.from_numpy(pool).to(dev).repeat((B + 7) // 8, 1)[:B].contiguous()
ptrs = [pcm.data_ptr() + i * n * 4 for i in range(B)]; ns = [n] * B
st.full_batch_device(fp, ptrs, ns); torch.cuda.synchronize()
h0 = st.host_timings(); t0 = time.perf_counter()
st.full_batch_device(fp, ptrs, ns); torch.cuda.synchronize()
wall = (time.perf_counter() - t0) * 1e3; h1 = st.host_timings()
d = {k: h1[k] - h0[k] for k in h1}
d["wall_ms"] = wall
d["between_passes_and_other_host_ms"] = wall - d["mel_ms"] - d["encode_launch_ms"] - d["decode_ms"]
print(json.dumps({"utterances": B, **{k: round(v, 2) for k, v in d.items()}}))
