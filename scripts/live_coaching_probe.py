"""Live-coaching pattern (reference src-tauri/src/live_coaching.rs:70-97): a recording that grows by 3 s per peek, every
peek transcribing all of it through whisper_full with default parameters. Total time of the peeks with and without the
window memo, large-v3-turbo dimensions (synthetic weights), results compared peek by peek."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import model_file
from handy_amd import whisper_rs as wr
from handy_amd.synth import synth_pcm

total_s = int(sys.argv[1]) if len(sys.argv) > 1 else 150
ctx = wr.WhisperContext(model_file("large-v3-turbo"))
rec = synth_pcm(5, total_s * 16000)
rec[10 * 16000:] *= 0.5
fp = wr.FullParams(wr.SamplingStrategy.BeamSearch(3, -1.0)); fp.set_language("en")
out = {}
texts = {}
for name, cap in (("memo_off", 0), ("memo_on", 16)):
    st = ctx.create_batch_state(1, 3, 1); st.set_window_cache(cap)
    st.full_batch(fp, [rec[:16000 * 3]])             # warm-up
    t0 = time.perf_counter(); per = []; tx = []
    for secs in range(3, total_s + 1, 3):
        t1 = time.perf_counter()
        st.full_batch(fp, [rec[: secs * 16000]])
        per.append((time.perf_counter() - t1) * 1e3)
        tx.append([(s.t0, s.t1, tuple(s.tokens)) for s in st.batch_segments(0)])
    out[name] = {"total_ms": (time.perf_counter() - t0) * 1e3, "last_peek_ms": per[-1], "max_peek_ms": max(per), **st.window_cache_stats()}
    texts[name] = tx
    st.free()
out["identical_results"] = texts["memo_off"] == texts["memo_on"]
out["recording_s"] = total_s; out["peeks"] = len(range(3, total_s + 1, 3))
print(json.dumps(out))
