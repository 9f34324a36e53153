"""Model load time (whisper_init_from_file_with_params_no_state): large-v3-turbo dimensions, f16 and block-quantised files
(expanded to fp16 at load). Page cache warm (the files were just written)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handy_amd import whisper_rs as wr, ggml_file as gf
out = []
for name, ftype in (("f16", 1), ("q8_0", 7), ("q5_0", 8), ("q4_1", 3)):
    path = f"/tmp/wmi_load_probe_{name}.bin"
    t0 = time.time(); gf.write_synthetic_model(path, "large-v3-turbo", seed=0, ftype=ftype); tw = time.time() - t0
    best = 1e9
    for _ in range(3):
        t0 = time.time(); ctx = wr.WhisperContext(path); dt = time.time() - t0; ctx.free(); best = min(best, dt)
    out.append({"file": name, "size_MB": os.path.getsize(path) / 1e6, "load_s": best, "write_s": tw})
    os.remove(path)
    print(out[-1], flush=True)
print(json.dumps({"probe": "model load time, large-v3-turbo dimensions", "results": out}))
