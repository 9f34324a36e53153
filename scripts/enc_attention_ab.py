"""Encoder attention kernels side by side (harness library): outputs of the two kernels compared bit for bit on the score
profiles of the parity test, then both timed at the headline shape (128 utterances x 20 heads x 1500 keys) inside one process.
Usage: enc_attention_ab.py [rounds]"""
import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from handy_amd import whisper_rs as wr

L = C.CDLL(os.environ.get("WMI_HARNESS_LIB_PATH", os.path.join(os.path.dirname(wr.__file__), "libwhisper_mi355_harness.so")))
L.wmi_debug_enc_attention.restype = C.c_int
L.wmi_debug_enc_attention.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
L.wmi_bench_enc_attention.restype = C.c_int
L.wmi_bench_enc_attention.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]


def run(B, H, T, q, k, v):
    out = np.empty((B, T, H * 64), np.uint16)
    assert L.wmi_debug_enc_attention(B, H, T, q.ctypes.data_as(C.c_void_p), k.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) == 0
    return out


bad = wide_bad = 0
worst_a = worst_w = 0.0
for T in (1500, 1472, 100, 40, 64, 128, 129, 192, 200, 31, 32, 33, 95, 96, 97, 257, 300):
    for shape in ("random", "rising", "spike", "falling", "low_start"):
        B, H = 2, 3
        rng = np.random.default_rng(T * 10 + len(shape))
        q = rng.standard_normal((B, H, T, 64)) * 0.35; k = rng.standard_normal((B, H, T, 64)) * 0.7; v = rng.standard_normal((B, H, T, 64))
        j = np.arange(T)[None, None, :, None] / max(T - 1, 1)
        if shape != "random": q[..., :8] = (1.0 + 0.5 * (np.arange(T) % 5)[None, None, :, None]) * 0.5
        if shape == "rising": k[..., :8] = j * 7.5
        elif shape == "spike": k[..., :8] = 0.0; k[:, :, (2 * T) // 3, :8] = 4.0
        elif shape == "falling": k[..., :8] = -10.0 * (np.arange(T)[None, None, :, None] >= 64); k[:, :, :64, :8] = 0.0
        elif shape == "low_start": k[..., :8] = 0.0; k[:, :, :min(64, T // 2), :8] = -6.0
        q16, k16, v16 = (np.ascontiguousarray(x.astype(np.float16)) for x in (q, k, v))
        L.wmi_debug_set_enc_attention(3); a = run(B, H, T, q16, k16, v16)
        L.wmi_debug_set_enc_attention(1); b = run(B, H, T, q16, k16, v16)
        n = int((a != b).sum())
        if n: bad += 1; print("DIFFERENT", T, shape, n, float(np.abs(a.view(np.float16).astype(np.float64) - b.view(np.float16).astype(np.float64)).max()), flush=True)
        # the 64-queries-per-wave kernel rounds its probabilities at another reference: compared with the float64 softmax
        L.wmi_debug_set_enc_attention(0); w = run(B, H, T, q16, k16, v16)
        sc = np.einsum("bhqd,bhkd->bhqk", q16.astype(np.float64), k16.astype(np.float64))
        pr = np.exp(sc - sc.max(-1, keepdims=True)); pr /= pr.sum(-1, keepdims=True)
        ref = np.einsum("bhqk,bhkd->bqhd", pr, v16.astype(np.float64)).reshape(B, T, H * 64)
        ea = float(np.abs(a.view(np.float16).astype(np.float64) - ref).max()); ew = float(np.abs(w.view(np.float16).astype(np.float64) - ref).max())
        worst_a = max(worst_a, ea); worst_w = max(worst_w, ew)
        if not (ew <= 4e-3) or not np.isfinite(w.view(np.float16).astype(np.float64)).all(): wide_bad += 1; print("WIDE OFF", T, shape, ea, ew, flush=True)
print(json.dumps({"bitwise_cases_different": bad, "wide_cases_over_4e-3": wide_bad, "max_err_vs_float64": {"enc_attention_kernel": worst_a, "enc_attention_wide_kernel": worst_w}}), flush=True)

ms = C.c_float()
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    for var in [3, 1, 0] + [int(x) for x in os.environ.get("WMI_ATTN_EXTRA", "").split(",") if x]:
        L.wmi_debug_set_enc_attention(var)
        rc = L.wmi_bench_enc_attention(128, 20, 1500, 20, C.byref(ms))
        flop = 4.0 * 128 * 20 * 1500 * 1500 * 64
        print(json.dumps({"round": rnd, "kernel": {3: "enc_attention_kernel", 1: "enc_attention_pipe_kernel", 0: "enc_attention_wide_kernel"}.get(var, "pipe ablation %d" % (var - 10)), "rc": rc, "us": round(ms.value * 1e3, 1), "TFLOPs": round(flop / ms.value / 1e9, 1)}), flush=True)
