"""LayerNorm kernel A/B: M rows of width 1280 (fp32 in, fp16 out), ms and TB/s of the 6 bytes per element it must move."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handy_amd import whisper_rs as wr
L = wr.load_library(wr.HARNESS_LIB_PATH)      # the benchmark-harness build of the library
L.wmi_bench_layer_norm.restype = C.c_int
L.wmi_bench_layer_norm.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
M = int(sys.argv[1]) if len(sys.argv) > 1 else 192000
out = {}
for rep in range(2):
    for v in (100, 1, 8, 9, 10, 11, 12, 13):
        ms = C.c_float()
        assert L.wmi_bench_layer_norm(M, 1280, v, 20, C.byref(ms)) == 0, v
        out.setdefault(v, []).append(round(M * 1280 * 6 / ms.value / 1e9, 3))
print(json.dumps({"rows": M, "TBps_by_variant": out}))
