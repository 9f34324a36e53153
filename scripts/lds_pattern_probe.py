"""16-byte LDS accesses under different lane -> (row, slot) patterns: cycles-per-instruction ratios against the linear pattern."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handy_amd import whisper_rs as wr
L = wr.load_library(wr.HARNESS_LIB_PATH)      # the benchmark-harness build of the library
L.wmi_bench_lds_pattern.restype = C.c_int
L.wmi_bench_lds_pattern.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
names = ["linear", "one column of 64 rows", "4 lanes/row, (row>>1)&7", "4 lanes/row, row&7", "32 rows/half-wave, (row>>1)&7", "32 rows/half-wave, row&7", "16 rows/quarter-wave, row&7"]
out = {}
for rd in (0, 1):
    base = None
    for p in range(7):
        ms = C.c_float()
        assert L.wmi_bench_lds_pattern(p, rd, 50000, C.byref(ms)) == 0
        base = base or ms.value
        out[("read " if rd else "write ") + names[p]] = {"ms": round(ms.value, 3), "vs_linear": round(ms.value / base, 2)}
print(json.dumps(out, indent=1))
