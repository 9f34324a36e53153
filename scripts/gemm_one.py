"""Runs one GEMM variant on one shape (for counter collection)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handy_amd import whisper_rs as wr
L = wr.load_library(wr.HARNESS_LIB_PATH)      # the benchmark-harness build of the library
L.wmi_bench_gemm.restype = C.c_int
L.wmi_bench_gemm.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float)]
v = int(sys.argv[1]); M, N, K = (int(x) for x in sys.argv[2:5])
ms = C.c_float()
rc = L.wmi_bench_gemm(M, N, K, v, 5, C.byref(ms))
print(rc, ms.value * 1e3, "us", 2.0 * M * N * K / (ms.value * 1e-3) / 1e12, "TF/s")
