"""Times the fp32 read-modify-write (residual) epilogue of the 256x256 GEMM (variant 31) beside the f16 one (15)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handy_amd import whisper_rs as wr
L = wr.load_library(wr.HARNESS_LIB_PATH)      # the benchmark-harness build of the library
L.wmi_bench_gemm.restype = C.c_int
L.wmi_bench_gemm.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float)]
for (M, N, K) in [(24000, 1280, 1280), (24000, 1280, 5120)]:
    for rnd in range(2):
        for v in (15, 31):
            ms = C.c_float()
            rc = L.wmi_bench_gemm(M, N, K, v, 10, C.byref(ms))
            print(f"M={M} N={N} K={K} round {rnd} variant {v}: rc={rc} {ms.value*1e3:8.1f} us {2.0*M*N*K/(ms.value*1e-3)/1e12:7.1f} TF/s", flush=True)
