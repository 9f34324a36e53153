"""Times the decoder's skinny split-K GEMM (variant 30) and the 64x64 fused kernel (variant 6) on the decode shapes."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handy_amd import whisper_rs as wr
L = wr.load_library(wr.HARNESS_LIB_PATH)      # the benchmark-harness build of the library
L.wmi_bench_gemm.restype = C.c_int
L.wmi_bench_gemm.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float)]
Ms = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [128]
for M in Ms:
    for (N, K) in [(1280, 1280), (3840, 1280), (5120, 1280), (1280, 5120)]:
        for v in (6, 30):
            ms = C.c_float()
            rc = L.wmi_bench_gemm(M, N, K, v, 50, C.byref(ms))
            gbs = (N * K * 2) / (ms.value * 1e-3) / 1e9 if rc == 0 and ms.value > 0 else 0
            print(f"M={M} N={N} K={K} variant {v}: rc={rc} {ms.value*1e3:7.1f} us  weights at {gbs:7.0f} GB/s", flush=True)
