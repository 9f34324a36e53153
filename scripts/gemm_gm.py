"""Sweep of the tile-order knob (tile rows per L2 group) on the encoder GEMM shapes, shipped kernel (variant 15) and its
fp32 residual-epilogue form (variant 31)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handy_amd import whisper_rs as wr
L = wr.load_library(wr.HARNESS_LIB_PATH)      # the benchmark-harness build of the library
L.wmi_bench_gemm.restype = C.c_int
L.wmi_bench_gemm.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float)]
L.wmi_debug_set_tile_group.argtypes = [C.c_int]
for (M, N, K) in [(int(sys.argv[1]) if len(sys.argv) > 1 else 24000, n, k) for (n, k) in ((1280, 5120), (1280, 1280), (5120, 1280), (3840, 1280))]:
    for gm in (3, 4, 5, 6, 8, 12, 16, 32):
        L.wmi_debug_set_tile_group(gm)
        best = 1e9
        for rep in range(3):
            ms = C.c_float(); rc = L.wmi_bench_gemm(M, N, K, 15, 10, C.byref(ms))
            best = min(best, ms.value)
        print(f"M={M} N={N} K={K} gm={gm:3d} {best*1e3:8.1f} us {2.0*M*N*K/(best*1e-3)/1e12:7.1f} TF/s", flush=True)
