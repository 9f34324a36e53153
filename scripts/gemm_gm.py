import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handy_amd import whisper_rs as wr
L = wr.load_library()
L.wmi_bench_gemm.restype = C.c_int
L.wmi_bench_gemm.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float)]
L.wmi_debug_set_tile_group.argtypes = [C.c_int]
for (M, N, K) in [(24000, 5120, 1280), (24000, 1280, 5120), (24000, 3840, 1280)]:
    for gm in (1, 2, 4, 8, 16, 94):
        L.wmi_debug_set_tile_group(gm)
        for v in (13, 1):
            ms = C.c_float(); rc = L.wmi_bench_gemm(M, N, K, v, 10, C.byref(ms))
            print(f"M={M} N={N} K={K} gm={gm:3d} variant {v:2d} {ms.value*1e3:8.1f} us {2.0*M*N*K/(ms.value*1e-3)/1e12:7.1f} TF/s", flush=True)
