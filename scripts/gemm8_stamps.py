"""Cycle stamps of the 8-phase GEMM (harness variant 53): where a phase's cycles go, per wave."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handy_amd import whisper_rs as wr
L = wr.load_library(wr.HARNESS_LIB_PATH)      # the benchmark-harness build of the library
L.wmi_bench_gemm.restype = C.c_int
L.wmi_bench_gemm.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float)]
L.wmi_debug_read_gemm_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
NB = 2048
shapes = [tuple(int(x) for x in a.split('x')) for a in sys.argv[1:]] or [(192000, 1280, 5120), (192000, 5120, 1280)]
for (M, N, K) in shapes:
    ms = C.c_float()
    L.wmi_bench_gemm(M, N, K, 53, 3, C.byref(ms))
    buf = (C.c_ulonglong * (NB * 64))()
    L.wmi_debug_read_gemm_stamps(buf, NB * 64)
    a = np.frombuffer(buf, dtype=np.uint64).astype(np.float64).reshape(NB, 8, 8)
    nk = a[0, 0, 3]
    print(f"M={M} N={N} K={K}: {ms.value*1e3:.1f} us/launch (stamped build), nk={nk:.0f}; cycles per K-step (4 phases), mean over the last {NB} workgroups")
    for w in range(8):
        r, dma, b1, mm, b2 = (a[:, w, i].mean() / nk for i in (0, 1, 2, 4, 5))
        print(f"  wave {w}: read+issue {r:7.1f}  dma wait {dma:7.1f}  barrier1 {b1:7.1f}  multiply {mm:7.1f}  barrier2 {b2:7.1f}  sum {r + dma + b1 + mm + b2:7.1f}")
