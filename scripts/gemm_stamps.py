"""Cycle stamps of the stamped GEMM build (variant 14): prologue / K-loop / epilogue per wave and the per-CU timeline."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handy_amd import whisper_rs as wr
L = wr.load_library(wr.HARNESS_LIB_PATH)      # the benchmark-harness build of the library
L.wmi_bench_gemm.restype = C.c_int
L.wmi_bench_gemm.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float)]
L.wmi_debug_read_gemm_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
NB = 2048
VARIANT = int(os.environ.get("WMI_STAMP_VARIANT", "14"))
shapes = [tuple(int(x) for x in a.split('x')) for a in sys.argv[1:]] or [(24000, 1280, 5120), (24000, 5120, 1280), (24000, 3840, 1280)]
for (M, N, K) in shapes:
    ms = C.c_float()
    L.wmi_bench_gemm(M, N, K, VARIANT, 3, C.byref(ms))
    buf = (C.c_ulonglong * (NB * 64))()
    L.wmi_debug_read_gemm_stamps(buf, NB * 64)
    nblk = min(NB, ((M + 255) // 256) * (N // 256))
    a = np.frombuffer(buf, dtype=np.uint64).astype(np.float64).reshape(NB, 8, 8)[:nblk]
    nk = a[0, 0, 3]
    t0 = a[:, :, 4].min()
    print(f"M={M} N={N} K={K}: {ms.value*1e3:.1f} us/launch, {nblk} blocks, nk={nk:.0f}; kernel span {a[:, :, 7].max() - t0:.0f} cycles")
    for w in range(8):
        print(f"  wave {w}: prologue {a[:, w, 5].mean():8.0f}  loop/kstep: wait {a[:, w, 0].mean()/(nk-1):7.1f} barrier {a[:, w, 1].mean()/(nk-1):7.1f} compute {a[:, w, 2].mean()/nk:7.1f}"
              f"  epilogue issue {a[:, w, 6].mean():8.0f}  block life {(a[:, w, 7] - a[:, w, 4]).mean():8.0f}")
    life = a[:, :, 7].max(axis=1) - a[:, :, 4].min(axis=1)
    print(f"  block life: mean {life.mean():.0f} min {life.min():.0f} max {life.max():.0f}; sum of lives / 256 CUs = {life.sum()/256:.0f}")
    starts = np.sort(a[:, :, 4].min(axis=1) - t0)
    print("  block start times (every 128th):", " ".join(f"{x:.0f}" for x in starts[::128]))
