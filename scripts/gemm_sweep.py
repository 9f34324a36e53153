"""A/B of GEMM tile variants on the encoder's shapes (run on the GPU box)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handy_amd import whisper_rs as wr
L = wr.load_library(os.environ.get("WMI_HARNESS_LIB_PATH") or wr.HARNESS_LIB_PATH)      # the benchmark-harness build of the library (WMI_HARNESS_LIB_PATH: another build, for A/B runs)
L.wmi_bench_gemm.restype = C.c_int
L.wmi_bench_gemm.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float)]
names = {0: "256x128 8w 3st", 1: "256x256 8w 2st", 2: "128x128 4w 3st", 3: "128x128 4w 2st", 4: "256x128 8w 2st", 5: "128x256 8w 3st", 6: "64x64 4w 4st", 7: "128x64 4w 4st", 9: "256x256 spread-dma", 10: "256x128 3st spread", 11: "256x256 no-gload", 12: "256x256 mfma-only", 13: "256x256 no-mfma", 14: "stamped", 15: "256x256 spread first half (shipped)", 22: "loads + MFMA (no LDS reads)", 23: "LDS reads only", 24: "loads only", 25: "loads only, same tile", 26: "full loop, same tile", 27: "buffer_load lds staging", 16: "shipped, bias + GELU epilogue", 31: "shipped, residual epilogue", 50: "8-phase", 60: "8-phase persistent", 64: "persistent loop, 1 tile/WG", 61: "persistent residual", 62: "persistent bias + GELU", 63: "persistent fp32 rows", 56: "8-phase fp32 rows", 54: "8-phase no stagger", 55: "8-phase dma first", 51: "8-phase residual", 52: "8-phase bias + GELU", 30: "skinny split-K (no consumer)"}
MROWS = int(os.environ.get('WMI_SWEEP_ROWS', '24000'))
shapes = [(MROWS, 5120, 1280), (MROWS, 1280, 5120), (MROWS, 2560, 1280), (MROWS, 1280, 1280)]
if os.environ.get('WMI_SWEEP_DEC'):       # the decoder's shapes: few rows, weights read once
    shapes = [(MROWS, 1280, 1280), (MROWS, 3840, 1280), (MROWS, 5120, 1280), (MROWS, 1280, 5120), (MROWS, 768, 768), (MROWS, 768, 3072)]
if os.environ.get('WMI_SWEEP_SHAPES'):     # explicit shapes: "M,N,K;M,N,K;..."
    shapes = [tuple(int(v) for v in t.split(',')) for t in os.environ['WMI_SWEEP_SHAPES'].split(';')]
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 9, 15]
L.wmi_debug_gemm_variant_diff.restype = C.c_int
L.wmi_debug_gemm_variant_diff.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float)]
for v in variants:
    for (M, N, K) in [(1000, 512, 256), (3000, 1280, 1280), (70000, 1280, 1280), (33000, 2560, 256)]:
        d = C.c_float()
        if v in (30, 31, 43, 51, 61): continue    # read-modify-write / partial-sum outputs: not comparable through this harness (the library refuses them too)
        ref = 16 if v in (52, 62) else 56 if v in (56, 63) else 6
        rc = L.wmi_debug_gemm_variant_diff(M, N, K, v, ref, C.byref(d))
        print(f"variant {v} vs {ref} on M={M} N={N} K={K}: rc={rc} max|diff|={d.value}", flush=True)
for (M, N, K) in shapes:
    for rnd in range(2):
        for v in variants:
            ms = C.c_float()
            rc = L.wmi_bench_gemm(M, N, K, v, 10, C.byref(ms))
            tf = 2.0 * M * N * K / (ms.value * 1e-3) / 1e12 if rc == 0 and ms.value > 0 else 0
            print(f"M={M} N={N} K={K} round {rnd} variant {v} ({names.get(v, '?'):15s}) rc={rc} {ms.value*1e3:8.1f} us {tf:7.1f} TF/s", flush=True)
