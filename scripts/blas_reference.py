"""Yardstick only (not on the product path): the vendor BLAS behind torch.matmul on the encoder's GEMM shapes, fp16 in, fp32
accumulate, C = A . W^T with both operands K-contiguous, next to this repo's kernel on the same box."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from handy_amd import whisper_rs as wr
L = wr.load_library(wr.HARNESS_LIB_PATH)      # the benchmark-harness build of the library
L.wmi_bench_gemm.restype = C.c_int
L.wmi_bench_gemm.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float)]
dev = torch.device("cuda", 0)
out = []
for (M, N, K) in [(int(sys.argv[1]) if len(sys.argv) > 1 else 24000, n, k) for (n, k) in ((5120, 1280), (1280, 5120), (3840, 1280), (1280, 1280))]:
    A = torch.randn(M, K, device=dev, dtype=torch.float16)
    W = torch.randn(N, K, device=dev, dtype=torch.float16)
    for _ in range(3):
        torch.matmul(A, W.t())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        torch.matmul(A, W.t())
    e1.record(); torch.cuda.synchronize()
    ms_blas = e0.elapsed_time(e1) / 20
    ms = C.c_float()
    assert L.wmi_bench_gemm(M, N, K, 15, 20, C.byref(ms)) == 0
    fl = 2.0 * M * N * K
    out.append({"M": M, "N": N, "K": K, "vendor_blas_TFLOPs": fl / ms_blas / 1e9, "this_kernel_TFLOPs": fl / ms.value / 1e9})
    print(out[-1], flush=True)
print(json.dumps({"probe": "encoder GEMM shapes: torch.matmul (vendor BLAS) vs gemm_nt_kernel 256x256", "results": out}))
