"""The no-allocation log-mel entry point (wmi_pcm_to_mel_device_noalloc) against the entry points it is a variant of
(whisper_pcm_to_mel_with_state + wmi_get_mel), through the C ABI: calls in a row, later ones with smaller inputs, so that
what an earlier call left in the library's fixed device array would show. Device buffers come straight from the HIP
runtime the library itself is linked against (no torch: its wheel carries a HIP runtime of its own)."""
import ctypes as C

import numpy as np
import pytest

from handy_amd.synth import synth_pcm

pytestmark = pytest.mark.gpu


class _Hip:
    def __init__(self):
        self.L = C.CDLL("libamdhip64.so.7")
        self.L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.L.hipFree.argtypes = [C.c_void_p]
        self.L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.L.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]

    def malloc(self, n):
        p = C.c_void_p()
        assert self.L.hipMalloc(C.byref(p), max(n, 4)) == 0
        return p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.malloc(arr.nbytes)
        if arr.nbytes:
            assert self.L.hipMemcpy(p, arr.ctypes.data, arr.nbytes, 1) == 0
        return p

    def download(self, p, shape, dtype=np.float32):
        out = np.empty(shape, dtype)
        assert self.L.hipMemcpy(out.ctypes.data, p, out.nbytes, 2) == 0      # synchronises with the null stream's work
        return out


@pytest.mark.parametrize("arch", ["micro-v3", "micro.en"])
def test_no_alloc_log_mel_equals_the_allocating_path(model_path, arch):
    from handy_amd import whisper_rs as wr
    ctx = wr.WhisperContext(model_path(arch))
    st = ctx.create_state()
    hip = _Hip()
    n_mel = ctx.n_mels()
    cap = n_mel * 6000
    out_dev = hip.malloc((cap + 64) * 4)
    for seed, n, n_out in ((1, 480000, 6000), (2, 77777, 3000), (3, 480000, 3000), (4, 1600, 100), (5, 0, 64)):
        pcm = synth_pcm(seed, max(n, 1))[:n]
        pcm_dev = hip.upload(pcm if n else np.zeros(1, np.float32))
        assert hip.L.hipMemset(out_dev, 0xFF, (cap + 64) * 4) == 0                      # NaN pattern everywhere
        assert hip.L.hipDeviceSynchronize() == 0
        assert ctx.pcm_to_mel_device_noalloc(pcm_dev, n, out_dev, n_out, 0) == 0         # enqueued on the null stream
        got_all = hip.download(out_dev, (cap + 64,))
        got = got_all[: n_mel * n_out].reshape(n_mel, n_out)
        if n:
            st.pcm_to_mel(pcm)
        else:
            assert ctx.L.whisper_pcm_to_mel_with_state(ctx.h, st.h, wr._fp(np.zeros(1, np.float32)), 0, 1) == 0
        ref = st.get_mel(n_out)
        assert np.isfinite(got).all()
        assert np.array_equal(got, ref), (seed, n, float(np.abs(got - ref).max()))       # same kernels: identical bits
        assert np.isnan(got_all[n_mel * n_out:]).all()                                    # nothing written past the result
        hip.L.hipFree(pcm_dev)
    # sizes beyond the stated maximum are refused and write nothing
    big = hip.upload(np.zeros(480001, np.float32))
    assert hip.L.hipMemset(out_dev, 0, (cap + 64) * 4) == 0
    assert ctx.pcm_to_mel_device_noalloc(big, 480001, out_dev, 3000, 0) == -2
    assert ctx.pcm_to_mel_device_noalloc(big, 1000, out_dev, 6001, 0) == -2
    assert not hip.download(out_dev, (cap + 64,)).any()
    hip.L.hipFree(big); hip.L.hipFree(out_dev)
    st.free(); ctx.free()
