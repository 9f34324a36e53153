"""The no-allocation log-mel entry point (wmi_pcm_to_mel_device_noalloc) against the entry points it is a variant of
(whisper_pcm_to_mel_with_state + wmi_get_mel), through the C ABI: calls in a row, later ones with smaller inputs, so that
what an earlier call left in the library's fixed device array would show; some on the null stream, some on a non-blocking
stream; the amount of free device memory must not move across a call. Device buffers come from the library's own wmi_dev_*
helpers, i.e. from the HIP runtime the library itself is linked against (no torch: its wheel carries a runtime of its own)."""
import ctypes as C

import numpy as np
import pytest

from handy_amd.synth import synth_pcm

pytestmark = pytest.mark.gpu


class _Hip:
    """Device buffers through the library's own wmi_dev_* helpers: the very runtime the entry point under test runs on."""

    def __init__(self, L):
        self.L = L

    def malloc(self, n):
        p = self.L.wmi_dev_alloc(max(n, 4))
        assert p
        return p

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.malloc(arr.nbytes)
        if arr.nbytes:
            assert self.L.wmi_dev_upload(p, arr.ctypes.data, arr.nbytes) == 0
        return p

    def download(self, p, shape, dtype=np.float32):
        out = np.empty(shape, dtype)
        assert self.L.wmi_dev_download(out.ctypes.data, p, out.nbytes) == 0      # synchronises with the null stream's work
        return out


@pytest.mark.parametrize("arch", ["micro-v3", "micro.en"])
def test_no_alloc_log_mel_equals_the_allocating_path(model_path, arch):
    from handy_amd import whisper_rs as wr
    ctx = wr.WhisperContext(model_path(arch))
    st = ctx.create_state()
    hip = _Hip(ctx.L)
    n_mel = ctx.n_mels()
    cap = n_mel * 6000
    out_dev = hip.malloc((cap + 64) * 4)
    stream = ctx.L.wmi_dev_stream_create()             # the later calls run on a non-blocking stream of the caller's
    assert stream
    for k, (seed, n, n_out) in enumerate(((1, 480000, 6000), (2, 77777, 3000), (3, 480000, 3000), (4, 1600, 100), (5, 0, 64))):
        pcm = synth_pcm(seed, max(n, 1))[:n]
        pcm_dev = hip.upload(pcm if n else np.zeros(1, np.float32))
        assert ctx.L.wmi_dev_memset(out_dev, 0xFF, (cap + 64) * 4) == 0                   # NaN pattern everywhere
        assert ctx.L.wmi_dev_sync() == 0
        free_before = ctx.L.wmi_dev_free_bytes()
        st_arg = stream if k >= 2 else 0
        assert ctx.pcm_to_mel_device_noalloc(pcm_dev, n, out_dev, n_out, st_arg) == 0     # enqueued, not waited for
        if st_arg:
            assert ctx.L.wmi_dev_stream_sync(stream) == 0
        assert ctx.L.wmi_dev_sync() == 0
        assert ctx.L.wmi_dev_free_bytes() == free_before, "the call changed the amount of free device memory"
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
        ctx.L.wmi_dev_free(pcm_dev)
    # sizes beyond the stated maximum are refused and write nothing
    big = hip.upload(np.zeros(480001, np.float32))
    assert ctx.L.wmi_dev_memset(out_dev, 0, (cap + 64) * 4) == 0
    assert ctx.pcm_to_mel_device_noalloc(big, 480001, out_dev, 3000, 0) == -2
    assert ctx.pcm_to_mel_device_noalloc(big, 1000, out_dev, 6001, 0) == -2
    assert not hip.download(out_dev, (cap + 64,)).any()
    ctx.L.wmi_dev_free(big); ctx.L.wmi_dev_free(out_dev); ctx.L.wmi_dev_stream_destroy(stream)
    st.free(); ctx.free()
