"""The step in front of the transcription path (resample to 16 kHz frames + smoothed voice-activity gate): the CPU
restatement against independent checks, and the host-side gate of the library against the restatement.  No GPU."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ingest_oracle as io   # noqa: E402


def test_transform_block_sizes():
    """FftFixedIn::new(in_hz, 16000, 1024, 1, 1): the requested chunk rounded up to whole output periods of the ratio."""
    assert io.fft_sizes(48000) == (3072, 1024)
    assert io.fft_sizes(44100) == (3087, 1120)
    assert io.fft_sizes(32000) == (2048, 1024)
    assert io.fft_sizes(22050) == (1764, 1280)
    assert io.fft_sizes(8000) == (512, 1024)


@pytest.mark.parametrize("hz", [48000, 44100, 22050, 8000])
def test_resampler_restatement_reproduces_in_band_sinusoids(hz):
    """A band-limited input comes out as the same signal at 16 kHz, delayed by the filter's centre (fft_size_in / 2 input
    samples): error <= 5e-5 away from the ends (observed 3e-8 .. 2e-5, the largest for the 6.9 kHz tone at 48 kHz)."""
    n_in, _ = io.fft_sizes(hz)
    t = np.arange(2 * hz) / hz
    fs = [220.0, 1000.0, 3100.0] + ([6900.0] if hz > 16000 else [])
    x = sum(0.2 * np.sin(2 * np.pi * f * t + i) for i, f in enumerate(fs))
    y = io.resample_stream(x, hz)
    tt = np.arange(len(y)) / 16000 - (n_in // 2) / hz
    ref = sum(0.2 * np.sin(2 * np.pi * f * tt + i) for i, f in enumerate(fs))
    assert np.abs(y[3000:-3000] - ref[3000:-3000]).max() <= 5e-5


def test_resampler_restatement_against_scipy_resample_poly():
    """Independent resampler (polyphase FIR, 601-tap Kaiser low-pass) on band-limited noise at 48 kHz, where the delay is a whole
    number of output samples (512). The two low-passes differ in their transition bands only and the input holds nothing above
    6 kHz: they agree within 1e-5 (observed 3.3e-6)."""
    from scipy.signal import firwin, resample_poly
    rng = np.random.default_rng(5)
    n = 48000 * 2
    spec = np.fft.rfft(rng.standard_normal(n))
    spec[int(6000 / 48000 * n):] = 0
    x = np.fft.irfft(spec, n); x *= 0.3 / np.abs(x).max()
    y = io.resample_stream(x, 48000)
    ref = resample_poly(x, 1, 3, window=firwin(601, 0.95 / 3, window=("kaiser", 14.0)))
    d = 512
    m = min(len(y) - d, len(ref)) - 3000
    assert np.abs(y[d + 3000: d + m] - ref[3000:m]).max() <= 1e-5


def test_stream_length_and_framing():
    """Input is zero-padded to whole 1024-chunks (finish), only whole transform blocks come out, frames are 480 samples with the
    last one zero-padded; 16 kHz passes through."""
    for hz, n, want_stream in [(48000, 3072, 1024), (48000, 3071, 1024), (48000, 3073, 1024), (48000, 2048, 0), (48000, 6144, 2048),
                               (44100, 3087, 1120), (44100, 3000, 0), (8000, 100, 2048)]:
        assert len(io.resample_stream(np.zeros(n), hz)) == want_stream, (hz, n)
        assert io.resample_frames(np.zeros(n), hz).shape == (-(-want_stream // 480), 480)
    x = np.arange(1000, dtype=np.float32)
    f = io.resample_frames(x, 16000)
    assert f.shape == (3, 480) and np.array_equal(f.reshape(-1)[:1000], x) and not f.reshape(-1)[1000:].any()


def test_smoothed_gate_known_answers():
    """Hand-derived from smoothed.rs:41-96 with prefill 2, hangover 1, onset 2."""
    g = lambda v: io.smoothed_gate(np.array(v, dtype=bool), 2, 1, 2).tolist()
    assert g([0, 0, 0, 0]) == []
    assert g([0, 1, 0, 1, 0]) == []                          # a lone voiced frame never reaches the onset count
    assert g([0, 0, 0, 1, 1]) == [2, 3, 4]                   # onset at frame 4: the pre-roll (2 frames) and the frame itself
    assert g([1, 1]) == [0, 1]                               # the buffer holds what has arrived so far
    assert g([0, 0, 0, 1, 1, 1, 0, 0, 0]) == [2, 3, 4, 5, 6]            # one hangover frame, then the frame that ends speech is dropped
    assert g([1, 1, 0, 0, 1, 1, 0]) == [0, 1, 2, 3, 4, 5, 6]            # frame 3 ends speech (dropped), onset at 5 re-emits 3..5 from the pre-roll
    assert g([1, 1, 0, 0, 1, 1]) == [0, 1, 2, 3, 4, 5]
    assert io.smoothed_gate(np.array([1, 1, 0, 1, 1], dtype=bool), 3, 0, 2).tolist() == [0, 1, 1, 2, 3, 4]   # frame 1 twice: still in the pre-roll


def _lib():
    from handy_amd import whisper_rs as wr
    return wr.load_library()


def test_library_gate_plan_equals_restatement():
    """wmi_ingest_gate_plan is host code: random detector outputs under several parameter sets, the reference's defaults
    (managers/audio.rs:123: 15, 15, 2; onichan_conversation.rs:135: 10, 10, 2) among them."""
    L = _lib()
    rng = np.random.default_rng(11)
    for prefill, hangover, onset in [(15, 15, 2), (10, 10, 2), (0, 0, 1), (3, 0, 2), (0, 4, 3), (2, 1, 0)]:
        for p in (0.05, 0.3, 0.6, 0.95):
            v = (rng.random(400) < p)
            runs = np.repeat(v[::4], 4)                      # longer runs as well
            for seq in (v, runs, np.zeros(0, dtype=bool)):
                ref = io.smoothed_gate(seq, prefill, hangover, onset)
                u8 = np.ascontiguousarray(seq, dtype=np.uint8)
                ptr = u8.ctypes.data_as(C.POINTER(C.c_ubyte))
                n = L.wmi_ingest_gate_plan(ptr, len(u8), prefill, hangover, onset, None, 0)
                assert n == len(ref)
                src = np.zeros(max(n, 1), dtype=np.int64)
                assert L.wmi_ingest_gate_plan(ptr, len(u8), prefill, hangover, onset, src.ctypes.data_as(C.POINTER(C.c_longlong)), n) == n
                assert np.array_equal(src[:n], ref)
    assert L.wmi_ingest_gate_plan(None, 5, 1, 1, 1, None, 0) == -1
    assert L.wmi_ingest_gate_plan(None, 0, -1, 1, 1, None, 0) == -1


def test_library_frame_counts_equal_restatement():
    L = _lib()
    for hz in (8000, 11025, 16000, 22050, 24000, 32000, 44100, 48000):
        for n in (0, 1, 479, 480, 1023, 1024, 1025, 3071, 3072, 3073, 6174, 100000, 1234567):
            assert L.wmi_ingest_n_frames(hz, n) == io.resample_frames(np.zeros(n, dtype=np.float32), hz).shape[0], (hz, n)
    assert L.wmi_ingest_n_frames(0, 10) == -1 and L.wmi_ingest_n_frames(48000, -1) == -1


def test_gate_frames_pads_short_clips():
    frames = np.arange(40 * 480, dtype=np.float32).reshape(40, 480)
    prob = np.zeros(40); prob[10:14] = 0.9
    out, src = io.gate_frames(frames, prob, 0.3, 2, 1, 2)
    assert src.tolist() == [9, 10, 11, 12, 13, 14] and len(out) == 20000 and not out[6 * 480:].any()
    assert np.array_equal(out[: 6 * 480], frames[9:15].reshape(-1))
    out, src = io.gate_frames(frames, np.zeros(40))
    assert len(out) == 0 and len(src) == 0
