"""Mirror of the reference's capture-side interface for bulk files, on the GPU: `FrameResampler`
(src-tauri/src/audio_toolkit/audio/resampler.rs:7-99) and `SmoothedVad` (audio_toolkit/vad/smoothed.rs:5-105) over whole
recordings in device memory, through the C ABI of libwhisper_mi355.so (`wmi_ingest_*`).  The per-frame speech probability
(Silero in the reference, silero.rs:41-50) comes from the caller.

Device memory comes from the HIP runtime the library itself is linked against, through the library's own `wmi_dev_*`
helpers (`DeviceBuffer`; no runtime is looked up by name here); anything else that exposes
`data_ptr()` and `numel()` (a torch CUDA tensor, say) is accepted as input or as `out=` too -- the caller then passes the
stream its data lives on."""
import ctypes as C
from typing import Callable, Optional

import numpy as np

from . import whisper_rs as wr

WHISPER_SAMPLE_RATE = 16000
FRAME_SAMPLES = 480         # 30 ms

def _runtime():
    return wr.load_library()


class DeviceBuffer:
    """float32 array in device memory on the current HIP device."""

    def __init__(self, shape):
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.n = int(np.prod(self.shape)) if self.shape else 1
        self.ptr = _runtime().wmi_dev_alloc(max(self.n, 1) * 4)
        if not self.ptr:
            raise wr.WhisperError(f"device allocation of {self.n * 4} bytes failed")

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = cls(a.shape)
        if a.size and _runtime().wmi_dev_upload(b.ptr, a.ctypes.data, a.nbytes) != 0:
            raise wr.WhisperError("hipMemcpy to the device failed")
        return b

    def numpy(self, n: Optional[int] = None) -> np.ndarray:
        """Copy back (waits for the null stream's work). n: only the first n floats, flat."""
        out = np.empty(self.shape if n is None else (int(n),), dtype=np.float32)
        if out.size and _runtime().wmi_dev_download(out.ctypes.data, self.ptr, out.nbytes) != 0:
            raise wr.WhisperError("hipMemcpy from the device failed")
        return out

    def data_ptr(self): return self.ptr
    def numel(self): return self.n

    def free(self):
        if self.ptr:
            _runtime().wmi_dev_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _as_device(x):
    if hasattr(x, "data_ptr") and hasattr(x, "numel"):
        return x
    return DeviceBuffer.from_numpy(x)


class FrameResampler:
    """FrameResampler::new(in_hz, out_hz, 30 ms); `process` is push(all samples) followed by finish()."""

    def __init__(self, in_hz: int, out_hz: int = WHISPER_SAMPLE_RATE, frame_ms: int = 30):
        if out_hz != WHISPER_SAMPLE_RATE or frame_ms != 30:
            raise ValueError("the engine resamples to 16 kHz in 30 ms frames, as the reference's recorder does")
        self.in_hz = int(in_hz)
        self.L = wr.load_library()
        if self.L.wmi_ingest_n_frames(self.in_hz, 0) < 0:
            raise ValueError(f"bad sample rate {in_hz}")

    def n_frames(self, n_in: int) -> int:
        return int(self.L.wmi_ingest_n_frames(self.in_hz, int(n_in)))

    def process(self, pcm, out=None, stream: int = 0):
        """pcm: float32 samples at in_hz (numpy array -> uploaded, or device memory) -> [n_frames, 480] float32 at 16 kHz in
        device memory (`out`, or a new DeviceBuffer). Enqueued on `stream` (a hipStream_t value; 0 = the null stream)."""
        x = _as_device(pcm)
        nf = self.n_frames(x.numel())
        if out is None:
            out = DeviceBuffer((nf, FRAME_SAMPLES))
        elif out.numel() < nf * FRAME_SAMPLES:
            raise ValueError("out is too small")
        rc = self.L.wmi_ingest_resample(C.c_void_p(x.data_ptr()), x.numel(), self.in_hz, C.c_void_p(out.data_ptr()), nf, C.c_void_p(stream))
        if rc != 0:
            raise wr.WhisperError(f"wmi_ingest_resample failed: {rc}")
        if not (x is pcm):
            out._keep = x                  # the upload must outlive the kernel
        return out


class SmoothedVad:
    """SmoothedVad::new(inner, prefill_frames, hangover_frames, onset_frames) applied to a whole recording: `plan` returns the
    source frame of every frame the recorder would have appended; `gather` collects them on the device."""

    def __init__(self, prefill_frames: int = 15, hangover_frames: int = 15, onset_frames: int = 2):
        self.prefill, self.hangover, self.onset = int(prefill_frames), int(hangover_frames), int(onset_frames)
        self.L = wr.load_library()

    def plan(self, is_voice) -> np.ndarray:
        v = np.ascontiguousarray(np.asarray(is_voice, dtype=np.uint8))
        vp = v.ctypes.data_as(C.POINTER(C.c_ubyte))
        n = self.L.wmi_ingest_gate_plan(vp, len(v), self.prefill, self.hangover, self.onset, None, 0)
        if n < 0:
            raise ValueError("bad gate parameters")
        src = np.zeros(max(int(n), 1), dtype=np.int64)
        self.L.wmi_ingest_gate_plan(vp, len(v), self.prefill, self.hangover, self.onset, src.ctypes.data_as(C.POINTER(C.c_longlong)), int(n))
        return src[: int(n)]

    def gather(self, frames, src: np.ndarray, stream: int = 0):
        """frames [n_frames, 480] in device memory, src from `plan` -> (DeviceBuffer, number of speech samples in it); clips under
        one second are padded to 1.25 s, as stop_recording does. Waits for `stream`."""
        src = np.ascontiguousarray(src, dtype=np.int64)
        total = len(src) * FRAME_SAMPLES
        cap = max(total, WHISPER_SAMPLE_RATE * 5 // 4)
        out = DeviceBuffer(cap)
        n_out = C.c_longlong(0)
        rc = self.L.wmi_ingest_gather(C.c_void_p(frames.data_ptr()), frames.numel() // FRAME_SAMPLES, src.ctypes.data_as(C.POINTER(C.c_longlong)),
                                      len(src), C.c_void_p(out.data_ptr()), cap, C.byref(n_out), C.c_void_p(stream))
        if rc != 0:
            raise wr.WhisperError(f"wmi_ingest_gather failed: {rc}")
        return out, int(n_out.value)


def ingest(pcm, in_hz: int, speech_prob: Callable, threshold: float = 0.3, vad: Optional[SmoothedVad] = None):
    """One recording from its native rate to the samples Handy's recorder would hand to `transcribe`
    (managers/audio.rs:121-123 defaults): resample, ask `speech_prob(frames as a host array) -> per-frame probabilities`,
    gate, gather. Returns (speech DeviceBuffer, number of samples, source frames)."""
    frames = FrameResampler(in_hz).process(pcm)
    vad = vad or SmoothedVad()
    prob = np.asarray(speech_prob(frames.numpy()), dtype=np.float32)
    src = vad.plan(prob > threshold)
    speech, n = vad.gather(frames, src)
    return speech, n, src
