"""Mirror of the reference's capture-side interface for bulk files, on the GPU: `FrameResampler`
(src-tauri/src/audio_toolkit/audio/resampler.rs:7-99) and `SmoothedVad` (audio_toolkit/vad/smoothed.rs:5-105) over whole
recordings in device memory, through the C ABI of libwhisper_mi355.so (`wmi_ingest_*`).  torch is used for device memory
only.  The per-frame speech probability (Silero in the reference, silero.rs:41-50) comes from the caller."""
import ctypes as C
from typing import Callable, Optional

import numpy as np

from . import whisper_rs as wr

WHISPER_SAMPLE_RATE = 16000
FRAME_SAMPLES = 480         # 30 ms


def _ptr(t):
    return C.c_void_p(t.data_ptr())


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

    def process(self, pcm_dev):
        """pcm_dev: 1-D float32 CUDA tensor at in_hz -> [n_frames, 480] float32 CUDA tensor at 16 kHz."""
        import torch
        if not (pcm_dev.is_cuda and pcm_dev.dtype == torch.float32 and pcm_dev.dim() == 1 and pcm_dev.is_contiguous()):
            raise ValueError("expected a contiguous 1-D float32 CUDA tensor")
        nf = self.n_frames(pcm_dev.numel())
        out = torch.empty((nf, FRAME_SAMPLES), dtype=torch.float32, device=pcm_dev.device)
        with torch.cuda.device(pcm_dev.device):
            st = torch.cuda.current_stream().cuda_stream
            rc = self.L.wmi_ingest_resample(_ptr(pcm_dev), pcm_dev.numel(), self.in_hz, _ptr(out), nf, C.c_void_p(st))
        if rc != 0:
            raise wr.WhisperError(f"wmi_ingest_resample failed: {rc}")
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

    def gather(self, frames_dev, src: np.ndarray):
        """frames_dev [n_frames, 480] on the device, src from `plan` -> 1-D float32 CUDA tensor of the speech samples
        (clips under one second padded to 1.25 s, as stop_recording does)."""
        import torch
        src = np.ascontiguousarray(src, dtype=np.int64)
        total = len(src) * FRAME_SAMPLES
        cap = max(total, WHISPER_SAMPLE_RATE * 5 // 4)
        out = torch.empty(cap, dtype=torch.float32, device=frames_dev.device)
        n_out = C.c_longlong(0)
        with torch.cuda.device(frames_dev.device):
            st = torch.cuda.current_stream().cuda_stream
            rc = self.L.wmi_ingest_gather(_ptr(frames_dev), frames_dev.shape[0], src.ctypes.data_as(C.POINTER(C.c_longlong)), len(src),
                                          _ptr(out), cap, C.byref(n_out), C.c_void_p(st))
        if rc != 0:
            raise wr.WhisperError(f"wmi_ingest_gather failed: {rc}")
        return out[: n_out.value]


def ingest(pcm_dev, in_hz: int, speech_prob: Callable, threshold: float = 0.3, vad: Optional[SmoothedVad] = None):
    """One recording from its native rate to the samples Handy's recorder would hand to `transcribe`
    (managers/audio.rs:121-123 defaults): resample, ask `speech_prob(frames_dev) -> per-frame probabilities` (host array),
    gate, gather."""
    frames = FrameResampler(in_hz).process(pcm_dev)
    vad = vad or SmoothedVad()
    prob = np.asarray(speech_prob(frames), dtype=np.float32)
    src = vad.plan(prob > threshold)
    return vad.gather(frames, src), src
