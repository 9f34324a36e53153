"""CPU restatement of the step in front of Handy's transcription path: resample to 16 kHz in 30 ms frames, then the smoothed
voice-activity gate.  TEST INFRASTRUCTURE ONLY: imported by tests/ and by nothing the product ships.

PARITY UNPINNED for the resampler: the arithmetic is rubato 0.16.2's `FftFixedIn<f32>` (reference
src-tauri/Cargo.lock:5119-5122, called from src-tauri/src/audio_toolkit/audio/resampler.rs:24), a crates.io dependency that is
not in the reference tree; the reference holds no test or vector for it.  What is restated below is rubato's published
algorithm as recalled ([recalled] marks), evaluated in float64; `tests/test_cpu_ingest.py` checks it against
scipy.signal.resample_poly, an independent band-limited resampler, on band-limited input.

The framing (`FrameResampler::push` / `finish`), the smoothing state machine (`SmoothedVad::push_frame`) and the recorder's
gather are in the reference tree and are restated line by line (citations at each function).
"""
import math

import numpy as np

OUT_HZ = 16000            # constants::WHISPER_SAMPLE_RATE
CHUNK_IN = 1024           # resampler.rs:5  RESAMPLER_CHUNK_SIZE
FRAME = 480               # resampler.rs:17 with recorder.rs:260-264: 30 ms at 16 kHz


def fft_sizes(in_hz, out_hz=OUT_HZ, chunk_in=CHUNK_IN):
    """[recalled] rubato FftFixedIn::new(in, out, chunk_size_in, sub_chunks = 1, channels = 1): the FFT block is the
    requested chunk rounded up to a whole number of output periods of the rate ratio."""
    g = math.gcd(in_hz, out_hz)
    min_chunk_out = out_hz // g
    fft_chunks = -(-chunk_in // min_chunk_out)
    return fft_chunks * in_hz // g, fft_chunks * out_hz // g          # (fft_size_in, fft_size_out)


def blackman_harris2(n):
    """[recalled] rubato windows.rs: periodic 4-term Blackman-Harris, squared."""
    x = np.arange(n, dtype=np.float64) / n
    w = 0.35875 - 0.48829 * np.cos(2 * np.pi * x) + 0.14128 * np.cos(4 * np.pi * x) - 0.01168 * np.cos(6 * np.pi * x)
    return w * w


def filter_time(n_in, n_out):
    """[recalled] rubato FftResampler::new + make_sincs(npoints = fft_size_in, factor = 1): windowed sinc centred at
    fft_size_in / 2, unit sum, scaled by 1 / (2 fft_size_in) (the two unnormalised transforms)."""
    cutoff = 0.4 ** (16.0 / n_in)
    if n_in > n_out:
        cutoff *= n_out / n_in
    t = (np.arange(n_in, dtype=np.float64) - (n_in // 2)) * cutoff
    h = blackman_harris2(n_in) * np.sinc(t)            # numpy's sinc is sin(pi x) / (pi x)
    h /= h.sum()
    return h / (2 * n_in)


def filter_spectrum(n_in, n_out):
    """Bins 0 .. new_len-1 of the filter's 2 fft_size_in-point transform; the bins from new_len on are dropped
    ([recalled] FftResampler::resample_unit: new_len = fft_size_out when decimating, fft_size_in + 1 otherwise)."""
    new_len = n_out if n_in > n_out else n_in + 1
    ht = np.zeros(2 * n_in); ht[:n_in] = filter_time(n_in, n_out)
    return np.fft.rfft(ht)[:new_len]


def resample_stream(pcm, in_hz, out_hz=OUT_HZ):
    """What `FftFixedIn::process` returns over all the 1024-sample chunks `FrameResampler::push` + `finish` hand it
    (resampler.rs:43-75): the input, zero-padded to whole chunks, is consumed in blocks of fft_size_in (a trailing partial block
    is never transformed); every block is zero-padded to twice its length, transformed, multiplied with the filter, cut to
    2 fft_size_out points, transformed back, and the second half of each result is added to the first half of the next."""
    pcm = np.asarray(pcm, dtype=np.float64)
    n_in, n_out = fft_sizes(in_hz, out_hz)
    padded = -(-len(pcm) // CHUNK_IN) * CHUNK_IN
    x = np.zeros(padded); x[: len(pcm)] = pcm
    n_blocks = padded // n_in
    F = filter_spectrum(n_in, n_out)
    out = np.zeros(n_blocks * n_out)
    overlap = np.zeros(n_out)
    for b in range(n_blocks):
        X = np.fft.rfft(x[b * n_in:(b + 1) * n_in], 2 * n_in)
        Y = np.zeros(n_out + 1, dtype=np.complex128)
        Y[: len(F)] = X[: len(F)] * F
        y = np.fft.irfft(Y, 2 * n_out) * (2 * n_out)             # realfft's inverse is unnormalised
        out[b * n_out:(b + 1) * n_out] = y[:n_out] + overlap
        overlap = y[n_out:]
    return out.astype(np.float32)


def resample_frames(pcm, in_hz, out_hz=OUT_HZ):
    """FrameResampler (resampler.rs:37-98): the resampled stream in 480-sample frames, the last one zero-padded; at 16 kHz the
    samples pass through untouched (:38-41)."""
    s = np.asarray(pcm, dtype=np.float32) if in_hz == out_hz else resample_stream(pcm, in_hz, out_hz)
    n_frames = -(-len(s) // FRAME)
    out = np.zeros(n_frames * FRAME, dtype=np.float32)
    out[: len(s)] = s
    return out.reshape(n_frames, FRAME)


def smoothed_gate(is_voice, prefill, hangover, onset):
    """SmoothedVad::push_frame over a whole recording (smoothed.rs:41-96) followed by the recorder's `handle_frame`
    (recorder.rs:290-295): returns the source frame index of every frame appended to the output, in order.  A frame still in
    the pre-roll buffer when speech starts again is appended again, as in the reference."""
    src, buf = [], []
    hang = ons = 0
    in_speech = False
    for i, v in enumerate(is_voice):
        buf.append(i)                                  # :43-46
        while len(buf) > prefill + 1:
            buf.pop(0)
        if not in_speech and v:                        # :53-71
            ons += 1
            if ons >= onset:
                in_speech = True; hang = hangover; ons = 0
                src.extend(buf)
        elif in_speech and v:                          # :74-77
            hang = hangover
            src.append(i)
        elif in_speech and not v:                      # :80-88
            if hang > 0:
                hang -= 1
                src.append(i)
            else:
                in_speech = False
        else:                                          # :91-94
            ons = 0
    return np.asarray(src, dtype=np.int64)


def gate_frames(frames, prob, threshold=0.3, prefill=15, hangover=15, onset=2):
    """Silero's decision per frame (silero.rs:46: prob > threshold; managers/audio.rs:121-123 for the defaults), the smoothed
    gate, the gather, and stop_recording's padding of clips under one second to 1.25 s (managers/audio.rs:402-411)."""
    src = smoothed_gate(np.asarray(prob) > threshold, prefill, hangover, onset)
    out = np.asarray(frames, dtype=np.float32)[src].reshape(-1)
    if 0 < len(out) < OUT_HZ:
        out = np.concatenate([out, np.zeros(OUT_HZ * 5 // 4 - len(out), dtype=np.float32)])
    return out, src
