"""A batch-aware front for the engine, shaped like Handy's TranscriptionManager (Whisper arm only).

Handy funnels every transcription through one mutex, one utterance per call
(reference: src-tauri/src/managers/transcription.rs:344-466, lock at :383); its only concurrent caller, the Discord
voice loop, spawns one blocking `tm.transcribe(samples)` task per speaker and lets them queue on that mutex
(src-tauri/src/discord_conversation.rs:406-507). `TranscriptionManager.transcribe` below keeps that call shape and
its observable behaviour --

  * empty audio returns "" without touching the engine                       (transcription.rs:358-362)
  * no model loaded raises "Model is not loaded for transcription."          (transcription.rs:372-375)
  * language "auto" -> None, "zh-Hans" / "zh-Hant" -> "zh"                   (transcription.rs:394-405)
  * WhisperInferenceParams { language, translate, ..Default }                (transcription.rs:407-411)
  * the result text is trimmed                                                (transcription.rs:455)

-- but calls that arrive while a batch is being collected (or while the GPU is busy with the previous one) are
coalesced into ONE wmi_full_batch over all of them, which is what the batch engine is for. The engine state also keeps a
memo of finished 30 s windows, so the live-coaching loop, which re-transcribes the whole growing recording every 3 s
(src-tauri/src/live_coaching.rs:84-97), only pays for the windows that changed. Custom-word correction,
model download/selection, idle unloading and event emission stay in Handy (out of scope here).
"""
from __future__ import annotations

import threading
import time
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import whisper_rs as wr


def normalize_language(selected_language: str) -> Optional[str]:
    """Handy's settings value -> the code handed to Whisper (transcription.rs:394-405)."""
    if selected_language == "auto":
        return None
    if selected_language in ("zh-Hans", "zh-Hant"):
        return "zh"
    return selected_language


@dataclass
class _Request:
    audio: np.ndarray
    done: threading.Event = field(default_factory=threading.Event)
    text: str = ""
    error: Optional[BaseException] = None


class TranscriptionManager:
    """Thread-safe `transcribe(audio) -> str`; concurrent callers share GPU batches."""

    def __init__(self, max_batch: int = 16, collect_ms: float = 3.0, selected_language: str = "auto",
                 translate_to_english: bool = False, gpu_device: int = 0, engine_factory=None, window_cache: int = 16):
        self.max_batch = max(1, max_batch)
        self.collect_ms = collect_ms
        self.selected_language = selected_language
        self.translate_to_english = translate_to_english
        self.gpu_device = gpu_device
        # 30 s windows remembered between calls: live coaching transcribes the whole growing recording every 3 s
        # (reference: src-tauri/src/live_coaching.rs:84-97); 0 turns the memo off
        self.window_cache = window_cache
        # engine_factory(path, gpu_device, max_batch) -> object with transcribe_batch(list of arrays, WhisperInferenceParams) -> list with a
        # str (or, for an utterance that failed on its own, an exception) per array, and close(); the default drives libwhisper_mi355.so, tests inject a stand-in to exercise the queueing on a CPU
        self._engine_factory = engine_factory or _HipBatchEngine
        self._engine = None
        self._lock = threading.Lock()
        self._cv = threading.Condition(self._lock)
        self._queue: List[_Request] = []
        self._stop = False
        self._worker: Optional[threading.Thread] = None
        self.batches_run = 0           # observability for tests: how many engine calls served the requests so far
        self.requests_served = 0

    # ---- model lifecycle (transcription.rs:137-151, 189-243)
    def load_model(self, path: str):
        self.unload_model()
        eng = self._engine_factory(path, self.gpu_device, self.max_batch)
        if self.window_cache > 0 and hasattr(eng, "set_window_cache"):
            eng.set_window_cache(self.window_cache)
        with self._lock:
            self._engine = eng
            self._stop = False
            self._worker = threading.Thread(target=self._run, name="wmi-batcher", daemon=True)
            self._worker.start()

    def is_model_loaded(self) -> bool:
        with self._lock:
            return self._engine is not None

    def unload_model(self):
        with self._lock:
            worker, self._worker = self._worker, None
            self._stop = True
            self._cv.notify_all()
        if worker:
            worker.join()
        with self._lock:
            eng, self._engine = self._engine, None
            pending, self._queue = self._queue, []
        for r in pending:                      # callers that lost the race with unload get the same error as "not loaded"
            r.error = RuntimeError("Model is not loaded for transcription.")
            r.done.set()
        if eng:
            eng.close()

    # ---- the call Handy makes
    def transcribe(self, audio) -> str:
        audio = np.ascontiguousarray(audio, dtype=np.float32)
        if audio.size == 0:
            return ""
        req = _Request(audio)
        with self._lock:
            if self._engine is None:
                raise RuntimeError("Model is not loaded for transcription.")
            self._queue.append(req)
            self._cv.notify_all()
        req.done.wait()
        if req.error is not None:
            raise RuntimeError(f"Whisper transcription failed: {req.error}") if not isinstance(req.error, RuntimeError) else req.error
        return req.text.strip()

    # ---- batching worker
    def _run(self):
        while True:
            with self._lock:
                while not self._queue and not self._stop:
                    self._cv.wait()
                if self._stop:
                    return
                # a first request is here: give the callers racing with it a moment to join, up to max_batch
                deadline = time.monotonic() + self.collect_ms / 1e3
                while len(self._queue) < self.max_batch and not self._stop:
                    left = deadline - time.monotonic()
                    if left <= 0:
                        break
                    self._cv.wait(left)
                batch, self._queue = self._queue[:self.max_batch], self._queue[self.max_batch:]
                eng = self._engine
                params = wr.WhisperInferenceParams(language=normalize_language(self.selected_language), translate=self.translate_to_english)
            if not batch:
                continue
            try:
                texts = eng.transcribe_batch([r.audio for r in batch], params)
                for r, t in zip(batch, texts):
                    if isinstance(t, BaseException):   # this utterance failed on its own; the other callers keep their text
                        r.error = t
                    else:
                        r.text = t
            except BaseException as e:         # the batch as a whole failed: every caller of it sees the failure
                for r in batch:
                    r.error = e
            with self._lock:
                self.batches_run += 1
                self.requests_served += len(batch)
            for r in batch:
                r.done.set()


class _HipBatchEngine:
    """One context + one batch state of libwhisper_mi355.so (there is no CPU path: the library raises without a GPU)."""

    def __init__(self, path: str, gpu_device: int, max_batch: int):
        self.ctx = wr.WhisperContext(path, wr.WhisperContextParameters(gpu_device))
        self.state = self.ctx.create_batch_state(max_batch, 5, 0)
        self._proto = wr.WhisperEngine()

    def transcribe_batch(self, audios, params) -> List[str]:
        fp = self._proto._full_params(params)
        self.state.full_batch(fp, audios, allow_failed=True)      # a failed clip fails only its own caller (below)
        out = []
        for u in range(len(audios)):
            rc = self.state.batch_rc(u)
            out.append(wr.WhisperError("GenericError", rc) if rc != 0 else "".join(s.text for s in self.state.batch_segments(u)))
        return out

    def set_window_cache(self, n: int):
        self.state.set_window_cache(n)

    def close(self):
        self.state.free()
        self.ctx.free()
