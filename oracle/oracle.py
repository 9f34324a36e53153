"""ctypes binding of oracle/libwhisper_oracle.so (the CPU checker).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg. The product package (handy_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libwhisper_oracle.so")


class WoParams(C.Structure):
    _fields_ = [
        ("strategy", C.c_int), ("n_max_text_ctx", C.c_int), ("offset_ms", C.c_int), ("duration_ms", C.c_int),
        ("translate", C.c_int), ("no_context", C.c_int), ("no_timestamps", C.c_int), ("single_segment", C.c_int),
        ("print_special", C.c_int), ("max_tokens", C.c_int), ("audio_ctx", C.c_int), ("tdrz_enable", C.c_int),
        ("language", C.c_char_p), ("detect_language", C.c_int), ("suppress_blank", C.c_int), ("suppress_nst", C.c_int),
        ("temperature", C.c_float), ("max_initial_ts", C.c_float), ("length_penalty", C.c_float),
        ("temperature_inc", C.c_float), ("entropy_thold", C.c_float), ("logprob_thold", C.c_float),
        ("no_speech_thold", C.c_float), ("best_of", C.c_int), ("beam_size", C.c_int),
        ("beam_candidates_topk", C.c_int), ("prompt_tokens", C.POINTER(C.c_int32)), ("prompt_n_tokens", C.c_int),
        ("force_n_tokens", C.c_int),
        ("logits_filter_callback", C.c_void_p), ("logits_filter_callback_user_data", C.c_void_p),
        ("encoder_begin_callback", C.c_void_p), ("encoder_begin_callback_user_data", C.c_void_p),
    ]


LOGITS_FILTER = C.CFUNCTYPE(None, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_float), C.c_void_p)
ENCODER_BEGIN = C.CFUNCTYPE(C.c_int, C.c_void_p)


class WoTokenData(C.Structure):
    _fields_ = [("id", C.c_int32), ("tid", C.c_int32), ("p", C.c_float), ("plog", C.c_float), ("pt", C.c_float),
                ("ptsum", C.c_float), ("t0", C.c_int64), ("t1", C.c_int64), ("t_dtw", C.c_int64), ("vlen", C.c_float)]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "whisper_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libwhisper_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, ci, fp = C.c_void_p, C.c_int, C.POINTER(C.c_float)
        L.wo_load.restype = vp; L.wo_load.argtypes = [C.c_char_p]
        L.wo_free_model.argtypes = [vp]
        L.wo_hparams.argtypes = [vp, C.POINTER(C.c_int)]
        L.wo_token_special.restype = ci; L.wo_token_special.argtypes = [vp, C.c_char_p]
        L.wo_token_to_str.restype = C.POINTER(C.c_char); L.wo_token_to_str.argtypes = [vp, ci, C.POINTER(ci)]
        L.wo_lang_id.restype = ci; L.wo_lang_id.argtypes = [C.c_char_p]
        L.wo_lang_str.restype = C.c_char_p; L.wo_lang_str.argtypes = [ci]
        L.wo_new_state.restype = vp; L.wo_new_state.argtypes = [vp]
        L.wo_free_state.argtypes = [vp]
        L.wo_pcm_to_mel.restype = ci; L.wo_pcm_to_mel.argtypes = [vp, fp, ci]
        L.wo_mel_n_len.restype = ci; L.wo_mel_n_len.argtypes = [vp]
        L.wo_mel_n_len_org.restype = ci; L.wo_mel_n_len_org.argtypes = [vp]
        L.wo_mel_data.restype = fp; L.wo_mel_data.argtypes = [vp]
        L.wo_set_mel.restype = ci; L.wo_set_mel.argtypes = [vp, fp, ci, ci]
        L.wo_encode.restype = ci; L.wo_encode.argtypes = [vp, ci, ci]
        for n in ("wo_enc_out", "wo_conv_out", "wo_logits"):
            getattr(L, n).restype = fp; getattr(L, n).argtypes = [vp]
        for n in ("wo_cross_k", "wo_cross_v"):
            getattr(L, n).restype = fp; getattr(L, n).argtypes = [vp, ci]
        L.wo_decode.restype = ci; L.wo_decode.argtypes = [vp, ci, C.POINTER(C.c_int32), ci, ci, ci]
        L.wo_process_logits.argtypes = [vp, C.POINTER(WoParams), fp, C.POINTER(C.c_int32), ci, ci, ci, C.c_float, fp, fp, fp]
        L.wo_default_params.restype = WoParams; L.wo_default_params.argtypes = [ci]
        L.wo_full.restype = ci; L.wo_full.argtypes = [vp, C.POINTER(WoParams), fp, ci]
        L.wo_full_n_segments.restype = ci; L.wo_full_n_segments.argtypes = [vp]
        L.wo_full_segment_t0.restype = C.c_int64; L.wo_full_segment_t0.argtypes = [vp, ci]
        L.wo_full_segment_t1.restype = C.c_int64; L.wo_full_segment_t1.argtypes = [vp, ci]
        L.wo_full_segment_text.restype = C.c_char_p; L.wo_full_segment_text.argtypes = [vp, ci]
        L.wo_full_n_tokens.restype = ci; L.wo_full_n_tokens.argtypes = [vp, ci]
        L.wo_full_token_data.restype = WoTokenData; L.wo_full_token_data.argtypes = [vp, ci, ci]
        L.wo_full_lang_id.restype = ci; L.wo_full_lang_id.argtypes = [vp]
        L.wo_n_fallbacks.restype = ci; L.wo_n_fallbacks.argtypes = [vp]
        L.wo_n_threads.restype = ci
        _lib = L
    return _lib


def _fptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class OracleModel:
    def __init__(self, path: str):
        self.L = lib()
        self.h = self.L.wo_load(path.encode())
        if not self.h:
            raise RuntimeError(f"oracle: cannot load {path}")
        hp = (C.c_int * 11)()
        self.L.wo_hparams(self.h, hp)
        (self.n_vocab, self.n_audio_ctx, self.n_audio_state, self.n_audio_head, self.n_audio_layer, self.n_text_ctx,
         self.n_text_state, self.n_text_head, self.n_text_layer, self.n_mels, self.ftype) = list(hp)

    def special(self, name: str) -> int:
        return self.L.wo_token_special(self.h, name.encode())

    def token_bytes(self, tid: int) -> bytes:
        n = C.c_int()
        p = self.L.wo_token_to_str(self.h, tid, C.byref(n))
        return C.string_at(p, n.value)

    def close(self):
        if self.h:
            self.L.wo_free_model(self.h)
            self.h = None


class OracleState:
    def __init__(self, model: OracleModel):
        self.m = model
        self.L = model.L
        self.h = self.L.wo_new_state(model.h)

    def close(self):
        if self.h:
            self.L.wo_free_state(self.h)
            self.h = None

    def pcm_to_mel(self, pcm: np.ndarray) -> np.ndarray:
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        rc = self.L.wo_pcm_to_mel(self.h, _fptr(pcm), len(pcm))
        if rc:
            raise RuntimeError(f"wo_pcm_to_mel {rc}")
        n_len = self.L.wo_mel_n_len(self.h)
        return np.ctypeslib.as_array(self.L.wo_mel_data(self.h), shape=(self.m.n_mels, n_len)).copy()

    @property
    def n_len_org(self) -> int:
        return self.L.wo_mel_n_len_org(self.h)

    def set_mel(self, mel: np.ndarray):
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        assert self.L.wo_set_mel(self.h, _fptr(mel), mel.shape[1], mel.shape[0]) == 0

    def encode(self, seek: int = 0, audio_ctx: int = 0) -> np.ndarray:
        rc = self.L.wo_encode(self.h, seek, audio_ctx)
        if rc:
            raise RuntimeError(f"wo_encode {rc}")
        n_ctx = audio_ctx or self.m.n_audio_ctx
        return np.ctypeslib.as_array(self.L.wo_enc_out(self.h), shape=(n_ctx, self.m.n_audio_state)).copy()

    def conv_out(self, n_ctx: Optional[int] = None) -> np.ndarray:
        n_ctx = n_ctx or self.m.n_audio_ctx
        return np.ctypeslib.as_array(self.L.wo_conv_out(self.h), shape=(n_ctx, self.m.n_audio_state)).copy()

    def cross_kv(self, layer: int, n_ctx: Optional[int] = None):
        n_ctx = n_ctx or self.m.n_audio_ctx
        k = np.ctypeslib.as_array(self.L.wo_cross_k(self.h, layer), shape=(n_ctx, self.m.n_text_state)).copy()
        v = np.ctypeslib.as_array(self.L.wo_cross_v(self.h, layer), shape=(n_ctx, self.m.n_text_state)).copy()
        return k, v

    def decode(self, tokens: List[int], n_past: int = 0, dec: int = 0, logits_all: bool = False) -> np.ndarray:
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        rc = self.L.wo_decode(self.h, dec, t.ctypes.data_as(C.POINTER(C.c_int32)), len(t), n_past, int(logits_all))
        if rc:
            raise RuntimeError(f"wo_decode {rc}")
        rows = len(t) if logits_all else 1
        return np.ctypeslib.as_array(self.L.wo_logits(self.h), shape=(rows, self.m.n_vocab)).copy()

    def process_logits(self, params: WoParams, logits: np.ndarray, tokens_cur: List[int], has_ts: bool, seek_delta: int,
                       temperature: float):
        V = self.m.n_vocab
        lin = np.ascontiguousarray(logits, dtype=np.float32)
        out_l, out_lp, out_p = (np.empty(V, np.float32) for _ in range(3))
        t = np.ascontiguousarray(tokens_cur, dtype=np.int32)
        self.L.wo_process_logits(self.m.h, C.byref(params), _fptr(lin), t.ctypes.data_as(C.POINTER(C.c_int32)), len(t),
                                 int(has_ts), seek_delta, temperature, _fptr(out_l), _fptr(out_lp), _fptr(out_p))
        return out_l, out_lp, out_p

    def full(self, params: WoParams, pcm: np.ndarray):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        rc = self.L.wo_full(self.h, C.byref(params), _fptr(pcm), len(pcm))
        if rc:
            raise RuntimeError(f"wo_full {rc}")
        segs = []
        for i in range(self.L.wo_full_n_segments(self.h)):
            toks = [self.L.wo_full_token_data(self.h, i, j) for j in range(self.L.wo_full_n_tokens(self.h, i))]
            segs.append({"t0": self.L.wo_full_segment_t0(self.h, i), "t1": self.L.wo_full_segment_t1(self.h, i),
                         "text": self.L.wo_full_segment_text(self.h, i),
                         "tokens": [t.id for t in toks], "plog": [t.plog for t in toks], "p": [t.p for t in toks]})
        return segs

    @property
    def n_fallbacks(self) -> int:
        return self.L.wo_n_fallbacks(self.h)


def default_params(strategy: int = 0) -> WoParams:
    return lib().wo_default_params(strategy)
