"""Python mirror of the whisper-rs / transcribe-rs call surface Handy uses, bound with ctypes to
libwhisper_mi355.so through the whisper.h C ABI (include/whisper.h, include/wmi.h).

Handy's Rust host reaches the engine as (reference: src-tauri/src/managers/transcription.rs)
    WhisperEngine::new()                     :228
    engine.load_model(&path)                 :229
    engine.transcribe_samples(audio, params) :413-415  -> result.text
    engine.unload_model()                    :145
and transcribe-rs implements those on whisper-rs' WhisperContext / WhisperState / FullParams.
The Rust toolchain is not available in this build environment, so the same names, argument
meaning and error behaviour are mirrored here for the parity tests and the benchmark; the Rust
binding a maintainer would use is rust/whisper-rs-sys.

There is no CPU fallback: if the shared library is missing or no HIP device is present, loading
fails with an exception.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwhisper_mi355.so")
# the same sources built with -DWMI_HARNESS: GEMM / LayerNorm / LDS A/B kernels and their wmi_bench_* entry points (scripts/ only)
HARNESS_LIB_PATH = os.path.join(_HERE, "libwhisper_mi355_harness.so")

whisper_token = C.c_int32


class WhisperError(RuntimeError):
    """Mirrors whisper_rs::WhisperError (InitError, UnableToCalculateSpectrogram, FailedToEncode, ...)."""

    def __init__(self, kind: str, code: int = 0):
        super().__init__(f"{kind} (code {code})")
        self.kind, self.code = kind, code


class _Ahead(C.Structure):
    _fields_ = [("n_text_layer", C.c_int), ("n_head", C.c_int)]


class _Aheads(C.Structure):
    _fields_ = [("n_heads", C.c_size_t), ("heads", C.POINTER(_Ahead))]


class whisper_context_params(C.Structure):
    _fields_ = [("use_gpu", C.c_bool), ("flash_attn", C.c_bool), ("gpu_device", C.c_int),
                ("dtw_token_timestamps", C.c_bool), ("dtw_aheads_preset", C.c_int), ("dtw_n_top", C.c_int),
                ("dtw_aheads", _Aheads), ("dtw_mem_size", C.c_size_t)]


class whisper_token_data(C.Structure):
    _fields_ = [("id", C.c_int32), ("tid", C.c_int32), ("p", C.c_float), ("plog", C.c_float), ("pt", C.c_float),
                ("ptsum", C.c_float), ("t0", C.c_int64), ("t1", C.c_int64), ("t_dtw", C.c_int64), ("vlen", C.c_float)]


class _Greedy(C.Structure):
    _fields_ = [("best_of", C.c_int)]


class _Beam(C.Structure):
    _fields_ = [("beam_size", C.c_int), ("patience", C.c_float)]


class whisper_full_params(C.Structure):
    _fields_ = [
        ("strategy", C.c_int), ("n_threads", C.c_int), ("n_max_text_ctx", C.c_int), ("offset_ms", C.c_int), ("duration_ms", C.c_int),
        ("translate", C.c_bool), ("no_context", C.c_bool), ("no_timestamps", C.c_bool), ("single_segment", C.c_bool),
        ("print_special", C.c_bool), ("print_progress", C.c_bool), ("print_realtime", C.c_bool), ("print_timestamps", C.c_bool),
        ("token_timestamps", C.c_bool), ("thold_pt", C.c_float), ("thold_ptsum", C.c_float), ("max_len", C.c_int),
        ("split_on_word", C.c_bool), ("max_tokens", C.c_int), ("debug_mode", C.c_bool), ("audio_ctx", C.c_int),
        ("tdrz_enable", C.c_bool), ("suppress_regex", C.c_char_p), ("initial_prompt", C.c_char_p),
        ("prompt_tokens", C.POINTER(C.c_int32)), ("prompt_n_tokens", C.c_int), ("language", C.c_char_p),
        ("detect_language", C.c_bool), ("suppress_blank", C.c_bool), ("suppress_nst", C.c_bool),
        ("temperature", C.c_float), ("max_initial_ts", C.c_float), ("length_penalty", C.c_float),
        ("temperature_inc", C.c_float), ("entropy_thold", C.c_float), ("logprob_thold", C.c_float), ("no_speech_thold", C.c_float),
        ("greedy", _Greedy), ("beam_search", _Beam),
        ("new_segment_callback", C.c_void_p), ("new_segment_callback_user_data", C.c_void_p),
        ("progress_callback", C.c_void_p), ("progress_callback_user_data", C.c_void_p),
        ("encoder_begin_callback", C.c_void_p), ("encoder_begin_callback_user_data", C.c_void_p),
        ("abort_callback", C.c_void_p), ("abort_callback_user_data", C.c_void_p),
        ("logits_filter_callback", C.c_void_p), ("logits_filter_callback_user_data", C.c_void_p),
        ("grammar_rules", C.c_void_p), ("n_grammar_rules", C.c_size_t), ("i_start_rule", C.c_size_t), ("grammar_penalty", C.c_float),
    ]


# callback types of whisper_full_params (include/whisper.h)
LOGITS_FILTER_CALLBACK = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.POINTER(whisper_token_data), C.c_int, C.POINTER(C.c_float), C.c_void_p)
ENCODER_BEGIN_CALLBACK = C.CFUNCTYPE(C.c_bool, C.c_void_p, C.c_void_p, C.c_void_p)


class wmi_kernel_stat(C.Structure):
    _fields_ = [("ms", C.c_double), ("work", C.c_double), ("launches", C.c_longlong)]


KERNEL_CLASSES = ["mel", "enc_gemm", "enc_attn", "enc_other", "dec_gemm", "dec_cross_attn", "dec_other"]

# every symbol include/whisper.h and include/wmi.h declare (checked by the CPU test-suite)
WMI_NODE_MAX_TOKENS = 240


class wmi_node_record(C.Structure):
    """include/wmi.h: one utterance's outcome of a node-level job."""
    _fields_ = [("utt", C.c_int), ("rc", C.c_int), ("device", C.c_int), ("lang_id", C.c_int), ("n_segments", C.c_int), ("n_tokens", C.c_int),
                ("t0", C.c_int64), ("t1", C.c_int64), ("tokens", C.c_int32 * WMI_NODE_MAX_TOKENS)]


DECLARED_SYMBOLS = """
whisper_context_default_params whisper_context_default_params_by_ref whisper_init_from_file_with_params
whisper_init_from_file_with_params_no_state whisper_init_from_buffer_with_params whisper_init_from_buffer_with_params_no_state
whisper_init_state whisper_free whisper_free_state whisper_free_params whisper_free_context_params
whisper_pcm_to_mel whisper_pcm_to_mel_with_state whisper_set_mel whisper_set_mel_with_state whisper_encode whisper_encode_with_state
whisper_decode whisper_decode_with_state whisper_get_logits whisper_get_logits_from_state whisper_tokenize whisper_token_count
whisper_lang_max_id whisper_lang_id whisper_lang_str whisper_lang_str_full whisper_lang_auto_detect whisper_lang_auto_detect_with_state
whisper_n_len whisper_n_len_from_state whisper_n_vocab whisper_n_text_ctx whisper_n_audio_ctx whisper_is_multilingual
whisper_model_n_vocab whisper_model_n_audio_ctx whisper_model_n_audio_state whisper_model_n_audio_head whisper_model_n_audio_layer
whisper_model_n_text_ctx whisper_model_n_text_state whisper_model_n_text_head whisper_model_n_text_layer whisper_model_n_mels
whisper_model_ftype whisper_model_type whisper_model_type_readable
whisper_token_to_str whisper_token_eot whisper_token_sot whisper_token_solm whisper_token_prev whisper_token_nosp whisper_token_not
whisper_token_beg whisper_token_lang whisper_token_translate whisper_token_transcribe
whisper_print_timings whisper_reset_timings whisper_print_system_info whisper_log_set
whisper_full_default_params whisper_full_default_params_by_ref whisper_full whisper_full_with_state whisper_full_parallel
whisper_full_n_segments whisper_full_n_segments_from_state whisper_full_lang_id whisper_full_lang_id_from_state
whisper_full_get_segment_t0 whisper_full_get_segment_t0_from_state whisper_full_get_segment_t1 whisper_full_get_segment_t1_from_state
whisper_full_get_segment_speaker_turn_next whisper_full_get_segment_speaker_turn_next_from_state
whisper_full_get_segment_text whisper_full_get_segment_text_from_state whisper_full_n_tokens whisper_full_n_tokens_from_state
whisper_full_get_token_text whisper_full_get_token_text_from_state whisper_full_get_token_id whisper_full_get_token_id_from_state
whisper_full_get_token_data whisper_full_get_token_data_from_state whisper_full_get_token_p whisper_full_get_token_p_from_state
wmi_init_state_batch wmi_full_batch wmi_batch_n_utts wmi_batch_n_segments wmi_batch_segment_t0 wmi_batch_segment_t1
wmi_batch_segment_text wmi_batch_n_tokens wmi_batch_token_data wmi_batch_lang_id wmi_batch_n_fallbacks wmi_batch_rc
wmi_set_force_tokens wmi_set_beam_candidates_sampled wmi_mel_batch_device wmi_mel_batch_device_normalized wmi_pcm_to_mel_device_noalloc wmi_get_mel wmi_get_encoder_output wmi_get_cross_kv
wmi_decode_logits_all wmi_debug_gemm wmi_debug_gemm_gelu wmi_debug_expand_blocks wmi_debug_quant_rows wmi_debug_enc_attention wmi_set_window_cache wmi_get_window_cache_stats wmi_set_pipeline_groups wmi_set_encoder_lanes wmi_set_profiling wmi_set_profiling_classes wmi_get_kernel_stats wmi_reset_kernel_stats wmi_get_host_timings wmi_set_quant_decode wmi_set_quant_resident wmi_set_decoder_lanes wmi_set_pass_graphs wmi_version
wmi_ingest_n_frames wmi_ingest_resample wmi_ingest_gate_plan wmi_ingest_gather
wmi_ctx_for_device wmi_node_init wmi_node_free wmi_node_n_devices wmi_node_state wmi_node_context wmi_node_run wmi_node_last_gather
wmi_node_last_batches wmi_node_run_custom
wmi_dev_alloc wmi_dev_free wmi_dev_upload wmi_dev_download wmi_dev_memset wmi_dev_sync wmi_dev_free_bytes wmi_dev_stream_create
wmi_dev_stream_sync wmi_dev_stream_destroy
""".split()

_lib = None


def load_library(path: Optional[str] = None):
    """dlopen the engine. Raises if the HIP extension has not been built: there is no other path."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("WMI_LIB_PATH") or LIB_PATH          # WMI_LIB_PATH: another build of the library (A/B runs)
    if not os.path.exists(p):
        raise WhisperError(f"InitError: {p} is missing; build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(p)
    vp, ci, fp = C.c_void_p, C.c_int, C.POINTER(C.c_float)
    L.whisper_context_default_params.restype = whisper_context_params
    L.whisper_init_from_file_with_params_no_state.restype = vp
    L.whisper_init_from_file_with_params_no_state.argtypes = [C.c_char_p, whisper_context_params]
    L.whisper_init_from_buffer_with_params_no_state.restype = vp
    L.whisper_init_from_buffer_with_params_no_state.argtypes = [vp, C.c_size_t, whisper_context_params]
    L.whisper_init_state.restype = vp; L.whisper_init_state.argtypes = [vp]
    L.wmi_init_state_batch.restype = vp; L.wmi_init_state_batch.argtypes = [vp, ci, ci, ci]
    L.whisper_free.argtypes = [vp]; L.whisper_free_state.argtypes = [vp]
    L.whisper_full_default_params.restype = whisper_full_params; L.whisper_full_default_params.argtypes = [ci]
    L.whisper_full_with_state.restype = ci; L.whisper_full_with_state.argtypes = [vp, vp, whisper_full_params, fp, ci]
    L.wmi_full_batch.restype = ci
    L.wmi_full_batch.argtypes = [vp, vp, whisper_full_params, C.POINTER(C.c_void_p), C.POINTER(ci), ci, ci]
    L.whisper_pcm_to_mel_with_state.restype = ci; L.whisper_pcm_to_mel_with_state.argtypes = [vp, vp, fp, ci, ci]
    L.whisper_set_mel_with_state.restype = ci; L.whisper_set_mel_with_state.argtypes = [vp, vp, fp, ci, ci]
    L.whisper_encode_with_state.restype = ci; L.whisper_encode_with_state.argtypes = [vp, vp, ci, ci]
    L.whisper_decode_with_state.restype = ci; L.whisper_decode_with_state.argtypes = [vp, vp, C.POINTER(C.c_int32), ci, ci, ci]
    L.whisper_get_logits_from_state.restype = fp; L.whisper_get_logits_from_state.argtypes = [vp]
    L.wmi_decode_logits_all.restype = ci; L.wmi_decode_logits_all.argtypes = [vp, vp, C.POINTER(C.c_int32), ci, ci, fp]
    L.whisper_lang_auto_detect_with_state.restype = ci; L.whisper_lang_auto_detect_with_state.argtypes = [vp, vp, ci, ci, fp]
    L.whisper_n_len_from_state.restype = ci; L.whisper_n_len_from_state.argtypes = [vp]
    L.whisper_tokenize.restype = ci; L.whisper_tokenize.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int32), ci]
    L.whisper_lang_id.restype = ci; L.whisper_lang_id.argtypes = [C.c_char_p]
    L.whisper_lang_str.restype = C.c_char_p; L.whisper_lang_str.argtypes = [ci]
    L.whisper_lang_max_id.restype = ci
    for n in ("n_vocab", "n_text_ctx", "n_audio_ctx", "is_multilingual", "model_n_audio_state", "model_n_audio_head", "model_n_audio_layer",
              "model_n_text_layer", "model_n_mels", "model_ftype", "token_eot", "token_sot", "token_solm", "token_prev", "token_nosp",
              "token_not", "token_beg", "token_translate", "token_transcribe"):
        f = getattr(L, "whisper_" + n); f.restype = ci; f.argtypes = [vp]
    L.whisper_token_lang.restype = ci; L.whisper_token_lang.argtypes = [vp, ci]
    L.whisper_token_to_str.restype = C.POINTER(C.c_char); L.whisper_token_to_str.argtypes = [vp, ci]
    L.whisper_model_type_readable.restype = C.c_char_p; L.whisper_model_type_readable.argtypes = [vp]
    L.whisper_print_system_info.restype = C.c_char_p
    L.whisper_print_timings.restype = None; L.whisper_print_timings.argtypes = [vp]
    L.whisper_reset_timings.restype = None; L.whisper_reset_timings.argtypes = [vp]
    L.whisper_full_default_params_by_ref.restype = C.POINTER(whisper_full_params); L.whisper_full_default_params_by_ref.argtypes = [ci]
    L.whisper_free_params.restype = None; L.whisper_free_params.argtypes = [C.POINTER(whisper_full_params)]
    L.whisper_full_get_segment_speaker_turn_next_from_state.restype = C.c_bool
    L.whisper_full_get_segment_speaker_turn_next_from_state.argtypes = [vp, ci]
    L.wmi_version.restype = C.c_char_p
    L.whisper_full_n_segments_from_state.restype = ci; L.whisper_full_n_segments_from_state.argtypes = [vp]
    L.whisper_full_lang_id_from_state.restype = ci; L.whisper_full_lang_id_from_state.argtypes = [vp]
    L.whisper_full_get_segment_t0_from_state.restype = C.c_int64; L.whisper_full_get_segment_t0_from_state.argtypes = [vp, ci]
    L.whisper_full_get_segment_t1_from_state.restype = C.c_int64; L.whisper_full_get_segment_t1_from_state.argtypes = [vp, ci]
    L.whisper_full_get_segment_text_from_state.restype = C.POINTER(C.c_char); L.whisper_full_get_segment_text_from_state.argtypes = [vp, ci]
    L.whisper_full_n_tokens_from_state.restype = ci; L.whisper_full_n_tokens_from_state.argtypes = [vp, ci]
    L.whisper_full_get_token_data_from_state.restype = whisper_token_data; L.whisper_full_get_token_data_from_state.argtypes = [vp, ci, ci]
    L.whisper_full_get_token_id_from_state.restype = ci; L.whisper_full_get_token_id_from_state.argtypes = [vp, ci, ci]
    L.wmi_batch_n_utts.restype = ci; L.wmi_batch_n_utts.argtypes = [vp]
    L.wmi_batch_n_segments.restype = ci; L.wmi_batch_n_segments.argtypes = [vp, ci]
    L.wmi_batch_segment_t0.restype = C.c_int64; L.wmi_batch_segment_t0.argtypes = [vp, ci, ci]
    L.wmi_batch_segment_t1.restype = C.c_int64; L.wmi_batch_segment_t1.argtypes = [vp, ci, ci]
    L.wmi_batch_segment_text.restype = C.POINTER(C.c_char); L.wmi_batch_segment_text.argtypes = [vp, ci, ci]
    L.wmi_batch_n_tokens.restype = ci; L.wmi_batch_n_tokens.argtypes = [vp, ci, ci]
    L.wmi_batch_token_data.restype = whisper_token_data; L.wmi_batch_token_data.argtypes = [vp, ci, ci, ci]
    L.wmi_batch_lang_id.restype = ci; L.wmi_batch_lang_id.argtypes = [vp, ci]
    L.wmi_batch_n_fallbacks.restype = ci; L.wmi_batch_n_fallbacks.argtypes = [vp, ci]
    L.wmi_batch_rc.restype = ci; L.wmi_batch_rc.argtypes = [vp, ci]
    L.wmi_set_force_tokens.argtypes = [vp, ci]; L.wmi_set_beam_candidates_sampled.argtypes = [vp, ci]
    L.wmi_mel_batch_device.restype = ci; L.wmi_mel_batch_device.argtypes = [vp, vp, vp, C.POINTER(C.c_longlong), C.POINTER(ci), ci]
    L.wmi_mel_batch_device_normalized.restype = ci; L.wmi_mel_batch_device_normalized.argtypes = [vp, vp, vp, C.POINTER(C.c_longlong), C.POINTER(ci), ci, vp, ci]
    L.wmi_pcm_to_mel_device_noalloc.restype = ci; L.wmi_pcm_to_mel_device_noalloc.argtypes = [vp, vp, ci, vp, ci, vp]
    L.wmi_get_mel.restype = ci; L.wmi_get_mel.argtypes = [vp, ci, fp, ci]
    L.wmi_get_encoder_output.restype = ci; L.wmi_get_encoder_output.argtypes = [vp, fp]
    L.wmi_get_cross_kv.restype = ci; L.wmi_get_cross_kv.argtypes = [vp, ci, fp, fp]
    L.wmi_debug_gemm.restype = ci; L.wmi_debug_gemm.argtypes = [ci, ci, ci, fp, fp, fp]
    L.wmi_debug_gemm_gelu.restype = ci; L.wmi_debug_gemm_gelu.argtypes = [ci, ci, ci, fp, fp, fp, vp]
    L.wmi_debug_expand_blocks.restype = ci; L.wmi_debug_expand_blocks.argtypes = [ci, vp, C.c_longlong, vp]
    L.wmi_set_profiling.argtypes = [vp, ci]
    L.wmi_set_pipeline_groups.argtypes = [vp, ci]
    L.wmi_set_window_cache.argtypes = [vp, ci]
    L.wmi_get_window_cache_stats.argtypes = [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    L.wmi_set_encoder_lanes.argtypes = [vp, ci]; L.wmi_set_encoder_lanes.restype = ci
    L.wmi_set_profiling_classes.argtypes = [vp, C.c_uint]
    L.wmi_get_kernel_stats.restype = ci; L.wmi_get_kernel_stats.argtypes = [vp, C.POINTER(wmi_kernel_stat)]
    L.wmi_reset_kernel_stats.argtypes = [vp]
    L.wmi_get_host_timings.restype = ci; L.wmi_get_host_timings.argtypes = [vp, C.POINTER(C.c_double)]
    L.wmi_set_quant_decode.restype = ci; L.wmi_set_quant_decode.argtypes = [vp, ci]
    L.wmi_set_quant_resident.restype = None; L.wmi_set_quant_resident.argtypes = [ci]
    L.wmi_set_decoder_lanes.restype = ci; L.wmi_set_decoder_lanes.argtypes = [vp, ci]
    L.wmi_set_pass_graphs.restype = None; L.wmi_set_pass_graphs.argtypes = [vp, ci]
    ll, llp = C.c_longlong, C.POINTER(C.c_longlong)
    L.wmi_ingest_n_frames.restype = ll; L.wmi_ingest_n_frames.argtypes = [ci, ll]
    L.wmi_ingest_resample.restype = ci; L.wmi_ingest_resample.argtypes = [vp, ll, ci, vp, ll, vp]
    L.wmi_ingest_gate_plan.restype = ll; L.wmi_ingest_gate_plan.argtypes = [C.POINTER(C.c_ubyte), ll, ci, ci, ci, llp, ll]
    L.wmi_ingest_gather.restype = ci; L.wmi_ingest_gather.argtypes = [vp, ll, llp, ll, vp, ll, llp, vp]
    sz = C.c_size_t
    L.wmi_dev_alloc.restype = vp; L.wmi_dev_alloc.argtypes = [sz]
    L.wmi_dev_free.restype = None; L.wmi_dev_free.argtypes = [vp]
    L.wmi_dev_upload.restype = ci; L.wmi_dev_upload.argtypes = [vp, vp, sz]
    L.wmi_dev_download.restype = ci; L.wmi_dev_download.argtypes = [vp, vp, sz]
    L.wmi_dev_memset.restype = ci; L.wmi_dev_memset.argtypes = [vp, ci, sz]
    L.wmi_dev_sync.restype = ci; L.wmi_dev_sync.argtypes = []
    L.wmi_dev_free_bytes.restype = ll; L.wmi_dev_free_bytes.argtypes = []
    L.wmi_dev_stream_create.restype = vp; L.wmi_dev_stream_create.argtypes = []
    L.wmi_dev_stream_sync.restype = ci; L.wmi_dev_stream_sync.argtypes = [vp]
    L.wmi_dev_stream_destroy.restype = None; L.wmi_dev_stream_destroy.argtypes = [vp]
    if path is None:
        _lib = L
    return L


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class SamplingStrategy:
    """whisper_rs::SamplingStrategy"""

    @staticmethod
    def Greedy(best_of: int = 5):
        return ("greedy", best_of, -1.0)

    @staticmethod
    def BeamSearch(beam_size: int = 5, patience: float = -1.0):
        return ("beam", beam_size, patience)


class FullParams:
    """whisper_rs::FullParams: setters named as in the crate; holds the by-value C struct."""

    def __init__(self, strategy=("greedy", 5, -1.0)):
        L = load_library()
        kind, k, patience = strategy
        self.c = L.whisper_full_default_params(0 if kind == "greedy" else 1)
        if kind == "greedy":
            self.c.greedy.best_of = k
        else:
            self.c.beam_search.beam_size = k
            self.c.beam_search.patience = patience
        self._keep = {}

    def set_language(self, lang: Optional[str]):
        self._keep["language"] = lang.encode() if lang is not None else None
        self.c.language = self._keep["language"]

    def set_detect_language(self, v: bool): self.c.detect_language = v
    def set_translate(self, v: bool): self.c.translate = v
    def set_no_context(self, v: bool): self.c.no_context = v
    def set_no_timestamps(self, v: bool): self.c.no_timestamps = v
    def set_single_segment(self, v: bool): self.c.single_segment = v
    def set_print_special(self, v: bool): self.c.print_special = v
    def set_print_progress(self, v: bool): self.c.print_progress = v
    def set_print_realtime(self, v: bool): self.c.print_realtime = v
    def set_print_timestamps(self, v: bool): self.c.print_timestamps = v
    def set_suppress_blank(self, v: bool): self.c.suppress_blank = v
    def set_suppress_non_speech_tokens(self, v: bool): self.c.suppress_nst = v
    def set_suppress_nst(self, v: bool): self.c.suppress_nst = v
    def set_temperature(self, v: float): self.c.temperature = v
    def set_temperature_inc(self, v: float): self.c.temperature_inc = v
    def set_max_initial_ts(self, v: float): self.c.max_initial_ts = v
    def set_length_penalty(self, v: float): self.c.length_penalty = v
    def set_entropy_thold(self, v: float): self.c.entropy_thold = v
    def set_logprob_thold(self, v: float): self.c.logprob_thold = v
    def set_no_speech_thold(self, v: float): self.c.no_speech_thold = v
    def set_max_tokens(self, v: int): self.c.max_tokens = v
    def set_audio_ctx(self, v: int): self.c.audio_ctx = v
    def set_offset_ms(self, v: int): self.c.offset_ms = v
    def set_duration_ms(self, v: int): self.c.duration_ms = v
    def set_n_max_text_ctx(self, v: int): self.c.n_max_text_ctx = v
    def set_n_threads(self, v: int): self.c.n_threads = v
    def set_tdrz_enable(self, v: bool): self.c.tdrz_enable = v

    def set_filter_logits_callback(self, fn):
        """fn(tokens: list of whisper_token_data, logits: float32 array of n_vocab, writable) -- whisper_logits_filter_callback"""
        def tramp(ctx, _state, tokens, n_tokens, logits, _ud):
            n_vocab = load_library().whisper_n_vocab(ctx)
            fn([tokens[i] for i in range(n_tokens)], np.ctypeslib.as_array(logits, shape=(n_vocab,)))
        self._keep["logits_filter"] = LOGITS_FILTER_CALLBACK(tramp)
        self.c.logits_filter_callback = C.cast(self._keep["logits_filter"], C.c_void_p)

    def set_encoder_begin_callback(self, fn):
        """fn() -> bool, called before each 30 s window is encoded; False ends the call with the segments so far"""
        self._keep["encoder_begin"] = ENCODER_BEGIN_CALLBACK(lambda _c, _s, _u: bool(fn()))
        self.c.encoder_begin_callback = C.cast(self._keep["encoder_begin"], C.c_void_p)

    def set_initial_prompt(self, s: str):
        self._keep["initial_prompt"] = s.encode()
        self.c.initial_prompt = self._keep["initial_prompt"]

    def set_tokens(self, toks: Sequence[int]):
        arr = (C.c_int32 * len(toks))(*toks)
        self._keep["tokens"] = arr
        self.c.prompt_tokens = C.cast(arr, C.POINTER(C.c_int32))
        self.c.prompt_n_tokens = len(toks)


class WhisperContextParameters:
    def __init__(self, gpu_device: int = 0):
        self.c = load_library().whisper_context_default_params()
        self.c.gpu_device = gpu_device

    def use_gpu(self, v: bool): self.c.use_gpu = v; return self
    def gpu_device(self, d: int): self.c.gpu_device = d; return self


class WhisperContext:
    """whisper_rs::WhisperContext::new_with_params"""

    def __init__(self, path: str, params: Optional[WhisperContextParameters] = None):
        self.L = load_library()
        params = params or WhisperContextParameters()
        self.h = self.L.whisper_init_from_file_with_params_no_state(os.fsencode(path), params.c)
        if not self.h:
            raise WhisperError("InitError")

    @classmethod
    def from_buffer(cls, data: bytes, params: Optional[WhisperContextParameters] = None):
        self = cls.__new__(cls)
        self.L = load_library()
        params = params or WhisperContextParameters()
        buf = C.create_string_buffer(data, len(data))
        self.h = self.L.whisper_init_from_buffer_with_params_no_state(C.cast(buf, C.c_void_p), len(data), params.c)
        if not self.h:
            raise WhisperError("InitError")
        return self

    def create_state(self) -> "WhisperState":
        h = self.L.whisper_init_state(self.h)
        if not h:
            raise WhisperError("InitError")
        return WhisperState(self, h)

    def create_batch_state(self, max_utts: int, max_decoders: int = 1, enc_chunk: int = 0) -> "WhisperState":
        h = self.L.wmi_init_state_batch(self.h, max_utts, max_decoders, enc_chunk)
        if not h:
            raise WhisperError("InitError")
        return WhisperState(self, h)

    def pcm_to_mel_device_noalloc(self, pcm_dev_ptr: int, n_samples: int, out_dev_ptr: int, n_frames_out: int, stream: int = 0) -> int:
        """wmi_pcm_to_mel_device_noalloc: device pointers in, work enqueued on `stream`, nothing allocated; returns the C code."""
        return self.L.wmi_pcm_to_mel_device_noalloc(self.h, pcm_dev_ptr, n_samples, out_dev_ptr, n_frames_out, stream or None)

    def n_vocab(self): return self.L.whisper_n_vocab(self.h)
    def n_text_ctx(self): return self.L.whisper_n_text_ctx(self.h)
    def n_audio_ctx(self): return self.L.whisper_n_audio_ctx(self.h)
    def is_multilingual(self): return bool(self.L.whisper_is_multilingual(self.h))
    def n_mels(self): return self.L.whisper_model_n_mels(self.h)
    def n_audio_state(self): return self.L.whisper_model_n_audio_state(self.h)
    def n_text_layer(self): return self.L.whisper_model_n_text_layer(self.h)
    def token_eot(self): return self.L.whisper_token_eot(self.h)
    def token_sot(self): return self.L.whisper_token_sot(self.h)
    def token_beg(self): return self.L.whisper_token_beg(self.h)
    def token_not(self): return self.L.whisper_token_not(self.h)
    def token_prev(self): return self.L.whisper_token_prev(self.h)
    def token_transcribe(self): return self.L.whisper_token_transcribe(self.h)
    def token_translate(self): return self.L.whisper_token_translate(self.h)
    def token_lang(self, lang_id: int): return self.L.whisper_token_lang(self.h, lang_id)

    def token_to_bytes(self, tid: int) -> bytes:
        return C.string_at(self.L.whisper_token_to_str(self.h, tid))

    def tokenize(self, text: str, max_tokens: int = 1024) -> List[int]:
        arr = (C.c_int32 * max_tokens)()
        n = self.L.whisper_tokenize(self.h, text.encode(), arr, max_tokens)
        if n < 0:
            raise WhisperError("InvalidText", n)
        return list(arr[:n])

    def free(self):
        if self.h:
            self.L.whisper_free(self.h)
            self.h = None


@dataclass
class Segment:
    start: float
    end: float
    text: str
    t0: int = 0
    t1: int = 0
    tokens: List[int] = field(default_factory=list)
    plog: List[float] = field(default_factory=list)


class WhisperState:
    """whisper_rs::WhisperState"""

    def __init__(self, ctx: WhisperContext, h):
        self.ctx, self.L, self.h = ctx, ctx.L, h

    def free(self):
        if self.h:
            self.L.whisper_free_state(self.h)
            self.h = None

    # -- full pipeline
    def full(self, params: FullParams, data: np.ndarray) -> int:
        data = np.ascontiguousarray(data, dtype=np.float32)
        if data.size == 0:
            raise WhisperError("NoSamples")
        rc = self.L.whisper_full_with_state(self.ctx.h, self.h, params.c, _fp(data), data.size)
        if rc == -1: raise WhisperError("UnableToCalculateSpectrogram", rc)
        if rc == 7 or rc == -6: raise WhisperError("FailedToEncode", rc)
        if rc == 8 or rc in (-7, -8): raise WhisperError("FailedToDecode", rc)
        if rc != 0: raise WhisperError("GenericError", rc)
        return rc

    def _utterance_codes(self, n: int, allow_failed: bool) -> List[int]:
        """wmi_full_batch returns 0 when the batch ran; an utterance that failed on its own carries its code in wmi_batch_rc.
        Unless the caller asks for the codes (allow_failed), any failed utterance raises: an empty transcript must not pass
        for a result."""
        rcs = [self.L.wmi_batch_rc(self.h, u) for u in range(n)]
        if not allow_failed:
            bad = [(u, rc) for u, rc in enumerate(rcs) if rc != 0]
            if bad:
                raise WhisperError(f"GenericError: utterance {bad[0][0]} of the batch failed ({len(bad)} in all)", bad[0][1])
        return rcs

    def full_batch(self, params: FullParams, pcm_list: Sequence[np.ndarray], allow_failed: bool = False) -> List[int]:
        """whisper_full over a batch of host clips. Returns the per-utterance codes (all 0 unless allow_failed)."""
        arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in pcm_list]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        ns = (C.c_int * len(arrs))(*[a.size for a in arrs])
        rc = self.L.wmi_full_batch(self.ctx.h, self.h, params.c, ptrs, ns, len(arrs), 0)
        if rc != 0:
            raise WhisperError("GenericError", rc)
        return self._utterance_codes(len(arrs), allow_failed)

    def full_batch_device(self, params: FullParams, dev_ptrs: Sequence[int], n_samples: Sequence[int], allow_failed: bool = False) -> List[int]:
        ptrs = (C.c_void_p * len(dev_ptrs))(*dev_ptrs)
        ns = (C.c_int * len(dev_ptrs))(*n_samples)
        rc = self.L.wmi_full_batch(self.ctx.h, self.h, params.c, ptrs, ns, len(dev_ptrs), 1)
        if rc != 0:
            raise WhisperError("GenericError", rc)
        return self._utterance_codes(len(dev_ptrs), allow_failed)

    def full_n_segments(self) -> int: return self.L.whisper_full_n_segments_from_state(self.h)
    def full_lang_id_from_state(self) -> int: return self.L.whisper_full_lang_id_from_state(self.h)
    def full_get_segment_t0(self, i: int) -> int: return self.L.whisper_full_get_segment_t0_from_state(self.h, i)
    def full_get_segment_t1(self, i: int) -> int: return self.L.whisper_full_get_segment_t1_from_state(self.h, i)
    def full_get_segment_bytes(self, i: int) -> bytes: return C.string_at(self.L.whisper_full_get_segment_text_from_state(self.h, i))
    def full_get_segment_text(self, i: int) -> str: return self.full_get_segment_bytes(i).decode("utf-8", errors="replace")
    def full_n_tokens(self, i: int) -> int: return self.L.whisper_full_n_tokens_from_state(self.h, i)
    def full_get_token_id(self, i: int, j: int) -> int: return self.L.whisper_full_get_token_id_from_state(self.h, i, j)
    def full_get_token_data(self, i: int, j: int) -> whisper_token_data: return self.L.whisper_full_get_token_data_from_state(self.h, i, j)

    def batch_segments(self, utt: int) -> List[Segment]:
        out = []
        for i in range(self.L.wmi_batch_n_segments(self.h, utt)):
            t0, t1 = self.L.wmi_batch_segment_t0(self.h, utt, i), self.L.wmi_batch_segment_t1(self.h, utt, i)
            toks = [self.L.wmi_batch_token_data(self.h, utt, i, j) for j in range(self.L.wmi_batch_n_tokens(self.h, utt, i))]
            text = C.string_at(self.L.wmi_batch_segment_text(self.h, utt, i)).decode("utf-8", errors="replace")
            out.append(Segment(t0 / 100.0, t1 / 100.0, text, t0, t1, [t.id for t in toks], [t.plog for t in toks]))
        return out

    def batch_tokens(self, utt: int) -> List[int]:
        return [t for s in self.batch_segments(utt) for t in s.tokens]

    def batch_n_fallbacks(self, utt: int) -> int: return self.L.wmi_batch_n_fallbacks(self.h, utt)
    def batch_rc(self, utt: int) -> int: return self.L.wmi_batch_rc(self.h, utt)
    def batch_lang_id(self, utt: int) -> int: return self.L.wmi_batch_lang_id(self.h, utt)

    # -- low-level stages
    def pcm_to_mel(self, pcm: np.ndarray, threads: int = 1):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        if self.L.whisper_pcm_to_mel_with_state(self.ctx.h, self.h, _fp(pcm), pcm.size, threads) != 0:
            raise WhisperError("UnableToCalculateSpectrogram", -1)

    def get_mel(self, n_frames: int, utt: int = 0) -> np.ndarray:
        out = np.empty((self.ctx.n_mels(), n_frames), dtype=np.float32)
        if self.L.wmi_get_mel(self.h, utt, _fp(out), n_frames) != 0:
            raise WhisperError("GenericError")
        return out

    def n_len(self) -> int: return self.L.whisper_n_len_from_state(self.h)

    def set_mel(self, mel: np.ndarray):
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        if self.L.whisper_set_mel_with_state(self.ctx.h, self.h, _fp(mel), mel.shape[1], mel.shape[0]) != 0:
            raise WhisperError("InvalidMelBands")

    def encode(self, offset: int = 0, threads: int = 1):
        if self.L.whisper_encode_with_state(self.ctx.h, self.h, offset, threads) != 0:
            raise WhisperError("FailedToEncode", -1)

    def encoder_output(self) -> np.ndarray:
        out = np.empty((self.ctx.n_audio_ctx(), self.ctx.n_audio_state()), dtype=np.float32)
        if self.L.wmi_get_encoder_output(self.h, _fp(out)) != 0:
            raise WhisperError("GenericError")
        return out

    def cross_kv(self, layer: int):
        shp = (self.ctx.n_audio_ctx(), self.ctx.n_audio_state())
        k, v = np.empty(shp, np.float32), np.empty(shp, np.float32)
        if self.L.wmi_get_cross_kv(self.h, layer, _fp(k), _fp(v)) != 0:
            raise WhisperError("GenericError")
        return k, v

    def decode(self, tokens: Sequence[int], n_past: int = 0, threads: int = 1) -> np.ndarray:
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        if self.L.whisper_decode_with_state(self.ctx.h, self.h, t.ctypes.data_as(C.POINTER(C.c_int32)), len(t), n_past, threads) != 0:
            raise WhisperError("FailedToDecode", 1)
        return np.ctypeslib.as_array(self.L.whisper_get_logits_from_state(self.h), shape=(self.ctx.n_vocab(),)).copy()

    def decode_logits_all(self, tokens: Sequence[int], n_past: int = 0) -> np.ndarray:
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        out = np.empty((len(t), self.ctx.n_vocab()), dtype=np.float32)
        if self.L.wmi_decode_logits_all(self.ctx.h, self.h, t.ctypes.data_as(C.POINTER(C.c_int32)), len(t), n_past, _fp(out)) != 0:
            raise WhisperError("FailedToDecode", 1)
        return out

    def lang_detect(self, offset_ms: int = 0):
        probs = np.zeros(self.L.whisper_lang_max_id() + 1, dtype=np.float32)
        lid = self.L.whisper_lang_auto_detect_with_state(self.ctx.h, self.h, offset_ms, 1, _fp(probs))
        if lid < 0:
            raise WhisperError("GenericError", lid)
        return lid, probs

    # -- knobs / stats
    def set_force_tokens(self, n: int): self.L.wmi_set_force_tokens(self.h, n)
    def set_beam_candidates_sampled(self, v: bool): self.L.wmi_set_beam_candidates_sampled(self.h, int(v))
    def set_profiling(self, on: bool): self.L.wmi_set_profiling(self.h, int(on))
    def set_pipeline_groups(self, n: int): self.L.wmi_set_pipeline_groups(self.h, int(n))

    def set_window_cache(self, max_windows: int): self.L.wmi_set_window_cache(self.h, int(max_windows))

    def window_cache_stats(self):
        h, s_ = C.c_longlong(0), C.c_longlong(0)
        self.L.wmi_get_window_cache_stats(self.h, C.byref(h), C.byref(s_))
        return {"hits": h.value, "stores": s_.value}
    def set_encoder_lanes(self, n: int):
        if self.L.wmi_set_encoder_lanes(self.h, int(n)) != 0: raise WhisperError("GenericError", -1)
    def set_profiling_classes(self, names):
        """Profile only the named kernel classes (e.g. ["enc_gemm"]); an empty list turns profiling off."""
        mask = 0
        for n in names: mask |= 1 << KERNEL_CLASSES.index(n)
        self.L.wmi_set_profiling_classes(self.h, mask)
    def reset_kernel_stats(self): self.L.wmi_reset_kernel_stats(self.h)
    def set_decoder_lanes(self, n: int):
        """Row lanes of a decoder pass (1 or 2 HIP streams)."""
        if self.L.wmi_set_decoder_lanes(self.h, int(n)) != 0: raise WhisperError("GenericError", -1)
    def set_pass_graphs(self, on: bool):
        """Decoder passes of a repeated shape replayed as HIP graphs (default on)."""
        self.L.wmi_set_pass_graphs(self.h, 1 if on else 0)
    def set_quant_decode(self, on: bool):
        """Few-row decode passes read block-quantised decoder weights as blocks (default) or their fp16 copies."""
        if self.L.wmi_set_quant_decode(self.h, 1 if on else 0) != 0: raise WhisperError("GenericError", -1)
    def host_timings(self):
        a = (C.c_double * 6)()
        self.L.wmi_get_host_timings(self.h, a)
        return dict(zip(("mel_ms", "encode_launch_ms", "decode_ms", "decode_launch_ms", "decode_wait_ms", "passes"), list(a)))

    def kernel_stats(self):
        arr = (wmi_kernel_stat * len(KERNEL_CLASSES))()
        if self.L.wmi_get_kernel_stats(self.h, arr) != 0:
            raise WhisperError("GenericError")
        return {n: {"ms": arr[i].ms, "work": arr[i].work, "launches": arr[i].launches} for i, n in enumerate(KERNEL_CLASSES)}


# ----------------------------------------------------------------------------------------------
# transcribe-rs mirror (the type Handy holds in LoadedEngine::Whisper, transcription.rs:31-35)
# ----------------------------------------------------------------------------------------------
@dataclass
class WhisperInferenceParams:
    """transcribe_rs::engines::whisper::WhisperInferenceParams (Handy sets language and translate,
    transcription.rs:407-411, and leaves the rest at Default)."""
    language: Optional[str] = None
    translate: bool = False
    print_special: bool = False
    print_progress: bool = False
    print_realtime: bool = False
    print_timestamps: bool = False
    suppress_blank: bool = True
    suppress_non_speech_tokens: bool = True
    no_speech_thold: float = 0.2
    initial_prompt: Optional[str] = None
    # upstream builds FullParams with BeamSearch { beam_size: 3, patience: -1.0 } (recalled; not in the reference tree)
    beam_size: int = 3


@dataclass
class TranscriptionResult:
    text: str
    segments: Optional[List[Segment]] = None


class WhisperEngine:
    """transcribe_rs::engines::whisper::WhisperEngine"""

    def __init__(self):
        self.ctx: Optional[WhisperContext] = None
        self.state: Optional[WhisperState] = None

    def load_model(self, path: str, gpu_device: int = 0):
        self.unload_model()
        self.ctx = WhisperContext(path, WhisperContextParameters(gpu_device))
        self.state = self.ctx.create_state()

    def unload_model(self):
        if self.state:
            self.state.free(); self.state = None
        if self.ctx:
            self.ctx.free(); self.ctx = None

    def _full_params(self, p: WhisperInferenceParams) -> FullParams:
        fp = FullParams(SamplingStrategy.BeamSearch(p.beam_size, -1.0) if p.beam_size > 1 else SamplingStrategy.Greedy(1))
        fp.set_language(p.language)
        fp.set_translate(p.translate)
        fp.set_print_special(p.print_special); fp.set_print_progress(p.print_progress)
        fp.set_print_realtime(p.print_realtime); fp.set_print_timestamps(p.print_timestamps)
        fp.set_suppress_blank(p.suppress_blank); fp.set_suppress_non_speech_tokens(p.suppress_non_speech_tokens)
        fp.set_no_speech_thold(p.no_speech_thold)
        if p.initial_prompt:
            fp.set_initial_prompt(p.initial_prompt)
        return fp

    def transcribe_samples(self, samples: np.ndarray, params: Optional[WhisperInferenceParams] = None) -> TranscriptionResult:
        if self.state is None:
            raise WhisperError("ModelNotLoaded")
        fp = self._full_params(params or WhisperInferenceParams())
        self.state.full(fp, samples)
        segs = self.state.batch_segments(0)
        return TranscriptionResult(text="".join(s.text for s in segs).strip(), segments=segs)
