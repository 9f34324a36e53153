//! FFI declarations of include/whisper.h (the subset whisper-rs 0.13.x calls), laid out exactly as the
//! C header. Field order and types MUST stay in lock-step with include/whisper.h: both structs cross
//! the ABI by value. tests/test_cpu_host.py checks the C side's offsets; `cargo test` here checks sizes.
#![allow(non_camel_case_types, non_upper_case_globals, non_snake_case)]
use std::os::raw::{c_char, c_float, c_int, c_uint, c_void};

pub type whisper_token = i32;
pub type whisper_pos = i32;
pub type whisper_seq_id = i32;

#[repr(C)] pub struct whisper_context { _private: [u8; 0] }
#[repr(C)] pub struct whisper_state { _private: [u8; 0] }
#[repr(C)] pub struct wmi_node { _private: [u8; 0] }

/// include/wmi.h: per-kernel-class device time (HIP events on the engine's stream)
#[repr(C)] #[derive(Clone, Copy, Default)] pub struct wmi_kernel_stat { pub ms: f64, pub work: f64, pub launches: i64 }
pub const WMI_K_COUNT: usize = 7;

/// include/wmi.h: one utterance's outcome of a node-level job (fixed size: it is what the GPUs exchange)
pub const WMI_NODE_MAX_TOKENS: usize = 240;
#[repr(C)] #[derive(Clone, Copy)]
pub struct wmi_node_record {
    pub utt: c_int, pub rc: c_int, pub device: c_int, pub lang_id: c_int, pub n_segments: c_int, pub n_tokens: c_int,
    pub t0: i64, pub t1: i64,
    pub tokens: [whisper_token; WMI_NODE_MAX_TOKENS],
}
pub type wmi_node_batch_fn = Option<unsafe extern "C" fn(*mut c_void, c_int, c_int, c_int, *mut wmi_node_record) -> c_int>;

pub const WHISPER_SAMPLE_RATE: u32 = 16000;
pub const WHISPER_N_FFT: u32 = 400;
pub const WHISPER_HOP_LENGTH: u32 = 160;
pub const WHISPER_CHUNK_SIZE: u32 = 30;

pub type whisper_alignment_heads_preset = c_int;
pub const whisper_alignment_heads_preset_WHISPER_AHEADS_NONE: whisper_alignment_heads_preset = 0;

#[repr(C)] #[derive(Clone, Copy)] pub struct whisper_ahead { pub n_text_layer: c_int, pub n_head: c_int }
#[repr(C)] #[derive(Clone, Copy)] pub struct whisper_aheads { pub n_heads: usize, pub heads: *const whisper_ahead }

#[repr(C)] #[derive(Clone, Copy)]
pub struct whisper_context_params {
    pub use_gpu: bool,
    pub flash_attn: bool,
    pub gpu_device: c_int,
    pub dtw_token_timestamps: bool,
    pub dtw_aheads_preset: whisper_alignment_heads_preset,
    pub dtw_n_top: c_int,
    pub dtw_aheads: whisper_aheads,
    pub dtw_mem_size: usize,
}

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct whisper_token_data {
    pub id: whisper_token, pub tid: whisper_token,
    pub p: c_float, pub plog: c_float, pub pt: c_float, pub ptsum: c_float,
    pub t0: i64, pub t1: i64, pub t_dtw: i64,
    pub vlen: c_float,
}

pub type whisper_sampling_strategy = c_int;
pub const whisper_sampling_strategy_WHISPER_SAMPLING_GREEDY: whisper_sampling_strategy = 0;
pub const whisper_sampling_strategy_WHISPER_SAMPLING_BEAM_SEARCH: whisper_sampling_strategy = 1;

pub type whisper_new_segment_callback = Option<unsafe extern "C" fn(*mut whisper_context, *mut whisper_state, c_int, *mut c_void)>;
pub type whisper_progress_callback = Option<unsafe extern "C" fn(*mut whisper_context, *mut whisper_state, c_int, *mut c_void)>;
pub type whisper_encoder_begin_callback = Option<unsafe extern "C" fn(*mut whisper_context, *mut whisper_state, *mut c_void) -> bool>;
pub type ggml_abort_callback = Option<unsafe extern "C" fn(*mut c_void) -> bool>;
pub type whisper_logits_filter_callback =
    Option<unsafe extern "C" fn(*mut whisper_context, *mut whisper_state, *const whisper_token_data, c_int, *mut c_float, *mut c_void)>;
pub type ggml_log_callback = Option<unsafe extern "C" fn(c_int, *const c_char, *mut c_void)>;

#[repr(C)] #[derive(Clone, Copy)] pub struct whisper_grammar_element { pub type_: c_int, pub value: u32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct whisper_full_params__bindgen_ty_1 { pub best_of: c_int }
#[repr(C)] #[derive(Clone, Copy)] pub struct whisper_full_params__bindgen_ty_2 { pub beam_size: c_int, pub patience: c_float }

#[repr(C)] #[derive(Clone, Copy)]
pub struct whisper_full_params {
    pub strategy: whisper_sampling_strategy,
    pub n_threads: c_int,
    pub n_max_text_ctx: c_int,
    pub offset_ms: c_int,
    pub duration_ms: c_int,
    pub translate: bool,
    pub no_context: bool,
    pub no_timestamps: bool,
    pub single_segment: bool,
    pub print_special: bool,
    pub print_progress: bool,
    pub print_realtime: bool,
    pub print_timestamps: bool,
    pub token_timestamps: bool,
    pub thold_pt: c_float,
    pub thold_ptsum: c_float,
    pub max_len: c_int,
    pub split_on_word: bool,
    pub max_tokens: c_int,
    pub debug_mode: bool,
    pub audio_ctx: c_int,
    pub tdrz_enable: bool,
    pub suppress_regex: *const c_char,
    pub initial_prompt: *const c_char,
    pub prompt_tokens: *const whisper_token,
    pub prompt_n_tokens: c_int,
    pub language: *const c_char,
    pub detect_language: bool,
    pub suppress_blank: bool,
    /// `suppress_nst` in newer whisper.cpp headers; one bool at this offset either way
    pub suppress_non_speech_tokens: bool,
    pub temperature: c_float,
    pub max_initial_ts: c_float,
    pub length_penalty: c_float,
    pub temperature_inc: c_float,
    pub entropy_thold: c_float,
    pub logprob_thold: c_float,
    pub no_speech_thold: c_float,
    pub greedy: whisper_full_params__bindgen_ty_1,
    pub beam_search: whisper_full_params__bindgen_ty_2,
    pub new_segment_callback: whisper_new_segment_callback,
    pub new_segment_callback_user_data: *mut c_void,
    pub progress_callback: whisper_progress_callback,
    pub progress_callback_user_data: *mut c_void,
    pub encoder_begin_callback: whisper_encoder_begin_callback,
    pub encoder_begin_callback_user_data: *mut c_void,
    pub abort_callback: ggml_abort_callback,
    pub abort_callback_user_data: *mut c_void,
    pub logits_filter_callback: whisper_logits_filter_callback,
    pub logits_filter_callback_user_data: *mut c_void,
    pub grammar_rules: *mut *const whisper_grammar_element,
    pub n_grammar_rules: usize,
    pub i_start_rule: usize,
    pub grammar_penalty: c_float,
}

extern "C" {
    pub fn whisper_context_default_params() -> whisper_context_params;
    pub fn whisper_init_from_file_with_params(path_model: *const c_char, params: whisper_context_params) -> *mut whisper_context;
    pub fn whisper_init_from_file_with_params_no_state(path_model: *const c_char, params: whisper_context_params) -> *mut whisper_context;
    pub fn whisper_init_from_buffer_with_params(buffer: *mut c_void, buffer_size: usize, params: whisper_context_params) -> *mut whisper_context;
    pub fn whisper_init_from_buffer_with_params_no_state(buffer: *mut c_void, buffer_size: usize, params: whisper_context_params) -> *mut whisper_context;
    pub fn whisper_init_state(ctx: *mut whisper_context) -> *mut whisper_state;
    pub fn whisper_free(ctx: *mut whisper_context);
    pub fn whisper_free_state(state: *mut whisper_state);
    pub fn whisper_free_params(params: *mut whisper_full_params);
    pub fn whisper_free_context_params(params: *mut whisper_context_params);

    pub fn whisper_pcm_to_mel_with_state(ctx: *mut whisper_context, state: *mut whisper_state, samples: *const c_float, n_samples: c_int, n_threads: c_int) -> c_int;
    pub fn whisper_set_mel_with_state(ctx: *mut whisper_context, state: *mut whisper_state, data: *const c_float, n_len: c_int, n_mel: c_int) -> c_int;
    pub fn whisper_encode_with_state(ctx: *mut whisper_context, state: *mut whisper_state, offset: c_int, n_threads: c_int) -> c_int;
    pub fn whisper_decode_with_state(ctx: *mut whisper_context, state: *mut whisper_state, tokens: *const whisper_token, n_tokens: c_int, n_past: c_int, n_threads: c_int) -> c_int;
    pub fn whisper_get_logits_from_state(state: *mut whisper_state) -> *mut c_float;
    pub fn whisper_tokenize(ctx: *mut whisper_context, text: *const c_char, tokens: *mut whisper_token, n_max_tokens: c_int) -> c_int;
    pub fn whisper_token_count(ctx: *mut whisper_context, text: *const c_char) -> c_int;

    pub fn whisper_lang_max_id() -> c_int;
    pub fn whisper_lang_id(lang: *const c_char) -> c_int;
    pub fn whisper_lang_str(id: c_int) -> *const c_char;
    pub fn whisper_lang_str_full(id: c_int) -> *const c_char;
    pub fn whisper_lang_auto_detect_with_state(ctx: *mut whisper_context, state: *mut whisper_state, offset_ms: c_int, n_threads: c_int, lang_probs: *mut c_float) -> c_int;

    pub fn whisper_n_len_from_state(state: *mut whisper_state) -> c_int;
    pub fn whisper_n_vocab(ctx: *mut whisper_context) -> c_int;
    pub fn whisper_n_text_ctx(ctx: *mut whisper_context) -> c_int;
    pub fn whisper_n_audio_ctx(ctx: *mut whisper_context) -> c_int;
    pub fn whisper_is_multilingual(ctx: *mut whisper_context) -> c_int;
    pub fn whisper_model_n_vocab(ctx: *mut whisper_context) -> c_int;
    pub fn whisper_model_n_audio_ctx(ctx: *mut whisper_context) -> c_int;
    pub fn whisper_model_n_audio_state(ctx: *mut whisper_context) -> c_int;
    pub fn whisper_model_n_audio_head(ctx: *mut whisper_context) -> c_int;
    pub fn whisper_model_n_audio_layer(ctx: *mut whisper_context) -> c_int;
    pub fn whisper_model_n_text_ctx(ctx: *mut whisper_context) -> c_int;
    pub fn whisper_model_n_text_state(ctx: *mut whisper_context) -> c_int;
    pub fn whisper_model_n_text_head(ctx: *mut whisper_context) -> c_int;
    pub fn whisper_model_n_text_layer(ctx: *mut whisper_context) -> c_int;
    pub fn whisper_model_n_mels(ctx: *mut whisper_context) -> c_int;
    pub fn whisper_model_ftype(ctx: *mut whisper_context) -> c_int;
    pub fn whisper_model_type(ctx: *mut whisper_context) -> c_int;
    pub fn whisper_model_type_readable(ctx: *mut whisper_context) -> *const c_char;

    pub fn whisper_token_to_str(ctx: *mut whisper_context, token: whisper_token) -> *const c_char;
    pub fn whisper_token_eot(ctx: *mut whisper_context) -> whisper_token;
    pub fn whisper_token_sot(ctx: *mut whisper_context) -> whisper_token;
    pub fn whisper_token_solm(ctx: *mut whisper_context) -> whisper_token;
    pub fn whisper_token_prev(ctx: *mut whisper_context) -> whisper_token;
    pub fn whisper_token_nosp(ctx: *mut whisper_context) -> whisper_token;
    pub fn whisper_token_not(ctx: *mut whisper_context) -> whisper_token;
    pub fn whisper_token_beg(ctx: *mut whisper_context) -> whisper_token;
    pub fn whisper_token_lang(ctx: *mut whisper_context, lang_id: c_int) -> whisper_token;
    pub fn whisper_token_translate(ctx: *mut whisper_context) -> whisper_token;
    pub fn whisper_token_transcribe(ctx: *mut whisper_context) -> whisper_token;

    pub fn whisper_print_timings(ctx: *mut whisper_context);
    pub fn whisper_reset_timings(ctx: *mut whisper_context);
    pub fn whisper_print_system_info() -> *const c_char;
    pub fn whisper_log_set(log_callback: ggml_log_callback, user_data: *mut c_void);

    pub fn whisper_full_default_params(strategy: whisper_sampling_strategy) -> whisper_full_params;
    pub fn whisper_full_default_params_by_ref(strategy: whisper_sampling_strategy) -> *mut whisper_full_params;
    pub fn whisper_full(ctx: *mut whisper_context, params: whisper_full_params, samples: *const c_float, n_samples: c_int) -> c_int;
    pub fn whisper_full_with_state(ctx: *mut whisper_context, state: *mut whisper_state, params: whisper_full_params, samples: *const c_float, n_samples: c_int) -> c_int;
    pub fn whisper_full_parallel(ctx: *mut whisper_context, params: whisper_full_params, samples: *const c_float, n_samples: c_int, n_processors: c_int) -> c_int;

    pub fn whisper_full_n_segments_from_state(state: *mut whisper_state) -> c_int;
    pub fn whisper_full_lang_id_from_state(state: *mut whisper_state) -> c_int;
    pub fn whisper_full_get_segment_t0_from_state(state: *mut whisper_state, i_segment: c_int) -> i64;
    pub fn whisper_full_get_segment_t1_from_state(state: *mut whisper_state, i_segment: c_int) -> i64;
    pub fn whisper_full_get_segment_speaker_turn_next_from_state(state: *mut whisper_state, i_segment: c_int) -> bool;
    pub fn whisper_full_get_segment_text_from_state(state: *mut whisper_state, i_segment: c_int) -> *const c_char;
    pub fn whisper_full_n_tokens_from_state(state: *mut whisper_state, i_segment: c_int) -> c_int;
    pub fn whisper_full_get_token_text_from_state(ctx: *mut whisper_context, state: *mut whisper_state, i_segment: c_int, i_token: c_int) -> *const c_char;
    pub fn whisper_full_get_token_id_from_state(state: *mut whisper_state, i_segment: c_int, i_token: c_int) -> whisper_token;
    pub fn whisper_full_get_token_data_from_state(state: *mut whisper_state, i_segment: c_int, i_token: c_int) -> whisper_token_data;
    pub fn whisper_full_get_token_p_from_state(state: *mut whisper_state, i_segment: c_int, i_token: c_int) -> c_float;

    // wmi.h: batch extension (no upstream counterpart)
    pub fn wmi_init_state_batch(ctx: *mut whisper_context, max_utts: c_int, max_decoders: c_int, enc_chunk: c_int) -> *mut whisper_state;
    /// Returns 0 when the batch ran; an utterance that failed on its own carries its code in `wmi_batch_rc` (check it
    /// for every utterance: a failed one has no segments, which otherwise reads as an empty transcript).
    pub fn wmi_full_batch(ctx: *mut whisper_context, state: *mut whisper_state, params: whisper_full_params, samples: *const *const c_float,
                          n_samples: *const c_int, n_utts: c_int, samples_on_device: c_int) -> c_int;
    pub fn wmi_batch_n_utts(state: *mut whisper_state) -> c_int;
    pub fn wmi_batch_n_segments(state: *mut whisper_state, utt: c_int) -> c_int;
    pub fn wmi_batch_segment_t0(state: *mut whisper_state, utt: c_int, seg: c_int) -> i64;
    pub fn wmi_batch_segment_t1(state: *mut whisper_state, utt: c_int, seg: c_int) -> i64;
    pub fn wmi_batch_segment_text(state: *mut whisper_state, utt: c_int, seg: c_int) -> *const c_char;
    pub fn wmi_batch_n_tokens(state: *mut whisper_state, utt: c_int, seg: c_int) -> c_int;
    pub fn wmi_batch_token_data(state: *mut whisper_state, utt: c_int, seg: c_int, tok: c_int) -> whisper_token_data;
    pub fn wmi_batch_lang_id(state: *mut whisper_state, utt: c_int) -> c_int;
    pub fn wmi_batch_rc(state: *mut whisper_state, utt: c_int) -> c_int;
    pub fn wmi_batch_n_fallbacks(state: *mut whisper_state, utt: c_int) -> c_int;
    pub fn wmi_set_force_tokens(state: *mut whisper_state, n: c_int);
    /// Remember finished interior 30 s windows between calls (growing recordings: live coaching); 0 = off.
    pub fn wmi_set_window_cache(state: *mut whisper_state, max_windows: c_int);
    pub fn wmi_get_window_cache_stats(state: *mut whisper_state, hits: *mut i64, stores: *mut i64);
    pub fn wmi_set_pipeline_groups(state: *mut whisper_state, n: c_int);
    pub fn wmi_set_encoder_lanes(state: *mut whisper_state, n: c_int) -> c_int;
    pub fn wmi_set_decoder_lanes(state: *mut whisper_state, n: c_int) -> c_int;
    pub fn wmi_set_pass_graphs(state: *mut whisper_state, on: c_int);
    pub fn wmi_set_beam_candidates_sampled(state: *mut whisper_state, sampled: c_int);
    pub fn wmi_set_quant_resident(on: c_int);
    pub fn wmi_set_quant_decode(state: *mut whisper_state, on: c_int) -> c_int;
    pub fn wmi_mel_batch_device(ctx: *mut whisper_context, state: *mut whisper_state, pcm_dev: *const c_float, offsets: *const i64,
                                n_samples: *const c_int, n_utts: c_int) -> c_int;
    pub fn wmi_mel_batch_device_normalized(ctx: *mut whisper_context, state: *mut whisper_state, pcm_dev: *const c_float, offsets: *const i64,
                                           n_samples: *const c_int, n_utts: c_int, mel_out_dev: *mut c_float, n_frames_out: c_int) -> c_int;
    /// The log-mel frontend without any allocation (callers under HIP graph capture, serving loops).
    pub fn wmi_pcm_to_mel_device_noalloc(ctx: *mut whisper_context, pcm_dev: *const c_float, n_samples: c_int, mel_out_dev: *mut c_float,
                                         n_frames_out: c_int, stream: *mut c_void) -> c_int;
    pub fn wmi_version() -> *const c_char;

    // wmi.h: profiling
    pub fn wmi_set_profiling(state: *mut whisper_state, enabled: c_int);
    pub fn wmi_set_profiling_classes(state: *mut whisper_state, class_mask: c_uint);
    pub fn wmi_get_kernel_stats(state: *mut whisper_state, out: *mut wmi_kernel_stat) -> c_int;
    pub fn wmi_reset_kernel_stats(state: *mut whisper_state);
    pub fn wmi_get_host_timings(state: *mut whisper_state, out: *mut f64) -> c_int;

    // wmi.h: inspection hooks of the parity tests
    pub fn wmi_get_mel(state: *mut whisper_state, utt: c_int, out: *mut c_float, n_frames: c_int) -> c_int;
    pub fn wmi_get_encoder_output(state: *mut whisper_state, out: *mut c_float) -> c_int;
    pub fn wmi_get_cross_kv(state: *mut whisper_state, layer: c_int, k_out: *mut c_float, v_out: *mut c_float) -> c_int;
    pub fn wmi_decode_logits_all(ctx: *mut whisper_context, state: *mut whisper_state, tokens: *const whisper_token, n_tokens: c_int,
                                 n_past: c_int, logits_out: *mut c_float) -> c_int;
    pub fn wmi_debug_gemm(m: c_int, n: c_int, k: c_int, a: *const c_float, w: *const c_float, c: *mut c_float) -> c_int;
    pub fn wmi_debug_gemm_gelu(m: c_int, n: c_int, k: c_int, a: *const c_float, w: *const c_float, bias: *const c_float, out: *mut u16) -> c_int;
    pub fn wmi_debug_expand_blocks(ggml_type: c_int, blocks: *const c_void, n_blocks: i64, out: *mut u16) -> c_int;
    pub fn wmi_debug_quant_rows(ggml_type: c_int, blocks: *const c_void, n: c_int, k: c_int, a_f16: *const u16, rows: c_int,
                                out_blocks: *mut c_float, out_f16: *mut c_float) -> c_int;
    pub fn wmi_debug_enc_attention(b: c_int, h: c_int, t: c_int, q: *const u16, k: *const u16, v: *const u16, out: *mut u16) -> c_int;

    // wmi.h: device memory of the HIP runtime the library is linked against (current device)
    pub fn wmi_dev_alloc(bytes: usize) -> *mut c_void;
    pub fn wmi_dev_free(p: *mut c_void);
    pub fn wmi_dev_upload(dst_dev: *mut c_void, src_host: *const c_void, bytes: usize) -> c_int;
    pub fn wmi_dev_download(dst_host: *mut c_void, src_dev: *const c_void, bytes: usize) -> c_int;
    pub fn wmi_dev_memset(dst_dev: *mut c_void, byte: c_int, bytes: usize) -> c_int;
    pub fn wmi_dev_sync() -> c_int;
    pub fn wmi_dev_free_bytes() -> i64;
    pub fn wmi_dev_stream_create() -> *mut c_void;
    pub fn wmi_dev_stream_sync(stream: *mut c_void) -> c_int;
    pub fn wmi_dev_stream_destroy(stream: *mut c_void);

    // wmi.h: node level -- one engine per GPU, data-parallel by utterance, result records gathered over RCCL
    pub fn wmi_ctx_for_device(path_model: *const c_char, device: c_int) -> *mut whisper_context;
    pub fn wmi_node_init(path_model: *const c_char, devices: *const c_int, n_devices: c_int, batch: c_int, max_decoders: c_int,
                         enc_chunk: c_int) -> *mut wmi_node;
    pub fn wmi_node_free(node: *mut wmi_node);
    pub fn wmi_node_n_devices(node: *mut wmi_node) -> c_int;
    pub fn wmi_node_state(node: *mut wmi_node, i: c_int) -> *mut whisper_state;
    pub fn wmi_node_context(node: *mut wmi_node, i: c_int) -> *mut whisper_context;
    pub fn wmi_node_run(node: *mut wmi_node, params: whisper_full_params, samples: *const *const c_float, n_samples: *const c_int,
                        n_utts: c_int, records_out: *mut wmi_node_record) -> c_int;
    pub fn wmi_node_last_gather(node: *mut wmi_node) -> c_int;
    pub fn wmi_node_last_batches(node: *mut wmi_node, i: c_int) -> c_int;
    pub fn wmi_node_run_custom(n_workers: c_int, n_utts: c_int, batch: c_int, func: wmi_node_batch_fn, user: *mut c_void,
                               records_out: *mut wmi_node_record, batches_per_worker: *mut c_int) -> c_int;

    // wmi.h: the recorder's resample + VAD gate for bulk files in device memory (audio_toolkit/audio/resampler.rs, vad/smoothed.rs)
    pub fn wmi_ingest_n_frames(in_hz: c_int, n_in: i64) -> i64;
    pub fn wmi_ingest_resample(pcm_dev: *const c_float, n_in: i64, in_hz: c_int, frames_dev: *mut c_float, n_frames_cap: i64, stream: *mut c_void) -> c_int;
    pub fn wmi_ingest_gate_plan(is_voice: *const u8, n_frames: i64, prefill: c_int, hangover: c_int, onset: c_int, src_out: *mut i64, cap: i64) -> i64;
    pub fn wmi_ingest_gather(frames_dev: *const c_float, n_frames: i64, src: *const i64, n_src: i64, out_dev: *mut c_float, out_cap: i64,
                             n_out: *mut i64, stream: *mut c_void) -> c_int;
}

/// whisper-rs installs a ggml log hook through this symbol; the engine has no ggml, so it is a no-op
/// kept on the Rust side (the shared library must not export any `ggml_*` symbol: Handy already isolates llama.cpp in a sidecar because of ggml symbol clashes).
#[no_mangle]
pub unsafe extern "C" fn ggml_log_set(_log_callback: ggml_log_callback, _user_data: *mut c_void) {}

#[cfg(test)]
mod tests {
    use super::*;
    #[test]
    fn struct_sizes_match_the_c_header() {
        // values printed by tests/test_cpu_host.py::test_struct_layout_matches_the_c_header on x86-64
        assert_eq!(std::mem::size_of::<whisper_token_data>(), 56);
        assert_eq!(std::mem::size_of::<whisper_context_params>(), 48);
        assert_eq!(std::mem::size_of::<whisper_full_params>(), 264);
        assert_eq!(std::mem::size_of::<wmi_node_record>(), 40 + 4 * WMI_NODE_MAX_TOKENS);
        assert_eq!(std::mem::size_of::<wmi_kernel_stat>(), 24);
    }
}
