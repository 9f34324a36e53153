/*
 * whisper.h -- C ABI of libwhisper_mi355.so, the MI355X-native Whisper engine.
 *
 * This is the boundary Handy reaches through transcribe-rs 0.2.1 -> whisper-rs 0.13.2 ->
 * whisper-rs-sys 0.11.1 (reference: src-tauri/Cargo.lock:7137-7155, 7830-7848; the only call
 * sites are src-tauri/src/managers/transcription.rs:228-229 (load) and :413-415 (transcribe)).
 * whisper-rs-sys binds whisper.cpp's whisper.h with bindgen; the declarations below keep that
 * header's names, argument order, by-value structs and error conventions (NULL on failed init,
 * non-zero int on failed run, nothing ever aborts) for the subset whisper-rs calls, so a
 * replacement -sys crate can link this library instead of the bundled whisper.cpp.
 * The upstream header is not part of the reference tree; struct layouts are restated from the
 * whisper.cpp 1.7.x line and must be re-checked against the exact pinned copy when one is at hand.
 */
#ifndef WHISPER_H
#define WHISPER_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WHISPER_API __attribute__((visibility("default")))

#define WHISPER_SAMPLE_RATE 16000
#define WHISPER_N_FFT       400
#define WHISPER_HOP_LENGTH  160
#define WHISPER_CHUNK_SIZE  30

struct whisper_context;
struct whisper_state;

typedef int32_t whisper_pos;
typedef int32_t whisper_token;
typedef int32_t whisper_seq_id;

enum whisper_alignment_heads_preset { WHISPER_AHEADS_NONE = 0 };

typedef struct whisper_ahead { int n_text_layer; int n_head; } whisper_ahead;
typedef struct whisper_aheads { size_t n_heads; const whisper_ahead *heads; } whisper_aheads;

/* replaces: whisper.cpp whisper_context_params (by value through whisper-rs WhisperContextParameters) */
struct whisper_context_params {
    bool use_gpu;
    bool flash_attn;
    int  gpu_device;            /* HIP device ordinal */
    bool dtw_token_timestamps;  /* accepted, ignored */
    enum whisper_alignment_heads_preset dtw_aheads_preset;
    int  dtw_n_top;
    struct whisper_aheads dtw_aheads;
    size_t dtw_mem_size;
};

typedef struct whisper_token_data {
    whisper_token id;   /* token id */
    whisper_token tid;  /* forced timestamp token id */
    float p;            /* probability of the token */
    float plog;         /* log probability of the token */
    float pt;           /* probability of the timestamp token */
    float ptsum;        /* sum of probabilities of all timestamp tokens */
    int64_t t0;         /* token-level timestamps (not produced: -1) */
    int64_t t1;
    int64_t t_dtw;
    float vlen;
} whisper_token_data;

enum whisper_sampling_strategy {
    WHISPER_SAMPLING_GREEDY,
    WHISPER_SAMPLING_BEAM_SEARCH,
};

typedef void (*whisper_new_segment_callback)(struct whisper_context *ctx, struct whisper_state *state, int n_new, void *user_data);
typedef void (*whisper_progress_callback)(struct whisper_context *ctx, struct whisper_state *state, int progress, void *user_data);
typedef bool (*whisper_encoder_begin_callback)(struct whisper_context *ctx, struct whisper_state *state, void *user_data);
typedef bool (*ggml_abort_callback)(void *data);
typedef void (*whisper_logits_filter_callback)(struct whisper_context *ctx, struct whisper_state *state,
                                               const whisper_token_data *tokens, int n_tokens, float *logits, void *user_data);
typedef struct whisper_grammar_element { int type; uint32_t value; } whisper_grammar_element;

/* replaces: whisper.cpp whisper_full_params (by value through whisper-rs FullParams) */
struct whisper_full_params {
    enum whisper_sampling_strategy strategy;

    int n_threads;          /* accepted, ignored (the work runs on the GPU) */
    int n_max_text_ctx;     /* max tokens to use from past text as prompt for the decoder */
    int offset_ms;          /* start offset in ms */
    int duration_ms;        /* audio duration to process in ms */

    bool translate;
    bool no_context;        /* do not use past transcription (if any) as initial prompt for the decoder */
    bool no_timestamps;     /* do not generate timestamps */
    bool single_segment;    /* force single segment output */
    bool print_special;     /* include special tokens in the segment text */
    bool print_progress;    /* accepted, ignored */
    bool print_realtime;    /* accepted, ignored */
    bool print_timestamps;  /* accepted, ignored */

    bool  token_timestamps; /* accepted, not implemented */
    float thold_pt;
    float thold_ptsum;
    int   max_len;
    bool  split_on_word;
    int   max_tokens;       /* max tokens per segment (0 = no limit) */

    bool debug_mode;
    int  audio_ctx;         /* overwrite the audio context size (0 = use default) */

    bool tdrz_enable;       /* enable tinydiarize speaker turn detection */

    const char *suppress_regex;   /* accepted, not implemented */

    const char *initial_prompt;
    const whisper_token *prompt_tokens;
    int prompt_n_tokens;

    const char *language;   /* for auto-detection, set to NULL, "" or "auto" */
    bool detect_language;

    bool suppress_blank;
    union { bool suppress_non_speech_tokens; bool suppress_nst; };  /* one field, both upstream spellings */

    float temperature;
    float max_initial_ts;
    float length_penalty;

    float temperature_inc;
    float entropy_thold;
    float logprob_thold;
    float no_speech_thold;  /* accepted, not implemented (as in the 1.7.1 line) */

    struct { int best_of; } greedy;
    struct { int beam_size; float patience; } beam_search;

    whisper_new_segment_callback new_segment_callback;
    void *new_segment_callback_user_data;
    whisper_progress_callback progress_callback;
    void *progress_callback_user_data;
    whisper_encoder_begin_callback encoder_begin_callback;
    void *encoder_begin_callback_user_data;
    ggml_abort_callback abort_callback;
    void *abort_callback_user_data;
    whisper_logits_filter_callback logits_filter_callback;   /* the logit rows make a round trip to host memory on every decoder pass while it is set */
    void *logits_filter_callback_user_data;

    const whisper_grammar_element **grammar_rules;           /* accepted, ignored */
    size_t n_grammar_rules;
    size_t i_start_rule;
    float  grammar_penalty;
};

typedef void (*ggml_log_callback)(int level, const char *text, void *user_data);

/* ---- context / state lifetime (whisper-rs WhisperContext::new_with_params, create_state, Drop) ---- */
WHISPER_API struct whisper_context_params whisper_context_default_params(void);
WHISPER_API struct whisper_context_params *whisper_context_default_params_by_ref(void);
WHISPER_API struct whisper_context *whisper_init_from_file_with_params(const char *path_model, struct whisper_context_params params);
WHISPER_API struct whisper_context *whisper_init_from_file_with_params_no_state(const char *path_model, struct whisper_context_params params);
WHISPER_API struct whisper_context *whisper_init_from_buffer_with_params(void *buffer, size_t buffer_size, struct whisper_context_params params);
WHISPER_API struct whisper_context *whisper_init_from_buffer_with_params_no_state(void *buffer, size_t buffer_size, struct whisper_context_params params);
WHISPER_API struct whisper_state *whisper_init_state(struct whisper_context *ctx);
WHISPER_API void whisper_free(struct whisper_context *ctx);
WHISPER_API void whisper_free_state(struct whisper_state *state);
WHISPER_API void whisper_free_params(struct whisper_full_params *params);
WHISPER_API void whisper_free_context_params(struct whisper_context_params *params);

/* ---- low-level stages (whisper-rs WhisperState::{pcm_to_mel, set_mel, encode, decode, get_logits}) ---- */
WHISPER_API int whisper_pcm_to_mel(struct whisper_context *ctx, const float *samples, int n_samples, int n_threads);
WHISPER_API int whisper_pcm_to_mel_with_state(struct whisper_context *ctx, struct whisper_state *state, const float *samples, int n_samples, int n_threads);
WHISPER_API int whisper_set_mel(struct whisper_context *ctx, const float *data, int n_len, int n_mel);
WHISPER_API int whisper_set_mel_with_state(struct whisper_context *ctx, struct whisper_state *state, const float *data, int n_len, int n_mel);
WHISPER_API int whisper_encode(struct whisper_context *ctx, int offset, int n_threads);
WHISPER_API int whisper_encode_with_state(struct whisper_context *ctx, struct whisper_state *state, int offset, int n_threads);
WHISPER_API int whisper_decode(struct whisper_context *ctx, const whisper_token *tokens, int n_tokens, int n_past, int n_threads);
WHISPER_API int whisper_decode_with_state(struct whisper_context *ctx, struct whisper_state *state, const whisper_token *tokens, int n_tokens, int n_past, int n_threads);
WHISPER_API float *whisper_get_logits(struct whisper_context *ctx);
WHISPER_API float *whisper_get_logits_from_state(struct whisper_state *state);
WHISPER_API int whisper_tokenize(struct whisper_context *ctx, const char *text, whisper_token *tokens, int n_max_tokens);
WHISPER_API int whisper_token_count(struct whisper_context *ctx, const char *text);

/* ---- language helpers ---- */
WHISPER_API int whisper_lang_max_id(void);
WHISPER_API int whisper_lang_id(const char *lang);
WHISPER_API const char *whisper_lang_str(int id);
WHISPER_API const char *whisper_lang_str_full(int id);
WHISPER_API int whisper_lang_auto_detect(struct whisper_context *ctx, int offset_ms, int n_threads, float *lang_probs);
WHISPER_API int whisper_lang_auto_detect_with_state(struct whisper_context *ctx, struct whisper_state *state, int offset_ms, int n_threads, float *lang_probs);

/* ---- model facts ---- */
WHISPER_API int whisper_n_len(struct whisper_context *ctx);
WHISPER_API int whisper_n_len_from_state(struct whisper_state *state);
WHISPER_API int whisper_n_vocab(struct whisper_context *ctx);
WHISPER_API int whisper_n_text_ctx(struct whisper_context *ctx);
WHISPER_API int whisper_n_audio_ctx(struct whisper_context *ctx);
WHISPER_API int whisper_is_multilingual(struct whisper_context *ctx);
WHISPER_API int whisper_model_n_vocab(struct whisper_context *ctx);
WHISPER_API int whisper_model_n_audio_ctx(struct whisper_context *ctx);
WHISPER_API int whisper_model_n_audio_state(struct whisper_context *ctx);
WHISPER_API int whisper_model_n_audio_head(struct whisper_context *ctx);
WHISPER_API int whisper_model_n_audio_layer(struct whisper_context *ctx);
WHISPER_API int whisper_model_n_text_ctx(struct whisper_context *ctx);
WHISPER_API int whisper_model_n_text_state(struct whisper_context *ctx);
WHISPER_API int whisper_model_n_text_head(struct whisper_context *ctx);
WHISPER_API int whisper_model_n_text_layer(struct whisper_context *ctx);
WHISPER_API int whisper_model_n_mels(struct whisper_context *ctx);
WHISPER_API int whisper_model_ftype(struct whisper_context *ctx);
WHISPER_API int whisper_model_type(struct whisper_context *ctx);
WHISPER_API const char *whisper_model_type_readable(struct whisper_context *ctx);

/* ---- tokens ---- */
WHISPER_API const char *whisper_token_to_str(struct whisper_context *ctx, whisper_token token);
WHISPER_API whisper_token whisper_token_eot(struct whisper_context *ctx);
WHISPER_API whisper_token whisper_token_sot(struct whisper_context *ctx);
WHISPER_API whisper_token whisper_token_solm(struct whisper_context *ctx);
WHISPER_API whisper_token whisper_token_prev(struct whisper_context *ctx);
WHISPER_API whisper_token whisper_token_nosp(struct whisper_context *ctx);
WHISPER_API whisper_token whisper_token_not(struct whisper_context *ctx);
WHISPER_API whisper_token whisper_token_beg(struct whisper_context *ctx);
WHISPER_API whisper_token whisper_token_lang(struct whisper_context *ctx, int lang_id);
WHISPER_API whisper_token whisper_token_translate(struct whisper_context *ctx);
WHISPER_API whisper_token whisper_token_transcribe(struct whisper_context *ctx);

/* ---- timings / system ---- */
WHISPER_API void whisper_print_timings(struct whisper_context *ctx);
WHISPER_API void whisper_reset_timings(struct whisper_context *ctx);
WHISPER_API const char *whisper_print_system_info(void);
WHISPER_API void whisper_log_set(ggml_log_callback log_callback, void *user_data);

/* ---- the full pipeline (whisper-rs FullParams / WhisperState::full) ---- */
WHISPER_API struct whisper_full_params whisper_full_default_params(enum whisper_sampling_strategy strategy);
WHISPER_API struct whisper_full_params *whisper_full_default_params_by_ref(enum whisper_sampling_strategy strategy);
WHISPER_API int whisper_full(struct whisper_context *ctx, struct whisper_full_params params, const float *samples, int n_samples);
/* returns 0 on success; -2 mel failure, -3 language detection failure, -4 too many decoders, -5 bad audio_ctx,
 * -6 encoder failure, -7/-8 decoder failure (whisper-rs maps non-zero to WhisperError) */
WHISPER_API int whisper_full_with_state(struct whisper_context *ctx, struct whisper_state *state, struct whisper_full_params params, const float *samples, int n_samples);
WHISPER_API int whisper_full_parallel(struct whisper_context *ctx, struct whisper_full_params params, const float *samples, int n_samples, int n_processors);

/* ---- result readout (whisper-rs WhisperState::full_n_segments, full_get_segment_text, ...) ---- */
WHISPER_API int whisper_full_n_segments(struct whisper_context *ctx);
WHISPER_API int whisper_full_n_segments_from_state(struct whisper_state *state);
WHISPER_API int whisper_full_lang_id(struct whisper_context *ctx);
WHISPER_API int whisper_full_lang_id_from_state(struct whisper_state *state);
WHISPER_API int64_t whisper_full_get_segment_t0(struct whisper_context *ctx, int i_segment);
WHISPER_API int64_t whisper_full_get_segment_t0_from_state(struct whisper_state *state, int i_segment);
WHISPER_API int64_t whisper_full_get_segment_t1(struct whisper_context *ctx, int i_segment);
WHISPER_API int64_t whisper_full_get_segment_t1_from_state(struct whisper_state *state, int i_segment);
WHISPER_API bool whisper_full_get_segment_speaker_turn_next(struct whisper_context *ctx, int i_segment);
WHISPER_API bool whisper_full_get_segment_speaker_turn_next_from_state(struct whisper_state *state, int i_segment);
WHISPER_API const char *whisper_full_get_segment_text(struct whisper_context *ctx, int i_segment);
WHISPER_API const char *whisper_full_get_segment_text_from_state(struct whisper_state *state, int i_segment);
WHISPER_API int whisper_full_n_tokens(struct whisper_context *ctx, int i_segment);
WHISPER_API int whisper_full_n_tokens_from_state(struct whisper_state *state, int i_segment);
WHISPER_API const char *whisper_full_get_token_text(struct whisper_context *ctx, int i_segment, int i_token);
WHISPER_API const char *whisper_full_get_token_text_from_state(struct whisper_context *ctx, struct whisper_state *state, int i_segment, int i_token);
WHISPER_API whisper_token whisper_full_get_token_id(struct whisper_context *ctx, int i_segment, int i_token);
WHISPER_API whisper_token whisper_full_get_token_id_from_state(struct whisper_state *state, int i_segment, int i_token);
WHISPER_API whisper_token_data whisper_full_get_token_data(struct whisper_context *ctx, int i_segment, int i_token);
WHISPER_API whisper_token_data whisper_full_get_token_data_from_state(struct whisper_state *state, int i_segment, int i_token);
WHISPER_API float whisper_full_get_token_p(struct whisper_context *ctx, int i_segment, int i_token);
WHISPER_API float whisper_full_get_token_p_from_state(struct whisper_state *state, int i_segment, int i_token);

#ifdef __cplusplus
}
#endif
#endif /* WHISPER_H */
