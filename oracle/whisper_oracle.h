/*
 * whisper_oracle.h -- CPU restatement of the Whisper inference path Handy reaches through
 * transcribe-rs 0.2.1 -> whisper-rs 0.13.2 -> whisper-rs-sys 0.11.1 (bundled whisper.cpp).
 *
 * TEST INFRASTRUCTURE ONLY. Nothing in the shipped library (handy_amd/) may include, link or
 * call this code; it is the checker that tests, the smoke run and the benchmark's CPU-baseline
 * leg compare the HIP engine against.
 *
 * PARITY UNPINNED: the arithmetic this file restates lives in third-party crates that are not
 * part of the reference tree (call sites: src-tauri/src/managers/transcription.rs:12-21,
 * 228-229, 413-415; pins: src-tauri/Cargo.lock:7137-7155, 7830-7848) and the reference holds
 * no test, fixture or golden vector for transcription. The restatement follows whisper.cpp's
 * published algorithm as recalled, and is cross-checked against the independent Hugging Face
 * transformers Whisper implementation on seeded synthetic weights (oracle/hf_golden.py).
 */
#ifndef WHISPER_ORACLE_H
#define WHISPER_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wo_model wo_model;
typedef struct wo_state wo_state;

typedef struct wo_params {
    int   strategy;          /* 0 greedy, 1 beam search */
    int   n_max_text_ctx;    /* 16384 */
    int   offset_ms, duration_ms;
    int   translate, no_context, no_timestamps, single_segment, print_special;
    int   max_tokens;
    int   audio_ctx;
    int   tdrz_enable;
    const char *language;    /* NULL / "" / "auto" => detect */
    int   detect_language;
    int   suppress_blank, suppress_nst;
    float temperature, max_initial_ts, length_penalty;
    float temperature_inc, entropy_thold, logprob_thold, no_speech_thold;
    int   best_of;
    int   beam_size;
    int   beam_candidates_topk; /* 0: candidates drawn from the distribution (upstream), 1: deterministic top-k */
    const int32_t *prompt_tokens; int prompt_n_tokens;
    int   force_n_tokens;    /* benchmark control: >0 => suppress EOT before, force EOT at, this many sampled tokens */
    /* whisper_full_params.logits_filter_callback [recalled]: called on every decoder's logits after the temperature and the
     * special-token suppression, before the non-speech list and the timestamp rules; may rewrite the logits in place.
     * (upstream also passes ctx and state, and the tokens as whisper_token_data; here: the token ids sampled so far) */
    void (*logits_filter_callback)(const int32_t *tokens, int n_tokens, float *logits, void *user_data);
    void *logits_filter_callback_user_data;
    /* whisper_full_params.encoder_begin_callback [recalled]: called before each window is encoded; returning 0 ends the
     * call with the segments produced so far */
    int  (*encoder_begin_callback)(void *user_data);
    void *encoder_begin_callback_user_data;
} wo_params;

typedef struct wo_token_data {
    int32_t id, tid;
    float   p, plog, pt, ptsum;
    int64_t t0, t1, t_dtw;
    float   vlen;
} wo_token_data;

wo_model *wo_load(const char *path);
void      wo_free_model(wo_model *m);
void      wo_hparams(const wo_model *m, int out[11]);
int       wo_token_special(const wo_model *m, const char *which); /* eot sot translate transcribe solm prev nosp not beg */
const char *wo_token_to_str(const wo_model *m, int id, int *len);
int       wo_lang_id(const char *lang);
const char *wo_lang_str(int id);

wo_state *wo_new_state(wo_model *m);
void      wo_free_state(wo_state *s);

/* log-mel of the whole input; returns 0 and fills state mel (n_mel x n_len, mel-major) */
int       wo_pcm_to_mel(wo_state *s, const float *pcm, int n);
int       wo_mel_n_len(const wo_state *s);
int       wo_mel_n_len_org(const wo_state *s);
const float *wo_mel_data(const wo_state *s);
int       wo_set_mel(wo_state *s, const float *data, int n_len, int n_mel);

/* encoder on the 3000-frame window starting at mel frame `seek`; fills enc_out and the cross K/V */
int       wo_encode(wo_state *s, int seek, int audio_ctx);
const float *wo_enc_out(const wo_state *s);          /* [n_ctx x n_state] */
const float *wo_conv_out(const wo_state *s);         /* [n_ctx x n_state] after conv2+GELU+pos-emb */
const float *wo_cross_k(const wo_state *s, int layer); /* [n_ctx x n_state] f16-rounded */
const float *wo_cross_v(const wo_state *s, int layer);

/* decoder for sequence `dec`: tokens at positions n_past.. ; logits_all!=0 keeps logits of every position */
int       wo_decode(wo_state *s, int dec, const int32_t *tokens, int n_tokens, int n_past, int logits_all);
const float *wo_logits(const wo_state *s);           /* [n_rows x n_vocab], n_rows = 1 or n_tokens */

/* logits processing of one decoder as whisper_full does it; exposed for known-answer tests */
void      wo_process_logits(const wo_model *m, const wo_params *p, const float *logits_in,
                            const int32_t *tokens_cur, int n_cur, int has_ts, int seek_delta,
                            float temperature, float *logits, float *logprobs, float *probs);

wo_params wo_default_params(int strategy);
int       wo_full(wo_state *s, const wo_params *p, const float *pcm, int n);
int       wo_full_n_segments(const wo_state *s);
int64_t   wo_full_segment_t0(const wo_state *s, int i);
int64_t   wo_full_segment_t1(const wo_state *s, int i);
const char *wo_full_segment_text(const wo_state *s, int i);
int       wo_full_n_tokens(const wo_state *s, int i);
wo_token_data wo_full_token_data(const wo_state *s, int i, int j);
int       wo_full_lang_id(const wo_state *s);
int       wo_n_fallbacks(const wo_state *s);
int       wo_n_threads(void);

#ifdef __cplusplus
}
#endif
#endif
