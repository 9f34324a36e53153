/*
 * whisper_oracle.c -- CPU restatement (plain C + OpenMP) of the Whisper path behind
 * Handy's TranscriptionManager::transcribe (src-tauri/src/managers/transcription.rs:391-416).
 *
 * TEST INFRASTRUCTURE ONLY (see whisper_oracle.h). PARITY UNPINNED: whisper.cpp (bundled in
 * whisper-rs-sys 0.11.1, src-tauri/Cargo.lock:7839-7848) is absent from the reference tree, so
 * every function below restates its published algorithm from recall; where a detail is a
 * recollection rather than a verified fact the comment says [recalled].
 *
 * Numerics follow whisper.cpp's CPU backend with f16 weights: matrix products take f16
 * weights and f16-rounded activations and accumulate in f32; K/V caches hold f16 values;
 * LayerNorm statistics and the mel filterbank accumulate in double; GELU is the tanh form
 * evaluated on the f16-rounded input and rounded to f16 (the content of ggml's f16 table).
 */
#include "whisper_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define WO_SAMPLE_RATE 16000
#define WO_N_FFT 400
#define WO_HOP 160
#define WO_CHUNK 30
#define WO_MAX_DECODERS 8

/* ------------------------------------------------------------------ f16 */
static inline float f16_to_f32(uint16_t h) {
    uint32_t s = (uint32_t)(h & 0x8000) << 16, e = (h >> 10) & 0x1F, m = h & 0x3FF, u;
    if (e == 0) {
        if (m == 0) u = s;
        else { int sh = 0; while (!(m & 0x400)) { m <<= 1; sh++; } m &= 0x3FF; u = s | ((uint32_t)(113 - sh) << 23) | (m << 13); }
    } else if (e == 31) u = s | 0x7F800000u | (m << 13);
    else u = s | ((e + 112) << 23) | (m << 13);
    float f; memcpy(&f, &u, 4); return f;
}
static inline uint16_t f32_to_f16(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t s = (x >> 16) & 0x8000; x &= 0x7FFFFFFF;
    if (x >= 0x7F800000u) return (uint16_t)(s | 0x7C00 | (x > 0x7F800000u ? 0x200 : 0));
    if (x >= 0x477FF000u) return (uint16_t)(s | 0x7C00);              /* overflow -> inf */
    if (x < 0x33000001u) return (uint16_t)s;                           /* underflow -> 0 */
    int e = (int)(x >> 23) - 127; uint32_t m = (x & 0x7FFFFF) | 0x800000;
    int shift; uint32_t h;
    if (e < -14) { shift = 13 + (-14 - e); h = 0; } else { shift = 13; h = (uint32_t)(e + 15) << 10; }
    uint32_t q = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (e >= -14) q &= 0x3FF;
    h += q;
    if (rem > half || (rem == half && (h & 1))) h++;
    return (uint16_t)(s | h);
}
static inline float r16(float f) { return f16_to_f32(f32_to_f16(f)); }

/* ------------------------------------------------------------------ model */
typedef struct { char name[96]; int n_dims; int ne[4]; float *data; size_t n; } wo_tensor;

typedef struct {
    float *ln1_w, *ln1_b, *q_w, *q_b, *k_w, *v_w, *v_b, *o_w, *o_b;
    float *lnx_w, *lnx_b, *xq_w, *xq_b, *xk_w, *xv_w, *xv_b, *xo_w, *xo_b;
    float *ln2_w, *ln2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} wo_layer;

struct wo_model {
    int n_vocab, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer;
    int n_text_ctx, n_text_state, n_text_head, n_text_layer, n_mels, ftype;
    int n_mel_f, n_fft_f; float *filters;
    int n_vocab_file; char **tok; int *tok_len;
    int t_eot, t_sot, t_translate, t_transcribe, t_solm, t_prev, t_nosp, t_not, t_beg, t_space;
    int n_tensors; wo_tensor *tensors;
    float *e_pe, *conv1_w, *conv1_b, *conv2_w, *conv2_b, *ln_post_w, *ln_post_b;
    float *d_pe, *d_te, *d_ln_w, *d_ln_b;
    wo_layer *enc, *dec;
    int *nst_ids; int n_nst;
};

static const char *k_lang[100][2] = {
 {"en","english"},{"zh","chinese"},{"de","german"},{"es","spanish"},{"ru","russian"},{"ko","korean"},{"fr","french"},
 {"ja","japanese"},{"pt","portuguese"},{"tr","turkish"},{"pl","polish"},{"ca","catalan"},{"nl","dutch"},{"ar","arabic"},
 {"sv","swedish"},{"it","italian"},{"id","indonesian"},{"hi","hindi"},{"fi","finnish"},{"vi","vietnamese"},{"he","hebrew"},
 {"uk","ukrainian"},{"el","greek"},{"ms","malay"},{"cs","czech"},{"ro","romanian"},{"da","danish"},{"hu","hungarian"},
 {"ta","tamil"},{"no","norwegian"},{"th","thai"},{"ur","urdu"},{"hr","croatian"},{"bg","bulgarian"},{"lt","lithuanian"},
 {"la","latin"},{"mi","maori"},{"ml","malayalam"},{"cy","welsh"},{"sk","slovak"},{"te","telugu"},{"fa","persian"},
 {"lv","latvian"},{"bn","bengali"},{"sr","serbian"},{"az","azerbaijani"},{"sl","slovenian"},{"kn","kannada"},
 {"et","estonian"},{"mk","macedonian"},{"br","breton"},{"eu","basque"},{"is","icelandic"},{"hy","armenian"},
 {"ne","nepali"},{"mn","mongolian"},{"bs","bosnian"},{"kk","kazakh"},{"sq","albanian"},{"sw","swahili"},
 {"gl","galician"},{"mr","marathi"},{"pa","punjabi"},{"si","sinhala"},{"km","khmer"},{"sn","shona"},{"yo","yoruba"},
 {"so","somali"},{"af","afrikaans"},{"oc","occitan"},{"ka","georgian"},{"be","belarusian"},{"tg","tajik"},
 {"sd","sindhi"},{"gu","gujarati"},{"am","amharic"},{"yi","yiddish"},{"lo","lao"},{"uz","uzbek"},{"fo","faroese"},
 {"ht","haitian creole"},{"ps","pashto"},{"tk","turkmen"},{"nn","nynorsk"},{"mt","maltese"},{"sa","sanskrit"},
 {"lb","luxembourgish"},{"my","myanmar"},{"bo","tibetan"},{"tl","tagalog"},{"mg","malagasy"},{"as","assamese"},
 {"tt","tatar"},{"haw","hawaiian"},{"ln","lingala"},{"ha","hausa"},{"ba","bashkir"},{"jw","javanese"},
 {"su","sundanese"},{"yue","cantonese"} };

int wo_lang_id(const char *lang) {
    if (!lang) return -1;
    for (int i = 0; i < 100; i++) if (!strcmp(lang, k_lang[i][0]) || !strcmp(lang, k_lang[i][1])) return i;
    return -1;
}
const char *wo_lang_str(int id) { return (id >= 0 && id < 100) ? k_lang[id][0] : NULL; }

static int is_multilingual(const wo_model *m) { return m->n_vocab >= 51865; }
static int num_languages(const wo_model *m) { return m->n_vocab - 51765 - (is_multilingual(m) ? 1 : 0); }
static int token_lang(const wo_model *m, int lang_id) { return m->t_sot + 1 + lang_id; }

static float *find_tensor(wo_model *m, const char *name) {
    for (int i = 0; i < m->n_tensors; i++) if (!strcmp(m->tensors[i].name, name)) return m->tensors[i].data;
    fprintf(stderr, "oracle: tensor '%s' missing\n", name);
    return NULL;
}

/* ggml block formats (32 elements): Q4_0 {f16 d, q[16]} x=d*(q-8); Q4_1 {f16 d, f16 m, q[16]} x=d*q+m;
 * Q5_0 {f16 d, u32 qh, q[16]} x=d*((q|bit<<4)-16); Q5_1 {f16 d, f16 m, u32 qh, q[16]}; Q8_0 {f16 d, i8 q[32]}.
 * Low nibbles are elements 0..15, high nibbles 16..31. */
static int dequant_block_bytes(int tt) { switch (tt) { case 2: return 18; case 3: return 20; case 6: return 22; case 7: return 24; case 8: return 34; } return 0; }
static void dequant(const uint8_t *src, int tt, size_t n, float *dst) {
    int bb = dequant_block_bytes(tt);
    for (size_t b = 0; b < n / 32; b++) {
        const uint8_t *p = src + b * bb; float *o = dst + b * 32;
        uint16_t dh; memcpy(&dh, p, 2); float d = f16_to_f32(dh), mn = 0; p += 2;
        if (tt == 3 || tt == 7) { uint16_t mh; memcpy(&mh, p, 2); mn = f16_to_f32(mh); p += 2; }
        if (tt == 8) { for (int j = 0; j < 32; j++) o[j] = (float)((int8_t)p[j]) * d; continue; }
        uint32_t qh = 0; if (tt == 6 || tt == 7) { memcpy(&qh, p, 4); p += 4; }
        for (int j = 0; j < 16; j++) {
            int lo = p[j] & 0x0F, hi = p[j] >> 4;
            if (tt == 6 || tt == 7) { lo |= ((qh >> j) & 1) << 4; hi |= ((qh >> (j + 16)) & 1) << 4; }
            if (tt == 2) { o[j] = (lo - 8) * d; o[j + 16] = (hi - 8) * d; }
            else if (tt == 6) { o[j] = (lo - 16) * d; o[j + 16] = (hi - 16) * d; }
            else { o[j] = lo * d + mn; o[j + 16] = hi * d + mn; }
        }
    }
}

static const char *k_non_speech[] = { "\"", "#", "(", ")", "*", "+", "/", ":", ";", "<", "=", ">", "@", "[", "\\", "]", "^",
    "_", "`", "{", "|", "}", "~", "\xe3\x80\x8c", "\xe3\x80\x8d", "\xe3\x80\x8e", "\xe3\x80\x8f", "<<", ">>", "<<<", ">>>", "--", "---", "-(", "-[",
    "('", "(\"", "((", "))", "(((", ")))", "[[", "]]", "{{", "}}", "\xe2\x99\xaa\xe2\x99\xaa", "\xe2\x99\xaa\xe2\x99\xaa\xe2\x99\xaa",
    "\xe2\x99\xa9", "\xe2\x99\xaa", "\xe2\x99\xab", "\xe2\x99\xac", "\xe2\x99\xad", "\xe2\x99\xae", "\xe2\x99\xaf" };

static int find_token(const wo_model *m, const char *s, int len) {
    for (int i = 0; i < m->n_vocab_file; i++) if (m->tok_len[i] == len && !memcmp(m->tok[i], s, len)) return i;
    return -1;
}

wo_model *wo_load(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    wo_model *m = (wo_model *)calloc(1, sizeof(*m));
    uint32_t magic; int32_t hp[11];
    if (fread(&magic, 4, 1, f) != 1 || magic != 0x67676d6c || fread(hp, 4, 11, f) != 11) { fclose(f); free(m); return NULL; }
    m->n_vocab = hp[0]; m->n_audio_ctx = hp[1]; m->n_audio_state = hp[2]; m->n_audio_head = hp[3]; m->n_audio_layer = hp[4];
    m->n_text_ctx = hp[5]; m->n_text_state = hp[6]; m->n_text_head = hp[7]; m->n_text_layer = hp[8]; m->n_mels = hp[9]; m->ftype = hp[10];
    int32_t nm[2];
    if (fread(nm, 4, 2, f) != 2) goto fail;
    m->n_mel_f = nm[0]; m->n_fft_f = nm[1];
    m->filters = (float *)malloc(sizeof(float) * nm[0] * nm[1]);
    if (fread(m->filters, 4, (size_t)nm[0] * nm[1], f) != (size_t)nm[0] * nm[1]) goto fail;
    int32_t nv;
    if (fread(&nv, 4, 1, f) != 1) goto fail;
    m->n_vocab_file = nv;
    m->tok = (char **)calloc(m->n_vocab > nv ? m->n_vocab : nv, sizeof(char *));
    m->tok_len = (int *)calloc(m->n_vocab > nv ? m->n_vocab : nv, sizeof(int));
    for (int i = 0; i < nv; i++) {
        uint32_t len; if (fread(&len, 4, 1, f) != 1) goto fail;
        m->tok[i] = (char *)malloc(len + 1); m->tok_len[i] = (int)len;
        if (len && fread(m->tok[i], 1, len, f) != len) goto fail;
        m->tok[i][len] = 0;
    }
    /* special token ids: English-only ids first, shifted for multilingual vocabularies */
    m->t_eot = 50256; m->t_sot = 50257; m->t_translate = 50357; m->t_transcribe = 50358; m->t_solm = 50359;
    m->t_prev = 50360; m->t_nosp = 50361; m->t_not = 50362; m->t_beg = 50363;
    if (is_multilingual(m)) {
        m->t_eot++; m->t_sot++;
        int dt = num_languages(m) - 98;
        m->t_translate += dt; m->t_transcribe += dt; m->t_solm += dt; m->t_prev += dt; m->t_nosp += dt; m->t_not += dt; m->t_beg += dt;
    }
    m->t_space = find_token(m, " ", 1);
    /* names for the ids past the end of the file's vocabulary */
    for (int i = nv; i < m->n_vocab; i++) {
        char w[64];
        if (i > m->t_beg) snprintf(w, sizeof w, "[_TT_%d]", i - m->t_beg);
        else if (i == m->t_eot) snprintf(w, sizeof w, "[_EOT_]");
        else if (i == m->t_sot) snprintf(w, sizeof w, "[_SOT_]");
        else if (i == m->t_translate) snprintf(w, sizeof w, "[_TRANSLATE_]");
        else if (i == m->t_transcribe) snprintf(w, sizeof w, "[_TRANSCRIBE_]");
        else if (i == m->t_solm) snprintf(w, sizeof w, "[_SOLM_]");
        else if (i == m->t_prev) snprintf(w, sizeof w, "[_PREV_]");
        else if (i == m->t_nosp) snprintf(w, sizeof w, "[_NOSP_]");
        else if (i == m->t_not) snprintf(w, sizeof w, "[_NOT_]");
        else if (i == m->t_beg) snprintf(w, sizeof w, "[_BEG_]");
        else if (i > m->t_sot && i <= m->t_sot + num_languages(m)) snprintf(w, sizeof w, "[_LANG_%s]", wo_lang_str(i - m->t_sot - 1));
        else snprintf(w, sizeof w, "[_extra_token_%d]", i);
        m->tok[i] = strdup(w); m->tok_len[i] = (int)strlen(w);
    }
    /* non-speech token ids: each symbol and " "+symbol, plus " -" and " '" */
    {
        int cap = 2 * (int)(sizeof k_non_speech / sizeof *k_non_speech) + 2;
        m->nst_ids = (int *)malloc(sizeof(int) * cap); m->n_nst = 0;
        for (size_t i = 0; i < sizeof k_non_speech / sizeof *k_non_speech; i++) {
            char b[32]; int id;
            int l = (int)strlen(k_non_speech[i]);
            if ((id = find_token(m, k_non_speech[i], l)) >= 0) m->nst_ids[m->n_nst++] = id;
            b[0] = ' '; memcpy(b + 1, k_non_speech[i], l);
            if ((id = find_token(m, b, l + 1)) >= 0) m->nst_ids[m->n_nst++] = id;
        }
        int id;
        if ((id = find_token(m, " -", 2)) >= 0) m->nst_ids[m->n_nst++] = id;
        if ((id = find_token(m, " '", 2)) >= 0) m->nst_ids[m->n_nst++] = id;
    }
    /* tensors */
    int cap = 0;
    for (;;) {
        int32_t h3[3];
        if (fread(h3, 4, 3, f) != 3) break;
        int n_dims = h3[0], name_len = h3[1], tt = h3[2];
        if (n_dims < 1 || n_dims > 4 || name_len <= 0 || name_len >= 96) goto fail;
        if (m->n_tensors == cap) { cap = cap ? cap * 2 : 64; m->tensors = (wo_tensor *)realloc(m->tensors, sizeof(wo_tensor) * cap); }
        wo_tensor *t = &m->tensors[m->n_tensors]; memset(t, 0, sizeof *t);
        t->n_dims = n_dims; size_t n = 1;
        for (int i = 0; i < n_dims; i++) { int32_t e; if (fread(&e, 4, 1, f) != 1 || e <= 0 || n > ((size_t)1 << 32) / (size_t)e) goto fail; t->ne[i] = e; n *= (size_t)e; }
        if (fread(t->name, 1, name_len, f) != (size_t)name_len) goto fail;
        t->n = n; t->data = (float *)malloc(sizeof(float) * n);
        if (!t->data) goto fail;
        m->n_tensors++;            /* from here on wo_free_model owns the buffer, also when the file ends inside the tensor */
        if (tt == 0) { if (fread(t->data, 4, n, f) != n) goto fail; }
        else if (tt == 1) {
            uint16_t *tmp = (uint16_t *)malloc(2 * n);
            if (fread(tmp, 2, n, f) != n) { free(tmp); goto fail; }
            for (size_t i = 0; i < n; i++) t->data[i] = f16_to_f32(tmp[i]);
            free(tmp);
        } else {
            int bb = dequant_block_bytes(tt);
            if (!bb || n % 32) goto fail;
            size_t nb = n / 32 * bb; uint8_t *tmp = (uint8_t *)malloc(nb);
            if (fread(tmp, 1, nb, f) != nb) { free(tmp); goto fail; }
            dequant(tmp, tt, n, t->data);
            /* quantised weights are expanded once at load and held as f16 */
            for (size_t i = 0; i < n; i++) t->data[i] = r16(t->data[i]);
            free(tmp);
        }
    }
    fclose(f); f = NULL;
    m->e_pe = find_tensor(m, "encoder.positional_embedding");
    m->conv1_w = find_tensor(m, "encoder.conv1.weight"); m->conv1_b = find_tensor(m, "encoder.conv1.bias");
    m->conv2_w = find_tensor(m, "encoder.conv2.weight"); m->conv2_b = find_tensor(m, "encoder.conv2.bias");
    m->ln_post_w = find_tensor(m, "encoder.ln_post.weight"); m->ln_post_b = find_tensor(m, "encoder.ln_post.bias");
    m->d_pe = find_tensor(m, "decoder.positional_embedding"); m->d_te = find_tensor(m, "decoder.token_embedding.weight");
    m->d_ln_w = find_tensor(m, "decoder.ln.weight"); m->d_ln_b = find_tensor(m, "decoder.ln.bias");
    if (!m->e_pe || !m->conv1_w || !m->conv2_w || !m->d_pe || !m->d_te || !m->d_ln_w) goto fail2;
    m->enc = (wo_layer *)calloc(m->n_audio_layer, sizeof(wo_layer));
    m->dec = (wo_layer *)calloc(m->n_text_layer, sizeof(wo_layer));
    for (int side = 0; side < 2; side++) {
        int nl = side ? m->n_text_layer : m->n_audio_layer;
        for (int i = 0; i < nl; i++) {
            wo_layer *L = side ? &m->dec[i] : &m->enc[i]; char n[96];
            const char *pre = side ? "decoder" : "encoder";
#define T(field, suffix) snprintf(n, sizeof n, "%s.blocks.%d.%s", pre, i, suffix); L->field = find_tensor(m, n); if (!L->field) goto fail2;
            T(ln1_w, "attn_ln.weight") T(ln1_b, "attn_ln.bias") T(q_w, "attn.query.weight") T(q_b, "attn.query.bias")
            T(k_w, "attn.key.weight") T(v_w, "attn.value.weight") T(v_b, "attn.value.bias") T(o_w, "attn.out.weight") T(o_b, "attn.out.bias")
            T(ln2_w, "mlp_ln.weight") T(ln2_b, "mlp_ln.bias") T(fc1_w, "mlp.0.weight") T(fc1_b, "mlp.0.bias") T(fc2_w, "mlp.2.weight") T(fc2_b, "mlp.2.bias")
            if (side) {
                T(lnx_w, "cross_attn_ln.weight") T(lnx_b, "cross_attn_ln.bias") T(xq_w, "cross_attn.query.weight") T(xq_b, "cross_attn.query.bias")
                T(xk_w, "cross_attn.key.weight") T(xv_w, "cross_attn.value.weight") T(xv_b, "cross_attn.value.bias")
                T(xo_w, "cross_attn.out.weight") T(xo_b, "cross_attn.out.bias")
            }
#undef T
        }
    }
    return m;
fail:
    if (f) fclose(f);
fail2:
    wo_free_model(m);
    return NULL;
}

void wo_free_model(wo_model *m) {
    if (!m) return;
    for (int i = 0; i < m->n_tensors; i++) free(m->tensors[i].data);
    free(m->tensors); free(m->filters);
    if (m->tok) { int n = m->n_vocab > m->n_vocab_file ? m->n_vocab : m->n_vocab_file; for (int i = 0; i < n; i++) free(m->tok[i]); }
    free(m->tok); free(m->tok_len); free(m->enc); free(m->dec); free(m->nst_ids); free(m);
}

void wo_hparams(const wo_model *m, int out[11]) {
    int v[11] = { m->n_vocab, m->n_audio_ctx, m->n_audio_state, m->n_audio_head, m->n_audio_layer, m->n_text_ctx,
                  m->n_text_state, m->n_text_head, m->n_text_layer, m->n_mels, m->ftype };
    memcpy(out, v, sizeof v);
}
int wo_token_special(const wo_model *m, const char *w) {
    if (!strcmp(w, "eot")) return m->t_eot; if (!strcmp(w, "sot")) return m->t_sot;
    if (!strcmp(w, "translate")) return m->t_translate; if (!strcmp(w, "transcribe")) return m->t_transcribe;
    if (!strcmp(w, "solm")) return m->t_solm; if (!strcmp(w, "prev")) return m->t_prev;
    if (!strcmp(w, "nosp")) return m->t_nosp; if (!strcmp(w, "not")) return m->t_not;
    if (!strcmp(w, "beg")) return m->t_beg; if (!strcmp(w, "space")) return m->t_space;
    return -1;
}
const char *wo_token_to_str(const wo_model *m, int id, int *len) {
    if (id < 0 || id >= m->n_vocab) { if (len) *len = 0; return ""; }
    if (len) *len = m->tok_len[id];
    return m->tok[id];
}
int wo_n_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ state */
typedef struct { wo_token_data *tok; int n, cap; int result_len; double sum_logprobs_all, sum_logprobs, avg_logprobs, entropy, score; } wo_sequence;

typedef struct { uint32_t mt[624]; int idx; } wo_rng;

typedef struct {
    wo_sequence seq; int seek_delta; int failed, completed, has_ts;
    float *probs, *logits, *logprobs;
    float *k, *v;            /* self K/V cache: [n_text_layer][n_text_ctx][n_state], f16-rounded */
    wo_rng rng;
} wo_decoder;

typedef struct { int64_t t0, t1; char *text; wo_token_data *tok; int n_tok; } wo_segment;

struct wo_state {
    wo_model *m;
    float *mel; int mel_n_len, mel_n_len_org, mel_n_mel;
    int n_ctx;                     /* audio context of the last encode */
    float *conv_out, *enc_out;
    float *cross_k, *cross_v;      /* [n_text_layer][n_ctx][n_state] f16-rounded */
    float *logits; int logits_rows;
    wo_decoder dec[2 * WO_MAX_DECODERS];
    int32_t *prompt_past; int n_prompt_past, cap_prompt_past;
    wo_segment *seg; int n_seg, cap_seg;
    int lang_id; int n_fallbacks;
};

wo_state *wo_new_state(wo_model *m) {
    wo_state *s = (wo_state *)calloc(1, sizeof *s);
    s->m = m;
    size_t d = m->n_audio_state;
    s->conv_out = (float *)malloc(sizeof(float) * m->n_audio_ctx * d);
    s->enc_out = (float *)malloc(sizeof(float) * m->n_audio_ctx * d);
    s->cross_k = (float *)malloc(sizeof(float) * m->n_text_layer * m->n_audio_ctx * m->n_text_state);
    s->cross_v = (float *)malloc(sizeof(float) * m->n_text_layer * m->n_audio_ctx * m->n_text_state);
    s->n_ctx = m->n_audio_ctx;
    return s;
}
static void seg_clear(wo_state *s) {
    for (int i = 0; i < s->n_seg; i++) { free(s->seg[i].text); free(s->seg[i].tok); }
    s->n_seg = 0;
}
void wo_free_state(wo_state *s) {
    if (!s) return;
    seg_clear(s); free(s->seg); free(s->mel); free(s->conv_out); free(s->enc_out); free(s->cross_k); free(s->cross_v); free(s->logits);
    for (int j = 0; j < 2 * WO_MAX_DECODERS; j++) { wo_decoder *D = &s->dec[j]; free(D->seq.tok); free(D->probs); free(D->logits); free(D->logprobs); free(D->k); free(D->v); }
    free(s->prompt_past); free(s);
}
static void dec_alloc(wo_state *s, int j) {
    wo_decoder *D = &s->dec[j]; wo_model *m = s->m;
    if (!D->k) {
        size_t n = (size_t)m->n_text_layer * m->n_text_ctx * m->n_text_state;
        D->k = (float *)calloc(n, sizeof(float)); D->v = (float *)calloc(n, sizeof(float));
    }
    if (!D->probs) {
        D->probs = (float *)malloc(sizeof(float) * m->n_vocab); D->logits = (float *)malloc(sizeof(float) * m->n_vocab);
        D->logprobs = (float *)malloc(sizeof(float) * m->n_vocab);
    }
}

/* ------------------------------------------------------------------ log-mel
 * Restates whisper.cpp log_mel_spectrogram [recalled]: periodic Hann(400); input laid out as
 * [reflect(samples[1..200]) | samples | zeros(480000+200)]; n_len = (n + 480000)/160 frames;
 * per frame f32 radix-2 FFT recursion that falls back to a direct DFT at odd length (25), with
 * shared 400-entry sin/cos tables; power spectrum of 201 bins; filterbank sum in double;
 * log10(max(.,1e-10)); then max-8 clamp over the WHOLE output and (x+4)/4. */
static float g_sin[WO_N_FFT], g_cos[WO_N_FFT], g_hann[WO_N_FFT]; static int g_tables = 0;
static void init_tables(void) {
    if (g_tables) return;
    for (int i = 0; i < WO_N_FFT; i++) {
        double th = (2.0 * M_PI * i) / WO_N_FFT;
        g_sin[i] = sinf((float)th); g_cos[i] = cosf((float)th);
        g_hann[i] = (float)(0.5 * (1.0 - cosf((float)((2.0 * M_PI * i) / WO_N_FFT))));
    }
    g_tables = 1;
}
static void dft_direct(const float *in, int N, float *out) {
    const int step = WO_N_FFT / N;
    for (int k = 0; k < N; k++) {
        float re = 0, im = 0;
        for (int n = 0; n < N; n++) { int idx = (k * n * step) % WO_N_FFT; re += in[n] * g_cos[idx]; im -= in[n] * g_sin[idx]; }
        out[2 * k] = re; out[2 * k + 1] = im;
    }
}
/* in: N samples followed by N scratch floats; out: 2N results followed by scratch for the halves */
static void fft_rec(float *in, int N, float *out) {
    if (N == 1) { out[0] = in[0]; out[1] = 0; return; }
    const int h = N / 2;
    if (N - 2 * h == 1) { dft_direct(in, N, out); return; }
    float *even = in + N;
    for (int i = 0; i < h; i++) even[i] = in[2 * i];
    float *even_f = out + 2 * N;
    fft_rec(even, h, even_f);
    float *odd = even;
    for (int i = 0; i < h; i++) odd[i] = in[2 * i + 1];
    float *odd_f = even_f + N;
    fft_rec(odd, h, odd_f);
    const int step = WO_N_FFT / N;
    for (int k = 0; k < h; k++) {
        float re = g_cos[k * step], im = -g_sin[k * step];
        float ro = odd_f[2 * k], io = odd_f[2 * k + 1];
        out[2 * k] = even_f[2 * k] + re * ro - im * io;
        out[2 * k + 1] = even_f[2 * k + 1] + re * io + im * ro;
        out[2 * (k + h)] = even_f[2 * k] - re * ro + im * io;
        out[2 * (k + h) + 1] = even_f[2 * k + 1] - re * io - im * ro;
    }
}

int wo_pcm_to_mel(wo_state *s, const float *pcm, int n) {
    wo_model *m = s->m; init_tables();
    const int n_mel = m->n_mels, n_fft = 1 + WO_N_FFT / 2;
    if (m->n_mel_f != n_mel || m->n_fft_f != n_fft) return -1;
    const int64_t pad1 = WO_SAMPLE_RATE * WO_CHUNK, pad2 = WO_N_FFT / 2;
    const int64_t np = n + pad1 + 2 * pad2;
    float *x = (float *)calloc(np, sizeof(float));
    memcpy(x + pad2, pcm, sizeof(float) * n);
    for (int i = 0; i < pad2; i++) x[i] = (1 + (pad2 - 1 - i) < n) ? pcm[1 + (pad2 - 1 - i)] : 0.0f;  /* reversed samples[1..200] */
    const int n_len = (int)((np - WO_N_FFT) / WO_HOP);
    const int n_len_org = 1 + (int)((n + pad2 - WO_N_FFT) / WO_HOP);
    free(s->mel); s->mel = (float *)malloc(sizeof(float) * n_mel * n_len);
    s->mel_n_len = n_len; s->mel_n_len_org = n_len_org; s->mel_n_mel = n_mel;
    const int64_t n_eff = n + pad2;     /* samples past this index are zero */
    int n_calc = (int)(n_eff / WO_HOP + 1); if (n_calc > n_len) n_calc = n_len;
#pragma omp parallel
    {
        float *fin = (float *)calloc(WO_N_FFT * 2, sizeof(float));
        float *fout = (float *)calloc(WO_N_FFT * 2 * 2 * 2, sizeof(float));
#pragma omp for schedule(static)
        for (int i = 0; i < n_len; i++) {
            if (i >= n_calc) { for (int j = 0; j < n_mel; j++) s->mel[(size_t)j * n_len + i] = (float)log10(1e-10); continue; }
            const int64_t off = (int64_t)i * WO_HOP;
            int64_t lim = n_eff - off; if (lim > WO_N_FFT) lim = WO_N_FFT;
            for (int j = 0; j < WO_N_FFT; j++) fin[j] = (j < lim) ? g_hann[j] * x[off + j] : 0.0f;
            fft_rec(fin, WO_N_FFT, fout);
            for (int j = 0; j < n_fft; j++) fout[j] = fout[2 * j] * fout[2 * j] + fout[2 * j + 1] * fout[2 * j + 1];
            for (int j = 0; j < n_mel; j++) {
                double sum = 0.0; const float *fl = m->filters + (size_t)j * n_fft;
                for (int k = 0; k < n_fft; k++) sum += fout[k] * fl[k];
                sum = log10(sum > 1e-10 ? sum : 1e-10);
                s->mel[(size_t)j * n_len + i] = (float)sum;
            }
        }
        free(fin); free(fout);
    }
    double mmax = -1e20;
    for (size_t i = 0; i < (size_t)n_mel * n_len; i++) if (s->mel[i] > mmax) mmax = s->mel[i];
    mmax -= 8.0;
    for (size_t i = 0; i < (size_t)n_mel * n_len; i++) {
        if (s->mel[i] < mmax) s->mel[i] = (float)mmax;
        s->mel[i] = (float)((s->mel[i] + 4.0) / 4.0);
    }
    free(x);
    return 0;
}
int wo_mel_n_len(const wo_state *s) { return s->mel_n_len; }
int wo_mel_n_len_org(const wo_state *s) { return s->mel_n_len_org; }
const float *wo_mel_data(const wo_state *s) { return s->mel; }
int wo_set_mel(wo_state *s, const float *data, int n_len, int n_mel) {
    if (n_mel != s->m->n_mels) return -1;
    free(s->mel); s->mel = (float *)malloc(sizeof(float) * n_len * n_mel);
    memcpy(s->mel, data, sizeof(float) * n_len * n_mel);
    s->mel_n_len = n_len; s->mel_n_len_org = n_len; s->mel_n_mel = n_mel;
    return 0;
}

/* ------------------------------------------------------------------ dense kernels */
typedef float v8f __attribute__((vector_size(32), aligned(4)));
static inline float hsum8(v8f v) { return ((v[0] + v[4]) + (v[2] + v[6])) + ((v[1] + v[5]) + (v[3] + v[7])); }

/* C[M x N] = A[M x K] * W[N x K]^T (+ bias[N]); lda/ldw/ldc in elements. K % 8 == 0. */
static void matmul_nt(const float *A, int lda, const float *W, int ldw, const float *bias, float *C, int ldc, int M, int N, int K) {
    const int MB = 4, NB = 2;
    const int mt = (M + MB - 1) / MB, nt = (N + NB - 1) / NB;
#pragma omp parallel for schedule(static) collapse(1) if ((long)M * N * K > 200000)
    for (int im = 0; im < mt; im++) {
        const int m0 = im * MB, mr = (M - m0 < MB) ? M - m0 : MB;
        for (int in = 0; in < nt; in++) {
            const int n0 = in * NB, nr = (N - n0 < NB) ? N - n0 : NB;
            if (mr == MB && nr == NB) {
                v8f a00 = {0}, a01 = {0}, a10 = {0}, a11 = {0}, a20 = {0}, a21 = {0}, a30 = {0}, a31 = {0};
                const float *p0 = A + (size_t)m0 * lda, *p1 = p0 + lda, *p2 = p1 + lda, *p3 = p2 + lda;
                const float *w0 = W + (size_t)n0 * ldw, *w1 = w0 + ldw;
                for (int k = 0; k < K; k += 8) {
                    v8f x0 = *(const v8f *)(w0 + k), x1 = *(const v8f *)(w1 + k), y;
                    y = *(const v8f *)(p0 + k); a00 += y * x0; a01 += y * x1;
                    y = *(const v8f *)(p1 + k); a10 += y * x0; a11 += y * x1;
                    y = *(const v8f *)(p2 + k); a20 += y * x0; a21 += y * x1;
                    y = *(const v8f *)(p3 + k); a30 += y * x0; a31 += y * x1;
                }
                float b0 = bias ? bias[n0] : 0, b1 = bias ? bias[n0 + 1] : 0;
                float *c = C + (size_t)m0 * ldc + n0;
                c[0] = hsum8(a00) + b0; c[1] = hsum8(a01) + b1; c += ldc;
                c[0] = hsum8(a10) + b0; c[1] = hsum8(a11) + b1; c += ldc;
                c[0] = hsum8(a20) + b0; c[1] = hsum8(a21) + b1; c += ldc;
                c[0] = hsum8(a30) + b0; c[1] = hsum8(a31) + b1;
            } else {
                for (int i = 0; i < mr; i++) for (int j = 0; j < nr; j++) {
                    v8f acc = {0}; const float *p = A + (size_t)(m0 + i) * lda, *w = W + (size_t)(n0 + j) * ldw;
                    for (int k = 0; k < K; k += 8) acc += *(const v8f *)(p + k) * *(const v8f *)(w + k);
                    C[(size_t)(m0 + i) * ldc + n0 + j] = hsum8(acc) + (bias ? bias[n0 + j] : 0);
                }
            }
        }
    }
}

static void round_f16_buf(const float *src, float *dst, size_t n) {
#pragma omp parallel for schedule(static) if (n > 65536)
    for (size_t i = 0; i < n; i++) dst[i] = r16(src[i]);
}

/* y = LN(x)*w + b, eps 1e-5, statistics in double */
static void layer_norm(const float *x, const float *w, const float *b, float *y, int M, int d) {
#pragma omp parallel for schedule(static) if (M > 16)
    for (int i = 0; i < M; i++) {
        const float *r = x + (size_t)i * d; float *o = y + (size_t)i * d;
        double sum = 0; for (int j = 0; j < d; j++) sum += r[j];
        float mean = (float)(sum / d);
        double sum2 = 0; for (int j = 0; j < d; j++) { float v = r[j] - mean; o[j] = v; sum2 += (double)(v * v); }
        float var = (float)(sum2 / d);
        const float scale = 1.0f / sqrtf(var + 1e-5f);
        for (int j = 0; j < d; j++) o[j] = o[j] * scale * w[j] + b[j];
    }
}

static inline float gelu_f16(float x) {
    if (x <= -10.0f) return 0.0f;
    if (x >= 10.0f) return x;
    float h = r16(x);
    float g = 0.5f * h * (1.0f + tanhf(0.79788456080286535587989211986876f * h * (1.0f + 0.044715f * h * h)));
    return r16(g);
}

/* ------------------------------------------------------------------ encoder */
int wo_encode(wo_state *s, int seek, int audio_ctx) {
    wo_model *m = s->m;
    const int d = m->n_audio_state, H = m->n_audio_head, dh = d / H, n_mel = m->n_mels;
    const int n_ctx = audio_ctx > 0 ? audio_ctx : m->n_audio_ctx;
    if (n_ctx > m->n_audio_ctx || !s->mel) return -1;
    s->n_ctx = n_ctx;
    const int T2 = 2 * n_ctx;
    /* window [seek, seek + 2*n_ctx) of the mel, zero beyond n_len; laid out [t][3 taps][n_mel] for conv1 */
    float *win = (float *)calloc((size_t)(T2 + 2) * n_mel, sizeof(float));    /* rows -1 .. T2 */
    {
        int i0 = seek < s->mel_n_len ? seek : s->mel_n_len, i1 = seek + T2 < s->mel_n_len ? seek + T2 : s->mel_n_len;
        for (int j = 0; j < n_mel; j++) for (int i = i0; i < i1; i++) win[(size_t)(i - i0 + 1) * n_mel + j] = r16(s->mel[(size_t)j * s->mel_n_len + i]);
    }
    /* conv weights [out][in][3] -> [out][tap][in] so a conv is one product over contiguous rows */
    const int K1 = 3 * n_mel, K2 = 3 * d;
    float *w1 = (float *)malloc(sizeof(float) * d * K1), *w2 = (float *)malloc(sizeof(float) * d * K2);
    for (int o = 0; o < d; o++) for (int c = 0; c < n_mel; c++) for (int k = 0; k < 3; k++) w1[(size_t)o * K1 + k * n_mel + c] = m->conv1_w[((size_t)o * n_mel + c) * 3 + k];
    for (int o = 0; o < d; o++) for (int c = 0; c < d; c++) for (int k = 0; k < 3; k++) w2[(size_t)o * K2 + k * d + c] = m->conv2_w[((size_t)o * d + c) * 3 + k];
    /* conv1 (k3, pad 1) + GELU -> h1 [T2 (+2 pad rows)][d] */
    float *h1 = (float *)calloc((size_t)(T2 + 2) * d, sizeof(float));
    matmul_nt(win, n_mel, w1, K1, m->conv1_b, h1 + d, d, T2, d, K1);
    for (size_t i = 0; i < (size_t)T2 * d; i++) h1[d + i] = gelu_f16(h1[d + i]);
    /* conv2 (k3, stride 2, pad 1) + GELU, then + positional embedding */
    float *x = s->conv_out;
    matmul_nt(h1, 2 * d, w2, K2, m->conv2_b, x, d, n_ctx, d, K2);
    for (size_t i = 0; i < (size_t)n_ctx * d; i++) x[i] = gelu_f16(x[i]) + m->e_pe[i];
    free(win); free(w1); free(w2); free(h1);

    float *cur = (float *)malloc(sizeof(float) * n_ctx * d);
    float *res = (float *)malloc(sizeof(float) * n_ctx * d);
    float *q = (float *)malloc(sizeof(float) * n_ctx * d), *k = (float *)malloc(sizeof(float) * n_ctx * d), *v = (float *)malloc(sizeof(float) * n_ctx * d);
    float *att = (float *)malloc(sizeof(float) * n_ctx * d);
    float *hid = (float *)malloc(sizeof(float) * n_ctx * 4 * d);
    memcpy(res, x, sizeof(float) * n_ctx * d);
    const float scale = 1.0f / sqrtf((float)dh);
    for (int il = 0; il < m->n_audio_layer; il++) {
        wo_layer *L = &m->enc[il];
        layer_norm(res, L->ln1_w, L->ln1_b, cur, n_ctx, d);
        round_f16_buf(cur, cur, (size_t)n_ctx * d);
        matmul_nt(cur, d, L->q_w, d, L->q_b, q, d, n_ctx, d, d);
        matmul_nt(cur, d, L->k_w, d, NULL, k, d, n_ctx, d, d);
        matmul_nt(cur, d, L->v_w, d, L->v_b, v, d, n_ctx, d, d);
        round_f16_buf(q, q, (size_t)n_ctx * d); round_f16_buf(k, k, (size_t)n_ctx * d); round_f16_buf(v, v, (size_t)n_ctx * d);
#pragma omp parallel
        {
            float *sc = (float *)malloc(sizeof(float) * n_ctx);
#pragma omp for schedule(dynamic, 16) collapse(2)
            for (int h = 0; h < H; h++) for (int i = 0; i < n_ctx; i++) {
                const float *qi = q + (size_t)i * d + h * dh; float mx = -INFINITY;
                for (int j = 0; j < n_ctx; j++) {
                    const float *kj = k + (size_t)j * d + h * dh; v8f acc = {0};
                    for (int c = 0; c < dh; c += 8) acc += *(const v8f *)(qi + c) * *(const v8f *)(kj + c);
                    sc[j] = hsum8(acc) * scale; if (sc[j] > mx) mx = sc[j];
                }
                double sum = 0; for (int j = 0; j < n_ctx; j++) { sc[j] = expf(sc[j] - mx); sum += sc[j]; }
                float inv = (float)(1.0 / sum);
                for (int j = 0; j < n_ctx; j++) sc[j] = r16(sc[j] * inv);
                float *o = att + (size_t)i * d + h * dh;
                for (int c = 0; c < dh; c++) o[c] = 0;
                for (int j = 0; j < n_ctx; j++) { const float p = sc[j]; const float *vj = v + (size_t)j * d + h * dh; for (int c = 0; c < dh; c++) o[c] += p * vj[c]; }
            }
            free(sc);
        }
        round_f16_buf(att, att, (size_t)n_ctx * d);
        matmul_nt(att, d, L->o_w, d, L->o_b, cur, d, n_ctx, d, d);
        for (size_t i = 0; i < (size_t)n_ctx * d; i++) res[i] += cur[i];
        layer_norm(res, L->ln2_w, L->ln2_b, cur, n_ctx, d);
        round_f16_buf(cur, cur, (size_t)n_ctx * d);
        matmul_nt(cur, d, L->fc1_w, d, L->fc1_b, hid, 4 * d, n_ctx, 4 * d, d);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < (size_t)n_ctx * 4 * d; i++) hid[i] = gelu_f16(hid[i]);
        matmul_nt(hid, 4 * d, L->fc2_w, 4 * d, L->fc2_b, cur, d, n_ctx, d, 4 * d);
        for (size_t i = 0; i < (size_t)n_ctx * d; i++) res[i] += cur[i];
    }
    layer_norm(res, m->ln_post_w, m->ln_post_b, s->enc_out, n_ctx, d);
    /* cross-attention K/V of every decoder layer, cached as f16 */
    round_f16_buf(s->enc_out, cur, (size_t)n_ctx * d);
    const int dt = m->n_text_state;
    for (int il = 0; il < m->n_text_layer; il++) {
        wo_layer *L = &m->dec[il];
        float *ck = s->cross_k + (size_t)il * m->n_audio_ctx * dt, *cv = s->cross_v + (size_t)il * m->n_audio_ctx * dt;
        matmul_nt(cur, d, L->xk_w, d, NULL, ck, dt, n_ctx, dt, d);
        matmul_nt(cur, d, L->xv_w, d, L->xv_b, cv, dt, n_ctx, dt, d);
        round_f16_buf(ck, ck, (size_t)n_ctx * dt); round_f16_buf(cv, cv, (size_t)n_ctx * dt);
    }
    free(cur); free(res); free(q); free(k); free(v); free(att); free(hid);
    return 0;
}
const float *wo_enc_out(const wo_state *s) { return s->enc_out; }
const float *wo_conv_out(const wo_state *s) { return s->conv_out; }
const float *wo_cross_k(const wo_state *s, int l) { return s->cross_k + (size_t)l * s->m->n_audio_ctx * s->m->n_text_state; }
const float *wo_cross_v(const wo_state *s, int l) { return s->cross_v + (size_t)l * s->m->n_audio_ctx * s->m->n_text_state; }

/* ------------------------------------------------------------------ decoder */
static void attend(const float *q, const float *K, const float *V, int n_kv, int ld, int H, int dh, float *out, float *sc) {
    const float scale = 1.0f / sqrtf((float)dh);
    for (int h = 0; h < H; h++) {
        const float *qh = q + h * dh; float mx = -INFINITY;
        for (int j = 0; j < n_kv; j++) {
            const float *kj = K + (size_t)j * ld + h * dh; float a = 0;
            for (int c = 0; c < dh; c++) a += qh[c] * kj[c];
            sc[j] = a * scale; if (sc[j] > mx) mx = sc[j];
        }
        double sum = 0; for (int j = 0; j < n_kv; j++) { sc[j] = expf(sc[j] - mx); sum += sc[j]; }
        float inv = (float)(1.0 / sum);
        float *o = out + h * dh; for (int c = 0; c < dh; c++) o[c] = 0;
        for (int j = 0; j < n_kv; j++) { float p = r16(sc[j] * inv); const float *vj = V + (size_t)j * ld + h * dh; for (int c = 0; c < dh; c++) o[c] += p * vj[c]; }
    }
}

int wo_decode(wo_state *s, int dec, const int32_t *tokens, int n_tokens, int n_past, int logits_all) {
    wo_model *m = s->m;
    const int d = m->n_text_state, H = m->n_text_head, dh = d / H, n_ctx = s->n_ctx, V = m->n_vocab;
    if (n_past + n_tokens > m->n_text_ctx || n_tokens <= 0) return -1;
    dec_alloc(s, dec);
    wo_decoder *D = &s->dec[dec];
    const int N = n_tokens;
    float *res = (float *)malloc(sizeof(float) * N * d), *cur = (float *)malloc(sizeof(float) * N * d);
    float *q = (float *)malloc(sizeof(float) * N * d), *att = (float *)malloc(sizeof(float) * N * d);
    float *hid = (float *)malloc(sizeof(float) * N * 4 * d);
    int sc_n = (n_ctx > m->n_text_ctx ? n_ctx : m->n_text_ctx);
    for (int i = 0; i < N; i++) {
        if (tokens[i] < 0 || tokens[i] >= V) { free(res); free(cur); free(q); free(att); free(hid); return -2; }
        for (int c = 0; c < d; c++) res[(size_t)i * d + c] = m->d_te[(size_t)tokens[i] * d + c] + m->d_pe[(size_t)(n_past + i) * d + c];
    }
    for (int il = 0; il < m->n_text_layer; il++) {
        wo_layer *L = &m->dec[il];
        float *Kc = D->k + (size_t)il * m->n_text_ctx * d, *Vc = D->v + (size_t)il * m->n_text_ctx * d;
        layer_norm(res, L->ln1_w, L->ln1_b, cur, N, d);
        round_f16_buf(cur, cur, (size_t)N * d);
        matmul_nt(cur, d, L->q_w, d, L->q_b, q, d, N, d, d);
        matmul_nt(cur, d, L->k_w, d, NULL, Kc + (size_t)n_past * d, d, N, d, d);
        matmul_nt(cur, d, L->v_w, d, L->v_b, Vc + (size_t)n_past * d, d, N, d, d);
        round_f16_buf(q, q, (size_t)N * d);
        round_f16_buf(Kc + (size_t)n_past * d, Kc + (size_t)n_past * d, (size_t)N * d);
        round_f16_buf(Vc + (size_t)n_past * d, Vc + (size_t)n_past * d, (size_t)N * d);
#pragma omp parallel
        {
            float *sc = (float *)malloc(sizeof(float) * sc_n);
#pragma omp for schedule(static)
            for (int i = 0; i < N; i++) attend(q + (size_t)i * d, Kc, Vc, n_past + i + 1, d, H, dh, att + (size_t)i * d, sc);
            free(sc);
        }
        round_f16_buf(att, att, (size_t)N * d);
        matmul_nt(att, d, L->o_w, d, L->o_b, cur, d, N, d, d);
        for (size_t i = 0; i < (size_t)N * d; i++) res[i] += cur[i];
        /* cross attention over the cached encoder K/V */
        layer_norm(res, L->lnx_w, L->lnx_b, cur, N, d);
        round_f16_buf(cur, cur, (size_t)N * d);
        matmul_nt(cur, d, L->xq_w, d, L->xq_b, q, d, N, d, d);
        round_f16_buf(q, q, (size_t)N * d);
        const float *ck = wo_cross_k(s, il), *cv = wo_cross_v(s, il);
#pragma omp parallel
        {
            float *sc = (float *)malloc(sizeof(float) * sc_n);
#pragma omp for schedule(static)
            for (int i = 0; i < N; i++) attend(q + (size_t)i * d, ck, cv, n_ctx, d, H, dh, att + (size_t)i * d, sc);
            free(sc);
        }
        round_f16_buf(att, att, (size_t)N * d);
        matmul_nt(att, d, L->xo_w, d, L->xo_b, cur, d, N, d, d);
        for (size_t i = 0; i < (size_t)N * d; i++) res[i] += cur[i];
        layer_norm(res, L->ln2_w, L->ln2_b, cur, N, d);
        round_f16_buf(cur, cur, (size_t)N * d);
        matmul_nt(cur, d, L->fc1_w, d, L->fc1_b, hid, 4 * d, N, 4 * d, d);
        for (size_t i = 0; i < (size_t)N * 4 * d; i++) hid[i] = gelu_f16(hid[i]);
        matmul_nt(hid, 4 * d, L->fc2_w, 4 * d, L->fc2_b, cur, d, N, d, 4 * d);
        for (size_t i = 0; i < (size_t)N * d; i++) res[i] += cur[i];
    }
    layer_norm(res, m->d_ln_w, m->d_ln_b, cur, N, d);
    round_f16_buf(cur, cur, (size_t)N * d);
    const int rows = logits_all ? N : 1;
    free(s->logits); s->logits = (float *)malloc(sizeof(float) * rows * V); s->logits_rows = rows;
    matmul_nt(cur + (size_t)(N - rows) * d, d, m->d_te, d, NULL, s->logits, V, rows, V, d);
    free(res); free(cur); free(q); free(att); free(hid);
    return 0;
}
const float *wo_logits(const wo_state *s) { return s->logits; }

/* ------------------------------------------------------------------ logits processing
 * Restates whisper_process_logits [recalled]. Order: temperature; suppress_blank (first sampled
 * position: eot and " "); not-token; all timestamps when no_timestamps; sot, nosp, solm (unless
 * tdrz), translate, transcribe, prev, language tokens; optional non-speech set; timestamp
 * pairing; max_initial_ts; monotonic timestamps; log-softmax; timestamp-mass rule. */
void wo_process_logits(const wo_model *m, const wo_params *p, const float *logits_in, const int32_t *cur, int n_cur,
                       int has_ts, int seek_delta, float temperature, float *logits, float *logprobs, float *probs) {
    const int V = m->n_vocab;
    memcpy(logits, logits_in, sizeof(float) * V);
    if (temperature > 0.0f) for (int i = 0; i < V; i++) logits[i] /= temperature;
    const int is_initial = n_cur == 0;
    if (p->suppress_blank && is_initial) { logits[m->t_eot] = -INFINITY; if (m->t_space >= 0) logits[m->t_space] = -INFINITY; }
    logits[m->t_not] = -INFINITY;
    if (p->no_timestamps) for (int i = m->t_beg; i < V; i++) logits[i] = -INFINITY;
    logits[m->t_sot] = -INFINITY; logits[m->t_nosp] = -INFINITY;
    if (!p->tdrz_enable) logits[m->t_solm] = -INFINITY;
    logits[m->t_translate] = -INFINITY; logits[m->t_transcribe] = -INFINITY; logits[m->t_prev] = -INFINITY;
    for (int i = 0; i < 100; i++) { int t = token_lang(m, i); if (t < V) logits[t] = -INFINITY; }
    if (p->logits_filter_callback) p->logits_filter_callback(cur, n_cur, logits, p->logits_filter_callback_user_data);
    if (p->suppress_nst) for (int i = 0; i < m->n_nst; i++) logits[m->nst_ids[i]] = -INFINITY;
    if (p->force_n_tokens > 0) {   /* benchmark control, not part of the upstream algorithm */
        if (n_cur < p->force_n_tokens) logits[m->t_eot] = -INFINITY;
        else { for (int i = 0; i < V; i++) if (i != m->t_eot) logits[i] = -INFINITY; logits[m->t_eot] = 0.0f; }
    }
    {
        const int last_ts = n_cur > 0 && cur[n_cur - 1] >= m->t_beg;
        const int pen_ts = n_cur < 2 || cur[n_cur - 2] >= m->t_beg;
        if (last_ts) {
            if (pen_ts) for (int i = m->t_beg; i < V; i++) logits[i] = -INFINITY;
            else for (int i = 0; i < m->t_eot; i++) logits[i] = -INFINITY;
        }
    }
    if (is_initial && p->max_initial_ts > 0.0f) {
        const float precision = (float)WO_CHUNK / m->n_audio_ctx;
        const int tid0 = (int)roundf(p->max_initial_ts / precision);
        for (int i = m->t_beg + tid0 + 1; i < V; i++) logits[i] = -INFINITY;
    }
    if (has_ts) { const int tid0 = seek_delta / 2; for (int i = m->t_beg; i < m->t_beg + tid0 && i < V; i++) logits[i] = -INFINITY; }
    {
        float mx = -INFINITY; for (int i = 0; i < V; i++) if (logits[i] > mx) mx = logits[i];
        float lse = 0; for (int i = 0; i < V; i++) if (logits[i] > -INFINITY) lse += expf(logits[i] - mx);
        lse = logf(lse) + mx;
        for (int i = 0; i < V; i++) logprobs[i] = logits[i] > -INFINITY ? logits[i] - lse : -INFINITY;
    }
    {
        float ts_lp = -INFINITY, mx = -INFINITY, lse = 0;
        for (int i = m->t_beg; i < V; i++) if (logprobs[i] > mx) mx = logprobs[i];
        for (int i = m->t_beg; i < V; i++) if (logprobs[i] > -INFINITY) lse += expf(logprobs[i] - mx);
        if (lse > 0.0f) ts_lp = logf(lse) + mx;
        float mt = -INFINITY; for (int i = 0; i < m->t_beg; i++) if (logprobs[i] > mt) mt = logprobs[i];
        if (ts_lp > mt) for (int i = 0; i < m->t_beg; i++) { logits[i] = -INFINITY; logprobs[i] = -INFINITY; }
    }
    for (int i = 0; i < V; i++) probs[i] = logits[i] == -INFINITY ? 0.0f : expf(logprobs[i]);
}

/* ------------------------------------------------------------------ sampling */
static void rng_seed(wo_rng *r, uint32_t s) { r->mt[0] = s; for (int i = 1; i < 624; i++) r->mt[i] = 1812433253u * (r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) + (uint32_t)i; r->idx = 624; }
static uint32_t rng_next(wo_rng *r) {
    if (r->idx >= 624) {
        for (int i = 0; i < 624; i++) { uint32_t y = (r->mt[i] & 0x80000000u) | (r->mt[(i + 1) % 624] & 0x7fffffffu); r->mt[i] = r->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfu : 0); }
        r->idx = 0;
    }
    uint32_t y = r->mt[r->idx++]; y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18; return y;
}
/* the mapping std::discrete_distribution applies in libstdc++: a 53-bit canonical double from two
 * 32-bit draws, then the first index whose normalised cumulative probability is >= it [recalled] */
static double rng_canonical(wo_rng *r) {
    double lo = (double)rng_next(r), hi = (double)rng_next(r);
    double v = (lo + hi * 4294967296.0) / 18446744073709551616.0;
    if (v >= 1.0) v = nextafter(1.0, 0.0);
    return v;
}
static int sample_discrete(const float *probs, int n, wo_rng *r) {
    double sum = 0; for (int i = 0; i < n; i++) sum += probs[i];
    const double u = rng_canonical(r);
    double c = 0;
    for (int i = 0; i < n - 1; i++) { c += probs[i] / sum; if (c >= u) return i; }
    return n - 1;
}
static void ts_stats(const wo_model *m, const float *probs, int *tid, float *pt, float *ptsum) {
    double sum = 0, mx = 0; *tid = m->t_beg;
    for (int i = m->t_beg; i < m->n_vocab; i++) { sum += probs[i]; if (mx < probs[i]) { mx = probs[i]; *tid = i; } }
    *pt = (float)(mx / (sum + 1e-10)); *ptsum = (float)sum;
}
static wo_token_data sample_token(const wo_model *m, wo_decoder *D, int best) {
    wo_token_data r; memset(&r, 0, sizeof r); r.t0 = r.t1 = r.t_dtw = -1;
    int tid; ts_stats(m, D->probs, &tid, &r.pt, &r.ptsum); r.tid = tid;
    if (best) { for (int i = 0; i < m->n_vocab; i++) if (r.p < D->probs[i]) { r.id = i; r.p = D->probs[i]; r.plog = D->logprobs[i]; } }
    else { r.id = sample_discrete(D->probs, m->n_vocab, &D->rng); r.p = D->probs[r.id]; r.plog = D->logprobs[r.id]; }
    if (r.id >= m->t_beg) { r.tid = r.id; r.pt = r.p; }
    return r;
}
static void sample_topk(const wo_model *m, wo_decoder *D, int k, int deterministic, wo_token_data *out) {
    int tid; float pt, ptsum; ts_stats(m, D->probs, &tid, &pt, &ptsum);
    int *used = NULL;
    if (deterministic) used = (int *)malloc(sizeof(int) * k);
    for (int i = 0; i < k; i++) {
        int id;
        if (deterministic) {      /* i-th largest probability, lowest id first on ties */
            id = -1; float best = -1.0f;
            for (int t = 0; t < m->n_vocab; t++) {
                int skip = 0; for (int u = 0; u < i; u++) if (used[u] == t) skip = 1;
                if (!skip && D->probs[t] > best) { best = D->probs[t]; id = t; }
            }
            used[i] = id;
        } else id = sample_discrete(D->probs, m->n_vocab, &D->rng);
        wo_token_data r; memset(&r, 0, sizeof r); r.t0 = r.t1 = r.t_dtw = -1;
        r.id = id; r.tid = tid; r.p = D->probs[id]; r.plog = D->logprobs[id]; r.pt = pt; r.ptsum = ptsum;
        if (r.id >= m->t_beg) { r.tid = r.id; r.pt = r.p; }
        out[i] = r;
    }
    free(used);
}

static void seq_push(wo_sequence *q, wo_token_data t) {
    if (q->n == q->cap) { q->cap = q->cap ? q->cap * 2 : 64; q->tok = (wo_token_data *)realloc(q->tok, sizeof(wo_token_data) * q->cap); }
    q->tok[q->n++] = t;
}
static void seq_copy(wo_sequence *dst, const wo_sequence *src) {
    wo_token_data *keep = dst->tok; int cap = dst->cap;
    if (cap < src->n + 1) { cap = src->n + 64; keep = (wo_token_data *)realloc(keep, sizeof(wo_token_data) * cap); }
    if (src->n) memcpy(keep, src->tok, sizeof(wo_token_data) * src->n);
    *dst = *src; dst->tok = keep; dst->cap = cap;
}
static void seq_score(const wo_params *p, wo_sequence *q) {
    if (q->result_len == 0) return;
    double r = 0; for (int i = 0; i < q->result_len; i++) r += q->tok[i].plog;
    q->sum_logprobs = r; q->avg_logprobs = r / q->result_len;
    double pen = q->result_len;
    if (p->length_penalty > 0.0f) pen = pow((5.0 + pen) / 6.0, p->length_penalty);
    q->score = r / pen;
    int i0 = q->result_len - 32 > 0 ? q->result_len - 32 : 0, cnt = q->result_len - i0; double ent = 0;
    for (int i = i0; i < q->result_len; i++) {
        int first = 1, c = 0;
        for (int j = i0; j < q->result_len; j++) if (q->tok[j].id == q->tok[i].id) { if (j < i) first = 0; c++; }
        if (first) { double pr = c / (double)cnt; ent -= pr * log(pr); }
    }
    q->entropy = ent;
}

/* ------------------------------------------------------------------ whisper_full restatement */
wo_params wo_default_params(int strategy) {
    wo_params p; memset(&p, 0, sizeof p);
    p.strategy = strategy; p.n_max_text_ctx = 16384; p.no_context = 1; p.language = "en";
    p.suppress_blank = 1; p.temperature = 0.0f; p.max_initial_ts = 1.0f; p.length_penalty = -1.0f;
    p.temperature_inc = 0.2f; p.entropy_thold = 2.4f; p.logprob_thold = -1.0f; p.no_speech_thold = 0.6f;
    p.best_of = strategy == 0 ? 5 : -1; p.beam_size = strategy == 1 ? 5 : -1;
    return p;
}

static int lang_auto_detect(wo_state *s, int seek, float *probs_out) {
    wo_model *m = s->m;
    if (seek < 0) return -1;
    if (seek >= s->mel_n_len_org) return -2;
    if (wo_encode(s, seek, 0) != 0) return -6;
    int32_t sot = m->t_sot;
    if (wo_decode(s, 0, &sot, 1, 0, 0) != 0) return -7;
    int best = -1; float bl = -INFINITY; int nl = num_languages(m) < 100 ? 100 : 100;
    double mx = -1e30;
    for (int i = 0; i < nl; i++) { int t = token_lang(m, i); if (t >= m->n_vocab) continue; float l = s->logits[t]; if (l > bl) { bl = l; best = i; } if (l > mx) mx = l; }
    if (probs_out) {
        double sum = 0; for (int i = 0; i < nl; i++) { int t = token_lang(m, i); if (t < m->n_vocab) sum += exp(s->logits[t] - mx); }
        for (int i = 0; i < nl; i++) { int t = token_lang(m, i); probs_out[i] = t < m->n_vocab ? (float)(exp(s->logits[t] - mx) / sum) : 0; }
    }
    return best;
}

static void push_segment(wo_state *s, int64_t t0, int64_t t1, const char *text, int tlen, const wo_token_data *tok, int n_tok) {
    if (s->n_seg == s->cap_seg) { s->cap_seg = s->cap_seg ? s->cap_seg * 2 : 16; s->seg = (wo_segment *)realloc(s->seg, sizeof(wo_segment) * s->cap_seg); }
    wo_segment *g = &s->seg[s->n_seg++];
    g->t0 = t0; g->t1 = t1; g->text = (char *)malloc(tlen + 1); memcpy(g->text, text, tlen); g->text[tlen] = 0;
    g->tok = (wo_token_data *)malloc(sizeof(wo_token_data) * (n_tok > 0 ? n_tok : 1)); memcpy(g->tok, tok, sizeof(wo_token_data) * n_tok); g->n_tok = n_tok;
}

typedef struct { int decoder_idx, seek_delta, has_ts; wo_sequence seq; } beam_cand;
static int cand_cmp(const void *a, const void *b) {
    const beam_cand *x = (const beam_cand *)a, *y = (const beam_cand *)b;
    if (x->seq.sum_logprobs_all != y->seq.sum_logprobs_all) return x->seq.sum_logprobs_all > y->seq.sum_logprobs_all ? -1 : 1;
    return x->decoder_idx - y->decoder_idx;
}
static int seq_tokens_equal(const wo_sequence *a, const wo_sequence *b) {
    if (a->n != b->n) return 0;
    for (int i = 0; i < a->n; i++) if (a->tok[i].id != b->tok[i].id) return 0;
    return 1;
}
static void kv_copy(wo_state *s, int src, int dst, int n_tok) {
    if (src == dst) return;
    wo_model *m = s->m; dec_alloc(s, dst);
    for (int il = 0; il < m->n_text_layer; il++) {
        size_t off = (size_t)il * m->n_text_ctx * m->n_text_state, n = (size_t)n_tok * m->n_text_state;
        memcpy(s->dec[dst].k + off, s->dec[src].k + off, sizeof(float) * n);
        memcpy(s->dec[dst].v + off, s->dec[src].v + off, sizeof(float) * n);
    }
}

int wo_full(wo_state *s, const wo_params *pin, const float *pcm, int n) {
    wo_model *m = s->m; wo_params P = *pin; wo_params *p = &P;
    seg_clear(s); s->n_fallbacks = 0;
    if (n > 0 && wo_pcm_to_mel(s, pcm, n) != 0) return -2;
    if (!s->mel) return -2;
    if (p->language == NULL || p->language[0] == 0 || !strcmp(p->language, "auto") || p->detect_language) {
        int id = is_multilingual(m) ? lang_auto_detect(s, 0, NULL) : 0;
        if (id < 0) return -3;
        s->lang_id = id; p->language = wo_lang_str(id);
        if (p->detect_language) return 0;
    }
    const int seek_start = p->offset_ms / 10;
    const int seek_end = p->duration_ms == 0 ? s->mel_n_len_org : seek_start + p->duration_ms / 10;
    if (seek_end < seek_start + 100) return 0;       /* under one second of audio: nothing is produced */

    float temps[16]; int n_temps = 0;
    if (p->temperature_inc > 0.0f) { for (float t = p->temperature; t < 1.0f + 1e-6f && n_temps < 16; t += p->temperature_inc) temps[n_temps++] = t; }
    else temps[n_temps++] = p->temperature;

    int n_decoders = 1;
    if (p->strategy == 0) n_decoders = p->best_of; else n_decoders = p->best_of > p->beam_size ? p->best_of : p->beam_size;
    if (n_decoders < 1) n_decoders = 1;
    if (n_decoders > WO_MAX_DECODERS) return -4;
    for (int j = 0; j < n_decoders; j++) { dec_alloc(s, j); rng_seed(&s->dec[j].rng, (uint32_t)j); }   /* decoder j's mt19937 starts from seed j [recalled]: with one shared seed the best_of decoders, which start from the same prompt logits, would all draw the same tokens */

    if (p->no_context) s->n_prompt_past = 0;
    if (p->prompt_tokens && p->prompt_n_tokens > 0) {
        int need = s->n_prompt_past + p->prompt_n_tokens;
        if (need > s->cap_prompt_past) { s->cap_prompt_past = need + 256; s->prompt_past = (int32_t *)realloc(s->prompt_past, sizeof(int32_t) * s->cap_prompt_past); }
        memmove(s->prompt_past + p->prompt_n_tokens, s->prompt_past, sizeof(int32_t) * s->n_prompt_past);
        memcpy(s->prompt_past, p->prompt_tokens, sizeof(int32_t) * p->prompt_n_tokens);
        s->n_prompt_past = need;
    }
    if (p->audio_ctx > m->n_audio_ctx) return -5;

    int32_t prompt_init[8]; int n_init = 0;
    prompt_init[n_init++] = m->t_sot;
    if (is_multilingual(m)) {
        const int lang_id = wo_lang_id(p->language);
        if (lang_id < 0) return -3;
        s->lang_id = lang_id;
        prompt_init[n_init++] = token_lang(m, lang_id);
        prompt_init[n_init++] = p->translate ? m->t_translate : m->t_transcribe;
    }
    {   /* distilled 2-layer decoders were trained without timestamps [recalled] */
        const int is_distil = m->n_text_layer == 2 && m->n_vocab != 51866;
        if (is_distil && !p->no_timestamps) p->no_timestamps = 1;
    }
    if (p->no_timestamps) prompt_init[n_init++] = m->t_not;

    int seek = seek_start;
    int32_t *prompt = (int32_t *)malloc(sizeof(int32_t) * (m->n_text_ctx + 16)); int n_prompt = 0;
    const int n_max = m->n_text_ctx / 2 - 4;
    beam_cand *cands = (beam_cand *)calloc((size_t)WO_MAX_DECODERS * WO_MAX_DECODERS, sizeof(beam_cand));
    wo_token_data *topk = (wo_token_data *)malloc(sizeof(wo_token_data) * WO_MAX_DECODERS);
    int rc = 0;

    while (1) {
        if (seek + 100 >= seek_end) break;
        if (p->encoder_begin_callback && !p->encoder_begin_callback(p->encoder_begin_callback_user_data)) break;
        if (wo_encode(s, seek, p->audio_ctx) != 0) { rc = -6; break; }
        if (seek > seek_start && seek + 500 >= seek_end) s->n_prompt_past = 0;
        int best_id = 0;
        for (int it = 0; it < n_temps; it++) {
            const float t_cur = temps[it];
            if (it > 0) s->n_fallbacks++;
            int n_cur = 1;
            if (p->strategy == 0) { if (t_cur > 0.0f) n_cur = p->best_of; }
            else { if (t_cur > 0.0f) n_cur = p->best_of; else n_cur = p->beam_size; }
            if (n_cur < 1) n_cur = 1;
            for (int j = 0; j < n_cur; j++) {
                wo_decoder *D = &s->dec[j];
                D->seq.n = 0; D->seq.result_len = 0; D->seq.sum_logprobs_all = 0; D->seq.sum_logprobs = -INFINITY;
                D->seq.avg_logprobs = -INFINITY; D->seq.entropy = 0; D->seq.score = -INFINITY;
                D->seek_delta = 100 * WO_CHUNK; D->failed = 0; D->completed = 0; D->has_ts = 0;
            }
            n_prompt = 0;
            if (s->n_prompt_past > 0 && t_cur < 0.5f && p->n_max_text_ctx > 0) {
                int n_take = p->n_max_text_ctx; if (n_take > m->n_text_ctx / 2) n_take = m->n_text_ctx / 2; if (n_take > s->n_prompt_past) n_take = s->n_prompt_past;
                prompt[n_prompt++] = m->t_prev;
                memcpy(prompt + n_prompt, s->prompt_past + s->n_prompt_past - n_take, sizeof(int32_t) * n_take); n_prompt += n_take;
            }
            memcpy(prompt + n_prompt, prompt_init, sizeof(int32_t) * n_init); n_prompt += n_init;
            if (wo_decode(s, 0, prompt, n_prompt, 0, 0) != 0) { rc = -7; goto done; }
            wo_process_logits(m, p, s->logits, NULL, 0, 0, s->dec[0].seek_delta, t_cur, s->dec[0].logits, s->dec[0].logprobs, s->dec[0].probs);
            for (int j = 1; j < n_cur; j++) {
                kv_copy(s, 0, j, n_prompt);
                memcpy(s->dec[j].probs, s->dec[0].probs, sizeof(float) * m->n_vocab);
                memcpy(s->dec[j].logits, s->dec[0].logits, sizeof(float) * m->n_vocab);
                memcpy(s->dec[j].logprobs, s->dec[0].logprobs, sizeof(float) * m->n_vocab);
            }
            for (int i = 0; i < n_max; i++) {
                int n_cand = 0;
                for (int j = 0; j < n_cur; j++) {
                    wo_decoder *D = &s->dec[j];
                    if (D->completed || D->failed) continue;
                    if (p->strategy == 0) {
                        wo_token_data t = sample_token(m, D, t_cur < 1e-6f);
                        seq_push(&D->seq, t); D->seq.sum_logprobs_all += t.plog;
                    } else {
                        sample_topk(m, D, p->beam_size, p->beam_candidates_topk, topk);
                        for (int c = 0; c < p->beam_size; c++) {
                            beam_cand *bc = &cands[n_cand++];
                            bc->decoder_idx = j; bc->seek_delta = D->seek_delta; bc->has_ts = D->has_ts;
                            seq_copy(&bc->seq, &D->seq); seq_push(&bc->seq, topk[c]); bc->seq.sum_logprobs_all += topk[c].plog;
                        }
                    }
                }
                if (p->strategy == 1) {
                    qsort(cands, n_cand, sizeof(beam_cand), cand_cmp);   /* [recalled] std::sort; ties broken by decoder index */
                    int cur_c = 0; int src_of[WO_MAX_DECODERS];
                    for (int j = 0; j < n_cur; j++) {
                        wo_decoder *D = &s->dec[j]; src_of[j] = -1;
                        if (D->completed || D->failed) continue;
                        if (cur_c >= n_cand) cur_c = 0;
                        beam_cand *c = &cands[cur_c++];
                        while (n_cand > cur_c && seq_tokens_equal(&cands[cur_c].seq, &c->seq) && i > 0) cur_c++;
                        D->seek_delta = c->seek_delta; D->has_ts = c->has_ts; seq_copy(&D->seq, &c->seq);
                        src_of[j] = c->decoder_idx;
                        kv_copy(s, c->decoder_idx, WO_MAX_DECODERS + j, n_prompt + i);
                    }
                    for (int j = 0; j < n_cur; j++) if (src_of[j] >= 0) kv_copy(s, WO_MAX_DECODERS + j, j, n_prompt + i);
                }
                for (int j = 0; j < n_cur; j++) {
                    wo_decoder *D = &s->dec[j];
                    if (D->completed || D->failed) continue;
                    const wo_token_data tk = D->seq.tok[D->seq.n - 1];
                    if (tk.id > m->t_beg) {
                        const int sd_new = 2 * (tk.id - m->t_beg);
                        if (D->has_ts && D->seek_delta > sd_new && D->seq.result_len < i) { D->failed = 1; continue; }
                        D->seek_delta = sd_new; D->seq.result_len = i + 1; D->has_ts = 1;
                    }
                    if (tk.id == m->t_eot || (p->max_tokens > 0 && i >= p->max_tokens) || (D->has_ts && seek + D->seek_delta + 100 >= seek_end)) {
                        if (D->seq.result_len == 0 && !p->no_timestamps) {
                            if (seek + D->seek_delta + 100 >= seek_end) D->seq.result_len = i + 1;
                            else { D->failed = 1; continue; }
                        }
                        if (p->single_segment || p->no_timestamps) { D->seq.result_len = i + 1; D->seek_delta = 100 * WO_CHUNK; }
                        D->completed = 1; continue;
                    }
                    if (i == n_max - 1 && (D->seq.result_len == 0 || D->seek_delta < 100 * WO_CHUNK / 2)) { D->failed = 1; continue; }
                }
                int all_done = 1;
                for (int j = 0; j < n_cur; j++) if (!s->dec[j].completed && !s->dec[j].failed) all_done = 0;
                if (all_done) break;
                const int n_past = n_prompt + i;
                for (int j = 0; j < n_cur; j++) {
                    wo_decoder *D = &s->dec[j];
                    if (D->failed || D->completed) continue;
                    int32_t tkn = D->seq.tok[D->seq.n - 1].id;
                    if (wo_decode(s, j, &tkn, 1, n_past, 0) != 0) { rc = -8; goto done; }
                    int32_t ids[512]; int nn = D->seq.n < 512 ? D->seq.n : 512;
                    for (int q = 0; q < nn; q++) ids[q] = D->seq.tok[D->seq.n - nn + q].id;
                    wo_process_logits(m, p, s->logits, ids, nn, D->has_ts, D->seek_delta, t_cur, D->logits, D->logprobs, D->probs);
                }
            }
            {
                double best_score = -INFINITY;
                for (int j = 0; j < n_cur; j++) {
                    wo_decoder *D = &s->dec[j];
                    if (D->failed) continue;
                    D->seq.n = D->seq.result_len;
                    seq_score(p, &D->seq);
                    if (D->seq.result_len > 32 && D->seq.entropy < p->entropy_thold) { D->failed = 1; continue; }
                    if (best_score < D->seq.score) { best_score = D->seq.score; best_id = j; }
                }
            }
            int success = 1;
            if (it != n_temps - 1) { wo_decoder *D = &s->dec[best_id]; if (D->failed || D->seq.avg_logprobs < p->logprob_thold) success = 0; }
            if (success) break;
        }
        {
            wo_decoder *B = &s->dec[best_id];
            int seek_delta = B->seek_delta; const int result_len = B->seq.result_len;
            const wo_token_data *tc = B->seq.tok; const int n_tc = B->seq.n;
            {   /* text context carried into the next window */
                int32_t *np = (int32_t *)malloc(sizeof(int32_t) * (n_prompt + result_len + 1)); int k = 0;
                if (n_prompt > 0 && prompt[0] == m->t_prev) for (int i = 1; i < n_prompt - n_init; i++) np[k++] = prompt[i];
                for (int i = 0; i < result_len && i < n_tc; i++) np[k++] = tc[i].id;
                free(s->prompt_past); s->prompt_past = np; s->n_prompt_past = k; s->cap_prompt_past = n_prompt + result_len + 1;
            }
            if (n_tc > 0) {
                int i0 = 0; int64_t t0 = seek + 2 * (tc[0].tid - m->t_beg);
                size_t cap_text = 64; for (int i = 0; i < n_tc; i++) cap_text += m->tok_len[tc[i].id];
                char *text = (char *)malloc(cap_text); int tl = 0;
                for (int i = 0; i < n_tc; i++) {
                    if (p->print_special || tc[i].id < m->t_eot) { int l; const char *w = wo_token_to_str(m, tc[i].id, &l); memcpy(text + tl, w, l); tl += l; }
                    if (tc[i].id > m->t_beg && !p->single_segment) {
                        const int64_t t1 = seek + 2 * (tc[i].tid - m->t_beg);
                        if (tl > 0) push_segment(s, t0, t1, text, tl, tc + i0, i - i0 + 1);
                        tl = 0;
                        while (i < n_tc && tc[i].id > m->t_beg) i++;
                        i--; t0 = t1; i0 = i + 1;
                    }
                }
                if (tl > 0) push_segment(s, t0, seek + seek_delta, text, tl, tc + i0, n_tc - i0);
                free(text);
            }
            seek += seek_delta;
        }
    }
done:
    for (int i = 0; i < WO_MAX_DECODERS * WO_MAX_DECODERS; i++) free(cands[i].seq.tok);
    free(cands); free(topk); free(prompt);
    return rc;
}

int wo_full_n_segments(const wo_state *s) { return s->n_seg; }
int64_t wo_full_segment_t0(const wo_state *s, int i) { return s->seg[i].t0; }
int64_t wo_full_segment_t1(const wo_state *s, int i) { return s->seg[i].t1; }
const char *wo_full_segment_text(const wo_state *s, int i) { return s->seg[i].text; }
int wo_full_n_tokens(const wo_state *s, int i) { return s->seg[i].n_tok; }
wo_token_data wo_full_token_data(const wo_state *s, int i, int j) { return s->seg[i].tok[j]; }
int wo_full_lang_id(const wo_state *s) { return s->lang_id; }
int wo_n_fallbacks(const wo_state *s) { return s->n_fallbacks; }
