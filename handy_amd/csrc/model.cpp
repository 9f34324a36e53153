// ggml Whisper file -> device tensors. The container layout is the legacy ggml ".bin" format of the
// files Handy's catalog downloads (reference: src-tauri/src/managers/model.rs:71-146: f16, Q4_1 and
// Q5_0 variants). Quantised tensors are expanded to fp16 once, at load time: the large matrices on the device from
// their uploaded blocks (k_load.hip), the few small vectors on the host.
#include "model.h"
#include <atomic>

#include <math.h>
#include <string.h>

#include <algorithm>
#include <memory>

namespace wmi {

// Whether block-quantised model files keep their decoder matrices resident as blocks next to the fp16 copies (for the
// few-row decode GEMMs). Off unless asked for: the few-row kernels are bound by launch and load latency, not by weight
// bytes, and expanding blocks in registers costs them more than the bytes save (measured: +1.4 us per projection).
static std::atomic<int> g_quant_resident{-1};
void set_quant_resident(bool on) { g_quant_resident.store(on ? 1 : 0); }
bool quant_resident() {
    const int v = g_quant_resident.load();
    if (v >= 0) return v != 0;
    const char *e = getenv("WMI_QUANT_RESIDENT");
    return e && atoi(e) != 0;
}


static const char *k_lang[kNumLangs][2] = {
    {"en", "english"}, {"zh", "chinese"}, {"de", "german"}, {"es", "spanish"}, {"ru", "russian"}, {"ko", "korean"}, {"fr", "french"},
    {"ja", "japanese"}, {"pt", "portuguese"}, {"tr", "turkish"}, {"pl", "polish"}, {"ca", "catalan"}, {"nl", "dutch"}, {"ar", "arabic"},
    {"sv", "swedish"}, {"it", "italian"}, {"id", "indonesian"}, {"hi", "hindi"}, {"fi", "finnish"}, {"vi", "vietnamese"}, {"he", "hebrew"},
    {"uk", "ukrainian"}, {"el", "greek"}, {"ms", "malay"}, {"cs", "czech"}, {"ro", "romanian"}, {"da", "danish"}, {"hu", "hungarian"},
    {"ta", "tamil"}, {"no", "norwegian"}, {"th", "thai"}, {"ur", "urdu"}, {"hr", "croatian"}, {"bg", "bulgarian"}, {"lt", "lithuanian"},
    {"la", "latin"}, {"mi", "maori"}, {"ml", "malayalam"}, {"cy", "welsh"}, {"sk", "slovak"}, {"te", "telugu"}, {"fa", "persian"},
    {"lv", "latvian"}, {"bn", "bengali"}, {"sr", "serbian"}, {"az", "azerbaijani"}, {"sl", "slovenian"}, {"kn", "kannada"},
    {"et", "estonian"}, {"mk", "macedonian"}, {"br", "breton"}, {"eu", "basque"}, {"is", "icelandic"}, {"hy", "armenian"},
    {"ne", "nepali"}, {"mn", "mongolian"}, {"bs", "bosnian"}, {"kk", "kazakh"}, {"sq", "albanian"}, {"sw", "swahili"},
    {"gl", "galician"}, {"mr", "marathi"}, {"pa", "punjabi"}, {"si", "sinhala"}, {"km", "khmer"}, {"sn", "shona"}, {"yo", "yoruba"},
    {"so", "somali"}, {"af", "afrikaans"}, {"oc", "occitan"}, {"ka", "georgian"}, {"be", "belarusian"}, {"tg", "tajik"},
    {"sd", "sindhi"}, {"gu", "gujarati"}, {"am", "amharic"}, {"yi", "yiddish"}, {"lo", "lao"}, {"uz", "uzbek"}, {"fo", "faroese"},
    {"ht", "haitian creole"}, {"ps", "pashto"}, {"tk", "turkmen"}, {"nn", "nynorsk"}, {"mt", "maltese"}, {"sa", "sanskrit"},
    {"lb", "luxembourgish"}, {"my", "myanmar"}, {"bo", "tibetan"}, {"tl", "tagalog"}, {"mg", "malagasy"}, {"as", "assamese"},
    {"tt", "tatar"}, {"haw", "hawaiian"}, {"ln", "lingala"}, {"ha", "hausa"}, {"ba", "bashkir"}, {"jw", "javanese"},
    {"su", "sundanese"}, {"yue", "cantonese"}};

int lang_id_from_str(const char *s) {
    if (!s) return -1;
    for (int i = 0; i < kNumLangs; i++)
        if (!strcmp(s, k_lang[i][0]) || !strcmp(s, k_lang[i][1])) return i;
    return -1;
}
const char *lang_str(int id) { return id >= 0 && id < kNumLangs ? k_lang[id][0] : nullptr; }
const char *lang_str_full(int id) { return id >= 0 && id < kNumLangs ? k_lang[id][1] : nullptr; }

static const char *k_non_speech[] = {"\"", "#", "(", ")", "*", "+", "/", ":", ";", "<", "=", ">", "@", "[", "\\", "]", "^", "_", "`", "{", "|", "}", "~",
                                     "\xe3\x80\x8c", "\xe3\x80\x8d", "\xe3\x80\x8e", "\xe3\x80\x8f", "<<", ">>", "<<<", ">>>", "--", "---", "-(", "-[", "('", "(\"",
                                     "((", "))", "(((", ")))", "[[", "]]", "{{", "}}", "\xe2\x99\xaa\xe2\x99\xaa", "\xe2\x99\xaa\xe2\x99\xaa\xe2\x99\xaa", "\xe2\x99\xa9",
                                     "\xe2\x99\xaa", "\xe2\x99\xab", "\xe2\x99\xac", "\xe2\x99\xad", "\xe2\x99\xae", "\xe2\x99\xaf"};

namespace {

struct TensorRef {
    int ttype = 0;
    std::vector<int> ne;
    const uint8_t *data = nullptr;
    size_t n = 0;
};

int block_bytes(int tt) {
    switch (tt) { case 2: return 18; case 3: return 20; case 6: return 22; case 7: return 24; case 8: return 34; }
    return 0;
}

inline float h2f(uint16_t h) { f16 v; memcpy(&v, &h, 2); return (float)v; }

// Expands one tensor to f32. Block formats (32 elements per block; low nibbles are elements 0..15, high 16..31):
// Q4_0 {f16 d, q[16]} d*(q-8) | Q4_1 {f16 d, f16 m, q[16]} d*q+m | Q5_0 {f16 d, u32 qh, q[16]} d*((q|b<<4)-16)
// Q5_1 {f16 d, f16 m, u32 qh, q[16]} d*(q|b<<4)+m | Q8_0 {f16 d, i8 q[32]} d*q
void expand_f32(const TensorRef &t, float *dst) {
    if (t.ttype == 0) { memcpy(dst, t.data, t.n * 4); return; }
    if (t.ttype == 1) { const uint16_t *s = (const uint16_t *)t.data; for (size_t i = 0; i < t.n; i++) dst[i] = h2f(s[i]); return; }
    const int bb = block_bytes(t.ttype), tt = t.ttype;
    for (size_t b = 0; b < t.n / 32; b++) {
        const uint8_t *p = t.data + b * bb; float *o = dst + b * 32;
        uint16_t dh; memcpy(&dh, p, 2); const float d = h2f(dh); float mn = 0; p += 2;
        if (tt == 3 || tt == 7) { uint16_t mh; memcpy(&mh, p, 2); mn = h2f(mh); p += 2; }
        if (tt == 8) { for (int j = 0; j < 32; j++) o[j] = (float)((int8_t)p[j]) * d; continue; }
        uint32_t qh = 0;
        if (tt == 6 || tt == 7) { memcpy(&qh, p, 4); p += 4; }
        for (int j = 0; j < 16; j++) {
            int lo = p[j] & 0x0F, hi = p[j] >> 4;
            if (tt == 6 || tt == 7) { lo |= ((qh >> j) & 1) << 4; hi |= ((qh >> (j + 16)) & 1) << 4; }
            if (tt == 2) { o[j] = (lo - 8) * d; o[j + 16] = (hi - 8) * d; }
            else if (tt == 6) { o[j] = (lo - 16) * d; o[j + 16] = (hi - 16) * d; }
            else { o[j] = lo * d + mn; o[j + 16] = hi * d + mn; }
        }
    }
}

struct Loader {
    const uint8_t *p, *end;
    bool ok = true;
    template <class T> T rd() {
        T v{};
        if (p + sizeof(T) > end) { ok = false; return v; }
        memcpy(&v, p, sizeof(T)); p += sizeof(T);
        return v;
    }
    const uint8_t *take(size_t n) {
        if ((size_t)(end - p) < n) { ok = false; return nullptr; }
        const uint8_t *r = p; p += n; return r;
    }
};

struct Uploader {
    Model *m;
    bool ok = true;
    template <class T> T *put(const T *host, size_t n, size_t n_alloc = 0) {
        if (n_alloc < n) n_alloc = n;
        void *d = nullptr;
        if (hipMalloc(&d, n_alloc * sizeof(T) + 256) != hipSuccess) { ok = false; log_msg(2, "load: hipMalloc of %zu bytes failed\n", n_alloc * sizeof(T)); return nullptr; }
        m->allocs.push_back(d); m->bytes += n_alloc * sizeof(T);
        if (n_alloc > n && hipMemset(d, 0, n_alloc * sizeof(T) + 256) != hipSuccess) ok = false;
        if (n && hipMemcpy(d, host, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) ok = false;
        return (T *)d;
    }

    // fp16 matrices are assembled on the device: f16 tensors go up straight from the file image, block-quantised ones
    // go up as blocks into a staging buffer and are expanded there.
    void *stage = nullptr;
    size_t stage_bytes = 0;
    f16 *alloc_f16(size_t n, size_t n_alloc = 0) {
        if (n_alloc < n) n_alloc = n;
        void *d = nullptr;
        if (hipMalloc(&d, n_alloc * sizeof(f16) + 256) != hipSuccess) { ok = false; log_msg(2, "load: hipMalloc of %zu bytes failed\n", n_alloc * sizeof(f16)); return nullptr; }
        m->allocs.push_back(d); m->bytes += n_alloc * sizeof(f16);
        if (n_alloc > n && hipMemset(d, 0, n_alloc * sizeof(f16) + 256) != hipSuccess) ok = false;
        return (f16 *)d;
    }
    void fill_f16(const TensorRef &t, f16 *dst) {
        if (!dst || !ok) return;
        if (t.ttype == 1) { if (hipMemcpy(dst, t.data, t.n * 2, hipMemcpyHostToDevice) != hipSuccess) ok = false; return; }
        if (t.ttype == 0) {
            std::vector<f16> v(t.n);
            const float *s = (const float *)t.data;
            for (size_t i = 0; i < t.n; i++) { float x; memcpy(&x, s + i, 4); v[i] = (f16)x; }
            if (hipMemcpy(dst, v.data(), t.n * 2, hipMemcpyHostToDevice) != hipSuccess) ok = false;
            return;
        }
        const size_t nb = t.n / 32, bytes = nb * (size_t)quant_block_bytes(t.ttype);
        if (bytes > stage_bytes) {
            if (stage) { (void)hipDeviceSynchronize(); (void)hipFree(stage); stage = nullptr; stage_bytes = 0; }
            if (hipMalloc(&stage, bytes + 256) != hipSuccess) { ok = false; log_msg(2, "load: hipMalloc of %zu staging bytes failed\n", bytes); return; }
            stage_bytes = bytes;
        }
        // the copy and the kernel both run on the null stream: the next tensor's copy waits for this one's expansion
        if (hipMemcpy(stage, t.data, bytes, hipMemcpyHostToDevice) != hipSuccess || !expand_blocks_f16(stage, t.ttype, (long)nb, dst, nullptr)) ok = false;
    }
    // The tensors `parts` (same block format, stacked along the rows) kept quantised: blocks through the staging buffer into the
    // aligned layout of QuantW. Returns ttype 0 when the parts are not block-quantised alike.
    QuantW put_quant(const std::vector<const TensorRef *> &parts) {
        QuantW q;
        if (!ok || parts.empty() || !parts[0]) return q;
        const int tt = parts[0]->ttype;
        size_t blocks = 0;
        for (const TensorRef *t : parts) { if (!t || t->ttype != tt || !quant_block_bytes(tt) || t->n % 32) return q; blocks += t->n / 32; }
        const size_t qb = tt == 8 ? 32 : 16;
        const bool high = tt == 6 || tt == 7;
        void *qs = nullptr, *dm = nullptr, *qh = nullptr;
        if (hipMalloc(&qs, blocks * qb + 256) != hipSuccess) { log_msg(2, "load: hipMalloc of %zu bytes failed\n", blocks * qb); ok = false; return q; }
        m->allocs.push_back(qs);
        if (hipMalloc(&dm, blocks * 4 + 256) != hipSuccess) { ok = false; return q; }
        m->allocs.push_back(dm);
        if (high) { if (hipMalloc(&qh, blocks * 4 + 256) != hipSuccess) { ok = false; return q; } m->allocs.push_back(qh); }
        m->bytes += blocks * (qb + 4 + (high ? 4 : 0));
        size_t off = 0;
        for (const TensorRef *t : parts) {
            const size_t nb = t->n / 32, bytes = nb * (size_t)quant_block_bytes(tt);
            if (bytes > stage_bytes) {
                if (stage) { (void)hipDeviceSynchronize(); (void)hipFree(stage); stage = nullptr; stage_bytes = 0; }
                if (hipMalloc(&stage, bytes + 256) != hipSuccess) { ok = false; return q; }
                stage_bytes = bytes;
            }
            if (hipMemcpy(stage, t->data, bytes, hipMemcpyHostToDevice) != hipSuccess ||
                !quant_repack(stage, tt, (long)nb, (char *)qs + off * qb, (unsigned *)dm + off, high ? (unsigned *)qh + off : nullptr, nullptr)) { ok = false; return q; }
            off += nb;
        }
        q.qs = (const unsigned char *)qs; q.dm = (const unsigned *)dm; q.qh = (const unsigned *)qh; q.ttype = tt;
        return q;
    }
    bool finish() {
        if (hipDeviceSynchronize() != hipSuccess) ok = false;
        if (stage) { (void)hipFree(stage); stage = nullptr; stage_bytes = 0; }
        return ok;
    }
};

}  // namespace

Model::~Model() {
    (void)hipSetDevice(device);
    for (void *p : allocs) (void)hipFree(p);
}

VocabIds Model::vocab_ids() const {
    VocabIds v;
    v.n_vocab = hp.n_vocab; v.eot = vocab.eot; v.sot = vocab.sot; v.translate = vocab.translate; v.transcribe = vocab.transcribe;
    v.solm = vocab.solm; v.prev = vocab.prev; v.nosp = vocab.nosp; v.tok_not = vocab.tok_not; v.beg = vocab.beg; v.space = vocab.space;
    v.n_lang = kNumLangs;
    return v;
}

Model *load_model(const uint8_t *data, size_t size, int device) {
    Loader L{data, data + size};
    if (L.rd<uint32_t>() != 0x67676d6c || !L.ok) { log_msg(2, "load: bad magic (not a ggml whisper file)\n"); return nullptr; }
    std::unique_ptr<Model> M(new Model());
    M->device = device;
    HParams &hp = M->hp;
    int32_t h[11];
    for (int i = 0; i < 11; i++) h[i] = L.rd<int32_t>();
    if (!L.ok) { log_msg(2, "load: truncated header\n"); return nullptr; }
    hp.n_vocab = h[0]; hp.n_audio_ctx = h[1]; hp.n_audio_state = h[2]; hp.n_audio_head = h[3]; hp.n_audio_layer = h[4];
    hp.n_text_ctx = h[5]; hp.n_text_state = h[6]; hp.n_text_head = h[7]; hp.n_text_layer = h[8]; hp.n_mels = h[9]; hp.ftype = h[10];
    const int d = hp.n_audio_state;
    if (hp.n_vocab < 51864 || hp.n_vocab > 52000 || d <= 0 || d % 64 != 0 || d > 1280 || hp.n_text_state != d || hp.n_audio_head <= 0 ||
        d / hp.n_audio_head != kHeadDim || hp.n_text_head != hp.n_audio_head || hp.n_mels <= 0 || hp.n_mels > 128 || hp.n_mels % 8 != 0 ||
        hp.n_text_ctx <= 0 || hp.n_text_ctx > kMaxTextCtx || hp.n_audio_ctx <= 0 || hp.n_audio_ctx > 1500 || hp.n_audio_layer <= 0 ||
        hp.n_text_layer <= 0 || hp.n_audio_layer > 64 || hp.n_text_layer > 64) {
        log_msg(2, "load: unsupported hyper-parameters (n_vocab %d n_state %d/%d heads %d mels %d text_ctx %d audio_ctx %d)\n", hp.n_vocab, d,
                hp.n_text_state, hp.n_audio_head, hp.n_mels, hp.n_text_ctx, hp.n_audio_ctx);
        return nullptr;
    }
    const int n_mel = L.rd<int32_t>(), n_fft = L.rd<int32_t>();
    if (!L.ok || n_mel != hp.n_mels || n_fft != kNBins) { log_msg(2, "load: bad mel filter table %dx%d\n", n_mel, n_fft); return nullptr; }
    const float *filt = (const float *)L.take((size_t)n_mel * n_fft * 4);
    if (!L.ok) { log_msg(2, "load: truncated mel filters\n"); return nullptr; }
    std::vector<float> filters(filt, filt + (size_t)n_mel * n_fft);

    // vocabulary
    Vocab &V = M->vocab;
    V.n_vocab = hp.n_vocab;
    const int n_file = L.rd<int32_t>();
    if (!L.ok || n_file <= 0 || n_file > hp.n_vocab) { log_msg(2, "load: bad vocabulary size %d\n", n_file); return nullptr; }
    V.id_to_token.resize(hp.n_vocab);
    for (int i = 0; i < n_file; i++) {
        const uint32_t len = L.rd<uint32_t>();
        const uint8_t *s = L.take(len);
        if (!L.ok) { log_msg(2, "load: truncated vocabulary\n"); return nullptr; }
        V.id_to_token[i].assign((const char *)s, len);
        V.token_to_id[V.id_to_token[i]] = i;
    }
    if (V.multilingual()) {
        V.eot++; V.sot++;
        const int dt = V.num_languages() - 98;
        V.translate += dt; V.transcribe += dt; V.solm += dt; V.prev += dt; V.nosp += dt; V.tok_not += dt; V.beg += dt;
    }
    for (int i = n_file; i < hp.n_vocab; i++) {
        std::string w;
        if (i > V.beg) w = "[_TT_" + std::to_string(i - V.beg) + "]";
        else if (i == V.eot) w = "[_EOT_]";
        else if (i == V.sot) w = "[_SOT_]";
        else if (i == V.translate) w = "[_TRANSLATE_]";
        else if (i == V.transcribe) w = "[_TRANSCRIBE_]";
        else if (i == V.solm) w = "[_SOLM_]";
        else if (i == V.prev) w = "[_PREV_]";
        else if (i == V.nosp) w = "[_NOSP_]";
        else if (i == V.tok_not) w = "[_NOT_]";
        else if (i == V.beg) w = "[_BEG_]";
        else if (i > V.sot && i <= V.sot + V.num_languages()) w = std::string("[_LANG_") + lang_str(i - V.sot - 1) + "]";
        else w = "[_extra_token_" + std::to_string(i) + "]";
        V.id_to_token[i] = w;
        V.token_to_id[w] = i;
    }
    { auto it = V.token_to_id.find(" "); V.space = it == V.token_to_id.end() ? -1 : it->second; }

    // tensor directory
    std::unordered_map<std::string, TensorRef> T;
    while (L.p < L.end) {
        const int n_dims = L.rd<int32_t>(), name_len = L.rd<int32_t>(), tt = L.rd<int32_t>();
        if (!L.ok || n_dims < 1 || n_dims > 4 || name_len <= 0 || name_len > 128) { log_msg(2, "load: corrupt tensor header\n"); return nullptr; }
        TensorRef t; t.ttype = tt; t.n = 1;
        for (int i = 0; i < n_dims; i++) { const int e = L.rd<int32_t>(); if (e <= 0) L.ok = false; t.ne.push_back(e); t.n *= (size_t)e; }
        const uint8_t *nm = L.take(name_len);
        if (!L.ok) { log_msg(2, "load: corrupt tensor header\n"); return nullptr; }
        std::string name((const char *)nm, name_len);
        size_t nbytes;
        if (tt == 0) nbytes = t.n * 4;
        else if (tt == 1) nbytes = t.n * 2;
        else if (block_bytes(tt) && t.n % 32 == 0) nbytes = t.n / 32 * block_bytes(tt);
        else { log_msg(2, "load: tensor '%s' has unsupported type %d\n", name.c_str(), tt); return nullptr; }
        t.data = L.take(nbytes);
        if (!L.ok) { log_msg(2, "load: file truncated inside tensor '%s'\n", name.c_str()); return nullptr; }
        T[name] = t;
    }
    bool missing = false;
    auto need = [&](const std::string &name, size_t n) -> const TensorRef * {
        auto it = T.find(name);
        if (it == T.end() || it->second.n != n) { log_msg(2, "load: tensor '%s' missing or of wrong size\n", name.c_str()); missing = true; return nullptr; }
        return &it->second;
    };
    auto f32v = [&](const std::string &name, size_t n) {
        std::vector<float> v(n, 0.0f);
        if (const TensorRef *t = need(name, n)) expand_f32(*t, v.data());
        return v;
    };

    if (hipSetDevice(device) != hipSuccess) { log_msg(2, "load: cannot select HIP device %d\n", device); return nullptr; }
    Uploader U{M.get()};
    auto f16_into = [&](const std::string &name, size_t n, f16 *dev) {      // f16 tensors are copied bit-for-bit
        if (const TensorRef *t = need(name, n)) U.fill_f16(*t, dev);
    };
    const size_t dd = (size_t)d * d;
    {   // mel tables
        std::vector<int> lo(n_mel, 0), hi(n_mel, 0);
        for (int j = 0; j < n_mel; j++) {
            int a = -1, b = -1;
            for (int k = 0; k < n_fft; k++) if (filters[(size_t)j * n_fft + k] != 0.0f) { if (a < 0) a = k; b = k; }
            if (a >= 0) { lo[j] = a; hi[j] = b + 1; }
        }
        std::vector<float> hann(kNFft);
        std::vector<float2> tw(kNFft);
        for (int i = 0; i < kNFft; i++) {
            hann[i] = (float)(0.5 * (1.0 - cosf((float)((2.0 * M_PI * i) / kNFft))));
            tw[i] = make_float2((float)cos(2.0 * M_PI * i / kNFft), (float)sin(2.0 * M_PI * i / kNFft));
        }
        M->mel.filters = U.put(filters.data(), filters.size());
        M->mel.flo = U.put(lo.data(), lo.size()); M->mel.fhi = U.put(hi.data(), hi.size());
        M->mel.hann = U.put(hann.data(), hann.size()); M->mel.tw400 = U.put(tw.data(), tw.size());
        std::vector<float> packed; std::vector<int> meta(n_mel, 0);
        for (int j = 0; j < n_mel; j++) {
            meta[j] = lo[j] | ((hi[j] - lo[j]) << 8) | ((int)packed.size() << 16);
            packed.insert(packed.end(), filters.begin() + (size_t)j * n_fft + lo[j], filters.begin() + (size_t)j * n_fft + hi[j]);
        }
        M->mel.fpacked = nullptr; M->mel.fmeta = nullptr; M->mel.n_packed = 0;
        if (packed.size() <= 1024 && n_mel <= 128) {           // the log-mel kernel's LDS table; wider tables are read densely
            M->mel.n_packed = (int)packed.size();
            if (packed.empty()) packed.push_back(0.0f);
            M->mel.fpacked = U.put(packed.data(), packed.size()); M->mel.fmeta = U.put(meta.data(), meta.size());
        }
        M->mel.n_mel = n_mel;
    }
    {   // conv stem: [out][in][3] -> [out][tap*in + c], K padded to the GEMM K-step with zero columns
        const int K1 = 3 * n_mel, K1p = round_up(K1, 64);
        M->conv1_kpad = K1p;
        std::vector<float> w = f32v("encoder.conv1.weight", (size_t)d * n_mel * 3);
        std::vector<f16> r((size_t)d * K1p, (f16)0.0f);
        for (int o = 0; o < d; o++) for (int c = 0; c < n_mel; c++) for (int k = 0; k < 3; k++) r[(size_t)o * K1p + k * n_mel + c] = (f16)w[((size_t)o * n_mel + c) * 3 + k];
        M->conv1_w = U.put(r.data(), r.size());
        w = f32v("encoder.conv2.weight", dd * 3);
        r.assign(dd * 3, (f16)0.0f);
        for (int o = 0; o < d; o++) for (int c = 0; c < d; c++) for (int k = 0; k < 3; k++) r[(size_t)o * 3 * d + (size_t)k * d + c] = (f16)w[((size_t)o * d + c) * 3 + k];
        M->conv2_w = U.put(r.data(), r.size());
        auto b1 = f32v("encoder.conv1.bias", d), b2 = f32v("encoder.conv2.bias", d);
        M->conv1_b = U.put(b1.data(), b1.size()); M->conv2_b = U.put(b2.data(), b2.size());
        auto pe = f32v("encoder.positional_embedding", (size_t)hp.n_audio_ctx * d);
        M->e_pe = U.put(pe.data(), pe.size());
        auto lw = f32v("encoder.ln_post.weight", d), lb = f32v("encoder.ln_post.bias", d);
        M->ln_post_w = U.put(lw.data(), lw.size()); M->ln_post_b = U.put(lb.data(), lb.size());
    }
    auto put_f32 = [&](const std::string &name, size_t n) { auto v = f32v(name, n); return U.put(v.data(), v.size()); };
    auto put_f16 = [&](const std::string &name, size_t n) { f16 *v = U.alloc_f16(n); f16_into(name, n, v); return v; };
    auto put_qkv = [&](const std::string &pre, f16 **w, float **b) {
        f16 *v = U.alloc_f16(3 * dd);
        if (v) { f16_into(pre + "query.weight", dd, v); f16_into(pre + "key.weight", dd, v + dd); f16_into(pre + "value.weight", dd, v + 2 * dd); }
        *w = v;
        std::vector<float> bb(3 * (size_t)d, 0.0f);
        auto q = f32v(pre + "query.bias", d), vv = f32v(pre + "value.bias", d);
        std::copy(q.begin(), q.end(), bb.begin()); std::copy(vv.begin(), vv.end(), bb.begin() + 2 * d);
        *b = U.put(bb.data(), bb.size());
    };
    M->enc.resize(hp.n_audio_layer);
    for (int i = 0; i < hp.n_audio_layer && !missing; i++) {
        EncLayer &E = M->enc[i];
        const std::string p = "encoder.blocks." + std::to_string(i) + ".";
        E.ln1_w = put_f32(p + "attn_ln.weight", d); E.ln1_b = put_f32(p + "attn_ln.bias", d);
        put_qkv(p + "attn.", &E.wqkv, &E.bqkv);
        E.wo = put_f16(p + "attn.out.weight", dd); E.bo = put_f32(p + "attn.out.bias", d);
        E.ln2_w = put_f32(p + "mlp_ln.weight", d); E.ln2_b = put_f32(p + "mlp_ln.bias", d);
        E.fc1 = put_f16(p + "mlp.0.weight", 4 * dd); E.b1 = put_f32(p + "mlp.0.bias", 4 * (size_t)d);
        E.fc2 = put_f16(p + "mlp.2.weight", 4 * dd); E.b2 = put_f32(p + "mlp.2.bias", d);
    }
    M->dec.resize(hp.n_text_layer);
    const bool keep_quant = quant_resident() && d % 64 == 0;
    {
        f16 *xkv = U.alloc_f16((size_t)hp.n_text_layer * 2 * dd);
        std::vector<float> xkvb((size_t)hp.n_text_layer * 2 * d, 0.0f);
        for (int i = 0; i < hp.n_text_layer && !missing; i++) {
            DecLayer &D = M->dec[i];
            const std::string p = "decoder.blocks." + std::to_string(i) + ".";
            D.ln1_w = put_f32(p + "attn_ln.weight", d); D.ln1_b = put_f32(p + "attn_ln.bias", d);
            put_qkv(p + "attn.", &D.wqkv, &D.bqkv);
            D.wo = put_f16(p + "attn.out.weight", dd); D.bo = put_f32(p + "attn.out.bias", d);
            D.lnx_w = put_f32(p + "cross_attn_ln.weight", d); D.lnx_b = put_f32(p + "cross_attn_ln.bias", d);
            D.xq = put_f16(p + "cross_attn.query.weight", dd); D.bxq = put_f32(p + "cross_attn.query.bias", d);
            D.xo = put_f16(p + "cross_attn.out.weight", dd); D.bxo = put_f32(p + "cross_attn.out.bias", d);
            D.ln2_w = put_f32(p + "mlp_ln.weight", d); D.ln2_b = put_f32(p + "mlp_ln.bias", d);
            D.fc1 = put_f16(p + "mlp.0.weight", 4 * dd); D.b1 = put_f32(p + "mlp.0.bias", 4 * (size_t)d);
            D.fc2 = put_f16(p + "mlp.2.weight", 4 * dd); D.b2 = put_f32(p + "mlp.2.bias", d);
            if (keep_quant && !missing) {       // block-quantised files: the decoder's matrices stay resident as blocks too (few-row decode reads those)
                auto tr = [&](const std::string &name) -> const TensorRef * { auto it = T.find(name); return it == T.end() ? nullptr : &it->second; };
                D.q_wqkv = U.put_quant({tr(p + "attn.query.weight"), tr(p + "attn.key.weight"), tr(p + "attn.value.weight")});
                D.q_wo = U.put_quant({tr(p + "attn.out.weight")});
                D.q_xq = U.put_quant({tr(p + "cross_attn.query.weight")});
                D.q_xo = U.put_quant({tr(p + "cross_attn.out.weight")});
                D.q_fc1 = U.put_quant({tr(p + "mlp.0.weight")});
                D.q_fc2 = U.put_quant({tr(p + "mlp.2.weight")});
            }
            if (xkv) {
                f16_into(p + "cross_attn.key.weight", dd, xkv + (size_t)(2 * i) * dd);
                f16_into(p + "cross_attn.value.weight", dd, xkv + (size_t)(2 * i + 1) * dd);
            }
            auto vb = f32v(p + "cross_attn.value.bias", d);
            std::copy(vb.begin(), vb.end(), xkvb.begin() + (size_t)(2 * i + 1) * d);
        }
        M->xkv_w = xkv;
        M->xkv_b = U.put(xkvb.data(), xkvb.size());
    }
    {
        M->n_vocab_pad = round_up(hp.n_vocab, 128);
        M->d_te = U.alloc_f16((size_t)hp.n_vocab * d, (size_t)M->n_vocab_pad * d);      // rows past n_vocab stay zero
        f16_into("decoder.token_embedding.weight", (size_t)hp.n_vocab * d, M->d_te);
        M->d_pe = put_f32("decoder.positional_embedding", (size_t)hp.n_text_ctx * d);
        M->d_ln_w = put_f32("decoder.ln.weight", d); M->d_ln_b = put_f32("decoder.ln.bias", d);
    }
    {   // non-speech token set as a bitmap: each symbol and " "+symbol, plus " -" and " '"
        std::vector<unsigned> bm((hp.n_vocab + 31) / 32 + 1, 0u);
        auto mark = [&](const std::string &s) { auto it = V.token_to_id.find(s); if (it != V.token_to_id.end() && it->second < n_file) bm[it->second >> 5] |= 1u << (it->second & 31); };
        for (const char *s : k_non_speech) { mark(s); mark(std::string(" ") + s); }
        mark(" -"); mark(" '");
        M->nst_bitmap = U.put(bm.data(), bm.size());
    }
    if (!U.finish() || missing) return nullptr;
    switch (hp.n_audio_layer) {
        case 4: M->type_name = "tiny"; break; case 6: M->type_name = "base"; break; case 12: M->type_name = "small"; break;
        case 24: M->type_name = "medium"; break; case 32: M->type_name = "large"; break; default: M->type_name = "custom"; break;
    }
    return M.release();
}

// Greedy longest-match over the vocabulary, preferring a leading-space form at word starts.
std::vector<int> tokenize(const Vocab &v, const std::string &text) {
    std::vector<int> out;
    size_t i = 0;
    size_t max_len = 0;
    for (const auto &kv : v.token_to_id) max_len = std::max(max_len, kv.first.size());
    while (i < text.size()) {
        size_t best = 0; int id = -1;
        for (size_t l = std::min(max_len, text.size() - i); l > 0; l--) {
            auto it = v.token_to_id.find(text.substr(i, l));
            if (it != v.token_to_id.end() && it->second < v.eot) { best = l; id = it->second; break; }
        }
        if (id < 0) { i++; continue; }
        out.push_back(id); i += best;
    }
    return out;
}

}  // namespace wmi
