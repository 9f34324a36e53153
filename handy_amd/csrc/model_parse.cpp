// Host-only half of the model loader: see model_parse.h.
#include "model_parse.h"

#include <math.h>
#include <string.h>

namespace wmi {

void log_msg(int level, const char *fmt, ...);

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


int ggml_block_bytes(int tt) {
    switch (tt) { case 2: return 18; case 3: return 20; case 6: return 22; case 7: return 24; case 8: return 34; }
    return 0;
}

namespace {

// IEEE half -> float by hand: this file also builds with compilers that have no _Float16
inline float h2f(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 31u, man = h & 1023u;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else { int e = -1; uint32_t mm = man; do { e++; mm <<= 1; } while (!(mm & 1024u)); bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((mm & 1023u) << 13); }
    } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((exp + 112u) << 23) | (man << 13);
    float f; memcpy(&f, &bits, 4); return f;
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


}  // namespace

// Expands one tensor to f32. Block formats (32 elements per block; low nibbles are elements 0..15, high 16..31):
// Q4_0 {f16 d, q[16]} d*(q-8) | Q4_1 {f16 d, f16 m, q[16]} d*q+m | Q5_0 {f16 d, u32 qh, q[16]} d*((q|b<<4)-16)
// Q5_1 {f16 d, f16 m, u32 qh, q[16]} d*(q|b<<4)+m | Q8_0 {f16 d, i8 q[32]} d*q
void expand_f32(const TensorRef &t, float *dst) {
    if (t.ttype == 0) { memcpy(dst, t.data, t.n * 4); return; }
    if (t.ttype == 1) { for (size_t i = 0; i < t.n; i++) { uint16_t h; memcpy(&h, t.data + 2 * i, 2); dst[i] = h2f(h); } return; }
    const int bb = ggml_block_bytes(t.ttype), tt = t.ttype;
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


bool parse_model_file(const uint8_t *data, size_t size, ParsedFile &F) {
    Loader L{data, data + size};
    if (!data || L.rd<uint32_t>() != 0x67676d6c || !L.ok) { log_msg(2, "load: bad magic (not a ggml whisper file)\n"); return false; }
    HParams &hp = F.hp;
    int32_t h[11];
    for (int i = 0; i < 11; i++) h[i] = L.rd<int32_t>();
    if (!L.ok) { log_msg(2, "load: truncated header\n"); return false; }
    hp.n_vocab = h[0]; hp.n_audio_ctx = h[1]; hp.n_audio_state = h[2]; hp.n_audio_head = h[3]; hp.n_audio_layer = h[4];
    hp.n_text_ctx = h[5]; hp.n_text_state = h[6]; hp.n_text_head = h[7]; hp.n_text_layer = h[8]; hp.n_mels = h[9]; hp.ftype = h[10];
    const int d = hp.n_audio_state;
    if (hp.n_vocab < 51864 || hp.n_vocab > 52000 || d <= 0 || d % 64 != 0 || d > 1280 || hp.n_text_state != d || hp.n_audio_head <= 0 ||
        d / hp.n_audio_head != 64 || d % hp.n_audio_head != 0 || hp.n_text_head != hp.n_audio_head || hp.n_mels <= 0 || hp.n_mels > 128 || hp.n_mels % 8 != 0 ||
        hp.n_text_ctx <= 0 || hp.n_text_ctx > 448 || hp.n_audio_ctx <= 0 || hp.n_audio_ctx > 1500 || hp.n_audio_layer <= 0 ||
        hp.n_text_layer <= 0 || hp.n_audio_layer > 64 || hp.n_text_layer > 64) {
        log_msg(2, "load: unsupported hyper-parameters (n_vocab %d n_state %d/%d heads %d mels %d text_ctx %d audio_ctx %d)\n", hp.n_vocab, d,
                hp.n_text_state, hp.n_audio_head, hp.n_mels, hp.n_text_ctx, hp.n_audio_ctx);
        return false;
    }
    const int n_mel = L.rd<int32_t>(), n_fft = L.rd<int32_t>();
    if (!L.ok || n_mel != hp.n_mels || n_fft != 201) { log_msg(2, "load: bad mel filter table %dx%d\n", n_mel, n_fft); return false; }
    const uint8_t *filt = L.take((size_t)n_mel * n_fft * 4);
    if (!L.ok) { log_msg(2, "load: truncated mel filters\n"); return false; }
    F.filters.resize((size_t)n_mel * n_fft);
    memcpy(F.filters.data(), filt, F.filters.size() * 4);          // the table sits at an arbitrary byte offset of the image

    // vocabulary
    Vocab &V = F.vocab;
    V.n_vocab = hp.n_vocab;
    const int n_file = L.rd<int32_t>();
    if (!L.ok || n_file <= 0 || n_file > hp.n_vocab) { log_msg(2, "load: bad vocabulary size %d\n", n_file); return false; }
    F.n_file_vocab = n_file;
    V.id_to_token.resize(hp.n_vocab);
    for (int i = 0; i < n_file; i++) {
        const uint32_t len = L.rd<uint32_t>();
        if (len > 4096) L.ok = false;                              // no token is longer than a few dozen bytes
        const uint8_t *s = L.ok ? L.take(len) : nullptr;
        if (!L.ok) { log_msg(2, "load: truncated or corrupt vocabulary\n"); return false; }
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
        else if (i > V.sot && i <= V.sot + V.num_languages() && lang_str(i - V.sot - 1)) w = std::string("[_LANG_") + lang_str(i - V.sot - 1) + "]";
        else w = "[_extra_token_" + std::to_string(i) + "]";
        V.id_to_token[i] = w;
        V.token_to_id[w] = i;
    }
    { auto it = V.token_to_id.find(" "); V.space = it == V.token_to_id.end() ? -1 : it->second; }

    // tensor directory
    while (L.p < L.end) {
        const int n_dims = L.rd<int32_t>(), name_len = L.rd<int32_t>(), tt = L.rd<int32_t>();
        if (!L.ok || n_dims < 1 || n_dims > 4 || name_len <= 0 || name_len > 128) { log_msg(2, "load: corrupt tensor header\n"); return false; }
        TensorRef t; t.ttype = tt; t.n = 1;
        for (int i = 0; i < n_dims; i++) {
            const int e = L.rd<int32_t>();
            if (e <= 0 || t.n > ((size_t)1 << 40) / (size_t)(e > 0 ? e : 1)) L.ok = false;       // no product of dimensions near an overflow
            t.ne.push_back(e); t.n *= (size_t)(e > 0 ? e : 1);
        }
        const uint8_t *nm = L.ok ? L.take(name_len) : nullptr;
        if (!L.ok) { log_msg(2, "load: corrupt tensor header\n"); return false; }
        std::string name((const char *)nm, name_len);
        size_t nbytes;
        if (tt == 0) nbytes = t.n * 4;
        else if (tt == 1) nbytes = t.n * 2;
        else if (ggml_block_bytes(tt) && t.n % 32 == 0) nbytes = t.n / 32 * ggml_block_bytes(tt);
        else { log_msg(2, "load: tensor '%s' has unsupported type %d\n", name.c_str(), tt); return false; }
        t.data = L.take(nbytes);
        if (!L.ok) { log_msg(2, "load: file truncated inside tensor '%s'\n", name.c_str()); return false; }
        F.tensors[name] = t;
    }
    return true;
}

bool check_model_tensors(const ParsedFile &F) {
    const HParams &hp = F.hp;
    const size_t d = (size_t)hp.n_audio_state, dd = d * d;
    bool ok = true;
    auto need = [&](const std::string &name, size_t n) {
        auto it = F.tensors.find(name);
        if (it == F.tensors.end() || it->second.n != n) { log_msg(2, "load: tensor '%s' missing or of wrong size\n", name.c_str()); ok = false; }
    };
    need("encoder.conv1.weight", d * hp.n_mels * 3); need("encoder.conv1.bias", d);
    need("encoder.conv2.weight", dd * 3); need("encoder.conv2.bias", d);
    need("encoder.positional_embedding", (size_t)hp.n_audio_ctx * d);
    need("encoder.ln_post.weight", d); need("encoder.ln_post.bias", d);
    auto block = [&](const std::string &p, bool cross) {
        need(p + "attn_ln.weight", d); need(p + "attn_ln.bias", d);
        need(p + "attn.query.weight", dd); need(p + "attn.query.bias", d); need(p + "attn.key.weight", dd);
        need(p + "attn.value.weight", dd); need(p + "attn.value.bias", d); need(p + "attn.out.weight", dd); need(p + "attn.out.bias", d);
        need(p + "mlp_ln.weight", d); need(p + "mlp_ln.bias", d);
        need(p + "mlp.0.weight", 4 * dd); need(p + "mlp.0.bias", 4 * d); need(p + "mlp.2.weight", 4 * dd); need(p + "mlp.2.bias", d);
        if (cross) {
            need(p + "cross_attn_ln.weight", d); need(p + "cross_attn_ln.bias", d);
            need(p + "cross_attn.query.weight", dd); need(p + "cross_attn.query.bias", d); need(p + "cross_attn.key.weight", dd);
            need(p + "cross_attn.value.weight", dd); need(p + "cross_attn.value.bias", d);
            need(p + "cross_attn.out.weight", dd); need(p + "cross_attn.out.bias", d);
        }
    };
    for (int i = 0; i < hp.n_audio_layer && ok; i++) block("encoder.blocks." + std::to_string(i) + ".", false);
    for (int i = 0; i < hp.n_text_layer && ok; i++) block("decoder.blocks." + std::to_string(i) + ".", true);
    need("decoder.token_embedding.weight", (size_t)hp.n_vocab * d);
    need("decoder.positional_embedding", (size_t)hp.n_text_ctx * d);
    need("decoder.ln.weight", d); need("decoder.ln.bias", d);
    return ok;
}

// whisper.cpp's tokenize() (whisper.cpp, "static std::vector<whisper_vocab::id> tokenize", as recalled: the pinned source is
// not in the reference tree): the text is cut into words by the GPT-2 pre-tokenisation pattern
//   's|'t|'re|'ve|'m|'ll|'d| ?[[:alpha:]]+| ?[[:digit:]]+| ?[^\s[:alpha:][:digit:]]+|\s+(?!\S)|\s+
// (std::regex, ECMAScript, "C" locale: classes are ASCII, bytes >= 0x80 are neither letters nor digits nor space), and every
// word is covered left to right by the longest vocabulary entries; a byte no entry starts at is skipped. The pattern is
// scanned by hand here: ordered alternatives, first match wins, and every byte starts some match, so the words tile the text.
// Control tokens (ids >= eot) are never produced from text.
namespace {
inline bool is_space(unsigned char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }
inline bool is_alpha(unsigned char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }
inline bool is_digit(unsigned char c) { return c >= '0' && c <= '9'; }
inline bool is_other(unsigned char c) { return !is_space(c) && !is_alpha(c) && !is_digit(c); }

size_t next_word_len(const std::string &t, size_t i) {
    const size_t n = t.size();
    if (t[i] == '\'') {
        static const char *const k_contractions[] = {"'s", "'t", "'re", "'ve", "'m", "'ll", "'d"};
        for (const char *c : k_contractions) {
            const size_t l = strlen(c);
            if (t.compare(i, l, c) == 0) return l;
        }
    }
    bool (*const classes[3])(unsigned char) = {is_alpha, is_digit, is_other};
    for (auto in_class : classes) {
        size_t j = i + (t[i] == ' ' && i + 1 < n && in_class((unsigned char)t[i + 1]) ? 1 : 0);
        const size_t first = j;
        while (j < n && in_class((unsigned char)t[j])) j++;
        if (j > first) return j - i;
    }
    size_t j = i;                                   // only white space starts here
    while (j < n && is_space((unsigned char)t[j])) j++;
    if (j < n && j - i > 1) return j - i - 1;       // \s+(?!\S): leave the last one to the word that follows
    return j - i;                                   // the run reaches the end of the text, or is a single character: \s+
}
}  // namespace

std::vector<int> tokenize(const Vocab &v, const std::string &text) {
    std::vector<int> out;
    for (size_t w = 0; w < text.size();) {
        const size_t wl = next_word_len(text, w);
        for (size_t i = w; i < w + wl;) {
            size_t j = w + wl;
            for (; j > i; j--) {
                const auto it = v.token_to_id.find(text.substr(i, j - i));
                if (it != v.token_to_id.end() && it->second < v.eot) { out.push_back(it->second); break; }
            }
            i = j > i ? j : i + 1;
        }
        w += wl;
    }
    return out;
}

}  // namespace wmi
