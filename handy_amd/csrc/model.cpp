// ggml Whisper file -> device tensors. The container layout is the legacy ggml ".bin" format of the
// files Handy's catalog downloads (reference: src-tauri/src/managers/model.rs:71-146: f16, Q4_1 and
// Q5_0 variants). Quantised tensors are expanded to fp16 once, at load time: the large matrices on the device from
// their uploaded blocks (k_load.hip), the few small vectors on the host.
#include "model.h"
#include "model_parse.h"
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


static const char *k_non_speech[] = {"\"", "#", "(", ")", "*", "+", "/", ":", ";", "<", "=", ">", "@", "[", "\\", "]", "^", "_", "`", "{", "|", "}", "~",
                                     "\xe3\x80\x8c", "\xe3\x80\x8d", "\xe3\x80\x8e", "\xe3\x80\x8f", "<<", ">>", "<<<", ">>>", "--", "---", "-(", "-[", "('", "(\"",
                                     "((", "))", "(((", ")))", "[[", "]]", "{{", "}}", "\xe2\x99\xaa\xe2\x99\xaa", "\xe2\x99\xaa\xe2\x99\xaa\xe2\x99\xaa", "\xe2\x99\xa9",
                                     "\xe2\x99\xaa", "\xe2\x99\xab", "\xe2\x99\xac", "\xe2\x99\xad", "\xe2\x99\xae", "\xe2\x99\xaf"};

namespace {

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
    ParsedFile F;
    if (!parse_model_file(data, size, F) || !check_model_tensors(F)) return nullptr;
    std::unique_ptr<Model> M(new Model());
    M->device = device;
    M->hp = F.hp;
    M->vocab = std::move(F.vocab);
    const HParams &hp = M->hp;
    Vocab &V = M->vocab;
    const int d = hp.n_audio_state, n_mel = hp.n_mels, n_fft = kNBins, n_file = F.n_file_vocab;
    const std::vector<float> &filters = F.filters;
    const std::unordered_map<std::string, TensorRef> &T = F.tensors;
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

}  // namespace wmi
