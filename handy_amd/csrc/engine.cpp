// Host side of the MI355X Whisper engine: workspaces, the batched encoder / decoder passes, and the
// whisper_full control loop restated as one resumable state machine per utterance so that B
// utterances share every kernel launch. The control flow follows whisper.cpp's whisper_full (bundled
// in whisper-rs-sys 0.11.1, reference src-tauri/Cargo.lock:7839-7848) as recalled; it is what
// Handy's TranscriptionManager::transcribe reaches (src-tauri/src/managers/transcription.rs:413-415).
#include "engine.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>

namespace wmi {

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ------------------------------------------------------------------------------------------ buffers
bool DevBuf::ensure(size_t bytes, bool zero) {
    if (bytes <= cap && p) return true;
    release();
    const size_t want = round_up_sz(bytes + 1024, 4096);
    if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; cap = 0; log_msg(2, "engine: hipMalloc(%zu) failed\n", want); return false; }
    cap = want;
    // the memset runs on the null stream; the engine's stream does not synchronise with it implicitly
    if (zero && (hipMemset(p, 0, want) != hipSuccess || hipDeviceSynchronize() != hipSuccess)) return false;
    return true;
}
void DevBuf::release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
bool PinnedBuf::ensure(size_t bytes) {
    if (bytes <= cap && p) return true;
    release();
    const size_t want = round_up_sz(bytes + 256, 4096);
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; cap = 0; log_msg(2, "engine: hipHostMalloc(%zu) failed\n", want); return false; }
    cap = want;
    return true;
}
void PinnedBuf::release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }

// ------------------------------------------------------------------------------------------ engine set-up
// Passes of up to this many rows take the split-K path for the narrow GEMMs (many small workgroups that request their
// whole weight slice at once, partial sums added by a consumer kernel); larger passes take the 64x64 tile with a fused
// epilogue, which accumulates K slice by slice in the same order -- the two paths give the same bits, so the threshold is
// a pure speed choice (WMI_SPLIT_ROWS overrides it for experiments): at 320 rows the partial sums of ten slices are five
// times the weight bytes.
static bool few_rows_path() {          // WMI_FEW_ROWS=0 turns the few-row kernels off (A/B measurements)
    static const bool v = [] { const char *e = getenv("WMI_FEW_ROWS"); return !e || atoi(e) != 0; }();
    return v;
}
static size_t split_max_rows() {
    static const size_t v = [] { const char *e = getenv("WMI_SPLIT_ROWS"); return e ? (size_t)atoi(e) : (size_t)64; }();
    return v;
}

Engine::Engine(Model *m, int max_utts, int max_dec, int enc_chunk) : model(m) {
    { const char *e = getenv("WMI_PASS_GRAPHS"); use_graphs_ = !e || atoi(e) != 0; }
    max_utts_ = std::max(1, max_utts);
    max_dec_ = std::min(8, std::max(1, max_dec));
    enc_chunk_ = enc_chunk > 0 ? enc_chunk : 128;        // big chunks: fewer, fuller GEMM launches (about 70 MB of workspace per utterance)
    enc_chunk_ = std::min(enc_chunk_, max_utts_);
    d_ = m->hp.n_audio_state; H_ = m->hp.n_audio_head; Le_ = m->hp.n_audio_layer; Ld_ = m->hp.n_text_layer;
    T_full_ = m->hp.n_audio_ctx; n_mel_ = m->hp.n_mels; V_ = m->hp.n_vocab; Vp_ = m->n_vocab_pad;
    Tp_full_ = round_up(T_full_, 128);
    if (hipSetDevice(m->device) != hipSuccess) return;
    if (hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking) != hipSuccess) return;
    for (int l = 0; l < kEncLanes; l++) if (hipStreamCreateWithFlags(&enc_streams_[l], hipStreamNonBlocking) != hipSuccess) return;
    if (hipStreamCreateWithFlags(&lane_stream_, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&lane_fork_, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&lane_join_, hipEventDisableTiming) != hipSuccess) return;
    if (const char *e = getenv("WMI_DEC_LANES")) set_decoder_lanes(atoi(e));
    prof_stream_ = stream_;
    ok_ = alloc_workspaces();
}

Engine::~Engine() {
    (void)hipSetDevice(model->device);
    if (stream_) (void)hipStreamSynchronize(stream_);
    clear_graphs();
    for (int l = 0; l < kEncLanes; l++) {
        if (enc_streams_[l]) (void)hipStreamSynchronize(enc_streams_[l]);
        for (DevBuf *b : {&ews_[l].melT, &ews_[l].h1, &ews_[l].x, &ews_[l].xn, &ews_[l].q, &ews_[l].k, &ews_[l].vt, &ews_[l].att, &ews_[l].hid, &ews_[l].enc16}) b->release();
    }
    for (DevBuf *b : {&pcm_, &mel_raw_, &mel_max_, &mel_utt_dev_, &mel_norm_tmp_, &enc32_,
                      &win_dev_, &slots_dev_, &ckv_, &dx_, &dxn_, &dq_, &datt_, &dhid_, &dxl_, &logits_, &probs_, &kpool_, &vpool_, &meta_dev_, &out_dev_, &sample_ws_, &splitk_ws_, &lane_ws_, &ckvt_})
        b->release();
    meta_host_.release(); out_host_.release(); probs_host_.release(); win_host_.release(); slots_host_.release();
    for (hipEvent_t e : enc_events_) (void)hipEventDestroy(e);
    for (auto &e : ev_pool_) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    if (stream_) (void)hipStreamDestroy(stream_);
    for (int l = 0; l < kEncLanes; l++) if (enc_streams_[l]) (void)hipStreamDestroy(enc_streams_[l]);
    if (lane_stream_) { (void)hipStreamSynchronize(lane_stream_); (void)hipStreamDestroy(lane_stream_); }
    if (lane_fork_) (void)hipEventDestroy(lane_fork_);
    if (lane_join_) (void)hipEventDestroy(lane_join_);
}

bool Engine::alloc_workspaces() {
    const int E = enc_chunk_, T = T_full_, d = d_;
    const size_t M1 = (size_t)E * (2 * T + 2), M2p = round_up(E * (T + 1), 256), Mp = round_up(E * T, 256);
    bool ok = true;
    const size_t qkv = (size_t)E * H_ * Tp_full_ * 64 * sizeof(f16);
    const int lanes = 1;                                        // the second lane's workspace is allocated when it is switched on
    for (int l = 0; l < lanes; l++) {
        EncWs &w = ews_[l];
        ok &= w.melT.ensure(((M1 + 512) * n_mel_ + 1024) * sizeof(f16));
        ok &= w.h1.ensure(((std::max(M1 + 2, 2 * M2p + 8) + 256) * d) * sizeof(f16));
        ok &= w.x.ensure(Mp * d * sizeof(float));
        ok &= w.xn.ensure(Mp * d * sizeof(f16));
        ok &= w.att.ensure(Mp * d * sizeof(f16));
        ok &= w.hid.ensure(Mp * 4 * d * sizeof(f16));
        ok &= w.enc16.ensure(Mp * d * sizeof(f16));
        ok &= w.q.ensure(qkv); ok &= w.k.ensure(qkv); ok &= w.vt.ensure(qkv);
    }
    ring_cap_ = 2 * max_utts_ + 4 * E;       // more than everything that can be queued at once, even after a wrap
    ok &= win_dev_.ensure(sizeof(WinDesc) * ring_cap_); ok &= slots_dev_.ensure(sizeof(int) * ring_cap_);
    ok &= win_host_.ensure(sizeof(WinDesc) * ring_cap_); ok &= slots_host_.ensure(sizeof(int) * ring_cap_);
    ok &= ckv_.ensure((size_t)max_utts_ * Ld_ * 2 * H_ * Tp_full_ * 64 * sizeof(f16));
    ok &= mel_max_.ensure(sizeof(unsigned) * max_utts_); ok &= mel_utt_dev_.ensure(sizeof(MelUtt) * max_utts_);
    // self-attention KV pool: pages of 16 tokens, shared by every sequence through block tables
    const int max_seqs = max_utts_ * max_dec_;
    const size_t page_bytes_all_layers = (size_t)Ld_ * 2 * H_ * kPageTokens * 64 * sizeof(f16);
    size_t want_pages = (size_t)max_seqs * kMaxPagesPerSeq + 8;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const size_t budget = free_b / 4;
        if (want_pages * page_bytes_all_layers > budget) want_pages = std::max<size_t>(budget / page_bytes_all_layers, (size_t)max_dec_ * kMaxPagesPerSeq + 8);
    }
    n_pages_ = (int)want_pages;
    layer_stride_ = (long)n_pages_ * H_ * kPageTokens * 64;
    ok &= kpool_.ensure((size_t)Ld_ * layer_stride_ * sizeof(f16)); ok &= vpool_.ensure((size_t)Ld_ * layer_stride_ * sizeof(f16));
    page_ref_.assign(n_pages_, 0);
    free_pages_.clear();
    for (int i = n_pages_ - 1; i >= 1; i--) free_pages_.push_back(i);       // page 0 stays unused: block tables are zero-filled
    lrows_cap_ = round_up(std::max(max_seqs, 8), 256);
    ok &= dxl_.ensure((size_t)lrows_cap_ * d * sizeof(f16));
    ok &= logits_.ensure((size_t)lrows_cap_ * Vp_ * sizeof(float));
    ok &= out_dev_.ensure((size_t)lrows_cap_ * (sizeof(SampleOut) + 8 * sizeof(TopkOut)));
    ok &= out_host_.ensure((size_t)lrows_cap_ * (sizeof(SampleOut) + 8 * sizeof(TopkOut)));
    ok &= sample_ws_.ensure(sample_workspace_bytes(lrows_cap_));
    logits_host_.assign(V_, 0.0f);
    ok &= ensure_rows(std::max(128, max_seqs));
    return ok;
}

bool Engine::set_encoder_lanes(int n) {
    n = std::max(1, std::min(n, kEncLanes));
    if (n > 1 && !ews_[1].x.p) {           // same sizes as lane 0
        (void)hipSetDevice(model->device);
        bool ok = true;
        const EncWs &a = ews_[0]; EncWs &w = ews_[1];
        ok &= w.melT.ensure(a.melT.cap); ok &= w.h1.ensure(a.h1.cap); ok &= w.x.ensure(a.x.cap); ok &= w.xn.ensure(a.xn.cap); ok &= w.att.ensure(a.att.cap);
        ok &= w.hid.ensure(a.hid.cap); ok &= w.enc16.ensure(a.enc16.cap); ok &= w.q.ensure(a.q.cap); ok &= w.k.ensure(a.k.cap); ok &= w.vt.ensure(a.vt.cap);
        if (!ok) return false;
    }
    encoder_lanes = n;
    return true;
}

bool Engine::ensure_rows(int rows) {
    if (rows <= rows_cap_) return true;
    const size_t R = round_up(rows, 256);
    bool ok = dx_.ensure(R * d_ * sizeof(float)) && dxn_.ensure(R * d_ * sizeof(f16)) && dq_.ensure(R * d_ * sizeof(f16)) &&
              datt_.ensure(R * d_ * sizeof(f16)) && dhid_.ensure(R * 4 * d_ * sizeof(f16));
    const size_t meta = R * sizeof(DecRow) + (size_t)lrows_cap_ * (sizeof(SampleRow) + sizeof(int) * (kMaxPagesPerSeq + 1)) + 2 * R * sizeof(int2) + 4096;
    ok = ok && meta_dev_.ensure(meta) && meta_host_.ensure(meta);
    // split-K partial sums: the largest ws[splits][rows][N] over the four weight shapes of a decoder layer (at d < 853 the
    // QKV and at d < 640 the fc1 projection split too, and their N is 3d and 4d)
    const int ws_rows = (int)std::min<size_t>(R, split_max_rows());
    size_t ws_bytes = 4096;
    for (const auto &nk : {std::pair<int, int>{3 * d_, d_}, {d_, d_}, {4 * d_, d_}, {d_, 4 * d_}})
        ws_bytes = std::max(ws_bytes, gemm_splitk_workspace_bytes(ws_rows, nk.first, nk.second));
    ok = ok && splitk_ws_.ensure(ws_bytes, false) && lane_ws_.ensure(ws_bytes, false);
    if (ok) rows_cap_ = (int)R;
    return ok;
}

// ------------------------------------------------------------------------------------------ profiling
void Engine::clear_graphs() {
    for (auto &kv : graphs_)
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    graphs_.clear();
}

void Engine::prof_begin(int cls, double work) {
    if (!profiling || !((profile_mask >> cls) & 1u)) return;
    if (ev_used_ == ev_pool_.size()) {
        EvPair e{};
        if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return;
        ev_pool_.push_back(e);
    }
    EvPair &e = ev_pool_[ev_used_];
    e.cls = cls; e.work = work;
    (void)hipEventRecord(e.a, prof_stream_);
    prof_open_ = true;
}
void Engine::prof_end() {
    if (!profiling || !prof_open_) return;
    (void)hipEventRecord(ev_pool_[ev_used_].b, prof_stream_);
    ev_used_++;
    prof_open_ = false;
    if (ev_used_ >= 200000) collect_stats();
}
void Engine::collect_stats() {
    if (!ev_used_) return;
    (void)hipStreamSynchronize(stream_);
    for (int l = 0; l < kEncLanes; l++) (void)hipStreamSynchronize(enc_streams_[l]);
    for (size_t i = 0; i < ev_used_; i++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev_pool_[i].a, ev_pool_[i].b) == hipSuccess) {
            wmi_kernel_stat &s = stats[ev_pool_[i].cls];
            s.ms += ms; s.work += ev_pool_[i].work; s.launches++;
        }
    }
    ev_used_ = 0;
}
void Engine::reset_stats() { collect_stats(); memset(stats, 0, sizeof stats); }

#define PROF(cls, work, call)                         \
    do {                                              \
        prof_begin(cls, work);                        \
        const bool ok__ = (call);                     \
        prof_end();                                   \
        if (!ok__) return false;                      \
    } while (0)

// ------------------------------------------------------------------------------------------ mel
bool Engine::compute_mel(const float *base, const std::vector<long> &offs, const std::vector<int> &n) {
    const int B = (int)n.size();
    if (B > max_utts_) return false;
    mel_utt_.assign(B, MelUtt{});
    utt_n_len_.assign(B, 0); utt_n_len_org_.assign(B, 0);
    long off = 0; int max_blocks = 1;
    double bytes = 0;
    for (int i = 0; i < B; i++) {
        MelUtt &u = mel_utt_[i];
        u.pcm_off = offs[i]; u.n = n[i];
        const int n_len = (n[i] + kSampleRate * kChunkSec) / kHop;
        int n_calc = (n[i] + kNFft / 2) / kHop + 1; if (n_calc > n_len) n_calc = n_len;
        u.n_calc = n_calc; u.n_frames = round_up(n_calc, kMelFramesPerBlock); u.mel_off = off;
        off += (long)u.n_frames * n_mel_;
        max_blocks = std::max(max_blocks, u.n_frames / kMelFramesPerBlock);
        utt_n_len_[i] = n_len; utt_n_len_org_[i] = 1 + (n[i] + kNFft / 2 - kNFft) / kHop;
        bytes += (double)n[i] * 4 + (double)u.n_frames * n_mel_ * 4;
    }
    if (!mel_raw_.ensure((size_t)off * sizeof(float), false)) return false;
    WMI_HIP_OK(hipMemcpyAsync(mel_utt_dev_.p, mel_utt_.data(), sizeof(MelUtt) * B, hipMemcpyHostToDevice, stream_));
    mel_user_ = false;
    PROF(WMI_K_MEL, bytes, mel_raw(base, mel_utt_dev_.as<MelUtt>(), B, max_blocks, model->mel, mel_raw_.as<float>(), mel_max_.as<unsigned>(), stream_));
    return true;
}

int Engine::pcm_to_mel(const float *pcm, int n) {
    if (!pcm || n < 0) return -1;
    (void)hipSetDevice(model->device);
    const double t0 = now_ms();
    if (!pcm_.ensure((size_t)std::max(n, 1) * sizeof(float), false)) return -1;
    if (n && hipMemcpyAsync(pcm_.p, pcm, (size_t)n * sizeof(float), hipMemcpyHostToDevice, stream_) != hipSuccess) return -1;
    if (!compute_mel(pcm_.as<float>(), {0}, {n})) return -1;
    if (hipStreamSynchronize(stream_) != hipSuccess) return -1;
    t_mel_ms += now_ms() - t0;
    return 0;
}

int Engine::mel_batch_device(const float *pcm_dev, const long long *offs, const int *n, int B, float *out_dev, int n_frames_out) {
    (void)hipSetDevice(model->device);
    if (out_dev && (n_frames_out <= 0 || n_frames_out % 4 != 0)) return -1;
    std::vector<long> o(offs, offs + B); std::vector<int> nn(n, n + B);
    if (!compute_mel(pcm_dev, o, nn)) return -1;
    if (out_dev) {      // the clamped, scaled log-mel -- what whisper_pcm_to_mel leaves -- for every utterance, into the caller's buffer
        prof_stream_ = stream_;
        prof_begin(WMI_K_MEL, 0.0);         // its bytes are counted with mel_raw_kernel's: PCM in, one log-mel out
        const bool ok = mel_normalize_batch(mel_raw_.as<float>(), mel_utt_dev_.as<MelUtt>(), B, n_mel_, mel_max_.as<unsigned>(), out_dev, n_frames_out, stream_);
        prof_end();
        if (!ok) return -1;
    }
    return hipStreamSynchronize(stream_) == hipSuccess ? 0 : -1;
}

int Engine::set_mel(const float *data, int n_len, int n_mel) {
    if (n_mel != n_mel_ || n_len <= 0 || !data) return -1;
    (void)hipSetDevice(model->device);
    if (!mel_raw_.ensure((size_t)n_len * n_mel * sizeof(float), false)) return -1;
    if (hipMemcpy(mel_raw_.p, data, (size_t)n_len * n_mel * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return -1;
    mel_utt_.assign(1, MelUtt{}); mel_utt_[0].n_frames = n_len; mel_utt_[0].n_calc = n_len;
    utt_n_len_.assign(1, n_len); utt_n_len_org_.assign(1, n_len);
    mel_user_ = true;
    return 0;
}

int Engine::get_mel(int utt, float *out, int n_frames) {
    if (utt < 0 || utt >= (int)mel_utt_.size() || n_frames <= 0) return -1;
    (void)hipSetDevice(model->device);
    const MelUtt &u = mel_utt_[utt];
    const size_t bytes = (size_t)n_frames * n_mel_ * sizeof(float);
    if (mel_user_) {
        for (int j = 0; j < n_mel_; j++)
            if (hipMemcpy(out + (size_t)j * n_frames, mel_raw_.as<float>() + (size_t)j * u.n_frames, sizeof(float) * std::min(n_frames, u.n_frames), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        return 0;
    }
    if (!mel_norm_tmp_.ensure(bytes, false)) return -1;
    if (!mel_normalize(mel_raw_.as<float>() + u.mel_off, n_mel_, u.n_frames, mel_max_.as<unsigned>() + utt, mel_norm_tmp_.as<float>(), n_frames, stream_)) return -1;
    if (hipMemcpyAsync(out, mel_norm_tmp_.p, bytes, hipMemcpyDeviceToHost, stream_) != hipSuccess) return -1;
    return hipStreamSynchronize(stream_) == hipSuccess ? 0 : -1;
}

// ------------------------------------------------------------------------------------------ encoder
hipEvent_t Engine::next_enc_event() {
    if (enc_events_used_ == enc_events_.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        enc_events_.push_back(e);
    }
    return enc_events_[enc_events_used_++];
}

bool Engine::encode_chunk(const std::vector<WinDesc> &wins, const std::vector<int> &slots, int T, bool want_f32, hipStream_t stream_, int lane) {
    // (the parameter shadows the member on purpose: every launch below goes to the caller's stream)
    const int E = (int)wins.size(), d = d_, T2 = 2 * T, Tp = Tp_full_;
    if (E <= 0) return true;
    if (E > enc_chunk_ || T <= 0 || T > T_full_ || lane < 0 || lane >= kEncLanes || !ews_[lane].x.p) return false;
    EncWs &w = ews_[lane];
    prof_stream_ = stream_;
    // chunk metadata goes through a pinned ring: the copy is asynchronous and a later chunk must not overwrite it
    if (ring_pos_ + E > ring_cap_) ring_pos_ = 0;
    WinDesc *hw = win_host_.as<WinDesc>() + ring_pos_, *dw = win_dev_.as<WinDesc>() + ring_pos_;
    int *hs = slots_host_.as<int>() + ring_pos_, *ds = slots_dev_.as<int>() + ring_pos_;
    ring_pos_ += E;
    memcpy(hw, wins.data(), sizeof(WinDesc) * E); memcpy(hs, slots.data(), sizeof(int) * E);
    WMI_HIP_OK(hipMemcpyAsync(dw, hw, sizeof(WinDesc) * E, hipMemcpyHostToDevice, stream_));
    WMI_HIP_OK(hipMemcpyAsync(ds, hs, sizeof(int) * E, hipMemcpyHostToDevice, stream_));
    const double win_bytes = (double)E * T2 * n_mel_ * 6;
    if (mel_user_) PROF(WMI_K_ENC_OTHER, win_bytes, mel_window_normalized(mel_raw_.as<float>(), dw, E, n_mel_, T2, w.melT.as<f16>(), stream_));
    else PROF(WMI_K_ENC_OTHER, win_bytes, mel_window(mel_raw_.as<float>(), mel_max_.as<unsigned>(), dw, E, n_mel_, T2, w.melT.as<f16>(), stream_));
    const int M1 = E * (T2 + 2), M2 = E * (T + 1), M = E * T;
    const int K1 = 3 * n_mel_;
    PROF(WMI_K_ENC_GEMM, 2.0 * E * T2 * d * K1, gemm_conv1(w.melT.as<f16>(), n_mel_, model->conv1_w, model->conv1_kpad, M1, d, w.h1.as<f16>(), model->conv1_b, T2, stream_));
    PROF(WMI_K_ENC_GEMM, 2.0 * M * d * 3 * d, gemm_conv2(w.h1.as<f16>(), model->conv2_w, M2, d, w.x.as<float>(), model->conv2_b, model->e_pe, T, stream_));
    const double ln_bytes = (double)M * d * 6;
    for (int l = 0; l < Le_; l++) {
        const EncLayer &L = model->enc[l];
        PROF(WMI_K_ENC_OTHER, ln_bytes, layer_norm(w.x.as<float>(), L.ln1_w, L.ln1_b, w.xn.as<f16>(), nullptr, M, d, nullptr, stream_));
        if (gemm_enc_qkv_is_split(M, d, T)) {
            PROF(WMI_K_ENC_GEMM, 2.0 * M * 2 * d * d, gemm_enc_qk(w.xn.as<f16>(), L.wqkv, M, d, H_, T, Tp, w.q.as<f16>(), w.k.as<f16>(), L.bqkv, stream_));
            PROF(WMI_K_ENC_GEMM, 2.0 * M * d * d, gemm_enc_vt(w.xn.as<f16>(), L.wqkv, M, d, H_, T, Tp, w.vt.as<f16>(), L.bqkv, stream_));
        } else {
            PROF(WMI_K_ENC_GEMM, 2.0 * M * 3 * d * d, gemm_enc_qkv(w.xn.as<f16>(), L.wqkv, M, d, H_, T, Tp, w.q.as<f16>(), w.k.as<f16>(), w.vt.as<f16>(), L.bqkv, stream_));
        }
        PROF(WMI_K_ENC_ATTN, 4.0 * E * (double)T * T * d, enc_attention(w.q.as<f16>(), w.k.as<f16>(), w.vt.as<f16>(), w.att.as<f16>(), E, H_, T, Tp, stream_));
        PROF(WMI_K_ENC_GEMM, 2.0 * M * d * d, gemm_residual(w.att.as<f16>(), d, L.wo, d, M, d, d, w.x.as<float>(), d, L.bo, stream_));
        PROF(WMI_K_ENC_OTHER, ln_bytes, layer_norm(w.x.as<float>(), L.ln2_w, L.ln2_b, w.xn.as<f16>(), nullptr, M, d, nullptr, stream_));
        PROF(WMI_K_ENC_GEMM, 2.0 * M * 4 * d * d, gemm_f16(w.xn.as<f16>(), d, L.fc1, d, M, 4 * d, d, w.hid.as<f16>(), 4 * d, L.b1, 1, 1.0f, stream_));
        PROF(WMI_K_ENC_GEMM, 2.0 * M * 4 * d * d, gemm_residual(w.hid.as<f16>(), 4 * d, L.fc2, 4 * d, M, d, 4 * d, w.x.as<float>(), d, L.b2, stream_));
    }
    float *e32 = nullptr;
    if (want_f32) { if (!enc32_.ensure((size_t)round_up(M, 256) * d * sizeof(float))) return false; e32 = enc32_.as<float>(); }
    PROF(WMI_K_ENC_OTHER, ln_bytes, layer_norm(w.x.as<float>(), model->ln_post_w, model->ln_post_b, w.enc16.as<f16>(), e32, M, d, nullptr, stream_));
    PROF(WMI_K_ENC_GEMM, 2.0 * M * (2.0 * Ld_ * d) * d,
         gemm_cross_kv(w.enc16.as<f16>(), model->xkv_w, M, d, H_, T, Tp, Ld_, ckv_.as<f16>(), model->xkv_b, ds, stream_));
    if (want_vt_ && ckvt_.p) PROF(WMI_K_ENC_OTHER, (double)E * Ld_ * T * d * 4, cross_v_transpose(ckv_.as<f16>(), ds, E, ckvt_.as<f16>(), H_, T, Tp, Ld_, stream_));
    prof_stream_ = this->stream_;
    enc_T_ = T;
    n_encode += E;
    return true;
}

int Engine::encode(int offset, int audio_ctx) {
    if (mel_utt_.empty()) return -1;
    (void)hipSetDevice(model->device);
    const int T = audio_ctx > 0 ? audio_ctx : T_full_;
    if (T > T_full_) return -1;
    const double t0 = now_ms();
    WinDesc w{mel_utt_[0].mel_off, mel_utt_[0].n_frames, utt_n_len_[0], offset, 0};
    if (!encode_chunk({w}, {0}, T, true, stream_, 0)) return -1;
    if (hipStreamSynchronize(stream_) != hipSuccess) return -1;
    t_encode_ms += now_ms() - t0;
    return 0;
}

int Engine::get_encoder_output(float *out) {
    if (!enc32_.p || enc_T_ <= 0) return -1;
    (void)hipSetDevice(model->device);
    return hipMemcpy(out, enc32_.p, (size_t)enc_T_ * d_ * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}

int Engine::get_cross_kv(int layer, float *k_out, float *v_out) {
    if (layer < 0 || layer >= Ld_ || enc_T_ <= 0) return -1;
    (void)hipSetDevice(model->device);
    std::vector<f16> tmp((size_t)H_ * Tp_full_ * 64);
    for (int kv = 0; kv < 2; kv++) {
        const f16 *src = ckv_.as<f16>() + ((size_t)(0 * Ld_ + layer) * 2 + kv) * H_ * Tp_full_ * 64;
        if (hipMemcpy(tmp.data(), src, tmp.size() * sizeof(f16), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        float *o = kv ? v_out : k_out;
        for (int h = 0; h < H_; h++) for (int t = 0; t < enc_T_; t++) for (int e = 0; e < 64; e++)
            o[(size_t)t * d_ + h * 64 + e] = (float)tmp[((size_t)h * Tp_full_ + t) * 64 + e];
    }
    return 0;
}

// ------------------------------------------------------------------------------------------ KV pages
int Engine::page_alloc() {
    if (free_pages_.empty()) return -1;
    const int p = free_pages_.back(); free_pages_.pop_back();
    page_ref_[p] = 1;
    return p;
}
void Engine::page_release(std::vector<int> &pages) {
    for (int p : pages) if (p > 0 && --page_ref_[p] == 0) free_pages_.push_back(p);
    pages.clear();
}
// Finds (allocating or un-sharing as needed) the cache slot for position `pos` of sequence d.
bool Engine::append_kv_slot(Decoder &d, int pos, std::vector<int2> &copies, int &kv_index) {
    const int pi = pos / kPageTokens, off = pos % kPageTokens;
    if (pi >= kMaxPagesPerSeq) return false;
    while ((int)d.pages.size() <= pi) { const int p = page_alloc(); if (p < 0) { log_msg(2, "engine: KV page pool exhausted\n"); return false; } d.pages.push_back(p); }
    int p = d.pages[pi];
    if (page_ref_[p] > 1) {            // shared with another beam: copy before the first private write
        const int np = page_alloc();
        if (np < 0) { log_msg(2, "engine: KV page pool exhausted\n"); return false; }
        if (off > 0) copies.push_back(int2{p, np});
        page_ref_[p]--;
        d.pages[pi] = np; p = np;
    }
    kv_index = p * kPageTokens + off;
    if (pos + 1 > d.n_kv) d.n_kv = pos + 1;
    return true;
}

// ------------------------------------------------------------------------------------------ decoder pass
bool Engine::decoder_pass(std::vector<DecRow> &rows, const std::vector<SampleRow> &srows, const std::vector<const Decoder *> &seqs,
                          const std::vector<int2> &page_copies, int T, bool want_raw_logits_host, float *logits_all_out, const LogitsFilter *filter) {
    const int R = (int)rows.size(), L = (int)srows.size(), S = (int)seqs.size(), C = (int)page_copies.size(), d = d_;
    if (R <= 0) return true;
    const double t_pass0 = now_ms();
    if (L > lrows_cap_ || S > lrows_cap_) { log_msg(2, "engine: decoder pass exceeds the state's capacity (%d logit rows)\n", L); return false; }
    if (!ensure_rows(std::max(R, C))) return false;
    // one packed metadata upload
    char *hb = meta_host_.as<char>();
    // runs of consecutive rows of one utterance (beams, prompt tokens), cut at 8: they share one sweep over the cross K/V
    const bool mq = want_vt_ && ckvt_.p;                 // MFMA kernel over the transposed V: groups of up to 16 rows
    const int group_cap = mq ? 16 : 8;
    xgroups_.clear();
    int max_group = 1;
    for (int r = 0; r < R;) {
        int e = r + 1;
        while (e < R && e - r < group_cap && rows[e].utt == rows[r].utt) e++;
        xgroups_.push_back(int2{r, e - r});
        max_group = std::max(max_group, e - r);
        r = e;
    }
    const int NG = (int)xgroups_.size();
    size_t o_rows = 0, o_srow = o_rows + sizeof(DecRow) * R, o_bt = o_srow + sizeof(SampleRow) * L, o_gather = o_bt + sizeof(int) * kMaxPagesPerSeq * S,
           o_copy = round_up_sz(o_gather + sizeof(int) * L, 8), o_grp = o_copy + sizeof(int2) * C, total = o_grp + sizeof(int2) * NG;
    if (total > meta_host_.cap) return false;
    memcpy(hb + o_rows, rows.data(), sizeof(DecRow) * R);
    if (L) memcpy(hb + o_srow, srows.data(), sizeof(SampleRow) * L);
    int *bt = (int *)(hb + o_bt);
    for (int s = 0; s < S; s++) for (int i = 0; i < kMaxPagesPerSeq; i++) bt[s * kMaxPagesPerSeq + i] = i < (int)seqs[s]->pages.size() ? seqs[s]->pages[i] : 0;
    int *gather = (int *)(hb + o_gather);
    for (int r = 0; r < R; r++) if (rows[r].logit_row >= 0) gather[rows[r].logit_row] = r;
    if (C) memcpy(hb + o_copy, page_copies.data(), sizeof(int2) * C);
    if (max_group > 1) memcpy(hb + o_grp, xgroups_.data(), sizeof(int2) * NG);
    bool any_probs = false;
    for (const SampleRow &s : srows) any_probs |= s.want_probs != 0;
    if (L > 0 && any_probs && (!probs_.ensure((size_t)lrows_cap_ * V_ * sizeof(float), false) || !probs_host_.ensure((size_t)lrows_cap_ * V_ * sizeof(float)))) return false;
    bool any_extra = any_probs;
    for (const SampleRow &s : srows) any_extra |= s.topk > 0;
    // Everything the pass puts on its stream, from the metadata upload to the download of the selections. What varies from one
    // pass to the next of the same shape -- tokens, positions, page tables, sampling state -- travels in the metadata buffer, so
    // the sequence of commands and their arguments is a function of the shape alone: see the graph replay below.
    auto enqueue = [&]() -> bool {
        WMI_HIP_OK(hipMemcpyAsync(meta_dev_.p, hb, total, hipMemcpyHostToDevice, stream_));
        char *db = meta_dev_.as<char>();
        const DecRow *d_rows = (const DecRow *)(db + o_rows);
        const SampleRow *d_srow = (const SampleRow *)(db + o_srow);
        const int *d_bt = (const int *)(db + o_bt), *d_gather = (const int *)(db + o_gather);
        const int2 *d_copy = (const int2 *)(db + o_copy);
        const int2 *d_grp = max_group > 1 ? (const int2 *)(db + o_grp) : nullptr;
        f16 *kp = kpool_.as<f16>(), *vp = vpool_.as<f16>();
        if (C) PROF(WMI_K_DEC_OTHER, 0, copy_pages(kp, vp, layer_stride_, Ld_, d_copy, C, H_, stream_));
        const double w1 = 2.0 * d * d;           // bytes of one d x d fp16 weight
        prof_stream_ = stream_;
        // The decoder layers over rows [r0, r0 + Rn) of the pass (cross-attention row groups [g0, g0 + NGn)) on stream `st`, partial sums
        // in `ws`. Three GEMM paths by row count, all with the same bits: at most 16 rows -- one launch per projection (LayerNorm in its
        // prologue, bias / residual / GELU in its epilogue), 8 launches per layer; up to split_max_rows() -- split-K with consumers that
        // also apply the epilogue and, for the residual GEMMs, the LayerNorm that follows; more -- fused-epilogue tiles + LayerNorm.
        // Leaves the residual stream of its rows in dx_; the final LayerNorm of the logit rows follows on the pass's stream.
        auto run_layers = [&](int r0, int Rn, int g0, int NGn, int max_grp, hipStream_t st, float *ws, size_t ws_cap) -> bool {
            float *x = dx_.as<float>() + (size_t)r0 * d;
            f16 *xn = dxn_.as<f16>() + (size_t)r0 * d, *q = dq_.as<f16>() + (size_t)r0 * d, *att = datt_.as<f16>() + (size_t)r0 * d, *hid = dhid_.as<f16>() + (size_t)r0 * 4 * d;
            const DecRow *rw = d_rows + r0;
            const double ln_bytes = (double)Rn * d * 6;
            PROF(WMI_K_DEC_OTHER, (double)Rn * d * 6, dec_embed(model->d_te, model->d_pe, rw, Rn, d, x, st));
            const bool may_split = (size_t)Rn <= split_max_rows();
            const int s_qkv = may_split ? gemm_pick_splits(3 * d, d) : 1, s_dd = may_split ? gemm_pick_splits(d, d) : 1,
                      s_fc1 = may_split ? gemm_pick_splits(4 * d, d) : 1, s_fc2 = may_split ? gemm_pick_splits(d, 4 * d) : 1;
            const bool fused_ln = s_dd > 1 && s_fc2 > 1;      // x += ... and the next LayerNorm in one kernel
            auto residual_ln = [&](const f16 *a, int K, const f16 *Wt, const float *bias, int splits, const float *lw, const float *lb) -> bool {
                PROF(WMI_K_DEC_GEMM, 2.0 * d * K, gemm_splitk(a, K, Wt, K, Rn, d, K, splits, ws, ws_cap, st));
                PROF(WMI_K_DEC_OTHER, ln_bytes, splitk_residual_ln(x, ws, splits, bias, lw, lb, xn, Rn, d, nullptr, Rn, st));
                return true;
            };
            auto cross = [&](int l) -> bool {
                if (mq && max_grp > 1)      // the group lists hold pass-wide row numbers: buffers from row 0
                    PROF(WMI_K_DEC_CROSS_ATTN, (double)NGn * 2.0 * T * d * 2, dec_cross_attention_mq(dq_.as<f16>(), ckv_.as<f16>(), ckvt_.as<f16>(), d_rows, d_grp + g0, NGn, d, H_, T, Tp_full_, Ld_, l, datt_.as<f16>(), st));
                else if (max_grp > 1)
                    PROF(WMI_K_DEC_CROSS_ATTN, (double)NGn * 2.0 * T * d * 2, dec_cross_attention(dq_.as<f16>(), ckv_.as<f16>(), d_rows, d_grp + g0, NGn, max_grp, Rn, d, H_, T, Tp_full_, Ld_, l, datt_.as<f16>(), st));
                else
                    PROF(WMI_K_DEC_CROSS_ATTN, (double)NGn * 2.0 * T * d * 2, dec_cross_attention(q, ckv_.as<f16>(), rw, nullptr, NGn, 1, Rn, d, H_, T, Tp_full_, Ld_, l, att, st));
                return true;
            };
            const bool few_rows = few_rows_path() && gemm_rows_ok(Rn, d, d, true) && gemm_rows_ok(Rn, d, 4 * d, false);
            if (fused_ln && !few_rows) PROF(WMI_K_DEC_OTHER, ln_bytes, layer_norm(x, model->dec[0].ln1_w, model->dec[0].ln1_b, xn, nullptr, Rn, d, nullptr, st));
            for (int l = 0; l < Ld_; l++) {
                const DecLayer &D = model->dec[l];
                f16 *kl = kp + l * layer_stride_, *vl = vp + l * layer_stride_;
                if (few_rows) {
                    auto qw = [&](const QuantW &qq) -> const QuantW * { return quant_decode && qq.ttype ? &qq : nullptr; };
                    PROF(WMI_K_DEC_GEMM, 3 * w1, gemm_rows_ln_dec_qkv(x, D.ln1_w, D.ln1_b, D.wqkv, Rn, d, H_, q, kl, vl, D.bqkv, rw, st, qw(D.q_wqkv)));
                    PROF(WMI_K_DEC_OTHER, 0, dec_self_attention(q, kl, vl, d_bt, rw, Rn, d, H_, att, st));
                    PROF(WMI_K_DEC_GEMM, w1, gemm_rows_residual(att, d, D.wo, Rn, d, d, x, d, D.bo, st, qw(D.q_wo)));
                    PROF(WMI_K_DEC_GEMM, w1, gemm_rows_ln_f16(x, D.lnx_w, D.lnx_b, D.xq, Rn, d, d, q, d, D.bxq, 0, 0.125f, st, qw(D.q_xq)));
                    if (!cross(l)) return false;
                    PROF(WMI_K_DEC_GEMM, w1, gemm_rows_residual(att, d, D.xo, Rn, d, d, x, d, D.bxo, st, qw(D.q_xo)));
                    PROF(WMI_K_DEC_GEMM, 4 * w1, gemm_rows_ln_f16(x, D.ln2_w, D.ln2_b, D.fc1, Rn, 4 * d, d, hid, 4 * d, D.b1, 1, 1.0f, st, qw(D.q_fc1)));
                    PROF(WMI_K_DEC_GEMM, 4 * w1, gemm_rows_residual(hid, 4 * d, D.fc2, Rn, d, 4 * d, x, d, D.b2, st, qw(D.q_fc2)));
                    continue;
                }
                if (!fused_ln) PROF(WMI_K_DEC_OTHER, ln_bytes, layer_norm(x, D.ln1_w, D.ln1_b, xn, nullptr, Rn, d, nullptr, st));
                if (s_qkv > 1) {
                    PROF(WMI_K_DEC_GEMM, 3 * w1, gemm_splitk(xn, d, D.wqkv, d, Rn, 3 * d, d, s_qkv, ws, ws_cap, st));
                    PROF(WMI_K_DEC_OTHER, (double)Rn * 3 * d * 2, splitk_dec_qkv(ws, s_qkv, D.bqkv, rw, Rn, d, H_, q, kl, vl, st));
                } else {
                    PROF(WMI_K_DEC_GEMM, 3 * w1, gemm_dec_qkv(xn, D.wqkv, Rn, d, H_, q, kl, vl, D.bqkv, rw, st));
                }
                PROF(WMI_K_DEC_OTHER, 0, dec_self_attention(q, kl, vl, d_bt, rw, Rn, d, H_, att, st));
                if (fused_ln) { if (!residual_ln(att, d, D.wo, D.bo, s_dd, D.lnx_w, D.lnx_b)) return false; }
                else {
                    PROF(WMI_K_DEC_GEMM, w1, gemm_residual(att, d, D.wo, d, Rn, d, d, x, d, D.bo, st, true));
                    PROF(WMI_K_DEC_OTHER, ln_bytes, layer_norm(x, D.lnx_w, D.lnx_b, xn, nullptr, Rn, d, nullptr, st));
                }
                if (s_dd > 1) {
                    PROF(WMI_K_DEC_GEMM, w1, gemm_splitk(xn, d, D.xq, d, Rn, d, d, s_dd, ws, ws_cap, st));
                    PROF(WMI_K_DEC_OTHER, (double)Rn * d * 2, splitk_f16(ws, s_dd, D.bxq, 0, 0.125f, q, Rn, d, st));
                } else {
                    PROF(WMI_K_DEC_GEMM, w1, gemm_f16(xn, d, D.xq, d, Rn, d, d, q, d, D.bxq, 0, 0.125f, st, true));
                }
                if (!cross(l)) return false;
                if (fused_ln) { if (!residual_ln(att, d, D.xo, D.bxo, s_dd, D.ln2_w, D.ln2_b)) return false; }
                else {
                    PROF(WMI_K_DEC_GEMM, w1, gemm_residual(att, d, D.xo, d, Rn, d, d, x, d, D.bxo, st, true));
                    PROF(WMI_K_DEC_OTHER, ln_bytes, layer_norm(x, D.ln2_w, D.ln2_b, xn, nullptr, Rn, d, nullptr, st));
                }
                if (s_fc1 > 1) {
                    PROF(WMI_K_DEC_GEMM, 4 * w1, gemm_splitk(xn, d, D.fc1, d, Rn, 4 * d, d, s_fc1, ws, ws_cap, st));
                    PROF(WMI_K_DEC_OTHER, (double)Rn * 4 * d * 2, splitk_f16(ws, s_fc1, D.b1, 1, 1.0f, hid, Rn, 4 * d, st));
                } else {
                    PROF(WMI_K_DEC_GEMM, 4 * w1, gemm_f16(xn, d, D.fc1, d, Rn, 4 * d, d, hid, 4 * d, D.b1, 1, 1.0f, st, true));
                }
                if (fused_ln) {
                    // (after the last layer the LayerNorm this writes is not used: the logit rows get theirs below)
                    const float *lw = l + 1 < Ld_ ? model->dec[l + 1].ln1_w : model->d_ln_w, *lb = l + 1 < Ld_ ? model->dec[l + 1].ln1_b : model->d_ln_b;
                    if (!residual_ln(hid, 4 * d, D.fc2, D.b2, s_fc2, lw, lb)) return false;
                } else {
                    PROF(WMI_K_DEC_GEMM, 4 * w1, gemm_residual(hid, 4 * d, D.fc2, 4 * d, Rn, d, 4 * d, x, d, D.b2, st, true));
                }
            }
            return true;
        };
        // Row lanes: the kernels of a pass of a few hundred rows are chains of dependent 5-30 us launches that leave most of the chip
        // idle (100-400 workgroups, each bound by its own latency chain). Two halves of the rows on two streams run two such chains
        // side by side; since no path's arithmetic depends on the rows that share its launches, the halves give the bits the whole
        // pass would. The pass's stream forks after the metadata upload and joins before the logits.
        const bool prof_dec = profiling && (profile_mask & ((1u << WMI_K_DEC_GEMM) | (1u << WMI_K_DEC_OTHER) | (1u << WMI_K_DEC_CROSS_ATTN)));
        int cut_row = 0, cut_grp = 0;
        if (dec_lanes_ > 1 && !prof_dec && R >= 2 * kDecLaneMinRows && lane_stream_ && lane_ws_.p) {
            // cut at the cross-attention group boundary nearest to the middle (rows of one utterance stay in one lane)
            int best = -1;
            for (int gi = 1; gi < NG; gi++) {
                const int r = xgroups_[gi].x;
                if (xgroups_[gi - 1].x + xgroups_[gi - 1].y == r && rows[r].utt != rows[r - 1].utt && (best < 0 || abs(r - R / 2) < abs(xgroups_[best].x - R / 2))) best = gi;
            }
            if (best > 0 && xgroups_[best].x >= kDecLaneMinRows && R - xgroups_[best].x >= kDecLaneMinRows) { cut_grp = best; cut_row = xgroups_[best].x; }
        }
        if (cut_row > 0) {
            int mg0 = 1, mg1 = 1;
            for (int gi = 0; gi < NG; gi++) (gi < cut_grp ? mg0 : mg1) = std::max(gi < cut_grp ? mg0 : mg1, xgroups_[gi].y);
            WMI_HIP_OK(hipEventRecord(lane_fork_, stream_));
            WMI_HIP_OK(hipStreamWaitEvent(lane_stream_, lane_fork_, 0));
            const bool ok1 = run_layers(cut_row, R - cut_row, cut_grp, NG - cut_grp, mg1, lane_stream_, lane_ws_.as<float>(), lane_ws_.cap);
            const bool ok0 = run_layers(0, cut_row, 0, cut_grp, mg0, stream_, splitk_ws_.as<float>(), splitk_ws_.cap);
            // join even when a launch failed: nothing may still run on the lane when this pass returns
            WMI_HIP_OK(hipEventRecord(lane_join_, lane_stream_));
            WMI_HIP_OK(hipStreamWaitEvent(stream_, lane_join_, 0));
            if (!ok0 || !ok1) { (void)hipStreamSynchronize(stream_); return false; }
        } else {
            if (!run_layers(0, R, 0, NG, max_group, stream_, splitk_ws_.as<float>(), splitk_ws_.cap)) return false;
        }
        if (L > 0) {
            PROF(WMI_K_DEC_OTHER, (double)L * d * 6, layer_norm(dx_.as<float>(), model->d_ln_w, model->d_ln_b, dxl_.as<f16>(), nullptr, L, d, d_gather, stream_));
            PROF(WMI_K_DEC_GEMM, 2.0 * Vp_ * d + 4.0 * L * Vp_, gemm_f32(dxl_.as<f16>(), d, model->d_te, d, L, Vp_, d, logits_.as<float>(), Vp_, nullptr, stream_));
            if (logits_all_out) WMI_HIP_OK(hipMemcpy2DAsync(logits_all_out, sizeof(float) * V_, logits_.p, sizeof(float) * Vp_, sizeof(float) * V_, L, hipMemcpyDeviceToHost, stream_));
            if (want_raw_logits_host) WMI_HIP_OK(hipMemcpyAsync(logits_host_.data(), logits_.as<float>() + (size_t)(L - 1) * Vp_, sizeof(float) * V_, hipMemcpyDeviceToHost, stream_));
            SampleOut *d_out = out_dev_.as<SampleOut>();
            TopkOut *d_topk = (TopkOut *)(out_dev_.as<char>() + sizeof(SampleOut) * lrows_cap_);
            if (filter) {
                // whisper_full hands the logits to the caller's filter after the temperature and the special-token suppression and
                // applies the remaining rules to what the filter leaves: first half of the rules, rows to the host and back, second half
                std::vector<SampleRow> half(srows);
                for (SampleRow &r : half) r.flags |= 512;
                filter_rows_.resize((size_t)L * V_);
                WMI_HIP_OK(hipMemcpyAsync(meta_dev_.as<char>() + o_srow, half.data(), sizeof(SampleRow) * L, hipMemcpyHostToDevice, stream_));
                if (!sample_rows(logits_.as<float>(), Vp_, L, d_srow, model->vocab_ids(), model->nst_bitmap, d_out, d_topk, 8, nullptr, false, sample_ws_.p, stream_)) return false;
                WMI_HIP_OK(hipMemcpy2DAsync(filter_rows_.data(), sizeof(float) * V_, logits_.p, sizeof(float) * Vp_, sizeof(float) * V_, L, hipMemcpyDeviceToHost, stream_));
                WMI_HIP_OK(hipStreamSynchronize(stream_));
                for (int r = 0; r < L; r++) (*filter)(r, filter_rows_.data() + (size_t)r * V_);
                for (SampleRow &r : half) r.flags = (r.flags & ~512) | 1024;
                WMI_HIP_OK(hipMemcpy2DAsync(logits_.p, sizeof(float) * Vp_, filter_rows_.data(), sizeof(float) * V_, sizeof(float) * V_, L, hipMemcpyHostToDevice, stream_));
                WMI_HIP_OK(hipMemcpyAsync(meta_dev_.as<char>() + o_srow, half.data(), sizeof(SampleRow) * L, hipMemcpyHostToDevice, stream_));
                WMI_HIP_OK(hipStreamSynchronize(stream_));        // `half` leaves scope below; the copies above read pageable host memory
            }
            PROF(WMI_K_DEC_OTHER, 8.0 * L * V_, sample_rows(logits_.as<float>(), Vp_, L, d_srow, model->vocab_ids(), model->nst_bitmap, d_out, d_topk, 8,
                                                           any_probs ? probs_.as<float>() : nullptr, any_extra, sample_ws_.p, stream_));
            WMI_HIP_OK(hipMemcpyAsync(out_host_.p, out_dev_.p, sizeof(SampleOut) * lrows_cap_ + sizeof(TopkOut) * 8 * (size_t)L, hipMemcpyDeviceToHost, stream_));
            if (any_probs) WMI_HIP_OK(hipMemcpyAsync(probs_host_.p, probs_.p, sizeof(float) * (size_t)L * V_, hipMemcpyDeviceToHost, stream_));
        }
        return true;
    };
    // Replay: a pass whose shape has been seen before runs as one HIP graph launch instead of the ~10 launches per layer it is
    // made of (a one-clip step pass of large-v3-turbo: 42 kernels and two copies). The graph is captured from the very code above
    // the second time a shape comes by -- the first time warms every kernel up, since nothing may be loaded or configured under
    // capture -- and is keyed by everything the commands' arguments depend on, the buffers' addresses included.
    bool launched = false;
    const bool prof_any = profiling && profile_mask != 0;
    if (use_graphs_ && !prof_any && !filter && !logits_all_out && !want_raw_logits_host && !any_probs && dec_lanes_ <= 1) {
        uintptr_t ptrs = 0;
        for (const void *q : {dx_.p, dxn_.p, dq_.p, datt_.p, dhid_.p, dxl_.p, logits_.p, meta_dev_.p, (void *)meta_host_.p, out_dev_.p, (void *)out_host_.p, ckv_.p, ckvt_.p,
                              kpool_.p, vpool_.p, splitk_ws_.p, sample_ws_.p})
            ptrs = ptrs * 1000003u ^ (uintptr_t)q;
        const PassKey key{{R, L, S, C, NG, max_group, T, (int)any_extra | (mq ? 2 : 0) | (quant_decode ? 4 : 0), (int)(ptrs & 0x7fffffff), (int)((ptrs >> 31) & 0x7fffffff)}};
        if (graphs_.size() > 64) clear_graphs();
        PassGraph &g = graphs_[key];
        if (g.exec) {
            launched = hipGraphLaunch(g.exec, stream_) == hipSuccess;
            if (!launched) { (void)hipGetLastError(); (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; g.bad = true; }
        } else if (!g.bad && ++g.seen >= 2) {
            if (hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                const bool enq = enqueue();
                hipGraph_t graph = nullptr;
                const hipError_t e = hipStreamEndCapture(stream_, &graph);
                if (enq && e == hipSuccess && graph && hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0) == hipSuccess && g.exec)
                    launched = hipGraphLaunch(g.exec, stream_) == hipSuccess;
                if (graph) (void)hipGraphDestroy(graph);
                if (!launched) {
                    (void)hipGetLastError();
                    if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
                    g.bad = true;
                    log_msg(1, "engine: a decoder pass of %d rows could not be captured as a graph; it keeps running as plain launches\n", R);
                }
            } else { (void)hipGetLastError(); g.bad = true; }
        }
    }
    if (!launched && !enqueue()) return false;
    const double t_launched = now_ms();
    WMI_HIP_OK(hipStreamSynchronize(stream_));
    t_dec_launch_ms += t_launched - t_pass0; t_dec_wait_ms += now_ms() - t_launched; n_passes++;
    n_decode += R;
    return true;
}

// whisper_decode(): positions n_past.. of the low-level sequence; raw logits of the last token land in logits_host_
int Engine::decode(const int *tokens, int n_tokens, int n_past, bool logits_all, float *logits_all_out) {
    if (!tokens || n_tokens <= 0 || n_past < 0 || n_past + n_tokens > model->hp.n_text_ctx || enc_T_ <= 0) return -1;
    if (logits_all && n_tokens > lrows_cap_) return -1;
    (void)hipSetDevice(model->device);
    const double t0 = now_ms();
    std::vector<DecRow> rows(n_tokens);
    std::vector<SampleRow> sr;
    std::vector<int2> copies;
    for (int i = 0; i < n_tokens; i++) {
        if (tokens[i] < 0 || tokens[i] >= V_) return -1;
        DecRow &r = rows[i];
        r.token = tokens[i]; r.pos = n_past + i; r.seq = 0; r.utt = 0; r.logit_row = -1;
        if (!append_kv_slot(ll_dec_, r.pos, copies, r.kv_index)) return -1;
        if (logits_all || i == n_tokens - 1) {
            r.logit_row = (int)sr.size();
            SampleRow s{}; s.flags = 256; s.max_initial = -1;
            sr.push_back(s);
        }
    }
    if (!decoder_pass(rows, sr, {&ll_dec_}, copies, enc_T_, true, logits_all ? logits_all_out : nullptr)) return -1;
    t_decode_ms += now_ms() - t0;
    return 0;
}

int Engine::lang_auto_detect(int offset_ms, float *probs) {
    const int seek = offset_ms / 10;
    if (seek < 0) return -1;
    if (mel_utt_.empty() || seek >= utt_n_len_org_[0]) return -2;
    if (encode(seek, 0) != 0) return -6;
    page_release(ll_dec_.pages); ll_dec_.n_kv = 0;
    const int sot = model->vocab.sot;
    if (decode(&sot, 1, 0, false, nullptr) != 0) return -7;
    int best = -1; float bl = -INFINITY; double mx = -1e30;
    for (int i = 0; i < kNumLangs; i++) { const int t = model->token_lang(i); if (t >= V_) continue; const float l = logits_host_[t]; if (l > bl) { bl = l; best = i; } mx = std::max(mx, (double)l); }
    if (probs) {
        double sum = 0;
        for (int i = 0; i < kNumLangs; i++) { const int t = model->token_lang(i); if (t < V_) sum += exp(logits_host_[t] - mx); }
        for (int i = 0; i < kNumLangs; i++) { const int t = model->token_lang(i); probs[i] = t < V_ ? (float)(exp(logits_host_[t] - mx) / sum) : 0.0f; }
    }
    return best;
}

// ------------------------------------------------------------------------------------------ whisper_full
namespace {

// the mapping std::discrete_distribution applies in libstdc++ (53-bit canonical double from two 32-bit
// draws, first index whose normalised cumulative probability reaches it), restated so the draw does not
// depend on the host's standard library
int sample_discrete(const float *probs, int n, std::mt19937 &rng) {
    double sum = 0;
    for (int i = 0; i < n; i++) sum += probs[i];
    const double lo = (double)rng(), hi = (double)rng();
    double u = (lo + hi * 4294967296.0) / 18446744073709551616.0;
    if (u >= 1.0) u = nextafter(1.0, 0.0);
    double c = 0;
    for (int i = 0; i < n - 1; i++) { c += probs[i] / sum; if (c >= u) return i; }
    return n - 1;
}

void sequence_score(const whisper_full_params &p, Sequence &q) {
    if (q.result_len == 0) return;
    double r = 0;
    for (int i = 0; i < q.result_len; i++) r += q.tokens[i].plog;
    q.sum_logprobs = r; q.avg_logprobs = r / q.result_len;
    double pen = q.result_len;
    if (p.length_penalty > 0.0f) pen = pow((5.0 + pen) / 6.0, p.length_penalty);
    q.score = r / pen;
    std::map<int, int> cnt; int n = 0;
    for (int i = std::max(0, q.result_len - 32); i < q.result_len; i++) { cnt[q.tokens[i].id]++; n++; }
    double ent = 0;
    for (auto &kv : cnt) { const double pr = kv.second / (double)n; ent -= pr * log(pr); }
    q.entropy = ent;
}

whisper_token_data make_token(int id, int tid, float p, float plog, float pt, float ptsum, int beg) {
    whisper_token_data t{};
    t.id = id; t.tid = tid; t.p = p; t.plog = plog; t.pt = pt; t.ptsum = ptsum; t.t0 = t.t1 = t.t_dtw = -1; t.vlen = 0;
    if (t.id >= beg) { t.tid = t.id; t.pt = t.p; }
    return t;
}

}  // namespace

struct Engine::Run {
    using LogitsFilter = Engine::LogitsFilter;
    Engine &E;
    whisper_context *ctx; whisper_state *st;
    whisper_full_params P;
    std::vector<Utt> utts;
    std::vector<int> initial_prompt_tokens;
    int n_decoders = 1, n_max = 0, single = 0;
    const Vocab &V;

    // window memo: host PCM of the running group (null when the caller's PCM is on the device), per-utterance sample counts,
    // the raw log-mel maxima (they enter the normalisation of every window), and a hash of the decoding parameters
    const float *const *memo_pcm = nullptr;
    const int *memo_n = nullptr;
    std::vector<unsigned> memo_mmax;
    uint64_t memo_params = 0;

    Run(Engine &e, whisper_context *c, whisper_state *s, const whisper_full_params &p) : E(e), ctx(c), st(s), P(p), V(e.model->vocab) {}

    static uint64_t mix(uint64_t h, uint64_t v) { h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2); h *= 0xff51afd7ed558ccdull; return h ^ (h >> 32); }
    static uint64_t hash_bytes(const void *p, size_t n, uint64_t h) {
        const unsigned char *b = (const unsigned char *)p;
        size_t i = 0;
        for (; i + 8 <= n; i += 8) { uint64_t v; memcpy(&v, b + i, 8); h = (h ^ v) * 0x9fb21c651e98df25ull; h ^= h >> 29; }
        uint64_t t = 0; if (i < n) memcpy(&t, b + i, n - i);
        return mix(h, t ^ (uint64_t)n);
    }
    uint64_t params_hash() const {
        uint64_t h = 0x6d656d6full;
        const int iv[] = {P.strategy, P.n_max_text_ctx, P.offset_ms, P.duration_ms, P.translate, P.no_context, P.no_timestamps, P.single_segment, P.print_special,
                          P.max_tokens, P.audio_ctx, P.tdrz_enable, P.suppress_blank, P.suppress_nst, P.greedy.best_of, P.beam_search.beam_size,
                          E.force_tokens, (int)E.beam_sampled, n_decoders, P.token_timestamps, P.max_len, P.split_on_word};
        h = hash_bytes(iv, sizeof iv, h);
        const float fv[] = {P.temperature, P.max_initial_ts, P.length_penalty, P.temperature_inc, P.entropy_thold, P.logprob_thold, P.no_speech_thold, P.beam_search.patience};
        h = hash_bytes(fv, sizeof fv, h);
        if (!initial_prompt_tokens.empty()) h = hash_bytes(initial_prompt_tokens.data(), initial_prompt_tokens.size() * sizeof(int), h);
        return h;
    }
    // Key of the window that starts at u.seek, or 0 when the window may not be remembered. Everything its outcome depends on
    // goes in: the samples under its 3000 frames (with the STFT overlap), the utterance-wide log-mel maximum, the parameters,
    // the language, the text context and, through the chain value, every earlier window of this call (hence the state of the
    // decoders' random generators). Only interior windows qualify: near the end of the audio the decode loop looks at the
    // remaining length.
    uint64_t window_key(const Utt &u) const {
        if (E.window_cache_cap <= 0 || !memo_pcm || u.seek + 3100 >= u.seek_end || P.logits_filter_callback) return 0;
        const long n = memo_n[u.index];
        const long s0 = std::max(0L, (long)u.seek * kHop - 200), s1 = std::min(n, ((long)u.seek + 3000) * kHop + 200);
        if (s1 <= s0) return 0;
        uint64_t h = mix(mix(u.chain, memo_params), ((uint64_t)(uint32_t)u.seek << 32) | (uint32_t)u.lang_id);
        h = mix(h, ((uint64_t)memo_mmax[u.mel_idx] << 1) | (u.no_timestamps ? 1u : 0u));
        if (!u.prompt_past.empty()) h = hash_bytes(u.prompt_past.data(), u.prompt_past.size() * sizeof(int), h);
        h = hash_bytes(memo_pcm[u.index] + s0, (size_t)(s1 - s0) * sizeof(float), h);
        return h ? h : 1;
    }
    // the window at u.seek was decoded before from identical inputs: take its outcome
    bool reuse_window(Utt &u) {
        const uint64_t key = window_key(u);
        if (!key) return false;
        for (const WindowMemo &m : E.window_memos) {
            if (m.key != key || m.rng.size() != u.dec.size()) continue;
            UttResult &res = E.results[u.index];
            res.segs.insert(res.segs.end(), m.segs.begin(), m.segs.end());
            res.n_fallbacks += m.n_fallbacks;
            if (single && P.new_segment_callback && !m.segs.empty()) P.new_segment_callback(ctx, st, (int)m.segs.size(), P.new_segment_callback_user_data);
            u.prompt_past = m.prompt_past;
            for (size_t j = 0; j < u.dec.size(); j++) u.dec[j].rng = m.rng[j];
            u.chain = m.chain_after;
            u.seek += m.seek_delta;
            u.encoded_at_seek = false;
            u.cur_key = 0;
            u.phase = u.seek + 100 >= u.seek_end ? Phase::Done : Phase::Encode;
            E.window_cache_hits++;
            return true;
        }
        return false;
    }
    void remember_window(Utt &u, size_t n_before, int seek_delta) {
        uint64_t after = mix(u.chain, mix(u.cur_key ? u.cur_key : 0x77ull + (uint64_t)u.seek, (uint64_t)seek_delta));
        const std::vector<Segment> &out = E.results[u.index].segs;
        for (size_t i = n_before; i < out.size(); i++)
            for (const whisper_token_data &t : out[i].tokens) after = mix(after, (uint64_t)(uint32_t)t.id);
        if (u.cur_key && E.window_cache_cap > 0) {
            WindowMemo m;
            m.key = u.cur_key; m.chain_after = after; m.seek_delta = seek_delta; m.n_fallbacks = E.results[u.index].n_fallbacks - u.fb0;
            m.segs.assign(out.begin() + n_before, out.end());
            m.prompt_past = u.prompt_past;
            for (const Decoder &d : u.dec) m.rng.push_back(d.rng);
            E.window_memos.push_back(std::move(m));
            while ((int)E.window_memos.size() > E.window_cache_cap) E.window_memos.pop_front();
            E.window_cache_stores++;
        }
        u.chain = after;
        u.cur_key = 0;
    }

    void release_decoders(Utt &u) { for (Decoder &d : u.dec) { E.page_release(d.pages); d.n_kv = 0; } }

    // returns false when the utterance is finished (phase Done, rc set)
    void init_utt(Utt &u) {
        const Model &M = *E.model;
        u.dec.resize(n_decoders);
        // decoder j draws from its own generator (seed j): the best_of decoders start from the same prompt logits and
        // cache, so one shared seed would make every one of them draw the same tokens
        for (int j = 0; j < n_decoders; j++) u.dec[j].rng = std::mt19937((uint32_t)j);
        u.n_len = E.utt_n_len_[u.mel_idx]; u.n_len_org = E.utt_n_len_org_[u.mel_idx];
        u.seek_start = P.offset_ms / 10;
        u.seek_end = P.duration_ms == 0 ? u.n_len_org : u.seek_start + P.duration_ms / 10;
        u.temps.clear();
        if (P.temperature_inc > 0.0f) { for (float t = P.temperature; t < 1.0f + 1e-6f && u.temps.size() < 16; t += P.temperature_inc) u.temps.push_back(t); }
        else u.temps.push_back(P.temperature);
        u.prompt_past = initial_prompt_tokens;         // caller's prompt tokens, then (one-utterance calls) the previous call's text context
        u.no_timestamps = P.no_timestamps;
        const bool detect = P.language == nullptr || P.language[0] == 0 || !strcmp(P.language, "auto") || P.detect_language;
        u.lang_id = 0;
        if (!detect) {
            u.lang_id = lang_id_from_str(P.language);
            if (u.lang_id < 0 && M.vocab.multilingual()) { u.rc = -3; u.phase = Phase::Done; return; }
            if (u.lang_id < 0) u.lang_id = 0;
        }
        if (detect && M.vocab.multilingual()) {
            if (0 >= u.n_len_org) { u.rc = -3; u.phase = Phase::Done; return; }
            u.seek = 0; u.phase = Phase::LangEncode;
            return;
        }
        after_language(u);
    }

    void after_language(Utt &u) {
        const Model &M = *E.model;
        E.results[u.index].lang_id = u.lang_id;
        if (P.detect_language) { u.phase = Phase::Done; return; }
        if (u.seek_end < u.seek_start + 100) { u.phase = Phase::Done; return; }     // under one second: nothing is produced
        u.prompt_init.clear();
        u.prompt_init.push_back(V.sot);
        if (V.multilingual()) {
            u.prompt_init.push_back(M.token_lang(u.lang_id));
            u.prompt_init.push_back(P.translate ? V.translate : V.transcribe);
        }
        const bool is_distil = M.hp.n_text_layer == 2 && M.hp.n_vocab != 51866;
        if (is_distil) u.no_timestamps = true;
        if (u.no_timestamps) u.prompt_init.push_back(V.tok_not);
        const bool reuse = u.encoded_at_seek && u.seek == u.seek_start;
        u.seek = u.seek_start;
        u.phase = Phase::Encode;
        if (reuse && u.seek + 100 < u.seek_end) begin_window(u);                      // language detection already encoded this window
    }

    // after the encoder ran for window u.seek
    void begin_window(Utt &u) {
        u.encoded_at_seek = true;
        u.cur_key = window_key(u);
        u.fb0 = E.results[u.index].n_fallbacks;
        if (u.seek > u.seek_start && u.seek + 500 >= u.seek_end) u.prompt_past.clear();
        u.it = 0; u.best = 0;
        start_iteration(u);
    }

    void start_iteration(Utt &u) {
        u.t_cur = u.temps[u.it];
        int n_cur = 1;
        if (P.strategy == WHISPER_SAMPLING_GREEDY) { if (u.t_cur > 0.0f) n_cur = P.greedy.best_of; }
        else { n_cur = u.t_cur > 0.0f ? P.greedy.best_of : P.beam_search.beam_size; }
        u.n_cur = std::max(1, std::min(n_cur, n_decoders));
        release_decoders(u);
        for (int j = 0; j < u.n_cur; j++) {
            Decoder &d = u.dec[j];
            d.seq = Sequence{}; d.seq.sum_logprobs = -INFINITY; d.seq.avg_logprobs = -INFINITY; d.seq.score = -INFINITY;
            d.seek_delta = 100 * kChunkSec; d.failed = d.completed = d.has_ts = false;
        }
        u.prompt.clear();
        if (!u.prompt_past.empty() && u.t_cur < 0.5f && P.n_max_text_ctx > 0) {
            const int n_take = std::min(std::min(P.n_max_text_ctx, E.model->hp.n_text_ctx / 2), (int)u.prompt_past.size());
            u.prompt.push_back(V.prev);
            u.prompt.insert(u.prompt.end(), u.prompt_past.end() - n_take, u.prompt_past.end());
        }
        u.prompt.insert(u.prompt.end(), u.prompt_init.begin(), u.prompt_init.end());
        u.n_prompt = (int)u.prompt.size();
        u.i = 0;
        u.phase = Phase::Prompt;
    }

    SampleRow sample_state(const Utt &u, const Decoder &d) const {
        SampleRow s{};
        s.temperature = u.t_cur;
        const auto &tk = d.seq.tokens;
        const int n = (int)tk.size();
        int f = 0;
        if (n == 0) f |= 1;
        if (n > 0 && tk[n - 1].id >= V.beg) f |= 2;
        if (n < 2 || tk[n - 2].id >= V.beg) f |= 4;
        if (d.has_ts) f |= 8;
        if (P.suppress_blank) f |= 16;
        if (u.no_timestamps) f |= 32;
        if (P.suppress_nst) f |= 64;
        if (P.tdrz_enable) f |= 128;
        s.flags = f;
        s.ts_min = d.seek_delta / 2;
        s.max_initial = -1;
        if (P.max_initial_ts > 0.0f) { const float precision = (float)kChunkSec / E.model->hp.n_audio_ctx; s.max_initial = (int)roundf(P.max_initial_ts / precision); }
        s.n_sampled = n; s.force_n = E.force_tokens;
        const bool beam = P.strategy == WHISPER_SAMPLING_BEAM_SEARCH;
        s.topk = (beam && !E.beam_sampled) ? std::min(8, P.beam_search.beam_size) : 0;
        s.want_probs = ((!beam && u.t_cur > 0.0f) || (beam && E.beam_sampled)) ? 1 : 0;
        return s;
    }

    // one sampling / bookkeeping step of whisper_full's inner loop; logits results are in each decoder's `so`
    void process_step(Utt &u) {
        const int i = u.i, beg = V.beg;
        struct Cand { int src; Sequence seq; int seek_delta; bool has_ts; };
        std::vector<Cand> cands;
        for (int j = 0; j < u.n_cur; j++) {
            Decoder &d = u.dec[j];
            if (d.completed || d.failed) continue;
            if (P.strategy == WHISPER_SAMPLING_GREEDY) {
                whisper_token_data t;
                if (u.t_cur < 1e-6f) t = make_token(d.so.id, d.so.tid, d.so.p, d.so.plog, d.so.pt, d.so.ptsum, beg);
                else { const int id = sample_discrete(d.probs.data(), E.V_, d.rng); const float p = d.probs[id]; t = make_token(id, d.so.tid, p, logf(p), d.so.pt, d.so.ptsum, beg); }
                d.seq.tokens.push_back(t); d.seq.sum_logprobs_all += t.plog;
            } else {
                const int k = std::min(8, P.beam_search.beam_size);
                for (int c = 0; c < k; c++) {
                    whisper_token_data t;
                    if (E.beam_sampled) { const int id = sample_discrete(d.probs.data(), E.V_, d.rng); const float p = d.probs[id]; t = make_token(id, d.so.tid, p, logf(p), d.so.pt, d.so.ptsum, beg); }
                    else { if (d.topk[c].id < 0) continue; t = make_token(d.topk[c].id, d.so.tid, d.topk[c].p, d.topk[c].plog, d.so.pt, d.so.ptsum, beg); }
                    Cand cd{j, d.seq, d.seek_delta, d.has_ts};
                    cd.seq.tokens.push_back(t); cd.seq.sum_logprobs_all += t.plog;
                    cands.push_back(std::move(cd));
                }
            }
        }
        if (P.strategy == WHISPER_SAMPLING_BEAM_SEARCH) {
            std::stable_sort(cands.begin(), cands.end(), [](const Cand &a, const Cand &b) {
                if (a.seq.sum_logprobs_all != b.seq.sum_logprobs_all) return a.seq.sum_logprobs_all > b.seq.sum_logprobs_all;
                return a.src < b.src;
            });
            auto same = [](const Sequence &a, const Sequence &b) {
                if (a.tokens.size() != b.tokens.size()) return false;
                for (size_t q = 0; q < a.tokens.size(); q++) if (a.tokens[q].id != b.tokens[q].id) return false;
                return true;
            };
            size_t cur = 0;
            std::vector<std::vector<int>> new_pages(u.n_cur);
            std::vector<int> taken(u.n_cur, 0), new_nkv(u.n_cur, 0);
            for (int j = 0; j < u.n_cur && !cands.empty(); j++) {
                Decoder &d = u.dec[j];
                if (d.completed || d.failed) continue;
                if (cur >= cands.size()) cur = 0;
                Cand &c = cands[cur++];
                while (cands.size() > cur && same(cands[cur].seq, c.seq) && i > 0) cur++;
                new_pages[j] = u.dec[c.src].pages; new_nkv[j] = u.dec[c.src].n_kv;          // share the parent's pages
                for (int p : new_pages[j]) E.page_ref_[p]++;
                taken[j] = 1;
                d.seek_delta = c.seek_delta; d.has_ts = c.has_ts; d.seq = c.seq;
            }
            for (int j = 0; j < u.n_cur; j++) if (taken[j]) { E.page_release(u.dec[j].pages); u.dec[j].pages = new_pages[j]; u.dec[j].n_kv = new_nkv[j]; }
        }
        for (int j = 0; j < u.n_cur; j++) {
            Decoder &d = u.dec[j];
            if (d.completed || d.failed) continue;
            const whisper_token_data tk = d.seq.tokens.back();
            if (tk.id > beg) {
                const int sd_new = 2 * (tk.id - beg);
                if (d.has_ts && d.seek_delta > sd_new && d.seq.result_len < i) { d.failed = true; continue; }
                d.seek_delta = sd_new; d.seq.result_len = i + 1; d.has_ts = true;
            }
            if (tk.id == V.eot || (P.max_tokens > 0 && i >= P.max_tokens) || (d.has_ts && u.seek + d.seek_delta + 100 >= u.seek_end)) {
                if (d.seq.result_len == 0 && !u.no_timestamps) {
                    if (u.seek + d.seek_delta + 100 >= u.seek_end) d.seq.result_len = i + 1;
                    else { d.failed = true; continue; }
                }
                if (P.single_segment || u.no_timestamps) { d.seq.result_len = i + 1; d.seek_delta = 100 * kChunkSec; }
                d.completed = true;
                continue;
            }
            if (i == n_max - 1 && (d.seq.result_len == 0 || d.seek_delta < 100 * kChunkSec / 2)) { d.failed = true; continue; }
        }
        bool all_done = true;
        for (int j = 0; j < u.n_cur; j++) if (!u.dec[j].completed && !u.dec[j].failed) all_done = false;
        if (all_done || i + 1 >= n_max) { finish_iteration(u); return; }
        u.phase = Phase::Step;              // rows for position n_prompt + i are built by the scheduler, then i advances
    }

    void finish_iteration(Utt &u) {
        double best_score = -INFINITY;
        for (int j = 0; j < u.n_cur; j++) {
            Decoder &d = u.dec[j];
            if (d.failed) continue;
            d.seq.tokens.resize(d.seq.result_len);
            sequence_score(P, d.seq);
            if (d.seq.result_len > 32 && d.seq.entropy < P.entropy_thold) { d.failed = true; continue; }
            if (best_score < d.seq.score) { best_score = d.seq.score; u.best = j; }
        }
        bool success = true;
        if (u.it != (int)u.temps.size() - 1) { const Decoder &d = u.dec[u.best]; if (d.failed || d.seq.avg_logprobs < P.logprob_thold) success = false; }
        if (!success) { u.it++; E.results[u.index].n_fallbacks++; start_iteration(u); return; }
        emit_window(u);
    }

    void emit_window(Utt &u) {
        Decoder &B = u.dec[u.best];
        const int seek_delta = B.seek_delta, result_len = B.seq.result_len, beg = V.beg;
        const auto &tc = B.seq.tokens;
        std::vector<int> np;
        if (!u.prompt.empty() && u.prompt.front() == V.prev) np.insert(np.end(), u.prompt.begin() + 1, u.prompt.end() - u.prompt_init.size());
        for (int i = 0; i < result_len && i < (int)tc.size(); i++) np.push_back(tc[i].id);
        u.prompt_past.swap(np);
        std::vector<Segment> &out = E.results[u.index].segs;
        const size_t n_before = out.size();
        if (!tc.empty()) {
            int i0 = 0; int64_t t0 = u.seek + 2 * (tc.front().tid - beg);
            std::string text; bool turn = false;
            for (int i = 0; i < (int)tc.size(); i++) {
                if (P.print_special || tc[i].id < V.eot) text += V.id_to_token[tc[i].id];
                if (P.tdrz_enable && tc[i].id == V.solm) turn = true;
                if (tc[i].id > beg && !P.single_segment) {
                    const int64_t t1 = u.seek + 2 * (tc[i].tid - beg);
                    if (!text.empty()) {
                        Segment s; s.t0 = t0; s.t1 = t1; s.text = text; s.speaker_turn_next = turn;
                        s.tokens.assign(tc.begin() + i0, tc.begin() + i + 1);
                        out.push_back(std::move(s));
                    }
                    text.clear();
                    while (i < (int)tc.size() && tc[i].id > beg) i++;
                    i--; t0 = t1; i0 = i + 1; turn = false;
                }
            }
            if (!text.empty()) {
                Segment s; s.t0 = t0; s.t1 = u.seek + seek_delta; s.text = text; s.speaker_turn_next = turn;
                s.tokens.assign(tc.begin() + i0, tc.end());
                out.push_back(std::move(s));
            }
        }
        if (single && P.new_segment_callback && out.size() > n_before) P.new_segment_callback(ctx, st, (int)(out.size() - n_before), P.new_segment_callback_user_data);
        remember_window(u, n_before, seek_delta);
        release_decoders(u);
        u.seek += seek_delta;
        u.encoded_at_seek = false;
        u.phase = Phase::Encode;
        if (u.seek + 100 >= u.seek_end) u.phase = Phase::Done;
    }

    int run(const float *const *pcm, const int *n, int B, bool on_device);
    int run_group(const float *const *pcm, const int *n, int g0, int G, bool on_device);
};

int Engine::Run::run(const float *const *pcm, const int *n, int B, bool on_device) {
    const Model &M = *E.model;
    n_decoders = 1;
    if (P.strategy == WHISPER_SAMPLING_GREEDY) n_decoders = P.greedy.best_of;
    else n_decoders = std::max(P.greedy.best_of, P.beam_search.beam_size);
    n_decoders = std::max(1, n_decoders);
    if (n_decoders > 8) { log_msg(2, "whisper_full: too many decoders requested (%d, max 8)\n", n_decoders); return -4; }
    // decoders beyond the state's capacity are only reachable through temperature fallback / beams
    if (n_decoders > E.max_dec_) n_decoders = E.max_dec_;
    if (P.strategy == WHISPER_SAMPLING_BEAM_SEARCH && P.beam_search.beam_size > E.max_dec_) { log_msg(2, "whisper_full: beam size %d exceeds the state's %d decoders\n", P.beam_search.beam_size, E.max_dec_); return -4; }
    if (P.audio_ctx > M.hp.n_audio_ctx) { log_msg(2, "whisper_full: audio_ctx is larger than the maximum allowed (%d > %d)\n", P.audio_ctx, M.hp.n_audio_ctx); return -5; }
    E.exp_n_audio_ctx = P.audio_ctx;
    // several sequences per utterance (beams, best_of): keep a transposed copy of the cross V so that their cross-attention
    // can share one MFMA sweep over K and V; without the memory for it the shared-sweep VALU kernel is used
    E.want_vt_ = n_decoders > 1 && E.ckvt_.ensure((size_t)E.max_utts_ * E.Ld_ * E.H_ * 64 * E.Tp_full_ * sizeof(f16));
    n_max = M.hp.n_text_ctx / 2 - 4;
    single = B == 1;
    initial_prompt_tokens.clear();
    if (P.prompt_tokens && P.prompt_n_tokens > 0) initial_prompt_tokens.assign(P.prompt_tokens, P.prompt_tokens + P.prompt_n_tokens);
    else if (P.initial_prompt && P.initial_prompt[0]) initial_prompt_tokens = tokenize(V, P.initial_prompt);
    // the ids index the token embedding on the device: reject what is not a token of this model
    for (int t : initial_prompt_tokens)
        if (t < 0 || t >= M.hp.n_vocab) { log_msg(2, "whisper_full: prompt token %d is outside the vocabulary (%d tokens)\n", t, M.hp.n_vocab); return -2; }
    // text context kept from the previous call of this state (whisper_full with no_context = false): the caller's prompt
    // tokens go in front of it. Only the one-utterance call has a "previous call" in this sense.
    if (P.no_context || !single) E.carry_prompt_past.clear();
    if (single && !E.carry_prompt_past.empty()) initial_prompt_tokens.insert(initial_prompt_tokens.end(), E.carry_prompt_past.begin(), E.carry_prompt_past.end());
    E.results.assign(B, UttResult{});
    int rc = 0;
    for (int g0 = 0; g0 < B; g0 += E.max_utts_) {
        const int r = run_group(pcm, n, g0, std::min(E.max_utts_, B - g0), on_device);
        if (r != 0 && rc == 0) rc = r;
    }
    // whisper_full leaves the state's text context updated after every call, whatever no_context was (the flag only clears it at
    // the start of a call): a call with no_context = false after one with no_context = true conditions on that call's text.
    // A failed utterance leaves what was there.
    if (single && rc == 0 && !utts.empty() && utts[0].rc == 0) E.carry_prompt_past = utts[0].prompt_past;
    return rc;
}

int Engine::Run::run_group(const float *const *pcm, const int *n, int g0, int G, bool on_device) {
    const Model &M = *E.model;
    (void)hipSetDevice(M.device);
    // ---- log-mel for the whole group
    double t0 = now_ms();
    std::vector<long> offs(G); std::vector<int> nn(G);
    const float *base = nullptr;
    if (on_device) {
        base = pcm[g0];
        for (int i = 0; i < G; i++) { offs[i] = (long)(pcm[g0 + i] - base); nn[i] = n[g0 + i]; }
    } else {
        size_t total = 0;
        for (int i = 0; i < G; i++) { offs[i] = (long)total; nn[i] = n[g0 + i]; total += (size_t)std::max(nn[i], 0); }
        if (!E.pcm_.ensure(std::max<size_t>(total, 1) * sizeof(float), false)) return -2;
        for (int i = 0; i < G; i++)
            if (nn[i] > 0 && hipMemcpyAsync(E.pcm_.as<float>() + offs[i], pcm[g0 + i], (size_t)nn[i] * sizeof(float), hipMemcpyHostToDevice, E.stream_) != hipSuccess) return -2;
        base = E.pcm_.as<float>();
    }
    for (int i = 0; i < G; i++) if (nn[i] < 0 || (nn[i] > 0 && !pcm[g0 + i])) return -2;
    if (!E.compute_mel(base, offs, nn)) return -2;
    utts.assign(G, Utt{});
    for (int i = 0; i < G; i++) { Utt &u = utts[i]; u.index = g0 + i; u.slot = i; u.mel_idx = i; init_utt(u); }
    if (hipStreamSynchronize(E.stream_) != hipSuccess) return -2;
    memo_pcm = nullptr;
    if (E.window_cache_cap > 0 && !on_device) {
        memo_mmax.assign(G, 0u);
        if (hipMemcpy(memo_mmax.data(), E.mel_max_.p, sizeof(unsigned) * G, hipMemcpyDeviceToHost) != hipSuccess) return -2;
        memo_pcm = pcm; memo_n = n; memo_params = params_hash();
    }
    E.t_mel_ms += now_ms() - t0;

    std::vector<DecRow> rows; std::vector<SampleRow> srows; std::vector<const Decoder *> seqs; std::vector<int2> copies;
    std::vector<PassRowRef> lref;
    std::vector<char> in_pass, gated;
    E.enc_events_used_ = 0;
    int rc = 0;
    for (;;) {
        if (single && P.abort_callback && P.abort_callback(P.abort_callback_user_data)) { rc = -6; break; }
        // ---- encoder for every utterance that needs a window: queued on the encoder stream, not waited for. Each chunk
        //      is followed by an event; an utterance joins decoder passes once its chunk's event has completed, so the
        //      decoder passes of earlier chunks (HBM-bound) run beside the encoder GEMMs of later ones (MFMA-bound).
        std::vector<int> need;
        for (int i = 0; i < G; i++) {
            Utt &u = utts[i];
            if (u.phase == Phase::Encode && u.seek + 100 >= u.seek_end) u.phase = Phase::Done;
            while (u.phase == Phase::Encode && reuse_window(u)) {}
            if (u.phase == Phase::Encode || u.phase == Phase::LangEncode) need.push_back(i);
        }
        if (!need.empty()) {
            t0 = now_ms();
            const int T = P.audio_ctx > 0 ? P.audio_ctx : M.hp.n_audio_ctx;
            for (size_t c0 = 0; c0 < need.size() && rc == 0; c0 += E.enc_chunk_) {
                std::vector<WinDesc> wins; std::vector<int> slots; std::vector<int> members;
                for (size_t c = c0; c < std::min(need.size(), c0 + (size_t)E.enc_chunk_); c++) {
                    Utt &u = utts[need[c]];
                    if (single && u.phase == Phase::Encode) {
                        if (P.progress_callback) P.progress_callback(ctx, st, u.seek_end > u.seek_start ? 100 * (u.seek - u.seek_start) / (u.seek_end - u.seek_start) : 0, P.progress_callback_user_data);
                        if (P.encoder_begin_callback && !P.encoder_begin_callback(ctx, st, P.encoder_begin_callback_user_data)) { u.phase = Phase::Done; continue; }
                    }
                    const MelUtt &mu = E.mel_utt_[u.mel_idx];
                    wins.push_back(WinDesc{mu.mel_off, mu.n_frames, u.n_len, u.seek, u.mel_idx});
                    slots.push_back(u.slot);
                    members.push_back(need[c]);
                }
                if (wins.empty()) continue;
                // language detection always uses the full audio context
                const int lane = (E.encoder_lanes > 1 && E.ews_[1].x.p) ? (E.next_lane_++ & 1) : 0;
                if (!E.encode_chunk(wins, slots, T, false, E.enc_streams_[lane], lane)) { rc = -6; break; }
                hipEvent_t ev = E.next_enc_event();
                if (!ev || hipEventRecord(ev, E.enc_streams_[lane]) != hipSuccess) { rc = -6; break; }
                for (int i : members) utts[i].enc_ev = ev;
            }
            if (rc) break;
            for (int i : need) {
                Utt &u = utts[i];
                if (u.phase == Phase::LangEncode) { u.encoded_at_seek = true; u.phase = Phase::LangDecode; }
                else if (u.phase == Phase::Encode) begin_window(u);
            }
            E.t_encode_ms += now_ms() - t0;       // launch time only
        }
        // ---- which utterances can decode now: those whose encoder chunk has completed and whose pipeline group is open.
        //      The batch is cut into E.pipeline_groups index ranges; a group decodes once every earlier group is finished,
        //      so decoder passes stay full (a pass costs nearly the same for 16 rows as for 64: it streams all the weights)
        //      while the encoder is still busy with the later groups.
        if (E.pipeline_groups <= 1) {              // no overlap: everything queued on the encoder lanes finishes first
            bool pending = false;
            for (int i = 0; i < G; i++) pending |= utts[i].enc_ev != nullptr;
            if (pending) {
                for (int l = 0; l < kEncLanes; l++) if (hipStreamSynchronize(E.enc_streams_[l]) != hipSuccess) rc = -6;
                if (rc) break;
                for (int i = 0; i < G; i++) utts[i].enc_ev = nullptr;
            }
        }
        const int n_groups = std::max(1, std::min(E.pipeline_groups, G));
        int open_group = n_groups;
        for (int i = 0; i < G; i++) if (utts[i].phase != Phase::Done) { open_group = (int)((long)i * n_groups / G); break; }
        bool any_ready = false, any_pending = false;
        hipEvent_t first_pending = nullptr;
        gated.assign(G, 0);
        for (int i = 0; i < G; i++) {
            Utt &u = utts[i];
            if (u.phase != Phase::LangDecode && u.phase != Phase::Prompt && u.phase != Phase::Step) continue;
            if ((int)((long)i * n_groups / G) > open_group) { gated[i] = 1; continue; }
            if (u.enc_ev) {
                const hipError_t q = hipEventQuery(u.enc_ev);
                if (q == hipSuccess) u.enc_ev = nullptr;
                else if (q == hipErrorNotReady) { any_pending = true; if (!first_pending) first_pending = u.enc_ev; }
                else { rc = -6; break; }
            }
            if (!u.enc_ev) any_ready = true;
        }
        if (rc) break;
        if (!any_ready && any_pending) {           // nothing to decode yet: wait for the oldest chunk in the queue
            if (hipEventSynchronize(first_pending) != hipSuccess) { rc = -6; break; }
            continue;
        }
        // ---- one decoder pass over every ready sequence that wants logits
        rows.clear(); srows.clear(); seqs.clear(); copies.clear(); lref.clear();
        in_pass.assign(G, 0);
        for (int i = 0; i < G; i++) {
            Utt &u = utts[i];
            if (u.enc_ev || gated[i]) continue;
            // an utterance whose sequences find no room in the K/V page pool fails alone: its rows leave the pass again
            const size_t r0 = rows.size(), s0 = srows.size(), q0 = seqs.size(), c0 = copies.size(), l0 = lref.size();
            bool fail = false;
            if (u.phase == Phase::LangDecode) {
                Decoder &d = u.dec[0];
                E.page_release(d.pages); d.n_kv = 0;
                DecRow r{V.sot, 0, (int)seqs.size(), u.slot, 0, (int)srows.size()};
                if (!E.append_kv_slot(d, 0, copies, r.kv_index)) fail = true;
                else {
                    SampleRow s{}; s.flags = 256; s.max_initial = -1; s.topk = 0;
                    rows.push_back(r); srows.push_back(s); seqs.push_back(&d); lref.push_back({i, -1});
                    in_pass[i] = 1;
                }
            } else if (u.phase == Phase::Prompt) {
                Decoder &d = u.dec[0];
                const int sidx = (int)seqs.size();
                seqs.push_back(&d);
                for (int t = 0; t < u.n_prompt; t++) {
                    DecRow r{u.prompt[t], t, sidx, u.slot, 0, -1};
                    if (!E.append_kv_slot(d, t, copies, r.kv_index)) { fail = true; break; }
                    if (t == u.n_prompt - 1) { r.logit_row = (int)srows.size(); srows.push_back(sample_state(u, d)); lref.push_back({i, 0}); }
                    rows.push_back(r);
                }
                if (!fail) in_pass[i] = 1;
            } else if (u.phase == Phase::Step) {
                for (int j = 0; j < u.n_cur; j++) {
                    Decoder &d = u.dec[j];
                    if (d.failed || d.completed) continue;
                    DecRow r{d.seq.tokens.back().id, u.n_prompt + u.i, (int)seqs.size(), u.slot, 0, (int)srows.size()};
                    if (!E.append_kv_slot(d, r.pos, copies, r.kv_index)) { fail = true; break; }
                    // the rules for the NEXT token see the sequence including the token being fed now
                    srows.push_back(sample_state(u, d));
                    rows.push_back(r); seqs.push_back(&d); lref.push_back({i, j});
                }
                if (!fail) { u.i++; in_pass[i] = 1; }
            }
            if (fail) {
                rows.resize(r0); srows.resize(s0); seqs.resize(q0); copies.resize(c0); lref.resize(l0);
                release_decoders(u);
                in_pass[i] = 0;
                u.rc = -8; u.phase = Phase::Done;
            }
        }
        if (rows.empty()) {
            bool all_done = true;
            for (int i = 0; i < G; i++) all_done &= utts[i].phase == Phase::Done;
            if (all_done) break;
            continue;                      // the only candidates of this round failed; others still wait for their encoder chunk
        }
        t0 = now_ms();
        const int T = E.enc_T_;
        LogitsFilter filter;
        if (P.logits_filter_callback)
            filter = [&](int lr, float *logits) {
                if (lref[lr].dec < 0) return;                                    // language detection row: no rules, no filter
                const Utt &fu = utts[lref[lr].utt];
                const std::vector<whisper_token_data> &tk = fu.dec[lref[lr].dec].seq.tokens;
                P.logits_filter_callback(ctx, st, tk.data(), fu.phase == Phase::Prompt ? 0 : (int)tk.size(), logits, P.logits_filter_callback_user_data);
            };
        if (!E.decoder_pass(rows, srows, seqs, copies, T, false, nullptr, P.logits_filter_callback ? &filter : nullptr)) { rc = -7; break; }
        E.t_decode_ms += now_ms() - t0;
        const SampleOut *so = E.out_host_.as<SampleOut>();
        const TopkOut *tk = (const TopkOut *)(E.out_host_.as<char>() + sizeof(SampleOut) * E.lrows_cap_);
        for (size_t lr = 0; lr < lref.size(); lr++) {
            Utt &u = utts[lref[lr].utt];
            if (lref[lr].dec < 0) continue;
            Decoder &d = u.dec[lref[lr].dec];
            d.so = so[lr];
            for (int c = 0; c < 8; c++) d.topk[c] = tk[lr * 8 + c];
            if (srows[lr].want_probs) d.probs.assign(E.probs_host_.as<float>() + lr * (size_t)E.V_, E.probs_host_.as<float>() + (lr + 1) * (size_t)E.V_);
        }
        // ---- advance the state machines of the utterances that were in the pass
        for (int i = 0; i < G; i++) {
            Utt &u = utts[i];
            if (!in_pass[i]) continue;
            if (u.phase == Phase::LangDecode) {
                // raw logits of this row are still on the device: fetch the language slice
                int lrow = -1;
                for (size_t lr = 0; lr < lref.size(); lr++) if (lref[lr].utt == i && lref[lr].dec < 0) lrow = (int)lr;
                std::vector<float> lg(kNumLangs + 1, 0.0f);
                const int first = V.sot + 1, cnt = std::min(kNumLangs, E.V_ - first);
                if (lrow < 0 || hipMemcpy(lg.data(), E.logits_.as<float>() + (size_t)lrow * E.Vp_ + first, sizeof(float) * cnt, hipMemcpyDeviceToHost) != hipSuccess) { u.rc = -3; u.phase = Phase::Done; continue; }
                int best = 0;
                for (int l = 1; l < cnt; l++) if (lg[l] > lg[best]) best = l;
                u.lang_id = best;
                E.page_release(u.dec[0].pages); u.dec[0].n_kv = 0;
                after_language(u);
            } else if (u.phase == Phase::Prompt) {
                for (int j = 1; j < u.n_cur; j++) {           // every decoder starts from the prompt's cache and logits
                    Decoder &d = u.dec[j];
                    E.page_release(d.pages);
                    d.pages = u.dec[0].pages; d.n_kv = u.dec[0].n_kv;
                    for (int p : d.pages) E.page_ref_[p]++;
                    d.so = u.dec[0].so; memcpy(d.topk, u.dec[0].topk, sizeof d.topk); d.probs = u.dec[0].probs;
                }
                process_step(u);
            } else if (u.phase == Phase::Step) {
                process_step(u);
            }
        }
    }
    if (rc) for (int l = 0; l < kEncLanes; l++) (void)hipStreamSynchronize(E.enc_streams_[l]);        // an early exit must not leave encoder work behind the caller's back
    // rc is the failure of the batch as a whole (a HIP error, an abort); an utterance that failed on its own (no language
    // found, no room in the K/V pool) carries its code in its result and does not fail the others
    for (Utt &u : utts) { release_decoders(u); if (u.rc != 0) E.results[u.index].rc = u.rc; else if (rc != 0) E.results[u.index].rc = rc; }
    return rc;
}

int Engine::full_batch(whisper_context *ctx, whisper_state *st, const whisper_full_params &params, const float *const *pcm, const int *n, int B, bool on_device) {
    if (B <= 0 || !pcm || !n) return -2;
    Run run(*this, ctx, st, params);
    page_release(ll_dec_.pages); ll_dec_.n_kv = 0;
    return run.run(pcm, n, B, on_device);
}

}  // namespace wmi
