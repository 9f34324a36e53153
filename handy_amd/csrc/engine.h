// Batched whisper_full engine: per-utterance state machines on the host drive batched encoder
// chunks and decoder passes on one GPU stream.
#pragma once
#include <array>
#include <functional>
#include <map>
#include <random>
#include <string>
#include <deque>
#include <vector>

#include "../../include/wmi.h"
#include "model.h"

namespace wmi {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool ensure(size_t bytes, bool zero = true);
    void release();
    template <class T> T *as() const { return (T *)p; }
};
struct PinnedBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool ensure(size_t bytes);
    void release();
    template <class T> T *as() const { return (T *)p; }
};

struct Segment {
    int64_t t0 = 0, t1 = 0;
    std::string text;
    std::vector<whisper_token_data> tokens;
    bool speaker_turn_next = false;
};
struct UttResult {
    std::vector<Segment> segs;
    int lang_id = 0, n_fallbacks = 0, rc = 0;
};

struct Sequence {
    std::vector<whisper_token_data> tokens;
    int result_len = 0;
    double sum_logprobs_all = 0, sum_logprobs = 0, avg_logprobs = 0, entropy = 0, score = 0;
};

struct Decoder {
    Sequence seq;
    int seek_delta = 0;
    bool failed = false, completed = false, has_ts = false;
    std::vector<int> pages;      // self-attention KV pages of this sequence, in position order
    int n_kv = 0;                // tokens whose K/V are in the cache
    std::mt19937 rng;
    SampleOut so{};              // selection computed from the latest logits of this sequence
    TopkOut topk[8];
    std::vector<float> probs;    // full probability row when the host has to draw from it
};

enum class Phase { LangEncode, LangDecode, Encode, Prompt, Step, Done };

struct Utt {
    int index = 0;               // position in the caller's batch
    int slot = 0;                // cross-attention K/V slot == position inside the running group
    int mel_idx = 0;
    int n_len = 0, n_len_org = 0;
    Phase phase = Phase::Done;
    int seek = 0, seek_start = 0, seek_end = 0;
    int it = 0, n_cur = 1, i = 0, n_prompt = 0;
    float t_cur = 0;
    std::vector<float> temps;
    std::vector<int> prompt, prompt_past, prompt_init;
    std::vector<Decoder> dec;
    int best = 0;
    int lang_id = 0;
    bool no_timestamps = false;
    bool encoded_at_seek = false; // the cross K/V in the slot already belong to window `seek`
    uint64_t chain = 0;          // hash of every window of this call so far (inputs and outcomes)
    uint64_t cur_key = 0;        // key of the window being decoded (0: not eligible for the window memo)
    int fb0 = 0;                 // fallback count when the window began
    hipEvent_t enc_ev = nullptr; // recorded on the encoder stream behind this utterance's chunk; cleared once seen complete
    int rc = 0;
};

// Result of one interior 30 s window, kept so that a later call whose audio starts with the same samples (Handy's live
// coaching re-transcribes the whole growing recording every few seconds) can take the window's outcome instead of
// encoding and decoding it again. A window is only taken when everything that determines it is identical: see
// Engine::Run::window_key.
struct WindowMemo {
    uint64_t key = 0, chain_after = 0;
    int seek_delta = 0, n_fallbacks = 0;
    std::vector<Segment> segs;
    std::vector<int> prompt_past;
    std::vector<std::mt19937> rng;
};

struct PassRowRef { int utt, dec; };   // which (utterance, decoder) a logit row belongs to

class Engine {
public:
    Engine(Model *m, int max_utts, int max_dec, int enc_chunk);
    ~Engine();
    bool ok() const { return ok_; }

    // whisper_full semantics over a batch
    int full_batch(whisper_context *ctx, whisper_state *st, const whisper_full_params &params, const float *const *pcm, const int *n, int B, bool on_device);

    // low-level stages (utterance slot 0)
    int pcm_to_mel(const float *pcm, int n);
    int mel_batch_device(const float *pcm_dev, const long long *offs, const int *n, int B, float *out_dev = nullptr, int n_frames_out = 0);
    int set_mel(const float *data, int n_len, int n_mel);
    int encode(int offset, int audio_ctx);
    int decode(const int *tokens, int n_tokens, int n_past, bool logits_all, float *logits_all_out);
    int lang_auto_detect(int offset_ms, float *probs);
    int n_len() const { return utt_n_len_org_.empty() ? 0 : utt_n_len_org_[0]; }
    float *logits_host() { return logits_host_.data(); }
    int get_mel(int utt, float *out, int n_frames);
    int get_encoder_output(float *out);
    int get_cross_kv(int layer, float *k_out, float *v_out);

    Model *model;
    std::vector<UttResult> results;
    int force_tokens = 0;
    std::vector<int> carry_prompt_past;  // text context of the last one-utterance call (used when the next one sets no_context = false)
    int window_cache_cap = 0;            // windows remembered between calls (0 = off)
    std::deque<WindowMemo> window_memos;
    long window_cache_hits = 0, window_cache_stores = 0;
    int pipeline_groups = 1;             // index ranges of a batch that decode one after the other while later ones still encode
    int encoder_lanes = 1;               // 2: alternate encoder chunks between two streams/workspaces
    bool set_encoder_lanes(int n);
    bool beam_sampled = false;
    int exp_n_audio_ctx = 0;

    // profiling
    bool profiling = false;
    bool quant_decode = true;
    void set_decoder_lanes(int n) { dec_lanes_ = n < 1 ? 1 : (n > 2 ? 2 : n); }
    void set_pass_graphs(bool on) { use_graphs_ = on; if (!on) clear_graphs(); }    // decoder passes of a repeated shape replayed as HIP graphs (default on)           // few-row decode passes read the decoder weights as the file's blocks where the model holds them
    unsigned profile_mask = ~0u;        // bit per wmi_kernel_class: which classes get event pairs while profiling is on
    wmi_kernel_stat stats[WMI_K_COUNT]{};
    void collect_stats();
    void reset_stats();
    // timings (ms, host clock around synchronised phases)
    double t_mel_ms = 0, t_encode_ms = 0, t_decode_ms = 0, t_load_ms = 0;
    int n_encode = 0, n_decode = 0;
    double t_dec_launch_ms = 0, t_dec_wait_ms = 0;   // decoder passes: host time building + launching, then waiting for the stream
    long n_passes = 0;

private:
    bool ok_ = false;
    int max_utts_, max_dec_, enc_chunk_;
    hipStream_t stream_ = nullptr;       // decoder passes, log-mel, the low-level whisper_encode/decode calls
    hipStream_t prof_stream_ = nullptr;  // where the profiling events of the kernels being launched go
    int d_, H_, Le_, Ld_, T_full_, n_mel_, V_, Vp_;

    // mel
    DevBuf pcm_, mel_raw_, mel_max_, mel_utt_dev_, mel_norm_tmp_;
    std::vector<MelUtt> mel_utt_;
    std::vector<int> utt_n_len_, utt_n_len_org_;
    bool mel_user_ = false;          // whisper_set_mel: mel_raw_ holds an already normalised mel

    // encoder workspace
    // Two encoder lanes (stream + workspace set): whisper_full alternates chunks between them, so while one chunk's
    // kernel drains (last partial round of tiles, launch gap) the other chunk's kernel already fills the idle CUs.
    struct EncWs { DevBuf melT, h1, x, xn, q, k, vt, att, hid, enc16; };
    static constexpr int kEncLanes = 2;
    EncWs ews_[kEncLanes];
    hipStream_t enc_streams_[kEncLanes] = {nullptr, nullptr};
    // second row lane of a decoder pass (see decoder_pass): its stream, fork / join events and split-K workspace
    static constexpr int kDecLaneMinRows = 16;
    int dec_lanes_ = 1;
    hipStream_t lane_stream_ = nullptr;
    hipEvent_t lane_fork_ = nullptr, lane_join_ = nullptr;
    DevBuf lane_ws_;
    int next_lane_ = 0;
    DevBuf enc32_, win_dev_, slots_dev_;
    PinnedBuf win_host_, slots_host_;    // per-chunk metadata: a ring, so that several chunks can be queued before the first one runs
    int ring_pos_ = 0, ring_cap_ = 0;
    std::vector<hipEvent_t> enc_events_; // pool of encoder-chunk completion events
    size_t enc_events_used_ = 0;
    hipEvent_t next_enc_event();
    int enc_T_ = 0;                  // audio context of the last encode (cross K/V length)
    DevBuf ckv_;
    DevBuf ckvt_;                    // transposed copy of the cross V, kept only while runs decode several sequences per utterance
    bool want_vt_ = false;
    int Tp_full_ = 0;

    // decoder workspace
    int rows_cap_ = 0, lrows_cap_ = 0;
    DevBuf dx_, dxn_, dq_, datt_, dhid_, dxl_, logits_, probs_;
    DevBuf kpool_, vpool_;
    int n_pages_ = 0;
    long layer_stride_ = 0;
    std::vector<int> free_pages_, page_ref_;
    DevBuf meta_dev_, out_dev_, sample_ws_, splitk_ws_;
    PinnedBuf meta_host_, out_host_, probs_host_;
    std::vector<float> logits_host_;
    std::vector<float> filter_rows_;  // logit rows on the host while the caller's logits_filter_callback looks at them
    Decoder ll_dec_;                 // sequence used by whisper_decode()
    std::vector<int2> xgroups_;      // cross-attention row groups of the pass being built

    // decoder passes replayed as HIP graphs, by pass shape (decoder_pass)
    struct PassKey { std::array<int, 10> v; bool operator<(const PassKey &o) const { return v < o.v; } };
    struct PassGraph { hipGraphExec_t exec = nullptr; int seen = 0; bool bad = false; };
    std::map<PassKey, PassGraph> graphs_;
    bool use_graphs_ = true;
    void clear_graphs();

    // event pool for profiling
    struct EvPair { hipEvent_t a, b; int cls; double work; };
    std::vector<EvPair> ev_pool_;
    size_t ev_used_ = 0;
    void prof_begin(int cls, double work);
    void prof_end();
    bool prof_open_ = false;

    bool alloc_workspaces();
    bool ensure_rows(int rows);
    bool compute_mel(const float *pcm_dev_base, const std::vector<long> &offs, const std::vector<int> &n);
    bool encode_chunk(const std::vector<WinDesc> &wins, const std::vector<int> &slots, int T, bool want_f32, hipStream_t s, int lane);
    struct PassOut { std::vector<SampleOut> so; std::vector<TopkOut> topk; };
    // filter (optional): called per logit row with that row's logits in host memory, between the two halves of the rules
    using LogitsFilter = std::function<void(int logit_row, float *logits)>;
    bool decoder_pass(std::vector<DecRow> &rows, const std::vector<SampleRow> &srows, const std::vector<const Decoder *> &seqs,
                      const std::vector<int2> &page_copies, int T, bool want_raw_logits_host, float *logits_all_out, const LogitsFilter *filter = nullptr);
    // KV pages
    int page_alloc();
    void page_release(std::vector<int> &pages);
    bool append_kv_slot(Decoder &d, int pos, std::vector<int2> &copies, int &kv_index);

    // whisper_full pieces
    struct Run;
    friend struct Run;
};

}  // namespace wmi

struct whisper_context {
    wmi::Model *model = nullptr;
    whisper_state *default_state = nullptr;
    whisper_context_params params{};
    double t_load_ms = 0;
};
struct whisper_state {
    wmi::Engine *engine = nullptr;
    whisper_context *ctx = nullptr;
    std::string token_text_tmp;
};
