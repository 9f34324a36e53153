// Internal shared declarations of the MI355X Whisper engine (host + device).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>

namespace wmi {

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kSampleRate = 16000;
constexpr int kNFft = 400;
constexpr int kHop = 160;
constexpr int kChunkSec = 30;
constexpr int kNBins = 201;
constexpr int kHeadDim = 64;          // every Whisper size uses 64-wide heads
constexpr int kPageTokens = 16;       // self-attention KV page
constexpr int kMaxTextCtx = 448;
constexpr int kMaxPagesPerSeq = kMaxTextCtx / kPageTokens;

// logging hook (set through whisper_log_set)
void log_msg(int level, const char *fmt, ...);

#define WMI_HIP_OK(expr)                                                                      \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) {                                                              \
            ::wmi::log_msg(2, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); \
            return false;                                                                     \
        }                                                                                     \
    } while (0)

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline size_t round_up_sz(size_t x, size_t m) { return (x + m - 1) / m * m; }

// ---------------------------------------------------------------------------------------------
// Per-row decoder metadata, uploaded once per decoder pass.
struct DecRow {
    int32_t token;      // input token id
    int32_t pos;        // position in the text context
    int32_t seq;        // block-table row (one per live decoder sequence)
    int32_t utt;        // cross-attention K/V slot of the utterance
    int32_t kv_index;   // page * kPageTokens + offset where this row's self K/V is stored
    int32_t logit_row;  // >= 0: this row produces logits into that row of the logits buffer
};

// Per-logit-row sampling state (the inputs whisper_process_logits reads from the decoder).
struct SampleRow {
    float   temperature;
    int32_t flags;          // bit0 is_initial, bit1 last_was_ts, bit2 penultimate_was_ts, bit3 has_ts,
                            // bit4 suppress_blank, bit5 no_timestamps, bit6 suppress_nst, bit7 tdrz_enable,
                            // bit8 raw (no rules: used by whisper_decode / language detection)
                            // bit9 / bit10: only the rules before / after the point where whisper_full calls logits_filter_callback
    int32_t ts_min;         // monotonic rule: timestamps below token_beg + ts_min are masked (has_ts)
    int32_t max_initial;    // first-position rule: timestamps above token_beg + max_initial masked; <0 off
    int32_t n_sampled;      // tokens sampled so far in this sequence
    int32_t force_n;        // benchmark control: >0 => no EOT before, only EOT at, this many tokens
    int32_t topk;           // 0: argmax only; k>0: also write the k most probable tokens
    int32_t want_probs;     // 1: write the full probability row (host-side sampling at T>0)
};

struct SampleOut {          // one per logit row
    int32_t id, tid;
    float   p, plog, pt, ptsum;
    float   lse;            // log-sum-exp of the masked logits (for host-side logprob of other ids)
    int32_t pad;
};

struct TopkOut { int32_t id; float p, plog; int32_t pad; };

struct VocabIds {
    int32_t n_vocab, eot, sot, translate, transcribe, solm, prev, nosp, tok_not, beg, space, n_lang;
};

}  // namespace wmi
