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
// 1/sqrt(64), applied to the encoder's Q after its rounding to fp16 (exact): Q and K carry the reference's bits.
constexpr float kEncQScale = 0.125f;
// Position of key t in the encoder's transposed V (Vt[b][h][e][.]): within every group of 16 keys the middle two 4-key chunks
// trade places (bits 2 and 3 of t swap), so that the two chunks one fragment of the attention's P V product needs sit in one
// 16-byte piece, which the attention kernel's LDS-DMA then moves whole.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int enc_vt_pos(int t) { return (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1); }
constexpr int kPageTokens = 16;       // self-attention KV page
constexpr int kMaxTextCtx = 448;
constexpr int kMaxPagesPerSeq = kMaxTextCtx / kPageTokens;

#if defined(__HIPCC__)
// tanh-form GELU of the f16-rounded input, rounded to f16 (the values of ggml's f16 lookup table), for the four values of
// a lane at once on packed fp32 pairs (v_pk_mul_f32 / v_pk_fma_f32): 0.5 h (1 + tanh u) = h (1 - 1/(exp(2u) + 1)) with the
// fused 1 + m*h and h - h*r written out as fma, so that every kernel that includes this gets the same bits; it is about half the instructions, and the
// fc1 epilogue is bound by their number (two transcendentals and ~10 other operations per value, 128 values per lane
// and tile).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f16x4 gelu_f16_dev4(f32x4 x) {
    const f16x4 h16 = {(f16)x[0], (f16)x[1], (f16)x[2], (f16)x[3]};
    f16x4 out;
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const f32x2 h = {(float)h16[2 * p], (float)h16[2 * p + 1]};
        const f32x2 a = 0.79788456080286535587989211986876f * h;
        const f32x2 m = 0.044715f * h;
        const f32x2 t = __builtin_elementwise_fma(m, h, f32x2{1.0f, 1.0f});
        const f32x2 u = (a * t) * 2.8853900817779268f;
        const f32x2 e = {__builtin_amdgcn_exp2f(u[0]), __builtin_amdgcn_exp2f(u[1])};
        const f32x2 d = e + 1.0f;
        const f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        const f32x2 g = __builtin_elementwise_fma(-r, h, h);
        // No guards here: for h >= 10 the reciprocal is below 2^-125 and h - h*r rounds to h, for h <= -10 exp(2u) is below
        // 2^-125, r is exactly 1 and h - h is +0 -- the values the guarded scalar form returns. (The two differ for
        // infinite inputs only, NaN here; an infinite activation turns into NaN in the next LayerNorm either way.)
        out[2 * p] = (f16)g[0];
        out[2 * p + 1] = (f16)g[1];
    }
    return out;
}
#endif

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
