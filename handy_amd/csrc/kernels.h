// Host-callable launchers of the HIP kernels (one translation unit per k_*.hip file).
#pragma once
#include "wmi_common.h"

namespace wmi {

// ---- k_gemm.hip : C[m][n] = sum_k A[m][k] W[n][k] with fused epilogues ----------------------
bool gemm_f16(const f16 *A, long lda, const f16 *W, long ldw, int M, int N, int K, f16 *out, long ldc, const float *bias, int gelu, float scale, hipStream_t st, bool dec = false);
bool gemm_f32(const f16 *A, long lda, const f16 *W, long ldw, int M, int N, int K, float *out, long ldc, const float *bias, hipStream_t st);
bool gemm_residual(const f16 *A, long lda, const f16 *W, long ldw, int M, int N, int K, float *x, long ldc, const float *bias, hipStream_t st, bool dec = false);
// A [N][K] weight matrix kept block-quantised on the device (ggml tensor types 2 Q4_0, 3 Q4_1, 6 Q5_0, 7 Q5_1, 8 Q8_0;
// 32 weights per block) for the few-row decoder GEMMs, which expand the blocks in registers to the very fp16 values the
// load-time expansion writes: a third to a half of the fp16 bytes per decoded token.
struct QuantW {
    const unsigned char *qs = nullptr;   // [N][K/32][16] nibble bytes, ggml order (types 2, 3, 6, 7); [N][K/32][32] int8 (type 8)
    const unsigned *dm = nullptr;        // [N][K/32] {f16 d, f16 m}; m = 0 for the formats without a minimum
    const unsigned *qh = nullptr;        // [N][K/32] fifth bits (types 6, 7)
    int ttype = 0;                       // 0: not resident
};
bool quant_repack(const void *blocks, int ttype, long n_blocks, void *qs, unsigned *dm, unsigned *qh, hipStream_t st);
// Few-row decoder GEMMs (<= 16 rows): one launch per projection, LayerNorm in the prologue, the split-K path's bits
bool gemm_rows_ok(int R, int N, int K, bool ln);
bool gemm_rows_ln_dec_qkv(const float *x, const float *ln_w, const float *ln_b, const f16 *W, int R, int d, int H, f16 *q, f16 *kc, f16 *vc, const float *bias, const DecRow *rows, hipStream_t st, const QuantW *qw = nullptr);
bool gemm_rows_ln_f16(const float *x, const float *ln_w, const float *ln_b, const f16 *W, int R, int N, int K, f16 *out, long ldc, const float *bias, int gelu, float scale, hipStream_t st, const QuantW *qw = nullptr);
bool gemm_rows_residual(const f16 *A, long lda, const f16 *W, int R, int N, int K, float *x, long ldc, const float *bias, hipStream_t st, const QuantW *qw = nullptr);
bool gemm_conv1(const f16 *melT, int n_mel, const f16 *W, int Kpad, int M, int d, f16 *h1, const float *bias, int T2, hipStream_t st);
bool gemm_conv2(const f16 *h1, const f16 *W, int M, int d, float *x, const float *bias, const float *pe, int T, hipStream_t st);
bool gemm_enc_qkv(const f16 *xn, const f16 *W, int M, int d, int H, int T, int Tp, f16 *Q, f16 *K, f16 *Vt, const float *bias, hipStream_t st);
// the same projection as two launches (Q and K; V transposed) when gemm_enc_qkv_is_split says the batch is big enough
bool gemm_enc_qkv_is_split(int M, int d, int T);
bool gemm_enc_qk(const f16 *xn, const f16 *W, int M, int d, int H, int T, int Tp, f16 *Q, f16 *K, const float *bias, hipStream_t st);
bool gemm_enc_vt(const f16 *xn, const f16 *W, int M, int d, int H, int T, int Tp, f16 *Vt, const float *bias, hipStream_t st);
bool gemm_cross_kv(const f16 *enc, const f16 *W, int M, int d, int H, int T, int Tp, int n_layer, f16 *out, const float *bias, const int *slots, hipStream_t st);
bool gemm_dec_qkv(const f16 *xn, const f16 *W, int M, int d, int H, f16 *q, f16 *kc, f16 *vc, const float *bias, const DecRow *rows, hipStream_t st);

// split-K path of the decoder (partials in ws, added up by the splitk_* consumers of k_decoder.hip)
int gemm_pick_splits(int N, int K);                       // 1 = use the fused-epilogue kernels
size_t gemm_splitk_workspace_bytes(int rows, int N, int K);   // bytes of ws[splits][rows][N] for this weight shape (0: it does not split)
bool gemm_splitk(const f16 *A, long lda, const f16 *W, long ldw, int M, int N, int K, int splits, float *ws, size_t ws_bytes, hipStream_t st);

#ifdef WMI_HARNESS        // benchmark harness build only (libwhisper_mi355_harness.so)
extern int g_tile_group_rows;
bool gemm_read_stamps(unsigned long long *out, int n);
bool gemm_variant(int variant, const f16 *A, long lda, const f16 *W, long ldw, int M, int N, int K, f16 *out, long ldc, hipStream_t st);
#endif

// ---- k_load.hip : load-time expansion of block-quantised tensors ------------------------------
int quant_block_bytes(int ttype);                          // bytes per 32-weight block of a ggml block type, 0 if not one
// blocks (device, 2-byte aligned) of ggml type `ttype` -> dst[n_blocks * 32] fp16 (device, 16-byte aligned)
bool expand_blocks_f16(const void *blocks, int ttype, long n_blocks, f16 *dst, hipStream_t st);

// ---- k_mel.hip : log-mel frontend -----------------------------------------------------------
struct MelTables {           // device-resident constants of one model
    const float *filters;    // [n_mel][201] dense table from the model file
    const int   *flo, *fhi;  // per mel bin: nonzero support [lo, hi)
    const float *fpacked;    // the supports' weights, concatenated in mel order; null when they exceed the kernel's LDS table
    const int   *fmeta;      // per mel bin: lo | count << 8 | offset into fpacked << 16
    int n_packed;
    const float *hann;       // [400]
    const float2 *tw400;     // [400] (cos, sin)(2*pi*m/400)
    int n_mel;
};
struct MelUtt {              // one per utterance
    long pcm_off;            // offset (samples) into the PCM buffer
    int  n;                  // samples
    int  n_frames;           // frames stored (multiple of kMelFramesPerBlock), raw layout [n_mel][n_frames]
    int  n_calc;             // frames that see any audio
    long mel_off;            // offset (floats) of this utterance's raw log-mel
};
constexpr int kMelFramesPerBlock = 32;
// raw log10 mel (before the max-8 clamp) for every utterance + per-utterance max (as ordered uint)
bool mel_raw(const float *pcm, const MelUtt *utts, int n_utt, int max_blocks_per_utt, const MelTables &tb, float *mel_raw, unsigned *mel_max, hipStream_t st);
// normalised copy: out[j][i] for i < n_out frames (clamp to max-8, (x+4)/4); frames past n_frames read as -10
bool mel_normalize(const float *raw, int n_mel, int n_frames, const unsigned *mel_max, float *out, int n_out, hipStream_t st);
// the same for every utterance of a mel_raw call at once: out[b][n_mel][n_out] (n_out % 4 == 0)
bool mel_normalize_batch(const float *raw, const MelUtt *utts, int n_utt, int n_mel, const unsigned *mel_max, float *out, int n_out, hipStream_t st);
// The same frontend for one utterance without any allocation (PCM and result in the caller's device memory, the kernels'
// intermediate data in a fixed device array of the library): at most kMelNoAllocMaxSamples samples in, n_out <=
// kMelNoAllocMaxFramesOut frames out. mel_noalloc_init looks the array's device address up (once per device, that device
// current); mel_noalloc returns 0, -2 for sizes beyond the maximum, another negative code on failure.
constexpr int kMelNoAllocMaxSamples = kSampleRate * kChunkSec;                        // 30 s
constexpr int kMelNoAllocMaxFramesOut = 2 * kSampleRate * kChunkSec / kHop;           // 6000: the padded length of a 30 s clip
bool mel_noalloc_init(int device);
int mel_noalloc(const float *pcm_dev, int n, const MelTables &tb, float *out_dev, int n_out, int device, hipStream_t st);
// encoder input windows: for chunk entry e, frames [seek[e], seek[e]+2T) of utterance utt[e] -> fp16 [e*(2T+2)+1+t][n_mel]
struct WinDesc { long mel_off; int n_frames; int n_len; int seek; int max_idx; };
bool mel_window(const float *raw, const unsigned *mel_max, const WinDesc *wins, int n_win, int n_mel, int T2, f16 *melT, hipStream_t st);
// same, from an already normalised mel (whisper_set_mel): n_frames == n_len, no clamp applied
bool mel_window_normalized(const float *mel, const WinDesc *wins, int n_win, int n_mel, int T2, f16 *melT, hipStream_t st);

// ---- k_encoder.hip --------------------------------------------------------------------------
// y = LayerNorm(x) * w + b (eps 1e-5); writes fp16 (out16) and/or fp32 (out32); rows gathered through `gather` if given
#ifdef WMI_HARNESS
bool lds_pattern_bench(int pattern, int do_read, int iters, float *ms_out);   // LDS access-pattern micro-benchmark
bool layer_norm_variant(int variant, const float *x, const float *w, const float *b, f16 *out16, int M, int d, hipStream_t st);   // LayerNorm A/B
#endif
bool layer_norm(const float *x, const float *w, const float *b, f16 *out16, float *out32, int M, int d, const int *gather, hipStream_t st);
// non-causal attention over T keys (padded to Tp), heads of 64: Q,K [B][H][Tp][64], Vt [B][H][64][Tp] with key t at enc_vt_pos(t) -> att [B*T][H*64]
bool enc_attention(const f16 *Q, const f16 *K, const f16 *Vt, f16 *att, int B, int H, int T, int Tp, hipStream_t st);
#ifdef WMI_HARNESS
void enc_attention_set_variant(int v);   // harness: 0 enc_attention_kernel, 1 enc_attention_pipe_kernel, 11..17 its ablations
#endif

// ---- k_decoder.hip --------------------------------------------------------------------------
bool dec_embed(const f16 *te, const float *pe, const DecRow *rows, int M, int d, float *x, hipStream_t st);
// causal self-attention over the paged cache: kc/vc [page][H][16][64], block_table [seq][kMaxPagesPerSeq]
bool dec_self_attention(const f16 *q, const f16 *kc, const f16 *vc, const int *block_table, const DecRow *rows, int M, int d, int H, f16 *att, hipStream_t st);
// cross-attention over cached encoder K/V: ckv [slot][layer][2][H][Tp][64]
// groups: {first row, count <= max_group <= 8} runs of consecutive rows of one utterance (nullptr / max_group 1: one row per workgroup)
bool dec_cross_attention(const f16 *q, const f16 *ckv, const DecRow *rows, const int2 *groups, int n_groups, int max_group, int M, int d, int H, int T, int Tp,
                         int n_layer, int layer, f16 *att, hipStream_t st);
// logits rules + log-softmax + argmax / top-k / probabilities for every logit row
// (the masked logits are written back in place; nst_bitmap has one bit per token id)
// any_extra: some row has topk > 0 or want_probs; workspace: sample_workspace_bytes(max rows) of device memory
size_t sample_workspace_bytes(int max_rows);
bool sample_rows(float *logits, long ld, int R, const SampleRow *srow, const VocabIds &vi, const unsigned *nst_bitmap,
                 SampleOut *out, TopkOut *topk, int topk_stride, float *probs, bool any_extra, void *workspace, hipStream_t st);
// copy KV page contents in every layer (beam fork of a partially filled page): pairs of (src_page, dst_page)
bool copy_pages(f16 *kpool, f16 *vpool, long layer_stride, int n_layer, const int2 *pairs, int n_pairs, int H, hipStream_t st);
// MFMA form for groups of up to 16 rows (count in groups[].y), reading V through its transposed copy vt[slot][layer][head][64][Tp]
bool cross_v_transpose(const f16 *ckv, const int *slots, int n_slots, f16 *vt, int H, int T, int Tp, int n_layer, hipStream_t st);
bool dec_cross_attention_mq(const f16 *q, const f16 *ckv, const f16 *vt, const DecRow *rows, const int2 *groups, int n_groups, int d, int H, int T, int Tp,
                            int n_layer, int layer, f16 *att, hipStream_t st);
// consumers of split-K partials ws[split][M][N] (summed in split order, then the same arithmetic as the fused epilogues)
bool splitk_residual_ln(float *x, const float *ws, int splits, const float *bias, const float *ln_w, const float *ln_b, f16 *out, int M, int d,
                        const int *gather, int n_out, hipStream_t st);         // x[m] += sum + bias; out[i] = LN(x[gather ? gather[i] : i])
bool splitk_f16(const float *ws, int splits, const float *bias, int gelu, float scale, f16 *out, int M, int N, hipStream_t st);
bool splitk_dec_qkv(const float *ws, int splits, const float *bias, const DecRow *rows, int M, int d, int H, f16 *q, f16 *kc, f16 *vc, hipStream_t st);

}  // namespace wmi
