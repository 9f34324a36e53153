// fp16 x fp16 -> fp32 "NT" GEMM for gfx950:  C[m][n] = sum_k Act[m][k] * W[n][k]   (+ fused epilogue)
//
// Both operands are K-contiguous (activations row-major, weights stored [out][in] as in the ggml
// file), so every MFMA fragment is one 16-byte run along K. Structure per workgroup (8 waves as 4x2
// on a 256x128 tile, or 4 waves as 2x2 on a 64x64 tile):
//   * BM x 64 activation tile and BN x 64 weight tile per K-step, copied global->LDS with the
//     16-byte LDS-DMA form (__builtin_amdgcn_global_load_lds); 3-4 LDS stages so that one or two
//     whole K-steps of loads stay in flight across each barrier; one barrier per K-step.
//   * LDS rows are 128 B; 16-byte chunk c of row r sits in slot c ^ (r & 7). The DMA writes LDS
//     linearly (wave base + lane*16), so the permutation is applied to each lane's SOURCE address
//     and again to the fragment read address: ds_read_b128 is then conflict-free.
//   * v_mfma_f32_16x16x32_f16 with the WEIGHT rows as the A operand and the activation rows as the
//     B operand: a lane ends up holding 4 consecutive output columns n of one row m, so row-major
//     epilogue accesses are 8- or 16-byte vectors.
//   * 1-D grid remapped so that each XCD (private L2) owns a contiguous run of tiles that share
//     activation row panels.
// Bound: MFMA (dense fp16 peak ~2.5 PFLOP/s). Algorithmic work: 2*M*N*K flop per launch.
#include "kernels.h"

#include <atomic>

namespace wmi {

constexpr int BK = 64;             // halfs per K-step (128 B rows in LDS)

__device__ __forceinline__ float gelu_f16_dev(float x) {
    // tanh-form GELU evaluated on the f16-rounded input and rounded to f16 (the values ggml's f16
    // lookup table holds), with the table's guards: x <= -10 -> 0, x >= 10 -> x.
    // 0.5 h (1 + tanh u) = h (1 - 1/(exp(2u) + 1)); exp and reciprocal are the hardware approximations,
    // whose error (~1 ulp of f32) disappears in the rounding to f16.
    const float h = (float)(f16)x;
    const float u = 0.79788456080286535587989211986876f * h * (1.0f + 0.044715f * h * h);
    const float e = __builtin_amdgcn_exp2f(u * 2.8853900817779268f);          // exp(2u)
    const float g = (float)(f16)(h - h * __builtin_amdgcn_rcpf(e + 1.0f));
    return x >= 10.0f ? x : (x <= -10.0f ? 0.0f : g);
}

// ------------------------------------------------------------------------------------------ epilogues
// Each epilogue receives one lane's 4 accumulators: row m, columns n..n+3 (n % 4 == 0).

// An epilogue whose output is f16 and contiguous along n also offers value() (the 4 results of one
// lane, not stored) and store8() (8 consecutive columns of one row, n % 8 == 0): the big-tile kernel
// then pairs lanes so that every store instruction moves 16 bytes per lane.

struct EpiF16 {            // out[m][n] = f16(act((acc + bias) * scale))
    static constexpr bool kWide = true;
    f16 *out; long ldc; const float *bias; int gelu; float scale;
    __device__ __forceinline__ f16x4 value(int m, int n, f32x4 a) const {
        if (bias) { const f32x4 b = *(const f32x4 *)(bias + n); a += b; }
        a *= scale;
        if (gelu) return gelu_f16_dev4(a);
        return f16x4{(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3]};
    }
    // The staged whole-tile epilogue asks for a column's bias once (col_bias) and names the mode (bit 0: bias, bit 1: GELU) at
    // compile time, so that its 32 value_b() calls per lane are straight-line code: same arithmetic, same bits as value().
    static constexpr bool kColBias = true;
    static constexpr int kModes = 4;
    __device__ __forceinline__ int mode() const { return (bias ? 1 : 0) | (gelu ? 2 : 0); }
    __device__ __forceinline__ f32x4 col_bias(int n) const { return bias ? *(const f32x4 *)(bias + n) : f32x4{0, 0, 0, 0}; }
    template <int MODE> __device__ __forceinline__ f16x4 value_b(int m, int n, f32x4 a, f32x4 b) const {
        if (MODE & 1) a += b;
        a *= scale;
        if (MODE & 2) return gelu_f16_dev4(a);
        return f16x4{(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3]};
    }
    __device__ __forceinline__ bool wide(int n) const { return true; }
    __device__ __forceinline__ bool wide_block(int n0, int bn) const { return true; }
    __device__ __forceinline__ void store8(int m, int n, f16x8 h) const { *(f16x8 *)(out + (long)m * ldc + n) = h; }
    __device__ __forceinline__ void operator()(int m, int n, f32x4 a) const { *(f16x4 *)(out + (long)m * ldc + n) = value(m, n, a); }
};

struct EpiF32 {            // out[m][n] = acc + bias
    static constexpr bool kWide = false;
    float *out; long ldc; const float *bias;
    __device__ __forceinline__ void operator()(int m, int n, f32x4 a) const {
        if (bias) { const f32x4 b = *(const f32x4 *)(bias + n); a += b; }
        *(f32x4 *)(out + (long)m * ldc + n) = a;
    }
};

struct EpiResidual {       // x[m][n] += acc + bias   (fp32 residual stream, in place)
    static constexpr bool kWide = false;
    static constexpr bool kReadsOutput = true;      // load() early, apply() later: the staged epilogue hides the read's latency
    float *x; long ldc; const float *bias;
    __device__ __forceinline__ f32x4 load(int m, int n) const { return *(const f32x4 *)(x + (long)m * ldc + n); }
    __device__ __forceinline__ void apply(int m, int n, f32x4 a, f32x4 r) const {
        const f32x4 b = *(const f32x4 *)(bias + n);
        *(f32x4 *)(x + (long)m * ldc + n) = r + (a + b);
    }
    __device__ __forceinline__ void operator()(int m, int n, f32x4 a) const { apply(m, n, a, load(m, n)); }
};

struct EpiConv1 {          // rows have period P = 2T+2 per utterance; h1[m+1][n] = t < 2T ? gelu(acc+b) : 0
    static constexpr bool kWide = true;
    f16 *h1; int d; const float *bias; int period; int t_valid;
    __device__ __forceinline__ f16x4 value(int m, int n, f32x4 a) const {
        const int t = m % period;
        f16x4 h = {(f16)0, (f16)0, (f16)0, (f16)0};
        if (t < t_valid) {
            const f32x4 b = *(const f32x4 *)(bias + n); a += b;
            h = f16x4{(f16)gelu_f16_dev(a[0]), (f16)gelu_f16_dev(a[1]), (f16)gelu_f16_dev(a[2]), (f16)gelu_f16_dev(a[3])};
        }
        return h;
    }
    __device__ __forceinline__ bool wide(int n) const { return true; }
    __device__ __forceinline__ bool wide_block(int n0, int bn) const { return true; }
    __device__ __forceinline__ void store8(int m, int n, f16x8 h) const { *(f16x8 *)(h1 + (long)(m + 1) * d + n) = h; }
    __device__ __forceinline__ void operator()(int m, int n, f32x4 a) const { *(f16x4 *)(h1 + (long)(m + 1) * d + n) = value(m, n, a); }
};

struct EpiConv2 {          // rows have period T+1; x[b*T+t][n] = gelu(acc+b) + pos_emb[t][n], junk row t==T dropped
    static constexpr bool kWide = false;
    float *x; int d; const float *bias; const float *pe; int T;
    __device__ __forceinline__ void operator()(int m, int n, f32x4 a) const {
        const int b = m / (T + 1), t = m - b * (T + 1);
        if (t >= T) return;
        const f32x4 bb = *(const f32x4 *)(bias + n); a += bb;
        const f32x4 p = *(const f32x4 *)(pe + (long)t * d + n);
        f32x4 o = {gelu_f16_dev(a[0]) + p[0], gelu_f16_dev(a[1]) + p[1], gelu_f16_dev(a[2]) + p[2], gelu_f16_dev(a[3]) + p[3]};
        *(f32x4 *)(x + ((long)b * T + t) * d + n) = o;
    }
};

struct EpiEncQKV {         // n in [0,d): Q*kEncQScale -> Q[b][h][t][64]; [d,2d): K -> K[b][h][t][64]; [2d,3d): V -> Vt[b][h][64][enc_vt_pos(t)]
    static constexpr bool kWide = true;
    f16 *Q, *K, *Vt; const float *bias; int d, H, T, Tp;
    int b0 = 0, base = 0;                   // begin_tile: utterance of the tile's first row, and that utterance's first row
    __device__ __forceinline__ void begin_tile(int m0, int n0) { b0 = m0 / T; base = b0 * T; }
    __device__ __forceinline__ f16x4 value(int m, int n, f32x4 a) const {
        const f32x4 bb = *(const f32x4 *)(bias + n); a += bb;
        if (n < d) a *= kEncQScale;
        return f16x4{(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3]};
    }
    static constexpr bool kColBias = true;
    static constexpr int kModes = 1;
    __device__ __forceinline__ int mode() const { return 0; }
    __device__ __forceinline__ f32x4 col_bias(int n) const { return *(const f32x4 *)(bias + n); }
    template <int MODE> __device__ __forceinline__ f16x4 value_b(int m, int n, f32x4 a, f32x4 b) const {
        a += b;
        if (n < d) a *= kEncQScale;
        return f16x4{(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3]};
    }
    __device__ __forceinline__ bool wide(int n) const { return n < 2 * d; }      // the V part is stored transposed, element by element
    __device__ __forceinline__ bool wide_block(int n0, int bn) const { return n0 + bn <= 2 * d; }
    __device__ __forceinline__ void store8(int m, int n, f16x8 o) const {       // begin_tile has run
        int b = b0, t = m - base;
        if (T >= 256) { if (t >= T) { t -= T; b++; } } else { b = m / T; t = m - b * T; }
        const int part = n >= d, c = n - part * d, h = c >> 6, e = c & 63;
        *(f16x8 *)((part ? K : Q) + (((long)b * H + h) * Tp + t) * 64 + e) = o;
    }
    __device__ __forceinline__ void operator()(int m, int n, f32x4 a) const {
        const int b = m / T, t = m - b * T;
        const f32x4 bb = *(const f32x4 *)(bias + n); a += bb;
        const int part = n / d, c = n - part * d, h = c >> 6, e = c & 63;
        if (part == 0) {
            f16x4 o = {(f16)(a[0] * kEncQScale), (f16)(a[1] * kEncQScale), (f16)(a[2] * kEncQScale), (f16)(a[3] * kEncQScale)};
            *(f16x4 *)(Q + (((long)b * H + h) * Tp + t) * 64 + e) = o;
        } else if (part == 1) {
            f16x4 o = {(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3]};
            *(f16x4 *)(K + (((long)b * H + h) * Tp + t) * 64 + e) = o;
        } else {
            f16 *p = Vt + (((long)b * H + h) * 64 + e) * Tp + enc_vt_pos(t);
            p[0] = (f16)a[0]; p[Tp] = (f16)a[1]; p[2 * Tp] = (f16)a[2]; p[3 * Tp] = (f16)a[3];
        }
    }
};

// The V projection computed transposed, Vt = Wv xn^T: output row m = h*64 + e (a weight row), column n = b*T + t (an
// activation row), so that the rows of a tile run along t -- the direction Vt[b][h][e][t] is contiguous in -- and go
// through the wide staged epilogue like every other f16 output. (As columns of the fused QKV GEMM the V part leaves the
// accumulators as 2-byte stores scattered over 64 rows of Vt: 45 000 cycles per tile against 9 000.)
// T % 4 == 0: four consecutive columns never straddle two utterances.
struct EpiVt {
    static constexpr bool kWide = true;
    f16 *Vt; const float *bias; int H, T, Tp, rows;
    int b0 = 0, base = 0;                   // begin_tile: utterance of the tile's first column, and that utterance's first column
    __device__ __forceinline__ void begin_tile(int m0, int n0) { b0 = n0 / T; base = b0 * T; }
    __device__ __forceinline__ f16x4 value(int m, int n, f32x4 a) const {
        const float b = bias[m];
        return f16x4{(f16)(a[0] + b), (f16)(a[1] + b), (f16)(a[2] + b), (f16)(a[3] + b)};
    }
    __device__ __forceinline__ bool wide(int n) const { return true; }
    __device__ __forceinline__ bool wide_block(int n0, int bn) const { return true; }
    __device__ __forceinline__ void store4(int m, int n, f16x4 o) const {
        if (n >= rows) return;
        int b = b0, t = n - base;
        if (T >= 256) { if (t >= T) { t -= T; b++; } } else { b = n / T; t = n - b * T; }
        *(f16x4 *)(Vt + (((long)b * H + (m >> 6)) * 64 + (m & 63)) * Tp + enc_vt_pos(t)) = o;      // t % 4 == 0: a whole 4-key chunk
    }
    __device__ __forceinline__ void store8(int m, int n, f16x8 o) const {
        store4(m, n, f16x4{o[0], o[1], o[2], o[3]});
        store4(m, n + 4, f16x4{o[4], o[5], o[6], o[7]});
    }
    __device__ __forceinline__ void operator()(int m, int n, f32x4 a) const { store4(m, n, value(m, n, a)); }
};

struct EpiCrossKV {        // n = (layer*2 + kv)*d + h*64 + e ; out[slot(b)][layer][kv][h][t][64]
    static constexpr bool kWide = true;
    f16 *out; const float *bias; const int *slots; int d, H, T, Tp, n_layer;
    int b0 = -1, base = 0, lk0 = 0, nbase = 0;      // begin_tile: utterance / (layer, k|v) of the tile's first row / column
    __device__ __forceinline__ void begin_tile(int m0, int n0) { b0 = m0 / T; base = b0 * T; lk0 = n0 / d; nbase = lk0 * d; }
    __device__ __forceinline__ f16 *dest(int m, int n) const {
        int b, t, lk, c;
        if (b0 >= 0 && T >= 256 && d >= 256) {
            b = b0; t = m - base; if (t >= T) { t -= T; b++; }
            lk = lk0; c = n - nbase; if (c >= d) { c -= d; lk++; }
        } else {
            b = m / T; t = m - b * T; lk = n / d; c = n - lk * d;
        }
        const int h = c >> 6, e = c & 63;
        const long slot = slots[b];
        return out + ((((slot * n_layer * 2 + lk) * H + h) * (long)Tp) + t) * 64 + e;
    }
    __device__ __forceinline__ f16x4 value(int m, int n, f32x4 a) const {
        const f32x4 bb = *(const f32x4 *)(bias + n); a += bb;
        return f16x4{(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3]};
    }
    static constexpr bool kColBias = true;
    static constexpr int kModes = 1;
    __device__ __forceinline__ int mode() const { return 0; }
    __device__ __forceinline__ f32x4 col_bias(int n) const { return *(const f32x4 *)(bias + n); }
    template <int MODE> __device__ __forceinline__ f16x4 value_b(int m, int n, f32x4 a, f32x4 b) const {
        a += b;
        return f16x4{(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3]};
    }
    __device__ __forceinline__ bool wide(int n) const { return true; }
    __device__ __forceinline__ bool wide_block(int n0, int bn) const { return true; }
    __device__ __forceinline__ void store8(int m, int n, f16x8 o) const { *(f16x8 *)dest(m, n) = o; }
    __device__ __forceinline__ void operator()(int m, int n, f32x4 a) const { *(f16x4 *)dest(m, n) = value(m, n, a); }
};

struct EpiDecQKV {         // q -> qbuf[m][d] * 1/8 ; k,v -> paged cache [page][H][16][64] at rows[m].kv_index
    static constexpr bool kWide = false;
    f16 *q, *kc, *vc; const float *bias; const DecRow *rows; int d, H;
    __device__ __forceinline__ void operator()(int m, int n, f32x4 a) const {
        const f32x4 bb = *(const f32x4 *)(bias + n); a += bb;
        const int part = n / d, c = n - part * d;
        if (part == 0) {
            f16x4 o = {(f16)(a[0] * 0.125f), (f16)(a[1] * 0.125f), (f16)(a[2] * 0.125f), (f16)(a[3] * 0.125f)};
            *(f16x4 *)(q + (long)m * d + c) = o;
        } else {
            const int kv = rows[m].kv_index, page = kv / kPageTokens, off = kv % kPageTokens, h = c >> 6, e = c & 63;
            f16x4 o = {(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3]};
            f16 *base = part == 1 ? kc : vc;
            *(f16x4 *)(base + (((long)page * H + h) * kPageTokens + off) * 64 + e) = o;
        }
    }
};

struct EpiPartial {        // split-K partial sums: ws[split][m][n] = acc (no bias); the consumer kernel adds the splits in order
    static constexpr bool kWide = false;
    static constexpr bool kSplitK = true;
    float *ws; long ld; long split_stride; int split;
    __device__ __forceinline__ void operator()(int m, int n, f32x4 a) const { *(f32x4 *)(ws + split * split_stride + (long)m * ld + n) = a; }
};

template <class Epi, class = void> struct has_col_bias { static constexpr bool value = false; };
template <class Epi> struct has_col_bias<Epi, decltype((void)Epi::kColBias)> { static constexpr bool value = Epi::kColBias; };
template <class Epi, class = void> struct is_splitk { static constexpr bool value = false; };
template <class Epi> struct is_splitk<Epi, decltype((void)Epi::kSplitK)> { static constexpr bool value = Epi::kSplitK; };
// Epilogues that map a row (or column) index to (utterance, frame) offer begin_tile(m0, n0): one integer division per
// tile there instead of one per store (a 256-wide tile spans at most two utterances, so a compare finishes the job;
// the compiler's division sequence is ~30 instructions, and the staged epilogue makes 16-32 stores per lane).
template <class Epi, class = void> struct epi_modes { static constexpr int value = 1; };
template <class Epi> struct epi_modes<Epi, decltype((void)Epi::kModes)> { static constexpr int value = Epi::kModes; };
template <class Epi, class = void> struct has_begin_tile { static constexpr bool value = false; };
template <class Epi> struct has_begin_tile<Epi, decltype((void)&Epi::begin_tile)> { static constexpr bool value = true; };
template <class Epi, class = void> struct reads_output { static constexpr bool value = false; };
template <class Epi> struct reads_output<Epi, decltype((void)Epi::kReadsOutput)> { static constexpr bool value = Epi::kReadsOutput; };

// ------------------------------------------------------------------------------------------ kernel
// Workgroup id -> output tile. Physical ids p, p+8, ... share an XCD (one private L2 each): every XCD
// gets a contiguous run of logical ids, and logical ids walk the tile grid in groups of 4-6 tile rows x
// all tile columns, column-major inside a group, so the ~32 tiles an XCD works on at one time form a
// patch that shares a few activation panels and weight panels through that L2.
#ifdef WMI_HARNESS
__device__ unsigned long long g_gemm_stamps[2048 * 8 * 8];      // cycle stamps of the stamped harness variants (see gemm_variant)
int g_tile_group_rows = 0;     // harness knob: tile rows per L2 group; 0 = chosen from the tile-column count
#else
// the product library has no stamped kernels and no knob: the stamp branches below are compiled out (ABL is 0 everywhere)
static __device__ unsigned long long g_gemm_stamps[8];
static constexpr int g_tile_group_rows = 0;
#endif

__device__ __forceinline__ void tile_coords(int p, int nbm, int nbn, int GM, int &bm, int &bn) {
    const int nblk = nbm * nbn;
    const int q = nblk >> 3, r = nblk & 7, x = p & 7, i = p >> 3;
    const int l = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    const int gsz = GM * nbn, gid = l / gsz, first = gid * GM, rows = min(nbm - first, GM), w = l - gid * gsz;
    bn = w / rows;
    bm = first + (w - bn * rows);
}

// Output of one wave's TM x TN grid of 16x16 accumulator tiles. A lane holds row (lane & 15), columns
// 4*(lane >> 4)..+3 of each tile. For f16 row-major destinations two vertically adjacent tiles X, Y are
// exchanged between lane pairs with v_permlane16_swap (odd 16-lane rows of one register <-> even rows of
// the other): afterwards a lane holds 8 consecutive columns (8*(lane >> 5)..+7) of one row of X (even
// 16-lane row) or Y (odd row) and stores them as one 16-byte vector -- half the store instructions, and the
// store tail of a tile is bound by their issue rate.
template <int TM, int TN, class Epi>
__device__ __forceinline__ void write_tile(const Epi &epi, f32x4 (&acc)[TN][TM], int M, int mw, int nw, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
    if constexpr (Epi::kWide && TM % 2 == 0) {
#pragma unroll
        for (int i = 0; i < TN; i++) {
            const int n = nw + i * 16;
            if (epi.wide(n)) {
#pragma unroll
                for (int j = 0; j < TM; j += 2) {
                    const int mx = mw + j * 16 + fr, my = mx + 16;
                    union { f16x4 h; unsigned u[2]; } x, y;
                    x.h = epi.value(mx, n + fq * 4, acc[i][j]);
                    y.h = epi.value(my, n + fq * 4, acc[i][j + 1]);
                    const auto s0 = __builtin_amdgcn_permlane16_swap(x.u[0], y.u[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(x.u[1], y.u[1], false, false);
                    union { f16x8 h; unsigned u[4]; } o;
                    o.u[0] = s0[0]; o.u[1] = s1[0]; o.u[2] = s0[1]; o.u[3] = s1[1];
                    const int m = (fq & 1) ? my : mx;
                    if (m < M) epi.store8(m, n + (fq >> 1) * 8, o.h);
                }
            } else {
#pragma unroll
                for (int j = 0; j < TM; j++) {
                    const int m = mw + j * 16 + fr;
                    if (m < M) epi(m, n + fq * 4, acc[i][j]);
                }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < TN; i++)
#pragma unroll
            for (int j = 0; j < TM; j++) {
                const int m = mw + j * 16 + fr;
                if (m < M) epi(m, nw + i * 16 + fq * 4, acc[i][j]);
            }
    }
}

// Output of a whole 256 x 256 workgroup tile through LDS (the 128 KB the K loop no longer needs): every wave
// converts its accumulators to f16 (pairing lanes as in write_tile, so LDS writes are 16 bytes per lane), the tile
// is laid out [256 rows][512 B] with 16-byte chunk c of row r at chunk c ^ (r & 7) (conflict-free for the
// 8-lane write groups and the 16-lane read groups), and is then stored to memory as whole rows: one wave
// instruction = 2 rows x 512 contiguous bytes. Direct stores from the accumulator layout touch 32 bytes per
// row and instruction, which the memory side counts (and moves) as half-used 64-byte requests.
// Which rows / columns of the 256 x 256 workgroup tile a wave's accumulator tiles acc[i][j] (i: 4 column tiles, j: 8 row
// tiles of 16) cover. MapBlock: one 128 x 64 block per wave (2 x 4 waves). MapSplit: four 64 x 32 blocks, one in each
// quarter of the tile -- the layout of the 8-phase kernel, whose waves take their operands from all four half-tiles.
struct MapBlock {
    int wr, wc;
    __device__ __forceinline__ int row(int j) const { return wr * 128 + j * 16; }
    __device__ __forceinline__ int col(int i) const { return wc * 64 + i * 16; }
};
struct MapSplit {
    int wr, wc;
    __device__ __forceinline__ int row(int j) const { return (j >> 2) * 128 + wr * 64 + (j & 3) * 16; }
    __device__ __forceinline__ int col(int i) const { return (i >> 1) * 128 + wc * 32 + (i & 1) * 16; }
};

template <class Epi, int XB = 0, class Map = MapBlock>          // XB (harness): 1 = no global stores, 2 = no LDS round trip (stores of the raw pairs, wrong values)
__device__ __forceinline__ void write_tile_staged(const Epi &epi, f32x4 (&acc)[4][8], int M, int m0, int n0, Map map, int wave, int lane, char *smem) {
    const int fr = lane & 15, fq = lane >> 4;
    // a lane's four column groups keep their bias for all eight row tiles: four loads, requested before the barrier
    f32x4 cb[4];
    if constexpr (has_col_bias<Epi>::value) {
#pragma unroll
        for (int i = 0; i < 4; i++) cb[i] = epi.col_bias(n0 + map.col(i) + fq * 4);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): my last fragment reads are done
    __builtin_amdgcn_s_barrier();                // ... and everybody else's: the stage buffers are free
    auto convert = [&](auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const int rx = map.row(j) + fr, ry = rx + 16, cn = map.col(i);
                union { f16x4 h; unsigned u[2]; } x, y;
                if constexpr (has_col_bias<Epi>::value) {
                    x.h = epi.template value_b<MODE>(m0 + rx, n0 + cn + fq * 4, acc[i][j], cb[i]);
                    y.h = epi.template value_b<MODE>(m0 + ry, n0 + cn + fq * 4, acc[i][j + 1], cb[i]);
                } else {
                    x.h = epi.value(m0 + rx, n0 + cn + fq * 4, acc[i][j]);
                    y.h = epi.value(m0 + ry, n0 + cn + fq * 4, acc[i][j + 1]);
                }
                const auto s0 = __builtin_amdgcn_permlane16_swap(x.u[0], y.u[0], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(x.u[1], y.u[1], false, false);
                union { f16x8 h; unsigned u[4]; } o;
                o.u[0] = s0[0]; o.u[1] = s1[0]; o.u[2] = s0[1]; o.u[3] = s1[1];
                const int row = (fq & 1) ? ry : rx, chunk = (cn >> 3) + (fq >> 1);
                if (XB & 2) { const int p = i * 4 + j / 2, r2 = p * 16 + wave * 2 + (lane >> 5); if (m0 + r2 < M) epi.store8(m0 + r2, n0 + (lane & 31) * 8, o.h); }
                else *(f16x8 *)(smem + row * 512 + ((chunk ^ (row & 7)) << 4)) = o.h;
            }
    };
    if constexpr (has_col_bias<Epi>::value) {
        if constexpr (Epi::kModes == 4) {
            switch (epi.mode()) {                // wave-uniform: one straight-line body per mode
                case 0: convert(std::integral_constant<int, 0>{}); break;
                case 1: convert(std::integral_constant<int, 1>{}); break;
                case 2: convert(std::integral_constant<int, 2>{}); break;
                default: convert(std::integral_constant<int, 3>{}); break;
            }
        } else convert(std::integral_constant<int, 0>{});
    } else convert(std::integral_constant<int, 0>{});
    if (XB & 2) return;
    __syncthreads();
    if (m0 + 256 <= M) {                         // no row of the tile lies past M: straight-line stores
#pragma unroll
        for (int p = 0; p < 16; p++) {
            const int row = p * 16 + wave * 2 + (lane >> 5), chunk = lane & 31;
            const f16x8 v = *(const f16x8 *)(smem + row * 512 + ((chunk ^ (row & 7)) << 4));
            if (XB & 1) asm volatile("" ::"v"(v));
            else epi.store8(m0 + row, n0 + chunk * 8, v);
        }
        return;
    }
#pragma unroll
    for (int p = 0; p < 16; p++) {
        const int row = p * 16 + wave * 2 + (lane >> 5), chunk = lane & 31;
        const f16x8 v = *(const f16x8 *)(smem + row * 512 + ((chunk ^ (row & 7)) << 4));
        if (XB & 1) asm volatile("" ::"v"(v));
        else if (m0 + row < M) epi.store8(m0 + row, n0 + chunk * 8, v);
    }
}

// The same for epilogues that work on fp32 values (residual add, fp32 output, ...): the raw accumulators of one
// 128-row half at a time (128 x 256 x 4 B = 128 KB) go through LDS and come back one whole row per wave
// instruction, so the epilogue's loads and stores are 1 KB contiguous instead of 16 rows x 64 B.
// An epilogue that adds to what the output already holds has its 16 rows per wave and half requested before the
// accumulators go to LDS; the barriers here are the raw s_barrier after an LDS-only wait, so those requests stay in
// flight across them (a __syncthreads() waits for every outstanding load).
template <class Epi, class Map = MapBlock>
__device__ __forceinline__ void write_tile_staged_f32(const Epi &epi, f32x4 (&acc)[4][8], int M, int m0, int n0, Map map, int wave, int lane, char *smem) {
    const int fr = lane & 15, fq = lane >> 4;
    constexpr bool RMW = reads_output<Epi>::value;
    f32x4 old[RMW ? 16 : 1];
    auto request = [&](int hh) {
        if constexpr (RMW) {
#pragma unroll
            for (int p = 0; p < 16; p++) {
                const int m = m0 + hh * 128 + p * 8 + wave;
                old[p] = f32x4{0, 0, 0, 0};
                if (m < M) old[p] = epi.load(m, n0 + lane * 4);
            }
        }
    };
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    request(0);
    lds_barrier();                               // everybody's last fragment reads are done: the stage buffers are free
#pragma unroll
    for (int hh = 0; hh < 2; hh++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if ((map.row(j) >> 7) != hh) continue;           // this row tile lies in the other 128-row half
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int row = (map.row(j) & 127) + fr, chunk = (map.col(i) >> 2) + fq;
                *(f32x4 *)(smem + row * 1024 + ((chunk ^ (row & 7)) << 4)) = acc[i][j];
            }
        }
        lds_barrier();
#pragma unroll
        for (int p = 0; p < 16; p++) {
            const int row = p * 8 + wave, m = m0 + hh * 128 + row;
            const f32x4 v = *(const f32x4 *)(smem + row * 1024 + ((lane ^ (row & 7)) << 4));
            if (m < M) {
                if constexpr (RMW) epi.apply(m, n0 + lane * 4, v, old[p]);
                else epi(m, n0 + lane * 4, v);
            }
        }
        if (hh == 0) { request(1); lds_barrier(); }
    }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// Wait until at most min(younger, MAXS) stages of LPS LDS-DMA instructions each are still outstanding.
template <int MAXS, int LPS> __device__ __forceinline__ void wait_stages(int younger) {
    static_assert(MAXS * LPS <= 63, "vmcnt is a 6-bit count");
    if constexpr (MAXS <= 0) wait_vmcnt<0>();
    else { if (younger >= MAXS) wait_vmcnt<MAXS * LPS>(); else wait_stages<MAXS - 1, LPS>(younger); }
}

// BM x BN output tile, WM x WN waves (each owns a (BM/WM) x (BN/WN) sub-tile), NST LDS stages.
// Stage s+NST-1 is issued while stage s is multiplied, so NST-2 whole stages stay in flight across
// each barrier: the wait before the barrier is a COUNTED vmcnt (never 0 in the steady state) and the
// barrier is the raw s_barrier (a __syncthreads() would drain the LDS-DMA queue).
// MODE bits. Schedule (shipped on the 256x256 tile): kSpread issues the next stage's LDS-DMA between the MFMA
// groups instead of in one burst after the barrier; kFirstHalf confines that to the first k-half of the step,
// so the loads get the second half's MFMAs as extra time to land. Harness-only builds: kNoLoads / kNoLdsReads /
// kNoMfma drop one ingredient of the loop (ablation timing), kStamps records s_memtime per loop section.
enum : int { kNoLoads = 1, kNoLdsReads = 2, kNoMfma = 4, kSpread = 8, kStamps = 16, kFirstHalf = 32, kSameTile = 64, kBufferLoad = 128, kNoStagger = 256, kIssueFirst = 512, kVSplit = 1024, kOneTile = 2048 };   // kSameTile: every workgroup loads tile (0,0): all L2 hits
// kVSplit (product, 64x64 tile): the K loop is cut into slices of `vs` steps whose sums are added in slice order -- the very
// additions the split-K path makes (one partial per slice, summed by its consumer), so the two paths agree bit for bit.

template <int BM, int BN, int WM, int WN, int NST, class Epi, int ABL = 0>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nt_kernel(const f16 *__restrict__ A, long lda, const f16 *__restrict__ W, long ldw,
                                                                int M, int N, int K, int gm, Epi epi, int vs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = WM * WN;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;     // 16x16 tiles per wave along m and n
    constexpr int LA = BM / 8 / NW, LB = BN / 8 / NW, LPS = LA + LB;   // LDS-DMA instructions per wave per stage
    static_assert(LA >= 1 && LB >= 1, "tile too small for the wave count");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: LDS-DMA destinations (M0) and tile offsets need no VALU
    int bm, bn;
    tile_coords(blockIdx.x, (M + BM - 1) / BM, N / BN, gm, bm, bn);
    const int m0 = bm * BM, n0 = bn * BN;
    const int wr = wave / WN, wc = wave % WN;
    if constexpr (is_splitk<Epi>::value) {       // blockIdx.y = K-slice; K is the slice length
        A += (long)blockIdx.y * K; W += (long)blockIdx.y * K;
        epi.split = blockIdx.y;
    }

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; i++)
#pragma unroll
        for (int j = 0; j < TM; j++) acc[i][j] = f32x4{0, 0, 0, 0};
    constexpr bool VS = (ABL & kVSplit) != 0;
    f32x4 vsum[VS ? TN : 1][VS ? TM : 1];
    int vs_left = vs;                  // K-steps to the end of the current slice
    bool vs_first = true;

    // --- staging: one LDS-DMA wave-instruction moves 8 rows x 128 B
    const int srow = lane >> 3, sslot = lane & 7;
    const f16 *a_src = A + (long)(((ABL & kSameTile) ? 0 : m0) + wave * 8 + srow) * lda + ((sslot ^ srow) << 3);      // (row & 7) == srow for every piece
    const f16 *w_src = W + (long)(((ABL & kSameTile) ? 0 : n0) + wave * 8 + srow) * ldw + ((sslot ^ srow) << 3);
    auto stage = [&](int kt, int buf) {
        char *base = smem + buf * STAGE + wave * 1024;
#pragma unroll
        for (int i = 0; i < LA; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a_src + (long)i * NW * 8 * lda + (long)kt * BK),
                                             (__attribute__((address_space(3))) void *)(base + i * NW * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < LB; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(w_src + (long)i * NW * 8 * ldw + (long)kt * BK),
                                             (__attribute__((address_space(3))) void *)(base + A_BYTES + i * NW * 1024), 16, 0, 0);
    };

    // kBufferLoad (harness): the same pieces through buffer_load ... lds -- 32-bit lane offsets against a per-workgroup
    // descriptor, the piece and K-step offsets in an SGPR, so no 64-bit vector address arithmetic per piece
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void *)(A + (long)((ABL & kSameTile) ? 0 : m0) * lda), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)(W + (long)((ABL & kSameTile) ? 0 : n0) * ldw), 0, 0x7fffffff, 0x00020000);
    const int voff_a = (int)(((long)(wave * 8 + srow) * lda + ((sslot ^ srow) << 3)) * 2), voff_w = (int)(((long)(wave * 8 + srow) * ldw + ((sslot ^ srow) << 3)) * 2);
    auto stage_piece = [&](int kt, int buf, int i, bool is_a) {
        char *base = smem + buf * STAGE + wave * 1024;
        if (ABL & kBufferLoad) {
            if (is_a) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void *)(base + i * NW * 1024), 16, voff_a, (int)(((long)i * NW * 8 * lda + (long)kt * BK) * 2), 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void *)(base + A_BYTES + i * NW * 1024), 16, voff_w, (int)(((long)i * NW * 8 * ldw + (long)kt * BK) * 2), 0, 0);
            return;
        }
        if (is_a)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a_src + (long)i * NW * 8 * lda + (long)kt * BK),
                                             (__attribute__((address_space(3))) void *)(base + i * NW * 1024), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(w_src + (long)i * NW * 8 * ldw + (long)kt * BK),
                                             (__attribute__((address_space(3))) void *)(base + A_BYTES + i * NW * 1024), 16, 0, 0);
    };
    const int nk = K / BK;
#pragma unroll
    for (int s = 0; s < NST - 1; s++)
        if (s < nk) stage(s, s);
    const int fr = lane & 15, fq = lane >> 4;
    int buf = 0, nbuf = NST - 1;       // buffer being multiplied / buffer the next stage goes to
    unsigned long long t_wait = 0, t_bar = 0, t_comp = 0, t_prev = 0, t_start = 0, t_pro = 0;
    if (ABL & 16) t_start = t_prev = __builtin_amdgcn_s_memtime();
    for (int kt = 0; kt < nk; kt++) {
        // stage kt has landed once at most min(NST-2, nk-1-kt) younger stages are still outstanding
        const int younger = nk - 1 - kt;
        wait_stages<NST - 2, LPS>(younger);
        if ((ABL & 16) && kt > 0) { const unsigned long long t = __builtin_amdgcn_s_memtime(); t_wait += t - t_prev; t_prev = t; }
        __builtin_amdgcn_s_barrier();
        if (ABL & 16) { const unsigned long long t = __builtin_amdgcn_s_memtime(); if (kt > 0) t_bar += t - t_prev; else t_pro = t - t_prev; t_prev = t; }
        const bool more = kt + NST - 1 < nk;
        if ((ABL & kNoLoads) == 0 && (ABL & kSpread) == 0 && more) stage(kt + NST - 1, nbuf);
        const char *sa = smem + ((ABL & 1) ? 0 : buf) * STAGE, *sb = sa + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            f16x8 fa[TM], fw[TN];
            if ((ABL & 2) == 0 || kt == 0) {
#pragma unroll
                for (int j = 0; j < TM; j++) {
                    const int row = wr * (BM / WM) + j * 16 + fr;
                    fa[j] = *(const f16x8 *)(sa + row * 128 + (((ks * 4 + fq) ^ (row & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < TN; i++) {
                    const int row = wc * (BN / WN) + i * 16 + fr;
                    fw[i] = *(const f16x8 *)(sb + row * 128 + (((ks * 4 + fq) ^ (row & 7)) << 4));
                }
            }
            if (ABL & 4) {
                if ((ABL & kSpread) && !(ABL & kNoLoads) && more && ks == 0) stage(kt + NST - 1, nbuf);
#pragma unroll
                for (int j = 0; j < TM; j++) asm volatile("" ::"v"(fa[j]));
#pragma unroll
                for (int i = 0; i < TN; i++) asm volatile("" ::"v"(fw[i]));
            } else {
#pragma unroll
                for (int i = 0; i < TN; i++) {
                    if ((ABL & 8) && more && (!(ABL & 32) || ks == 0)) {          // spread the LDS-DMA issue over the MFMA groups
                        constexpr int per = (ABL & 32) ? (LPS + TN - 1) / TN : (LPS + 2 * TN - 1) / (2 * TN);
#pragma unroll
                        for (int u = 0; u < per; u++) {
                            const int li = ((ABL & 32) ? i : ks * TN + i) * per + u;
                            if (li < LA) stage_piece(kt + NST - 1, nbuf, li, true);
                            else if (li < LPS) stage_piece(kt + NST - 1, nbuf, li - LA, false);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int j = 0; j < TM; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[i], fa[j], acc[i][j], 0, 0, 0);
                    if (ABL & 8) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        buf = buf + 1 == NST ? 0 : buf + 1;
        nbuf = nbuf + 1 == NST ? 0 : nbuf + 1;
        if (ABL & 16) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t = __builtin_amdgcn_s_memtime(); t_comp += t - t_prev; t_prev = t; }
        if constexpr (VS) {
            if (--vs_left == 0) {          // end of a slice: fold its sum into the total, in slice order
#pragma unroll
                for (int i = 0; i < TN; i++)
#pragma unroll
                    for (int j = 0; j < TM; j++) {
                        vsum[i][j] = vs_first ? acc[i][j] : vsum[i][j] + acc[i][j];
                        acc[i][j] = f32x4{0, 0, 0, 0};
                    }
                vs_first = false; vs_left = vs;
            }
        }
    }
    if constexpr (VS) {
#pragma unroll
        for (int i = 0; i < TN; i++)
#pragma unroll
            for (int j = 0; j < TM; j++) acc[i][j] = vsum[i][j];
    }
    if constexpr (has_begin_tile<Epi>::value) epi.begin_tile(m0, n0);       // every tile shape is at most 256 x 256
    if constexpr (BM == 256 && BN == 256 && WM == 2 && WN == 4 && Epi::kWide) {
        if (epi.wide_block(n0, BN)) write_tile_staged<Epi, (ABL >> 8) & 3>(epi, acc, M, m0, n0, MapBlock{wr, wc}, wave, lane, smem);
        else write_tile<TM, TN>(epi, acc, M, m0 + wr * (BM / WM), n0 + wc * (BN / WN), lane);
    } else if constexpr (BM == 256 && BN == 256 && WM == 2 && WN == 4 && !is_splitk<Epi>::value) {
        write_tile_staged_f32(epi, acc, M, m0, n0, MapBlock{wr, wc}, wave, lane, smem);
    } else {
        write_tile<TM, TN>(epi, acc, M, m0 + wr * (BM / WM), n0 + wc * (BN / WN), lane);
    }
    if ((ABL & 16) && (int)blockIdx.x >= max(0, (int)gridDim.x - 2048)) {       // the last 2048 workgroups: the steady state of a long launch
        const unsigned long long t_issue = __builtin_amdgcn_s_memtime();
        wait_vmcnt<0>();
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            unsigned long long *o = g_gemm_stamps + ((long)((int)blockIdx.x - max(0, (int)gridDim.x - 2048)) * 8 + (wave & 7)) * 8;
            o[0] = t_wait; o[1] = t_bar; o[2] = t_comp; o[3] = nk; o[4] = t_start; o[5] = t_pro; o[6] = t_issue - t_prev; o[7] = t_end;
        }
    }
}

// ------------------------------------------------------------------------------------------ 8-phase kernel
// 256 x 256 output tile, 8 waves, K in steps of 64 -- the same tile as above with the K-step cut into four phases of 16 MFMAs
// and the two waves of each SIMD running half a phase apart, so that while one multiplies, the other reads its fragments
// and issues its share of the LDS-DMA; the matrix pipe of a SIMD then sees one wave's MFMAs after the other's, back to back.
//   * Each operand tile of a K-step is two half-tiles of 128 rows x 128 B (activation rows 0-127 "AL" / 128-255 "AH", weight
//     rows "BL" / "BH"); LDS holds 8 half-tile slots (two K-steps). A wave owns a 64 x 32 block in each quarter of the
//     output tile: rows wr*64.. of AL and of AH, columns wc*32.. of BL and of BH. Per K-step it reads BL + AL fragments
//     (phase 0), BH (1), AH (2), nothing (3; the BL fragments are still in registers) and multiplies one quarter per phase.
//     A slot is therefore free again two phases after the phase that read it, long before the K-step is over.
//   * The stream of half-tiles AL, BL, BH, AH of K-step 0, 1, ... is issued one per phase, six ahead: phase p issues index
//     p + 6 into the slot whose last reader was phase <= p + 4 - 6 ... i.e. at least two phases (four barriers) back.
//   * After issuing, a wave waits for all but its 8 youngest LDS-DMA instructions (4 half-tiles in flight): everything up
//     to index p + 2 has landed, which is all phase p + 1 reads. The barrier between the wait and the read publishes the
//     other waves' pieces. Counted vmcnt, raw s_barrier: nothing in the loop drains the queue.
//   * Waves 4-7 start one barrier late and waves 0-3 pass one extra barrier at the end; every barrier is still passed
//     by all eight waves.
// Accumulation order along K is the same as in gemm_nt_kernel: results are bit-identical.
template <class Epi, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_nt8_kernel(const f16 *__restrict__ A, long lda, const f16 *__restrict__ W, long ldw,
                                                       int M, int N, int K, int gm, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SLOT = 128 * 128;                        // bytes of one half-tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    int bm, bn;
    tile_coords(blockIdx.x, (M + 255) / 256, N / 256, gm, bm, bn);
    const int m0 = bm * 256, n0 = bn * 256;

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = f32x4{0, 0, 0, 0};

    // staging: half-tile = 16 pieces of 8 rows x 128 B; wave w moves pieces w and w + 8 (rows 8w.. and 64 + 8w..)
    const int srow = lane >> 3, sslot = lane & 7;
    const f16 *a_src = A + (long)(m0 + wave * 8 + srow) * lda + ((sslot ^ srow) << 3);
    const f16 *w_src = W + (long)(n0 + wave * 8 + srow) * ldw + ((sslot ^ srow) << 3);
    const int nk = K / BK, total = 4 * nk;
    auto issue = [&](int q) {          // q = 4 * K-step + {0 AL, 1 BL, 2 BH, 3 AH}
        const int kt = q >> 2, h = q & 3;
        char *dst = smem + (((kt & 1) << 2) + h) * SLOT + wave * 1024;
        const bool is_a = h == 0 || h == 3;
        const f16 *src = (is_a ? a_src : w_src) + (long)(h >= 2 ? 128 : 0) * (is_a ? lda : ldw) + (long)kt * BK;
        const long step = 64 * (is_a ? lda : ldw);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + step), (__attribute__((address_space(3))) void *)(dst + 8192), 16, 0, 0);
    };
    // all but the n youngest LDS-DMA instructions of this wave have landed
    auto wait_dma = [&](int halves_in_flight) {
        if (halves_in_flight >= 4) wait_vmcnt<8>();
        else if (halves_in_flight == 3) wait_vmcnt<6>();
        else if (halves_in_flight == 2) wait_vmcnt<4>();
        else if (halves_in_flight == 1) wait_vmcnt<2>();
        else wait_vmcnt<0>();
    };
    const int fr = lane & 15, fq = lane >> 4;
    // fragment addresses inside a slot: row (wave block + tile + fr), 16-byte chunk (ks * 4 + fq) ^ (row & 7); row & 7 == fr & 7
    const int a_off = (wr * 64 + fr) * 128, b_off = (wc * 32 + fr) * 128;
    const int c0 = ((fq) ^ (fr & 7)) << 4, c1 = ((4 + fq) ^ (fr & 7)) << 4;
    f16x8 fa[4][2], fbl[2][2], fbh[2][2];
    auto read_a = [&](const char *slot) {
#pragma unroll
        for (int j = 0; j < 4; j++) { fa[j][0] = *(const f16x8 *)(slot + a_off + j * 2048 + c0); fa[j][1] = *(const f16x8 *)(slot + a_off + j * 2048 + c1); }
    };
    auto read_b = [&](const char *slot, f16x8 (&fb)[2][2]) {
#pragma unroll
        for (int i = 0; i < 2; i++) { fb[i][0] = *(const f16x8 *)(slot + b_off + i * 2048 + c0); fb[i][1] = *(const f16x8 *)(slot + b_off + i * 2048 + c1); }
    };
    auto mma = [&](const f16x8 (&fb)[2][2], int ib, int jb) {      // quarter (column half ib, row half jb)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[ib * 2 + i][jb * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[i][ks], fa[j][ks], acc[ib * 2 + i][jb * 4 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // prologue: six half-tiles in flight, the first two landed
#pragma unroll
    for (int q = 0; q < 6; q++)
        if (q < total) issue(q);
    wait_dma(min(4, total - 2));
    __builtin_amdgcn_s_barrier();
    if (wr == 1 && !(ABL & kNoStagger)) __builtin_amdgcn_s_barrier();  // waves 4-7 run one barrier behind

    // ABL & kStamps (harness): cycles per section, summed over the loop: [0] read segment up to the wait, [1] the DMA wait,
    // [2] first barrier, [3] multiply segment, [4] second barrier
    unsigned long long tsec[5] = {0, 0, 0, 0, 0}, tprev = 0;
    auto stamp = [&](int sec) {
        if (ABL & kStamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tsec[sec] += t - tprev; tprev = t; }
    };
    if (ABL & kStamps) tprev = __builtin_amdgcn_s_memtime();
    for (int kt = 0; kt < nk; kt++) {
        const char *sb = smem + ((kt & 1) << 2) * SLOT;            // slots AL, BL, BH, AH of this K-step
        const int p = 4 * kt;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            // ---- read segment
            if ((ABL & kIssueFirst) && p + r + 6 < total) { issue(p + r + 6); __builtin_amdgcn_sched_barrier(0); }
            if (r == 0) { read_b(sb + SLOT, fbl); __builtin_amdgcn_sched_barrier(0); read_a(sb); }
            else if (r == 1) read_b(sb + 2 * SLOT, fbh);
            else if (r == 2) read_a(sb + 3 * SLOT);
            if (!(ABL & kIssueFirst) && p + r + 6 < total) issue(p + r + 6);
            stamp(0);
            wait_dma(total - 3 - (p + r));                          // indices <= p + r + 2 have landed
            stamp(1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            stamp(2);
            // ---- multiply segment
            if (r == 0) mma(fbl, 0, 0);
            else if (r == 1) mma(fbh, 1, 0);
            else if (r == 2) mma(fbh, 1, 1);
            else mma(fbl, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            stamp(3);
            __builtin_amdgcn_s_barrier();
            stamp(4);
        }
    }
    if (wr == 0 && !(ABL & kNoStagger)) __builtin_amdgcn_s_barrier();  // the barrier waves 4-7 passed at the start
    if ((ABL & kStamps) && lane == 0 && (int)blockIdx.x >= max(0, (int)gridDim.x - 2048)) {
        unsigned long long *o = g_gemm_stamps + ((long)((int)blockIdx.x - max(0, (int)gridDim.x - 2048)) * 8 + wave) * 8;
        o[0] = tsec[0]; o[1] = tsec[1]; o[2] = tsec[2]; o[3] = nk; o[4] = tsec[3]; o[5] = tsec[4]; o[6] = 0; o[7] = 0;
    }

    if constexpr (has_begin_tile<Epi>::value) epi.begin_tile(m0, n0);
    const MapSplit map{wr, wc};
    if constexpr (Epi::kWide) {
        if (epi.wide_block(n0, 256)) { write_tile_staged<Epi, 0, MapSplit>(epi, acc, M, m0, n0, map, wave, lane, smem); return; }
    } else if constexpr (!is_splitk<Epi>::value) {
        write_tile_staged_f32<Epi, MapSplit>(epi, acc, M, m0, n0, map, wave, lane, smem);
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int m = m0 + map.row(j) + fr;
            if (m < M) epi(m, n0 + map.col(i) + fq * 4, acc[i][j]);
        }
}

// ------------------------------------------------------------------------------------------ persistent 8-phase kernel
// The same loop with one workgroup per CU walking a list of tiles (tile ids blockIdx.x, + gridDim.x, ...: the ids one CU
// would have been handed by a grid of one workgroup per tile, so the L2 grouping of tile_coords is unchanged). What it buys:
//   * the stream of half-tiles does not stop at a tile boundary. K/64 is even, so the next tile's K-step 0 has the slot
//     parity the ring expects, and the last six phases of a tile issue the next tile's first six half-tiles into slots
//     0-5 exactly as they would issue the current tile's; the counted wait stays at four half-tiles in flight. A tile
//     therefore has no prologue (no cold first loads), and no workgroup has to retire, drain its stores and be
//     dispatched again between two tiles;
//   * those loads land while the epilogue runs. The epilogue stages through the 64 KB that are free meanwhile -- ring
//     slots 6 and 7 (the last K-step's BH / AH, read two and one phases before the loop ends) plus 32 KB above the ring
//     (LDS is 160 KB) -- in two 128-row halves (f16 results) or four 128 x 128 quarters (fp32 results), with the next
//     half's conversions (bias, GELU) computed before the barrier that frees the staging area;
//   * a barrier closes the epilogue: the next tile's phase 0 issues into slot 6.
// Accumulation order, epilogue arithmetic and results are those of gemm_nt8_kernel.
template <class Epi, int MODE>
__device__ __forceinline__ void write_tile_halves(const Epi &epi, f32x4 (&acc)[4][8], int M, int m0, int n0, MapSplit map, int wave, int lane, char *stg) {
    const int fr = lane & 15, fq = lane >> 4;
    f32x4 cb[4];
    if constexpr (has_col_bias<Epi>::value) {
#pragma unroll
        for (int i = 0; i < 4; i++) cb[i] = epi.col_bias(n0 + map.col(i) + fq * 4);
    }
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    const bool full = m0 + 256 <= M;
    f16x8 o[8];
    auto convert = [&](int hh) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int jj = 0; jj < 4; jj += 2) {
                const int j = hh * 4 + jj, rx = map.row(j) + fr, ry = rx + 16, cn = map.col(i);
                union { f16x4 h; unsigned u[2]; } x, y;
                if constexpr (has_col_bias<Epi>::value) {
                    x.h = epi.template value_b<MODE>(m0 + rx, n0 + cn + fq * 4, acc[i][j], cb[i]);
                    y.h = epi.template value_b<MODE>(m0 + ry, n0 + cn + fq * 4, acc[i][j + 1], cb[i]);
                } else {
                    x.h = epi.value(m0 + rx, n0 + cn + fq * 4, acc[i][j]);
                    y.h = epi.value(m0 + ry, n0 + cn + fq * 4, acc[i][j + 1]);
                }
                const auto s0 = __builtin_amdgcn_permlane16_swap(x.u[0], y.u[0], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(x.u[1], y.u[1], false, false);
                union { f16x8 h; unsigned u[4]; } v;
                v.u[0] = s0[0]; v.u[1] = s1[0]; v.u[2] = s0[1]; v.u[3] = s1[1];
                o[i * 2 + (jj >> 1)] = v.h;
            }
    };
    auto put = [&](int hh) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int jj = 0; jj < 4; jj += 2) {
                const int rx = (map.row(hh * 4 + jj) & 127) + fr, row = (fq & 1) ? rx + 16 : rx, chunk = (map.col(i) >> 3) + (fq >> 1);
                *(f16x8 *)(stg + row * 512 + ((chunk ^ (row & 7)) << 4)) = o[i * 2 + (jj >> 1)];
            }
    };
    auto drain = [&](int hh) {
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const int row = p * 16 + wave * 2 + (lane >> 5), chunk = lane & 31, m = m0 + hh * 128 + row;
            const f16x8 v = *(const f16x8 *)(stg + row * 512 + ((chunk ^ (row & 7)) << 4));
            if (full || m < M) epi.store8(m, n0 + chunk * 8, v);
        }
    };
    convert(0);
    lds_barrier();               // every wave has read its last fragments: slots 6 and 7 are free
    put(0);
    lds_barrier();
    drain(0);
    convert(1);
    lds_barrier();               // half 0 has left the staging area
    put(1);
    lds_barrier();
    drain(1);
    lds_barrier();               // ... and half 1: the next tile may issue into slot 6
}

template <class Epi>
__device__ __forceinline__ void write_tile_quarters_f32(const Epi &epi, f32x4 (&acc)[4][8], int M, int m0, int n0, MapSplit map, int wave, int lane, char *stg) {
    const int fr = lane & 15, fq = lane >> 4;
    constexpr bool RMW = reads_output<Epi>::value;
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    f32x4 old[RMW ? 8 : 1];
    auto request = [&](int pass) {        // quarter `pass`: rows (pass >> 1) * 128.., columns (pass & 1) * 128..
        if constexpr (RMW) {
#pragma unroll
            for (int p = 0; p < 8; p++) {
                const int m = m0 + (pass >> 1) * 128 + p * 16 + wave * 2 + (lane >> 5);
                old[p] = f32x4{0, 0, 0, 0};
                if (m < M) old[p] = epi.load(m, n0 + (pass & 1) * 128 + (lane & 31) * 4);
            }
        }
    };
    auto put = [&](int pass) {
        const int hh = pass >> 1, ci = pass & 1;
#pragma unroll
        for (int ii = 0; ii < 2; ii++)
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                const int row = (map.row(hh * 4 + jj) & 127) + fr, chunk = ((map.col(ci * 2 + ii) & 127) >> 2) + fq;
                *(f32x4 *)(stg + row * 512 + ((chunk ^ (row & 7)) << 4)) = acc[ci * 2 + ii][hh * 4 + jj];
            }
    };
    auto drain = [&](int pass) {
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const int row = p * 16 + wave * 2 + (lane >> 5), m = m0 + (pass >> 1) * 128 + row, n = n0 + (pass & 1) * 128 + (lane & 31) * 4;
            const f32x4 v = *(const f32x4 *)(stg + row * 512 + (((lane & 31) ^ (row & 7)) << 4));
            if (m < M) {
                if constexpr (RMW) epi.apply(m, n, v, old[p]);
                else epi(m, n, v);
            }
        }
    };
    // a quarter's old values are requested as soon as the previous quarter has used the registers: they travel while that
    // quarter leaves the staging area and this one enters it
    request(0);
    lds_barrier();               // slots 6 and 7 are free
#pragma unroll
    for (int pass = 0; pass < 4; pass++) {
        put(pass);
        lds_barrier();
        drain(pass);
        if (pass < 3) request(pass + 1);
        lds_barrier();           // after the last quarter: the next tile may issue into slot 6
    }
}

template <class Epi, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_nt8p_kernel(const f16 *__restrict__ A, long lda, const f16 *__restrict__ W, long ldw,
                                                        int M, int N, int K, int gm, Epi epi0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SLOT = 128 * 128;                        // bytes of one half-tile
    char *const stg = smem + 6 * SLOT;                     // the epilogue's 64 KB: slots 6, 7 and the 32 KB above the ring
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int nbm = (M + 255) / 256, nbn = N / 256, nblk = nbm * nbn;
    const int nk = K / BK, total = 4 * nk;                 // nk is even (checked by the launcher)
    // staging: half-tile = 16 pieces of 8 rows x 128 B; wave w moves pieces w and w + 8 (rows 8w.. and 64 + 8w..), through
    // buffer_load ... lds: the tile's first row is the descriptor's base, the half-tile and K-step a scalar offset, and a lane
    // contributes one 32-bit byte offset per operand -- no 64-bit vector arithmetic in the loop and two address registers in all
    // (with flat global addresses the compiler keeps eight 64-bit pointers alive beside the 128 accumulators, and spills).
    const int srow = lane >> 3, sslot = lane & 7;
    const int a_loff = (int)(((wave * 8 + srow) * lda + ((sslot ^ srow) << 3)) * 2);
    const int w_loff = (int)(((wave * 8 + srow) * ldw + ((sslot ^ srow) << 3)) * 2);
    int pid = blockIdx.x;                                  // the launcher keeps gridDim.x <= nblk
    int bm, bn;
    tile_coords(pid, nbm, nbn, gm, bm, bn);
    auto descriptor = [](const f16 *p) { return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, 0x7fffffff, 0x00020000); };
    __amdgpu_buffer_rsrc_t rs_a = descriptor(A + (long)bm * 256 * lda), rs_w = descriptor(W + (long)bn * 256 * ldw), rs_an = rs_a, rs_wn = rs_w;
    // q = 4 * K-step + {0 AL, 1 BL, 2 BH, 3 AH}; next: of the tile after this one
    auto issue = [&](bool next, int q) {
        const int kt = q >> 2, h = q & 3;
        char *dst = smem + (((kt & 1) << 2) + h) * SLOT + wave * 1024;
        const bool is_a = h == 0 || h == 3;
        const long ld = is_a ? lda : ldw;
        const int soff = (int)(((long)(h >= 2 ? 128 : 0) * ld + (long)kt * BK) * 2);
        const __amdgpu_buffer_rsrc_t rs = is_a ? (next ? rs_an : rs_a) : (next ? rs_wn : rs_w);
        const int loff = is_a ? a_loff : w_loff;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)dst, 16, loff, soff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(dst + 8192), 16, loff, soff + (int)(128 * ld), 0, 0);
    };
    const int fr = lane & 15, fq = lane >> 4;
    const int a_off = (wr * 64 + fr) * 128, b_off = (wc * 32 + fr) * 128;
    const int c0 = ((fq) ^ (fr & 7)) << 4, c1 = ((4 + fq) ^ (fr & 7)) << 4;
    f32x4 acc[4][8];
    f16x8 fa[4][2], fbl[2][2], fbh[2][2];
    auto read_a = [&](const char *slot) {
#pragma unroll
        for (int j = 0; j < 4; j++) { fa[j][0] = *(const f16x8 *)(slot + a_off + j * 2048 + c0); fa[j][1] = *(const f16x8 *)(slot + a_off + j * 2048 + c1); }
    };
    auto read_b = [&](const char *slot, f16x8 (&fb)[2][2]) {
#pragma unroll
        for (int i = 0; i < 2; i++) { fb[i][0] = *(const f16x8 *)(slot + b_off + i * 2048 + c0); fb[i][1] = *(const f16x8 *)(slot + b_off + i * 2048 + c1); }
    };
    auto mma = [&](const f16x8 (&fb)[2][2], int ib, int jb) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[ib * 2 + i][jb * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[i][ks], fa[j][ks], acc[ib * 2 + i][jb * 4 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // One phase. R: which quarter; MODE 0: issue index p + 6 of the current tile, 1: of the next tile (the stream runs on
    // across the tile boundary), 2: nothing left to issue; VM: vector-memory instructions that may stay in flight behind the wait
    // (8 = four half-tiles).
    // Everything that varies from phase to phase is a compile-time constant: the loop body is straight-line code.
    auto phase = [&](auto r_tag, auto mode_tag, auto vm_tag, int kt, const char *sb) {
        constexpr int r = decltype(r_tag)::value, mode = decltype(mode_tag)::value, vm = decltype(vm_tag)::value;
        if constexpr (r == 0) { read_b(sb + SLOT, fbl); __builtin_amdgcn_sched_barrier(0); read_a(sb); }
        else if constexpr (r == 1) read_b(sb + 2 * SLOT, fbh);
        else if constexpr (r == 2) read_a(sb + 3 * SLOT);
        if constexpr (mode == 0) issue(false, 4 * kt + r + 6);
        else if constexpr (mode == 1) issue(true, 4 * kt + r + 6 - total);
        wait_vmcnt<vm>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        if constexpr (r == 0) mma(fbl, 0, 0);
        else if constexpr (r == 1) mma(fbh, 1, 0);
        else if constexpr (r == 2) mma(fbh, 1, 1);
        else mma(fbl, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    // Four half-tiles stay in flight behind every wait. Behind an epilogue the queue also holds that epilogue's stores, younger
    // than half-tiles 2-5 of the new tile and older than what its phases issue; loads complete in order among themselves, so
    // "at most 8 outstanding" still implies that the half-tile a phase is about to read has landed, whatever the stores do.
    using V8 = std::integral_constant<int, 8>;

    // the first tile's prologue: six half-tiles in flight, the first two landed (total >= 8)
#pragma unroll
    for (int q = 0; q < 6; q++) issue(false, q);
    wait_vmcnt<8>();
    __builtin_amdgcn_s_barrier();

    for (;;) {
        const int npid = pid + (int)gridDim.x;
        const bool has_next = npid < nblk;
        int bm2 = bm, bn2 = bn;
        if (has_next) tile_coords(npid, nbm, nbn, gm, bm2, bn2);
        rs_an = descriptor(A + (long)bm2 * 256 * lda); rs_wn = descriptor(W + (long)bn2 * 256 * ldw);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) acc[i][j] = f32x4{0, 0, 0, 0};
        if (wr == 1) __builtin_amdgcn_s_barrier();             // waves 4-7 run one barrier behind

        const char *s0 = smem, *s1 = smem + 4 * SLOT;          // slots AL, BL, BH, AH of the even / odd K-steps
        for (int kt = 0; kt < nk - 2; kt += 2) {               // steady state: index p + 6 is of this tile, four half-tiles stay in flight
            phase(I0{}, I0{}, V8{}, kt, s0); phase(I1{}, I0{}, V8{}, kt, s0); phase(I2{}, I0{}, V8{}, kt, s0); phase(I3{}, I0{}, V8{}, kt, s0);
            phase(I0{}, I0{}, V8{}, kt + 1, s1); phase(I1{}, I0{}, V8{}, kt + 1, s1); phase(I2{}, I0{}, V8{}, kt + 1, s1); phase(I3{}, I0{}, V8{}, kt + 1, s1);
        }
        phase(I0{}, I0{}, V8{}, nk - 2, s0); phase(I1{}, I0{}, V8{}, nk - 2, s0);
        // the last six phases issue the next tile's first six half-tiles (after the last tile: this tile's, once more, so that
        // the code is the same; nobody reads them and the kernel waits for them before it ends)
        phase(I2{}, I1{}, V8{}, nk - 2, s0); phase(I3{}, I1{}, V8{}, nk - 2, s0);
        phase(I0{}, I1{}, V8{}, nk - 1, s1); phase(I1{}, I1{}, V8{}, nk - 1, s1); phase(I2{}, I1{}, V8{}, nk - 1, s1); phase(I3{}, I1{}, V8{}, nk - 1, s1);
        if (wr == 0) __builtin_amdgcn_s_barrier();             // the barrier waves 4-7 passed at the start

        Epi epi = epi0;
        const int m0 = bm * 256, n0 = bn * 256;
        if constexpr (has_begin_tile<Epi>::value) epi.begin_tile(m0, n0);
        // The epilogue's per-lane offsets are recomputed for every tile from opaque copies of the lane and wave numbers: left to
        // itself the compiler computes them once, before the tile loop, and then has to carry some twenty registers through the
        // K loop -- into scratch memory, whose reloads wait behind every load and store in flight.
        int lane_e = lane, wave_e = wave;
        asm volatile("" : "+v"(lane_e));
        asm volatile("" : "+s"(wave_e));
        const int fr = lane_e & 15, fq = lane_e >> 4;
        const MapSplit map{wave_e >> 2, wave_e & 3};
        bool done = false;
        if constexpr (Epi::kWide) {
            if (epi.wide_block(n0, 256)) {
                if constexpr (has_col_bias<Epi>::value && epi_modes<Epi>::value == 4) {
                    switch (epi.mode()) {                          // wave-uniform: one straight-line body per mode
                        case 0: write_tile_halves<Epi, 0>(epi, acc, M, m0, n0, map, wave_e, lane_e, stg); break;
                        case 1: write_tile_halves<Epi, 1>(epi, acc, M, m0, n0, map, wave_e, lane_e, stg); break;
                        case 2: write_tile_halves<Epi, 2>(epi, acc, M, m0, n0, map, wave_e, lane_e, stg); break;
                        default: write_tile_halves<Epi, 3>(epi, acc, M, m0, n0, map, wave_e, lane_e, stg); break;
                    }
                } else write_tile_halves<Epi, 0>(epi, acc, M, m0, n0, map, wave_e, lane_e, stg);
                done = true;
            }
        } else if constexpr (!is_splitk<Epi>::value) {
            write_tile_quarters_f32<Epi>(epi, acc, M, m0, n0, map, wave_e, lane_e, stg);
            done = true;
        }
        if (!done) {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int m = m0 + map.row(j) + fr;
                    if (m < M) epi(m, n0 + map.col(i) + fq * 4, acc[i][j]);
                }
        }
        if (!has_next) break;
        pid = npid; bm = bm2; bn = bn2; rs_a = rs_an; rs_w = rs_wn;
    }
    wait_vmcnt<0>();
}

// CUs of the current device, rounded down to a multiple of 8 (the tile order assumes workgroup ids p and p + 8 share an XCD)
static int persistent_grid() {
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int n = cus[dev].load(std::memory_order_acquire);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) return 0;
        n &= ~7;
        cus[dev].store(n, std::memory_order_release);
    }
    return n;
}
// tiles per workgroup from which the persistent kernel is used
static bool persistent_ok(int M, int N, int K) {
    const int g = persistent_grid();
    return g > 0 && N % 256 == 0 && K % (2 * BK) == 0 && K >= 2 * BK && (long)((M + 255) / 256) * (N / 256) >= 2L * g;
}

template <class Epi, int ABL = 0>
static bool launch8p(const f16 *A, long lda, const f16 *W, long ldw, int M, int N, int K, const Epi &epi, hipStream_t st, int group_rows = 0) {
    if (M <= 0) return true;
    const int g = persistent_grid();
    if (g <= 0 || N % 256 != 0 || K % (2 * BK) != 0 || K < 2 * BK) { log_msg(2, "gemm: unsupported shape M=%d N=%d K=%d (persistent 8-phase tile)\n", M, N, K); return false; }
    const int nblk = ((M + 255) / 256) * (N / 256);
    constexpr size_t lds = 10 * 128 * 128;
    static std::atomic<unsigned long long> attr_set{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (!((attr_set.load(std::memory_order_acquire) >> dev) & 1ull)) {
        if (hipFuncSetAttribute((const void *)gemm_nt8p_kernel<Epi, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
        attr_set.fetch_or(1ull << dev, std::memory_order_release);
    }
    const int nbn = N / 256, gm = group_rows > 0 ? group_rows : g_tile_group_rows > 0 ? g_tile_group_rows : (nbn <= 8 ? 6 : (nbn <= 16 ? 5 : 4));
    hipLaunchKernelGGL((gemm_nt8p_kernel<Epi, ABL>), dim3((ABL & kOneTile) ? nblk : std::min(nblk, g)), dim3(512), lds, st, A, lda, W, ldw, M, N, K, gm, epi);
    return hipGetLastError() == hipSuccess;
}

template <class Epi, int ABL = 0>
static bool launch8(const f16 *A, long lda, const f16 *W, long ldw, int M, int N, int K, const Epi &epi, hipStream_t st, int group_rows = 0) {
    if (M <= 0) return true;
    if (N % 256 != 0 || K % BK != 0 || K < 2 * BK) { log_msg(2, "gemm: unsupported shape M=%d N=%d K=%d (8-phase tile)\n", M, N, K); return false; }
    const int nblk = ((M + 255) / 256) * (N / 256);
    constexpr size_t lds = 8 * 128 * 128;
    static std::atomic<unsigned long long> attr_set{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (!((attr_set.load(std::memory_order_acquire) >> dev) & 1ull)) {
        if (hipFuncSetAttribute((const void *)gemm_nt8_kernel<Epi, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
        attr_set.fetch_or(1ull << dev, std::memory_order_release);
    }
    const int nbn = N / 256, gm = group_rows > 0 ? group_rows : g_tile_group_rows > 0 ? g_tile_group_rows : (nbn <= 8 ? 6 : (nbn <= 16 ? 5 : 4));
    hipLaunchKernelGGL((gemm_nt8_kernel<Epi, ABL>), dim3(nblk), dim3(512), lds, st, A, lda, W, ldw, M, N, K, gm, epi);
    return hipGetLastError() == hipSuccess;
}

template <int BM, int BN, int WM, int WN, int NST, class Epi, int ABL = 0>
static bool launch(const f16 *A, long lda, const f16 *W, long ldw, int M, int N, int K, const Epi &epi, hipStream_t st, int splits = 1, int group_rows = 0, int vs = 0) {
    if (M <= 0) return true;
    if (N % BN != 0 || K % BK != 0) { log_msg(2, "gemm: unsupported shape M=%d N=%d K=%d (tile %dx%d)\n", M, N, K, BM, BN); return false; }
    if ((ABL & kVSplit) && (vs < 1 || (K / BK) % vs != 0)) { log_msg(2, "gemm: %d K-steps do not divide into slices of %d\n", K / BK, vs); return false; }
    const int nblk = ((M + BM - 1) / BM) * (N / BN);
    constexpr size_t lds = (size_t)NST * (BM + BN) * 128;
    // the opt-in to more than 64 KB of dynamic LDS is per device (and per kernel instantiation): one bit per device id
    static std::atomic<unsigned long long> attr_set{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (!((attr_set.load(std::memory_order_acquire) >> dev) & 1ull)) {
        if (hipFuncSetAttribute((const void *)gemm_nt_kernel<BM, BN, WM, WN, NST, Epi, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
        attr_set.fetch_or(1ull << dev, std::memory_order_release);
    }
    // rows per L2 group: a group of ~30-80 tiles per XCD measured best on the encoder shapes (6 rows x 5 columns, 5 x 15, 4 x 20)
    const int nbn = N / BN, gm = group_rows > 0 ? group_rows : g_tile_group_rows > 0 ? g_tile_group_rows : (nbn <= 8 ? 6 : (nbn <= 16 ? 5 : 4));
    hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WM, WN, NST, Epi, ABL>), dim3(nblk, splits), dim3(64 * WM * WN), lds, st, A, lda, W, ldw, M, N, K, gm, epi, vs);
    return hipGetLastError() == hipSuccess;
}

// Tile choice: 256x256 (the 8-phase kernel -- its persistent form, one workgroup per CU walking the tiles, when there are two
// or more tiles per CU and K/64 is even; for K = 64 the 2-stage kernel with the LDS-DMA issue spread over the MFMA
// groups) when N allows it and the grid fills the chip, then
// 256x128 (8 waves, 3 stages, one workgroup per CU) when that grid still fills the chip,
// 64x64 (4 waves, 4 stages) for the skinny decoder-side problems. Activation buffers are allocated with
// their row count rounded up to 256, so the row tiles may over-read.
int skinny_steps(int N, int K);
// dec: a decoder GEMM. Weight shapes that the split-K path slices (skinny_steps > 0) are then accumulated slice by slice
// in the 64x64 kernel as well, so a row's result does not depend on which of the two paths its pass took.
template <class Epi>
static bool dispatch(const f16 *A, long lda, const f16 *W, long ldw, int M, int N, int K, const Epi &epi, hipStream_t st, bool dec = false) {
    const long big = (long)((M + 255) / 256) * (N / 128);
    // one clip's encoder (1500 rows) at N = 4d: 6 x 20 tiles of 256 x 256 leave half the CUs idle, 6 x 40 of 256 x 128 fill them
    // (large-v3-turbo fc1: 34.5 -> 27.0 us; same bits)
    if (M <= 1536 && N % 128 == 0 && N <= 8192 && big >= 192) return launch<256, 128, 4, 2, 3, Epi>(A, lda, W, ldw, M, N, K, epi, st);
    if (persistent_ok(M, N, K)) return launch8p<Epi>(A, lda, W, ldw, M, N, K, epi, st);       // two or more tiles per CU: the persistent form
    if (N % 256 == 0 && big >= 192 && K >= 2 * BK) return launch8<Epi>(A, lda, W, ldw, M, N, K, epi, st);
    if (N % 256 == 0 && big >= 192) return launch<256, 256, 2, 4, 2, Epi, kSpread | kFirstHalf>(A, lda, W, ldw, M, N, K, epi, st);
    if (N % 128 == 0 && M >= 192 && big >= 128) return launch<256, 128, 4, 2, 3, Epi>(A, lda, W, ldw, M, N, K, epi, st);
    if (dec) {
        const int ns = skinny_steps(N, K);
        if (ns > 0) return launch<64, 64, 2, 2, 4, Epi, kVSplit>(A, lda, W, ldw, M, N, K, epi, st, 1, 0, ns);
    }
    return launch<64, 64, 2, 2, 4, Epi>(A, lda, W, ldw, M, N, K, epi, st);
}

#ifdef WMI_HARNESS
// Tile-shape A/B harness (benchmark only): same kernel, explicit variant.
bool gemm_variant(int variant, const f16 *A, long lda, const f16 *W, long ldw, int M, int N, int K, f16 *out, long ldc, hipStream_t st) {
    const EpiF16 e{out, ldc, nullptr, 0, 1.0f};
    switch (variant) {
        case 0: return launch<256, 128, 4, 2, 3, EpiF16>(A, lda, W, ldw, M, N, K, e, st);
        case 1: return launch<256, 256, 2, 4, 2, EpiF16>(A, lda, W, ldw, M, N, K, e, st);
        case 50: return launch8<EpiF16>(A, lda, W, ldw, M, N, K, e, st);                                       // 8-phase kernel, plain f16 epilogue
        case 53: return launch8<EpiF16, kStamps>(A, lda, W, ldw, M, N, K, e, st);                              // ... with per-section cycle stamps
        case 54: return launch8<EpiF16, kNoStagger>(A, lda, W, ldw, M, N, K, e, st);                           // ... all eight waves in step
        case 55: return launch8<EpiF16, kIssueFirst>(A, lda, W, ldw, M, N, K, e, st);                          // ... LDS-DMA issued before the fragment reads
        case 56: return launch8<EpiF32>(A, lda, W, ldw, M, N, K, EpiF32{(float *)out, ldc, nullptr}, st);     // fp32 rows (out must hold M x N floats)
        case 60: return launch8p<EpiF16>(A, lda, W, ldw, M, N, K, e, st);                                      // persistent 8-phase kernel, plain f16 epilogue
        case 61: return launch8p<EpiResidual>(A, lda, W, ldw, M, N, K, EpiResidual{(float *)out, ldc, (const float *)out}, st);
        case 62: return launch8p<EpiF16>(A, lda, W, ldw, M, N, K, EpiF16{out, ldc, (const float *)W, 1, 1.0f}, st);  // bias + GELU
        case 64: return launch8p<EpiF16, kOneTile>(A, lda, W, ldw, M, N, K, e, st);                            // the persistent kernel's loop, one tile per workgroup
        case 63: return launch8p<EpiF32>(A, lda, W, ldw, M, N, K, EpiF32{(float *)out, ldc, nullptr}, st);    // fp32 rows
        case 51: return launch8<EpiResidual>(A, lda, W, ldw, M, N, K, EpiResidual{(float *)out, ldc, (const float *)out}, st);
        case 52: return launch8<EpiF16>(A, lda, W, ldw, M, N, K, EpiF16{out, ldc, (const float *)W, 1, 1.0f}, st);   // bias + GELU
        case 2: return launch<128, 128, 2, 2, 3, EpiF16>(A, lda, W, ldw, M, N, K, e, st);
        case 3: return launch<128, 128, 2, 2, 2, EpiF16>(A, lda, W, ldw, M, N, K, e, st);
        case 4: return launch<256, 128, 4, 2, 2, EpiF16>(A, lda, W, ldw, M, N, K, e, st);
        case 5: return launch<128, 256, 2, 4, 3, EpiF16>(A, lda, W, ldw, M, N, K, e, st);
        case 6: return launch<64, 64, 2, 2, 4, EpiF16>(A, lda, W, ldw, M, N, K, e, st);
        case 7: return launch<128, 64, 2, 2, 4, EpiF16>(A, lda, W, ldw, M, N, K, e, st);
        case 9: return launch<256, 256, 2, 4, 2, EpiF16, 8>(A, lda, W, ldw, M, N, K, e, st);    // LDS-DMA issue spread over the MFMA groups
        case 10: return launch<256, 128, 4, 2, 3, EpiF16, 8>(A, lda, W, ldw, M, N, K, e, st);
        case 15: return launch<256, 256, 2, 4, 2, EpiF16, 8 + 32>(A, lda, W, ldw, M, N, K, e, st);
        case 14: return launch<256, 256, 2, 4, 2, EpiF16, 8 + 16>(A, lda, W, ldw, M, N, K, e, st);   // with per-section cycle stamps
        case 40: return launch<256, 256, 2, 4, 2, EpiF16, 8 + 16 + 32>(A, lda, W, ldw, M, N, K, e, st);         // stamps on the shipped schedule
        case 43: return launch<256, 256, 2, 4, 2, EpiResidual, 8 + 16 + 32>(A, lda, W, ldw, M, N, K, EpiResidual{(float *)out, ldc, (const float *)out}, st);   // stamps, fp32 read-modify-write epilogue
        case 44: return launch<256, 256, 2, 4, 2, EpiF16, 8 + 16 + 32>(A, lda, W, ldw, M, N, K, EpiF16{out, ldc, (const float *)W, 1, 1.0f}, st);              // stamps, bias + GELU epilogue
        case 41: return launch<256, 256, 2, 4, 2, EpiF16, 8 + 16 + 32 + 256>(A, lda, W, ldw, M, N, K, e, st);   // ... epilogue without global stores
        case 42: return launch<256, 256, 2, 4, 2, EpiF16, 8 + 16 + 32 + 512>(A, lda, W, ldw, M, N, K, e, st);   // ... epilogue without the LDS round trip
        case 31: return launch<256, 256, 2, 4, 2, EpiResidual, kSpread | kFirstHalf>(A, lda, W, ldw, M, N, K, EpiResidual{(float *)out, ldc, (const float *)out}, st);   // fp32 read-modify-write epilogue (out must hold M x N floats; bias aliased, values irrelevant)
        case 30: return gemm_splitk(A, lda, W, ldw, M, N, K, gemm_pick_splits(N, K), (float *)out, gemm_splitk_workspace_bytes(M, N, K), st);   // the decoder's skinny path (out must hold the partials)
        case 22: return launch<256, 256, 2, 4, 2, EpiF16, kNoLdsReads | kSpread | kFirstHalf>(A, lda, W, ldw, M, N, K, e, st);   // loads + MFMA, fragments read once
        case 23: return launch<256, 256, 2, 4, 2, EpiF16, kNoLoads | kNoMfma>(A, lda, W, ldw, M, N, K, e, st);                   // LDS fragment reads only
        case 24: return launch<256, 256, 2, 4, 2, EpiF16, kNoLdsReads | kNoMfma | kSpread | kFirstHalf>(A, lda, W, ldw, M, N, K, e, st);   // loads only
        case 27: return launch<256, 256, 2, 4, 2, EpiF16, kSpread | kFirstHalf | kBufferLoad>(A, lda, W, ldw, M, N, K, e, st);   // staging through buffer_load ... lds
        case 25: return launch<256, 256, 2, 4, 2, EpiF16, kNoLdsReads | kNoMfma | kSpread | kFirstHalf | kSameTile>(A, lda, W, ldw, M, N, K, e, st);   // loads only, all L2 hits
        case 26: return launch<256, 256, 2, 4, 2, EpiF16, kSpread | kFirstHalf | kSameTile>(A, lda, W, ldw, M, N, K, e, st);                          // full loop, all L2 hits
        case 16: return launch<256, 256, 2, 4, 2, EpiF16, kSpread | kFirstHalf>(A, lda, W, ldw, M, N, K, EpiF16{out, ldc, (const float *)W, 1, 1.0f}, st);   // shipped kernel with the bias + GELU epilogue (bias values arbitrary)
        case 11: return launch<256, 256, 2, 4, 2, EpiF16, 1>(A, lda, W, ldw, M, N, K, e, st);   // no global loads in the loop
        case 12: return launch<256, 256, 2, 4, 2, EpiF16, 3>(A, lda, W, ldw, M, N, K, e, st);   // no global loads, no LDS reads
        case 13: return launch<256, 256, 2, 4, 2, EpiF16, 4>(A, lda, W, ldw, M, N, K, e, st);   // no MFMA
    }
    return false;
}

#endif  // WMI_HARNESS

// ------------------------------------------------------------------------------------------ skinny split-K GEMM
// The decoder multiplies a few hundred activation rows with weights that are read once per pass: the job is to
// pull the weight bytes through as many CUs as possible with as few latency round trips as possible. One
// workgroup (4 waves, each 32 rows of a 128-row x 64-column tile) owns 64 weight rows x one K-slice of NS*64 columns:
//   * the whole slice (<= 48 KB) is requested at once with LDS-DMA -- one HBM latency per workgroup;
//   * activation fragments do not go through LDS at all: every lane loads its 16-byte MFMA operands for the
//     whole slice straight from global memory (the activation panel is small and L2-resident), each wave its
//     own 32 rows, so no activation byte is fetched twice by a workgroup;
//   * one vmcnt(0) and one barrier publish the slice; then it is MFMAs only;
//   * the fp32 partial tile goes to ws[split][m][n]; the consumer kernels add the splits in a fixed order.
// Up to 8 such workgroups fit a CU, so a GEMM of 200-400 workgroups runs as one co-resident batch.
template <int NS>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const f16 *__restrict__ A, long lda, const f16 *__restrict__ W, long ldw,
                                                          int M, int N, float *__restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * 64, split = blockIdx.y, m0 = blockIdx.z * 128 + wave * 32;     // a wave owns 32 rows x all 64 columns
    const long k0 = (long)split * NS * 64;
    // weights: stage s = 64 rows x 128 B, 16-byte chunk c of row r at slot c ^ (r & 7); two DMA instructions per wave and stage
    {
        const int srow = lane >> 3, sslot = lane & 7;
        const f16 *src = W + (long)(n0 + wave * 8 + srow) * ldw + k0 + ((sslot ^ srow) << 3);
#pragma unroll
        for (int st = 0; st < NS; st++)
#pragma unroll
            for (int p = 0; p < 2; p++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (long)p * 32 * ldw + st * 64),
                                                 (__attribute__((address_space(3))) void *)(smem + st * 8192 + (p * 32 + wave * 8) * 128), 16, 0, 0);
    }
    const int fr = lane & 15, fq = lane >> 4;
    const f16 *a_ptr = A + (long)(m0 + fr) * lda + k0 + fq * 8;
    f16x8 fa[NS][2][2];                   // [K-step][k-half][row tile]: the wave's whole activation slice, requested at once
#pragma unroll
    for (int st = 0; st < NS; st++)
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int j = 0; j < 2; j++) fa[st][ks][j] = *(const f16x8 *)(a_ptr + (long)j * 16 * lda + st * 64 + ks * 32);
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = f32x4{0, 0, 0, 0};
    __builtin_amdgcn_s_waitcnt(0x0F70);    // vmcnt(0): my share of the weight slice has landed (and my activations)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int st = 0; st < NS; st++) {
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            f16x8 fw[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int row = i * 16 + fr;
                fw[i] = *(const f16x8 *)(smem + st * 8192 + row * 128 + (((ks * 4 + fq) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[i], fa[st][ks][j], acc[i][j], 0, 0, 0);
        }
    }
    float *o = ws + (long)split * M * N;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int m = m0 + j * 16 + fr, n = n0 + i * 16 + fq * 4;
            if (m < M) *(f32x4 *)(o + (long)m * N + n) = acc[i][j];
        }
}

// K-steps of 64 per workgroup (1..6; 0 = do not split): the largest slice that still gives >= 192 workgroups for one
// 128-row panel. The choice depends on the weight shape only, never on the number of rows, and the fused-epilogue 64x64
// kernel folds its accumulators at the same slice boundaries (kVSplit), so a sequence's numbers do not depend on what
// else shares its pass or on which of the two paths the pass took.
int skinny_steps(int N, int K) {
    if (N % 64 != 0 || K % BK != 0) return 0;
    const int tiles = N / 64;
    if (tiles >= 40) return 0;           // wide outputs have enough tiles for the fused-epilogue kernels, whose activation
                                         // traffic per weight byte is lower and which write no partial sums
    const int nk = K / BK;
    int pick = 0;
    for (int ns = 6; ns >= 1; ns--)
        if (nk % ns == 0 && nk / ns <= 16) { pick = ns; if (tiles * (nk / ns) >= 192) break; }
    return pick && (nk / pick) > 1 ? pick : 0;
}
int gemm_pick_splits(int N, int K) { const int ns = skinny_steps(N, K); return ns ? K / BK / ns : 1; }
// Bytes of partial sums one split GEMM of `rows` rows writes: ws[splits][rows][N]. 0 when (N, K) does not split.
size_t gemm_splitk_workspace_bytes(int rows, int N, int K) {
    const int splits = gemm_pick_splits(N, K);
    return splits > 1 ? (size_t)splits * rows * N * sizeof(float) : 0;
}
bool gemm_splitk(const f16 *A, long lda, const f16 *W, long ldw, int M, int N, int K, int splits, float *ws, size_t ws_bytes, hipStream_t st) {
    const int ns = splits > 0 && K % (splits * BK) == 0 ? K / BK / splits : 0;
    if (ns < 1 || ns > 6 || N % 64 != 0 || splits > 16) { log_msg(2, "gemm_splitk: bad split %d for M=%d N=%d K=%d\n", splits, M, N, K); return false; }
    if ((size_t)splits * M * N * sizeof(float) > ws_bytes) { log_msg(2, "gemm_splitk: %d x %d x %d partial sums exceed the %zu-byte workspace\n", splits, M, N, ws_bytes); return false; }
    const dim3 grid(N / 64, splits, (M + 127) / 128);
#define WMI_SKINNY(NS_) case NS_: hipLaunchKernelGGL((gemm_skinny_kernel<NS_>), grid, dim3(256), NS_ * 8192, st, A, lda, W, ldw, M, N, ws); break;
    switch (ns) { WMI_SKINNY(1) WMI_SKINNY(2) WMI_SKINNY(3) WMI_SKINNY(4) WMI_SKINNY(5) WMI_SKINNY(6) }
#undef WMI_SKINNY
    return hipGetLastError() == hipSuccess;
}

// ------------------------------------------------------------------------------------------ few-row GEMM with a LayerNorm prologue
// Decoder passes of at most 16 rows (one clip, its beams) are a chain of dependent launches of a few
// microseconds each, and their cost is the number of launches. This kernel is one launch per projection with nothing
// before or after it: a workgroup owns 16 output columns over the whole K and all rows,
//   * LN: it first normalises every row itself (layer_norm_kernel's arithmetic, operation for operation; the rows are a
//     few KB and L2-resident, so every workgroup redoing them is cheaper than a launch) and keeps the f16 result in LDS
//     (only the rows the pass has: one clip is one row, and three workgroups then fit a CU);
//     otherwise the f16 activation fragments go global -> registers;
//   * its four or eight waves share the K-slices of the split-K path (slice s to wave s mod 4 / 8), each wave streaming its weight
//     fragments global -> registers two chunks of eight 32-wide steps ahead -- no LDS round trip, no barrier in the loop;
//   * the per-slice sums meet in LDS and are added in slice order, then the usual epilogue functor runs.
// Same operand order per MFMA, same chain per slice, same fold order as gemm_skinny_kernel + its consumers and as the
// kVSplit 64x64 kernel: the three paths give the same bits.
// One lane's weight operand of one 32-wide K-step as it travels from memory to the MFMA: 8 halfs, or -- QT = the ggml tensor
// type of a block-quantised matrix -- its 8 quants, the block's {d, m} word and fifth-bit word. dequant8 turns the latter
// into the fp16 values the load-time expansion (expand_blocks_kernel) writes, bit for bit: quant x d (+ m, or - zero x d)
// is exact in fp32 for quants of at most 8 bits and an fp16 scale, so one fused multiply-add per value rounds exactly
// once where the expansion's multiply (exact) and add round once -- then the same conversion to fp16. (Only the sign of
// a zero weight can differ, which no sum of products can see.)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int QT>
__device__ __forceinline__ u32x4 load_wfrag(const f16 *W, long ldw, const QuantW &qw, int nb, int row, int t, int quad) {
    if constexpr (QT == 0) {
        return *(const u32x4 *)(W + (long)row * ldw + t * 32 + quad * 8);
    } else {
        const long blk = (long)row * nb + t;
        u32x4 r;
        const uint2 q = QT == 8 ? *(const uint2 *)(qw.qs + blk * 32 + quad * 8) : *(const uint2 *)(qw.qs + blk * 16 + (quad & 1) * 8);
        r[0] = q.x; r[1] = q.y; r[2] = qw.dm[blk];
        r[3] = (QT == 6 || QT == 7) ? qw.qh[blk] : 0u;
        return r;
    }
}
template <int QT>
__device__ __forceinline__ f16x8 wfrag_f16(u32x4 r, int quad) {
    if constexpr (QT == 0) {
        return __builtin_bit_cast(f16x8, r);
    } else {
        const unsigned dmw = r[2];                                   // (a bit_cast straight from the vector element reads element 0 with this compiler)
        const f16x2 dmh = __builtin_bit_cast(f16x2, dmw);
        const float d = (float)dmh[0];
        unsigned w0, w1;
        float c;
        if (QT == 8) {
            w0 = r[0] ^ 0x80808080u; w1 = r[1] ^ 0x80808080u;       // int8 + 128 as unsigned bytes
            c = -128.0f * d;
        } else {
            const int sh = (quad >> 1) * 4;                          // weights 0..15 are the low nibbles, 16..31 the high ones
            w0 = (r[0] >> sh) & 0x0F0F0F0Fu; w1 = (r[1] >> sh) & 0x0F0F0F0Fu;
            if (QT == 6 || QT == 7) {                                // fifth bit of weight v is bit v of the block's word
                const unsigned hb = (r[3] >> (8 * quad)) & 0xFFu;
                w0 |= (((hb & 0xFu) * 0x00204081u) & 0x01010101u) << 4;          // bit i -> byte i
                w1 |= (((hb >> 4) * 0x00204081u) & 0x01010101u) << 4;
            }
            c = (QT == 3 || QT == 7) ? (float)dmh[1] : (QT == 6 ? -16.0f : -8.0f) * d;
        }
        f16x8 o;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float a = __builtin_fmaf((float)((w0 >> (8 * i)) & 0xFFu), d, c), b = __builtin_fmaf((float)((w1 >> (8 * i)) & 0xFFu), d, c);
            asm volatile("" : "+v"(a), "+v"(b));                     // the fp32 values are pinned: no mixed-precision fusion with the conversion
            o[i] = (f16)a; o[4 + i] = (f16)b;
        }
        return o;
    }
}

template <int RT, int NWV, bool LN, int QT, class Epi>
__global__ __launch_bounds__(64 * NWV) void gemm_rows_kernel(const float *__restrict__ x, const float *__restrict__ ln_w, const float *__restrict__ ln_b,
                                                        const f16 *__restrict__ A, long lda, const f16 *__restrict__ W, long ldw,
                                                        int R, int K, int per_slice, int nslice, Epi epi, QuantW qw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CH = 8;                                          // 32-wide K-steps per register chunk
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4, n0 = blockIdx.x * 16;
    f32x4 *part = (f32x4 *)smem;                                   // [slice][row tile][lane]
    char *act = smem + (size_t)nslice * RT * 1024;                 // LN: [R rows][K halfs + 8]: rows 16 B apart in the banks
    const int stride = K * 2 + 16;
    const int nb = K >> 5;                                         // blocks of 32 per weight row
    const f16 *ap = LN ? nullptr : A + (long)fr * lda + fq * 8;
    const int psc = (per_slice + CH - 1) / CH;                      // chunks per slice
    const int my_slices = wave < nslice ? (nslice - wave + NWV - 1) / NWV : 0;
    const int nchunk = my_slices * psc;
    constexpr int AR = LN ? 1 : RT, AC = LN ? 1 : CH;
    u32x4 wa[CH], wb[CH];
    f16x8 aa[AR][AC], ab[AR][AC];
    f32x4 acc[RT];
    // chunk q of this wave: slice wave + NWV*(q / psc), steps t0 .. t0 + n - 1
    auto locate = [&](int q, int &si, int &ci, int &n) {
        si = q / psc; ci = q - si * psc;
        n = min(CH, per_slice - ci * CH);
        return (wave + NWV * si) * per_slice + ci * CH;
    };
    auto load = [&](u32x4 (&wr)[CH], f16x8 (&ar)[AR][AC], int q) {
        int si, ci, n;
        const int t0 = locate(q, si, ci, n);
#pragma unroll
        for (int i = 0; i < CH; i++)
            if (i < n) {
                wr[i] = load_wfrag<QT>(W, ldw, qw, nb, n0 + fr, t0 + i, fq);
                if constexpr (!LN) {
#pragma unroll
                    for (int j = 0; j < RT; j++) {               // rows past the pass are not fetched (one clip: 1 row of the tile's 16)
                        ar[j][i] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                        if (j * 16 + fr < R) ar[j][i] = *(const f16x8 *)(ap + (long)j * 16 * lda + (long)(t0 + i) * 32);
                    }
                }
            }
    };
    auto compute = [&](u32x4 (&wr)[CH], f16x8 (&ar)[AR][AC], int q) {
        int si, ci, n;
        const int t0 = locate(q, si, ci, n);
        if (ci == 0) {
#pragma unroll
            for (int j = 0; j < RT; j++) acc[j] = f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < CH; i++)
            if (i < n) {
                const f16x8 fw = wfrag_f16<QT>(wr[i], fq);
#pragma unroll
                for (int j = 0; j < RT; j++) {
                    f16x8 fa;
                    if constexpr (LN) {
                        fa = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                        if (j * 16 + fr < R) fa = *(const f16x8 *)(act + (size_t)(j * 16 + fr) * stride + ((t0 + i) * 32 + fq * 8) * 2);
                    } else fa = ar[j][i];
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw, fa, acc[j], 0, 0, 0);
                }
            }
        if (ci == psc - 1) {
            const int sl = wave + NWV * si;
#pragma unroll
            for (int j = 0; j < RT; j++) part[(sl * RT + j) * 64 + lane] = acc[j];
        }
    };
    // A projection whose K is one slice is one accumulation chain, i.e. one wave's work; streaming its weights through that
    // wave's registers alone would take K/256 dependent round trips. Then all waves fetch the workgroup's 16 weight rows
    // (requested here, before the LayerNorm) and park them in LDS beside the activations; wave 0 multiplies from LDS.
    const bool staged_w = LN && nslice == 1 && K <= 1280;
    char *wl = act + (size_t)R * stride;
    constexpr int WP = 1280 / 8 * 16 / (64 * NWV);                 // 16-byte pieces per thread at the widest K
    u32x4 wtmp[LN ? WP : 1];
    if constexpr (LN) {
        if (staged_w) {
            const int per_row = K >> 3;
#pragma unroll
            for (int i = 0; i < WP; i++) {
                const int p = tid + 64 * NWV * i;
                if (p < 16 * per_row) { const int r = p / per_row, c = p - r * per_row; wtmp[i] = load_wfrag<QT>(W, ldw, qw, nb, n0 + r, c >> 2, c & 3); }
            }
        }
    }
    if (!staged_w && nchunk > 0) load(wa, aa, 0);          // the first weight chunk is on its way while the rows are normalised
    if constexpr (LN) {
        const int nch = K >> 2;
        constexpr int RW = RT * 16 / NWV;                          // rows per wave: row = wave + NWV * rr
        f32x4 v[RW][5];
#pragma unroll
        for (int rr = 0; rr < RW; rr++) {                          // every row of the wave requested before the first is reduced
            const int row = wave + NWV * rr;
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const int c = lane + 64 * i;
                v[rr][i] = f32x4{0, 0, 0, 0};
                if (c < nch && row < R) v[rr][i] = *((const f32x4 *)(x + (long)row * K) + c);
            }
        }
#pragma unroll
        for (int rr = 0; rr < RW; rr++) {
            const int row = wave + NWV * rr;
            char *dst = act + (size_t)row * stride;
            if (row >= R) continue;                                // rows past the pass have no LDS row: their operand is zero
            float sum = 0.0f;
#pragma unroll
            for (int i = 0; i < 5; i++) sum += (v[rr][i][0] + v[rr][i][1]) + (v[rr][i][2] + v[rr][i][3]);
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
            const float mean = sum / (float)K;
            float sq = 0.0f;
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const int c = lane + 64 * i;
                if (c < nch) { v[rr][i] -= mean; sq += (v[rr][i][0] * v[rr][i][0] + v[rr][i][1] * v[rr][i][1]) + (v[rr][i][2] * v[rr][i][2] + v[rr][i][3] * v[rr][i][3]); }
            }
            for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
            const float scale = 1.0f / sqrtf(sq / (float)K + 1e-5f);
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const int c = lane + 64 * i;
                if (c < nch) {
                    const f32x4 ww = *(const f32x4 *)(ln_w + 4 * c), bb = *(const f32x4 *)(ln_b + 4 * c);
                    f32x4 y;
#pragma unroll
                    for (int k = 0; k < 4; k++) y[k] = v[rr][i][k] * scale * ww[k] + bb[k];
                    *(f16x4 *)(dst + 8 * c) = f16x4{(f16)y[0], (f16)y[1], (f16)y[2], (f16)y[3]};
                }
            }
        }
        if (staged_w) {
            const int per_row = K >> 3;
#pragma unroll
            for (int i = 0; i < WP; i++) {
                const int p = tid + 64 * NWV * i;
                if (p < 16 * per_row) { const int r = p / per_row, c = p - r * per_row; *(f16x8 *)(wl + (size_t)r * stride + c * 16) = wfrag_f16<QT>(wtmp[i], c & 3); }
            }
        }
        __syncthreads();
        if (staged_w) {
            if (wave == 0) {
#pragma unroll
                for (int j = 0; j < RT; j++) acc[j] = f32x4{0, 0, 0, 0};
                for (int t = 0; t < (K >> 5); t++) {
                    const f16x8 fw = *(const f16x8 *)(wl + (size_t)fr * stride + (t * 32 + fq * 8) * 2);
#pragma unroll
                    for (int j = 0; j < RT; j++) {
                        f16x8 fa = {0, 0, 0, 0, 0, 0, 0, 0};
                        if (j * 16 + fr < R) fa = *(const f16x8 *)(act + (size_t)(j * 16 + fr) * stride + (t * 32 + fq * 8) * 2);
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw, fa, acc[j], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int j = 0; j < RT; j++) part[j * 64 + lane] = acc[j];
            }
        }
    }
    for (int q = 0; q < (staged_w ? 0 : nchunk); q += 2) {
        if (q + 1 < nchunk) load(wb, ab, q + 1);
        compute(wa, aa, q);
        if (q + 2 < nchunk) load(wa, aa, q + 2);
        if (q + 1 < nchunk) compute(wb, ab, q + 1);
    }
    __syncthreads();
    if (wave < RT) {
        const int j = wave;
        f32x4 a = part[j * 64 + lane];
        for (int sl = 1; sl < nslice; sl++) a += part[(sl * RT + j) * 64 + lane];
        const int m = j * 16 + fr;
        if (m < R) epi(m, n0 + fq * 4, a);
    }
}

// Rows, widths and depths the few-row kernel takes: up to 16 rows (one MFMA row tile; with two, a workgroup pulls twice
// its weight bytes in activation fragments and the split-K path is faster: 259 against 224 ms per step on 32 clips of
// Whisper small), K a multiple of 64, N of 16; with the LayerNorm prologue K is the model width (<= 1280).
bool gemm_rows_ok(int R, int N, int K, bool ln) {
    return R >= 1 && R <= 16 && N % 16 == 0 && K % BK == 0 && K <= 8192 && (!ln || K <= 1280);
}
template <int RT, int NWV, bool LN, int QT, class Epi>
static bool launch_rows_rt(const float *x, const float *ln_w, const float *ln_b, const f16 *A, long lda, const f16 *W, long ldw, int R, int N, int K, const Epi &epi, hipStream_t st, const QuantW &qw) {
    const int ns = skinny_steps(N, K), per_slice = ns ? ns * 2 : K / 32, nslice = (K / 32) / per_slice;
    if (nslice < 1 || nslice > 16 || nslice * per_slice != K / 32) { log_msg(2, "gemm_rows: bad slicing of K=%d\n", K); return false; }
    const size_t lds = (size_t)nslice * RT * 1024 + (LN ? (size_t)R * (K * 2 + 16) : 0) + (LN && nslice == 1 && K <= 1280 ? (size_t)16 * (K * 2 + 16) : 0);
    if (lds > 160 * 1024) { log_msg(2, "gemm_rows: %zu bytes of LDS\n", lds); return false; }
    static std::atomic<unsigned long long> attr_set{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (lds > 64 * 1024 && !((attr_set.load(std::memory_order_acquire) >> dev) & 1ull)) {
        if (hipFuncSetAttribute((const void *)gemm_rows_kernel<RT, NWV, LN, QT, Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
        attr_set.fetch_or(1ull << dev, std::memory_order_release);
    }
    hipLaunchKernelGGL((gemm_rows_kernel<RT, NWV, LN, QT, Epi>), dim3(N / 16), dim3(64 * NWV), lds, st, x, ln_w, ln_b, A, lda, W, ldw, R, K, per_slice, nslice, epi, qw);
    return hipGetLastError() == hipSuccess;
}
template <bool LN, int QT, class Epi>
static bool launch_rows_qt(const float *x, const float *ln_w, const float *ln_b, const f16 *A, long lda, const f16 *W, long ldw, int R, int N, int K, const Epi &epi, hipStream_t st, const QuantW &qw) {
    // eight waves where the K-slices give every one of them work: twice the weight bytes in flight per workgroup
    const int ns = skinny_steps(N, K);
    if (ns && K / BK / ns >= 8) return launch_rows_rt<1, 8, LN, QT, Epi>(x, ln_w, ln_b, A, lda, W, ldw, R, N, K, epi, st, qw);
    return launch_rows_rt<1, 4, LN, QT, Epi>(x, ln_w, ln_b, A, lda, W, ldw, R, N, K, epi, st, qw);
}
template <bool LN, class Epi>
static bool launch_rows(const float *x, const float *ln_w, const float *ln_b, const f16 *A, long lda, const f16 *W, long ldw, int R, int N, int K, const Epi &epi, hipStream_t st, const QuantW *qw) {
    if (!gemm_rows_ok(R, N, K, LN)) { log_msg(2, "gemm_rows: unsupported shape R=%d N=%d K=%d\n", R, N, K); return false; }
    const QuantW none{};
    if (!qw || !qw->ttype) return launch_rows_qt<LN, 0, Epi>(x, ln_w, ln_b, A, lda, W, ldw, R, N, K, epi, st, none);
    if (!qw->qs || !qw->dm || ((qw->ttype == 6 || qw->ttype == 7) && !qw->qh)) return false;
    switch (qw->ttype) {
        case 2: return launch_rows_qt<LN, 2, Epi>(x, ln_w, ln_b, A, lda, W, ldw, R, N, K, epi, st, *qw);
        case 3: return launch_rows_qt<LN, 3, Epi>(x, ln_w, ln_b, A, lda, W, ldw, R, N, K, epi, st, *qw);
        case 6: return launch_rows_qt<LN, 6, Epi>(x, ln_w, ln_b, A, lda, W, ldw, R, N, K, epi, st, *qw);
        case 7: return launch_rows_qt<LN, 7, Epi>(x, ln_w, ln_b, A, lda, W, ldw, R, N, K, epi, st, *qw);
        case 8: return launch_rows_qt<LN, 8, Epi>(x, ln_w, ln_b, A, lda, W, ldw, R, N, K, epi, st, *qw);
    }
    return false;
}
// q, K/V pages <- LayerNorm(x) Wqkv^T + b
bool gemm_rows_ln_dec_qkv(const float *x, const float *ln_w, const float *ln_b, const f16 *W, int R, int d, int H, f16 *q, f16 *kc, f16 *vc, const float *bias, const DecRow *rows, hipStream_t st, const QuantW *qw) {
    return launch_rows<true>(x, ln_w, ln_b, nullptr, 0, W, d, R, 3 * d, d, EpiDecQKV{q, kc, vc, bias, rows, d, H}, st, qw);
}
// out <- f16(act((LayerNorm(x) W^T + b) * scale))
bool gemm_rows_ln_f16(const float *x, const float *ln_w, const float *ln_b, const f16 *W, int R, int N, int K, f16 *out, long ldc, const float *bias, int gelu, float scale, hipStream_t st, const QuantW *qw) {
    return launch_rows<true>(x, ln_w, ln_b, nullptr, 0, W, K, R, N, K, EpiF16{out, ldc, bias, gelu, scale}, st, qw);
}
// x += A W^T + b
bool gemm_rows_residual(const f16 *A, long lda, const f16 *W, int R, int N, int K, float *x, long ldc, const float *bias, hipStream_t st, const QuantW *qw) {
    return launch_rows<false>(nullptr, nullptr, nullptr, A, lda, W, K, R, N, K, EpiResidual{x, ldc, bias}, st, qw);
}

#ifdef WMI_HARNESS
bool gemm_read_stamps(unsigned long long *out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_stamps), sizeof(unsigned long long) * n) == hipSuccess;
}
#endif

bool gemm_f16(const f16 *A, long lda, const f16 *W, long ldw, int M, int N, int K, f16 *out, long ldc, const float *bias, int gelu, float scale, hipStream_t st, bool dec) {
    return dispatch(A, lda, W, ldw, M, N, K, EpiF16{out, ldc, bias, gelu, scale}, st, dec);
}
bool gemm_f32(const f16 *A, long lda, const f16 *W, long ldw, int M, int N, int K, float *out, long ldc, const float *bias, hipStream_t st) {
    return dispatch(A, lda, W, ldw, M, N, K, EpiF32{out, ldc, bias}, st);
}
bool gemm_residual(const f16 *A, long lda, const f16 *W, long ldw, int M, int N, int K, float *x, long ldc, const float *bias, hipStream_t st, bool dec) {
    return dispatch(A, lda, W, ldw, M, N, K, EpiResidual{x, ldc, bias}, st, dec);
}
bool gemm_conv1(const f16 *melT, int n_mel, const f16 *W, int Kpad, int M, int d, f16 *h1, const float *bias, int T2, hipStream_t st) {
    return dispatch(melT, n_mel, W, Kpad, M, d, Kpad, EpiConv1{h1, d, bias, T2 + 2, T2}, st);
}
bool gemm_conv2(const f16 *h1, const f16 *W, int M, int d, float *x, const float *bias, const float *pe, int T, hipStream_t st) {
    return dispatch(h1, 2L * d, W, 3L * d, M, d, 3 * d, EpiConv2{x, d, bias, pe, T}, st);
}
// Big batches: Q and K as one GEMM over the first 2d weight rows, V as its own transposed GEMM (see EpiVt) whose "rows"
// are the d weight rows and whose "columns" are the M activation rows; all tile rows of that GEMM form one L2 group.
// xn must then be readable up to its row count rounded up to 256 (the engine's activation buffers are).
bool gemm_enc_qkv_is_split(int M, int d, int T) {
    const int Mp = (M + 255) / 256 * 256;
    return d % 256 == 0 && T % 4 == 0 && (long)(d / 256) * (Mp / 128) >= 192;
}
bool gemm_enc_qk(const f16 *xn, const f16 *W, int M, int d, int H, int T, int Tp, f16 *Q, f16 *Kb, const float *bias, hipStream_t st) {
    return dispatch(xn, d, W, d, M, 2 * d, d, EpiEncQKV{Q, Kb, nullptr, bias, d, H, T, Tp}, st);
}
bool gemm_enc_vt(const f16 *xn, const f16 *W, int M, int d, int H, int T, int Tp, f16 *Vt, const float *bias, hipStream_t st) {
    if (!gemm_enc_qkv_is_split(M, d, T)) return false;
    const int Mp = (M + 255) / 256 * 256;
    if (persistent_ok(d, Mp, d)) return launch8p<EpiVt>(W + (long)2 * d * d, d, xn, d, d, Mp, d, EpiVt{Vt, bias + 2 * d, H, T, Tp, M}, st, d / 256);
    return launch8<EpiVt>(W + (long)2 * d * d, d, xn, d, d, Mp, d, EpiVt{Vt, bias + 2 * d, H, T, Tp, M}, st, d / 256);
}
bool gemm_enc_qkv(const f16 *xn, const f16 *W, int M, int d, int H, int T, int Tp, f16 *Q, f16 *Kb, f16 *Vt, const float *bias, hipStream_t st) {
    return dispatch(xn, d, W, d, M, 3 * d, d, EpiEncQKV{Q, Kb, Vt, bias, d, H, T, Tp}, st);
}
bool gemm_cross_kv(const f16 *enc, const f16 *W, int M, int d, int H, int T, int Tp, int n_layer, f16 *out, const float *bias, const int *slots, hipStream_t st) {
    return dispatch(enc, d, W, d, M, n_layer * 2 * d, d, EpiCrossKV{out, bias, slots, d, H, T, Tp, n_layer}, st);
}
bool gemm_dec_qkv(const f16 *xn, const f16 *W, int M, int d, int H, f16 *q, f16 *kc, f16 *vc, const float *bias, const DecRow *rows, hipStream_t st) {
    return dispatch(xn, d, W, d, M, 3 * d, d, EpiDecQKV{q, kc, vc, bias, rows, d, H}, st, true);
}

}  // namespace wmi
