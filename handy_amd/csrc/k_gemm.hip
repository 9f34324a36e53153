// fp16 x fp16 -> fp32 "NT" GEMM for gfx950:  C[m][n] = sum_k Act[m][k] * W[n][k]   (+ fused epilogue)
//
// Both operands are K-contiguous (activations row-major, weights stored [out][in] as in the ggml
// file), so every MFMA fragment is one 16-byte run along K. Structure per workgroup (256 threads,
// 4 waves in a 2x2 arrangement):
//   * BM x 64 activation tile and BN x 64 weight tile per K-step, copied global->LDS with the
//     16-byte LDS-DMA form (__builtin_amdgcn_global_load_lds); two LDS stages, the copy of step
//     k+1 is in flight while step k is multiplied; one barrier per K-step.
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

namespace wmi {

constexpr int BK = 64;             // halfs per K-step (128 B rows in LDS)

__device__ __forceinline__ float gelu_f16_dev(float x) {
    // tanh-form GELU evaluated on the f16-rounded input and rounded to f16 (the values ggml's
    // f16 lookup table holds); saturates like the table's guard: x<=-10 -> 0, x>=10 -> x.
    if (x <= -10.0f) return 0.0f;
    if (x >= 10.0f) return x;
    float h = (float)(f16)x;
    float u = 0.79788456080286535587989211986876f * h * (1.0f + 0.044715f * h * h);
    // tanh(u) = 1 - 2/(exp(2u)+1)
    float e = __expf(2.0f * u);
    float t = 1.0f - 2.0f / (e + 1.0f);
    float g = 0.5f * h * (1.0f + t);
    return (float)(f16)g;
}

// ------------------------------------------------------------------------------------------ epilogues
// Each epilogue receives one lane's 4 accumulators: row m, columns n..n+3 (n % 4 == 0).

struct EpiF16 {            // out[m][n] = f16(act((acc + bias) * scale))
    f16 *out; long ldc; const float *bias; int gelu; float scale;
    __device__ __forceinline__ void operator()(int m, int n, f32x4 a) const {
        if (bias) { const f32x4 b = *(const f32x4 *)(bias + n); a += b; }
        a *= scale;
        if (gelu) { a[0] = gelu_f16_dev(a[0]); a[1] = gelu_f16_dev(a[1]); a[2] = gelu_f16_dev(a[2]); a[3] = gelu_f16_dev(a[3]); }
        f16x4 h = {(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3]};
        *(f16x4 *)(out + (long)m * ldc + n) = h;
    }
};

struct EpiF32 {            // out[m][n] = acc + bias
    float *out; long ldc; const float *bias;
    __device__ __forceinline__ void operator()(int m, int n, f32x4 a) const {
        if (bias) { const f32x4 b = *(const f32x4 *)(bias + n); a += b; }
        *(f32x4 *)(out + (long)m * ldc + n) = a;
    }
};

struct EpiResidual {       // x[m][n] += acc + bias   (fp32 residual stream, in place)
    float *x; long ldc; const float *bias;
    __device__ __forceinline__ void operator()(int m, int n, f32x4 a) const {
        float *p = x + (long)m * ldc + n;
        const f32x4 b = *(const f32x4 *)(bias + n);
        const f32x4 r = *(const f32x4 *)p;
        *(f32x4 *)p = r + (a + b);
    }
};

struct EpiConv1 {          // rows have period P = 2T+2 per utterance; h1[m+1][n] = t < 2T ? gelu(acc+b) : 0
    f16 *h1; int d; const float *bias; int period; int t_valid;
    __device__ __forceinline__ void operator()(int m, int n, f32x4 a) const {
        const int t = m % period;
        f16x4 h = {(f16)0, (f16)0, (f16)0, (f16)0};
        if (t < t_valid) {
            const f32x4 b = *(const f32x4 *)(bias + n); a += b;
            h = f16x4{(f16)gelu_f16_dev(a[0]), (f16)gelu_f16_dev(a[1]), (f16)gelu_f16_dev(a[2]), (f16)gelu_f16_dev(a[3])};
        }
        *(f16x4 *)(h1 + (long)(m + 1) * d + n) = h;
    }
};

struct EpiConv2 {          // rows have period T+1; x[b*T+t][n] = gelu(acc+b) + pos_emb[t][n], junk row t==T dropped
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

struct EpiEncQKV {         // n in [0,d): Q*(1/8) -> Q[b][h][t][64]; [d,2d): K -> K[b][h][t][64]; [2d,3d): V -> Vt[b][h][64][Tp]
    f16 *Q, *K, *Vt; const float *bias; int d, H, T, Tp;
    __device__ __forceinline__ void operator()(int m, int n, f32x4 a) const {
        const int b = m / T, t = m - b * T;
        const f32x4 bb = *(const f32x4 *)(bias + n); a += bb;
        const int part = n / d, c = n - part * d, h = c >> 6, e = c & 63;
        if (part == 0) {
            f16x4 o = {(f16)(a[0] * 0.125f), (f16)(a[1] * 0.125f), (f16)(a[2] * 0.125f), (f16)(a[3] * 0.125f)};
            *(f16x4 *)(Q + (((long)b * H + h) * Tp + t) * 64 + e) = o;
        } else if (part == 1) {
            f16x4 o = {(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3]};
            *(f16x4 *)(K + (((long)b * H + h) * Tp + t) * 64 + e) = o;
        } else {
            f16 *p = Vt + (((long)b * H + h) * 64 + e) * Tp + t;
            p[0] = (f16)a[0]; p[Tp] = (f16)a[1]; p[2 * Tp] = (f16)a[2]; p[3 * Tp] = (f16)a[3];
        }
    }
};

struct EpiCrossKV {        // n = (layer*2 + kv)*d + h*64 + e ; out[slot(b)][layer][kv][h][t][64]
    f16 *out; const float *bias; const int *slots; int d, H, T, Tp, n_layer;
    __device__ __forceinline__ void operator()(int m, int n, f32x4 a) const {
        const int b = m / T, t = m - b * T;
        const f32x4 bb = *(const f32x4 *)(bias + n); a += bb;
        const int lk = n / d, c = n - lk * d, h = c >> 6, e = c & 63;
        const long slot = slots[b];
        f16x4 o = {(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3]};
        *(f16x4 *)(out + ((((slot * n_layer * 2 + lk) * H + h) * (long)Tp) + t) * 64 + e) = o;
    }
};

struct EpiDecQKV {         // q -> qbuf[m][d] * 1/8 ; k,v -> paged cache [page][H][16][64] at rows[m].kv_index
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

// ------------------------------------------------------------------------------------------ kernel
template <int BM, int BN, class Epi>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const f16 *__restrict__ A, long lda, const f16 *__restrict__ W, long ldw,
                                                      int M, int N, int K, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int TM = BM / 32, TN = BN / 32;     // 16x16 tiles per wave along m and n
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // XCD-aware remap: physical ids p, p+8, ... share an XCD; give each XCD a contiguous run of logical tiles
    const int nbn = N / BN, nbm = (M + BM - 1) / BM, nblk = nbm * nbn;
    int p = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, x = p & 7, i = p >> 3;
        p = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int bm = p / nbn, bn = p - bm * nbn;
    const int m0 = bm * BM, n0 = bn * BN;
    const int wr = wave >> 1, wc = wave & 1;

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; i++)
#pragma unroll
        for (int j = 0; j < TM; j++) acc[i][j] = f32x4{0, 0, 0, 0};

    // --- staging: one LDS-DMA wave-instruction moves 8 rows x 128 B
    const int srow = lane >> 3, sslot = lane & 7;
    auto stage = [&](int kt, int buf) {
        char *base = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < BM / 32; i++) {
            const int r0 = (i * 4 + wave) * 8, row = r0 + srow, chunk = sslot ^ (row & 7);
            const f16 *src = A + (long)(m0 + row) * lda + (long)kt * BK + chunk * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(base + r0 * 128), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < BN / 32; i++) {
            const int r0 = (i * 4 + wave) * 8, row = r0 + srow, chunk = sslot ^ (row & 7);
            const f16 *src = W + (long)(n0 + row) * ldw + (long)kt * BK + chunk * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(base + A_BYTES + r0 * 128), 16, 0, 0);
        }
    };

    const int nk = K / BK;
    stage(0, 0);
    const int fr = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nk; kt++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        const char *sa = smem + (kt & 1) * STAGE, *sb = sa + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            f16x8 fa[TM], fw[TN];
#pragma unroll
            for (int j = 0; j < TM; j++) {
                const int row = wr * (BM / 2) + j * 16 + fr;
                fa[j] = *(const f16x8 *)(sa + row * 128 + (((ks * 4 + fq) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TN; i++) {
                const int row = wc * (BN / 2) + i * 16 + fr;
                fw[i] = *(const f16x8 *)(sb + row * 128 + (((ks * 4 + fq) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TN; i++)
#pragma unroll
                for (int j = 0; j < TM; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[i], fa[j], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < TN; i++)
#pragma unroll
        for (int j = 0; j < TM; j++) {
            const int m = m0 + wr * (BM / 2) + j * 16 + fr;
            const int n = n0 + wc * (BN / 2) + i * 16 + fq * 4;
            if (m < M) epi(m, n, acc[i][j]);
        }
}

template <int BM, int BN, class Epi>
static bool launch(const f16 *A, long lda, const f16 *W, long ldw, int M, int N, int K, const Epi &epi, hipStream_t st) {
    if (M <= 0) return true;
    if (N % BN != 0 || K % BK != 0) { log_msg(2, "gemm: unsupported shape M=%d N=%d K=%d (tile %dx%d)\n", M, N, K, BM, BN); return false; }
    const int nblk = ((M + BM - 1) / BM) * (N / BN);
    const size_t lds = 2 * (BM + BN) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)gemm_nt_kernel<BM, BN, Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, Epi>), dim3(nblk), dim3(256), lds, st, A, lda, W, ldw, M, N, K, epi);
    return hipGetLastError() == hipSuccess;
}

// Tile choice: big tiles when the grid still fills the chip, smaller ones for skinny problems.
template <class Epi>
static bool dispatch(const f16 *A, long lda, const f16 *W, long ldw, int M, int N, int K, const Epi &epi, hipStream_t st) {
    const long big = (long)((M + 127) / 128) * (N / 128);
    if (N % 128 == 0 && big >= 192) return launch<128, 128, Epi>(A, lda, W, ldw, M, N, K, epi, st);
    if (M > 64) return launch<128, 64, Epi>(A, lda, W, ldw, M, N, K, epi, st);
    return launch<64, 64, Epi>(A, lda, W, ldw, M, N, K, epi, st);
}

bool gemm_f16(const f16 *A, long lda, const f16 *W, long ldw, int M, int N, int K, f16 *out, long ldc, const float *bias, int gelu, float scale, hipStream_t st) {
    return dispatch(A, lda, W, ldw, M, N, K, EpiF16{out, ldc, bias, gelu, scale}, st);
}
bool gemm_f32(const f16 *A, long lda, const f16 *W, long ldw, int M, int N, int K, float *out, long ldc, const float *bias, hipStream_t st) {
    return dispatch(A, lda, W, ldw, M, N, K, EpiF32{out, ldc, bias}, st);
}
bool gemm_residual(const f16 *A, long lda, const f16 *W, long ldw, int M, int N, int K, float *x, long ldc, const float *bias, hipStream_t st) {
    return dispatch(A, lda, W, ldw, M, N, K, EpiResidual{x, ldc, bias}, st);
}
bool gemm_conv1(const f16 *melT, int n_mel, const f16 *W, int Kpad, int M, int d, f16 *h1, const float *bias, int T2, hipStream_t st) {
    return dispatch(melT, n_mel, W, Kpad, M, d, Kpad, EpiConv1{h1, d, bias, T2 + 2, T2}, st);
}
bool gemm_conv2(const f16 *h1, const f16 *W, int M, int d, float *x, const float *bias, const float *pe, int T, hipStream_t st) {
    return dispatch(h1, 2L * d, W, 3L * d, M, d, 3 * d, EpiConv2{x, d, bias, pe, T}, st);
}
bool gemm_enc_qkv(const f16 *xn, const f16 *W, int M, int d, int H, int T, int Tp, f16 *Q, f16 *Kb, f16 *Vt, const float *bias, hipStream_t st) {
    return dispatch(xn, d, W, d, M, 3 * d, d, EpiEncQKV{Q, Kb, Vt, bias, d, H, T, Tp}, st);
}
bool gemm_cross_kv(const f16 *enc, const f16 *W, int M, int d, int H, int T, int Tp, int n_layer, f16 *out, const float *bias, const int *slots, hipStream_t st) {
    return dispatch(enc, d, W, d, M, n_layer * 2 * d, d, EpiCrossKV{out, bias, slots, d, H, T, Tp, n_layer}, st);
}
bool gemm_dec_qkv(const f16 *xn, const f16 *W, int M, int d, int H, f16 *q, f16 *kc, f16 *vc, const float *bias, const DecRow *rows, hipStream_t st) {
    return dispatch(xn, d, W, d, M, 3 * d, d, EpiDecQKV{q, kc, vc, bias, rows, d, H}, st);
}

}  // namespace wmi
