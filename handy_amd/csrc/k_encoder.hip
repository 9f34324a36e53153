// Encoder-side kernels for gfx950: row LayerNorm and the non-causal multi-head attention.
#include <atomic>
#include "kernels.h"
#include <type_traits>

namespace wmi {

// ------------------------------------------------------------------------------------------ LayerNorm
// One wave per row, the row held in registers (up to 5 float4 per lane), two-pass statistics (mean, then centred variance)
// like the reference. Bound: HBM (4*d in, 2*d out per row). The waves are persistent: each strides over the rows and
// requests its next row before it reduces the current one, and the weights and bias sit in LDS. A wave that lives for one
// row spends most of its life being launched and waiting for its single request: 3.7 TB/s of the 6 bytes per element
// against 5.5 TB/s this way. STREAM marks the activations as read-once / written-once for the caches (large inputs only).
template <bool STREAM>
__global__ __launch_bounds__(256) void layer_norm_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ b,
                                                         f16 *__restrict__ out16, float *__restrict__ out32, int M, int d, const int *__restrict__ gather) {
    __shared__ f32x4 sw[320], sb[320];
    const int nch = d >> 2, lane = threadIdx.x & 63;
    for (int c = threadIdx.x; c < nch; c += 256) { sw[c] = *(const f32x4 *)(w + 4 * c); sb[c] = *(const f32x4 *)(b + 4 * c); }
    __syncthreads();
    const int stride = gridDim.x * 4;
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    f32x4 v[5], nx[5];
    auto load = [&](f32x4 *dst, int r) {
        const long src = gather ? gather[r] : r;
        const f32x4 *xr = (const f32x4 *)(x + src * d);
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int c = lane + 64 * i;
            dst[i] = f32x4{0, 0, 0, 0};
            if (c < nch) dst[i] = STREAM ? __builtin_nontemporal_load(xr + c) : xr[c];
        }
    };
    if (row < M) load(v, row);
    for (; row < M; row += stride) {
        if (row + stride < M) load(nx, row + stride);
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < 5; i++) sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float mean = sum / (float)d;
        float sq = 0.0f;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int c = lane + 64 * i;
            if (c < nch) { v[i] -= mean; sq += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]); }
        }
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
        const float scale = 1.0f / sqrtf(sq / (float)d + 1e-5f);
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int c = lane + 64 * i;
            if (c < nch) {
                const f32x4 ww = sw[c], bb = sb[c];
                f32x4 y;
#pragma unroll
                for (int k = 0; k < 4; k++) y[k] = v[i][k] * scale * ww[k] + bb[k];
                if (out16) {
                    const f16x4 r = f16x4{(f16)y[0], (f16)y[1], (f16)y[2], (f16)y[3]};
                    f16x4 *dst = (f16x4 *)(out16 + (long)row * d + 4 * c);
                    if (STREAM) __builtin_nontemporal_store(r, dst); else *dst = r;
                }
                if (out32) *(f32x4 *)(out32 + (long)row * d + 4 * c) = y;
            }
        }
#pragma unroll
        for (int i = 0; i < 5; i++) v[i] = nx[i];
    }
}

bool layer_norm(const float *x, const float *w, const float *b, f16 *out16, float *out32, int M, int d, const int *gather, hipStream_t st) {
    if (M <= 0) return true;
    if (d % 4 != 0 || d > 1280) { log_msg(2, "layer_norm: unsupported width %d\n", d); return false; }
    const int wgs = std::min((M + 3) / 4, 4096);            // 16 workgroups per CU's worth of rows in flight, then the waves stride
    if ((size_t)M * d * 4 >= ((size_t)256 << 20)) hipLaunchKernelGGL((layer_norm_kernel<true>), dim3(wgs), dim3(256), 0, st, x, w, b, out16, out32, M, d, gather);
    else hipLaunchKernelGGL((layer_norm_kernel<false>), dim3(wgs), dim3(256), 0, st, x, w, b, out16, out32, M, d, gather);
    return hipGetLastError() == hipSuccess;
}

#ifdef WMI_HARNESS
// Benchmark-only variants of the LayerNorm kernel (fp16 output only): V bit 0 = nontemporal loads and stores, bit 1 = weights and
// bias requested before the statistics, bit 2 = 8 rows per workgroup.
template <int V>
__global__ __launch_bounds__((V & 4) ? 512 : 256) void layer_norm_variant_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ b,
                                                                                  f16 *__restrict__ out16, int M, int d) {
    const int rows_wg = (V & 4) ? 8 : 4;
    const int row = blockIdx.x * rows_wg + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const f32x4 *xr = (const f32x4 *)(x + (long)row * d);
    const int nch = d >> 2;
    f32x4 v[5], ww[5], bb[5];
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const int c = lane + 64 * i;
        v[i] = f32x4{0, 0, 0, 0};
        if (c < nch) v[i] = (V & 1) ? __builtin_nontemporal_load(xr + c) : xr[c];
    }
    if (V & 2) {
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int c = lane + 64 * i;
            if (c < nch) { ww[i] = *(const f32x4 *)(w + 4 * c); bb[i] = *(const f32x4 *)(b + 4 * c); }
        }
    }
#pragma unroll
    for (int i = 0; i < 5; i++) sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)d;
    float sq = 0.0f;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const int c = lane + 64 * i;
        if (c < nch) { v[i] -= mean; sq += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]); }
    }
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float scale = 1.0f / sqrtf(sq / (float)d + 1e-5f);
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const int c = lane + 64 * i;
        if (c < nch) {
            if (!(V & 2)) { ww[i] = *(const f32x4 *)(w + 4 * c); bb[i] = *(const f32x4 *)(b + 4 * c); }
            f32x4 y;
#pragma unroll
            for (int k = 0; k < 4; k++) y[k] = v[i][k] * scale * ww[i][k] + bb[i][k];
            const f16x4 r = f16x4{(f16)y[0], (f16)y[1], (f16)y[2], (f16)y[3]};
            f16x4 *dst = (f16x4 *)(out16 + (long)row * d + 4 * c);
            if (V & 1) __builtin_nontemporal_store(r, dst); else *dst = r;
        }
    }
}

// Persistent variant: waves stride over the rows, weights and bias live in LDS; PF = the next row is requested before the current
// one is reduced.
template <bool PF>
__global__ __launch_bounds__(256) void layer_norm_persistent_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ b,
                                                                     f16 *__restrict__ out16, int M, int d) {
    __shared__ f32x4 sw[320], sb[320];
    const int nch = d >> 2, lane = threadIdx.x & 63;
    for (int c = threadIdx.x; c < nch; c += 256) { sw[c] = *(const f32x4 *)(w + 4 * c); sb[c] = *(const f32x4 *)(b + 4 * c); }
    __syncthreads();
    const int stride = gridDim.x * 4;
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    f32x4 v[5], nx[5];
    auto load = [&](f32x4 *dst, int r) {
        const f32x4 *xr = (const f32x4 *)(x + (long)r * d);
#pragma unroll
        for (int i = 0; i < 5; i++) { const int c = lane + 64 * i; dst[i] = f32x4{0, 0, 0, 0}; if (c < nch) dst[i] = __builtin_nontemporal_load(xr + c); }
    };
    if (row < M) load(v, row);
    for (; row < M; row += stride) {
        if (PF && row + stride < M) load(nx, row + stride);
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < 5; i++) sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float mean = sum / (float)d;
        float sq = 0.0f;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int c = lane + 64 * i;
            if (c < nch) { v[i] -= mean; sq += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]); }
        }
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
        const float scale = 1.0f / sqrtf(sq / (float)d + 1e-5f);
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int c = lane + 64 * i;
            if (c < nch) {
                const f32x4 ww = sw[c], bb = sb[c];
                f32x4 y;
#pragma unroll
                for (int k = 0; k < 4; k++) y[k] = v[i][k] * scale * ww[k] + bb[k];
                __builtin_nontemporal_store(f16x4{(f16)y[0], (f16)y[1], (f16)y[2], (f16)y[3]}, (f16x4 *)(out16 + (long)row * d + 4 * c));
            }
        }
        if (PF) {
#pragma unroll
            for (int i = 0; i < 5; i++) v[i] = nx[i];
        } else if (row + stride < M) load(v, row + stride);
    }
}

bool layer_norm_variant(int variant, const float *x, const float *w, const float *b, f16 *out16, int M, int d, hipStream_t st) {
    if (M <= 0 || d % 4 != 0 || d > 1280) return false;
#define WMI_LNV(V_) case V_: hipLaunchKernelGGL((layer_norm_variant_kernel<V_>), dim3((M + ((V_ & 4) ? 7 : 3)) / ((V_ & 4) ? 8 : 4)), dim3((V_ & 4) ? 512 : 256), 0, st, x, w, b, out16, M, d); break;
    switch (variant) { WMI_LNV(0) WMI_LNV(1) WMI_LNV(2) WMI_LNV(3) WMI_LNV(4) WMI_LNV(5) WMI_LNV(6) WMI_LNV(7)
        case 100: return layer_norm(x, w, b, out16, nullptr, M, d, nullptr, st);
        case 8: case 9: case 10: case 11: case 12: case 13: {
            const int wgs = 256 * (variant < 10 ? 8 : variant < 12 ? 4 : 16);
            if (variant & 1) hipLaunchKernelGGL((layer_norm_persistent_kernel<true>), dim3(wgs), dim3(256), 0, st, x, w, b, out16, M, d);
            else hipLaunchKernelGGL((layer_norm_persistent_kernel<false>), dim3(wgs), dim3(256), 0, st, x, w, b, out16, M, d);
            break;
        }
        default: return false; }
#undef WMI_LNV
    return hipGetLastError() == hipSuccess;
}

#endif  // WMI_HARNESS

// ------------------------------------------------------------------------------------------ attention
// Non-causal softmax(Q K^T) V over T keys, heads of 64. Q is pre-scaled by 1/8 (kEncQScale, exact in fp16).
//   S^T = K Q^T      v_mfma_f32_32x32x16_f16, K rows as the A operand: a lane then holds scores of
//                    ONE query (its column) for 16 keys per 32-key half, so the reference exponent and the row sum
//                    are per-lane scalars plus one exchange with the other lane half.
//   O^T += V^T P^T   P^T is taken straight from the S^T accumulators (converted to fp16) as the B
//                    operand; V arrives pre-transposed ([dh][enc_vt_pos(key)], written that way by the
//                    QKV GEMM epilogues), read in the k-order the accumulator layout implies.
// K and V^T tiles of 64 keys are staged through LDS by LDS-DMA (double-buffered, XOR-swizzled on the source side).
// Nominally bound by the matrix pipe (4*T*T*64 flop per head); measured, it is bound by instruction issue: per score a lane
// issues one fused multiply-add, one exponential (8 cycles), one add and half a conversion -- about 18 cycles -- and an MFMA
// (32 cycles of the matrix pipe, 8 of issue) covers two scores per lane. The output tile goes through LDS so rows are stored whole.

constexpr float kLog2e = 1.4426950408889634f;

// Workgroup = 4 waves = 256 query rows of one (utterance, head); each wave owns 64 rows as two blocks of 32 (A and B). A K or
// V^T fragment read from LDS feeds two MFMAs (one per block) and a staged tile serves 256 queries: half the LDS reads and
// LDS-DMA instructions per score of a 32-queries-per-wave layout (enc_attention_kernel in the harness build, 9 % slower).
// The keys are worked through in halves of a staged tile (32 keys): per half, Q K^T of block A, then Q K^T of B interleaved
// with A's exponentials, then V^T P^T of A interleaved with B's exponentials, then V^T P^T of B -- independent work of the
// other block under every MFMA group but the first and last, pinned with scheduling barriers.
// The MFMAs accumulate raw scores (C = 0) and the reference exponent enters the exponential's argument through the fused
// multiply-add that scales by log2(e): p = exp2(s * log2e - m * log2e), one rounding.
// The reference m of a query is the maximum of its first 32 keys. Later halves exponentiate straight away and look at the sum
// afterwards: while a lane's 16 values add up to less than kRedo = 2^15, each of them is below 2^15 -- fp16 holds that at full
// relative precision (its maximum is 65504) and the sums are fp32, so the result is the same softmax, rounded alike. Only when
// some query of the wave exceeds that does the wave read that block's scores again, move m up to the new maximum and rescale
// what it has summed (wave-uniform branch, rare).
__global__ __launch_bounds__(256, 2) void enc_attention_wide_kernel(const f16 *__restrict__ Q, const f16 *__restrict__ K, const f16 *__restrict__ Vt,
                                                                    f16 *__restrict__ att, int H, int T, int Tp) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 16384];    // per stage: K tile 8 KB + V^T tile 8 KB; the output tile at the end
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: LDS-DMA destinations need no VALU
    // Workgroup ids are dealt round-robin to the 8 XCDs (one private L2 each): give every XCD a contiguous run of
    // logical ids, so that the query blocks of one (utterance, head), which all stream the same K and V, share an L2.
    const int nqb = (Tp + 255) / 256, nblk = gridDim.x;
    const int xq = nblk >> 3, xr = nblk & 7, xcd = blockIdx.x & 7, xi = blockIdx.x >> 3;
    const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
    const int qblk = logical % nqb, bh_i = logical / nqb;
    if (qblk * 256 >= T) return;                  // query rows that are all padding
    const long bh = bh_i;
    const int head = bh_i % H, b = bh_i / H;
    const int q0 = qblk * 256 + wave * 64;
    f16x8 qf[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; qb++) {
        const int qrow = min(q0 + qb * 32 + r, Tp - 1);          // rows past the padded length: any row, never stored
        const f16 *qp = Q + (bh * Tp + qrow) * 64 + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) qf[qb][ks] = *(const f16x8 *)(qp + 16 * ks);
    }
    const f16 *kg = K + bh * Tp * 64;
    const f16 *vg = Vt + bh * 64 * Tp;
    // Staging by LDS-DMA, 16 bytes a lane: one wave-instruction moves a piece of 8 tile rows x 128 B to consecutive LDS
    // addresses, so the XOR swizzle of the 16-byte slots is applied on the source side (lane (row, slot) fetches the slot
    // that belongs there). A stage is 8 pieces of K rows and 8 of V^T rows; wave w moves pieces w and w + 4 of each.
    // V^T needs no shuffling on the way: the P operand of the PV MFMA holds keys {16s + 4h + b, 16s + 8 + 4h + b} (b < 4) in
    // one fragment, and enc_vt_pos() already put those two 4-key chunks side by side in global memory.
    const int srow = lane >> 3, sslot = lane & 7;
    int koff[2], voff[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int row = (wave + 4 * i) * 8 + srow, src_slot = sslot ^ ((row >> 1) & 7);
        koff[i] = row * 64 + src_slot * 8;
        voff[i] = row * Tp + src_slot * 8;
    }
    auto stage = [&](int kt, int buf) {
        char *base = smem + buf * 16384 + wave * 1024;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(kg + (long)kt * 4096 + koff[i]),
                                             (__attribute__((address_space(3))) void *)(base + i * 4096), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vg + (long)kt * 64 + voff[i]),
                                             (__attribute__((address_space(3))) void *)(base + 8192 + i * 4096), 16, 0, 0);
        }
    };
    // every piece of a stage has landed (each wave waits for its own, the barrier joins them) and every wave is done reading
    // the other stage
    auto stage_barrier = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    // byte offset of this lane's fragment j (16-byte chunk 2 j + h, swizzled) inside its tile row: K fragments ks = j of key row
    // kh * 32 + r, V^T fragments of key chunk 2 j + h = 4 kh + 2 sidx + h of row blk * 32 + r
    int foff[4];
#pragma unroll
    for (int j = 0; j < 4; j++) foff[j] = r * 128 + (((2 * j + h) ^ ((r >> 1) & 7)) << 4);

    f32x16 o[2][2], zero;                        // o[query block][dh block]
#pragma unroll
    for (int i = 0; i < 16; i++) { o[0][0][i] = 0.0f; o[0][1][i] = 0.0f; o[1][0][i] = 0.0f; o[1][1][i] = 0.0f; zero[i] = 0.0f; }
    float m_run[2] = {0.0f, 0.0f}, nm[2] = {0.0f, 0.0f}, l_run[2] = {0.0f, 0.0f};      // nm = -m_run * log2(e)
    const int n_tiles = (T + 63) / 64, n_half = (T + 31) / 32;
    const bool ragged = (T & 31) != 0;
    constexpr float kRedo = 32768.0f;

    // One half of a staged tile: keys kt * 64 + KH * 32 .. + 31. FIRST: the very first half (sets the references); MASK: it
    // holds keys past T.
    auto half = [&](int kt, const char *kb, const char *vb, auto kh_tag, auto first_tag, auto mask_tag) __attribute__((always_inline)) {
        constexpr int KH = decltype(kh_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value, MASK = decltype(mask_tag)::value;
        f16x8 kf[4], vf[2][2];
        f32x16 sa, sb;
        f16x8 pa[2], pb[2];
        float lsum_a = 0.0f, lsum_b = 0.0f;
        auto read_k = [&]() {
#pragma unroll
            for (int ks = 0; ks < 4; ks++) kf[ks] = *(const f16x8 *)(kb + KH * 4096 + foff[ks]);
        };
        auto read_v = [&]() {
#pragma unroll
            for (int sidx = 0; sidx < 2; sidx++)
#pragma unroll
                for (int blk = 0; blk < 2; blk++) vf[sidx][blk] = *(const f16x8 *)(vb + blk * 4096 + foff[KH * 2 + sidx]);
        };
        auto mask = [&](f32x16 &s) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int key = kt * 64 + KH * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (key >= T) s[i] = -INFINITY;
            }
        };
        auto qk = [&](f32x16 &s, int qb) {
#pragma unroll
            for (int ks = 0; ks < 4; ks++) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qf[qb][ks], ks == 0 ? zero : s, 0, 0, 0);
            if (MASK) mask(s);
        };
        auto exps = [&](const f32x16 &s, float nmq, float &lsum, f16x8 (&pf)[2], auto i0_tag, auto n_tag) {
            constexpr int i0 = decltype(i0_tag)::value, n = decltype(n_tag)::value;
#pragma unroll
            for (int e = 0; e < n; e++) {
                const int i = i0 + e;
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[i], kLog2e, nmq));
                lsum += p;
                pf[i >> 3][i & 7] = (f16)p;
            }
        };
        auto exps_all = [&](const f32x16 &s, float nmq, float &lsum, f16x8 (&pf)[2]) {
            lsum = 0.0f;
            exps(s, nmq, lsum, pf, std::integral_constant<int, 0>{}, std::integral_constant<int, 16>{});
        };
        // the reference of block qb moves up to the maximum of s (first: is set to it); what has been summed is rescaled
        auto move_reference = [&](const f32x16 &s, int qb, bool first) {
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; i++) mx = fmaxf(mx, s[i]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = first ? mx : fmaxf(mx, m_run[qb]);
            if (!first) {
                const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * kLog2e);
                l_run[qb] *= alpha;
#pragma unroll
                for (int i = 0; i < 16; i++) { o[qb][0][i] *= alpha; o[qb][1][i] *= alpha; }
            }
            m_run[qb] = m_new;
            nm[qb] = -m_new * kLog2e;
        };
        auto redo = [&](f32x16 &s, int qb, float &lsum, f16x8 (&pf)[2]) {       // wave-uniform, rare
            asm volatile("" ::: "memory");
            read_k();
            qk(s, qb);
            move_reference(s, qb, false);
            exps_all(s, nm[qb], lsum, pf);
        };
        auto pv = [&](int qb, const f16x8 (&pf)[2], auto g_tag) {
            constexpr int g = decltype(g_tag)::value, sidx = g >> 1, blk = g & 1;
            o[qb][blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[sidx][blk], pf[sidx], o[qb][blk], 0, 0, 0);
        };

        read_k();
        read_v();
        qk(sa, 0);
        if constexpr (FIRST) {
            qk(sb, 1);
            move_reference(sa, 0, true);
            move_reference(sb, 1, true);
            exps_all(sa, nm[0], lsum_a, pa);
        } else {
            // Q K^T of block B, one MFMA per gap, under block A's exponentials (letting the first of these MFMAs go ahead of
            // the exponentials, so that the wave has something to issue while block A's last MFMA finishes, was 1.5 % slower;
            // reading the next half's K fragments one half ahead, under V^T P^T of block B, changed nothing: with two waves on
            // a SIMD neither wait is what the SIMD's issue slots are lost to; eight waves and 512 queries per workgroup, which
            // halves the LDS-DMA instructions per query once more, was 3.6 % slower: one workgroup per CU, its waves in step)
            __builtin_amdgcn_sched_barrier(0);
            sb = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qf[1][0], zero, 0, 0, 0);
            exps(sa, nm[0], lsum_a, pa, std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
            __builtin_amdgcn_sched_barrier(0);
            sb = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[1], qf[1][1], sb, 0, 0, 0);
            exps(sa, nm[0], lsum_a, pa, std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{});
            __builtin_amdgcn_sched_barrier(0);
            sb = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[2], qf[1][2], sb, 0, 0, 0);
            exps(sa, nm[0], lsum_a, pa, std::integral_constant<int, 8>{}, std::integral_constant<int, 4>{});
            __builtin_amdgcn_sched_barrier(0);
            sb = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[3], qf[1][3], sb, 0, 0, 0);
            exps(sa, nm[0], lsum_a, pa, std::integral_constant<int, 12>{}, std::integral_constant<int, 4>{});
            __builtin_amdgcn_sched_barrier(0);
            if (MASK) mask(sb);
            asm volatile("" : "+v"(lsum_a), "+v"(pa[0]), "+v"(pa[1]));
            if (__builtin_amdgcn_ballot_w64(lsum_a >= kRedo) != 0) redo(sa, 0, lsum_a, pa);
        }
        l_run[0] += lsum_a;
        // V^T P^T of block A under block B's exponentials
        __builtin_amdgcn_sched_barrier(0);
        pv(0, pa, std::integral_constant<int, 0>{});
        exps(sb, nm[1], lsum_b, pb, std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
        __builtin_amdgcn_sched_barrier(0);
        pv(0, pa, std::integral_constant<int, 1>{});
        exps(sb, nm[1], lsum_b, pb, std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{});
        __builtin_amdgcn_sched_barrier(0);
        pv(0, pa, std::integral_constant<int, 2>{});
        exps(sb, nm[1], lsum_b, pb, std::integral_constant<int, 8>{}, std::integral_constant<int, 4>{});
        __builtin_amdgcn_sched_barrier(0);
        pv(0, pa, std::integral_constant<int, 3>{});
        exps(sb, nm[1], lsum_b, pb, std::integral_constant<int, 12>{}, std::integral_constant<int, 4>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!FIRST) {
            asm volatile("" : "+v"(lsum_b), "+v"(pb[0]), "+v"(pb[1]));
            if (__builtin_amdgcn_ballot_w64(lsum_b >= kRedo) != 0) redo(sb, 1, lsum_b, pb);
        }
        l_run[1] += lsum_b;
        pv(1, pb, std::integral_constant<int, 0>{}); pv(1, pb, std::integral_constant<int, 1>{});
        pv(1, pb, std::integral_constant<int, 2>{}); pv(1, pb, std::integral_constant<int, 3>{});
    };
    using No = std::false_type; using Yes = std::true_type;
    using H0 = std::integral_constant<int, 0>; using H1 = std::integral_constant<int, 1>;
    auto tile = [&](int kt, auto first_tag, auto last_tag) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_tag)::value;
        const char *kb = smem + (kt & 1) * 16384, *vb = kb + 8192;
        if (kt + 1 < n_tiles) stage(kt + 1, (kt + 1) & 1);
        if constexpr (!LAST) {
            half(kt, kb, vb, H0{}, first_tag, No{});
            half(kt, kb, vb, H1{}, No{}, No{});
        } else if (n_half & 1) {                 // the tile's second half is all padding
            if (ragged) half(kt, kb, vb, H0{}, first_tag, Yes{}); else half(kt, kb, vb, H0{}, first_tag, No{});
        } else {
            half(kt, kb, vb, H0{}, first_tag, No{});
            if (ragged) half(kt, kb, vb, H1{}, No{}, Yes{}); else half(kt, kb, vb, H1{}, No{}, No{});
        }
        stage_barrier();
    };
    stage(0, 0);
    stage_barrier();
    if (n_tiles == 1) tile(0, Yes{}, Yes{});
    else {
        tile(0, Yes{}, No{});
        for (int kt = 1; kt + 1 < n_tiles; kt++) tile(kt, No{}, No{});
        tile(n_tiles - 1, No{}, Yes{});
    }
    // normalise, stage the [256 q][64 dh] tile in LDS, store whole rows
    char *ob = smem;
#pragma unroll
    for (int qb = 0; qb < 2; qb++) {
        const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32);
        const float inv = 1.0f / l_tot;
        const int orow = wave * 64 + qb * 32 + r;
#pragma unroll
        for (int blk = 0; blk < 2; blk++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const f32x16 &ov = o[qb][blk];
                const f16x4 v4 = {(f16)(ov[4 * g] * inv), (f16)(ov[4 * g + 1] * inv), (f16)(ov[4 * g + 2] * inv), (f16)(ov[4 * g + 3] * inv)};
                *(f16x4 *)(ob + orow * 128 + (((4 * blk + g) ^ (orow & 7)) << 4) + 8 * h) = v4;
            }
    }
    __syncthreads();
    const int d = H * 64;
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const int idx = p * 256 + tid, row = idx >> 3, ch = idx & 7, q = qblk * 256 + row;
        if (q < T) *(f16x8 *)(att + ((long)b * T + q) * d + head * 64 + ch * 8) = *(const f16x8 *)(ob + row * 128 + ((ch ^ (row & 7)) << 4));
    }
}

#ifdef WMI_HARNESS
// The 32-queries-per-wave layout the product kernel replaced (128 queries per workgroup, three waves per SIMD, whole 64-key
// tiles, the reference exponent as the C operand of a tile's first MFMA): kept for A/B runs (scripts/enc_attention_ab.py).
__global__ __launch_bounds__(256) void enc_attention_kernel(const f16 *__restrict__ Q, const f16 *__restrict__ K, const f16 *__restrict__ Vt,
                                                            f16 *__restrict__ att, int H, int T, int Tp) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 16384];    // per stage: K tile 8 KB + V^T tile 8 KB
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: LDS-DMA destinations need no VALU
    // Workgroup ids are dealt round-robin to the 8 XCDs (one private L2 each): give every XCD a contiguous run of
    // logical ids, so that the query blocks of one (utterance, head), which all stream the same K and V, share an L2.
    const int nqb = Tp / 128, nblk = gridDim.x;
    const int xq = nblk >> 3, xr = nblk & 7, xcd = blockIdx.x & 7, xi = blockIdx.x >> 3;
    const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
    const int qblk = logical % nqb, bh_i = logical / nqb;
    if (qblk * 128 >= T) return;                  // query rows that are all padding
    const long bh = bh_i;
    const int head = bh_i % H, b = bh_i / H;
    const int q0 = qblk * 128 + wave * 32;
    const f16 *qp = Q + (bh * Tp + q0 + r) * 64 + 8 * h;
    f16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) qf[ks] = *(const f16x8 *)(qp + 16 * ks);

    const f16 *kg = K + bh * Tp * 64;
    const f16 *vg = Vt + bh * 64 * Tp;
    // Staging by LDS-DMA, 16 bytes a lane: one wave-instruction moves a piece of 8 tile rows x 128 B to consecutive LDS
    // addresses, so the XOR swizzle of the 16-byte slots is applied on the source side (lane (row, slot) fetches the slot
    // that belongs there). A stage is 8 pieces of K rows and 8 of V^T rows; wave w moves pieces w and w + 4 of each.
    // V^T needs no shuffling on the way: the P operand of the PV MFMA holds keys {16s + 4h + b, 16s + 8 + 4h + b} (b < 4) in
    // one fragment, and enc_vt_pos() already put those two 4-key chunks side by side in global memory.
    const int srow = lane >> 3, sslot = lane & 7;
    int koff[2], voff[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int row = (wave + 4 * i) * 8 + srow, src_slot = sslot ^ ((row >> 1) & 7);
        koff[i] = row * 64 + src_slot * 8;
        voff[i] = row * Tp + src_slot * 8;
    }
    auto stage = [&](int kt, int buf) {
        char *base = smem + buf * 16384 + wave * 1024;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(kg + (long)kt * 4096 + koff[i]),
                                             (__attribute__((address_space(3))) void *)(base + i * 4096), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vg + (long)kt * 64 + voff[i]),
                                             (__attribute__((address_space(3))) void *)(base + 8192 + i * 4096), 16, 0, 0);
        }
    };
    // every piece of a stage has landed (each wave waits for its own, the barrier joins them) and every wave is done reading
    // the other stage
    auto stage_barrier = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    f32x16 o0, o1, minit;                         // minit = -m_run in all 16 elements: the C operand of a tile's first MFMA
#pragma unroll
    for (int i = 0; i < 16; i++) { o0[i] = 0.0f; o1[i] = 0.0f; minit[i] = 0.0f; }
    float m_run = 0.0f, l_run = 0.0f;             // m_run: the query's reference score, set by tile 0
    const int n_tiles = (T + 63) / 64;
    stage(0, 0);
    stage_barrier();
    // One key tile. The accumulators start at -m_run, so the MFMAs leave s - m_run, and p = 2^(log2(e) (s - m_run)) is one
    // multiply and one exponential per score. The reference m_run is the maximum of
    // tile 0. Later tiles exponentiate straight away and look at the sum afterwards: while a lane's 32 values add up to less
    // than kRedo = 2^15, each of them is below 2^15 -- fp16 holds that at full relative precision (its maximum is 65504) and
    // the sums are fp32, so the result is the same softmax, rounded alike. Only when some query of the wave exceeds that does
    // the wave compute the tile's scores again, move the reference to the new maximum and rescale what it has summed.
    // MASK = the tile holds keys past T (only ever the last one); FIRST = tile 0.
    constexpr float kRedo = 32768.0f;
    auto tile = [&](int kt, auto mask_tag, auto first_tag) {
        constexpr bool MASK = decltype(mask_tag)::value, FIRST = decltype(first_tag)::value;
        const char *kb = smem + (kt & 1) * 16384, *vb = kb + 8192;
        if (kt + 1 < n_tiles) stage(kt + 1, (kt + 1) & 1);
        f32x16 s[2];
        auto scores = [&]() {
#pragma unroll
            for (int kh = 0; kh < 2; kh++) {
                const int row = kh * 32 + r, sw = (row >> 1) & 7;
#pragma unroll
                for (int ks = 0; ks < 4; ks++) {
                    const f16x8 kf = *(const f16x8 *)(kb + row * 128 + (((2 * ks + h) ^ sw) << 4));
                    if (ks == 0) s[kh] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], minit, 0, 0, 0);
                    else s[kh] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[kh], 0, 0, 0);
                }
            }
            if (MASK) {
#pragma unroll
                for (int kh = 0; kh < 2; kh++)
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const int key = kt * 64 + kh * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                        if (key >= T) s[kh][i] = -INFINITY;
                    }
            }
        };
        float lsum;
        f16x8 pf[2][2];
        auto exponentials = [&]() {
            lsum = 0.0f;
#pragma unroll
            for (int kh = 0; kh < 2; kh++)
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const float p = __builtin_amdgcn_exp2f(s[kh][i] * kLog2e);
                    lsum += p;
                    pf[kh][i >> 3][i & 7] = (f16)p;
                }
        };
        scores();
        bool move = FIRST;
        if (!FIRST) {
            exponentials();
            move = __builtin_amdgcn_ballot_w64(lsum >= kRedo) != 0;          // wave-uniform
            if (move) {
                asm volatile("" ::: "memory");                                // a second read of the tile, not 32 more live registers
                scores();
            }
        }
        if (move) {
            float mx = -INFINITY;
#pragma unroll
            for (int kh = 0; kh < 2; kh++)
#pragma unroll
                for (int i = 0; i < 16; i++) mx = fmaxf(mx, s[kh][i]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float delta = FIRST ? mx : fmaxf(mx, 0.0f);      // queries whose maximum did not grow keep theirs (alpha == 1)
#pragma unroll
            for (int kh = 0; kh < 2; kh++)
#pragma unroll
                for (int i = 0; i < 16; i++) s[kh][i] -= delta;
            m_run += delta;
#pragma unroll
            for (int i = 0; i < 16; i++) minit[i] = -m_run;
            if (!FIRST) {
                const float alpha = __builtin_amdgcn_exp2f(-delta * kLog2e);
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < 16; i++) { o0[i] *= alpha; o1[i] *= alpha; }
            }
            exponentials();
        }
        l_run += lsum;
#pragma unroll
        for (int kh = 0; kh < 2; kh++)
#pragma unroll
            for (int sidx = 0; sidx < 2; sidx++) {
                const int p16 = kh * 4 + 2 * sidx + h;
#pragma unroll
                for (int blk = 0; blk < 2; blk++) {
                    const int row = blk * 32 + r;
                    const f16x8 vf = *(const f16x8 *)(vb + row * 128 + ((p16 ^ ((row >> 1) & 7)) << 4));
                    if (blk == 0) o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kh][sidx], o0, 0, 0, 0);
                    else o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kh][sidx], o1, 0, 0, 0);
                }
            }
        stage_barrier();
    };
    if (n_tiles == 1) {
        if (T & 63) tile(0, std::true_type{}, std::true_type{}); else tile(0, std::false_type{}, std::true_type{});
    } else {
        tile(0, std::false_type{}, std::true_type{});
        for (int kt = 1; kt + 1 < n_tiles; kt++) tile(kt, std::false_type{}, std::false_type{});
        if (T & 63) tile(n_tiles - 1, std::true_type{}, std::false_type{}); else tile(n_tiles - 1, std::false_type{}, std::false_type{});
    }
    // normalise, stage the [128 q][64 dh] tile in LDS, store whole rows
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    char *ob = smem;
#pragma unroll
    for (int blk = 0; blk < 2; blk++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const f32x16 &o = blk ? o1 : o0;
            const f16x4 v4 = {(f16)(o[4 * g] * inv), (f16)(o[4 * g + 1] * inv), (f16)(o[4 * g + 2] * inv), (f16)(o[4 * g + 3] * inv)};
            // 16-byte chunk (4 blk + g) of row (wave * 32 + r) sits at chunk ^ (row & 7): the 32 lanes of a half-wave write one
            // column of 32 different rows, which unswizzled is the same four banks 32 times over
            const int orow = wave * 32 + r;
            *(f16x4 *)(ob + orow * 128 + (((4 * blk + g) ^ (orow & 7)) << 4) + 8 * h) = v4;
        }
    __syncthreads();
    const int d = H * 64;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int idx = p * 256 + tid, row = idx >> 3, ch = idx & 7, q = qblk * 128 + row;
        if (q < T) *(f16x8 *)(att + ((long)b * T + q) * d + head * 64 + ch * 8) = *(const f16x8 *)(ob + row * 128 + ((ch ^ (row & 7)) << 4));
    }
}



// enc_attention_kernel (above), software-pipelined inside each wave. Iteration t of a wave works on three key tiles at once: the matrix
// pipe multiplies Q K^T of tile t + 1 and V^T P^T of tile t - 1 (16 MFMAs, none of which depends on anything this iteration
// computes) while the vector pipe exponentiates tile t's scores (32 per lane: multiply, exponential, add, convert), the two
// streams interleaved by hand -- one MFMA, then two scores' worth of vector work, pinned with scheduling barriers -- so that
// the matrix pipe's 32 cycles per MFMA run under the vector instructions instead of beside another wave's. In that kernel
// the overlap depends on another wave of the SIMD being in the other phase, and the per-tile barrier keeps a workgroup's
// waves in step. Costs a second set of score and probability registers (two waves per SIMD instead of three) and a third K
// stage in LDS; V^T stays double-buffered (V^T of tile t is requested in iteration t and multiplied in t + 1).
// The arithmetic is that kernel's, operation for operation: same accumulation chains, same order of the partial sums,
// and a tile that moves the reference m_run (rare) recomputes its own scores and those of the tile after it from the K tiles
// still in the ring, with the new -m_run as the accumulators' starting value. The outputs are bit-identical.
// ABL (harness only, wrong results): 1 no exponential, 2 no MFMA in the gaps, 3 no workgroup barrier, 4 no LDS fragment reads,
// 5 no K / V^T staging, 6 staging never waited for, 7 = 2 + 5
template <int ABL>
__global__ __launch_bounds__(256, 2) void enc_attention_pipe_kernel(const f16 *__restrict__ Q, const f16 *__restrict__ K, const f16 *__restrict__ Vt,
                                                                    f16 *__restrict__ att, int H, int T, int Tp) {
    __shared__ __attribute__((aligned(16))) char smem[5 * 8192];    // K tiles: 3 stages of 8 KB; V^T tiles: 2 stages of 8 KB
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nqb = Tp / 128, nblk = gridDim.x;
    const int xq = nblk >> 3, xr = nblk & 7, xcd = blockIdx.x & 7, xi = blockIdx.x >> 3;
    const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
    const int qblk = logical % nqb, bh_i = logical / nqb;
    if (qblk * 128 >= T) return;
    const long bh = bh_i;
    const int head = bh_i % H, b = bh_i / H;
    const int q0 = qblk * 128 + wave * 32;
    const f16 *qp = Q + (bh * Tp + q0 + r) * 64 + 8 * h;
    f16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) qf[ks] = *(const f16x8 *)(qp + 16 * ks);
    const f16 *kg = K + bh * Tp * 64;
    const f16 *vg = Vt + bh * 64 * Tp;
    const int srow = lane >> 3, sslot = lane & 7;
    int koff[2], voff[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int row = (wave + 4 * i) * 8 + srow, src_slot = sslot ^ ((row >> 1) & 7);
        koff[i] = row * 64 + src_slot * 8;
        voff[i] = row * Tp + src_slot * 8;
    }
    auto stage_k = [&](int kt, int slot) {
        if (ABL == 5 || ABL == 7) return;
        char *base = smem + slot * 8192 + wave * 1024;
#pragma unroll
        for (int i = 0; i < 2; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(kg + (long)kt * 4096 + koff[i]),
                                             (__attribute__((address_space(3))) void *)(base + i * 4096), 16, 0, 0);
    };
    auto stage_v = [&](int kt) {
        if (ABL == 5 || ABL == 7) return;
        char *base = smem + (3 + (kt & 1)) * 8192 + wave * 1024;
#pragma unroll
        for (int i = 0; i < 2; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vg + (long)kt * 64 + voff[i]),
                                             (__attribute__((address_space(3))) void *)(base + i * 4096), 16, 0, 0);
    };
    auto stage_barrier = [&]() {
        if (ABL != 6) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ABL != 3) __syncthreads();
    };
    // byte offset of this lane's fragment j (16-byte chunk 2 j + h, swizzled) inside its tile row: the same four offsets serve
    // the K fragments (ks = j) and the V^T fragments (key chunk 2 j + h = 4 kh + 2 sidx + h); rows kh * 32 + r / blk * 32 + r
    int foff[4];
#pragma unroll
    for (int j = 0; j < 4; j++) foff[j] = r * 128 + (((2 * j + h) ^ ((r >> 1) & 7)) << 4);
    f32x16 o0, o1, minit;
#pragma unroll
    for (int i = 0; i < 16; i++) { o0[i] = 0.0f; o1[i] = 0.0f; minit[i] = 0.0f; }
    float m_run = 0.0f, l_run = 0.0f;
    const int n_tiles = (T + 63) / 64;
    const bool ragged = (T & 63) != 0;
    constexpr float kRedo = 32768.0f;
    float lsum;

    // scores of tile kt (its K in ring slot kt % 3) relative to the current reference, all at once: tile 0 and the rare path
    auto scores = [&](f32x16 (&s)[2], int kt, int slot, bool mask) {
        const char *kb = smem + slot * 8192;
#pragma unroll
        for (int kh = 0; kh < 2; kh++)
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                const f16x8 kf = *(const f16x8 *)(kb + kh * 4096 + foff[ks]);
                if (ks == 0) s[kh] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], minit, 0, 0, 0);
                else s[kh] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[kh], 0, 0, 0);
            }
        if (mask) {
#pragma unroll
            for (int kh = 0; kh < 2; kh++)
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int key = kt * 64 + kh * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (key >= T) s[kh][i] = -INFINITY;
                }
        }
    };
    auto exponentials = [&](const f32x16 (&s)[2], f16x8 (&pf)[2][2]) {
        lsum = 0.0f;
#pragma unroll
        for (int kh = 0; kh < 2; kh++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const float p = __builtin_amdgcn_exp2f(s[kh][i] * kLog2e);
                lsum += p;
                pf[kh][i >> 3][i & 7] = (f16)p;
            }
    };
    auto move_reference = [&](f32x16 (&s)[2], bool first) {
        float mx = -INFINITY;
#pragma unroll
        for (int kh = 0; kh < 2; kh++)
#pragma unroll
            for (int i = 0; i < 16; i++) mx = fmaxf(mx, s[kh][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float delta = first ? mx : fmaxf(mx, 0.0f);
#pragma unroll
        for (int kh = 0; kh < 2; kh++)
#pragma unroll
            for (int i = 0; i < 16; i++) s[kh][i] -= delta;
        m_run += delta;
#pragma unroll
        for (int i = 0; i < 16; i++) minit[i] = -m_run;
        if (!first) {
            const float alpha = __builtin_amdgcn_exp2f(-delta * kLog2e);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 16; i++) { o0[i] *= alpha; o1[i] *= alpha; }
        }
    };
    // the V^T P^T product of one tile, all at once: the last tile, after the loop
    auto pv_all = [&](const f16x8 (&pf)[2][2], int kt) {
        const char *vb = smem + (3 + (kt & 1)) * 8192;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const f16x8 v0 = *(const f16x8 *)(vb + foff[j]), v1 = *(const f16x8 *)(vb + 4096 + foff[j]);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, pf[j >> 1][j & 1], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, pf[j >> 1][j & 1], o1, 0, 0, 0);
        }
    };

    // One iteration, t >= 1: sixteen gaps, each one MFMA (first the eight of V^T P^T of tile t - 1, then the eight of Q K^T of
    // tile t + 1), the LDS read of the fragment three MFMAs ahead, and two of tile t's scores; nothing crosses a gap boundary.
    // The last iteration multiplies a tile t + 1 that does not exist (whatever its ring slot holds; the scores are never looked
    // at): 8 of a wave's ~400 MFMAs, cheaper than a second copy of the loop body.
    auto body = [&](int t, int slot_cur, int slot_nxt, f32x16 (&sc)[2], f32x16 (&sn)[2], f16x8 (&pfp)[2][2], f16x8 (&pfc)[2][2]) {
        const char *kb = smem + slot_nxt * 8192;
        const char *vb = smem + (3 + ((t - 1) & 1)) * 8192;
        // fragment of MFMA g: g < 8: V^T rows (g & 1) * 32 + r, key chunk pair g >> 1; g >= 8: K rows ((g - 8) >> 2) * 32 + r, ks = g & 3
        f16x8 frag[16];
        auto fetch = [&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value;
            if constexpr (ABL == 4) { if constexpr (g < 16) frag[g] = qf[g & 3]; }
            else if constexpr (g < 8) frag[g] = *(const f16x8 *)(vb + (g & 1) * 4096 + foff[g >> 1]);
            else if constexpr (g < 16) frag[g] = *(const f16x8 *)(kb + ((g - 8) >> 2) * 4096 + foff[g & 3]);
        };
        auto gap = [&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value;
            if constexpr (ABL == 2 || ABL == 7) {
                if constexpr (g < 8) o0[g] += (float)frag[g][0]; else sn[(g - 8) >> 2][g & 3] = (float)frag[g][0];
            } else if constexpr (g < 8) {
                if constexpr (g & 1) o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag[g], pfp[g >> 2][(g >> 1) & 1], o1, 0, 0, 0);
                else o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag[g], pfp[g >> 2][(g >> 1) & 1], o0, 0, 0, 0);
            } else {
                constexpr int kh = (g - 8) >> 2, ks = g & 3;
                if constexpr (ks == 0) sn[kh] = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag[g], qf[ks], minit, 0, 0, 0);
                else sn[kh] = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag[g], qf[ks], sn[kh], 0, 0, 0);
            }
            fetch(std::integral_constant<int, g + 3>{});
#pragma unroll
            for (int e = 0; e < 2; e++) {
                constexpr int kh = g >> 3;
                const int i = (2 * g + e) & 15;
                const float p = ABL == 1 ? sc[kh][i] * kLog2e : __builtin_amdgcn_exp2f(sc[kh][i] * kLog2e);
                lsum += p;
                pfc[kh][i >> 3][i & 7] = (f16)p;
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        if (t + 2 < n_tiles) stage_k(t + 2, slot_cur == 0 ? 2 : slot_cur - 1);         // the slot tile t - 1 has left
        stage_v(t);
        lsum = 0.0f;
        fetch(std::integral_constant<int, 0>{}); fetch(std::integral_constant<int, 1>{}); fetch(std::integral_constant<int, 2>{});
        __builtin_amdgcn_sched_barrier(0);
        gap(std::integral_constant<int, 0>{}); gap(std::integral_constant<int, 1>{}); gap(std::integral_constant<int, 2>{});
        gap(std::integral_constant<int, 3>{}); gap(std::integral_constant<int, 4>{}); gap(std::integral_constant<int, 5>{});
        gap(std::integral_constant<int, 6>{}); gap(std::integral_constant<int, 7>{}); gap(std::integral_constant<int, 8>{});
        gap(std::integral_constant<int, 9>{}); gap(std::integral_constant<int, 10>{}); gap(std::integral_constant<int, 11>{});
        gap(std::integral_constant<int, 12>{}); gap(std::integral_constant<int, 13>{}); gap(std::integral_constant<int, 14>{});
        gap(std::integral_constant<int, 15>{});
        // the probabilities and their sum exist here, not wherever their first reader is (code sinking would carry the
        // exponentials out of the gaps, into the blocks after the branches below)
        asm volatile("" : "+v"(lsum), "+v"(pfc[0][0]), "+v"(pfc[0][1]), "+v"(pfc[1][0]), "+v"(pfc[1][1]));
        const bool mask_next = ragged && t + 2 == n_tiles;
        if (mask_next) {
#pragma unroll
            for (int kh = 0; kh < 2; kh++)
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int key = (t + 1) * 64 + kh * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (key >= T) sn[kh][i] = -INFINITY;
                }
        }
        if (__builtin_amdgcn_ballot_w64(lsum >= kRedo) != 0) {               // wave-uniform, rare
            asm volatile("" ::: "memory");
            scores(sc, t, slot_cur, ragged && t + 1 == n_tiles);
            move_reference(sc, false);
            exponentials(sc, pfc);
            if (t + 1 < n_tiles) scores(sn, t + 1, slot_nxt, mask_next);
        }
        l_run += lsum;
        stage_barrier();
    };

    // tile 0 sets the reference; its exponentials and tile 1's scores one after the other
    stage_k(0, 0);
    if (n_tiles > 1) stage_k(1, 1);
    stage_v(0);
    stage_barrier();
    f32x16 sa[2], sb[2];
    f16x8 pa[2][2], pb[2][2];
    scores(sa, 0, 0, ragged && n_tiles == 1);
    move_reference(sa, true);
    if (n_tiles > 2) stage_k(2, 2);
    if (n_tiles > 1) scores(sb, 1, 1, ragged && n_tiles == 2);
    exponentials(sa, pa);
    l_run += lsum;
    stage_barrier();
    // iteration t: K of tile t in ring slot t % 3. Odd t: scores in sb, probabilities into pb; even t: sa, pa.
    int t = 1, slot = 1;
    for (; t + 1 < n_tiles; t += 2) {
        body(t, slot, slot == 2 ? 0 : slot + 1, sb, sa, pa, pb);
        slot = slot == 2 ? 0 : slot + 1;
        body(t + 1, slot, slot == 2 ? 0 : slot + 1, sa, sb, pb, pa);
        slot = slot == 2 ? 0 : slot + 1;
    }
    if (t < n_tiles) {
        body(t, slot, slot == 2 ? 0 : slot + 1, sb, sa, pa, pb);
        pv_all(pb, t);
    } else {
        pv_all(pa, n_tiles - 1);
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    char *ob = smem;
#pragma unroll
    for (int blk = 0; blk < 2; blk++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const f32x16 &o = blk ? o1 : o0;
            const f16x4 v4 = {(f16)(o[4 * g] * inv), (f16)(o[4 * g + 1] * inv), (f16)(o[4 * g + 2] * inv), (f16)(o[4 * g + 3] * inv)};
            const int orow = wave * 32 + r;
            *(f16x4 *)(ob + orow * 128 + (((4 * blk + g) ^ (orow & 7)) << 4) + 8 * h) = v4;
        }
    __syncthreads();
    const int d = H * 64;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int idx = p * 256 + tid, row = idx >> 3, ch = idx & 7, q = qblk * 128 + row;
        if (q < T) *(f16x8 *)(att + ((long)b * T + q) * d + head * 64 + ch * 8) = *(const f16x8 *)(ob + row * 128 + ((ch ^ (row & 7)) << 4));
    }
}

#endif  // WMI_HARNESS (enc_attention_pipe_kernel)

#ifdef WMI_HARNESS
// Micro-benchmark (harness only): time of 16-byte LDS writes / reads (four per iteration, 8 workgroups per CU) under a lane ->
// (row, slot) pattern; rows are 128 bytes.
//   0 linear (lane l -> byte 16 l)            1 one column: row l, slot 0
//   2 four lanes per row, slots 0,2,4,6, swizzle (row >> 1) & 7 (the attention kernel's tile writes)
//   3 as 2 with swizzle row & 7               4 32 rows per half-wave, slot = lane >> 5, swizzle (row >> 1) & 7 (its fragment reads)
//   5 as 4 with swizzle row & 7               6 16 rows per quarter-wave, slot = lane >> 4, swizzle row & 7 (the GEMM's fragment reads)
__global__ __launch_bounds__(256) void lds_pattern_kernel(int pattern, int do_read, int iters, float *sink) {
    __shared__ __attribute__((aligned(16))) char buf[16384];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int row = 0, slot = 0;
    switch (pattern) {
        case 0: row = lane >> 3; slot = lane & 7; break;
        case 1: row = lane; slot = 0; break;
        case 2: row = lane >> 2; slot = ((lane & 3) * 2) ^ ((row >> 1) & 7); break;
        case 3: row = lane >> 2; slot = ((lane & 3) * 2) ^ (row & 7); break;
        case 4: row = lane & 31; slot = (lane >> 5) ^ ((row >> 1) & 7); break;
        case 5: row = lane & 31; slot = (lane >> 5) ^ (row & 7); break;
        default: row = lane & 15; slot = (lane >> 4) ^ (row & 7); break;
    }
    char *p = buf + ((wave * 16 + row) & 127) * 128 + slot * 16;
    f32x4 v = {(float)lane, 1.0f, 2.0f, 3.0f}, acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; i++) {
        if (do_read) {
            f32x4 t0, t1, t2, t3;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %4\n\tds_read_b128 %3, %4\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"((unsigned)(size_t)p) : "memory");
            acc += t0 + t1 + t2 + t3;
        } else {
            asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %1\n\tds_write_b128 %0, %1\n\tds_write_b128 %0, %1" ::"v"((unsigned)(size_t)p), "v"(v) : "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (sink && acc[0] == 12345.678f) sink[threadIdx.x] = acc[0] + *(float *)p;
}
bool lds_pattern_bench(int pattern, int do_read, int iters, float *ms_out) {
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return false;
    hipLaunchKernelGGL(lds_pattern_kernel, dim3(2048), dim3(256), 0, nullptr, pattern, do_read, 16, (float *)nullptr);
    (void)hipEventRecord(e0, nullptr);
    hipLaunchKernelGGL(lds_pattern_kernel, dim3(2048), dim3(256), 0, nullptr, pattern, do_read, iters, (float *)nullptr);
    (void)hipEventRecord(e1, nullptr);
    const bool ok = hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(ms_out, e0, e1) == hipSuccess;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ok && hipGetLastError() == hipSuccess;
}

#endif  // WMI_HARNESS

#ifdef WMI_HARNESS
// which attention kernel enc_attention() launches in the harness library: 0 the product's (enc_attention_wide_kernel),
// 3 enc_attention_kernel, 1 enc_attention_pipe_kernel, 11..17 its ablations (WMI_ENC_ATTN at load, wmi_debug_set_enc_attention afterwards)
std::atomic<int> g_enc_attention_variant{[] { const char *e = getenv("WMI_ENC_ATTN"); return e ? atoi(e) : 0; }()};
void enc_attention_set_variant(int v) { g_enc_attention_variant.store(v, std::memory_order_relaxed); }
#endif

bool enc_attention(const f16 *Q, const f16 *K, const f16 *Vt, f16 *att, int B, int H, int T, int Tp, hipStream_t st) {
    if (B <= 0) return true;
    if (Tp % 128 != 0 || T > Tp) { log_msg(2, "enc_attention: bad padding T=%d Tp=%d\n", T, Tp); return false; }
#ifdef WMI_HARNESS
    const dim3 grid((Tp / 128) * H * B);
    switch (g_enc_attention_variant.load(std::memory_order_relaxed)) {
        case 3: hipLaunchKernelGGL(enc_attention_kernel, grid, dim3(256), 0, st, Q, K, Vt, att, H, T, Tp); return hipGetLastError() == hipSuccess;
#define WMI_PIPE(V_, A_) case V_: hipLaunchKernelGGL(enc_attention_pipe_kernel<A_>, grid, dim3(256), 0, st, Q, K, Vt, att, H, T, Tp); return hipGetLastError() == hipSuccess;
        WMI_PIPE(1, 0) WMI_PIPE(11, 1) WMI_PIPE(12, 2) WMI_PIPE(13, 3) WMI_PIPE(14, 4) WMI_PIPE(15, 5) WMI_PIPE(16, 6) WMI_PIPE(17, 7)
#undef WMI_PIPE
        default: break;
    }
#endif
    hipLaunchKernelGGL(enc_attention_wide_kernel, dim3(((Tp + 255) / 256) * H * B), dim3(256), 0, st, Q, K, Vt, att, H, T, Tp);
    return hipGetLastError() == hipSuccess;
}

}  // namespace wmi
