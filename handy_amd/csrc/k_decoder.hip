// Decoder-side kernels for gfx950: token embedding, single-query attention over the paged
// self-attention cache and over the cached encoder K/V, and the logits rules + selection.
// All of them are HBM/L2-bound byte movers: 16-byte loads, many independent loads in flight.
#include "kernels.h"

namespace wmi {

// ------------------------------------------------------------------------------------------ embedding
__global__ void dec_embed_kernel(const f16 *__restrict__ te, const float *__restrict__ pe, const DecRow *__restrict__ rows, int M, int d, float *__restrict__ x) {
    const int nch = d >> 2;
    const long total = (long)M * nch;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / nch), c = (int)(i - (long)r * nch) * 4;
        const DecRow row = rows[r];
        const f16x4 t = *(const f16x4 *)(te + (long)row.token * d + c);
        const f32x4 p = *(const f32x4 *)(pe + (long)row.pos * d + c);
        *(f32x4 *)(x + (long)r * d + c) = f32x4{(float)t[0] + p[0], (float)t[1] + p[1], (float)t[2] + p[2], (float)t[3] + p[3]};
    }
}
bool dec_embed(const f16 *te, const float *pe, const DecRow *rows, int M, int d, float *x, hipStream_t st) {
    if (M <= 0) return true;
    const long total = (long)M * (d / 4);
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(dec_embed_kernel, dim3(blocks), dim3(256), 0, st, te, pe, rows, M, d, x);
    return hipGetLastError() == hipSuccess;
}

// ------------------------------------------------------------------------------------------ single-query attention
// Lane layout: 8 lanes share one key (each owns 8 of the 64 head dims, one 16-byte load of K and
// of V per key); a wave covers 8 keys per pass. Every lane group keeps its own running
// (max, sum, acc[8]) over the keys it visits; the groups are merged once at the end.
struct AttnAcc { float m, l, a[8]; };

__device__ __forceinline__ void attn_init(AttnAcc &s) { s.m = -INFINITY; s.l = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) s.a[i] = 0.0f; }

__device__ __forceinline__ void attn_step(AttnAcc &s, const float *q, f16x8 k, f16x8 v) {
    float d0 = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) d0 = __builtin_fmaf(q[i], (float)k[i], d0);
    d0 += __shfl_xor(d0, 1); d0 += __shfl_xor(d0, 2); d0 += __shfl_xor(d0, 4);
    const float mn = fmaxf(s.m, d0);
    const float alpha = __expf(s.m - mn), p = __expf(d0 - mn);
    // the fused multiply-adds are written out: left to the compiler, "a * alpha + p * v" is fused around either product, and
    // the choice differed between instantiations of the kernels below (a row's result then depended, in its last bit and
    // once in a few thousand outputs, on how many rows shared its pass)
    s.l = __builtin_fmaf(s.l, alpha, p);
#pragma unroll
    for (int i = 0; i < 8; i++) s.a[i] = __builtin_fmaf(p, (float)v[i], s.a[i] * alpha);
    s.m = mn;
}
// merge with the lane `mask` away (same head-dim slice, other keys); a group that saw no key has m = -inf, l = 0
__device__ __forceinline__ void attn_merge_xor(AttnAcc &s, int mask) {
    const float om = __shfl_xor(s.m, mask), ol = __shfl_xor(s.l, mask);
    const float mn = fmaxf(s.m, om);
    const float fa = s.m == -INFINITY ? 0.0f : __expf(s.m - mn), fb = om == -INFINITY ? 0.0f : __expf(om - mn);
    s.l = __builtin_fmaf(ol, fb, s.l * fa);
#pragma unroll
    for (int i = 0; i < 8; i++) { const float oa = __shfl_xor(s.a[i], mask); s.a[i] = __builtin_fmaf(oa, fb, s.a[i] * fa); }
    s.m = mn;
}

// Merge of the four waves' partial states through LDS and the final store of one (row, head) output: 64 head dims by 8 threads.
__device__ __forceinline__ void attn_block_finish(AttnAcc &s, float (*red)[8][10], int tid, f16 *dst) {
    const int wave = tid >> 6, lane = tid & 63, g = lane >> 3, e8 = lane & 7;
    attn_merge_xor(s, 8); attn_merge_xor(s, 16); attn_merge_xor(s, 32);
    if (g == 0) { red[wave][e8][0] = s.m; red[wave][e8][1] = s.l;
#pragma unroll
        for (int i = 0; i < 8; i++) red[wave][e8][2 + i] = s.a[i]; }
    __syncthreads();
    if (tid < 8) {
        float mn = -INFINITY;
        for (int w = 0; w < 4; w++) mn = fmaxf(mn, red[w][tid][0]);
        float l = 0.0f, a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int w = 0; w < 4; w++) {
            const float mw = red[w][tid][0];
            const float f = mw == -INFINITY ? 0.0f : __expf(mw - mn);
            l = __builtin_fmaf(red[w][tid][1], f, l);
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = __builtin_fmaf(red[w][tid][2 + i], f, a[i]);
        }
        const float inv = 1.0f / l;
        f16x8 o;
#pragma unroll
        for (int i = 0; i < 8; i++) o[i] = (f16)(a[i] * inv);
        *(f16x8 *)(dst + tid * 8) = o;
    }
}

// Causal self-attention over the paged cache: 4 waves = 32 keys per pass, two passes in flight, so a sequence of a
// few hundred cached tokens costs a handful of dependent global-load round trips instead of one per 8 keys.
__global__ __launch_bounds__(256) void dec_self_attention_kernel(const f16 *__restrict__ qb, const f16 *__restrict__ kc, const f16 *__restrict__ vc,
                                                                 const int *__restrict__ block_table, const DecRow *__restrict__ rows, int d, int H, f16 *__restrict__ att) {
    __shared__ float red[4][8][10];
    const int head = blockIdx.x, m = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 3, e8 = lane & 7;
    const DecRow row = rows[m];
    const int n_kv = row.pos + 1;
    const int *bt = block_table + (long)row.seq * kMaxPagesPerSeq;
    float q[8];
    { const f16x8 qv = *(const f16x8 *)(qb + (long)m * d + head * 64 + e8 * 8);
#pragma unroll
      for (int i = 0; i < 8; i++) q[i] = (float)qv[i]; }
    auto kv_off = [&](int j) { return (((long)bt[j >> 4] * H + head) * kPageTokens + (j & 15)) * 64 + e8 * 8; };
    AttnAcc s; attn_init(s);
    int j = wave * 8 + g;
    for (; j + 32 < n_kv; j += 64) {
        const long o0 = kv_off(j), o1 = kv_off(j + 32);
        const f16x8 k0 = __builtin_nontemporal_load((const f16x8 *)(kc + o0)), v0 = __builtin_nontemporal_load((const f16x8 *)(vc + o0));
        const f16x8 k1 = __builtin_nontemporal_load((const f16x8 *)(kc + o1)), v1 = __builtin_nontemporal_load((const f16x8 *)(vc + o1));
        attn_step(s, q, k0, v0);
        attn_step(s, q, k1, v1);
    }
    for (; j < n_kv; j += 32) { const long o = kv_off(j); attn_step(s, q, __builtin_nontemporal_load((const f16x8 *)(kc + o)), __builtin_nontemporal_load((const f16x8 *)(vc + o))); }
    attn_block_finish(s, red, tid, att + (long)m * d + head * 64);
}
bool dec_self_attention(const f16 *q, const f16 *kc, const f16 *vc, const int *block_table, const DecRow *rows, int M, int d, int H, f16 *att, hipStream_t st) {
    if (M <= 0) return true;
    hipLaunchKernelGGL(dec_self_attention_kernel, dim3(H, M), dim3(256), 0, st, q, kc, vc, block_table, rows, d, H, att);
    return hipGetLastError() == hipSuccess;
}

// Cross-attention: T (1500) keys per (row, head), 384 KB of K+V streamed per pair; 4 waves = 32 keys per pass.
// The K and V loads carry the non-temporal hint: a decode step streams every utterance's cache once (31 GB at 1024
// utterances) and nothing of it is read again before the next step, so keeping it out of the caches' retention is
// worth 9 % of the kernel (6.1 -> 6.7 TB/s).
// Rows that belong to one utterance (the beams of a beam search, the tokens of a prompt) read the same K and V:
// a workgroup takes a group of up to NQ such rows (consecutive in the pass; the host passes {first row, count})
// and applies every key it loads to all of them, so the cache is streamed once per group, not once per row.
template <int NQ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 4))) void dec_cross_attention_kernel(const f16 *__restrict__ qb, const f16 *__restrict__ ckv, const DecRow *__restrict__ rows,
                                                                  const int2 *__restrict__ groups, int d, int H, int T, int Tp, int n_layer, int layer, f16 *__restrict__ att) {
    __shared__ float red[4][8][10];
    const int head = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 3, e8 = lane & 7;
    const int m0 = NQ == 1 ? (int)blockIdx.y : groups[blockIdx.y].x, cnt = NQ == 1 ? 1 : groups[blockIdx.y].y;
    const long slot = rows[m0].utt;
    const f16 *kp = ckv + ((((slot * n_layer + layer) * 2 + 0) * H + head) * (long)Tp) * 64 + e8 * 8;
    const f16 *vp = kp + (long)H * Tp * 64;
    float q[NQ][8];
    AttnAcc s[NQ];
#pragma unroll
    for (int c = 0; c < NQ; c++) {
        const f16x8 qv = *(const f16x8 *)(qb + (long)(m0 + (c < cnt ? c : 0)) * d + head * 64 + e8 * 8);
#pragma unroll
        for (int i = 0; i < 8; i++) q[c][i] = (float)qv[i];
        attn_init(s[c]);
    }
    int j = wave * 8 + g;
    if (NQ == 1) {
        // eight independent key rows (16 x 16-byte loads per lane) in flight per iteration: with one clip the launch is 20
        // workgroups of 384 KB each, and what a wave has in flight bounds its rate; the keys are still applied in order
        for (; j + 224 < T; j += 256) {
            f16x8 k[8], v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { k[u] = __builtin_nontemporal_load((const f16x8 *)(kp + (long)(j + 32 * u) * 64)); v[u] = __builtin_nontemporal_load((const f16x8 *)(vp + (long)(j + 32 * u) * 64)); }
#pragma unroll
            for (int u = 0; u < 8; u++) attn_step(s[0], q[0], k[u], v[u]);
        }
        for (; j + 96 < T; j += 128) {     // four independent key rows (8 x 16-byte loads per lane) in flight per iteration
            f16x8 k[4], v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { k[u] = __builtin_nontemporal_load((const f16x8 *)(kp + (long)(j + 32 * u) * 64)); v[u] = __builtin_nontemporal_load((const f16x8 *)(vp + (long)(j + 32 * u) * 64)); }
#pragma unroll
            for (int u = 0; u < 4; u++) attn_step(s[0], q[0], k[u], v[u]);
        }
    }
    for (; j + 32 < T; j += 64) {          // two independent key rows in flight per iteration
        const f16x8 k0 = __builtin_nontemporal_load((const f16x8 *)(kp + (long)j * 64)), v0 = __builtin_nontemporal_load((const f16x8 *)(vp + (long)j * 64));
        const f16x8 k1 = __builtin_nontemporal_load((const f16x8 *)(kp + (long)(j + 32) * 64)), v1 = __builtin_nontemporal_load((const f16x8 *)(vp + (long)(j + 32) * 64));
#pragma unroll
        for (int c = 0; c < NQ; c++) { attn_step(s[c], q[c], k0, v0); attn_step(s[c], q[c], k1, v1); }
    }
    for (; j < T; j += 32) {
        const f16x8 k0 = __builtin_nontemporal_load((const f16x8 *)(kp + (long)j * 64)), v0 = __builtin_nontemporal_load((const f16x8 *)(vp + (long)j * 64));
#pragma unroll
        for (int c = 0; c < NQ; c++) attn_step(s[c], q[c], k0, v0);
    }
#pragma unroll
    for (int c = 0; c < NQ; c++) {
        if (c < cnt) attn_block_finish(s[c], red, tid, att + (long)(m0 + c) * d + head * 64);      // cnt is uniform over the workgroup
        if (NQ > 1) __syncthreads();                                                                // red is reused by the next row
    }
}
// ---- cross-attention for groups of rows on MFMA (beam search: the beams of an utterance share its K and V)
// V of the cross cache, transposed per (slot, layer, head): vt[dim][key]. Made once per encoded window, only for
// runs that decode several sequences per utterance.
__global__ __launch_bounds__(256) void cross_v_transpose_kernel(const f16 *__restrict__ ckv, const int *__restrict__ slots, f16 *__restrict__ vt,
                                                                int H, int T, int Tp, int n_layer) {
    __shared__ f16 tile[64][66];
    const int t0 = blockIdx.x * 64, head = blockIdx.y % H, layer = blockIdx.y / H;
    const long slot = slots[blockIdx.z];
    const f16 *v = ckv + ((((slot * n_layer + layer) * 2 + 1) * H + head) * (long)Tp) * 64;
    f16 *o = vt + (((slot * n_layer + layer) * H + head) * 64L) * Tp;
    for (int i = threadIdx.x; i < 64 * 8; i += 256) {
        const int t = i >> 3, c = i & 7;
        f16x8 x = {0, 0, 0, 0, 0, 0, 0, 0};
        if (t0 + t < T) x = *(const f16x8 *)(v + (long)(t0 + t) * 64 + c * 8);
#pragma unroll
        for (int e = 0; e < 8; e++) tile[t][c * 8 + e] = x[e];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 8; i += 256) {
        const int e = i >> 3, c = i & 7;
        f16x8 x;
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = tile[c * 8 + k][e];
        *(f16x8 *)(o + (long)e * Tp + t0 + c * 8) = x;         // t0 + 64 <= Tp: Tp is a multiple of 128
    }
}
bool cross_v_transpose(const f16 *ckv, const int *slots, int n_slots, f16 *vt, int H, int T, int Tp, int n_layer, hipStream_t st) {
    if (n_slots <= 0) return true;
    hipLaunchKernelGGL(cross_v_transpose_kernel, dim3((T + 63) / 64, H * n_layer, n_slots), dim3(256), 0, st, ckv, slots, vt, H, T, Tp, n_layer);
    return hipGetLastError() == hipSuccess;
}

// One workgroup = one (group of <= 16 rows of one utterance, head); its 4 waves take the 32-key tiles round-robin.
//   S^T[key][row] = K Q^T    v_mfma_f32_16x16x32_f16, K rows straight from global memory as the A operand (16 B per lane),
//                            Q as the B operand: a lane ends up with one row's scores for 4 keys of each 16-key half
//   softmax                  running maximum per row kept equal in the 4 lanes that share the row (two shuffles per
//                            tile), because all of them feed one accumulator
//   O^T[dim][row] += V^T P^T P^T goes from the score registers straight into the B operand; V^T comes from the transposed
//                            copy; the K rows of a tile are taken in the order that makes a lane's 8 scores 8 consecutive
//                            keys, so its V^T operand is one 16-byte load per 16 dims
// Every K and V byte of the head is read once per group; the four waves' partial results meet in LDS. (No non-temporal
// hint on these loads, unlike the row-per-workgroup kernel: a 128-byte line here is consumed by several load
// instructions and has to stay cached between them -- with the hint the kernel drops from 4.7 to 3.6 TB/s.)
__global__ __launch_bounds__(256) void dec_cross_attention_mq_kernel(const f16 *__restrict__ qb, const f16 *__restrict__ ckv, const f16 *__restrict__ vt,
                                                                     const DecRow *__restrict__ rows, const int2 *__restrict__ groups, int d, int H, int T, int Tp,
                                                                     int n_layer, int layer, f16 *__restrict__ att) {
    __shared__ float red[4][16][66];       // per wave and row: 64 dims, running maximum, sum
    const int head = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int m0 = groups[blockIdx.y].x, cnt = groups[blockIdx.y].y;
    const long slot = rows[m0].utt;
    const f16 *kp = ckv + ((((slot * n_layer + layer) * 2 + 0) * H + head) * (long)Tp) * 64;
    const f16 *vp = vt + (((slot * n_layer + layer) * H + head) * 64L) * Tp;
    f16x8 qf[2];
    {
        const f16 *q = qb + (long)(m0 + (n < cnt ? n : 0)) * d + head * 64 + g * 8;
        qf[0] = *(const f16x8 *)q; qf[1] = *(const f16x8 *)(q + 32);
    }
    f32x4 o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) o[i] = f32x4{0, 0, 0, 0};
    float m_run = -INFINITY, l_run = 0.0f;
    const int n_tiles = (T + 31) / 32;
    // Key order inside a 32-key tile: row i of score tile u is key (i >> 2) * 8 + u * 4 + (i & 3), so that the 8 scores a lane
    // ends up with (rows 4g .. 4g+3 of both tiles) are the 8 consecutive keys 8g .. 8g+7 -- its V^T operand is then one
    // 16-byte load per 16 dims instead of two 8-byte ones.
    struct Frags { f16x8 k[2][2]; f16x8 v[4]; };
    auto load_tile = [&](Frags &f, int t) {
        const int key0 = t * 32;
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int ks = 0; ks < 2; ks++) f.k[u][ks] = *(const f16x8 *)(kp + (long)(key0 + (n >> 2) * 8 + u * 4 + (n & 3)) * 64 + ks * 32 + g * 8);
#pragma unroll
        for (int db = 0; db < 4; db++) f.v[db] = *(const f16x8 *)(vp + (long)(db * 16 + n) * Tp + key0 + g * 8);
    };
    auto compute_tile = [&](const Frags &f, int t) {
        f32x4 s[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            s[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.k[u][0], qf[0], f32x4{0, 0, 0, 0}, 0, 0, 0);
            s[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.k[u][1], qf[1], s[u], 0, 0, 0);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (t * 32 + g * 8 + u * 4 + r >= T) s[u][r] = -INFINITY;
                mx = fmaxf(mx, s[u][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);           // exp(-inf) = 0 on the first tile
        f16x8 pf;
        float ls = 0.0f;
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int r = 0; r < 4; r++) { const float p = __expf(s[u][r] - m_new); ls += p; pf[u * 4 + r] = (f16)p; }
        l_run = l_run * alpha + ls;
        m_run = m_new;
#pragma unroll
        for (int db = 0; db < 4; db++) {
            o[db] *= alpha;
            o[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.v[db], pf, o[db], 0, 0, 0);
        }
    };
    // two register sets: the next tile of this wave is in flight while the current one is multiplied
    Frags fa, fb;
    int t = wave;
    if (t < n_tiles) load_tile(fa, t);
    for (; t < n_tiles; t += 8) {
        if (t + 4 < n_tiles) load_tile(fb, t + 4);
        compute_tile(fa, t);
        if (t + 4 < n_tiles) {
            if (t + 8 < n_tiles) load_tile(fa, t + 8);
            compute_tile(fb, t + 4);
        }
    }
    l_run += __shfl_xor(l_run, 16); l_run += __shfl_xor(l_run, 32);
#pragma unroll
    for (int db = 0; db < 4; db++)
#pragma unroll
        for (int r = 0; r < 4; r++) red[wave][n][db * 16 + g * 4 + r] = o[db][r];
    if (g == 0) { red[wave][n][64] = m_run; red[wave][n][65] = l_run; }
    __syncthreads();
    for (int i = tid; i < cnt * 64; i += 256) {
        const int row = i >> 6, e = i & 63;
        float mn = -INFINITY;
        for (int w = 0; w < 4; w++) mn = fmaxf(mn, red[w][row][64]);
        float l = 0.0f, a = 0.0f;
        for (int w = 0; w < 4; w++) {
            const float mw = red[w][row][64];
            const float f = mw == -INFINITY ? 0.0f : __expf(mw - mn);
            l += red[w][row][65] * f; a += red[w][row][e] * f;
        }
        att[(long)(m0 + row) * d + head * 64 + e] = (f16)(a / l);
    }
}
bool dec_cross_attention_mq(const f16 *q, const f16 *ckv, const f16 *vt, const DecRow *rows, const int2 *groups, int n_groups, int d, int H, int T, int Tp,
                            int n_layer, int layer, f16 *att, hipStream_t st) {
    if (n_groups <= 0) return true;
    hipLaunchKernelGGL(dec_cross_attention_mq_kernel, dim3(H, n_groups), dim3(256), 0, st, q, ckv, vt, rows, groups, d, H, T, Tp, n_layer, layer, att);
    return hipGetLastError() == hipSuccess;
}

bool dec_cross_attention(const f16 *q, const f16 *ckv, const DecRow *rows, const int2 *groups, int n_groups, int max_group, int M, int d, int H, int T, int Tp,
                         int n_layer, int layer, f16 *att, hipStream_t st) {
    if (M <= 0) return true;
#define WMI_XATT(NQ_) hipLaunchKernelGGL((dec_cross_attention_kernel<NQ_>), dim3(H, NQ_ == 1 ? M : n_groups), dim3(256), 0, st, q, ckv, rows, groups, d, H, T, Tp, n_layer, layer, att)
    if (max_group <= 1 || !groups) WMI_XATT(1);
    else if (max_group <= 2) WMI_XATT(2);
    else if (max_group <= 4) WMI_XATT(4);
    else if (max_group <= 8) WMI_XATT(8);
    else return false;
#undef WMI_XATT
    return hipGetLastError() == hipSuccess;
}

// ------------------------------------------------------------------------------------------ logits rules + selection
// Restates whisper_process_logits + whisper_sample_token(best): suppression rules, timestamp pairing /
// monotonicity / initial bound, log-softmax, the "timestamp mass beats every text token" rule, then
// argmax (lowest id on ties) and timestamp statistics, with wavefront reductions.
// A row of ~52k logits is split over kSampleSplits workgroups so that a small batch still fills the chip:
//   1. sample_partial_kernel  masks its slice in place (later passes are plain reads that hit in L2) and
//      emits per-slice maxima (text / timestamp, with index) and exp-sums relative to the slice maximum;
//   2. sample_final_kernel    one wave per row folds the slices, applies the timestamp-mass rule, selects;
//   3. sample_extra_kernel    only for rows that asked for it: probabilities (host-side draws at T > 0) and the
//      per-slice top-k candidates;  4. topk_merge_kernel folds those into the row's k most probable tokens.
constexpr int kSampleSplits = 16;
constexpr int kSampleChunk = 3328;              // 13 x 256; 16 x 3328 >= 52000
constexpr int kSamplePer = kSampleChunk / 256;

struct ArgMax { float v; int i; };
__device__ __forceinline__ ArgMax amax2(ArgMax a, ArgMax b) { return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a; }
__device__ __forceinline__ ArgMax wave_amax(ArgMax a) {
    for (int o = 32; o > 0; o >>= 1) { ArgMax b; b.v = __shfl_xor(a.v, o); b.i = __shfl_xor(a.i, o); a = amax2(a, b); }
    return a;
}
__device__ __forceinline__ float wave_sum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ ArgMax block_amax(ArgMax a, ArgMax *sh) {
    a = wave_amax(a);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
    __syncthreads();
    ArgMax r = sh[0];
    for (int w = 1; w < 4; w++) r = amax2(r, sh[w]);
    return r;
}
__device__ float block_sum(float v, float *sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

struct SamplePartial { float mt_v; int mt_i; float ms_v; int ms_i; float m, st, ss; int pad; };

__global__ __launch_bounds__(256) void sample_partial_kernel(float *__restrict__ logits, long ld, const SampleRow *__restrict__ srow, VocabIds vi,
                                                             const unsigned *__restrict__ nst_bitmap, SamplePartial *__restrict__ part) {
    __shared__ ArgMax sh_a[4];
    __shared__ float sh_f[4];
    const int s = blockIdx.x, r = blockIdx.y, tid = threadIdx.x, V = vi.n_vocab;
    const SampleRow sr = srow[r];
    float *lg = logits + (long)r * ld;
    const int fl = sr.flags;
    const bool is_initial = fl & 1, last_ts = fl & 2, pen_ts = fl & 4, has_ts = fl & 8, sup_blank = fl & 16, no_ts = fl & 32,
               sup_nst = fl & 64, tdrz = fl & 128, raw = fl & 256, stage_a = fl & 512, stage_b = fl & 1024;
    // stage_a / stage_b: the rules in two launches with a host callback on the logits in between (logits_filter_callback):
    // the first applies the temperature and the special-token suppression, the second everything after them
    ArgMax mt = {-INFINITY, 0x7fffffff}, ms = {-INFINITY, 0x7fffffff};
    float val[kSamplePer];
#pragma unroll
    for (int k = 0; k < kSamplePer; k++) {
        const int i = s * kSampleChunk + k * 256 + tid;
        float v = -INFINITY;
        if (i < V) {
            v = lg[i];
            const float v_in = v;
            if (sr.temperature > 0.0f && !stage_b) v = v / sr.temperature;
            if (!raw) {
                bool kill = false;
                if (!stage_b) {
                    if (sup_blank && is_initial && (i == vi.eot || i == vi.space)) kill = true;
                    if (i == vi.tok_not || i == vi.sot || i == vi.nosp || i == vi.translate || i == vi.transcribe || i == vi.prev) kill = true;
                    if (i == vi.solm && !tdrz) kill = true;
                    if (no_ts && i >= vi.beg) kill = true;
                    if (i > vi.sot && i <= vi.sot + 100) kill = true;                       // language tokens
                }
                if (!stage_a) {
                    if (sup_nst && ((nst_bitmap[i >> 5] >> (i & 31)) & 1)) kill = true;
                    if (sr.force_n > 0) {
                        if (sr.n_sampled < sr.force_n) { if (i == vi.eot) kill = true; }
                        else { if (i != vi.eot) kill = true; else { kill = false; v = 0.0f; } }
                    }
                    if (last_ts) { if (pen_ts) { if (i >= vi.beg) kill = true; } else if (i < vi.eot) kill = true; }
                    if (is_initial && sr.max_initial >= 0 && i > vi.beg + sr.max_initial) kill = true;
                    if (has_ts && i >= vi.beg && i < vi.beg + sr.ts_min) kill = true;
                }
                if (kill) v = -INFINITY;
                // only what a rule changed goes back (a greedy step changes a few hundred of a row's 52 k values: the row is
                // then read once instead of read and rewritten)
                if (__float_as_uint(v) != __float_as_uint(v_in)) lg[i] = v;
            }
            if (i < vi.beg) { if (v > mt.v) { mt.v = v; mt.i = i; } }
            else if (v > ms.v) { ms.v = v; ms.i = i; }
        }
        val[k] = v;
    }
    mt = block_amax(mt, sh_a);
    ms = block_amax(ms, sh_a);
    const float m = fmaxf(mt.v, ms.v);
    float st = 0.0f, ss = 0.0f;
    if (m > -INFINITY) {
#pragma unroll
        for (int k = 0; k < kSamplePer; k++) {
            const int i = s * kSampleChunk + k * 256 + tid;
            if (val[k] > -INFINITY) { const float e = __expf(val[k] - m); if (i < vi.beg) st += e; else ss += e; }
        }
    }
    st = block_sum(st, sh_f);
    ss = block_sum(ss, sh_f);
    if (tid == 0) part[(long)r * kSampleSplits + s] = SamplePartial{mt.v, mt.i, ms.v, ms.i, m, st, ss, 0};
}

__global__ __launch_bounds__(64) void sample_final_kernel(const SamplePartial *__restrict__ part, const SampleRow *__restrict__ srow, VocabIds vi, SampleOut *__restrict__ out) {
    const int r = blockIdx.x, lane = threadIdx.x;
    SamplePartial p{-INFINITY, 0x7fffffff, -INFINITY, 0x7fffffff, -INFINITY, 0.0f, 0.0f, 0};
    if (lane < kSampleSplits) p = part[(long)r * kSampleSplits + lane];
    const ArgMax mt = wave_amax(ArgMax{p.mt_v, p.mt_i}), ms = wave_amax(ArgMax{p.ms_v, p.ms_i});
    const float mx = fmaxf(mt.v, ms.v);
    const float f = p.m > -INFINITY ? __expf(p.m - mx) : 0.0f;
    const float st = wave_sum(p.st * f), ss = wave_sum(p.ss * f);
    if (lane != 0) return;
    const bool raw = srow[r].flags & 256;
    const float lse = logf(st + ss) + mx;
    const float ts_lp = ss > 0.0f ? logf(ss) + mx - lse : -INFINITY;      // log of the total timestamp probability
    const bool ts_wins = !raw && ts_lp > mt.v - lse;
    ArgMax best = ts_wins ? ms : amax2(mt, ms);
    if (best.i == 0x7fffffff) best.i = 0;
    SampleOut o;
    o.id = best.i; o.plog = best.v - lse; o.p = __expf(o.plog); o.lse = lse; o.pad = ts_wins ? 1 : 0;
    const float sum_ts = ss > 0.0f ? __expf(ts_lp) : 0.0f;
    const float max_ts = ms.v > -INFINITY ? __expf(ms.v - lse) : 0.0f;
    o.tid = max_ts > 0.0f ? ms.i : vi.beg;
    o.pt = max_ts / (sum_ts + 1e-10f); o.ptsum = sum_ts;
    if (o.id >= vi.beg) { o.tid = o.id; o.pt = o.p; }
    out[r] = o;
}

struct TopkCand { float v; int i; };

__global__ __launch_bounds__(256) void sample_extra_kernel(const float *__restrict__ logits, long ld, const SampleRow *__restrict__ srow, VocabIds vi,
                                                           const SampleOut *__restrict__ out, float *__restrict__ probs, TopkCand *__restrict__ cand) {
    __shared__ ArgMax sh_a[4];
    const int s = blockIdx.x, r = blockIdx.y, tid = threadIdx.x, V = vi.n_vocab;
    const SampleRow sr = srow[r];
    if (!sr.want_probs && sr.topk <= 0) return;
    const SampleOut o = out[r];
    const bool ts_wins = o.pad != 0;
    const float *lg = logits + (long)r * ld;
    float val[kSamplePer];
#pragma unroll
    for (int k = 0; k < kSamplePer; k++) {
        const int i = s * kSampleChunk + k * 256 + tid;
        float v = -INFINITY;
        if (i < V) { v = lg[i]; if (ts_wins && i < vi.beg) v = -INFINITY; }
        val[k] = v;
        if (probs && sr.want_probs && i < V) probs[(long)r * V + i] = v == -INFINITY ? 0.0f : __expf(v - o.lse);
    }
    if (cand && sr.topk > 0) {
        for (int t = 0; t < sr.topk && t < 8; t++) {
            ArgMax a = {-INFINITY, 0x7fffffff};
#pragma unroll
            for (int k = 0; k < kSamplePer; k++) { const int i = s * kSampleChunk + k * 256 + tid; if (val[k] > a.v) { a.v = val[k]; a.i = i; } }
            a = block_amax(a, sh_a);
#pragma unroll
            for (int k = 0; k < kSamplePer; k++) if (s * kSampleChunk + k * 256 + tid == a.i) val[k] = -INFINITY;
            if (tid == 0) cand[((long)r * kSampleSplits + s) * 8 + t] = TopkCand{a.v, a.i};
        }
    }
}

__global__ __launch_bounds__(128) void topk_merge_kernel(const TopkCand *__restrict__ cand, const SampleRow *__restrict__ srow, const SampleOut *__restrict__ out,
                                                         TopkOut *__restrict__ topk, int topk_stride) {
    __shared__ ArgMax sh[2];
    const int r = blockIdx.x, tid = threadIdx.x;
    const int k = min(srow[r].topk, 8);
    if (k <= 0) return;
    const float lse = out[r].lse;
    TopkCand c = cand[(long)r * kSampleSplits * 8 + tid];           // 16 slices x 8 candidates
    if ((tid & 7) >= k) c = TopkCand{-INFINITY, 0x7fffffff};
    for (int t = 0; t < k; t++) {
        ArgMax a = wave_amax(ArgMax{c.v, c.i});
        __syncthreads();
        if ((tid & 63) == 0) sh[tid >> 6] = a;
        __syncthreads();
        a = amax2(sh[0], sh[1]);
        if (c.i == a.i) c.v = -INFINITY;
        if (tid == 0) {
            TopkOut o; o.id = (a.v == -INFINITY || a.i == 0x7fffffff) ? -1 : a.i; o.plog = a.v - lse; o.p = a.v > -INFINITY ? __expf(o.plog) : 0.0f; o.pad = 0;
            topk[(long)r * topk_stride + t] = o;
        }
    }
}

size_t sample_workspace_bytes(int max_rows) { return (size_t)max_rows * kSampleSplits * (sizeof(SamplePartial) + 8 * sizeof(TopkCand)); }

bool sample_rows(float *logits, long ld, int R, const SampleRow *srow, const VocabIds &vi, const unsigned *nst_bitmap,
                 SampleOut *out, TopkOut *topk, int topk_stride, float *probs, bool any_extra, void *workspace, hipStream_t st) {
    if (R <= 0) return true;
    if (vi.n_vocab > kSampleSplits * kSampleChunk) { log_msg(2, "sample_rows: vocabulary of %d exceeds the kernel's slicing\n", vi.n_vocab); return false; }
    SamplePartial *part = (SamplePartial *)workspace;
    TopkCand *cand = (TopkCand *)(part + (size_t)R * kSampleSplits);
    hipLaunchKernelGGL(sample_partial_kernel, dim3(kSampleSplits, R), dim3(256), 0, st, logits, ld, srow, vi, nst_bitmap, part);
    hipLaunchKernelGGL(sample_final_kernel, dim3(R), dim3(64), 0, st, part, srow, vi, out);
    if (any_extra) {
        hipLaunchKernelGGL(sample_extra_kernel, dim3(kSampleSplits, R), dim3(256), 0, st, logits, ld, srow, vi, out, probs, cand);
        hipLaunchKernelGGL(topk_merge_kernel, dim3(R), dim3(128), 0, st, cand, srow, out, topk, topk_stride);
    }
    return hipGetLastError() == hipSuccess;
}

// ------------------------------------------------------------------------------------------ KV page copy (beam fork)
__global__ void copy_pages_kernel(f16 *kpool, f16 *vpool, long layer_stride, const int2 *pairs, int page_elems) {
    const int2 pr = pairs[blockIdx.x];
    const long lo = (long)blockIdx.y * layer_stride;
    const f16x8 *ks = (const f16x8 *)(kpool + lo + (long)pr.x * page_elems), *vs = (const f16x8 *)(vpool + lo + (long)pr.x * page_elems);
    f16x8 *kd = (f16x8 *)(kpool + lo + (long)pr.y * page_elems), *vd = (f16x8 *)(vpool + lo + (long)pr.y * page_elems);
    for (int i = threadIdx.x; i < page_elems / 8; i += blockDim.x) { kd[i] = ks[i]; vd[i] = vs[i]; }
}
bool copy_pages(f16 *kpool, f16 *vpool, long layer_stride, int n_layer, const int2 *pairs, int n_pairs, int H, hipStream_t st) {
    if (n_pairs <= 0) return true;
    hipLaunchKernelGGL(copy_pages_kernel, dim3(n_pairs, n_layer), dim3(256), 0, st, kpool, vpool, layer_stride, pairs, H * kPageTokens * 64);
    return hipGetLastError() == hipSuccess;
}

// ------------------------------------------------------------------------------------------ split-K consumers
// The decoder's skinny GEMMs leave fp32 partial sums ws[split][M][N]; these kernels add the splits in index
// order and apply what the fused GEMM epilogues would have applied.

// One workgroup per output row i (source row m = gather ? gather[i] : i), one float4 column group per thread:
// x[m] += bias + sum_s ws[s][m]; out[i] = LayerNorm(x[m]). The split loads of a thread are independent, so all of them are in
// flight at once. The statistics are layer_norm_kernel's, operation for operation: the new row goes through LDS, every wave
// reads it in that kernel's lane-to-column map (column group lane + 64 i) and makes the same additions in the same order
// (all waves get the same bits; each then normalises its own columns). A residual GEMM with a fused epilogue followed by
// layer_norm() therefore gives the same bits as this kernel after a split GEMM.
__global__ __launch_bounds__(320) void splitk_residual_ln_kernel(float *__restrict__ x, const float *__restrict__ ws, int splits, long split_stride,
                                                                 const float *__restrict__ bias, const float *__restrict__ w, const float *__restrict__ b,
                                                                 f16 *__restrict__ out, int d, const int *__restrict__ gather) {
    __shared__ f32x4 row[320];
    const int i_out = blockIdx.x, c = threadIdx.x, lane = c & 63, nch = d >> 2;
    const long m = gather ? gather[i_out] : i_out;
    const bool on = c < nch;
    f32x4 v = {0, 0, 0, 0};
    if (on) {
        const float *p = ws + m * d + 4 * c;
        f32x4 a = *(const f32x4 *)p;
        int s = 1;
        for (; s + 3 < splits; s += 4) {
            const f32x4 a0 = *(const f32x4 *)(p + s * split_stride), a1 = *(const f32x4 *)(p + (s + 1) * split_stride),
                        a2 = *(const f32x4 *)(p + (s + 2) * split_stride), a3 = *(const f32x4 *)(p + (s + 3) * split_stride);
            a += a0; a += a1; a += a2; a += a3;
        }
        for (; s < splits; s++) a += *(const f32x4 *)(p + s * split_stride);
        f32x4 *xr = (f32x4 *)(x + m * d) + c;
        v = *xr + (a + *(const f32x4 *)(bias + 4 * c));
        *xr = v;
        row[c] = v;
    }
    __syncthreads();
    f32x4 u[5];
#pragma unroll
    for (int i = 0; i < 5; i++) { const int cc = lane + 64 * i; u[i] = cc < nch ? row[cc] : f32x4{0, 0, 0, 0}; }
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < 5; i++) sum += (u[i][0] + u[i][1]) + (u[i][2] + u[i][3]);
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)d;
    float sq = 0.0f;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const int cc = lane + 64 * i;
        if (cc < nch) { u[i] -= mean; sq += (u[i][0] * u[i][0] + u[i][1] * u[i][1]) + (u[i][2] * u[i][2] + u[i][3] * u[i][3]); }
    }
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float scale = 1.0f / sqrtf(sq / (float)d + 1e-5f);
    if (on) {
        v -= mean;
        const f32x4 ww = *(const f32x4 *)(w + 4 * c), bb = *(const f32x4 *)(b + 4 * c);
        f32x4 y;
#pragma unroll
        for (int k = 0; k < 4; k++) y[k] = v[k] * scale * ww[k] + bb[k];
        *(f16x4 *)(out + (long)i_out * d + 4 * c) = f16x4{(f16)y[0], (f16)y[1], (f16)y[2], (f16)y[3]};
    }
}
bool splitk_residual_ln(float *x, const float *ws, int splits, const float *bias, const float *ln_w, const float *ln_b, f16 *out, int M, int d,
                        const int *gather, int n_out, hipStream_t st) {
    if (n_out <= 0) return true;
    if (d % 4 != 0 || d > 1280 || splits < 1) { log_msg(2, "splitk_residual_ln: unsupported width %d\n", d); return false; }
    const int threads = ((d / 4 + 63) / 64) * 64;
    hipLaunchKernelGGL(splitk_residual_ln_kernel, dim3(n_out), dim3(threads), 0, st, x, ws, splits, (long)M * d, bias, ln_w, ln_b, out, d, gather);
    return hipGetLastError() == hipSuccess;
}

// out[m][n] = f16(act((sum_s ws[s][m][n] + bias[n]) * scale)), 4 columns per thread
__global__ __launch_bounds__(256) void splitk_f16_kernel(const float *__restrict__ ws, int splits, long split_stride, const float *__restrict__ bias, int gelu, float scale,
                                                         f16 *__restrict__ out, long n4, int N) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int n = (int)((i * 4) % N);
    f32x4 a = *(const f32x4 *)(ws + i * 4);
    for (int s = 1; s < splits; s++) a += *(const f32x4 *)(ws + s * split_stride + i * 4);
    if (bias) a += *(const f32x4 *)(bias + n);
    a *= scale;
    *(f16x4 *)(out + i * 4) = gelu ? gelu_f16_dev4(a) : f16x4{(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3]};       // the GEMM epilogue's GELU
}
bool splitk_f16(const float *ws, int splits, const float *bias, int gelu, float scale, f16 *out, int M, int N, hipStream_t st) {
    if (M <= 0) return true;
    if (N % 4 != 0 || splits < 1) return false;
    const long n4 = (long)M * N / 4;
    hipLaunchKernelGGL(splitk_f16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, ws, splits, (long)M * N, bias, gelu, scale, out, n4, N);
    return hipGetLastError() == hipSuccess;
}

// n in [0,d): q[m][n] = f16((sum + bias) / 8); [d,2d) / [2d,3d): K / V of row m into the paged cache at rows[m].kv_index
__global__ __launch_bounds__(256) void splitk_dec_qkv_kernel(const float *__restrict__ ws, int splits, long split_stride, const float *__restrict__ bias,
                                                             const DecRow *__restrict__ rows, int d, int H, f16 *__restrict__ q, f16 *__restrict__ kc, f16 *__restrict__ vc, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int N = 3 * d;
    const int m = (int)((i * 4) / N), n = (int)((i * 4) - (long)m * N);
    f32x4 a = *(const f32x4 *)(ws + i * 4);
    for (int s = 1; s < splits; s++) a += *(const f32x4 *)(ws + s * split_stride + i * 4);
    a += *(const f32x4 *)(bias + n);
    const int part = n / d, c = n - part * d;
    if (part == 0) {
        *(f16x4 *)(q + (long)m * d + c) = f16x4{(f16)(a[0] * 0.125f), (f16)(a[1] * 0.125f), (f16)(a[2] * 0.125f), (f16)(a[3] * 0.125f)};
    } else {
        const int kv = rows[m].kv_index, page = kv / kPageTokens, off = kv % kPageTokens, h = c >> 6, e = c & 63;
        f16 *base = part == 1 ? kc : vc;
        *(f16x4 *)(base + (((long)page * H + h) * kPageTokens + off) * 64 + e) = f16x4{(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3]};
    }
}
bool splitk_dec_qkv(const float *ws, int splits, const float *bias, const DecRow *rows, int M, int d, int H, f16 *q, f16 *kc, f16 *vc, hipStream_t st) {
    if (M <= 0) return true;
    if (d % 64 != 0 || splits < 1) return false;
    const long n4 = (long)M * 3 * d / 4;
    hipLaunchKernelGGL(splitk_dec_qkv_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, ws, splits, (long)M * 3 * d, bias, rows, d, H, q, kc, vc, n4);
    return hipGetLastError() == hipSuccess;
}

}  // namespace wmi
