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
    for (int i = 0; i < 8; i++) d0 += q[i] * (float)k[i];
    d0 += __shfl_xor(d0, 1); d0 += __shfl_xor(d0, 2); d0 += __shfl_xor(d0, 4);
    const float mn = fmaxf(s.m, d0);
    const float alpha = __expf(s.m - mn), p = __expf(d0 - mn);
    s.l = s.l * alpha + p;
#pragma unroll
    for (int i = 0; i < 8; i++) s.a[i] = s.a[i] * alpha + p * (float)v[i];
    s.m = mn;
}
// merge with the lane `mask` away (same head-dim slice, other keys); a group that saw no key has m = -inf, l = 0
__device__ __forceinline__ void attn_merge_xor(AttnAcc &s, int mask) {
    const float om = __shfl_xor(s.m, mask), ol = __shfl_xor(s.l, mask);
    const float mn = fmaxf(s.m, om);
    const float fa = s.m == -INFINITY ? 0.0f : __expf(s.m - mn), fb = om == -INFINITY ? 0.0f : __expf(om - mn);
    s.l = s.l * fa + ol * fb;
#pragma unroll
    for (int i = 0; i < 8; i++) { const float oa = __shfl_xor(s.a[i], mask); s.a[i] = s.a[i] * fa + oa * fb; }
    s.m = mn;
}

__global__ __launch_bounds__(64) void dec_self_attention_kernel(const f16 *__restrict__ qb, const f16 *__restrict__ kc, const f16 *__restrict__ vc,
                                                                const int *__restrict__ block_table, const DecRow *__restrict__ rows, int d, int H, f16 *__restrict__ att) {
    const int head = blockIdx.x, m = blockIdx.y, lane = threadIdx.x, g = lane >> 3, e8 = lane & 7;
    const DecRow row = rows[m];
    const int n_kv = row.pos + 1;
    const int *bt = block_table + (long)row.seq * kMaxPagesPerSeq;
    float q[8];
    { const f16x8 qv = *(const f16x8 *)(qb + (long)m * d + head * 64 + e8 * 8);
#pragma unroll
      for (int i = 0; i < 8; i++) q[i] = (float)qv[i]; }
    AttnAcc s; attn_init(s);
    for (int j = g; j < n_kv; j += 8) {
        const int page = bt[j >> 4], off = j & 15;
        const long o = (((long)page * H + head) * kPageTokens + off) * 64 + e8 * 8;
        attn_step(s, q, *(const f16x8 *)(kc + o), *(const f16x8 *)(vc + o));
    }
    attn_merge_xor(s, 8); attn_merge_xor(s, 16); attn_merge_xor(s, 32);
    if (g == 0) {
        const float inv = 1.0f / s.l;
        f16x8 o;
#pragma unroll
        for (int i = 0; i < 8; i++) o[i] = (f16)(s.a[i] * inv);
        *(f16x8 *)(att + (long)m * d + head * 64 + e8 * 8) = o;
    }
}
bool dec_self_attention(const f16 *q, const f16 *kc, const f16 *vc, const int *block_table, const DecRow *rows, int M, int d, int H, f16 *att, hipStream_t st) {
    if (M <= 0) return true;
    hipLaunchKernelGGL(dec_self_attention_kernel, dim3(H, M), dim3(64), 0, st, q, kc, vc, block_table, rows, d, H, att);
    return hipGetLastError() == hipSuccess;
}

// Cross-attention: T (1500) keys per (row, head), 384 KB of K+V streamed per pair; 4 waves = 32 keys per pass.
__global__ __launch_bounds__(256) void dec_cross_attention_kernel(const f16 *__restrict__ qb, const f16 *__restrict__ ckv, const DecRow *__restrict__ rows,
                                                                  int d, int H, int T, int Tp, int n_layer, int layer, f16 *__restrict__ att) {
    __shared__ float red[4][8][10];
    const int head = blockIdx.x, m = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 3, e8 = lane & 7;
    const long slot = rows[m].utt;
    const f16 *kp = ckv + ((((slot * n_layer + layer) * 2 + 0) * H + head) * (long)Tp) * 64 + e8 * 8;
    const f16 *vp = kp + (long)H * Tp * 64;
    float q[8];
    { const f16x8 qv = *(const f16x8 *)(qb + (long)m * d + head * 64 + e8 * 8);
#pragma unroll
      for (int i = 0; i < 8; i++) q[i] = (float)qv[i]; }
    AttnAcc s; attn_init(s);
    int j = wave * 8 + g;
    for (; j + 32 < T; j += 64) {          // two independent key rows in flight per iteration
        const f16x8 k0 = *(const f16x8 *)(kp + (long)j * 64), v0 = *(const f16x8 *)(vp + (long)j * 64);
        const f16x8 k1 = *(const f16x8 *)(kp + (long)(j + 32) * 64), v1 = *(const f16x8 *)(vp + (long)(j + 32) * 64);
        attn_step(s, q, k0, v0);
        attn_step(s, q, k1, v1);
    }
    for (; j < T; j += 32) attn_step(s, q, *(const f16x8 *)(kp + (long)j * 64), *(const f16x8 *)(vp + (long)j * 64));
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
            l += red[w][tid][1] * f;
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] += red[w][tid][2 + i] * f;
        }
        const float inv = 1.0f / l;
        f16x8 o;
#pragma unroll
        for (int i = 0; i < 8; i++) o[i] = (f16)(a[i] * inv);
        *(f16x8 *)(att + (long)m * d + head * 64 + tid * 8) = o;
    }
}
bool dec_cross_attention(const f16 *q, const f16 *ckv, const DecRow *rows, int M, int d, int H, int T, int Tp, int n_layer, int layer, f16 *att, hipStream_t st) {
    if (M <= 0) return true;
    hipLaunchKernelGGL(dec_cross_attention_kernel, dim3(H, M), dim3(256), 0, st, q, ckv, rows, d, H, T, Tp, n_layer, layer, att);
    return hipGetLastError() == hipSuccess;
}

// ------------------------------------------------------------------------------------------ logits rules + selection
// One workgroup per logit row. Restates whisper_process_logits + whisper_sample_token(best):
// suppression rules, timestamp pairing / monotonicity / initial bound, log-softmax, the
// "timestamp mass beats every text token" rule, then argmax (lowest id on ties) and timestamp
// statistics, all with wavefront reductions. Pass 1 writes the masked logits back in place so the
// later passes (sum, optional top-k, optional probabilities) are plain reads that hit in L2.
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

__global__ __launch_bounds__(256) void sample_rows_kernel(float *__restrict__ logits, long ld, const SampleRow *__restrict__ srow, VocabIds vi,
                                                          const unsigned *__restrict__ nst_bitmap, SampleOut *__restrict__ out,
                                                          TopkOut *__restrict__ topk, int topk_stride, float *__restrict__ probs) {
    __shared__ ArgMax sh_a[4];
    __shared__ float sh_f[4];
    const int r = blockIdx.x, tid = threadIdx.x, V = vi.n_vocab;
    const SampleRow sr = srow[r];
    float *lg = logits + (long)r * ld;
    const int fl = sr.flags;
    const bool is_initial = fl & 1, last_ts = fl & 2, pen_ts = fl & 4, has_ts = fl & 8, sup_blank = fl & 16, no_ts = fl & 32,
               sup_nst = fl & 64, tdrz = fl & 128, raw = fl & 256;
    const float inv_t = sr.temperature > 0.0f ? 1.0f / sr.temperature : 1.0f;
    ArgMax mt = {-INFINITY, 0x7fffffff}, ms = {-INFINITY, 0x7fffffff};      // text / timestamp maxima
    for (int i = tid; i < V; i += 256) {
        float v = lg[i];
        if (sr.temperature > 0.0f) v = v / sr.temperature;
        (void)inv_t;
        if (!raw) {
            bool kill = false;
            if (sup_blank && is_initial && (i == vi.eot || i == vi.space)) kill = true;
            if (i == vi.tok_not || i == vi.sot || i == vi.nosp || i == vi.translate || i == vi.transcribe || i == vi.prev) kill = true;
            if (i == vi.solm && !tdrz) kill = true;
            if (no_ts && i >= vi.beg) kill = true;
            if (i > vi.sot && i <= vi.sot + 100) kill = true;                       // language tokens
            if (sup_nst && ((nst_bitmap[i >> 5] >> (i & 31)) & 1)) kill = true;
            if (sr.force_n > 0) {
                if (sr.n_sampled < sr.force_n) { if (i == vi.eot) kill = true; }
                else { if (i != vi.eot) kill = true; else { kill = false; v = 0.0f; } }
            }
            if (last_ts) { if (pen_ts) { if (i >= vi.beg) kill = true; } else if (i < vi.eot) kill = true; }
            if (is_initial && sr.max_initial >= 0 && i > vi.beg + sr.max_initial) kill = true;
            if (has_ts && i >= vi.beg && i < vi.beg + sr.ts_min) kill = true;
            if (kill) v = -INFINITY;
            lg[i] = v;
        }
        if (i < vi.beg) { if (v > mt.v) { mt.v = v; mt.i = i; } }
        else if (v > ms.v) { ms.v = v; ms.i = i; }
    }
    mt = block_amax(mt, sh_a);
    ms = block_amax(ms, sh_a);
    const float mx = fmaxf(mt.v, ms.v);
    float st_ = 0.0f, ss_ = 0.0f;
    for (int i = tid; i < V; i += 256) {
        const float v = lg[i];
        if (v > -INFINITY) { const float e = __expf(v - mx); if (i < vi.beg) st_ += e; else ss_ += e; }
    }
    st_ = block_sum(st_, sh_f);
    ss_ = block_sum(ss_, sh_f);
    const float lse = logf(st_ + ss_) + mx;
    // timestamp mass rule: log(sum_ts p) > max text logprob  =>  only timestamps remain
    const float ts_lp = ss_ > 0.0f ? logf(ss_) + mx - lse : -INFINITY;
    const bool ts_wins = !raw && ts_lp > mt.v - lse;
    ArgMax best = ts_wins ? ms : amax2(mt, ms);
    if (best.i == 0x7fffffff) best.i = 0;
    if (tid == 0) {
        SampleOut o;
        o.id = best.i; o.plog = best.v - lse; o.p = __expf(o.plog); o.lse = lse; o.pad = ts_wins ? 1 : 0;
        const float sum_ts = ss_ > 0.0f ? __expf(logf(ss_) + mx - lse) : 0.0f;
        const float max_ts = ms.v > -INFINITY ? __expf(ms.v - lse) : 0.0f;
        o.tid = max_ts > 0.0f ? ms.i : vi.beg;
        o.pt = max_ts / (sum_ts + 1e-10f); o.ptsum = sum_ts;
        if (o.id >= vi.beg) { o.tid = o.id; o.pt = o.p; }
        out[r] = o;
    }
    if (probs && sr.want_probs) {
        float *pr = probs + (long)r * V;
        for (int i = tid; i < V; i += 256) {
            const float v = lg[i];
            pr[i] = (v == -INFINITY || (ts_wins && i < vi.beg)) ? 0.0f : __expf(v - lse);
        }
    }
    if (topk && sr.topk > 0) {
        int chosen[8];
        for (int k = 0; k < sr.topk && k < 8; k++) {
            ArgMax a = {-INFINITY, 0x7fffffff};
            for (int i = (ts_wins ? vi.beg : 0) + tid; i < V; i += 256) {
                const float v = lg[i];
                bool used = false;
                for (int u = 0; u < k; u++) used |= chosen[u] == i;
                if (!used && v > a.v) { a.v = v; a.i = i; }
            }
            a = block_amax(a, sh_a);
            chosen[k] = a.i;
            if (tid == 0) {
                TopkOut t; t.id = a.i == 0x7fffffff ? -1 : a.i; t.plog = a.v - lse; t.p = a.v > -INFINITY ? __expf(t.plog) : 0.0f; t.pad = 0;
                topk[(long)r * topk_stride + k] = t;
            }
        }
    }
}
bool sample_rows(float *logits, long ld, int R, const SampleRow *srow, const VocabIds &vi, const unsigned *nst_bitmap,
                 SampleOut *out, TopkOut *topk, int topk_stride, float *probs, hipStream_t st) {
    if (R <= 0) return true;
    hipLaunchKernelGGL(sample_rows_kernel, dim3(R), dim3(256), 0, st, logits, ld, srow, vi, nst_bitmap, out, topk, topk_stride, probs);
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

}  // namespace wmi
