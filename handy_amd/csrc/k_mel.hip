// Log-mel frontend for gfx950 (restates whisper.cpp's log_mel_spectrogram semantics):
//   padded = [reflect(samples[1..200]) | samples | zeros], frame i = padded[160 i .. 160 i + 400) * Hann(400, periodic)
//   400-point real DFT -> 201 power bins -> filterbank (double accumulation) -> log10(max(., 1e-10))
//   then, per utterance, clamp to (max - 8) and map to (x + 4) / 4.
//
// One workgroup (256 threads) produces 32 consecutive frames:
//   1. the 5360 input samples the 32 frames share are read once from HBM into LDS (coalesced);
//   2. each 400-sample real frame is packed into a 200-point complex sequence and transformed by a
//      fp32 mixed-radix FFT (200 = 8 x 25). The radix-8 pass applies the Hann window as it reads the
//      samples and keeps its window values and twiddles in registers (a thread serves one residue
//      n2 for several frames); the 25-point transforms run entirely in registers (5 x 5, constant
//      twiddles), so a frame crosses LDS twice instead of four times;
//   3. the real-FFT unpacking computes bins k and 200-k from the same two values and leaves the 201
//      power values in place of the transform;
//   4. the filterbank is applied over each mel bin's nonzero support only (bit-identical to the
//      dense sum, whose other terms are exact zeros), accumulated in double like the reference;
//   5. raw log10 values go to HBM mel-major (128 B per mel row per workgroup) and the per-utterance
//      maximum is folded with one atomicMax per workgroup. The clamp/scale pass is fused into
//      whatever consumes the mel (encoder window extraction) because it needs the global maximum.
// Bound: HBM. Algorithmic bytes per 30 s utterance: 480000*4 in + n_mel*3000*4 out.
#include "kernels.h"

#include <algorithm>
#include <atomic>
#include <mutex>

namespace wmi {

constexpr int F = kMelFramesPerBlock;          // frames per workgroup
constexpr int NS = (F - 1) * kHop + kNFft;     // samples shared by the frames
constexpr int ZS = 201;                        // complex elements per frame row in LDS (odd: frames land on different banks)

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }   // a * (-i)
__device__ __forceinline__ float2 mul_pi(float2 a) { return make_float2(-a.y, a.x); }   // a * (+i)

__device__ __forceinline__ void dft4(float2 a, float2 b, float2 c, float2 d, float2 *o) {
    float2 t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = csub(b, d);
    o[0] = cadd(t0, t2); o[2] = csub(t0, t2);
    o[1] = cadd(t1, mul_mi(t3)); o[3] = cadd(t1, mul_pi(t3));
}
__device__ __forceinline__ void dft8(float2 *v) {
    float2 e[4], o[4];
    dft4(v[0], v[2], v[4], v[6], e);
    dft4(v[1], v[3], v[5], v[7], o);
    const float r = 0.70710678118654752440f;
    o[1] = make_float2(r * (o[1].x + o[1].y), r * (o[1].y - o[1].x));      // * (1 - i)/sqrt2
    o[2] = mul_mi(o[2]);                                                      // * (-i)
    o[3] = make_float2(r * (o[3].y - o[3].x), -r * (o[3].x + o[3].y));     // * (-1 - i)/sqrt2
#pragma unroll
    for (int k = 0; k < 4; k++) { v[k] = cadd(e[k], o[k]); v[k + 4] = csub(e[k], o[k]); }
}
__device__ __forceinline__ void dft5(float2 *v) {
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
    const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
    float2 t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]), t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
    float2 x0 = make_float2(v[0].x + t1.x + t2.x, v[0].y + t1.y + t2.y);
    float2 m1 = make_float2(v[0].x + c1 * t1.x + c2 * t2.x, v[0].y + c1 * t1.y + c2 * t2.y);
    float2 m2 = make_float2(v[0].x + c2 * t1.x + c1 * t2.x, v[0].y + c2 * t1.y + c1 * t2.y);
    float2 n1 = make_float2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y);
    float2 n2 = make_float2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y);
    v[0] = x0;
    v[1] = cadd(m1, mul_mi(n1)); v[4] = cadd(m1, mul_pi(n1));
    v[2] = cadd(m2, mul_mi(n2)); v[3] = cadd(m2, mul_pi(n2));
}

__device__ __forceinline__ unsigned ordered_bits(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

// (cos, sin)(2 pi j / 25), the float values of the 400-entry table at 16 j
__device__ constexpr float kW25[25][2] = {
    {1.0f, 0.0f}, {0.968583167f, 0.24868989f}, {0.876306653f, 0.481753677f},
    {0.72896862f, 0.684547126f}, {0.535826802f, 0.844327927f}, {0.309017003f, 0.95105654f},
    {0.0627905205f, 0.998026729f}, {-0.187381312f, 0.982287228f}, {-0.425779283f, 0.904827058f},
    {-0.637423992f, 0.770513237f}, {-0.809017003f, 0.587785244f}, {-0.92977649f, 0.368124545f},
    {-0.992114723f, 0.125333235f}, {-0.992114723f, -0.125333235f}, {-0.92977649f, -0.368124545f},
    {-0.809017003f, -0.587785244f}, {-0.637423992f, -0.770513237f}, {-0.425779283f, -0.904827058f},
    {-0.187381312f, -0.982287228f}, {0.0627905205f, -0.998026729f}, {0.309017003f, -0.95105654f},
    {0.535826802f, -0.844327927f}, {0.72896862f, -0.684547126f}, {0.876306653f, -0.481753677f},
    {0.968583167f, -0.24868989f},
};

// |X[k]|^2 of the 400-point real transform from the packed 200-point one: X[k] = E[k] + W400^k O[k],
// E = (Z[k] + conj Z[200-k])/2, O = (Z[k] - conj Z[200-k])/(2i); a = Z[k], b = Z[200-k], w = (cos, sin)(2 pi k/400)
__device__ __forceinline__ float unpack_power(float2 a, float2 b, float2 w) {
    b.y = -b.y;
    const float2 e = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y + b.y));
    const float2 dlt = make_float2(0.5f * (a.x - b.x), 0.5f * (a.y - b.y));
    const float2 o = mul_mi(dlt);
    const float2 xo = cmul(o, make_float2(w.x, -w.y));
    const float re = e.x + xo.x, im = e.y + xo.y;
    return re * re + im * im;
}

constexpr int NSR = (NS + 255) / 256;          // sample loads per thread
constexpr int kMaxPackedFilter = 1024;         // nonzero-support weights staged in LDS (Whisper's filterbanks hold ~400)

// PACKED: the filterbank's nonzero supports, concatenated, sit in LDS (meta word per mel bin: lo | count << 8 | offset << 16);
// otherwise (a model file with a dense or very wide table) the weights are read from the dense table in global memory.
// The utterance descriptors come from `utts` (one per blockIdx.y) or, when that is null, from the kernel argument `single`
// (a launch for one utterance then needs no descriptor in device memory).
template <bool PACKED>
__global__ __launch_bounds__(256) void mel_raw_kernel(const float *__restrict__ pcm, const MelUtt *__restrict__ utts, MelUtt single, MelTables tb,
                                                      float *__restrict__ mel_out, unsigned *__restrict__ mel_max) {
    __shared__ __align__(16) float smp[NSR * 256];
    __shared__ float2 z[F * ZS];
    __shared__ float fw[PACKED ? kMaxPackedFilter : 1];
    __shared__ int fmeta[PACKED ? 128 : 1];
    __shared__ float wmax[4];
    const MelUtt u = utts ? utts[blockIdx.y] : single;
    const int f0 = blockIdx.x * F;
    if (f0 >= u.n_frames) return;
    const int tid = threadIdx.x;
    float *out = mel_out + u.mel_off;
    if (f0 >= u.n_calc) {                      // frames that see no audio: power 0 -> log10(1e-10)
        for (int i = tid; i < F * tb.n_mel; i += 256) out[(long)(i / F) * u.n_frames + f0 + (i % F)] = -10.0f;
        if (tid == 0) atomicMax(mel_max + blockIdx.y, ordered_bits(-10.0f));
        return;
    }
    // The samples of this workgroup's frames: padded position p = 160 f0 + i holds sample |p - 200| (reflection on the left),
    // zero past the end. All loads are issued before the first is consumed; the buffer descriptor's range check supplies the
    // zeros (its record count is the number of samples left from the first one this workgroup can touch).
    {
        const long s0 = f0 ? (long)f0 * kHop - 200 : 0;
        const long left = (long)u.n - s0;
        const int nrec = left <= 0 ? 0 : (left > NSR * 256 ? NSR * 256 : (int)left);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(pcm + u.pcm_off + s0), 0, nrec * 4, 0x00020000);
        float v[NSR];
#pragma unroll
        for (int r = 0; r < NSR; r++) {
            const int i = tid + 256 * r;
            const int idx = f0 ? i : (i < 200 ? 200 - i : i - 200);
            v[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, idx * 4, 0, 0));
        }
        if (PACKED) {
            for (int i = tid; i < tb.n_packed; i += 256) fw[i] = tb.fpacked[i];
            if (tid < tb.n_mel) fmeta[tid] = tb.fmeta[tid];
        }
#pragma unroll
        for (int r = 0; r < NSR; r++) smp[tid + 256 * r] = v[r];
    }
    // register tables: pass 1's window values and twiddles for residue n2, the unpacking pass's twiddles for bins k and 200-k
    const int n2 = tid % 25, fg = tid / 25;
    float2 hw[8], w8[8];
#pragma unroll
    for (int a = 0; a < 8; a++) {
        hw[a] = make_float2(tb.hann[50 * a + 2 * n2], tb.hann[50 * a + 2 * n2 + 1]);
        const float2 w = tb.tw400[(2 * n2 * a) % 400];
        w8[a] = make_float2(w.x, -w.y);
    }
    const int uk = tid % 101, ug = tid / 101;
    const float2 wk = tb.tw400[uk], wr = tb.tw400[200 - uk];
    __syncthreads();
    // pass 1: packed sequence z[m] = w[2m] x[2m] + i w[2m+1] x[2m+1], m = 25 a + n2; radix-8 over a, twiddle W200^(n2 k1),
    // stored at [k1][n2]. Thread (n2, fg) serves frames fg, fg + 8, ...
    if (fg < 8) {
#pragma unroll 2
        for (int f = fg; f < F; f += 8) {
            const float *sp = smp + f * kHop + 2 * n2;
            float2 v[8];
#pragma unroll
            for (int a = 0; a < 8; a++) {
                const float2 sv = *reinterpret_cast<const float2 *>(sp + 50 * a);
                v[a] = make_float2(hw[a].x * sv.x, hw[a].y * sv.y);
            }
            dft8(v);
#pragma unroll
            for (int k1 = 1; k1 < 8; k1++) v[k1] = cmul(v[k1], w8[k1]);
            float2 *zz = z + f * ZS + n2;
#pragma unroll
            for (int k1 = 0; k1 < 8; k1++) zz[25 * k1] = v[k1];
        }
    }
    __syncthreads();
    // pass 2: thread (frame, k1) transforms its 25 values in registers: radix-5 over a of y[5a+b], twiddle W25^(b c),
    // radix-5 over b; result k = k1 + 8 (c + 5 e). Every thread has read its row before any thread writes.
    {
        const int f = tid >> 3, k1 = tid & 7;
        float2 t[25];
        {
            const float2 *y = z + f * ZS + k1 * 25;
#pragma unroll
            for (int b = 0; b < 5; b++) {
                float2 v[5];
#pragma unroll
                for (int a = 0; a < 5; a++) v[a] = y[5 * a + b];
                dft5(v);
                t[b] = v[0];
#pragma unroll
                for (int c = 1; c < 5; c++) t[5 * c + b] = cmul(v[c], make_float2(kW25[(b * c) % 25][0], -kW25[(b * c) % 25][1]));
            }
        }
        __syncthreads();
        float2 *zo = z + f * ZS + k1;
#pragma unroll
        for (int c = 0; c < 5; c++) {
            float2 v[5];
#pragma unroll
            for (int b = 0; b < 5; b++) v[b] = t[5 * c + b];
            dft5(v);
#pragma unroll
            for (int e = 0; e < 5; e++) zo[8 * c + 40 * e] = v[e];
        }
    }
    __syncthreads();
    // real-FFT unpacking in place: thread (k, g) turns Z[k], Z[200-k] of frames g, g + 2, ... into the two power values;
    // bin k ends up in z[k].x (bin 200 in the row's spare 201st element)
    if (tid < 202) {
#pragma unroll 4
        for (int f = ug; f < F; f += 2) {
            float2 *zf = z + f * ZS;
            const float2 a = zf[uk], b = zf[uk == 0 ? 0 : 200 - uk];
            const float pk = unpack_power(a, b, wk), pr = unpack_power(b, a, wr);
            zf[uk].x = pk;
            if (uk != 100) zf[200 - uk].x = pr;
        }
    }
    __syncthreads();
    float lmax = -1e30f;
    {
        const int f = tid & (F - 1), jg = tid / F;
        const float *pw = reinterpret_cast<const float *>(z + f * ZS);
        // log10 through the hardware log2 (about 1 ulp of fp32, i.e. ~1e-6 absolute here, against a parity bound of 1e-3);
        // a double-precision log10 per output value would be most of this kernel's instructions
        auto emit = [&](int j, double sum) {
            const float v = __builtin_amdgcn_logf((float)(sum > 1e-10 ? sum : 1e-10)) * 0.30102999566398119521f;
            out[(long)j * u.n_frames + f0 + f] = v;
            lmax = fmaxf(lmax, v);
        };
        for (int j = jg; j < tb.n_mel; j += 256 / F) {
            double sum = 0.0;
            if (PACKED) {
                const int m = fmeta[j], cnt = (m >> 8) & 255;
                const float *w = fw + (m >> 16), *p = pw + 2 * (m & 255);
                for (int k = 0; k < cnt; k++) sum += (double)(p[2 * k] * w[k]);
            } else {
                const int lo = tb.flo[j], hi = tb.fhi[j];
                const float *fl = tb.filters + j * kNBins;
                for (int k = lo; k < hi; k++) sum += (double)(pw[2 * k] * fl[k]);
            }
            emit(j, sum);
        }
    }
    for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
    if ((tid & 63) == 0) wmax[tid >> 6] = lmax;
    __syncthreads();
    if (tid == 0) atomicMax(mel_max + blockIdx.y, ordered_bits(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
}

bool mel_raw(const float *pcm, const MelUtt *utts, int n_utt, int max_blocks_per_utt, const MelTables &tb, float *mel_raw_out, unsigned *mel_max, hipStream_t st) {
    if (n_utt <= 0 || max_blocks_per_utt <= 0) return true;
    WMI_HIP_OK(hipMemsetAsync(mel_max, 0, sizeof(unsigned) * n_utt, st));
    if (tb.n_mel > 128) return false;
    if (tb.fpacked) hipLaunchKernelGGL(mel_raw_kernel<true>, dim3(max_blocks_per_utt, n_utt), dim3(256), 0, st, pcm, utts, MelUtt{}, tb, mel_raw_out, mel_max);
    else hipLaunchKernelGGL(mel_raw_kernel<false>, dim3(max_blocks_per_utt, n_utt), dim3(256), 0, st, pcm, utts, MelUtt{}, tb, mel_raw_out, mel_max);
    return hipGetLastError() == hipSuccess;
}

__device__ __forceinline__ float mel_norm_value(float raw, float mmax) {
    // mirrors: mmax (double) -= 8; if (x < mmax) x = mmax; x = (x + 4) / 4
    const double lim = (double)mmax - 8.0;
    float v = raw;
    if ((double)v < lim) v = (float)lim;
    return (float)(((double)v + 4.0) / 4.0);
}

__global__ void mel_normalize_kernel(const float *raw, int n_mel, int n_frames, const unsigned *mel_max, float *out, int n_out) {
    const float mmax = fmaxf(from_ordered_bits(*mel_max), -10.0f);
    const long total = (long)n_mel * n_out;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i / n_out), t = (int)(i - (long)j * n_out);
        const float r = t < n_frames ? raw[(long)j * n_frames + t] : -10.0f;
        out[i] = mel_norm_value(r, mmax);
    }
}
bool mel_normalize(const float *raw, int n_mel, int n_frames, const unsigned *mel_max, float *out, int n_out, hipStream_t st) {
    const long total = (long)n_mel * n_out;
    if (total <= 0) return true;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(mel_normalize_kernel, dim3(blocks), dim3(256), 0, st, raw, n_mel, n_frames, mel_max, out, n_out);
    return hipGetLastError() == hipSuccess;
}

// The clamp / scale pass for a whole batch: out[b][j][t], t < n_out, = the normalised log-mel of utterance b (frames past the
// ones computed carry the floor value). It runs right behind mel_raw_kernel, whose output (184 MB for an hour of audio at
// 128 mel bins) is still in the Infinity Cache; four frames per thread, 16-byte loads and stores.
__global__ __launch_bounds__(256) void mel_normalize_batch_kernel(const float *__restrict__ raw, const MelUtt *__restrict__ utts, int n_mel,
                                                                  const unsigned *__restrict__ mel_max, float *__restrict__ out, int n_out) {
    const MelUtt u = utts[blockIdx.y];
    const float mmax = fmaxf(from_ordered_bits(mel_max[blockIdx.y]), -10.0f);
    const int q_per_row = n_out >> 2;                      // n_out % 4 == 0 (checked by the launcher)
    const long total = (long)n_mel * q_per_row;
    const float *src = raw + u.mel_off;
    float *dst = out + (long)blockIdx.y * n_mel * n_out;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int j = (int)(i / q_per_row), t = (int)(i - (long)j * q_per_row) * 4;
        f32x4 v = {-10.0f, -10.0f, -10.0f, -10.0f};
        if (t + 4 <= u.n_frames) v = *(const f32x4 *)(src + (long)j * u.n_frames + t);          // n_frames % 32 == 0: whole or nothing
        *(f32x4 *)(dst + (long)j * n_out + t) = f32x4{mel_norm_value(v[0], mmax), mel_norm_value(v[1], mmax), mel_norm_value(v[2], mmax), mel_norm_value(v[3], mmax)};
    }
}
bool mel_normalize_batch(const float *raw, const MelUtt *utts, int n_utt, int n_mel, const unsigned *mel_max, float *out, int n_out, hipStream_t st) {
    if (n_utt <= 0) return true;
    if (n_out <= 0 || n_out % 4 != 0) return false;
    const long total = (long)n_mel * (n_out >> 2);
    const int bx = (int)std::min<long>((total + 255) / 256, 96);
    hipLaunchKernelGGL(mel_normalize_batch_kernel, dim3(bx, n_utt), dim3(256), 0, st, raw, utts, n_mel, mel_max, out, n_out);
    return hipGetLastError() == hipSuccess;
}

// ------------------------------------------------------------------------------------------ log-mel without allocations
// The frontend for callers that may not allocate (code under HIP graph capture, serving loops, harnesses that forbid
// hipMalloc): one utterance of at most kMelNoAllocMaxSamples samples, PCM and result in the caller's device memory. What
// the two kernels hand to each other -- the raw log-mel and the utterance maximum, which the library's other entry
// points keep in buffers they allocate -- sits in a fixed array of this code object instead. Every GPU the library runs
// on holds its own copy of the array at its own address until the process ends; the array's name is only a host-side
// handle, so the device address is looked up once per device (that device current) and the kernels get the pointer as
// an argument. The array keeps the previous call's contents: the maximum is cleared per call, the raw log-mel is only
// read where this call wrote it. Calls that overlap on one GPU would share the array and race.
constexpr int kMelNoAllocMaxFrames = ((kMelNoAllocMaxSamples + kNFft / 2) / kHop + 1 + F - 1) / F * F;      // 3008 for 30 s
constexpr size_t kMelScratchHeader = 256;                                                                    // [0, 4): the utterance maximum
constexpr size_t kMelScratchBytes = kMelScratchHeader + (size_t)128 * kMelNoAllocMaxFrames * sizeof(float);  // 1 540 352 bytes
static __device__ __align__(256) unsigned char scratch[kMelScratchBytes];

constexpr int kMaxDevices = 64;
static std::atomic<unsigned char *> g_scratch_ptr[kMaxDevices];
static std::mutex g_scratch_mu;

// Set-up, at context creation: `device` must be the current device. The look-up may load the code object on that device
// (an allocation), which is why it happens here and never inside the entry point.
bool mel_noalloc_init(int device) {
    if (device < 0 || device >= kMaxDevices) return false;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (g_scratch_ptr[device].load(std::memory_order_acquire)) return true;
    void *p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p) { log_msg(2, "mel_noalloc_init: hipGetSymbolAddress failed on device %d: %s\n", device, hipGetErrorString(e)); return false; }
    g_scratch_ptr[device].store((unsigned char *)p, std::memory_order_release);
    return true;
}

// pcm_dev[n] -> out_dev[n_mel][n_out] (normalised log-mel, frames past the audio carry the floor value). `device` is current.
// Launches two kernels and one 256-byte memset on `st`; allocates nothing.
int mel_noalloc(const float *pcm_dev, int n, const MelTables &tb, float *out_dev, int n_out, int device, hipStream_t st) {
    if (!pcm_dev || !out_dev || n < 0 || n_out <= 0 || tb.n_mel <= 0 || tb.n_mel > 128) return -1;
    if (n > kMelNoAllocMaxSamples || n_out > kMelNoAllocMaxFramesOut) {
        log_msg(2, "wmi_pcm_to_mel_device_noalloc: %d samples / %d frames exceed the entry point's maximum of %d samples / %d frames\n", n, n_out, kMelNoAllocMaxSamples, kMelNoAllocMaxFramesOut);
        return -2;
    }
    // the array's address on this device was looked up when the context was created; it is not looked up here (that can allocate)
    unsigned char *ptr = device >= 0 && device < kMaxDevices ? g_scratch_ptr[device].load(std::memory_order_acquire) : nullptr;
    if (!ptr) { log_msg(2, "wmi_pcm_to_mel_device_noalloc: no context has been created on device %d\n", device); return -3; }
    unsigned *mel_max = (unsigned *)ptr;
    float *raw = (float *)(ptr + kMelScratchHeader);
    MelUtt u{};
    u.pcm_off = 0; u.n = n; u.mel_off = 0;
    const int n_len = (n + kSampleRate * kChunkSec) / kHop;
    u.n_calc = (n + kNFft / 2) / kHop + 1; if (u.n_calc > n_len) u.n_calc = n_len;
    u.n_frames = (u.n_calc + F - 1) / F * F;
    if (u.n_frames > kMelNoAllocMaxFrames) return -2;
    if (hipMemsetAsync(ptr, 0, kMelScratchHeader, st) != hipSuccess) return -4;
    if (tb.fpacked) hipLaunchKernelGGL(mel_raw_kernel<true>, dim3(u.n_frames / F, 1), dim3(256), 0, st, pcm_dev, (const MelUtt *)nullptr, u, tb, raw, mel_max);
    else hipLaunchKernelGGL(mel_raw_kernel<false>, dim3(u.n_frames / F, 1), dim3(256), 0, st, pcm_dev, (const MelUtt *)nullptr, u, tb, raw, mel_max);
    if (hipGetLastError() != hipSuccess) return -4;
    if (!mel_normalize(raw, tb.n_mel, u.n_frames, mel_max, out_dev, n_out, st)) return -4;
    return 0;
}

// window extraction + normalisation + transpose to [frame][mel] fp16 (the conv1 operand layout)
template <bool NORMALIZED>
__global__ __launch_bounds__(256) void mel_window_kernel(const float *__restrict__ src, const unsigned *__restrict__ mel_max, const WinDesc *__restrict__ wins,
                                                         int n_mel, int T2, f16 *__restrict__ melT) {
    __shared__ float tile[128 * 65];
    const WinDesc w = wins[blockIdx.y];
    const int t0 = blockIdx.x * 64;
    const float mmax = NORMALIZED ? 0.0f : fmaxf(from_ordered_bits(mel_max[w.max_idx]), -10.0f);
    const float *raw = src + w.mel_off;
    for (int i = threadIdx.x; i < n_mel * 64; i += 256) {
        const int j = i >> 6, tt = i & 63, t = t0 + tt, fr = w.seek + t;
        float v = 0.0f;
        if (t < T2 && fr < w.n_len) {
            if (NORMALIZED) v = raw[(long)j * w.n_frames + fr];
            else v = mel_norm_value(fr < w.n_frames ? raw[(long)j * w.n_frames + fr] : -10.0f, mmax);
        }
        tile[j * 65 + tt] = v;
    }
    __syncthreads();
    f16 *dst = melT + ((long)blockIdx.y * (T2 + 2) + 1 + t0) * n_mel;
    for (int i = threadIdx.x; i < n_mel * 64; i += 256) {
        const int tt = i / n_mel, j = i - tt * n_mel;
        if (t0 + tt < T2) dst[(long)tt * n_mel + j] = (f16)tile[j * 65 + tt];
    }
}
bool mel_window(const float *raw, const unsigned *mel_max, const WinDesc *wins, int n_win, int n_mel, int T2, f16 *melT, hipStream_t st) {
    if (n_win <= 0) return true;
    if (n_mel > 128) return false;
    hipLaunchKernelGGL(mel_window_kernel<false>, dim3((T2 + 63) / 64, n_win), dim3(256), 0, st, raw, mel_max, wins, n_mel, T2, melT);
    return hipGetLastError() == hipSuccess;
}
bool mel_window_normalized(const float *mel, const WinDesc *wins, int n_win, int n_mel, int T2, f16 *melT, hipStream_t st) {
    if (n_win <= 0) return true;
    if (n_mel > 128) return false;
    hipLaunchKernelGGL(mel_window_kernel<true>, dim3((T2 + 63) / 64, n_win), dim3(256), 0, st, mel, nullptr, wins, n_mel, T2, melT);
    return hipGetLastError() == hipSuccess;
}

}  // namespace wmi
