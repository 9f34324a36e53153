// The step in front of the transcription path, for bulk files: resample to 16 kHz in 30 ms frames and gather the frames a
// voice-activity gate lets through -- what Handy's recorder does per 30 ms of microphone audio on the CPU
// (reference: src-tauri/src/audio_toolkit/audio/resampler.rs:16-98, audio_toolkit/vad/smoothed.rs:41-96,
// audio_toolkit/audio/recorder.rs:280-299, managers/audio.rs:402-411). The per-frame speech probability itself (Silero,
// an ONNX model run by the `vad-rs` crate) stays with the caller.
//
// Resampler: rubato's synchronous FFT resampler `FftFixedIn<f32>::new(in_hz, 16000, 1024, 1, 1)` as resampler.rs:24 builds it.
// The input stream is cut into blocks of fft_size_in samples; a block is zero-padded to twice its length, transformed,
// multiplied with the transform of a windowed-sinc low-pass, cut (or zero-extended) to 2 fft_size_out points, transformed back,
// and the second half of every result is added to the first half of the next one.  Every step is linear with real
// coefficients, so one complex transform carries two blocks (one in the real part, one in the imaginary part) from PCM to
// output samples.  One workgroup per pair of blocks; the transforms are Stockham passes of radix 4/2/3/5/7 between two
// LDS buffers (the sizes are multiples of in_hz / gcd and 16000 / gcd: 6144 and 2048 points at 48 kHz, 6174 and 2240 at
// 44.1 kHz).  The tail a pair hands to the next pair's first block meets that block's own samples through one float
// atomic add each onto a zeroed output: two addends, so the sum does not depend on which arrives first.
#include <hip/hip_runtime.h>

#include <cmath>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/wmi.h"
#include "wmi_common.h"

namespace wmi {
namespace {

constexpr int kOutHz = 16000;       // constants::WHISPER_SAMPLE_RATE
constexpr int kChunkIn = 1024;      // resampler.rs:5
constexpr int kFrame = 480;         // 30 ms at 16 kHz (recorder.rs:260-264)
constexpr int kMaxStages = 16;
constexpr int kThreads = 1024;
constexpr size_t kMaxLds = 160 * 1024 - 1024;

struct FftPlan {
    int n, n_stage;
    int radix[kMaxStages];
};

struct ResamplePlan {               // passed to the kernel by value
    int n_in, n_out, new_len;       // fft_size_in, fft_size_out, bins kept
    FftPlan fwd, inv;               // 2 n_in and 2 n_out points
    const float2 *tables;           // exp(-2 pi i m / (2 n_in)), m < n_in | exp(+2 pi i m / (2 n_out)), m < n_out | the filter's
                                    // transform, bins 0 .. new_len-1 (the second half of a twiddle table is minus the first)
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// v <- DFT_R(v) with exp(-2 pi i r q / R) (INV: exp(+...)); R in {2, 3, 4, 5, 7}. The generic form is R*R complex
// multiply-adds with literal roots; after unrolling, the products with 0 and +-1 fold away.
template <int R, bool INV>
__device__ __forceinline__ void dft_small(float2 (&v)[R]) {
    if constexpr (R == 2) {
        const float2 a = v[0], b = v[1];
        v[0] = make_float2(a.x + b.x, a.y + b.y);
        v[1] = make_float2(a.x - b.x, a.y - b.y);
    } else if constexpr (R == 4) {
        const float2 s0 = make_float2(v[0].x + v[2].x, v[0].y + v[2].y), d0 = make_float2(v[0].x - v[2].x, v[0].y - v[2].y);
        const float2 s1 = make_float2(v[1].x + v[3].x, v[1].y + v[3].y), d1 = make_float2(v[1].x - v[3].x, v[1].y - v[3].y);
        // forward: -i d1 = (d1.y, -d1.x); inverse: +i d1 = (-d1.y, d1.x)
        const float2 j1 = INV ? make_float2(-d1.y, d1.x) : make_float2(d1.y, -d1.x);
        v[0] = make_float2(s0.x + s1.x, s0.y + s1.y);
        v[1] = make_float2(d0.x + j1.x, d0.y + j1.y);
        v[2] = make_float2(s0.x - s1.x, s0.y - s1.y);
        v[3] = make_float2(d0.x - j1.x, d0.y - j1.y);
    } else {
        float c[R], s[R];               // cos, sin of 2 pi k / R
        if constexpr (R == 3) {
            const float cc[3] = {1.0f, -0.5f, -0.5f}, ss[3] = {0.0f, 0.86602540378443865f, -0.86602540378443865f};
#pragma unroll
            for (int k = 0; k < R; ++k) { c[k] = cc[k]; s[k] = ss[k]; }
        } else if constexpr (R == 5) {
            const float cc[5] = {1.0f, 0.30901699437494742f, -0.80901699437494742f, -0.80901699437494742f, 0.30901699437494742f};
            const float ss[5] = {0.0f, 0.95105651629515357f, 0.58778525229247313f, -0.58778525229247313f, -0.95105651629515357f};
#pragma unroll
            for (int k = 0; k < R; ++k) { c[k] = cc[k]; s[k] = ss[k]; }
        } else {
            static_assert(R == 7, "radix");
            const float cc[7] = {1.0f, 0.62348980185873353f, -0.22252093395631440f, -0.90096886790241913f, -0.90096886790241913f, -0.22252093395631440f, 0.62348980185873353f};
            const float ss[7] = {0.0f, 0.78183148246802981f, 0.97492791218182361f, 0.43388373911755812f, -0.43388373911755812f, -0.97492791218182361f, -0.78183148246802981f};
#pragma unroll
            for (int k = 0; k < R; ++k) { c[k] = cc[k]; s[k] = ss[k]; }
        }
        float2 o[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            float2 acc = v[0];
#pragma unroll
            for (int r = 1; r < R; ++r) {
                const int k = (r * q) % R;
                const float wr = c[k], wi = INV ? s[k] : -s[k];
                acc.x += v[r].x * wr - v[r].y * wi;
                acc.y += v[r].x * wi + v[r].y * wr;
            }
            o[q] = acc;
        }
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = o[q];
    }
}

// Where a pass reads from and writes to. The first forward pass takes the PCM of the two blocks straight from global memory
// (the upper half of the zero-padded input is never stored anywhere), the last forward pass multiplies with the filter and
// drops or moves the bins as it writes, the last inverse pass adds the halves and writes the output samples.
struct LdsSrc {
    const float2 *p;
    __device__ __forceinline__ float2 operator()(int i) const { return p[i]; }
};
struct PcmSrc {
    const float *p0, *p1;
    int left0, left1;                   // samples of block 0 / block 1 that exist (0 .. n_in)
    __device__ __forceinline__ float2 operator()(int i) const {
        float2 v = make_float2(0.0f, 0.0f);
        if (i < left0) v.x = p0[i];
        if (i < left1) v.y = p1[i];
        return v;
    }
};
struct LdsSink {
    float2 *p;
    template <int R>
    __device__ __forceinline__ void put(int o, int Ns, const float2 (&v)[R]) const {
#pragma unroll
        for (int r = 0; r < R; ++r) p[o + r * Ns] = v[r];
    }
};
// bins below new_len times the filter; their mirror images times its conjugate, moved to the end of the 2 n_out-point spectrum;
// everything else is dropped
struct SpectrumSink {
    float2 *z;
    const float2 *filt;
    int new_len, n2i, n2o;
    template <int R>
    __device__ __forceinline__ void put(int o, int Ns, const float2 (&v)[R]) const {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int k = o + r * Ns;
            if (k < new_len) {
                const float2 f = filt[k];
                z[k] = cmul(v[r], f);
                if (2 * k == n2i) z[n2o - k] = cmul(v[r], make_float2(f.x, -f.y));      // the input's Nyquist bin is its own mirror image (kept when interpolating)
            } else {
                const int kk = n2i - k;
                if (kk < new_len) { const float2 f = filt[kk]; z[n2o - kk] = cmul(v[r], make_float2(f.x, -f.y)); }
            }
        }
    }
};
// The last inverse pass has an even radix, so the thread that holds sample n of the result also holds sample n + n_out.
struct OutSink {
    float *o0, *o1;
    int n_out;
    bool has1, has2;
    template <int R>
    __device__ __forceinline__ void put(int o, int Ns, const float2 (&v)[R]) const {
        if constexpr (R % 2 == 0) {
#pragma unroll
            for (int r = 0; r < R / 2; ++r) {
                const int n = o + r * Ns;
                const float2 h = v[r], t = v[r + R / 2];
                atomicAdd(o0 + n, h.x);                             // + the tail of block 2w-1, from another workgroup
                if (has1) o1[n] = h.y + t.x;
                if (has2) atomicAdd(o1 + n_out + n, t.y);           // the tail of block 2w+1, for block 2w+2
            }
        }
    }
};

// One Stockham pass: butterfly j takes in[j + r n/R] times tw^(r (j mod Ns)) and writes out[(j - j mod Ns) R + j mod Ns + r Ns].
// tw is the first half of the table exp(-+2 pi i m / n) (the second half is minus the first).
template <int R, bool INV, class Src, class Sink>
__device__ __forceinline__ void fft_pass(int n, int Ns, const float2 *__restrict__ tw, int tid, const Src &src, const Sink &sink) {
    const int m = n / R;
    const int tstep = m / Ns;
    const int half = n >> 1;
    const float inv_ns = 1.0f / (float)Ns;
    for (int j = tid; j < m; j += kThreads) {
        const int k = j - (int)(((float)j + 0.5f) * inv_ns) * Ns;       // j mod Ns: j < 2^20, so the product is off by far less than 0.5 / Ns
        float2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = src(j + r * m);
        if (k > 0) {
#pragma unroll
            for (int r = 1; r < R; ++r) {
                const int mi = r * k * tstep;
                const float2 w = tw[mi >= half ? mi - half : mi];
                v[r] = cmul(v[r], mi >= half ? make_float2(-w.x, -w.y) : w);
            }
        }
        dft_small<R, INV>(v);
        sink.template put<R>((j - k) * R + k, Ns, v);
    }
}

template <bool INV, class Src, class Sink>
__device__ __forceinline__ void fft_pass_any(int R, int n, int Ns, const float2 *__restrict__ tw, int tid, const Src &src, const Sink &sink) {
    switch (R) {
        case 4: fft_pass<4, INV>(n, Ns, tw, tid, src, sink); break;
        case 2: fft_pass<2, INV>(n, Ns, tw, tid, src, sink); break;
        case 3: fft_pass<3, INV>(n, Ns, tw, tid, src, sink); break;
        case 5: fft_pass<5, INV>(n, Ns, tw, tid, src, sink); break;
        default: fft_pass<7, INV>(n, Ns, tw, tid, src, sink); break;
    }
}

// pcm[n_valid] (zeros beyond) -> out[n_blocks * n_out]: pair w is blocks 2w and 2w+1; the workgroups are persistent and stride over
// the pairs with both twiddle tables and the filter in LDS. `out` must be zero before the launch: the first block of a pair
// shares its samples with the tail of the pair before it. Both transforms have at least two passes (1024 points or more).
__global__ __launch_bounds__(kThreads) void resample_blocks_kernel(const float *__restrict__ pcm, long long n_valid, ResamplePlan P, long long n_blocks, float *__restrict__ out) {
    extern __shared__ float2 lds[];
    const int n2i = 2 * P.n_in, n2o = 2 * P.n_out;
    const int nmax = n2i > n2o ? n2i : n2o;
    float2 *a = lds, *b = lds + nmax;
    float2 *twf = b + nmax, *twi = twf + P.n_in, *filt = twi + P.n_out;
    const int tid = threadIdx.x;
    for (int i = tid; i < P.n_in + P.n_out + P.new_len; i += kThreads) twf[i] = P.tables[i];
    const long long n_pairs = (n_blocks + 1) / 2;
    for (long long w = blockIdx.x; w < n_pairs; w += gridDim.x) {
        const long long b0 = 2 * w, b1 = b0 + 1;
        const bool has1 = b1 < n_blocks, has2 = b1 + 1 < n_blocks;
        const long long l0 = n_valid - b0 * P.n_in, l1 = has1 ? n_valid - b1 * P.n_in : 0;
        PcmSrc pcm_src;
        pcm_src.p0 = pcm + b0 * P.n_in; pcm_src.p1 = pcm + (has1 ? b1 : b0) * P.n_in;
        pcm_src.left0 = (int)(l0 < 0 ? 0 : l0 > P.n_in ? P.n_in : l0);
        pcm_src.left1 = (int)(l1 < 0 ? 0 : l1 > P.n_in ? P.n_in : l1);
        float2 *cur = a, *oth = b;
        int Ns = 1;
        fft_pass_any<false>(P.fwd.radix[0], n2i, Ns, twf, tid, pcm_src, LdsSink{cur});
        Ns *= P.fwd.radix[0];
        __syncthreads();
        for (int s = 1; s + 1 < P.fwd.n_stage; ++s) {
            fft_pass_any<false>(P.fwd.radix[s], n2i, Ns, twf, tid, LdsSrc{cur}, LdsSink{oth});
            Ns *= P.fwd.radix[s];
            __syncthreads();
            float2 *t = cur; cur = oth; oth = t;
        }
        for (int k = P.new_len + tid; k <= n2o - P.new_len; k += kThreads) oth[k] = make_float2(0.0f, 0.0f);       // the bins nobody writes
        fft_pass_any<false>(P.fwd.radix[P.fwd.n_stage - 1], n2i, Ns, twf, tid, LdsSrc{cur}, SpectrumSink{oth, filt, P.new_len, n2i, n2o});
        __syncthreads();
        { float2 *t = cur; cur = oth; oth = t; }
        Ns = 1;
        for (int s = 0; s + 1 < P.inv.n_stage; ++s) {
            fft_pass_any<true>(P.inv.radix[s], n2o, Ns, twi, tid, LdsSrc{cur}, LdsSink{oth});
            Ns *= P.inv.radix[s];
            __syncthreads();
            float2 *t = cur; cur = oth; oth = t;
        }
        fft_pass_any<true>(P.inv.radix[P.inv.n_stage - 1], n2o, Ns, twi, tid, LdsSrc{cur}, OutSink{out + b0 * P.n_out, out + b1 * P.n_out, P.n_out, has1, has2});
        __syncthreads();
    }
}

// out[i * 480 + t] = frames[src[i] * 480 + t]; both 16-byte aligned (a frame is 1920 bytes)
__global__ __launch_bounds__(256) void gather_frames_kernel(const float4 *__restrict__ frames, const long long *__restrict__ src, long long n_src, float4 *__restrict__ out) {
    constexpr int Q = kFrame / 4;
    const long long total = n_src * Q;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long i = e / Q;
        const int q = (int)(e - i * Q);
        out[e] = frames[src[i] * Q + q];
    }
}

// n = product of radices 4, 2, 3, 5, 7; even_last: the odd ones first and a 2 or 4 at the end (the pass that writes the output
// samples needs sample n and sample n + n/2 in one thread).
bool factor_radices(int n, FftPlan &fp, bool even_last) {
    fp.n = n; fp.n_stage = 0;
    const int fwd_order[5] = {4, 2, 3, 5, 7}, inv_order[5] = {3, 5, 7, 2, 4};
    int twos = 0;
    while (n % 2 == 0) { ++twos; n /= 2; }
    int count[8] = {0};
    count[4] = twos / 2; count[2] = twos % 2;
    for (int r : {3, 5, 7}) while (n % r == 0) { ++count[r]; n /= r; }
    if (n != 1 || (even_last && twos == 0)) return false;
    for (int r : (even_last ? inv_order : fwd_order))
        for (int c = 0; c < count[r]; ++c) {
            if (fp.n_stage == kMaxStages) return false;
            fp.radix[fp.n_stage++] = r;
        }
    return fp.n_stage >= 2;
}

int gcd_int(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }

// rubato FftFixedIn::new(in_hz, out_hz, chunk_size_in = 1024, sub_chunks = 1, channels = 1): the transform block is the
// requested chunk rounded up to a whole number of output periods of the rate ratio.
void fft_sizes(int in_hz, int &n_in, int &n_out) {
    const int g = gcd_int(in_hz, kOutHz);
    const int min_chunk_out = kOutHz / g;
    const int fft_chunks = (kChunkIn + min_chunk_out - 1) / min_chunk_out;
    n_in = (int)((long long)fft_chunks * in_hz / g);
    n_out = (int)((long long)fft_chunks * kOutHz / g);
}

struct DevicePlan {
    ResamplePlan p{};
    size_t lds = 0;
    int n_cu = 256;
    bool ok = false;
};
std::mutex g_plan_mu;
std::map<std::pair<int, int>, DevicePlan> g_plans;      // (device, in_hz); a plan's device memory lives until the process ends

// The low-pass of rubato's FftResampler::new: make_sincs(fft_size_in, 1, cutoff, BlackmanHarris2) -- a sinc centred at
// fft_size_in / 2 under the squared periodic 4-term Blackman-Harris window, unit sum -- scaled by 1 / (2 fft_size_in) for the two
// unnormalised transforms, zero-padded to 2 fft_size_in and transformed. Evaluated in double here.
void filter_spectrum(int n_in, int n_out, int new_len, std::vector<float2> &F) {
    const double pi = 3.14159265358979323846;
    double cutoff = std::pow(0.4, 16.0 / n_in);
    if (n_in > n_out) cutoff *= (double)n_out / n_in;
    std::vector<double> h(n_in);
    double sum = 0.0;
    for (int j = 0; j < n_in; ++j) {
        const double x = (double)j / n_in;
        double w = 0.35875 - 0.48829 * std::cos(2 * pi * x) + 0.14128 * std::cos(4 * pi * x) - 0.01168 * std::cos(6 * pi * x);
        w *= w;
        const double t = ((double)j - (double)(n_in / 2)) * cutoff;
        const double sc = t == 0.0 ? 1.0 : std::sin(pi * t) / (pi * t);
        h[j] = w * sc;
        sum += h[j];
    }
    const int n2 = 2 * n_in;
    std::vector<double> ct(n2), st(n2);
    for (int m = 0; m < n2; ++m) { ct[m] = std::cos(2 * pi * m / n2); st[m] = std::sin(2 * pi * m / n2); }
    const double scale = 1.0 / (sum * n2);
    F.resize(new_len);
    for (int k = 0; k < new_len; ++k) {
        double re = 0.0, im = 0.0;
        int idx = 0;                                    // (j k) mod n2
        for (int j = 0; j < n_in; ++j) {
            re += h[j] * ct[idx];
            im -= h[j] * st[idx];
            idx += k; if (idx >= n2) idx -= n2;
        }
        F[k] = make_float2((float)(re * scale), (float)(im * scale));
    }
}

const DevicePlan *get_plan(int in_hz) {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto key = std::make_pair(device, in_hz);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) return &it->second;
    // built on the side and entered into the table only when every step has succeeded: a rate that cannot be supported is
    // refused again next time by the same checks, and a transient failure (out of memory) does not stick to the rate
    DevicePlan dp;
    ResamplePlan &p = dp.p;
    fft_sizes(in_hz, p.n_in, p.n_out);
    p.new_len = p.n_in > p.n_out ? p.n_out : p.n_in + 1;
    const int n2i = 2 * p.n_in, n2o = 2 * p.n_out;
    dp.lds = ((size_t)2 * (n2i > n2o ? n2i : n2o) + p.n_in + p.n_out + p.new_len) * sizeof(float2);
    if (!factor_radices(n2i, p.fwd, false) || !factor_radices(n2o, p.inv, true) || dp.lds > kMaxLds) {
        log_msg(2, "wmi_ingest: %d Hz needs transforms of %d and %d points (radices 2, 3, 5, 7 only, %zu bytes of LDS at most): not supported\n", in_hz, n2i, n2o, kMaxLds);
        return nullptr;
    }
    const double pi = 3.14159265358979323846;
    std::vector<float2> host((size_t)p.n_in + p.n_out + p.new_len);
    for (int m = 0; m < p.n_in; ++m) host[m] = make_float2((float)std::cos(2 * pi * m / n2i), (float)-std::sin(2 * pi * m / n2i));
    for (int m = 0; m < p.n_out; ++m) host[p.n_in + m] = make_float2((float)std::cos(2 * pi * m / n2o), (float)std::sin(2 * pi * m / n2o));
    std::vector<float2> F;
    filter_spectrum(p.n_in, p.n_out, p.new_len, F);
    for (int k = 0; k < p.new_len; ++k) host[(size_t)p.n_in + p.n_out + k] = F[k];
    float2 *dev = nullptr;
    if (hipMalloc((void **)&dev, host.size() * sizeof(float2)) != hipSuccess) return nullptr;
    if (hipMemcpy(dev, host.data(), host.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(dev); return nullptr; }
    p.tables = dev;
    if (hipFuncSetAttribute((const void *)resample_blocks_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds) != hipSuccess) { (void)hipFree(dev); return nullptr; }
    int n_cu = 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n_cu > 0) dp.n_cu = n_cu;
    dp.ok = true;
    return &(g_plans[key] = dp);
}

// samples the resampler returns for n_in input samples (whole transform blocks of the input zero-padded to whole 1024-chunks)
long long stream_len(int in_hz, long long n_in) {
    if (in_hz == kOutHz) return n_in;
    int bi, bo;
    fft_sizes(in_hz, bi, bo);
    const long long padded = (n_in + kChunkIn - 1) / kChunkIn * kChunkIn;
    return padded / bi * bo;
}

}  // namespace
}  // namespace wmi

using namespace wmi;

extern "C" {

WHISPER_API long long wmi_ingest_n_frames(int in_hz, long long n_in) {
    if (in_hz <= 0 || n_in < 0) return -1;
    return (stream_len(in_hz, n_in) + kFrame - 1) / kFrame;
}

WHISPER_API int wmi_ingest_resample(const float *pcm_dev, long long n_in, int in_hz, float *frames_dev, long long n_frames_cap, void *stream) {
    if (in_hz <= 0 || n_in < 0 || (n_in > 0 && !pcm_dev)) return -1;
    const long long n_stream = stream_len(in_hz, n_in);
    const long long n_frames = (n_stream + kFrame - 1) / kFrame;
    if (n_frames > n_frames_cap) return -2;
    if (n_frames == 0) return 0;
    if (!frames_dev) return -1;
    hipStream_t st = (hipStream_t)stream;
    if (in_hz == kOutHz) {                              // resampler.rs:38-41: no resampler, the frames are cut from the input
        if (hipMemcpyAsync(frames_dev, pcm_dev, (size_t)n_in * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return -4;
        if (n_frames * kFrame > n_in && hipMemsetAsync(frames_dev + n_in, 0, (size_t)(n_frames * kFrame - n_in) * sizeof(float), st) != hipSuccess) return -4;
        return 0;
    }
    const DevicePlan *dp = get_plan(in_hz);
    if (!dp) return -3;
    if (hipMemsetAsync(frames_dev, 0, (size_t)n_frames * kFrame * sizeof(float), st) != hipSuccess) return -4;
    const long long n_blocks = n_stream / dp->p.n_out;
    const long long n_pairs = (n_blocks + 1) / 2;
    const long long n_wg = n_pairs < dp->n_cu ? n_pairs : dp->n_cu;             // one workgroup fills a CU's LDS
    hipLaunchKernelGGL(resample_blocks_kernel, dim3((unsigned)n_wg), dim3(kThreads), dp->lds, st, pcm_dev, n_in, dp->p, n_blocks, frames_dev);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

WHISPER_API long long wmi_ingest_gate_plan(const unsigned char *is_voice, long long n_frames, int prefill, int hangover, int onset, long long *src_out, long long cap) {
    if (n_frames < 0 || prefill < 0 || hangover < 0 || onset < 0 || (n_frames > 0 && !is_voice)) return -1;
    long long n = 0;
    auto emit = [&](long long f) { if (src_out && n < cap) src_out[n] = f; ++n; };
    long long buffered = 0;                 // frames in the pre-roll buffer: the last `buffered` frames up to the current one
    int hang = 0, ons = 0;
    bool in_speech = false;
    for (long long i = 0; i < n_frames; ++i) {
        buffered = buffered < (long long)prefill + 1 ? buffered + 1 : (long long)prefill + 1;
        const bool v = is_voice[i] != 0;
        if (!in_speech && v) {
            if (++ons >= onset) {
                in_speech = true; hang = hangover; ons = 0;
                for (long long f = i - buffered + 1; f <= i; ++f) emit(f);
            }
        } else if (in_speech && v) {
            hang = hangover;
            emit(i);
        } else if (in_speech) {
            if (hang > 0) { --hang; emit(i); }
            else in_speech = false;
        } else {
            ons = 0;
        }
    }
    return n;
}

WHISPER_API int wmi_ingest_gather(const float *frames_dev, long long n_frames, const long long *src, long long n_src, float *out_dev, long long out_cap, long long *n_out, void *stream) {
    if (n_src < 0 || n_frames < 0 || (n_src > 0 && (!src || !frames_dev)) || !n_out) return -1;
    for (long long i = 0; i < n_src; ++i) if (src[i] < 0 || src[i] >= n_frames) return -1;
    long long total = n_src * kFrame;
    const long long padded = (total > 0 && total < kOutHz) ? kOutHz * 5 / 4 : total;       // managers/audio.rs:402-411
    *n_out = padded;
    if (padded > out_cap) return -2;
    if (padded == 0) return 0;
    if (!out_dev || ((uintptr_t)frames_dev & 15) || ((uintptr_t)out_dev & 15)) return -1;
    hipStream_t st = (hipStream_t)stream;
    long long *src_dev = nullptr;
    if (hipMalloc((void **)&src_dev, (size_t)n_src * sizeof(long long)) != hipSuccess) return -4;
    int rc = 0;
    if (hipMemcpyAsync(src_dev, src, (size_t)n_src * sizeof(long long), hipMemcpyHostToDevice, st) != hipSuccess) rc = -4;
    if (rc == 0) {
        const long long want = (total / 4 + 255) / 256;
        const unsigned blocks = (unsigned)(want < 1 ? 1 : want > 8192 ? 8192 : want);
        hipLaunchKernelGGL(gather_frames_kernel, dim3(blocks), dim3(256), 0, st, (const float4 *)frames_dev, src_dev, n_src, (float4 *)out_dev);
        if (hipGetLastError() != hipSuccess) rc = -4;
    }
    if (rc == 0 && padded > total && hipMemsetAsync(out_dev + total, 0, (size_t)(padded - total) * sizeof(float), st) != hipSuccess) rc = -4;
    if (hipStreamSynchronize(st) != hipSuccess) rc = -4;
    (void)hipFree(src_dev);
    return rc;
}

}  // extern "C"
