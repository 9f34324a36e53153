// Load-time expansion of ggml block-quantised tensors to fp16, on the device: the file's blocks are uploaded as they are
// (0.56 to 1.06 bytes per weight) and written out once as the fp16 operands the GEMMs read.
//
// Block formats (32 weights per block; low nibbles are weights 0..15, high nibbles 16..31):
//   type 2 Q4_0 {f16 d, q[16]}                 d*(q-8)            18 bytes
//   type 3 Q4_1 {f16 d, f16 m, q[16]}          d*q+m              20 bytes
//   type 6 Q5_0 {f16 d, u32 qh, q[16]}         d*((q|b<<4)-16)    22 bytes
//   type 7 Q5_1 {f16 d, f16 m, u32 qh, q[16]}  d*(q|b<<4)+m       24 bytes
//   type 8 Q8_0 {f16 d, i8 q[32]}              d*q                34 bytes
// The arithmetic is fp32 -- the product of an integer of at most 8 bits and an fp16 scale is exact there, so fusing it
// with the addition of the minimum changes nothing -- then one rounding to fp16: the same bits a scalar host expansion gives.
#include "kernels.h"

namespace wmi {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int TT> struct BlockFmt;
template <> struct BlockFmt<2> { static constexpr int bytes = 18, qofs = 2, has_min = 0, has_high = 0; };
template <> struct BlockFmt<3> { static constexpr int bytes = 20, qofs = 4, has_min = 1, has_high = 0; };
template <> struct BlockFmt<6> { static constexpr int bytes = 22, qofs = 6, has_min = 0, has_high = 1; };
template <> struct BlockFmt<7> { static constexpr int bytes = 24, qofs = 8, has_min = 1, has_high = 1; };
template <> struct BlockFmt<8> { static constexpr int bytes = 34, qofs = 2, has_min = 0, has_high = 0; };

// One thread per block. Blocks are 2-byte aligned (every block size is even and the staging buffer is 256-byte aligned),
// so the block is read as 16-bit words; the 64 output bytes leave as four 16-byte stores.
template <int TT>
__global__ __launch_bounds__(256) void expand_blocks_kernel(const unsigned short *__restrict__ src, long n_blocks, f16 *__restrict__ dst) {
    using F = BlockFmt<TT>;
    const long b = (long)blockIdx.x * 256 + threadIdx.x;
    if (b >= n_blocks) return;
    const unsigned short *p = src + b * (F::bytes / 2);
    unsigned short w[F::bytes / 2];
#pragma unroll
    for (int i = 0; i < F::bytes / 2; i++) w[i] = p[i];
    f16 dh; __builtin_memcpy(&dh, &w[0], 2);
    const float d = (float)dh;
    float mn = 0.0f;
    if (F::has_min) { f16 mh; __builtin_memcpy(&mh, &w[1], 2); mn = (float)mh; }
    unsigned qh = 0;
    if (F::has_high) qh = (unsigned)w[F::qofs / 2 - 2] | ((unsigned)w[F::qofs / 2 - 1] << 16);
    float o[32];
    if (TT == 8) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const unsigned short v = w[1 + j];
            o[2 * j] = __fmul_rn((float)(signed char)(v & 0xFF), d);
            o[2 * j + 1] = __fmul_rn((float)(signed char)(v >> 8), d);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const unsigned short v = w[F::qofs / 2 + (j >> 1)];
            const int byte = (j & 1) ? (v >> 8) : (v & 0xFF);
            int lo = byte & 0x0F, hi = byte >> 4;
            if (F::has_high) { lo |= ((qh >> j) & 1) << 4; hi |= ((qh >> (j + 16)) & 1) << 4; }
            if (F::has_min) {
                o[j] = __fadd_rn(__fmul_rn((float)lo, d), mn);
                o[j + 16] = __fadd_rn(__fmul_rn((float)hi, d), mn);
            } else {
                const int zero = F::has_high ? 16 : 8;
                o[j] = __fmul_rn((float)(lo - zero), d);
                o[j + 16] = __fmul_rn((float)(hi - zero), d);
            }
        }
    }
    f16x8 *out = (f16x8 *)(dst + b * 32);
#pragma unroll
    for (int c = 0; c < 4; c++) {
        f16x8 v;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            // the fp32 value is pinned in a register: folded into the conversion, the multiply becomes a mixed-precision
            // fma with a +0 addend, which turns the -0 of (zero quant) x (negative scale) into +0
            float x = o[c * 8 + i];
            asm volatile("" : "+v"(x));
            v[i] = (f16)x;
        }
        out[c] = v;
    }
}

// The same blocks kept quantised for the decoder's few-row GEMMs, in a layout whose pieces are aligned: per block one
// word {d, m} (m = 0 for the formats without a minimum), one word of fifth bits (5-bit formats) and the 16 nibble bytes
// (32 int8 for Q8_0) in ggml's order. One thread per block.
template <int TT>
__global__ __launch_bounds__(256) void repack_blocks_kernel(const unsigned short *__restrict__ src, long n_blocks, unsigned *__restrict__ qs,
                                                            unsigned *__restrict__ dm, unsigned *__restrict__ qh) {
    using F = BlockFmt<TT>;
    const long b = (long)blockIdx.x * 256 + threadIdx.x;
    if (b >= n_blocks) return;
    const unsigned short *p = src + b * (F::bytes / 2);
    dm[b] = (unsigned)p[0] | (F::has_min ? (unsigned)p[1] << 16 : 0u);
    if (F::has_high) qh[b] = (unsigned)p[F::qofs / 2 - 2] | ((unsigned)p[F::qofs / 2 - 1] << 16);
    constexpr int NW = TT == 8 ? 8 : 4;                         // 32-bit words of quants per block
#pragma unroll
    for (int i = 0; i < NW; i++) qs[b * NW + i] = (unsigned)p[F::qofs / 2 + 2 * i] | ((unsigned)p[F::qofs / 2 + 2 * i + 1] << 16);
}

}  // namespace

bool quant_repack(const void *blocks, int ttype, long n_blocks, void *qs, unsigned *dm, unsigned *qh, hipStream_t st) {
    if (n_blocks <= 0) return true;
    if (!quant_block_bytes(ttype) || ((uintptr_t)blocks & 1) || ((uintptr_t)qs & 3) || !dm || ((ttype == 6 || ttype == 7) && !qh)) return false;
    const long grid = (n_blocks + 255) / 256;
    if (grid > 0x7fffffffL) return false;
    const unsigned short *s = (const unsigned short *)blocks;
    unsigned *q = (unsigned *)qs;
    switch (ttype) {
        case 2: repack_blocks_kernel<2><<<dim3((unsigned)grid), 256, 0, st>>>(s, n_blocks, q, dm, qh); break;
        case 3: repack_blocks_kernel<3><<<dim3((unsigned)grid), 256, 0, st>>>(s, n_blocks, q, dm, qh); break;
        case 6: repack_blocks_kernel<6><<<dim3((unsigned)grid), 256, 0, st>>>(s, n_blocks, q, dm, qh); break;
        case 7: repack_blocks_kernel<7><<<dim3((unsigned)grid), 256, 0, st>>>(s, n_blocks, q, dm, qh); break;
        case 8: repack_blocks_kernel<8><<<dim3((unsigned)grid), 256, 0, st>>>(s, n_blocks, q, dm, qh); break;
    }
    return hipGetLastError() == hipSuccess;
}

int quant_block_bytes(int ttype) {
    switch (ttype) { case 2: return 18; case 3: return 20; case 6: return 22; case 7: return 24; case 8: return 34; }
    return 0;
}

bool expand_blocks_f16(const void *blocks, int ttype, long n_blocks, f16 *dst, hipStream_t st) {
    if (n_blocks <= 0) return true;
    if (!quant_block_bytes(ttype) || ((uintptr_t)blocks & 1) || ((uintptr_t)dst & 15)) return false;
    const long grid = (n_blocks + 255) / 256;
    if (grid > 0x7fffffffL) return false;
    const unsigned short *s = (const unsigned short *)blocks;
    switch (ttype) {
        case 2: expand_blocks_kernel<2><<<dim3((unsigned)grid), 256, 0, st>>>(s, n_blocks, dst); break;
        case 3: expand_blocks_kernel<3><<<dim3((unsigned)grid), 256, 0, st>>>(s, n_blocks, dst); break;
        case 6: expand_blocks_kernel<6><<<dim3((unsigned)grid), 256, 0, st>>>(s, n_blocks, dst); break;
        case 7: expand_blocks_kernel<7><<<dim3((unsigned)grid), 256, 0, st>>>(s, n_blocks, dst); break;
        case 8: expand_blocks_kernel<8><<<dim3((unsigned)grid), 256, 0, st>>>(s, n_blocks, dst); break;
    }
    return hipGetLastError() == hipSuccess;
}

}  // namespace wmi
