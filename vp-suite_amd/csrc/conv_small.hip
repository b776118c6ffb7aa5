// conv_small.hip — the EF glue's few-channel stride-1 layers as streaming kernels (ef_blocks.py:15-49: conv1 1|3 -> 16 3x3 +
// LeakyReLU in front of the encoder, the final 16 -> 1|3 1x1 convolution behind the forecaster, and the latter's adjoint).
// No matrix shape worth an MFMA tile: 9-27 or 16 products per output, HBM-bound by the 16-channel side. The implicit-GEMM
// kernel ran them with 64-wide tiles for 1, 3 or 16 useful columns at 2-4x their HBM time. fp32 FMA
// throughout (exact operands: these layers need no bf16 split).
#include <stdlib.h>

#include "vpx_internal.h"

namespace vpx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned short cs_bf16_bits(float v) {
    __bf16 h = (__bf16)v;
    return __builtin_bit_cast(unsigned short, h);
}

// few -> many: y[p][co] = act(bias[co] + sum_{tap, ci} x[p + tap][ci] * w(co, ci, tap)); one thread per (pixel, 4 output
// channels): a wave's stores are contiguous. Optional second output in split-bf16 operand format (cell2.hip) for the ConvLSTM
// block that consumes it. TR (k = 1 only): w is the [CI][CO] tensor of the transposed layer (the adjoint of many -> few).
template <int CI, int K>
__global__ __launch_bounds__(256) void conv_few_to_many_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ y,
                                                               char* __restrict__ y_sp, long long npix, int H, int W, int CO, int pad,
                                                               float leaky, int tr) {
    constexpr int KK = K * K;
    __shared__ float wl[KK * CI * 64];   // [tap][ci][co], CO <= 64
    for (int e = threadIdx.x; e < KK * CI * CO; e += 256) {
        const int co = e % CO, r = e / CO, ci = r % CI, tap = r / CI;
        wl[e] = tr ? w[(ci * CO + co) * KK + tap] : w[(co * CI + ci) * KK + tap];
    }
    __syncthreads();
    const int q = CO >> 2;   // channel quads per pixel
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long p = gid / q;
    if (p >= npix) return;
    const int cg = (int)(gid - p * q);
    const int xx = (int)(p % W);
    const long long r = p / W;
    const int yy = (int)(r % H);
    const long long row0 = (r / H) * H;
    f32x4 acc = bias ? *reinterpret_cast<const f32x4*>(bias + cg * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const int iy = yy + ky - pad, ix = xx + kx - pad;
            const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const float* src = x + ((row0 + iy) * W + ix) * CI;
#pragma unroll
            for (int ci = 0; ci < CI; ++ci) {
                const float v = in ? src[ci] : 0.f;
                acc += v * *reinterpret_cast<const f32x4*>(wl + ((ky * K + kx) * CI + ci) * CO + cg * 4);
            }
        }
    if (leaky != 0.0f) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = acc[e] > 0.f ? acc[e] : acc[e] * leaky;
    }
    if (y) *reinterpret_cast<f32x4*>(y + p * CO + cg * 4) = acc;
    if (y_sp) {
        unsigned h[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned short hb = cs_bf16_bits(acc[e]);
            h[e] = hb;
            l[e] = cs_bf16_bits(acc[e] - __builtin_bit_cast(float, (unsigned)hb << 16));
        }
        char* dst = y_sp + p * CO * 4 + (cg >> 1) * 32 + (cg & 1) * 8;   // [pixel][group of 8][8 hi | 8 lo]
        *reinterpret_cast<uint2*>(dst) = uint2{h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
        *reinterpret_cast<uint2*>(dst + 16) = uint2{l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
    }
}

// few -> 16 channels (the first layer of both EF encoders): one thread per PIXEL and all 16 channels — a quarter of the input loads
// and address arithmetic of the kernel above, every weight read an LDS broadcast, 16-byte stores (64 contiguous bytes per lane).
// 1 -> 16 3x3 at 64x64, 1280 frames, operand-format output: 0.143 -> see DESIGN.md 3.5.
template <int CI, int K>
__global__ __launch_bounds__(256) void conv_few_to_16_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ y,
                                                             char* __restrict__ y_sp, long long npix, int H, int W, int pad, float leaky, int tr) {
    constexpr int KK = K * K, CO = 16;
    __shared__ __attribute__((aligned(16))) float wl[KK * CI * CO];   // [tap][ci][co]
    for (int e = threadIdx.x; e < KK * CI * CO; e += 256) {
        const int co = e % CO, r = e / CO, ci = r % CI, tap = r / CI;
        wl[e] = tr ? w[(ci * CO + co) * KK + tap] : w[(co * CI + ci) * KK + tap];
    }
    __syncthreads();
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    const int xx = (int)(p % W);
    const long long r = p / W;
    const int yy = (int)(r % H);
    const long long row0 = (r / H) * H;
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = bias ? *reinterpret_cast<const f32x4*>(bias + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const int iy = yy + ky - pad, ix = xx + kx - pad;
            const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const float* src = x + ((row0 + iy) * W + ix) * CI;
#pragma unroll
            for (int ci = 0; ci < CI; ++ci) {
                const float v = in ? src[ci] : 0.f;
                const float* wr = wl + ((ky * K + kx) * CI + ci) * CO;
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] += v * *reinterpret_cast<const f32x4*>(wr + q * 4);
            }
        }
    if (leaky != 0.0f) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[q][e] = acc[q][e] > 0.f ? acc[q][e] : acc[q][e] * leaky;
    }
    if (y) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(y + p * CO + q * 4) = acc[q];
    }
    if (y_sp) {   // [pixel][group of 8 channels][8 hi | 8 lo]
        unsigned h[16], l[16];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned short hb = cs_bf16_bits(acc[q][e]);
                h[q * 4 + e] = hb;
                l[q * 4 + e] = cs_bf16_bits(acc[q][e] - __builtin_bit_cast(float, (unsigned)hb << 16));
            }
        char* dst = y_sp + p * CO * 4;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            *reinterpret_cast<uint4*>(dst + g * 32) = uint4{h[g * 8] | (h[g * 8 + 1] << 16), h[g * 8 + 2] | (h[g * 8 + 3] << 16),
                                                            h[g * 8 + 4] | (h[g * 8 + 5] << 16), h[g * 8 + 6] | (h[g * 8 + 7] << 16)};
            *reinterpret_cast<uint4*>(dst + g * 32 + 16) = uint4{l[g * 8] | (l[g * 8 + 1] << 16), l[g * 8 + 2] | (l[g * 8 + 3] << 16),
                                                                 l[g * 8 + 4] | (l[g * 8 + 5] << 16), l[g * 8 + 6] | (l[g * 8 + 7] << 16)};
        }
    }
}

// many -> few (1x1): y[p][co] = act(bias[co] + sum_ci x[p][ci] * w[co][ci]), CO = 1 or 3; one thread per pixel, 16-byte loads
template <int CO>
__global__ __launch_bounds__(256) void conv_many_to_few_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ y, long long npix,
                                                               int CI, float leaky) {
    __shared__ float wl[CO * 64];
    for (int e = threadIdx.x; e < CO * CI; e += 256) wl[e] = w[e];
    __syncthreads();
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    float acc[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) acc[co] = bias ? bias[co] : 0.f;
    const float* src = x + p * CI;
    for (int c = 0; c < CI; c += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + co * CI + c);
            acc[co] += v[0] * wv[0] + v[1] * wv[1] + v[2] * wv[2] + v[3] * wv[3];
        }
    }
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        float v = acc[co];
        if (leaky != 0.0f) v = v > 0.f ? v : v * leaky;
        y[p * CO + co] = v;
    }
}

// 0: not a few-channel layer; 1: few -> many 3x3; 2: many -> few 1x1; 3: the 1x1 transposed layer (adjoint of 2)
int conv_small_kind(const vpx_conv_desc* d) {
    static int env = -1;   // VPX_CONV_SMALL=0: implicit-GEMM kernel for these layers too (experiments)
    if (env < 0) env = dev_switch("VPX_CONV_SMALL", 1);
    if (!env || d->stride != 1 || d->kh != d->kw) return 0;
    if (!d->transposed && d->kh == 3 && d->pad == 1 && (d->Ci == 1 || d->Ci == 3) && (d->Co & 7) == 0 && d->Co <= 64) return 1;
    if (!d->transposed && d->kh == 1 && d->pad == 0 && (d->Ci & 3) == 0 && d->Ci <= 64 && (d->Co == 1 || d->Co == 3)) return 2;
    if (d->transposed && d->kh == 1 && d->pad == 0 && (d->Ci == 1 || d->Ci == 3) && (d->Co & 7) == 0 && d->Co <= 64) return 3;
    return 0;
}

hipError_t launch_conv_small(const vpx_conv_desc* d, int kind, const float* x, const float* w, const float* bias, float* y, char* y_sp,
                             hipStream_t s) {
    const long long npix = (long long)d->N * d->H * d->W;
    if (kind == 2) {
        const dim3 grid((unsigned)((npix + 255) / 256));
        if (d->Co == 1) VPX_LAUNCH(conv_many_to_few_kernel<1>, grid, dim3(256), 0, s, x, w, bias, y, npix, d->Ci, d->leaky_slope);
        else VPX_LAUNCH(conv_many_to_few_kernel<3>, grid, dim3(256), 0, s, x, w, bias, y, npix, d->Ci, d->leaky_slope);
        return vpx_hip_last_error();
    }
    const int tr = kind == 3;
    if (d->Co == 16) {   // one thread per pixel
        const dim3 g16((unsigned)((npix + 255) / 256));
        if (kind == 1 && d->Ci == 1) VPX_LAUNCH((conv_few_to_16_kernel<1, 3>), g16, dim3(256), 0, s, x, w, bias, y, y_sp, npix, d->H, d->W, 1, d->leaky_slope, 0);
        else if (kind == 1) VPX_LAUNCH((conv_few_to_16_kernel<3, 3>), g16, dim3(256), 0, s, x, w, bias, y, y_sp, npix, d->H, d->W, 1, d->leaky_slope, 0);
        else if (d->Ci == 1) VPX_LAUNCH((conv_few_to_16_kernel<1, 1>), g16, dim3(256), 0, s, x, w, bias, y, y_sp, npix, d->H, d->W, 0, d->leaky_slope, tr);
        else VPX_LAUNCH((conv_few_to_16_kernel<3, 1>), g16, dim3(256), 0, s, x, w, bias, y, y_sp, npix, d->H, d->W, 0, d->leaky_slope, tr);
        return vpx_hip_last_error();
    }
    const long long threads = npix * (d->Co / 4);
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (kind == 1 && d->Ci == 1) VPX_LAUNCH((conv_few_to_many_kernel<1, 3>), grid, dim3(256), 0, s, x, w, bias, y, y_sp, npix, d->H, d->W, d->Co, 1, d->leaky_slope, 0);
    else if (kind == 1) VPX_LAUNCH((conv_few_to_many_kernel<3, 3>), grid, dim3(256), 0, s, x, w, bias, y, y_sp, npix, d->H, d->W, d->Co, 1, d->leaky_slope, 0);
    else if (d->Ci == 1) VPX_LAUNCH((conv_few_to_many_kernel<1, 1>), grid, dim3(256), 0, s, x, w, bias, y, y_sp, npix, d->H, d->W, d->Co, 0, d->leaky_slope, tr);
    else VPX_LAUNCH((conv_few_to_many_kernel<3, 1>), grid, dim3(256), 0, s, x, w, bias, y, y_sp, npix, d->H, d->W, d->Co, 0, d->leaky_slope, tr);
    return vpx_hip_last_error();
}

}  // namespace vpx
