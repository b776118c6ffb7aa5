// pointwise.hip — HBM-bound helpers of the hot path: layout adaptors (the reference is NCHW, the kernels are NHWC).
#include "vpx_internal.h"

namespace vpx {

// [N][R][C] -> [N][C][R] through a 32x33 LDS tile (coalesced on both sides).
__global__ void transpose_rc_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int C) {
    __shared__ float tile[32][33];
    const size_t img = (size_t)blockIdx.z * R * C;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int r = r0 + i, c = c0 + threadIdx.x;
        if (r < R && c < C) tile[i][threadIdx.x] = src[img + (size_t)r * C + c];
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int c = c0 + i, r = r0 + threadIdx.x;
        if (r < R && c < C) dst[img + (size_t)c * R + r] = tile[threadIdx.x][i];
    }
}

static hipError_t launch_transpose(const float* src, float* dst, int N, int R, int C, hipStream_t s) {
    // gridDim.z is limited to 65535: fold large N into several launches
    for (int n0 = 0; n0 < N; n0 += 65535) {
        const int nn = (N - n0 < 65535) ? (N - n0) : 65535;
        dim3 grid((C + 31) / 32, (R + 31) / 32, nn);
        VPX_LAUNCH(transpose_rc_kernel, grid, dim3(32, 8), 0, s, src + (size_t)n0 * R * C,
                           dst + (size_t)n0 * R * C, R, C);
    }
    return vpx_hip_last_error();
}

hipError_t launch_nchw_to_nhwc(const float* src, float* dst, int N, int C, int H, int W, hipStream_t s) {
    return launch_transpose(src, dst, N, C, H * W, s);  // [N][C][HW] -> [N][HW][C]
}
hipError_t launch_nhwc_to_nchw(const float* src, float* dst, int N, int C, int H, int W, hipStream_t s) {
    return launch_transpose(src, dst, N, H * W, C, s);  // [N][HW][C] -> [N][C][HW]
}

// ---------------------------------------------------------------------------------------------------------------
// decoupling-loss tail. F.normalize(dim=2, eps=1e-12) then cosine_similarity(dim=2, eps=1e-8), abs, mean.
// One thread per (b, co) walks the H*W axis (channels are contiguous -> coalesced across the wave).
// ---------------------------------------------------------------------------------------------------------------
// block = 32 channels x 32 pixel slices: every thread sums its slice of the H*W axis with four loads of each operand in flight, LDS
// combines the slices in a fixed order. (Round 3 had 8 slices and a serial walk: 40 us for a 32x32 map at B = 4 — 16 blocks on the
// whole chip, one load in flight per thread.)
__global__ __launch_bounds__(1024) void decouple_stats_kernel(const float* __restrict__ yc, const float* __restrict__ ym,
                                                              float* __restrict__ stats, int B, int HW, int Ch) {
    constexpr int NS = 32;
    __shared__ float red[3][NS][32];
    const int b = blockIdx.x, lane32 = threadIdx.x & 31, co = blockIdx.y * 32 + lane32, slice = threadIdx.x >> 5;
    float scc = 0.f, smm = 0.f, scm = 0.f;
    if (co < Ch) {
        const float* pc = yc + (size_t)b * HW * Ch + co;
        const float* pm = ym + (size_t)b * HW * Ch + co;
        int p = slice;
        for (; p + 3 * NS < HW; p += 4 * NS) {
            float a[4], m[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { a[k] = pc[(size_t)(p + k * NS) * Ch]; m[k] = pm[(size_t)(p + k * NS) * Ch]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) { scc += a[k] * a[k]; smm += m[k] * m[k]; scm += a[k] * m[k]; }
        }
        for (; p < HW; p += NS) {
            const float a = pc[(size_t)p * Ch], m = pm[(size_t)p * Ch];
            scc += a * a; smm += m * m; scm += a * m;
        }
    }
    red[0][slice][lane32] = scc;
    red[1][slice][lane32] = smm;
    red[2][slice][lane32] = scm;
    __syncthreads();
    if (slice == 0 && co < Ch) {
        scc = smm = scm = 0.f;
        for (int s = 0; s < NS; ++s) { scc += red[0][s][lane32]; smm += red[1][s][lane32]; scm += red[2][s][lane32]; }
        const float nc = fmaxf(sqrtf(scc), 1e-12f), nm = fmaxf(sqrtf(smm), 1e-12f);
        // after normalisation |u| = sqrt(scc)/nc, |v| = sqrt(smm)/nm (1 unless degenerate); cos = u.v / max(|u||v|, 1e-8)
        const float un = sqrtf(scc) / nc, vn = sqrtf(smm) / nm;
        const float c = (scm / (nc * nm)) / fmaxf(un * vn, 1e-8f);
        float* st = stats + ((size_t)b * Ch + co) * 4;
        st[0] = scc; st[1] = smm; st[2] = scm; st[3] = fabsf(c);
    }
}

__global__ void decouple_mean_kernel(const float* __restrict__ stats, float* __restrict__ value, int n) {
    __shared__ float red[256];
    // eight loads in flight per thread (the serial walk kept one: 17.6 us for 16 384 values, most of it latency); fixed order
    float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int i = threadIdx.x;
    for (; i + 7 * 256 < n; i += 8 * 256) {
#pragma unroll
        for (int k = 0; k < 8; ++k) p[k] += stats[(size_t)(i + k * 256) * 4 + 3];
    }
    for (int k = 0; i < n; i += 256, ++k) p[k] += stats[(size_t)i * 4 + 3];
    const float acc = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *value = red[0] / (float)n;
}

// d|cos|/dY for the non-degenerate case (norms above the eps clamps; the clamps have zero gradient otherwise)
__global__ void decouple_bwd_pointwise_kernel(const float* __restrict__ yc, const float* __restrict__ ym,
                                              const float* __restrict__ stats, const float* __restrict__ dvalue,
                                              float* __restrict__ dyc, float* __restrict__ dym, int B, int HW, int Ch) {
    const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const long long n = (long long)B * HW * Ch;
    if (e >= n) return;
    const int co = (int)(e % Ch);
    const int b = (int)(e / ((long long)HW * Ch));
    const float* st = stats + ((size_t)b * Ch + co) * 4;
    const float scc = st[0], smm = st[1], scm = st[2];
    const float nc = sqrtf(scc), nm = sqrtf(smm);
    float gc = 0.f, gm = 0.f;
    if (nc > 1e-12f && nm > 1e-12f) {
        const float c = scm / (nc * nm);
        const float g = (c > 0.f ? 1.f : (c < 0.f ? -1.f : 0.f)) * dvalue[0] / (float)(B * Ch);
        const float a = yc[e], m = ym[e];
        gc = g * (m / (nc * nm) - c * a / scc);
        gm = g * (a / (nc * nm) - c * m / smm);
    }
    dyc[e] = gc;
    dym[e] = gm;
}

__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, long long n) {
    const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (e < n) y[e] += x[e];
}

hipError_t launch_decouple_stats(const float* yc, const float* ym, float* stats, int B, int HW, int Ch, hipStream_t s) {
    VPX_LAUNCH(decouple_stats_kernel, dim3(B, (Ch + 31) / 32), dim3(1024), 0, s, yc, ym, stats, B, HW, Ch);
    return vpx_hip_last_error();
}
hipError_t launch_decouple_mean(const float* stats, float* value, int n, hipStream_t s) {
    VPX_LAUNCH(decouple_mean_kernel, dim3(1), dim3(256), 0, s, stats, value, n);
    return vpx_hip_last_error();
}
hipError_t launch_decouple_bwd_pointwise(const float* yc, const float* ym, const float* stats, const float* dvalue,
                                         float* dyc, float* dym, int B, int HW, int Ch, hipStream_t s) {
    const long long n = (long long)B * HW * Ch;
    VPX_LAUNCH(decouple_bwd_pointwise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, yc, ym, stats,
                       dvalue, dyc, dym, B, HW, Ch);
    return vpx_hip_last_error();
}
hipError_t launch_axpy(float* y, const float* x, long long n, hipStream_t s) {
    VPX_LAUNCH(axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, y, x, n);
    return vpx_hip_last_error();
}

// ---- ConvLSTM gate / state update on pre-activations (K-split step on small maps) ---------------------------------
// Same arithmetic as EpiConvLSTM (conv_gemm.hip): peepholes on the previous cell for i,f and on the NEW cell for o
// (conv_lstm_hzzone.py:62-68). One thread per (pixel, channel); `pre` may alias ea.gates (a thread reads its four
// values before it writes them).
__global__ __launch_bounds__(256) void convlstm_pointwise_kernel(const ConvLSTMStepArgs a, const float* pre, long long pre_bstride, int B,
                                                                 long long HW) {
    const long long e = blockIdx.x * 256LL + threadIdx.x;
    const int Ch = a.Ch;
    if (e >= (long long)B * HW * Ch) return;
    const int ch = (int)(e % Ch);
    const long long bp = e / Ch;          // b * HW + pix
    const long long pix = bp % HW;
    const int b = (int)(bp / HW);
    const float* p4 = pre + (size_t)b * pre_bstride + pix * 4 * Ch + ch;   // pre_bstride: elements between batch items
    float ai = p4[a.gate_pos[0] * Ch], af = p4[a.gate_pos[1] * Ch], ag = p4[a.gate_pos[2] * Ch], ao = p4[a.gate_pos[3] * Ch];
    if (a.bias) {
        ai += a.bias[a.gate_pos[0] * Ch + ch]; af += a.bias[a.gate_pos[1] * Ch + ch];
        ag += a.bias[a.gate_pos[2] * Ch + ch]; ao += a.bias[a.gate_pos[3] * Ch + ch];
    }
    const float cp = a.c_in ? a.c_in[e] : 0.0f;
    if (a.wci) { ai += a.wci[pix * Ch + ch] * cp; af += a.wcf[pix * Ch + ch] * cp; }
    const float i_ = sigmoid_f(ai), f_ = sigmoid_f(af), g_ = tanh_f(ag);
    const float cn = lstm_c(f_, cp, i_, g_);
    if (a.wco) ao += a.wco[pix * Ch + ch] * cn;
    const float o_ = sigmoid_f(ao);
    a.c_out[e] = cn;
    a.h_out[(size_t)b * a.h_bstride + pix * Ch + ch] = o_ * tanh_f(cn);
    if (a.gates) {
        float* gs = a.gates + bp * 4 * Ch + ch;
        gs[0] = i_; gs[Ch] = f_; gs[2 * Ch] = g_; gs[3 * Ch] = o_;
    }
}

hipError_t launch_convlstm_pointwise(const ConvLSTMStepArgs& ea, const float* pre, int B, long long HW, hipStream_t s,
                                     long long pre_bstride) {
    const long long n = (long long)B * HW * ea.Ch;
    if (pre_bstride <= 0) pre_bstride = HW * 4 * ea.Ch;
    VPX_LAUNCH(convlstm_pointwise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ea, pre, pre_bstride, B, HW);
    return vpx_hip_last_error();
}

}  // namespace vpx
