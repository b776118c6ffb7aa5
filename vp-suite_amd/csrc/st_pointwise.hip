// st_pointwise.hip — the pointwise stages that follow K-SPLIT c5 launches of the ST-LSTM step (round 4). On small grids (fewer than
// 96 pixel tiles of 16x16: BASELINE configs[4]'s per-GPU shards, small-batch predrnn-pp) a c5 workgroup cannot run the whole K of a
// tile — the chip would stand empty — so the K of every convolution is cut into chunks that run as separate JOBS of the same launch
// and write fp32 partial sums into their own buffers (no atomics: bit-reproducible). These kernels add the partials in a fixed order
// and apply what the fused epilogues apply on large grids:
//   st_gates_ks_kernel   pre-activations [B,HW,7Ch] (i,f,g,o | i',f',g') -> c_new, m_new, delta_c, delta_m, o_pre, saved gates, split copies
//   st_out_ks_kernel     h_new = sigmoid(o_pre + conv_o(mem)) * tanh(conv_last(mem))  (predrnn.py:80-81)
//   sum_partials_kernel  out (+)= sum of partial buffers (data gradients)
#include "cell2_dev.h"
#include "vpx_host.h"

namespace vpx {

struct F8v { f32x4 a, b; };
__device__ __forceinline__ F8v ld8v(const float* p) { return F8v{*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4)}; }
__device__ __forceinline__ void st8v(float* p, const float (&v)[8]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
__device__ __forceinline__ void st8split(char* sp, size_t pix, int Ch, int c, const float (&v)[8]) {   // c % 8 == 0
    unsigned h[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c2_split(v[i], h[i], l[i]);
    uint4* d = reinterpret_cast<uint4*>(sp + pix * ((size_t)Ch * 4u) + (size_t)c * 4u);
    d[0] = uint4{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
    d[1] = uint4{l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
}

// sum over the ks partial buffers of the 8 values at element offset e (buffers `stride` floats apart)
__device__ __forceinline__ void sum8(const float* p0, long long stride, int ks, size_t e, float (&v)[8]) {
    F8v s = ld8v(p0 + e);
    for (int k = 1; k < ks; ++k) { const F8v t = ld8v(p0 + (size_t)k * stride + e); s.a += t.a; s.b += t.b; }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = i < 4 ? s.a[i] : s.b[i - 4];
}

__global__ __launch_bounds__(256) void st_gates_ks_kernel(const STGatesKSArgs a) {
    const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const int Ch = a.Ch, G = Ch >> 3;
    if (t >= a.npix * G) return;
    const long long pix = t / G;
    const int ch = (int)(t - pix * G) * 8;
    const size_t e = (size_t)pix * Ch + ch, pe = (size_t)pix * 7 * Ch + ch;
    float pi[8], pf[8], pg[8], po[8];
    {   // c group
        sum8(a.part, a.pstride, a.ks, pe, pi); sum8(a.part, a.pstride, a.ks, pe + Ch, pf);
        sum8(a.part, a.pstride, a.ks, pe + 2 * Ch, pg); sum8(a.part, a.pstride, a.ks, pe + 3 * Ch, po);
        const F8v cin = ld8v(a.c + e);
        float gi[8], gf[8], gg[8], dl[8], sn[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            gi[i] = sigmoid_f(pi[i]); gf[i] = sigmoid_f(pf[i] + a.fbias); gg[i] = tanh_f(pg[i]);
            dl[i] = gi[i] * gg[i];
            sn[i] = gf[i] * (i < 4 ? cin.a[i] : cin.b[i - 4]) + dl[i];
        }
        st8v(a.c_new + e, sn); st8v(a.delta_c + e, dl); st8v(a.o_pre + e, po);
        if (a.gates_c) { float* gs = a.gates_c + (size_t)pix * 3 * Ch + ch; st8v(gs, gi); st8v(gs + Ch, gf); st8v(gs + 2 * Ch, gg); }
        if (a.cn_sp) st8split(a.cn_sp, (size_t)pix, Ch, ch, sn);
    }
    {   // m group
        sum8(a.part, a.pstride, a.ks, pe + 4 * Ch, pi); sum8(a.part, a.pstride, a.ks, pe + 5 * Ch, pf); sum8(a.part, a.pstride, a.ks, pe + 6 * Ch, pg);
        const F8v min_ = ld8v(a.m + e);
        float gi[8], gf[8], gg[8], dl[8], sn[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            gi[i] = sigmoid_f(pi[i]); gf[i] = sigmoid_f(pf[i] + a.fbias); gg[i] = tanh_f(pg[i]);
            dl[i] = gi[i] * gg[i];
            sn[i] = gf[i] * (i < 4 ? min_.a[i] : min_.b[i - 4]) + dl[i];
        }
        st8v(a.m_new + e, sn); st8v(a.delta_m + e, dl);
        if (a.gates_m) { float* gs = a.gates_m + (size_t)pix * 3 * Ch + ch; st8v(gs, gi); st8v(gs + Ch, gf); st8v(gs + 2 * Ch, gg); }
        if (a.mn_sp) st8split(a.mn_sp, (size_t)pix, Ch, ch, sn);
    }
}

__global__ __launch_bounds__(256) void st_out_ks_kernel(const STOutKSArgs a) {
    const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (t >= a.n / 8) return;
    const size_t e = (size_t)t * 8;
    float co[8];
    sum8(a.part, a.pstride, a.ks, e, co);
    const F8v op = ld8v(a.o_pre + e), lc = ld8v(a.lc + e);
    float o[8], tl[8], hn[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        o[i] = sigmoid_f((i < 4 ? op.a[i] : op.b[i - 4]) + co[i]);
        tl[i] = tanh_f(i < 4 ? lc.a[i] : lc.b[i - 4]);
        hn[i] = o[i] * tl[i];
    }
    st8v(a.h_new + e, hn);
    if (a.o_save) { st8v(a.o_save + e, o); st8v(a.tl_save + e, tl); }
    if (a.h_sp) st8split(a.h_sp, e / a.Ch, a.Ch, (int)(e % a.Ch), hn);
}

__global__ __launch_bounds__(256) void sum_partials_kernel(float* __restrict__ out, const float* __restrict__ part, long long pstride, int ks,
                                                          long long n, int accumulate) {
    const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (t >= n / 4) return;
    const size_t e = (size_t)t * 4;
    f32x4 s = accumulate ? *reinterpret_cast<const f32x4*>(out + e) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < ks; ++k) s += *reinterpret_cast<const f32x4*>(part + (size_t)k * pstride + e);
    *reinterpret_cast<f32x4*>(out + e) = s;
}

hipError_t launch_st_gates_ks(const STGatesKSArgs& a, hipStream_t s) {
    const long long n = a.npix * (a.Ch >> 3);
    VPX_LAUNCH(st_gates_ks_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    return vpx_hip_last_error();
}
hipError_t launch_st_out_ks(const STOutKSArgs& a, hipStream_t s) {
    const long long n = a.n / 8;
    VPX_LAUNCH(st_out_ks_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    return vpx_hip_last_error();
}
// several sums in one launch (blockIdx.y = job): the K-split data gradients of an ST-LSTM step's backward leave up to five of them behind
struct SumJobs { int n, _p; float* out[5]; const float* part[5]; long long cnt[5]; int ks[5], acc[5]; };
__global__ __launch_bounds__(256) void sum_partials_multi_kernel(const SumJobs J) {
    const int j = blockIdx.y;
    const long long n = J.cnt[j];
    const float* const part = J.part[j];
    float* const out = J.out[j];
    const int ks = J.ks[j];
    for (long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; e < n; e += (long long)gridDim.x * 1024) {
        f32x4 s = J.acc[j] ? *reinterpret_cast<const f32x4*>(out + e) : f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < ks; ++k) s += *reinterpret_cast<const f32x4*>(part + (size_t)k * n + e);
        *reinterpret_cast<f32x4*>(out + e) = s;
    }
}
hipError_t launch_sum_partials_multi(int njobs, float* const* out, const float* const* part, const long long* n, const int* ks, const int* acc, hipStream_t s) {
    if (njobs < 1) return hipSuccess;
    if (njobs > 5) return hipErrorInvalidValue;
    SumJobs J{};
    J.n = njobs;
    long long mx = 0;
    for (int i = 0; i < njobs; ++i) {
        if (n[i] & 3) return hipErrorInvalidValue;
        J.out[i] = out[i]; J.part[i] = part[i]; J.cnt[i] = n[i]; J.ks[i] = ks[i]; J.acc[i] = acc[i];
        if (n[i] > mx) mx = n[i];
    }
    VPX_LAUNCH(sum_partials_multi_kernel, dim3((unsigned)((mx / 4 + 255) / 256), njobs), dim3(256), 0, s, J);
    return vpx_hip_last_error();
}
hipError_t launch_sum_partials(float* out, const float* part, long long pstride, int ks, long long n, int accumulate, hipStream_t s) {
    VPX_LAUNCH(sum_partials_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, out, part, pstride, ks, n, accumulate);
    return vpx_hip_last_error();
}

}  // namespace vpx
