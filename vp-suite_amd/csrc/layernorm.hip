// layernorm.hip — LayerNorm([C,H,W]) per sample (predrnn.py:27,31,35,39) forward/backward and the pointwise gate stages
// of the LayerNorm ST-LSTM variant, where the three input convolutions cannot be fused into one accumulator because each
// is normalised on its own before the gates add them.
// All tensors NHWC: a sample is one contiguous run of n = HW*C floats; gamma/beta are [HW, C] (transposed once per call
// from the reference's [C,H,W]).
#include "vpx_host.h"

namespace vpx {

constexpr int LN_CHUNKS = 64;

// mode 0: partial[b][chunk] = sum(x) ; mode 1: sum((x - mean[b])^2)
__global__ void ln_partial_kernel(const float* __restrict__ x, long long n, const float* __restrict__ stats, int mode,
                                  double* __restrict__ partial) {
    __shared__ double red[256];
    const int b = blockIdx.x, chunk = blockIdx.y;
    const long long per = (n + LN_CHUNKS - 1) / LN_CHUNKS;
    const long long lo = chunk * per, hi = (lo + per < n) ? lo + per : n;
    const float* p = x + (size_t)b * n;
    const float m = mode ? stats[2 * b] : 0.0f;
    double acc = 0.0;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        const float v = p[i] - m;
        acc += mode ? (double)v * v : (double)v;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(size_t)b * LN_CHUNKS + chunk] = red[0];
}

// stats[b] = (mean, rstd)
__global__ void ln_finalize_kernel(const double* __restrict__ partial, long long n, int mode, float* __restrict__ stats, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double s = 0.0;
    for (int c = 0; c < LN_CHUNKS; ++c) s += partial[(size_t)b * LN_CHUNKS + c];
    if (mode == 0) stats[2 * b] = (float)(s / (double)n);
    else stats[2 * b + 1] = (float)(1.0 / sqrt(s / (double)n + 1e-5));
}

// y = xhat * gamma + beta, xhat = (x - mean) * rstd ; xhat saved for the backward (may be null)
__global__ void ln_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ xhat,
                                long long n, int B) {
    const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (e >= n * B) return;
    const int b = (int)(e / n);
    const long long i = e - (long long)b * n;
    const float xh = (x[e] - stats[2 * b]) * stats[2 * b + 1];
    if (xhat) xhat[e] = xh;
    y[e] = xh * gamma[i] + beta[i];
}

hipError_t launch_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* xhat,
                                float* stats, double* partial, int B, long long n, hipStream_t s) {
    VPX_LAUNCH(ln_partial_kernel, dim3(B, LN_CHUNKS), dim3(256), 0, s, x, n, stats, 0, partial);
    VPX_LAUNCH(ln_finalize_kernel, dim3((B + 63) / 64), dim3(64), 0, s, partial, n, 0, stats, B);
    VPX_LAUNCH(ln_partial_kernel, dim3(B, LN_CHUNKS), dim3(256), 0, s, x, n, stats, 1, partial);
    VPX_LAUNCH(ln_finalize_kernel, dim3((B + 63) / 64), dim3(64), 0, s, partial, n, 1, stats, B);
    const long long tot = n * B;
    VPX_LAUNCH(ln_apply_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, x, stats, gamma, beta, y,
                       xhat, n, B);
    return vpx_hip_last_error();
}

// ---- backward -------------------------------------------------------------------------------------------------------
// dy is read through a channel-block map: channel c of the normalised tensor is channel blk[c / Cb] * Cb + c % Cb of a
// [.., ldy]-wide gradient tensor (the ST-LSTM keeps all gate gradients in one [B,HW,7Ch] tensor).
struct LNBwdArgs {
    const float* dy; int ldy; int Cb; int blk[8];
    const float* xhat; const float* stats; const float* gamma;
    int B, HW, C;
    double* partial;   // [B][LN_CHUNKS][2]
    float* sums;       // [B][2] : mean(dxhat), mean(dxhat * xhat)
    float* du;         // [B,HW,C]
    float* dgamma; float* dbeta;   // [HW,C], overwritten
};

__device__ __forceinline__ float ln_dy(const LNBwdArgs& a, int b, int p, int c) {
    const int src = a.Cb ? a.blk[c / a.Cb] * a.Cb + c % a.Cb : c;
    return a.dy[((size_t)b * a.HW + p) * a.ldy + src];
}

__global__ void ln_bwd_partial_kernel(const LNBwdArgs a) {
    __shared__ double r1[256], r2[256];
    const int b = blockIdx.x, chunk = blockIdx.y;
    const long long n = (long long)a.HW * a.C;
    const long long per = (n + LN_CHUNKS - 1) / LN_CHUNKS;
    const long long lo = chunk * per, hi = (lo + per < n) ? lo + per : n;
    double s1 = 0.0, s2 = 0.0;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        const int p = (int)(i / a.C), c = (int)(i - (long long)p * a.C);
        const float dxh = ln_dy(a, b, p, c) * a.gamma[i];
        s1 += dxh;
        s2 += (double)dxh * a.xhat[(size_t)b * n + i];
    }
    r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { r1[threadIdx.x] += r1[threadIdx.x + s]; r2[threadIdx.x] += r2[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a.partial[((size_t)b * LN_CHUNKS + chunk) * 2] = r1[0];
        a.partial[((size_t)b * LN_CHUNKS + chunk) * 2 + 1] = r2[0];
    }
}

__global__ void ln_bwd_finalize_kernel(const LNBwdArgs a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    double s1 = 0.0, s2 = 0.0;
    for (int c = 0; c < LN_CHUNKS; ++c) {
        s1 += a.partial[((size_t)b * LN_CHUNKS + c) * 2];
        s2 += a.partial[((size_t)b * LN_CHUNKS + c) * 2 + 1];
    }
    const double n = (double)a.HW * a.C;
    a.sums[2 * b] = (float)(s1 / n);
    a.sums[2 * b + 1] = (float)(s2 / n);
}

// one thread per (pixel, channel): loops over the batch -> dgamma / dbeta need no atomics
__global__ void ln_bwd_apply_kernel(const LNBwdArgs a) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const long long n = (long long)a.HW * a.C;
    if (i >= n) return;
    const int p = (int)(i / a.C), c = (int)(i - (long long)p * a.C);
    const float g = a.gamma[i];
    float dg = 0.f, db = 0.f;
    for (int b = 0; b < a.B; ++b) {
        const float dy = ln_dy(a, b, p, c);
        const float xh = a.xhat[(size_t)b * n + i];
        dg += dy * xh;
        db += dy;
        a.du[(size_t)b * n + i] = a.stats[2 * b + 1] * (dy * g - a.sums[2 * b] - xh * a.sums[2 * b + 1]);
    }
    a.dgamma[i] = dg;
    a.dbeta[i] = db;
}

hipError_t launch_layernorm_bwd(const float* dy, int ldy, int Cb, const int* blk, const float* xhat, const float* stats,
                                const float* gamma, int B, int HW, int C, double* partial, float* sums, float* du,
                                float* dgamma, float* dbeta, hipStream_t s) {
    LNBwdArgs a{};
    a.dy = dy; a.ldy = ldy; a.Cb = Cb;
    for (int i = 0; i < 8; ++i) a.blk[i] = blk ? blk[i] : i;
    a.xhat = xhat; a.stats = stats; a.gamma = gamma; a.B = B; a.HW = HW; a.C = C;
    a.partial = partial; a.sums = sums; a.du = du; a.dgamma = dgamma; a.dbeta = dbeta;
    VPX_LAUNCH(ln_bwd_partial_kernel, dim3(B, LN_CHUNKS), dim3(256), 0, s, a);
    VPX_LAUNCH(ln_bwd_finalize_kernel, dim3((B + 63) / 64), dim3(64), 0, s, a);
    const long long n = (long long)HW * C;
    VPX_LAUNCH(ln_bwd_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    return vpx_hip_last_error();
}

// ---- pointwise gate stages of the LayerNorm ST-LSTM (predrnn.py:61-81 on already-normalised conv outputs) -------------
__device__ __forceinline__ float ln_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * v)); }
__device__ __forceinline__ float ln_tanh(float v) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * v)); }

__global__ void st_ln_gates_kernel(const STLNGateArgs a) {
    const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const int Ch = a.Ch;
    if (e >= a.npix * Ch) return;
    const long long pix = e / Ch;
    const int ch = (int)(e - pix * Ch);
    const float* xc = a.xc + pix * 7 * Ch + ch;   // (i,f,g,i',f',g',o)
    const float* hc = a.hc + pix * 4 * Ch + ch;   // (i,f,g,o)
    const float* mc = a.mc + pix * 3 * Ch + ch;   // (i,f,g)
    const float i_ = ln_sigmoid(xc[0] + hc[0]);
    const float f_ = ln_sigmoid(xc[Ch] + hc[Ch] + 1.0f);
    const float g_ = ln_tanh(xc[2 * Ch] + hc[2 * Ch]);
    const float dc = i_ * g_;
    const float cn = f_ * a.c[e] + dc;
    const float ip = ln_sigmoid(xc[3 * Ch] + mc[0]);
    const float fp = ln_sigmoid(xc[4 * Ch] + mc[Ch] + 1.0f);
    const float gp = ln_tanh(xc[5 * Ch] + mc[2 * Ch]);
    const float dm = ip * gp;
    const float mn = fp * a.m[e] + dm;
    a.c_new[e] = cn; a.m_new[e] = mn; a.delta_c[e] = dc; a.delta_m[e] = dm;
    a.o_pre[e] = xc[6 * Ch] + hc[3 * Ch];
    a.mem[pix * 2 * Ch + ch] = cn;
    a.mem[pix * 2 * Ch + Ch + ch] = mn;
    if (a.gates_c) {
        float* gc = a.gates_c + pix * 3 * Ch + ch;
        gc[0] = i_; gc[Ch] = f_; gc[2 * Ch] = g_;
        float* gm = a.gates_m + pix * 3 * Ch + ch;
        gm[0] = ip; gm[Ch] = fp; gm[2 * Ch] = gp;
    }
}

__global__ void st_ln_out_kernel(const float* __restrict__ o_pre, const float* __restrict__ oc, const float* __restrict__ lc,
                                 float* __restrict__ h_new, float* __restrict__ o_save, float* __restrict__ tl_save,
                                 long long n) {
    const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float o_ = ln_sigmoid(o_pre[e] + (oc ? oc[e] : 0.0f));  // oc == null: o_pre already holds the full pre-activation
    const float tl = ln_tanh(lc[e]);
    h_new[e] = o_ * tl;
    if (o_save) { o_save[e] = o_; tl_save[e] = tl; }
}

hipError_t launch_st_ln_gates(const STLNGateArgs& a, hipStream_t s) {
    const long long n = a.npix * a.Ch;
    VPX_LAUNCH(st_ln_gates_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    return vpx_hip_last_error();
}
hipError_t launch_st_ln_out(const float* o_pre, const float* oc, const float* lc, float* h_new, float* o_save,
                            float* tl_save, long long n, hipStream_t s) {
    VPX_LAUNCH(st_ln_out_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, o_pre, oc, lc, h_new, o_save,
                       tl_save, n);
    return vpx_hip_last_error();
}

}  // namespace vpx


// ---- standalone LayerNorm([C,H,W]) over NHWC samples (the action-conditional ST-LSTM cell normalises each of its biased
//      convolutions on its own: predrnn.py:102-136) -----------------------------------------------------------------------
extern "C" {

size_t vpx_layernorm_workspace_bytes(int B) {
    if (B < 1) return 0;
    return vpx::align256((size_t)B * vpx::LN_CHUNKS * 2 * sizeof(double)) + vpx::align256((size_t)B * 2 * sizeof(float)) + 256;
}

int vpx_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* xhat, float* stats, int B, long long n,
                      void* workspace, size_t workspace_bytes, void* stream) {
    using namespace vpx;
    if (!x || !gamma || !beta || !y || !stats || B < 1 || n < 1) { set_error("vpx_layernorm_fwd: bad argument"); return VPX_ERR_ARG; }
    if (!workspace || workspace_bytes < vpx_layernorm_workspace_bytes(B)) { set_error("vpx_layernorm_fwd: workspace too small"); return VPX_ERR_WORKSPACE; }
    double* partial = reinterpret_cast<double*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    VPX_CHECK_HIP(launch_layernorm_fwd(x, gamma, beta, y, xhat, stats, partial, B, n, (hipStream_t)stream));
    return VPX_OK;
}

int vpx_layernorm_bwd(const float* dy, const float* xhat, const float* stats, const float* gamma, float* dx, float* dgamma, float* dbeta,
                      int B, int HW, int C, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace vpx;
    if (!dy || !xhat || !stats || !gamma || !dx || !dgamma || !dbeta || B < 1 || HW < 1 || C < 1) { set_error("vpx_layernorm_bwd: bad argument"); return VPX_ERR_ARG; }
    if (!workspace || workspace_bytes < vpx_layernorm_workspace_bytes(B)) { set_error("vpx_layernorm_bwd: workspace too small"); return VPX_ERR_WORKSPACE; }
    char* w = reinterpret_cast<char*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    double* partial = reinterpret_cast<double*>(w);
    float* sums = reinterpret_cast<float*>(w + align256((size_t)B * LN_CHUNKS * 2 * sizeof(double)));
    VPX_CHECK_HIP(launch_layernorm_bwd(dy, C, 0, nullptr, xhat, stats, gamma, B, HW, C, partial, sums, dx, dgamma, dbeta, (hipStream_t)stream));
    return VPX_OK;
}

}  // extern "C"
