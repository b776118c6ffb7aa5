// Training tail of the hot path's caller: the prediction loss with its gradient in one pass, and
// one Adam update over the flat parameter / gradient buckets. Both are streaming, HBM-bound kernels.
//   vpx_mse_loss      MSE summed over (c,h,w), averaged over frames (t) then samples (b):
//                     vp_suite/base/base_measure.py:55-57 with nn.MSELoss(reduction="none") (measure/image_wise.py:25),
//                     scaled and summed by PredictionLossProvider.get_losses (measure/loss_provider.py:48-51)
//   vpx_adam_step     torch.optim.Adam(params, lr) as constructed in vp_suite/vpsuite.py:353 (betas 0.9/0.999, eps 1e-8,
//                     no weight decay, no amsgrad), called once per iteration at base_model.py:176
#include <hip/hip_runtime.h>
#include <math.h>
#include "vpx_internal.h"
#include "vpx_host.h"

namespace vpx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MSE_THREADS = 256;
constexpr int MSE_MAX_BLOCKS = 1024;

// grad (optional) = 2 * scale / n_frames * (pred - target); partial[block] = sum (pred - target)^2 in double
__global__ __launch_bounds__(MSE_THREADS) void mse_partial_kernel(const float* __restrict__ pred,
                                                                  const float* __restrict__ target, long long n,
                                                                  float gscale, float* __restrict__ grad,
                                                                  double* __restrict__ partial) {
    __shared__ double red[MSE_THREADS / 64];
    double acc = 0.0;
    const long long stride = (long long)gridDim.x * MSE_THREADS * 4;
    const bool vec = (((uintptr_t)pred | (uintptr_t)target | (uintptr_t)grad) & 15) == 0;
    for (long long e = ((long long)blockIdx.x * MSE_THREADS + threadIdx.x) * 4; e < n; e += stride) {
        if (vec && e + 3 < n) {
            const f32x4 p = *reinterpret_cast<const f32x4*>(pred + e);
            const f32x4 t = *reinterpret_cast<const f32x4*>(target + e);
            f32x4 d;
#pragma unroll
            for (int k = 0; k < 4; ++k) { d[k] = p[k] - t[k]; acc += (double)d[k] * d[k]; d[k] *= gscale; }
            if (grad) *reinterpret_cast<f32x4*>(grad + e) = d;
        } else {
            for (long long k = e; k < n && k < e + 4; ++k) {
                const float d = pred[k] - target[k];
                acc += (double)d * d;
                if (grad) grad[k] = d * gscale;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < MSE_THREADS / 64; ++w) s += red[w];
        partial[blockIdx.x] = s;
    }
}

// loss = scale / n_frames * sum(partial): one wave, fixed summation order (deterministic)
__global__ void mse_final_kernel(const double* __restrict__ partial, int nblocks, double mult, float* __restrict__ loss) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 64) acc += partial[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (threadIdx.x == 0) *loss = (float)(acc * mult);
}

struct AdamArgs {
    float* p; const float* g; float* m; float* v;
    long long n;
    // scalars are prepared in double on the host exactly as torch/optim/adam.py does in Python floats, then rounded once
    float beta1, beta2, omb1, omb2;      // beta, 1 - beta
    float step_size, bc2_sqrt, eps;      // lr / (1 - beta1^t), sqrt(1 - beta2^t)
    float weight_decay, grad_scale;
};

// torch.optim.Adam single-tensor update (torch/optim/adam.py _single_tensor_adam, amsgrad = False, maximize = False):
//   g += wd * p ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adam_kernel(const AdamArgs a) {
    const long long stride = (long long)gridDim.x * 256 * 4;
    const float step_size = a.step_size;
    for (long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; e < a.n; e += stride) {
        if (e + 3 < a.n) {  // the buckets are 256-byte aligned allocations: vector path
            f32x4 p = *reinterpret_cast<const f32x4*>(a.p + e);
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(a.g + e);
            f32x4 m = *reinterpret_cast<const f32x4*>(a.m + e);
            f32x4 v = *reinterpret_cast<const f32x4*>(a.v + e);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float g = g4[k] * a.grad_scale;
                if (a.weight_decay != 0.0f) g += a.weight_decay * p[k];
                m[k] = a.beta1 * m[k] + a.omb1 * g;
                v[k] = a.beta2 * v[k] + a.omb2 * g * g;
                const float denom = sqrtf(v[k]) / a.bc2_sqrt + a.eps;
                p[k] -= step_size * (m[k] / denom);
            }
            *reinterpret_cast<f32x4*>(a.p + e) = p;
            *reinterpret_cast<f32x4*>(a.m + e) = m;
            *reinterpret_cast<f32x4*>(a.v + e) = v;
        } else {
            for (long long k = e; k < a.n; ++k) {
                float g = a.g[k] * a.grad_scale;
                if (a.weight_decay != 0.0f) g += a.weight_decay * a.p[k];
                const float m = a.beta1 * a.m[k] + a.omb1 * g;
                const float v = a.beta2 * a.v[k] + a.omb2 * g * g;
                a.m[k] = m; a.v[k] = v;
                a.p[k] -= step_size * (m / (sqrtf(v) / a.bc2_sqrt + a.eps));
            }
        }
    }
}

}  // namespace vpx

using namespace vpx;

extern "C" {

size_t vpx_mse_loss_workspace_bytes(void) { return MSE_MAX_BLOCKS * sizeof(double) + 256; }

int vpx_mse_loss(const float* pred, const float* target, long long n_elements, long long n_frames, float scale, float* loss,
                 float* dpred, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!pred || !target || !loss || n_elements < 1 || n_frames < 1) { set_error("vpx_mse_loss: bad argument"); return VPX_ERR_ARG; }
    if (!workspace || workspace_bytes < vpx_mse_loss_workspace_bytes()) { set_error("vpx_mse_loss: workspace too small"); return VPX_ERR_WORKSPACE; }
    hipStream_t stream = (hipStream_t)stream_;
    double* partial = reinterpret_cast<double*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    long long blocks = (n_elements + MSE_THREADS * 4 - 1) / (MSE_THREADS * 4);
    if (blocks > MSE_MAX_BLOCKS) blocks = MSE_MAX_BLOCKS;
    const double mult = (double)scale / (double)n_frames;
    VPX_LAUNCH(mse_partial_kernel, dim3((unsigned)blocks), dim3(MSE_THREADS), 0, stream, pred, target, n_elements,
                       (float)(2.0 * mult), dpred, partial);
    VPX_CHECK_HIP(vpx_hip_last_error());
    VPX_LAUNCH(mse_final_kernel, dim3(1), dim3(64), 0, stream, partial, (int)blocks, mult, loss);
    VPX_CHECK_HIP(vpx_hip_last_error());
    return VPX_OK;
}

int vpx_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, double lr, double beta1,
                  double beta2, double eps, double weight_decay, int step, double grad_scale, void* stream_) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 1 || step < 1) { set_error("vpx_adam_step: bad argument"); return VPX_ERR_ARG; }
    if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0) {
        set_error("vpx_adam_step: buckets must be 16-byte aligned");
        return VPX_ERR_ARG;
    }
    AdamArgs a{param, grad, exp_avg, exp_avg_sq, n, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2),
               (float)(lr / (1.0 - pow(beta1, step))), (float)sqrt(1.0 - pow(beta2, step)), (float)eps,
               (float)weight_decay, (float)grad_scale};
    long long blocks = (n + 1023) / 1024;
    if (blocks > 2048) blocks = 2048;
    VPX_LAUNCH(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, a);
    VPX_CHECK_HIP(vpx_hip_last_error());
    return VPX_OK;
}

}  // extern "C"
