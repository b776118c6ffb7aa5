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
        hipLaunchKernelGGL(transpose_rc_kernel, grid, dim3(32, 8), 0, s, src + (size_t)n0 * R * C,
                           dst + (size_t)n0 * R * C, R, C);
    }
    return hipGetLastError();
}

hipError_t launch_nchw_to_nhwc(const float* src, float* dst, int N, int C, int H, int W, hipStream_t s) {
    return launch_transpose(src, dst, N, C, H * W, s);  // [N][C][HW] -> [N][HW][C]
}
hipError_t launch_nhwc_to_nchw(const float* src, float* dst, int N, int C, int H, int W, hipStream_t s) {
    return launch_transpose(src, dst, N, H * W, C, s);  // [N][HW][C] -> [N][C][HW]
}

}  // namespace vpx
