// trajgru.hip — the non-convolution half of a TrajGRU step (vp_suite/model_blocks/traj_gru.py:148-162, 190-203) as HIP
// kernels, forward and backward, on NHWC tensors:
//   warp    L bilinear warps of the hidden state along the generated flow fields (grid_sample semantics of the reference:
//           bilinear, zero padding, align_corners=False, applied to the grid "pixel - flow" normalised by (W-1), (H-1)),
//           written channel-concatenated [B,HW,L*C] — the operand of the 1x1 `ret` convolution
//   gates   reset / update gates, candidate memory and the state update, one pass
// Both are HBM-bound streaming kernels; the five convolutions of the step run on the implicit-GEMM kernel (conv_gemm.hip).
#include "vpx_host.h"

namespace vpx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// source coordinate of the reference's warp: vgrid = x - f, normalised 2*v/(W-1) - 1 (traj_gru.py:158-159), un-normalised by
// grid_sample with align_corners=False: ((n + 1) * W - 1) / 2  ->  v * W/(W-1) - 0.5
struct WarpGeo { int B, H, W, C, L; float kx, ky; };

__device__ __forceinline__ void warp_coords(const WarpGeo& g, int x, int y, float fx, float fy, int& x0, int& y0, float& ax, float& ay) {
    const float sx = ((float)x - fx) * g.kx - 0.5f;
    const float sy = ((float)y - fy) * g.ky - 0.5f;
    const float flx = floorf(sx), fly = floorf(sy);
    x0 = (int)flx; y0 = (int)fly;
    ax = sx - flx; ay = sy - fly;
}

// one thread per (b, pixel, l, 4-channel group)
__global__ __launch_bounds__(256) void trajgru_warp_fwd_kernel(const WarpGeo g, const float* __restrict__ h, const float* __restrict__ flows,
                                                               float* __restrict__ warped) {
    const int c4n = g.C >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)g.B * g.H * g.W * g.L * c4n;
    if (idx >= total) return;
    const int c4 = (int)(idx % c4n);
    long long r = idx / c4n;
    const int l = (int)(r % g.L); r /= g.L;
    const int pix = (int)(r % ((long long)g.H * g.W));
    const int b = (int)(r / ((long long)g.H * g.W));
    const int y = pix / g.W, x = pix - y * g.W;
    const float* fl = flows + ((size_t)b * g.H * g.W + pix) * (2 * g.L) + 2 * l;
    int x0, y0;
    float ax, ay;
    warp_coords(g, x, y, fl[0], fl[1], x0, y0, ax, ay);
    const float* hb = h + (size_t)b * g.H * g.W * g.C + c4 * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int xx = x0 + dx, yy = y0 + dy;
            if (xx < 0 || xx >= g.W || yy < 0 || yy >= g.H) continue;   // zero padding
            const float wgt = (dx ? ax : 1.0f - ax) * (dy ? ay : 1.0f - ay);
            const f32x4 v = *reinterpret_cast<const f32x4*>(hb + ((size_t)yy * g.W + xx) * g.C);
            acc += v * wgt;
        }
    *reinterpret_cast<f32x4*>(warped + ((size_t)b * g.H * g.W + pix) * ((size_t)g.L * g.C) + (size_t)l * g.C + c4 * 4) = acc;
}

// backward: one thread per (b, pixel, l) walks the channels: dh (+)= scatter of d_warped with the bilinear weights (float
// atomics: the scatter targets are data dependent — like torch's grid_sample backward the sum order is not reproducible),
// d_flow = -(d sample / d coordinate) summed over the channels (the warp uses pixel - flow)
// FIXED (vpx_set_deterministic(1) / vpx_trajgru_warp_bwd_det): the scatter adds 2^40-scaled 64-bit integers instead of floats —
// integer addition is associative, so the sum no longer depends on the order in which the atomics arrive (bit-reproducible);
// resolution 9e-13, range +-8.4e6 per element. trajgru_fixed_to_float_kernel then adds the converted sums onto dh.
constexpr double WARP_FIXED_SCALE = 1099511627776.0;   // 2^40
__device__ __forceinline__ void warp_fixed_add(long long* acc, size_t i, float v) {
    atomicAdd(reinterpret_cast<unsigned long long*>(acc) + i, (unsigned long long)__double2ll_rn((double)v * WARP_FIXED_SCALE));
}
__global__ __launch_bounds__(256) void trajgru_fixed_to_float_kernel(const long long* __restrict__ acc, float* __restrict__ dh, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dh[i] += (float)((double)acc[i] * (1.0 / WARP_FIXED_SCALE));
}

template <bool FIXED>
__global__ __launch_bounds__(256) void trajgru_warp_bwd_kernel(const WarpGeo g, const float* __restrict__ h, const float* __restrict__ flows,
                                                               const float* __restrict__ dwarped, float* __restrict__ dh,
                                                               float* __restrict__ dflows, long long* __restrict__ acc) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)g.B * g.H * g.W * g.L;
    if (idx >= total) return;
    const int l = (int)(idx % g.L);
    long long r = idx / g.L;
    const int pix = (int)(r % ((long long)g.H * g.W));
    const int b = (int)(r / ((long long)g.H * g.W));
    const int y = pix / g.W, x = pix - y * g.W;
    const size_t fo = ((size_t)b * g.H * g.W + pix) * (2 * g.L) + 2 * l;
    int x0, y0;
    float ax, ay;
    warp_coords(g, x, y, flows[fo], flows[fo + 1], x0, y0, ax, ay);
    const float* hb = h + (size_t)b * g.H * g.W * g.C;
    float* dhb = dh ? dh + (size_t)b * g.H * g.W * g.C : nullptr;
    long long* accb = acc ? acc + (size_t)b * g.H * g.W * g.C : nullptr;
    const float* dw = dwarped + ((size_t)b * g.H * g.W + pix) * ((size_t)g.L * g.C) + (size_t)l * g.C;
    bool ok[2][2];
    size_t off[2][2];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int xx = x0 + dx, yy = y0 + dy;
            ok[dy][dx] = xx >= 0 && xx < g.W && yy >= 0 && yy < g.H;
            off[dy][dx] = ok[dy][dx] ? ((size_t)yy * g.W + xx) * g.C : 0;
        }
    float gx = 0.f, gy = 0.f;
    for (int c = 0; c < g.C; ++c) {
        const float d = dw[c];
        const float v00 = ok[0][0] ? hb[off[0][0] + c] : 0.f, v01 = ok[0][1] ? hb[off[0][1] + c] : 0.f;
        const float v10 = ok[1][0] ? hb[off[1][0] + c] : 0.f, v11 = ok[1][1] ? hb[off[1][1] + c] : 0.f;
        gx += d * ((1.0f - ay) * (v01 - v00) + ay * (v11 - v10));
        gy += d * ((1.0f - ax) * (v10 - v00) + ax * (v11 - v01));
        if (FIXED) {
            if (accb) {
                if (ok[0][0]) warp_fixed_add(accb, off[0][0] + c, d * (1.0f - ax) * (1.0f - ay));
                if (ok[0][1]) warp_fixed_add(accb, off[0][1] + c, d * ax * (1.0f - ay));
                if (ok[1][0]) warp_fixed_add(accb, off[1][0] + c, d * (1.0f - ax) * ay);
                if (ok[1][1]) warp_fixed_add(accb, off[1][1] + c, d * ax * ay);
            }
        } else if (dhb) {
            if (ok[0][0]) unsafeAtomicAdd(dhb + off[0][0] + c, d * (1.0f - ax) * (1.0f - ay));
            if (ok[0][1]) unsafeAtomicAdd(dhb + off[0][1] + c, d * ax * (1.0f - ay));
            if (ok[1][0]) unsafeAtomicAdd(dhb + off[1][0] + c, d * (1.0f - ax) * ay);
            if (ok[1][1]) unsafeAtomicAdd(dhb + off[1][1] + c, d * ax * ay);
        }
    }
    if (dflows) {
        dflows[fo] = -gx * g.kx;       // d sx / d f_x = -W/(W-1)
        dflows[fo + 1] = -gy * g.ky;
    }
}

// ---- gates -------------------------------------------------------------------------------------------------------------
struct GruArgs {
    long long n;              // B*HW*C
    int HWC, C;               // elements per image, channels
    int act; float slope;     // candidate activation: 0 leaky/relu (slope), 1 sigmoid
    const float* i2h; long long i2h_bs;   // [B][HW][3C] slice of the input projection (batch stride in elements) or null
    const float* h2h;         // [B][HW][3C]
    const float* prev;        // [B][HW][C]
    float* next;              // [B][HW][C]
    float* save;              // [B][HW][3C]: r, u, candidate (post-activation) — or null
};

__device__ __forceinline__ float gru_act(float v, int act, float slope) {
    return act == 1 ? sigmoid_f(v) : (v > 0.0f ? v : v * slope);
}

__global__ __launch_bounds__(256) void trajgru_gates_fwd_kernel(const GruArgs a) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= a.n) return;
    const int b = (int)(e / a.HWC);
    const int rem = (int)(e - (long long)b * a.HWC);
    const int pix = rem / a.C, c = rem - pix * a.C;
    const size_t g3 = ((size_t)b * (a.HWC / a.C) + pix) * (3 * a.C) + c;
    float r_pre = a.h2h[g3], u_pre = a.h2h[g3 + a.C];
    const float hm = a.h2h[g3 + 2 * a.C];
    float m_pre = 0.f;
    if (a.i2h) {
        const float* ip = a.i2h + (size_t)b * a.i2h_bs + (size_t)pix * (3 * a.C) + c;
        r_pre += ip[0]; u_pre += ip[a.C]; m_pre = ip[2 * a.C];
    }
    const float r = sigmoid_f(r_pre), u = sigmoid_f(u_pre);
    const float m = gru_act(m_pre + r * hm, a.act, a.slope);
    a.next[e] = u * a.prev[e] + (1.0f - u) * m;     // traj_gru.py:203
    if (a.save) { a.save[g3] = r; a.save[g3 + a.C] = u; a.save[g3 + 2 * a.C] = m; }
}

struct GruBwdArgs {
    long long n; int HWC, C; int act; float slope;
    const float* dnext;       // [B][HW][C] total gradient of next_h
    const float* h2h; const float* prev; const float* save;
    float* di2h; long long di2h_bs;   // slice of the d(input projection) slab, or null
    float* dh2h;              // [B][HW][3C]
    float* dprev;             // [B][HW][C]: direct path u * dnext (written)
};

__global__ __launch_bounds__(256) void trajgru_gates_bwd_kernel(const GruBwdArgs a) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= a.n) return;
    const int b = (int)(e / a.HWC);
    const int rem = (int)(e - (long long)b * a.HWC);
    const int pix = rem / a.C, c = rem - pix * a.C;
    const size_t g3 = ((size_t)b * (a.HWC / a.C) + pix) * (3 * a.C) + c;
    const float r = a.save[g3], u = a.save[g3 + a.C], m = a.save[g3 + 2 * a.C];
    const float hm = a.h2h[g3 + 2 * a.C];
    const float dn = a.dnext[e];
    const float du = dn * (a.prev[e] - m) * u * (1.0f - u);
    const float dm = dn * (1.0f - u);
    const float dact = a.act == 1 ? m * (1.0f - m) : (m > 0.0f ? 1.0f : a.slope);
    const float dpre = dm * dact;
    const float dr = dpre * hm * r * (1.0f - r);
    a.dprev[e] = dn * u;
    a.dh2h[g3] = dr; a.dh2h[g3 + a.C] = du; a.dh2h[g3 + 2 * a.C] = dpre * r;
    if (a.di2h) {
        float* ip = a.di2h + (size_t)b * a.di2h_bs + (size_t)pix * (3 * a.C) + c;
        ip[0] = dr; ip[a.C] = du; ip[2 * a.C] = dpre;
    }
}

}  // namespace vpx

using namespace vpx;

extern "C" {

static int warp_geo(WarpGeo& g, int B, int H, int W, int C, int L, const char* who) {
    if (B < 1 || H < 1 || W < 1 || C < 1 || L < 1) { set_error("%s: non-positive dimension", who); return VPX_ERR_ARG; }
    if (C & 3) { set_error("%s: the channel count must be a multiple of 4 (got %d)", who, C); return VPX_ERR_UNSUPPORTED; }
    g = WarpGeo{B, H, W, C, L, (float)W / (float)(W > 1 ? W - 1 : 1), (float)H / (float)(H > 1 ? H - 1 : 1)};
    return VPX_OK;
}

int vpx_trajgru_warp_fwd(const float* h, const float* flows, float* warped, int B, int H, int W, int C, int L, void* stream) {
    WarpGeo g;
    int rc = warp_geo(g, B, H, W, C, L, "vpx_trajgru_warp_fwd");
    if (rc != VPX_OK) return rc;
    if (!h || !flows || !warped) { set_error("vpx_trajgru_warp_fwd: NULL tensor argument"); return VPX_ERR_ARG; }
    const long long total = (long long)B * H * W * L * (C / 4);
    VPX_LAUNCH(trajgru_warp_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, h, flows, warped);
    VPX_CHECK_HIP(vpx_hip_last_error());
    return VPX_OK;
}

int vpx_trajgru_warp_bwd(const float* h, const float* flows, const float* dwarped, float* dh, float* dflows, int B, int H, int W,
                         int C, int L, void* stream) {
    WarpGeo g;
    int rc = warp_geo(g, B, H, W, C, L, "vpx_trajgru_warp_bwd");
    if (rc != VPX_OK) return rc;
    if (!h || !flows || !dwarped) { set_error("vpx_trajgru_warp_bwd: NULL tensor argument"); return VPX_ERR_ARG; }
    if (vpx::g_deterministic && dh) {
        set_error("vpx_trajgru_warp_bwd: deterministic mode is on — call vpx_trajgru_warp_bwd_det (it needs a workspace for the fixed-point sums)");
        return VPX_ERR_UNSUPPORTED;
    }
    const long long total = (long long)B * H * W * L;
    VPX_LAUNCH(trajgru_warp_bwd_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, h, flows,
                       dwarped, dh, dflows, (long long*)nullptr);
    VPX_CHECK_HIP(vpx_hip_last_error());
    return VPX_OK;
}

size_t vpx_trajgru_warp_bwd_det_workspace_bytes(int B, int H, int W, int C) {
    if (B < 1 || H < 1 || W < 1 || C < 1) return 0;
    return (size_t)B * H * W * C * sizeof(long long) + 256;
}

int vpx_trajgru_warp_bwd_det(const float* h, const float* flows, const float* dwarped, float* dh, float* dflows, int B, int H, int W,
                             int C, int L, void* workspace, size_t workspace_bytes, void* stream_) {
    WarpGeo g;
    int rc = warp_geo(g, B, H, W, C, L, "vpx_trajgru_warp_bwd_det");
    if (rc != VPX_OK) return rc;
    if (!h || !flows || !dwarped) { set_error("vpx_trajgru_warp_bwd_det: NULL tensor argument"); return VPX_ERR_ARG; }
    if (dh && (!workspace || workspace_bytes < vpx_trajgru_warp_bwd_det_workspace_bytes(B, H, W, C))) {
        set_error("vpx_trajgru_warp_bwd_det: workspace too small");
        return VPX_ERR_WORKSPACE;
    }
    hipStream_t stream = (hipStream_t)stream_;
    long long* acc = dh ? reinterpret_cast<long long*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255) : nullptr;
    const long long n = (long long)B * H * W * C, total = (long long)B * H * W * L;
    if (acc) VPX_CHECK_HIP(vpx_memset_async(acc, 0, (size_t)n * sizeof(long long), stream));
    VPX_LAUNCH(trajgru_warp_bwd_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, g, h, flows, dwarped, dh,
                       dflows, acc);
    VPX_CHECK_HIP(vpx_hip_last_error());
    if (acc) {
        VPX_LAUNCH(trajgru_fixed_to_float_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, acc, dh, n);
        VPX_CHECK_HIP(vpx_hip_last_error());
    }
    return VPX_OK;
}

int vpx_trajgru_gates_fwd(const float* i2h, long long i2h_bstride, const float* h2h, const float* prev, float* next, float* save,
                          int B, int HW, int C, int act, float slope, void* stream) {
    if (B < 1 || HW < 1 || C < 1 || !h2h || !prev || !next || (act != 0 && act != 1)) { set_error("vpx_trajgru_gates_fwd: bad argument"); return VPX_ERR_ARG; }
    GruArgs a{(long long)B * HW * C, HW * C, C, act, slope, i2h, i2h_bstride, h2h, prev, next, save};
    VPX_LAUNCH(trajgru_gates_fwd_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    VPX_CHECK_HIP(vpx_hip_last_error());
    return VPX_OK;
}

int vpx_trajgru_gates_bwd(const float* dnext, const float* h2h, const float* prev, const float* save, float* di2h,
                          long long di2h_bstride, float* dh2h, float* dprev, int B, int HW, int C, int act, float slope, void* stream) {
    if (B < 1 || HW < 1 || C < 1 || !dnext || !h2h || !prev || !save || !dh2h || !dprev || (act != 0 && act != 1)) {
        set_error("vpx_trajgru_gates_bwd: bad argument");
        return VPX_ERR_ARG;
    }
    GruBwdArgs a{(long long)B * HW * C, HW * C, C, act, slope, dnext, h2h, prev, save, di2h, di2h_bstride, dh2h, dprev};
    VPX_LAUNCH(trajgru_gates_bwd_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    VPX_CHECK_HIP(vpx_hip_last_error());
    return VPX_OK;
}

int vpx_axpy(float* y, const float* x, long long n, void* stream) {
    if (!y || !x || n < 0) { set_error("vpx_axpy: bad argument"); return VPX_ERR_ARG; }
    if (n) VPX_CHECK_HIP(launch_axpy(y, x, n, (hipStream_t)stream));
    return VPX_OK;
}

/* d(pre-activation) = dy * LeakyReLU'(y) (slope >= 0; from the sign of the activated output y) and its column sums (bias
 * gradient), one pass, fixed summation order. dys and db may each be NULL. workspace: COLSUM_BLOCKS * cols floats. */
size_t vpx_leaky_bwd_workspace_bytes(int cols) { return cols < 1 ? 0 : align256((size_t)COLSUM_BLOCKS * cols * sizeof(float)) + 256; }
int vpx_leaky_bwd(const float* dy, const float* y, float slope, float* dys, float* db, long long rows, int cols, void* workspace,
                  size_t workspace_bytes, void* stream) {
    if (!dy || !y || rows < 1 || cols < 1 || slope < 0.0f) { set_error("vpx_leaky_bwd: bad argument"); return VPX_ERR_ARG; }
    if (!workspace || workspace_bytes < vpx_leaky_bwd_workspace_bytes(cols)) { set_error("vpx_leaky_bwd: workspace too small"); return VPX_ERR_WORKSPACE; }
    float* part = reinterpret_cast<float*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    VPX_CHECK_HIP(launch_colsum(dy, y, slope, dys, db, part, rows, cols, (hipStream_t)stream));
    return VPX_OK;
}

}  // extern "C"
