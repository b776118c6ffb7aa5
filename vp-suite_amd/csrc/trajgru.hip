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

namespace {

int warp_geo(WarpGeo& g, int B, int H, int W, int C, int L, const char* who) {
    if (B < 1 || H < 1 || W < 1 || C < 1 || L < 1) { set_error("%s: non-positive dimension", who); return VPX_ERR_ARG; }
    if (C & 3) { set_error("%s: the channel count must be a multiple of 4 (got %d)", who, C); return VPX_ERR_UNSUPPORTED; }
    g = WarpGeo{B, H, W, C, L, (float)W / (float)(W > 1 ? W - 1 : 1), (float)H / (float)(H > 1 ? H - 1 : 1)};
    return VPX_OK;
}

int tg_warp_fwd(const float* h, const float* flows, float* warped, int B, int H, int W, int C, int L, void* stream) {
    WarpGeo g;
    int rc = warp_geo(g, B, H, W, C, L, "vpx_trajgru_seq: warp_fwd");
    if (rc != VPX_OK) return rc;
    if (!h || !flows || !warped) { set_error("vpx_trajgru_seq: warp_fwd: NULL tensor argument"); return VPX_ERR_ARG; }
    const long long total = (long long)B * H * W * L * (C / 4);
    VPX_LAUNCH(trajgru_warp_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, h, flows, warped);
    VPX_CHECK_HIP(vpx_hip_last_error());
    return VPX_OK;
}

int tg_warp_bwd(const float* h, const float* flows, const float* dwarped, float* dh, float* dflows, int B, int H, int W,
                         int C, int L, void* stream) {
    WarpGeo g;
    int rc = warp_geo(g, B, H, W, C, L, "vpx_trajgru_seq: warp_bwd");
    if (rc != VPX_OK) return rc;
    if (!h || !flows || !dwarped) { set_error("vpx_trajgru_seq: warp_bwd: NULL tensor argument"); return VPX_ERR_ARG; }
    if (vpx::g_deterministic && dh) {
        set_error("vpx_trajgru_seq: warp_bwd: deterministic mode needs the fixed-point form");
        return VPX_ERR_UNSUPPORTED;
    }
    const long long total = (long long)B * H * W * L;
    VPX_LAUNCH(trajgru_warp_bwd_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, h, flows,
                       dwarped, dh, dflows, (long long*)nullptr);
    VPX_CHECK_HIP(vpx_hip_last_error());
    return VPX_OK;
}

size_t tg_warp_bwd_det_workspace_bytes(int B, int H, int W, int C) {
    if (B < 1 || H < 1 || W < 1 || C < 1) return 0;
    return (size_t)B * H * W * C * sizeof(long long) + 256;
}

int tg_warp_bwd_det(const float* h, const float* flows, const float* dwarped, float* dh, float* dflows, int B, int H, int W,
                             int C, int L, void* workspace, size_t workspace_bytes, void* stream_) {
    WarpGeo g;
    int rc = warp_geo(g, B, H, W, C, L, "vpx_trajgru_seq: warp_bwd_det");
    if (rc != VPX_OK) return rc;
    if (!h || !flows || !dwarped) { set_error("vpx_trajgru_seq: warp_bwd_det: NULL tensor argument"); return VPX_ERR_ARG; }
    if (dh && (!workspace || workspace_bytes < tg_warp_bwd_det_workspace_bytes(B, H, W, C))) {
        set_error("vpx_trajgru_seq: warp_bwd_det: workspace too small");
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

int tg_gates_fwd(const float* i2h, long long i2h_bstride, const float* h2h, const float* prev, float* next, float* save,
                          int B, int HW, int C, int act, float slope, void* stream) {
    if (B < 1 || HW < 1 || C < 1 || !h2h || !prev || !next || (act != 0 && act != 1)) { set_error("vpx_trajgru_seq: gates_fwd: bad argument"); return VPX_ERR_ARG; }
    GruArgs a{(long long)B * HW * C, HW * C, C, act, slope, i2h, i2h_bstride, h2h, prev, next, save};
    VPX_LAUNCH(trajgru_gates_fwd_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    VPX_CHECK_HIP(vpx_hip_last_error());
    return VPX_OK;
}

int tg_gates_bwd(const float* dnext, const float* h2h, const float* prev, const float* save, float* di2h,
                          long long di2h_bstride, float* dh2h, float* dprev, int B, int HW, int C, int act, float slope, void* stream) {
    if (B < 1 || HW < 1 || C < 1 || !dnext || !h2h || !prev || !save || !dh2h || !dprev || (act != 0 && act != 1)) {
        set_error("vpx_trajgru_seq: gates_bwd: bad argument");
        return VPX_ERR_ARG;
    }
    GruBwdArgs a{(long long)B * HW * C, HW * C, C, act, slope, dnext, h2h, prev, save, di2h, di2h_bstride, dh2h, dprev};
    VPX_LAUNCH(trajgru_gates_bwd_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    VPX_CHECK_HIP(vpx_hip_last_error());
    return VPX_OK;
}

}  // namespace

// =====================================================================================================================
// The whole sequence behind one entry point each way (round 5): the time loop of traj_gru.py:164-214 and its BPTT schedule as
// library launches over ONE caller workspace (carved and bounds-checked like every other entry point). Round 2-4 drove the step
// pieces above from Python (an autograd Function that re-sized a shared torch workspace per call site); the pieces are internal now.
// Layout: time-major NHWC — x [T][B][H*W][Cin], hs [T][B][H*W][C] — so a time slice is a dense batch of images.
namespace {

constexpr int TG_F = 32;   // channels of the flow generator's hidden layer (traj_gru.py:108-122, fixed by the reference)

struct TGLayout {
    size_t n_h, n3, n_fl, n_f1, n_warp, n_x;
    size_t wpk, slabs, colsum, det;
    size_t w_max, b_max;   // largest per-step parameter gradient (temporaries of the accumulation)
};

int tg_check(const vpx_trajgru_desc* d, const char* who) {
    if (!d) { set_error("%s: desc is NULL", who); return VPX_ERR_ARG; }
    if (d->B < 1 || d->T < 1 || d->C < 1 || d->H < 1 || d->W < 1 || d->L < 1 || d->Cin < 1) { set_error("%s: non-positive dimension", who); return VPX_ERR_ARG; }
    if (d->k_i2h < 1 || !(d->k_i2h & 1) || d->k_i2h > 7) { set_error("%s: i2h kernel must be odd and <= 7 (got %d)", who, d->k_i2h); return VPX_ERR_ARG; }
    if (d->C & 3) { set_error("%s: the channel count must be a multiple of 4 (got %d)", who, d->C); return VPX_ERR_UNSUPPORTED; }
    if (d->precision < VPX_PREC_F32 || d->precision > VPX_PREC_BF16) { set_error("%s: unknown precision %d", who, d->precision); return VPX_ERR_UNSUPPORTED; }
    if (!(d->slope > 0.0f)) { set_error("%s: the activation must be LeakyReLU with a positive slope", who); return VPX_ERR_UNSUPPORTED; }
    return VPX_OK;
}

TGLayout tg_layout(const vpx_trajgru_desc* d) {
    TGLayout L{};
    const size_t px = (size_t)d->B * d->H * d->W;
    const int C = d->C, Cin = d->Cin, k = d->k_i2h, LC = d->L * C, F = TG_F, L2 = 2 * d->L;
    L.n_h = px * C; L.n3 = px * 3 * C; L.n_fl = px * L2; L.n_f1 = px * F; L.n_warp = px * LC; L.n_x = px * Cin;
    auto mx = [](size_t& a, size_t b) { if (b > a) a = b; };
    // forward packs, and the transposed packs of the data gradients (same bound with Ci / Co exchanged)
    const int cv[5][3] = {{Cin, 3 * C, k}, {C, F, 5}, {Cin, F, 5}, {F, L2, 5}, {LC, 3 * C, 1}};
    for (auto& c : cv) {
        mx(L.wpk, plain_conv_wpk_floats(c[0], c[1], c[2], c[2]));
        mx(L.wpk, plain_conv_wpk_floats(c[1], c[0], c[2], c[2]));
        mx(L.w_max, (size_t)c[0] * c[1] * c[2] * c[2]);
        mx(L.b_max, (size_t)c[1]);
    }
    // weight-gradient K-slice slabs: the per-step layers over B images, i2h over all T*B
    const int ns1 = wgrad_slices(d->B, d->H, d->W), nsT = wgrad_slices(d->B * d->T, d->H, d->W);
    mx(L.slabs, (size_t)nsT * k * k * 3 * C * Cin);
    mx(L.slabs, (size_t)ns1 * 25 * F * C);
    mx(L.slabs, (size_t)ns1 * 25 * F * Cin);
    mx(L.slabs, (size_t)ns1 * 25 * L2 * F);
    mx(L.slabs, (size_t)ns1 * 3 * C * LC);
    L.colsum = (size_t)COLSUM_BLOCKS * (3 * C > F ? 3 * C : F);
    if (L.colsum < (size_t)COLSUM_BLOCKS * L2) L.colsum = (size_t)COLSUM_BLOCKS * L2;
    L.det = tg_warp_bwd_det_workspace_bytes(d->B, d->H, d->W, C) / 4 + 64;
    return L;
}

// y = act(conv(x; w) + b [+ y]) on N images (the forward of traj_ops' `conv`)
int tg_conv(hipStream_t s, int prec, int N, int H, int W, const float* x, const float* w, const float* b, float* y, int Ci, int Co, int k,
            bool accumulate, float leaky, float* wpk) {
    const ConvGeo g{N, H, W};
    return plain_conv(s, prec, g, x, Ci, Ci, w, (long long)Ci * k * k, k * k, k, k, Co, false, b, y, Co, accumulate, wpk, leaky);
}
// dx = conv^T(dy; w), dw = wgrad(dy, x), db = colsum(dy); each destination may be NULL
int tg_conv_bwd(hipStream_t s, int prec, int N, int H, int W, const float* x, const float* w, const float* dy, float* dx, float* dw, float* db,
                int Ci, int Co, int k, float* wpk, float* slabs, float* db_part) {
    const ConvGeo g{N, H, W};
    int rc;
    if (dx && (rc = plain_conv(s, prec, g, dy, Co, Co, w, (long long)Ci * k * k, k * k, k, k, Ci, true, nullptr, dx, Ci, false, wpk))) return rc;
    if (dw && (rc = plain_wgrad(s, prec, g, dy, Co, x, Ci, k, k, slabs, dw))) return rc;
    if (db) VPX_CHECK_HIP(launch_colsum(dy, nullptr, 0.f, nullptr, db, db_part, (long long)N * H * W, Co, s));
    return VPX_OK;
}

}  // namespace

extern "C" {

size_t vpx_trajgru_reserve_bytes(const vpx_trajgru_desc* d) {
    if (tg_check(d, "vpx_trajgru_reserve_bytes") != VPX_OK || !(d->flags & VPX_FLAG_SAVE_FOR_BWD)) return 0;
    const TGLayout L = tg_layout(d);
    // per step: flows, flow features, the `ret` output, the gate kernel's (r, u, candidate)
    return (size_t)d->T * (align256(L.n_fl * 4) + align256(L.n_f1 * 4) + 2 * align256(L.n3 * 4)) + 256;
}

size_t vpx_trajgru_workspace_bytes(const vpx_trajgru_desc* d) {
    if (tg_check(d, "vpx_trajgru_workspace_bytes") != VPX_OK) return 0;
    const TGLayout L = tg_layout(d);
    const size_t T = (size_t)d->T;
    // forward: pack, input projection of all frames, warped operand, a zero state, one step's flows / features / ret output
    size_t fwd = align256(L.wpk * 4) + align256(T * L.n3 * 4) + align256(L.n_warp * 4) + align256(L.n_h * 4) +
                 align256(L.n_fl * 4) + align256(L.n_f1 * 4) + 2 * align256(L.n3 * 4);
    size_t bwd = 0;
    if (d->flags & VPX_FLAG_SAVE_FOR_BWD)
        bwd = align256(L.wpk * 4) + align256(L.slabs * 4) + align256(L.colsum * 4) + align256(L.det * 4) +
              4 * align256(L.n_h * 4) +                                  // carry, dprev, dh_tmp, zero state
              align256(T * L.n3 * 4) + align256(T * L.n_x * 4) +         // d(input projection) and the flow branch's dx, all steps
              align256(L.n3 * 4) + 2 * align256(L.n_warp * 4) + align256(L.n_fl * 4) + 2 * align256(L.n_f1 * 4) +
              align256(L.w_max * 4) + align256(L.b_max * 4);
    return (fwd > bwd ? fwd : bwd) + 512;
}

int vpx_trajgru_seq_fwd(const vpx_trajgru_desc* d, const float* x, const float* h0, const float* const* params, float* hs, void* reserve,
                        size_t reserve_bytes, void* workspace, size_t workspace_bytes, void* stream_) {
    int rc = tg_check(d, "vpx_trajgru_seq_fwd");
    if (rc != VPX_OK) return rc;
    if ((!x && !h0) || !params || !hs) { set_error("vpx_trajgru_seq_fwd: NULL tensor argument (x and h0 must not both be NULL)"); return VPX_ERR_ARG; }
    for (int i = 0; i < 10; ++i) if (!params[i]) { set_error("vpx_trajgru_seq_fwd: parameter %d is NULL", i); return VPX_ERR_ARG; }
    const bool save = (d->flags & VPX_FLAG_SAVE_FOR_BWD) != 0;
    if (save && (!reserve || reserve_bytes < vpx_trajgru_reserve_bytes(d))) { set_error("vpx_trajgru_seq_fwd: reserve too small"); return VPX_ERR_WORKSPACE; }
    if (!workspace || workspace_bytes < vpx_trajgru_workspace_bytes(d)) { set_error("vpx_trajgru_seq_fwd: workspace too small"); return VPX_ERR_WORKSPACE; }
    const float *i2h_w = params[0], *i2h_b = params[1], *i2f_w = params[2], *i2f_b = params[3], *h2f_w = params[4], *h2f_b = params[5],
                *fl_w = params[6], *fl_b = params[7], *ret_w = params[8], *ret_b = params[9];
    hipStream_t stream = (hipStream_t)stream_;
    const TGLayout L = tg_layout(d);
    const int B = d->B, T = d->T, C = d->C, Cin = d->Cin, H = d->H, W = d->W, HW = H * W, k = d->k_i2h, F = TG_F, prec = d->precision;
    Carver ws(workspace, workspace_bytes);
    float* wpk = ws.take(L.wpk);
    float* i2h_all = ws.take((size_t)T * L.n3);
    float* warped = ws.take(L.n_warp);
    float* zero_h = ws.take(L.n_h);
    float* fl1 = ws.take(L.n_fl);
    float* f11 = ws.take(L.n_f1);
    float* h2h1 = ws.take(L.n3);
    VPX_CHECK_CARVE(ws, "vpx_trajgru_seq_fwd");
    float *flows = nullptr, *f1 = nullptr, *h2h = nullptr, *gsave = nullptr;
    if (save) {
        Carver rs(reserve, reserve_bytes);
        flows = rs.take((size_t)T * L.n_fl); f1 = rs.take((size_t)T * L.n_f1); h2h = rs.take((size_t)T * L.n3); gsave = rs.take((size_t)T * L.n3);
        VPX_CHECK_CARVE(rs, "vpx_trajgru_seq_fwd (reserve)");
    }
    if (x && (rc = tg_conv(stream, prec, T * B, H, W, x, i2h_w, i2h_b, i2h_all, Cin, 3 * C, k, false, 0.f, wpk))) return rc;   // (:171-173) all frames, one launch
    if (!h0) VPX_CHECK_HIP(vpx_memset_async(zero_h, 0, L.n_h * 4, stream));
    for (int t = 0; t < T; ++t) {
        const float* prev = t == 0 ? (h0 ? h0 : zero_h) : hs + (size_t)(t - 1) * L.n_h;
        float* fl_t = save ? flows + (size_t)t * L.n_fl : fl1;
        float* f1_t = save ? f1 + (size_t)t * L.n_f1 : f11;
        float* h2h_t = save ? h2h + (size_t)t * L.n3 : h2h1;
        // flow generator (:134-146): f1 = leaky(i2f(x_t) + h2f(h_{t-1})), flows = conv5x5(f1)
        if ((rc = tg_conv(stream, prec, B, H, W, prev, h2f_w, h2f_b, f1_t, C, F, 5, false, x ? 0.f : d->slope, wpk))) return rc;
        if (x && (rc = tg_conv(stream, prec, B, H, W, x + (size_t)t * L.n_x, i2f_w, i2f_b, f1_t, Cin, F, 5, true, d->slope, wpk))) return rc;
        if ((rc = tg_conv(stream, prec, B, H, W, f1_t, fl_w, fl_b, fl_t, F, 2 * d->L, 5, false, 0.f, wpk))) return rc;
        // L warps of h_{t-1} along -flow (:148-162, :184-187), then the 1x1 `ret` convolution (:188)
        if ((rc = tg_warp_fwd(prev, fl_t, warped, B, H, W, C, d->L, stream))) return rc;
        if ((rc = tg_conv(stream, prec, B, H, W, warped, ret_w, ret_b, h2h_t, d->L * C, 3 * C, 1, false, 0.f, wpk))) return rc;
        // gates + state update (:190-203)
        if ((rc = tg_gates_fwd(x ? i2h_all + (size_t)t * L.n3 : nullptr, (long long)HW * 3 * C, h2h_t, prev, hs + (size_t)t * L.n_h,
                               save ? gsave + (size_t)t * L.n3 : nullptr, B, HW, C, 0, d->slope, stream))) return rc;
    }
    return VPX_OK;
}

int vpx_trajgru_seq_bwd(const vpx_trajgru_desc* d, const float* x, const float* h0, const float* const* params, const float* hs,
                        const void* reserve, size_t reserve_bytes, const float* dout, const float* dhT, float* dx, float* dh0,
                        float* const* dparams, void* workspace, size_t workspace_bytes, void* stream_) {
    int rc = tg_check(d, "vpx_trajgru_seq_bwd");
    if (rc != VPX_OK) return rc;
    if (!(d->flags & VPX_FLAG_SAVE_FOR_BWD)) { set_error("vpx_trajgru_seq_bwd: desc lacks VPX_FLAG_SAVE_FOR_BWD"); return VPX_ERR_ARG; }
    if ((!x && !h0) || !params || !hs || !reserve || !dparams) { set_error("vpx_trajgru_seq_bwd: NULL tensor argument"); return VPX_ERR_ARG; }
    for (int i = 0; i < 10; ++i) {
        if (!params[i]) { set_error("vpx_trajgru_seq_bwd: parameter %d is NULL", i); return VPX_ERR_ARG; }
        if (!dparams[i] && (x || i >= 4)) { set_error("vpx_trajgru_seq_bwd: parameter gradient %d is NULL", i); return VPX_ERR_ARG; }
    }
    if (dx && !x) { set_error("vpx_trajgru_seq_bwd: dx requested but x is NULL"); return VPX_ERR_ARG; }
    if (reserve_bytes < vpx_trajgru_reserve_bytes(d)) { set_error("vpx_trajgru_seq_bwd: reserve too small"); return VPX_ERR_WORKSPACE; }
    if (!workspace || workspace_bytes < vpx_trajgru_workspace_bytes(d)) { set_error("vpx_trajgru_seq_bwd: workspace too small"); return VPX_ERR_WORKSPACE; }
    const float *i2h_w = params[0], *i2f_w = params[2], *h2f_w = params[4], *fl_w = params[6], *ret_w = params[8];
    float *d_i2h_w = dparams[0], *d_i2h_b = dparams[1], *d_i2f_w = dparams[2], *d_i2f_b = dparams[3], *d_h2f_w = dparams[4], *d_h2f_b = dparams[5],
          *d_fl_w = dparams[6], *d_fl_b = dparams[7], *d_ret_w = dparams[8], *d_ret_b = dparams[9];
    hipStream_t stream = (hipStream_t)stream_;
    const TGLayout L = tg_layout(d);
    const int B = d->B, T = d->T, C = d->C, Cin = d->Cin, H = d->H, W = d->W, HW = H * W, k = d->k_i2h, F = TG_F, prec = d->precision, LC = d->L * C, L2 = 2 * d->L;
    const float slope = d->slope;
    Carver rs(const_cast<void*>(reserve), reserve_bytes);
    const float* flows = rs.take((size_t)T * L.n_fl);
    const float* f1 = rs.take((size_t)T * L.n_f1);
    const float* h2h = rs.take((size_t)T * L.n3);
    const float* gsave = rs.take((size_t)T * L.n3);
    VPX_CHECK_CARVE(rs, "vpx_trajgru_seq_bwd (reserve)");
    Carver ws(workspace, workspace_bytes);
    float* wpk = ws.take(L.wpk);
    float* slabs = ws.take(L.slabs);
    float* cpart = ws.take(L.colsum);
    float* det_ws = ws.take(L.det);
    float* carry = ws.take(L.n_h);
    float* dprev = ws.take(L.n_h);
    float* dh_tmp = ws.take(L.n_h);
    float* zero_h = ws.take(L.n_h);
    float* di2h_all = ws.take((size_t)T * L.n3);
    float* dx_i2f = ws.take((size_t)T * L.n_x);
    float* dh2h = ws.take(L.n3);
    float* dwarped = ws.take(L.n_warp);
    float* warped = ws.take(L.n_warp);
    float* dflows = ws.take(L.n_fl);
    float* df1 = ws.take(L.n_f1);
    float* df1s = ws.take(L.n_f1);
    float* t_w = ws.take(L.w_max);
    float* t_b = ws.take(L.b_max);
    VPX_CHECK_CARVE(ws, "vpx_trajgru_seq_bwd");
    const bool det = vpx::g_deterministic != 0;
    auto axpy = [&](float* y, const float* v, size_t n) -> int { VPX_CHECK_HIP(launch_axpy(y, v, (long long)n, stream)); return VPX_OK; };
    auto zero = [&](float* p, size_t n) -> int { VPX_CHECK_HIP(vpx_memset_async(p, 0, n * 4, stream)); return VPX_OK; };
    // parameter gradients accumulate over the steps; the flow features' bias sum is the gradient of BOTH i2f_conv1.bias and h2f_conv1.bias
    const size_t n_ret = (size_t)3 * C * LC, n_fl_w = (size_t)L2 * F * 25, n_h2f = (size_t)F * C * 25, n_i2f = (size_t)F * Cin * 25;
    if ((rc = zero(d_ret_w, n_ret)) || (rc = zero(d_ret_b, 3 * C)) || (rc = zero(d_fl_w, n_fl_w)) || (rc = zero(d_fl_b, L2)) ||
        (rc = zero(d_h2f_w, n_h2f)) || (rc = zero(d_h2f_b, F))) return rc;
    if (x && (rc = zero(d_i2f_w, n_i2f))) return rc;
    if (!h0 && (rc = zero(zero_h, L.n_h))) return rc;
    if (dhT) VPX_CHECK_HIP(vpx_memcpy_async(carry, dhT, L.n_h * 4, hipMemcpyDeviceToDevice, stream));
    else if ((rc = zero(carry, L.n_h))) return rc;
    for (int t = T - 1; t >= 0; --t) {
        const float* prev = t == 0 ? (h0 ? h0 : zero_h) : hs + (size_t)(t - 1) * L.n_h;
        const float* fl_t = flows + (size_t)t * L.n_fl;
        const float* f1_t = f1 + (size_t)t * L.n_f1;
        if (dout && (rc = axpy(carry, dout + (size_t)t * L.n_h, L.n_h))) return rc;   // total gradient of h_t
        if ((rc = tg_gates_bwd(carry, h2h + (size_t)t * L.n3, prev, gsave + (size_t)t * L.n3, x ? di2h_all + (size_t)t * L.n3 : nullptr,
                               (long long)HW * 3 * C, dh2h, dprev, B, HW, C, 0, slope, stream))) return rc;
        // `ret` (1x1) backward needs the warped operand again: recomputed (a streaming kernel) instead of stored for all t
        if ((rc = tg_warp_fwd(prev, fl_t, warped, B, H, W, C, d->L, stream))) return rc;
        if ((rc = tg_conv_bwd(stream, prec, B, H, W, warped, ret_w, dh2h, dwarped, t_w, t_b, LC, 3 * C, 1, wpk, slabs, cpart))) return rc;
        if ((rc = axpy(d_ret_w, t_w, n_ret)) || (rc = axpy(d_ret_b, t_b, 3 * C))) return rc;
        if (det) rc = tg_warp_bwd_det(prev, fl_t, dwarped, dprev, dflows, B, H, W, C, d->L, det_ws, L.det * 4, stream);
        else rc = tg_warp_bwd(prev, fl_t, dwarped, dprev, dflows, B, H, W, C, d->L, stream);
        if (rc) return rc;
        if ((rc = tg_conv_bwd(stream, prec, B, H, W, f1_t, fl_w, dflows, df1, t_w, t_b, F, L2, 5, wpk, slabs, cpart))) return rc;
        if ((rc = axpy(d_fl_w, t_w, n_fl_w)) || (rc = axpy(d_fl_b, t_b, L2))) return rc;
        VPX_CHECK_HIP(launch_colsum(df1, f1_t, slope, df1s, t_b, cpart, (long long)B * HW, F, stream));   // LeakyReLU' and the bias sum, one pass
        if ((rc = axpy(d_h2f_b, t_b, F))) return rc;
        if ((rc = tg_conv_bwd(stream, prec, B, H, W, prev, h2f_w, df1s, dh_tmp, t_w, nullptr, C, F, 5, wpk, slabs, cpart))) return rc;
        if ((rc = axpy(d_h2f_w, t_w, n_h2f)) || (rc = axpy(dprev, dh_tmp, L.n_h))) return rc;
        if (x) {
            if ((rc = tg_conv_bwd(stream, prec, B, H, W, x + (size_t)t * L.n_x, i2f_w, df1s, dx_i2f + (size_t)t * L.n_x, t_w, nullptr, Cin, F, 5,
                                  wpk, slabs, cpart))) return rc;
            if ((rc = axpy(d_i2f_w, t_w, n_i2f))) return rc;
        }
        float* sw = carry; carry = dprev; dprev = sw;   // gradient of h_{t-1} through this step
    }
    if (x) {
        VPX_CHECK_HIP(vpx_memcpy_async(d_i2f_b, d_h2f_b, (size_t)F * 4, hipMemcpyDeviceToDevice, stream));
        if ((rc = tg_conv_bwd(stream, prec, T * B, H, W, x, i2h_w, di2h_all, dx, d_i2h_w, d_i2h_b, Cin, 3 * C, k, wpk, slabs, cpart))) return rc;
        if (dx && (rc = axpy(dx, dx_i2f, (size_t)T * L.n_x))) return rc;
    }
    if (dh0) VPX_CHECK_HIP(vpx_memcpy_async(dh0, carry, L.n_h * 4, hipMemcpyDeviceToDevice, stream));
    return VPX_OK;
}

}  // extern "C"
