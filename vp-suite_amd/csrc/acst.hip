// acst.hip — pointwise half of the action-conditional ST-LSTM cell (vp_suite/model_blocks/predrnn.py:139-169) as HIP kernels,
// forward and backward, NHWC: the conv_h(h) * conv_a(a) product (:144), both gate groups, the state updates, the output gate.
// The six convolutions of the cell run on the implicit-GEMM kernel; these kernels replace the ~40 elementwise ATen ops
// (and their autograd nodes) between them. HBM-bound streaming kernels, one thread per (pixel, channel).
#include "vpx_host.h"

namespace vpx {

struct AcstGateArgs {
    long long npix; int Ch; float forget_bias;
    const float* xc;   // [npix][7Ch]  (i, f, g, i', f', g', o)
    const float* hc;   // [npix][4Ch]  (i, f, g, o)
    const float* ac;   // [npix][4Ch]  multiplies hc elementwise, or null (plain ST-LSTM arithmetic on conv outputs)
    const float* mc;   // [npix][3Ch]  (i', f', g')
    const float* c; const float* m;
    float* c_new; float* m_new; float* delta_c; float* delta_m; float* o_pre;
    float* mem;        // [npix][2Ch] = (c_new | m_new): operand of conv_o / conv_last
    float* save;       // [npix][6Ch] post-activation (i, f, g, i', f', g') or null
};

__global__ __launch_bounds__(256) void acst_gates_fwd_kernel(const AcstGateArgs a) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const int Ch = a.Ch;
    if (e >= a.npix * Ch) return;
    const long long pix = e / Ch;
    const int ch = (int)(e - pix * Ch);
    const float* xc = a.xc + pix * 7 * Ch + ch;
    const float* hc = a.hc + pix * 4 * Ch + ch;
    const float* mc = a.mc + pix * 3 * Ch + ch;
    float h0 = hc[0], h1 = hc[Ch], h2 = hc[2 * Ch], h3 = hc[3 * Ch];
    if (a.ac) {
        const float* ac = a.ac + pix * 4 * Ch + ch;
        h0 *= ac[0]; h1 *= ac[Ch]; h2 *= ac[2 * Ch]; h3 *= ac[3 * Ch];
    }
    const float i_ = sigmoid_f(xc[0] + h0), f_ = sigmoid_f(xc[Ch] + h1 + a.forget_bias), g_ = tanh_f(xc[2 * Ch] + h2);
    const float dc = i_ * g_, cn = f_ * a.c[e] + dc;
    const float ip = sigmoid_f(xc[3 * Ch] + mc[0]), fp = sigmoid_f(xc[4 * Ch] + mc[Ch] + a.forget_bias), gp = tanh_f(xc[5 * Ch] + mc[2 * Ch]);
    const float dm = ip * gp, mn = fp * a.m[e] + dm;
    a.c_new[e] = cn; a.m_new[e] = mn; a.delta_c[e] = dc; a.delta_m[e] = dm;
    a.o_pre[e] = xc[6 * Ch] + h3;
    a.mem[pix * 2 * Ch + ch] = cn;
    a.mem[pix * 2 * Ch + Ch + ch] = mn;
    if (a.save) {
        float* s = a.save + pix * 6 * Ch + ch;
        s[0] = i_; s[Ch] = f_; s[2 * Ch] = g_; s[3 * Ch] = ip; s[4 * Ch] = fp; s[5 * Ch] = gp;
    }
}

struct AcstGateBwdArgs {
    long long npix; int Ch;
    const float* hc; const float* ac; const float* c; const float* m; const float* save;
    const float* d_cn; const float* d_mn; const float* d_dc; const float* d_dm; const float* d_opre; const float* d_mem;  // any may be null
    float* dxc; float* dhc; float* dac; float* dmc; float* dc; float* dm;   // dac null iff ac null
};

__global__ __launch_bounds__(256) void acst_gates_bwd_kernel(const AcstGateBwdArgs a) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const int Ch = a.Ch;
    if (e >= a.npix * Ch) return;
    const long long pix = e / Ch;
    const int ch = (int)(e - pix * Ch);
    const float* s = a.save + pix * 6 * Ch + ch;
    const float i_ = s[0], f_ = s[Ch], g_ = s[2 * Ch], ip = s[3 * Ch], fp = s[4 * Ch], gp = s[5 * Ch];
    float Dcn = a.d_cn ? a.d_cn[e] : 0.f, Dmn = a.d_mn ? a.d_mn[e] : 0.f;
    if (a.d_mem) { Dcn += a.d_mem[pix * 2 * Ch + ch]; Dmn += a.d_mem[pix * 2 * Ch + Ch + ch]; }
    const float Ddc = Dcn + (a.d_dc ? a.d_dc[e] : 0.f), Ddm = Dmn + (a.d_dm ? a.d_dm[e] : 0.f);
    const float da_i = Ddc * g_ * i_ * (1.f - i_), da_f = Dcn * a.c[e] * f_ * (1.f - f_), da_g = Ddc * i_ * (1.f - g_ * g_);
    const float da_ip = Ddm * gp * ip * (1.f - ip), da_fp = Dmn * a.m[e] * fp * (1.f - fp), da_gp = Ddm * ip * (1.f - gp * gp);
    const float da_o = a.d_opre ? a.d_opre[e] : 0.f;
    a.dc[e] = Dcn * f_;
    a.dm[e] = Dmn * fp;
    float* dx = a.dxc + pix * 7 * Ch + ch;
    dx[0] = da_i; dx[Ch] = da_f; dx[2 * Ch] = da_g; dx[3 * Ch] = da_ip; dx[4 * Ch] = da_fp; dx[5 * Ch] = da_gp; dx[6 * Ch] = da_o;
    float* dmc = a.dmc + pix * 3 * Ch + ch;
    dmc[0] = da_ip; dmc[Ch] = da_fp; dmc[2 * Ch] = da_gp;
    float* dh = a.dhc + pix * 4 * Ch + ch;
    if (a.ac) {
        const float* hc = a.hc + pix * 4 * Ch + ch;
        const float* ac = a.ac + pix * 4 * Ch + ch;
        float* da = a.dac + pix * 4 * Ch + ch;
        dh[0] = da_i * ac[0]; dh[Ch] = da_f * ac[Ch]; dh[2 * Ch] = da_g * ac[2 * Ch]; dh[3 * Ch] = da_o * ac[3 * Ch];
        da[0] = da_i * hc[0]; da[Ch] = da_f * hc[Ch]; da[2 * Ch] = da_g * hc[2 * Ch]; da[3 * Ch] = da_o * hc[3 * Ch];
    } else {
        dh[0] = da_i; dh[Ch] = da_f; dh[2 * Ch] = da_g; dh[3 * Ch] = da_o;
    }
}

// h_new = sigmoid(o_pre + oc) * tanh(lc) (predrnn.py:166-167) and its backward
__global__ __launch_bounds__(256) void st_out_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ o, const float* __restrict__ tl,
                                                         float* __restrict__ d_o, float* __restrict__ d_lc, long long n) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const float g = dh[e], ov = o[e], t = tl[e];
    d_o[e] = g * t * ov * (1.f - ov);      // gradient of the sigmoid's argument: goes to o_pre and to conv_o's output alike
    d_lc[e] = g * ov * (1.f - t * t);
}

}  // namespace vpx

using namespace vpx;

extern "C" {

int vpx_acst_gates_fwd(const float* xc, const float* hc, const float* ac, const float* mc, const float* c, const float* m,
                       float* c_new, float* m_new, float* delta_c, float* delta_m, float* o_pre, float* mem, float* save,
                       long long npix, int Ch, float forget_bias, void* stream) {
    if (!xc || !hc || !mc || !c || !m || !c_new || !m_new || !delta_c || !delta_m || !o_pre || !mem || npix < 1 || Ch < 1) {
        set_error("vpx_acst_gates_fwd: bad argument");
        return VPX_ERR_ARG;
    }
    AcstGateArgs a{npix, Ch, forget_bias, xc, hc, ac, mc, c, m, c_new, m_new, delta_c, delta_m, o_pre, mem, save};
    VPX_LAUNCH(acst_gates_fwd_kernel, dim3((unsigned)((npix * Ch + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    VPX_CHECK_HIP(vpx_hip_last_error());
    return VPX_OK;
}

int vpx_acst_gates_bwd(const float* hc, const float* ac, const float* c, const float* m, const float* save, const float* d_cn,
                       const float* d_mn, const float* d_dc, const float* d_dm, const float* d_opre, const float* d_mem, float* dxc,
                       float* dhc, float* dac, float* dmc, float* dc, float* dm, long long npix, int Ch, void* stream) {
    if (!hc || !c || !m || !save || !dxc || !dhc || !dmc || !dc || !dm || npix < 1 || Ch < 1 || ((ac != nullptr) != (dac != nullptr))) {
        set_error("vpx_acst_gates_bwd: bad argument");
        return VPX_ERR_ARG;
    }
    AcstGateBwdArgs a{npix, Ch, hc, ac, c, m, save, d_cn, d_mn, d_dc, d_dm, d_opre, d_mem, dxc, dhc, dac, dmc, dc, dm};
    VPX_LAUNCH(acst_gates_bwd_kernel, dim3((unsigned)((npix * Ch + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    VPX_CHECK_HIP(vpx_hip_last_error());
    return VPX_OK;
}

int vpx_st_out_fwd(const float* o_pre, const float* oc, const float* lc, float* h_new, float* o_save, float* tl_save, long long n,
                   void* stream) {
    if (!o_pre || !lc || !h_new || n < 1 || ((o_save != nullptr) != (tl_save != nullptr))) { set_error("vpx_st_out_fwd: bad argument"); return VPX_ERR_ARG; }
    VPX_CHECK_HIP(launch_st_ln_out(o_pre, oc, lc, h_new, o_save, tl_save, n, (hipStream_t)stream));
    return VPX_OK;
}

int vpx_st_out_bwd(const float* dh, const float* o, const float* tl, float* d_o, float* d_lc, long long n, void* stream) {
    if (!dh || !o || !tl || !d_o || !d_lc || n < 1) { set_error("vpx_st_out_bwd: bad argument"); return VPX_ERR_ARG; }
    VPX_LAUNCH(st_out_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dh, o, tl, d_o, d_lc, n);
    VPX_CHECK_HIP(vpx_hip_last_error());
    return VPX_OK;
}

}  // extern "C"
