// acst.hip — the action-conditional ST-LSTM cell (vp_suite/model_blocks/predrnn.py:86-169) as ONE library call each way
// (vpx_acstlstm_step_fwd / _bwd, round 6; until then the Python block composed the stages below itself): six biased convolutions on the
// implicit-GEMM kernel, optional LayerNorms (layernorm.hip), and the pointwise half as HIP kernels, forward and backward, NHWC: the
// conv_h(h) * conv_a(a) product (:144), both gate groups, the state updates, the output gate — these replace the ~40 elementwise ATen ops
// (and their autograd nodes) between the convolutions. HBM-bound streaming kernels, one thread per (pixel, channel).
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

// ---- the whole cell step (round 6): six biased convolutions (+ LayerNorms) + the two pointwise stages behind ONE entry point each way ----
struct ACS {   // sizes
    size_t n_state, HW;
    int B, Cin, Ch, H, W, k;
    size_t wpk_max, slab_floats;
};
static ACS acs_sizes(const vpx_acstlstm_desc* d) {
    ACS s{};
    s.B = d->B; s.Cin = d->Cin; s.Ch = d->Ch; s.H = d->H; s.W = d->W; s.k = d->k;
    s.HW = (size_t)d->H * d->W;
    s.n_state = (size_t)d->B * s.HW * d->Ch;
    const int Ch = d->Ch, Cin = d->Cin, k = d->k;
    size_t m = 0;
    auto mx = [&](size_t v) { if (v > m) m = v; };
    // forward layers and their adjoints (data gradients): (C -> Co, k)
    mx(plain_conv_wpk_floats(Cin, 7 * Ch, k, k)); mx(plain_conv_wpk_floats(7 * Ch, Cin, k, k));
    mx(plain_conv_wpk_floats(Ch, 4 * Ch, k, k)); mx(plain_conv_wpk_floats(4 * Ch, Ch, k, k));
    mx(plain_conv_wpk_floats(Ch, 3 * Ch, k, k)); mx(plain_conv_wpk_floats(3 * Ch, Ch, k, k));
    mx(plain_conv_wpk_floats(2 * Ch, Ch, k, k)); mx(plain_conv_wpk_floats(Ch, 2 * Ch, k, k));
    mx(plain_conv_wpk_floats(2 * Ch, Ch, 1, 1)); mx(plain_conv_wpk_floats(Ch, 2 * Ch, 1, 1));
    s.wpk_max = m;
    size_t w = (size_t)7 * Ch * Cin * k * k;
    if ((size_t)4 * Ch * Ch * k * k > w) w = (size_t)4 * Ch * Ch * k * k;
    if ((size_t)2 * Ch * Ch * k * k > w) w = (size_t)2 * Ch * Ch * k * k;
    s.slab_floats = w * wgrad_slices(d->B, d->H, d->W);
    return s;
}
static const char* acs_check(const vpx_acstlstm_desc* d) {
    if (!d) return "NULL descriptor";
    if (d->B < 1 || d->Cin < 1 || d->Ch < 1 || d->H < 1 || d->W < 1) return "sizes must be positive";
    if (d->k < 1 || !(d->k & 1) || d->k > 7) return "the five k x k convolutions need an odd kernel size up to 7";
    if (d->precision < VPX_PREC_F32 || d->precision > VPX_PREC_BF16) return "unknown precision";
    return nullptr;
}
constexpr int ACS_LN_MULT[5] = {7, 4, 4, 3, 1};   // channels / Ch of conv_x, conv_h, conv_a, conv_m, conv_o
constexpr int ACS_LN_PLANES = 19;

struct ACSReserve { float *hc, *ac, *save, *mem, *o, *tl, *xhat[5], *st[5]; };
static ACSReserve acs_carve_reserve(void* reserve, const ACS& s, bool ln) {
    char* r = (char*)reserve;
    const size_t plane = align256(s.n_state * 4);
    ACSReserve R{};
    auto take = [&](int planes) { float* p = (float*)r; r += planes * plane; return p; };
    R.hc = take(4); R.ac = take(4); R.save = take(6); R.mem = take(2); R.o = take(1); R.tl = take(1);
    if (ln) {
        for (int i = 0; i < 5; ++i) R.xhat[i] = take(ACS_LN_MULT[i]);
        for (int i = 0; i < 5; ++i) R.st[i] = (float*)r + 2 * s.B * i;
    }
    return R;
}

}  // namespace vpx

using namespace vpx;

extern "C" {

size_t vpx_acstlstm_reserve_bytes(const vpx_acstlstm_desc* d) {
    if (acs_check(d) || !(d->flags & VPX_FLAG_SAVE_FOR_BWD)) return 0;
    const ACS s = acs_sizes(d);
    return (size_t)(18 + (d->layer_norm ? ACS_LN_PLANES : 0)) * align256(s.n_state * 4) + align256((size_t)10 * d->B * 4) + 256;
}

size_t vpx_acstlstm_workspace_bytes(const vpx_acstlstm_desc* d) {
    if (const char* e = acs_check(d)) { set_error("vpx_acstlstm_workspace_bytes: %s", e); return 0; }
    const ACS s = acs_sizes(d);
    const size_t plane = align256(s.n_state * 4), per_sample = align256(s.HW * d->Ch * 4);
    const size_t ln_fwd = d->layer_norm ? 2 * ACS_LN_PLANES * per_sample : 0;
    // forward: xc(7) hc(4) ac(4) mc(3) o_pre oc lc mem(2) save-less gates; LayerNorm parameters transposed (gamma, beta)
    const size_t fwd = 23 * plane + ln_fwd;
    // backward: d_o d_lc du_o dmem(2) dxc(7) dhc(4) dac(4) dmc(3) dm + LayerNorm: du_x(7) du_h(4) du_a(4) du_m(3); slabs; bias partials;
    // LayerNorm parameters (gamma: 19) and their gradients (2 x 19)
    const size_t bwd = (size_t)(24 + (d->layer_norm ? ACS_LN_PLANES : 0)) * plane + align256(s.slab_floats * 4) +
                       align256((size_t)COLSUM_BLOCKS * 7 * d->Ch * 4) + (d->layer_norm ? 3 * ACS_LN_PLANES * per_sample : 0);
    return (fwd > bwd ? fwd : bwd) + align256(s.wpk_max * 4) + align256((size_t)d->B * 64 * 2 * 8) + align256((size_t)10 * d->B * 4) + 64 * 256;
}

// params: (conv_x, conv_h, conv_a, conv_m, conv_o, conv_last) x (weight OIHW, bias); ln: (x, h, a, m, o) x (gamma, beta), [C,H,W] each
int vpx_acstlstm_step_fwd(const vpx_acstlstm_desc* d, const float* x, const float* h, const float* c, const float* m, const float* act,
                          const float* const* params, const float* const* ln, float* h_new, float* c_new, float* m_new, float* delta_c,
                          float* delta_m, void* reserve, size_t reserve_bytes, void* workspace, size_t workspace_bytes, void* stream_) {
    if (const char* e = acs_check(d)) { set_error("vpx_acstlstm_step_fwd: %s", e); return VPX_ERR_ARG; }
    if (!x || !h || !c || !m || !act || !params || !h_new || !c_new || !m_new || !delta_c || !delta_m) { set_error("vpx_acstlstm_step_fwd: NULL tensor argument"); return VPX_ERR_ARG; }
    for (int i = 0; i < 12; ++i) if (!params[i]) { set_error("vpx_acstlstm_step_fwd: parameter %d is NULL (every convolution of the cell has a bias)", i); return VPX_ERR_ARG; }
    const bool use_ln = d->layer_norm != 0, save = (d->flags & VPX_FLAG_SAVE_FOR_BWD) != 0;
    if (use_ln) {
        if (!ln) { set_error("vpx_acstlstm_step_fwd: layer_norm set but ln is NULL"); return VPX_ERR_ARG; }
        for (int i = 0; i < 10; ++i) if (!ln[i]) { set_error("vpx_acstlstm_step_fwd: LayerNorm parameter %d is NULL", i); return VPX_ERR_ARG; }
    }
    if (save && (!reserve || reserve_bytes < vpx_acstlstm_reserve_bytes(d))) { set_error("vpx_acstlstm_step_fwd: reserve too small"); return VPX_ERR_WORKSPACE; }
    if (!workspace || workspace_bytes < vpx_acstlstm_workspace_bytes(d)) { set_error("vpx_acstlstm_step_fwd: workspace too small"); return VPX_ERR_WORKSPACE; }
    hipStream_t stream = (hipStream_t)stream_;
    const ACS s = acs_sizes(d);
    const int B = s.B, Cin = s.Cin, Ch = s.Ch, k = s.k, prec = d->precision;
    const ConvGeo g{B, s.H, s.W};
    ACSReserve R{};
    if (save) R = acs_carve_reserve(reserve, s, use_ln);
    Carver ws(workspace, workspace_bytes);
    float* wpk = ws.take(s.wpk_max);
    double* partial = (double*)ws.take((size_t)B * 64 * 2 * 2);
    float* st_tmp = ws.take((size_t)10 * B);
    float* xc = ws.take(7 * s.n_state);
    float* mc = ws.take(3 * s.n_state);
    float* hc = save ? R.hc : ws.take(4 * s.n_state);
    float* ac = save ? R.ac : ws.take(4 * s.n_state);
    float* mem = save ? R.mem : ws.take(2 * s.n_state);
    float* o_pre = ws.take(s.n_state);
    float* oc = ws.take(s.n_state);
    float* lc = ws.take(s.n_state);
    float *gam[5] = {}, *bet[5] = {};
    if (use_ln)
        for (int i = 0; i < 5; ++i) { gam[i] = ws.take(s.HW * ACS_LN_MULT[i] * Ch); bet[i] = ws.take(s.HW * ACS_LN_MULT[i] * Ch); }
    VPX_CHECK_CARVE(ws, "vpx_acstlstm_step_fwd");
    if (use_ln)
        for (int i = 0; i < 5; ++i) {
            VPX_CHECK_HIP(launch_nchw_to_nhwc(ln[2 * i], gam[i], 1, ACS_LN_MULT[i] * Ch, s.H, s.W, stream));
            VPX_CHECK_HIP(launch_nchw_to_nhwc(ln[2 * i + 1], bet[i], 1, ACS_LN_MULT[i] * Ch, s.H, s.W, stream));
        }
    const long long n1 = (long long)s.HW * Ch;
    int rc;
    // the four input convolutions, each with its bias and (optionally) its LayerNorm (predrnn.py:139-142 with :102-136)
    struct In { const float* src; int C; int mult; float* dst; int pi; } in[4] = {{x, Cin, 7, xc, 0}, {h, Ch, 4, hc, 1}, {act, Ch, 4, ac, 2}, {m, Ch, 3, mc, 3}};
    for (int i = 0; i < 4; ++i) {
        const In& L = in[i];
        if ((rc = plain_conv(stream, prec, g, L.src, L.C, L.C, params[2 * L.pi], (long long)L.C * k * k, k * k, k, k, L.mult * Ch, false, params[2 * L.pi + 1],
                             L.dst, L.mult * Ch, false, wpk))) return rc;
        if (use_ln)
            VPX_CHECK_HIP(launch_layernorm_fwd(L.dst, gam[i], bet[i], L.dst, save ? R.xhat[i] : nullptr, save ? R.st[i] : st_tmp + 2 * B * i, partial, B,
                                               L.mult * n1, stream));
    }
    // conv_h(h) * conv_a(a), both gate groups, the state updates, mem = (c_new | m_new)  (predrnn.py:143-164)
    {
        AcstGateArgs a{(long long)((size_t)B * s.HW), Ch, d->forget_bias, xc, hc, ac, mc, c, m, c_new, m_new, delta_c, delta_m, o_pre, mem, save ? R.save : nullptr};
        VPX_LAUNCH(acst_gates_fwd_kernel, dim3((unsigned)((a.npix * Ch + 255) / 256)), dim3(256), 0, stream, a);
        VPX_CHECK_HIP(vpx_hip_last_error());
    }
    // conv_o(mem) (+ LayerNorm), conv_last(mem), output gate  (predrnn.py:165-167)
    if ((rc = plain_conv(stream, prec, g, mem, 2 * Ch, 2 * Ch, params[8], (long long)2 * Ch * k * k, k * k, k, k, Ch, false, params[9], oc, Ch, false, wpk))) return rc;
    if (use_ln)
        VPX_CHECK_HIP(launch_layernorm_fwd(oc, gam[4], bet[4], oc, save ? R.xhat[4] : nullptr, save ? R.st[4] : st_tmp + 8 * B, partial, B, n1, stream));
    if ((rc = plain_conv(stream, prec, g, mem, 2 * Ch, 2 * Ch, params[10], (long long)2 * Ch, 1, 1, 1, Ch, false, params[11], lc, Ch, false, wpk))) return rc;
    VPX_CHECK_HIP(launch_st_ln_out(o_pre, oc, lc, h_new, save ? R.o : nullptr, save ? R.tl : nullptr, (long long)s.n_state, stream));
    return VPX_OK;
}

int vpx_acstlstm_step_bwd(const vpx_acstlstm_desc* d, const float* x, const float* h, const float* c, const float* m, const float* act,
                          const float* const* params, const float* const* ln, const void* reserve, size_t reserve_bytes, const float* dh_new,
                          const float* dc_new, const float* dm_new, const float* ddc, const float* ddm, float* dx, float* dh, float* dc, float* dm,
                          float* dact, float* const* dparams, float* const* dln, void* workspace, size_t workspace_bytes, void* stream_) {
    if (const char* e = acs_check(d)) { set_error("vpx_acstlstm_step_bwd: %s", e); return VPX_ERR_ARG; }
    if (!(d->flags & VPX_FLAG_SAVE_FOR_BWD)) { set_error("vpx_acstlstm_step_bwd: the forward must have run with VPX_FLAG_SAVE_FOR_BWD"); return VPX_ERR_ARG; }
    if (!x || !h || !c || !m || !act || !params || !dh_new) { set_error("vpx_acstlstm_step_bwd: NULL tensor argument"); return VPX_ERR_ARG; }
    for (int i = 0; i < 12; i += 2) if (!params[i]) { set_error("vpx_acstlstm_step_bwd: weight %d is NULL", i / 2); return VPX_ERR_ARG; }
    const bool use_ln = d->layer_norm != 0;
    if (use_ln) {
        if (!ln) { set_error("vpx_acstlstm_step_bwd: layer_norm set but ln is NULL"); return VPX_ERR_ARG; }
        for (int i = 0; i < 10; i += 2) if (!ln[i]) { set_error("vpx_acstlstm_step_bwd: LayerNorm weight %d is NULL", i / 2); return VPX_ERR_ARG; }
    }
    if (!reserve || reserve_bytes < vpx_acstlstm_reserve_bytes(d)) { set_error("vpx_acstlstm_step_bwd: reserve too small"); return VPX_ERR_WORKSPACE; }
    if (!workspace || workspace_bytes < vpx_acstlstm_workspace_bytes(d)) { set_error("vpx_acstlstm_step_bwd: workspace too small"); return VPX_ERR_WORKSPACE; }
    hipStream_t stream = (hipStream_t)stream_;
    const ACS s = acs_sizes(d);
    const int B = s.B, Cin = s.Cin, Ch = s.Ch, k = s.k, prec = d->precision, HW = (int)s.HW;
    const long long npix = (long long)B * HW;
    const ConvGeo g{B, s.H, s.W};
    const ACSReserve R = acs_carve_reserve(const_cast<void*>(reserve), s, use_ln);
    Carver ws(workspace, workspace_bytes);
    float* wpk = ws.take(s.wpk_max);
    double* partial = (double*)ws.take((size_t)B * 64 * 2 * 2);
    float* sums = ws.take((size_t)10 * B);
    float* d_o = ws.take(s.n_state);
    float* d_lc = ws.take(s.n_state);
    float* dmem = ws.take(2 * s.n_state);
    float* dxc = ws.take(7 * s.n_state);
    float* dhc = ws.take(4 * s.n_state);
    float* dac = ws.take(4 * s.n_state);
    float* dmc = ws.take(3 * s.n_state);
    float* dc_s = ws.take(s.n_state);
    float* dm_s = ws.take(s.n_state);
    float* slabs = ws.take(s.slab_floats);
    float* db_part = ws.take((size_t)COLSUM_BLOCKS * 7 * Ch);
    float *du[5] = {dxc, dhc, dac, dmc, d_o}, *gam[5] = {}, *dgam[5] = {}, *dbet[5] = {};
    if (use_ln)
        for (int i = 0; i < 5; ++i) {
            du[i] = ws.take(ACS_LN_MULT[i] * s.n_state);
            gam[i] = ws.take(s.HW * ACS_LN_MULT[i] * Ch); dgam[i] = ws.take(s.HW * ACS_LN_MULT[i] * Ch); dbet[i] = ws.take(s.HW * ACS_LN_MULT[i] * Ch);
        }
    VPX_CHECK_CARVE(ws, "vpx_acstlstm_step_bwd");
    if (use_ln)
        for (int i = 0; i < 5; ++i) VPX_CHECK_HIP(launch_nchw_to_nhwc(ln[2 * i], gam[i], 1, ACS_LN_MULT[i] * Ch, s.H, s.W, stream));
    float* const* dp = dparams;
    auto want = [&](int i) { return dp && dp[i]; };
    const int blk[8] = {0, 1, 2, 3, 4, 5, 6, 0};
    int rc;
    // A: h_new = sigmoid(o_pre + oc) * tanh(lc): the gradient of the sigmoid's argument (goes to o_pre and to conv_o's output alike) and of lc
    VPX_LAUNCH(st_out_bwd_kernel, dim3((unsigned)((s.n_state + 255) / 256)), dim3(256), 0, stream, dh_new, R.o, R.tl, d_o, d_lc, (long long)s.n_state);
    VPX_CHECK_HIP(vpx_hip_last_error());
    // B: conv_o (through its LayerNorm) and conv_last back to mem = (c_new | m_new); their weight and bias gradients
    if (use_ln) VPX_CHECK_HIP(launch_layernorm_bwd(d_o, Ch, Ch, blk, R.xhat[4], R.st[4], gam[4], B, HW, Ch, partial, sums, du[4], dgam[4], dbet[4], stream));
    if ((rc = plain_conv(stream, prec, g, du[4], Ch, Ch, params[8], (long long)2 * Ch * k * k, k * k, k, k, 2 * Ch, true, nullptr, dmem, 2 * Ch, false, wpk))) return rc;
    if ((rc = plain_conv(stream, prec, g, d_lc, Ch, Ch, params[10], (long long)2 * Ch, 1, 1, 1, 2 * Ch, true, nullptr, dmem, 2 * Ch, true, wpk))) return rc;
    if (want(8) && (rc = plain_wgrad(stream, prec, g, du[4], Ch, R.mem, 2 * Ch, k, k, slabs, dp[8]))) return rc;
    if (want(9)) VPX_CHECK_HIP(launch_colsum(du[4], nullptr, 0.f, nullptr, dp[9], db_part, npix, Ch, stream));
    if (want(10) && (rc = plain_wgrad(stream, prec, g, d_lc, Ch, R.mem, 2 * Ch, 1, 1, slabs, dp[10]))) return rc;
    if (want(11)) VPX_CHECK_HIP(launch_colsum(d_lc, nullptr, 0.f, nullptr, dp[11], db_part, npix, Ch, stream));
    // C: the gate stage: gradients of the four (normalised) conv outputs, dc and the direct part of dm
    {
        AcstGateBwdArgs a{npix, Ch, R.hc, R.ac, c, m, R.save, dc_new, dm_new, ddc, ddm, d_o, dmem, dxc, dhc, dac, dmc, dc ? dc : dc_s, dm ? dm : dm_s};
        VPX_LAUNCH(acst_gates_bwd_kernel, dim3((unsigned)((npix * Ch + 255) / 256)), dim3(256), 0, stream, a);
        VPX_CHECK_HIP(vpx_hip_last_error());
    }
    // D: through the LayerNorms of conv_x / conv_h / conv_a / conv_m
    if (use_ln) {
        float* dy4[4] = {dxc, dhc, dac, dmc};
        for (int i = 0; i < 4; ++i)
            VPX_CHECK_HIP(launch_layernorm_bwd(dy4[i], ACS_LN_MULT[i] * Ch, Ch, blk, R.xhat[i], R.st[i], gam[i], B, HW, ACS_LN_MULT[i] * Ch, partial, sums, du[i],
                                               dgam[i], dbet[i], stream));
    }
    // E / F: data, weight and bias gradients of the four input convolutions (dm: the conv part adds to the gate stage's direct part)
    struct In { const float* src; int C; int mult; float* dsrc; int pi; bool acc; } in[4] = {{x, Cin, 7, dx, 0, false}, {h, Ch, 4, dh, 1, false},
                                                                                              {act, Ch, 4, dact, 2, false}, {m, Ch, 3, dm, 3, true}};
    for (int i = 0; i < 4; ++i) {
        const In& L = in[i];
        const int Co = L.mult * Ch;
        if (L.dsrc && (rc = plain_conv(stream, prec, g, du[i], Co, Co, params[2 * L.pi], (long long)L.C * k * k, k * k, k, k, L.C, true, nullptr, L.dsrc, L.C, L.acc, wpk)))
            return rc;
        if (want(2 * L.pi) && (rc = plain_wgrad(stream, prec, g, du[i], Co, L.src, L.C, k, k, slabs, dp[2 * L.pi]))) return rc;
        if (want(2 * L.pi + 1)) VPX_CHECK_HIP(launch_colsum(du[i], nullptr, 0.f, nullptr, dp[2 * L.pi + 1], db_part, npix, Co, stream));
    }
    // G: LayerNorm parameter gradients back to the reference's [C,H,W]
    if (use_ln && dln)
        for (int i = 0; i < 5; ++i) {
            if (dln[2 * i]) VPX_CHECK_HIP(launch_nhwc_to_nchw(dgam[i], dln[2 * i], 1, ACS_LN_MULT[i] * Ch, s.H, s.W, stream));
            if (dln[2 * i + 1]) VPX_CHECK_HIP(launch_nhwc_to_nchw(dbet[i], dln[2 * i + 1], 1, ACS_LN_MULT[i] * Ch, s.H, s.W, stream));
        }
    return VPX_OK;
}

}  // extern "C"
