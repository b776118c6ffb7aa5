// stlstm_ln_api.hip — LayerNorm variant of the ST-LSTM cell (predrnn.py:24-40 + 57-83): conv_x / conv_h / conv_m /
// conv_o are each followed by LayerNorm([C,H,W]) before the gates add them, so they run as separate plain convolutions
// (same implicit-GEMM kernel), LayerNorm kernels and pointwise gate kernels. Called from vpx_stlstm_step_fwd/_bwd when
// desc.layer_norm != 0. LN parameters arrive in the reference's [C,H,W] layout and are transposed to [HW,C] per call.
#include "vpx_host.h"

namespace vpx {

struct STLN {  // sizes
    size_t n_state, n_x, HW;
    int B, Cin, Ch, H, W, k;
    size_t wpk_max;      // floats
    size_t slab_floats;
    int slices;
};

static STLN stln_sizes(const vpx_stlstm_desc* d) {
    STLN s{};
    s.B = d->B; s.Cin = d->Cin; s.Ch = d->Ch; s.H = d->H; s.W = d->W; s.k = d->k;
    s.HW = (size_t)d->H * d->W;
    s.n_state = (size_t)d->B * s.HW * d->Ch;
    s.n_x = (size_t)d->B * s.HW * d->Cin;
    const int Ch = d->Ch, Cin = d->Cin, k = d->k;
    size_t m = plain_conv_wpk_floats(Cin, 7 * Ch, k, k);
    auto mx = [&](size_t v) { if (v > m) m = v; };
    mx(plain_conv_wpk_floats(Ch, 4 * Ch, k, k)); mx(plain_conv_wpk_floats(Ch, 3 * Ch, k, k));
    mx(plain_conv_wpk_floats(2 * Ch, Ch, k, k)); mx(plain_conv_wpk_floats(2 * Ch, Ch, 1, 1));
    mx(plain_conv_wpk_floats(7 * Ch, Cin, k, k)); mx(plain_conv_wpk_floats(4 * Ch, Ch, k, k));
    mx(plain_conv_wpk_floats(3 * Ch, Ch, k, k)); mx(plain_conv_wpk_floats(Ch, 2 * Ch, k, k));
    mx(plain_conv_wpk_floats(Ch, 2 * Ch, 1, 1));
    s.wpk_max = m;
    s.slices = wgrad_slices(d->B, d->H, d->W);
    size_t w = (size_t)7 * Ch * Cin * k * k;
    if ((size_t)4 * Ch * Ch * k * k > w) w = (size_t)4 * Ch * Ch * k * k;
    if ((size_t)2 * Ch * Ch * k * k > w) w = (size_t)2 * Ch * Ch * k * k;
    s.slab_floats = w * s.slices;
    return s;
}

size_t stlstm_ln_reserve_bytes(const vpx_stlstm_desc* d) {
    const STLN s = stln_sizes(d);
    // xhat_x(7) xhat_h(4) xhat_m(3) xhat_o(1) gates_c(3) gates_m(3) o tl mem(2) = 25 planes + 4 x stats[B][2]
    return 25 * align256(s.n_state * 4) + align256((size_t)8 * d->B * 4) + 256;
}

size_t stlstm_ln_workspace_bytes(const vpx_stlstm_desc* d) {
    const STLN s = stln_sizes(d);
    const size_t plane = align256(s.n_state * 4);
    // fwd: xc(7) hc(4) mc(3) o_pre oc lc (3) + LN params gamma/beta (2 x 15 planes of one sample) ; bwd: dG7(7) dlc du_o
    // dcn dmn dm(5) du_x(7) du_h(4) du_m(3) + gamma(15) dgamma(15) dbeta(15)
    const size_t per_sample = align256(s.HW * d->Ch * 4);
    size_t fwd = 19 * plane + 30 * per_sample;
    size_t bwd = 28 * plane + 60 * per_sample + align256(s.slab_floats * 4);
    size_t b = (fwd > bwd ? fwd : bwd) + 5 * align256(s.wpk_max * 4) + align256((size_t)d->B * 64 * 2 * 8) + align256((size_t)d->B * 2 * 4);
    if (d->layout == VPX_LAYOUT_NCHW) b += 2 * align256(s.n_x * 4) + 14 * plane;
    return b + 64 * 256;  // alignment slack of the individual carves
}

struct LNReserve {
    float *xhat_x, *xhat_h, *xhat_m, *xhat_o, *gates_c, *gates_m, *o, *tl, *mem, *st_x, *st_h, *st_m, *st_o;
};
static LNReserve carve_reserve(void* reserve, const STLN& s) {
    char* r = (char*)reserve;
    const size_t plane = align256(s.n_state * 4);
    LNReserve R{};
    auto take = [&](int planes) { float* p = (float*)r; r += planes * plane; return p; };
    R.xhat_x = take(7); R.xhat_h = take(4); R.xhat_m = take(3); R.xhat_o = take(1);
    R.gates_c = take(3); R.gates_m = take(3); R.o = take(1); R.tl = take(1); R.mem = take(2);
    float* st = (float*)r;
    R.st_x = st; R.st_h = st + 2 * s.B; R.st_m = st + 4 * s.B; R.st_o = st + 6 * s.B;
    return R;
}

// transposes the 8 LN parameter tensors ([C,H,W] -> [HW,C]); order x_g,x_b,h_g,h_b,m_g,m_b,o_g,o_b
static int ln_params_nhwc(const float* const* ln, float* dst[8], Carver& ws, const STLN& s, hipStream_t stream, bool cached = false) {
    const int mult[4] = {7, 4, 3, 1};
    for (int i = 0; i < 8; ++i) {
        const int C = mult[i / 2] * s.Ch;
        dst[i] = ws.take(s.HW * C);
        if (!ln[i]) { set_error("stlstm (layer_norm): LayerNorm parameter %d is NULL", i); return VPX_ERR_ARG; }
        if (!cached) VPX_CHECK_HIP(launch_nchw_to_nhwc(ln[i], dst[i], 1, C, s.H, s.W, stream));
    }
    return VPX_OK;
}

int stlstm_ln_fwd(const vpx_stlstm_desc* d, const float* x, const float* h, const float* c, const float* m,
                  const float* Wx, const float* Wh, const float* Wm, const float* Wo, const float* Wlast,
                  const float* const* ln, float* h_new, float* c_new, float* m_new, float* delta_c, float* delta_m,
                  void* reserve, Carver& ws, hipStream_t stream) {
    const STLN s = stln_sizes(d);
    const int B = s.B, Cin = s.Cin, Ch = s.Ch, k = s.k, prec = d->precision;
    const ConvGeo g{B, s.H, s.W};
    const bool save = (d->flags & VPX_FLAG_SAVE_FOR_BWD) != 0;
    LNReserve R{};
    if (save) R = carve_reserve(reserve, s);
    // VPX_FLAG_WEIGHTS_PACKED: the caller kept this workspace since a call with the same weights, LayerNorm parameters and desc —
    // the five weight packs (one region each) and the transposed LayerNorm parameters are still in it (first carves: fixed offsets)
    const bool packed = (d->flags & VPX_FLAG_WEIGHTS_PACKED) != 0;
    float* wpk5[5];
    for (auto& p : wpk5) p = ws.take(s.wpk_max);
    float* lnp[8];
    int rc;
    if ((rc = ln_params_nhwc(ln, lnp, ws, s, stream, packed))) return rc;
    double* partial = (double*)ws.take((size_t)B * 64 * 2 * 2);
    float* st_tmp = ws.take((size_t)8 * B);
    float* xc = ws.take(7 * s.n_state);
    float* hc = ws.take(4 * s.n_state);
    float* mc = ws.take(3 * s.n_state);
    float* o_pre = ws.take(s.n_state);
    float* oc = ws.take(s.n_state);
    float* lc = ws.take(s.n_state);
    float* mem_ws = ws.take(2 * s.n_state);
    VPX_CHECK_CARVE(ws, "vpx_stlstm_step_fwd (LayerNorm)");
    float* mem = save ? R.mem : mem_ws;
    const long long n1 = (long long)s.HW * Ch;
    // conv_x / conv_h / conv_m, each followed by its own LayerNorm (predrnn.py:58-60 with :24-36)
    if ((rc = plain_conv(stream, prec, g, x, Cin, Cin, Wx, (long long)Cin * k * k, k * k, k, k, 7 * Ch, false, nullptr, xc, 7 * Ch, false, wpk5[0], 0.0f, packed))) return rc;
    VPX_CHECK_HIP(launch_layernorm_fwd(xc, lnp[0], lnp[1], xc, save ? R.xhat_x : nullptr, save ? R.st_x : st_tmp, partial, B, 7 * n1, stream));
    if ((rc = plain_conv(stream, prec, g, h, Ch, Ch, Wh, (long long)Ch * k * k, k * k, k, k, 4 * Ch, false, nullptr, hc, 4 * Ch, false, wpk5[1], 0.0f, packed))) return rc;
    VPX_CHECK_HIP(launch_layernorm_fwd(hc, lnp[2], lnp[3], hc, save ? R.xhat_h : nullptr, save ? R.st_h : st_tmp + 2 * B, partial, B, 4 * n1, stream));
    if ((rc = plain_conv(stream, prec, g, m, Ch, Ch, Wm, (long long)Ch * k * k, k * k, k, k, 3 * Ch, false, nullptr, mc, 3 * Ch, false, wpk5[2], 0.0f, packed))) return rc;
    VPX_CHECK_HIP(launch_layernorm_fwd(mc, lnp[4], lnp[5], mc, save ? R.xhat_m : nullptr, save ? R.st_m : st_tmp + 4 * B, partial, B, 3 * n1, stream));
    STLNGateArgs ga{};
    ga.npix = (long long)B * s.HW; ga.Ch = Ch; ga.xc = xc; ga.hc = hc; ga.mc = mc; ga.c = c; ga.m = m;
    ga.c_new = c_new; ga.m_new = m_new; ga.delta_c = delta_c; ga.delta_m = delta_m; ga.o_pre = o_pre; ga.mem = mem;
    ga.gates_c = save ? R.gates_c : nullptr; ga.gates_m = save ? R.gates_m : nullptr;
    VPX_CHECK_HIP(launch_st_ln_gates(ga, stream));
    // conv_o(mem) + LayerNorm, conv_last(mem)   (predrnn.py:80-81)
    if ((rc = plain_conv(stream, prec, g, mem, 2 * Ch, 2 * Ch, Wo, (long long)2 * Ch * k * k, k * k, k, k, Ch, false, nullptr, oc, Ch, false, wpk5[3], 0.0f, packed))) return rc;
    VPX_CHECK_HIP(launch_layernorm_fwd(oc, lnp[6], lnp[7], oc, save ? R.xhat_o : nullptr, save ? R.st_o : st_tmp + 6 * B, partial, B, n1, stream));
    if ((rc = plain_conv(stream, prec, g, mem, 2 * Ch, 2 * Ch, Wlast, (long long)2 * Ch, 1, 1, 1, Ch, false, nullptr, lc, Ch, false, wpk5[4], 0.0f, packed))) return rc;
    VPX_CHECK_HIP(launch_st_ln_out(o_pre, oc, lc, h_new, save ? R.o : nullptr, save ? R.tl : nullptr, (long long)s.n_state, stream));
    return VPX_OK;
}

int stlstm_ln_bwd(const vpx_stlstm_desc* d, const float* x, const float* h, const float* c, const float* m,
                  const float* Wx, const float* Wh, const float* Wm, const float* Wo, const float* Wlast,
                  const float* const* ln, const void* reserve, const float* dh_new, const float* dc_new,
                  const float* dm_new, const float* ddc, const float* ddm, float* dx, float* dh, float* dc, float* dm,
                  float* dWx, float* dWh, float* dWm, float* dWo, float* dWlast, float* const* dln, Carver& ws,
                  hipStream_t stream) {
    const STLN s = stln_sizes(d);
    const int B = s.B, Cin = s.Cin, Ch = s.Ch, k = s.k, prec = d->precision;
    const int HW = (int)s.HW, ldG = 7 * Ch;
    const ConvGeo g{B, s.H, s.W};
    const LNReserve R = carve_reserve(const_cast<void*>(reserve), s);
    float* wpk = ws.take(s.wpk_max);
    double* partial = (double*)ws.take((size_t)B * 64 * 2 * 2);
    float* sums = ws.take((size_t)2 * B);
    float* dG7 = ws.take(7 * s.n_state);
    float* dlc = ws.take(s.n_state);
    float* du_o = ws.take(s.n_state);
    float* dcn = ws.take(s.n_state);
    float* dmn = ws.take(s.n_state);
    float* dm_scratch = ws.take(s.n_state);
    float* du_x = ws.take(7 * s.n_state);
    float* du_h = ws.take(4 * s.n_state);
    float* du_m = ws.take(3 * s.n_state);
    float* dmem = ws.take(2 * s.n_state);
    float* slabs = ws.take(s.slab_floats);
    float* lnp[8];
    int rc;
    if ((rc = ln_params_nhwc(ln, lnp, ws, s, stream))) return rc;
    float* dlnp[8];
    const int mult[4] = {7, 4, 3, 1};
    for (int i = 0; i < 8; ++i) dlnp[i] = ws.take(s.HW * mult[i / 2] * Ch);
    VPX_CHECK_CARVE(ws, "vpx_stlstm_step_bwd (LayerNorm)");
    float* dm_out = dm ? dm : dm_scratch;

    // A: through h_new = o * tanh(lc): d(o pre-activation) -> dG7 block 3, d conv_last
    {
        STBwdOutArgs a{(long long)s.n_state, Ch, ldG, 3 * Ch, -1, dh_new, R.o, R.tl, dG7, dlc, 0};
        VPX_CHECK_HIP(launch_st_bwd_out(a, stream));
    }
    // B: LayerNorm of conv_o backward (dy = dG7 block 3), then grads of mem through conv_o and conv_last
    const int blk_o[8] = {3, 0, 0, 0, 0, 0, 0, 0};
    VPX_CHECK_HIP(launch_layernorm_bwd(dG7, ldG, Ch, blk_o, R.xhat_o, R.st_o, lnp[6], B, HW, Ch, partial, sums, du_o, dlnp[6], dlnp[7], stream));
    if ((rc = plain_conv(stream, prec, g, du_o, Ch, Ch, Wo, (long long)2 * Ch * k * k, k * k, k, k, 2 * Ch, true, nullptr, dmem, 2 * Ch, false, wpk))) return rc;
    if ((rc = plain_conv(stream, prec, g, dlc, Ch, Ch, Wlast, (long long)2 * Ch, 1, 1, 1, 2 * Ch, true, nullptr, dmem, 2 * Ch, true, wpk))) return rc;
    // split dmem [B,HW,2Ch] into the two state gradients the gate stage expects
    VPX_CHECK_HIP(vpx_memcpy2d_async(dcn, (size_t)Ch * 4, dmem, (size_t)2 * Ch * 4, (size_t)Ch * 4, (size_t)B * HW, hipMemcpyDeviceToDevice, stream));
    VPX_CHECK_HIP(vpx_memcpy2d_async(dmn, (size_t)Ch * 4, dmem + Ch, (size_t)2 * Ch * 4, (size_t)Ch * 4, (size_t)B * HW, hipMemcpyDeviceToDevice, stream));
    // C: gate groups -> dG7 blocks (i,f,g | o | i',f',g') w.r.t. the SUMS of normalised conv outputs
    {
        STBwdGateArgs a{};
        a.npix = (long long)B * HW; a.Ch = Ch; a.ldG = ldG;
        a.gates_c = R.gates_c; a.gates_m = R.gates_m; a.c = c; a.m = m;
        a.dcn_ext = dc_new; a.dmn_ext = dm_new; a.ddc_ext = ddc; a.ddm_ext = ddm;
        a.dcn_conv = dcn; a.dmn_conv = dmn; a.dG7 = dG7; a.dc = dc; a.dm = dm_out;
        VPX_CHECK_HIP(launch_st_bwd_gates(a, stream));
    }
    // D: LayerNorm backward of conv_x (Wx row order i,f,g,i',f',g',o), conv_h (i,f,g,o), conv_m (i,f,g)
    const int blk_x[8] = {0, 1, 2, 4, 5, 6, 3, 0}, blk_h[8] = {0, 1, 2, 3, 0, 0, 0, 0}, blk_m[8] = {4, 5, 6, 0, 0, 0, 0, 0};
    VPX_CHECK_HIP(launch_layernorm_bwd(dG7, ldG, Ch, blk_x, R.xhat_x, R.st_x, lnp[0], B, HW, 7 * Ch, partial, sums, du_x, dlnp[0], dlnp[1], stream));
    VPX_CHECK_HIP(launch_layernorm_bwd(dG7, ldG, Ch, blk_h, R.xhat_h, R.st_h, lnp[2], B, HW, 4 * Ch, partial, sums, du_h, dlnp[2], dlnp[3], stream));
    VPX_CHECK_HIP(launch_layernorm_bwd(dG7, ldG, Ch, blk_m, R.xhat_m, R.st_m, lnp[4], B, HW, 3 * Ch, partial, sums, du_m, dlnp[4], dlnp[5], stream));
    // E: data gradients
    if (dx && (rc = plain_conv(stream, prec, g, du_x, 7 * Ch, 7 * Ch, Wx, (long long)Cin * k * k, k * k, k, k, Cin, true, nullptr, dx, Cin, false, wpk))) return rc;
    if (dh && (rc = plain_conv(stream, prec, g, du_h, 4 * Ch, 4 * Ch, Wh, (long long)Ch * k * k, k * k, k, k, Ch, true, nullptr, dh, Ch, false, wpk))) return rc;
    if (dm && (rc = plain_conv(stream, prec, g, du_m, 3 * Ch, 3 * Ch, Wm, (long long)Ch * k * k, k * k, k, k, Ch, true, nullptr, dm, Ch, true, wpk))) return rc;
    // F: weight gradients
    if (dWx && (rc = plain_wgrad(stream, prec, g, du_x, 7 * Ch, x, Cin, k, k, slabs, dWx))) return rc;
    if (dWh && (rc = plain_wgrad(stream, prec, g, du_h, 4 * Ch, h, Ch, k, k, slabs, dWh))) return rc;
    if (dWm && (rc = plain_wgrad(stream, prec, g, du_m, 3 * Ch, m, Ch, k, k, slabs, dWm))) return rc;
    if (dWo && (rc = plain_wgrad(stream, prec, g, du_o, Ch, R.mem, 2 * Ch, k, k, slabs, dWo))) return rc;
    if (dWlast && (rc = plain_wgrad(stream, prec, g, dlc, Ch, R.mem, 2 * Ch, 1, 1, slabs, dWlast))) return rc;
    // G: LayerNorm parameter gradients back to the reference's [C,H,W]
    if (dln)
        for (int i = 0; i < 8; ++i)
            if (dln[i]) VPX_CHECK_HIP(launch_nhwc_to_nchw(dlnp[i], dln[i], 1, mult[i / 2] * Ch, s.H, s.W, stream));
    return VPX_OK;
}

}  // namespace vpx
