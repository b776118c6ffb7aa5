// stlstm_api.hip — vpx_stlstm_step_fwd / _bwd: one Spatio-Temporal LSTM cell step (predrnn.py:57-83) as four launches of
// the implicit-GEMM kernel:
//   c group   conv([x|h]; Wx rows i,f,g,o + Wh rows i,f,g,o)  -> c_new, delta_c, o_pre           (fused gates)
//   m group   conv([x|m]; Wx rows i',f',g' + Wm rows i,f,g)   -> m_new, delta_m                  (fused gates)
//   conv_last 1x1 over mem = [c_new|m_new]                    -> lc
//   conv_o    kxk over mem, epilogue h_new = sigmoid(o_pre + conv_o) * tanh(lc)
// The LayerNorm variant (predrnn.py:24-40) is not implemented in this round: it fails loudly.
#include "vpx_host.h"

using namespace vpx;

namespace {

struct STLayout {
    int taps, tiles32, tiles128, ng_l, ksplit_l;
    int mw_g, mw_o;               // workgroup form of the dual gate launch / the conv_o launch (2: 8 waves, 16x16 pixels)
    int o_split, o_ng, o_tiles;   // conv_o as a K-split plain conv accumulating into o_pre (small maps) instead of the fused launch
    int nstage_g, chunks_g;            // gate groups: segments (x: Cin, recurrent: Ch), k x k
    ConvStage stage_g[MAX_STAGE];
    int nstage_o, chunks_o;            // conv_o: segments (c_new: Ch, m_new: Ch), k x k
    ConvStage stage_o[MAX_STAGE];
    int nstage_l, chunks_l;            // conv_last: same segments, 1 x 1
    ConvStage stage_l[MAX_STAGE];
    size_t n_state, n_x;
    size_t wpk_c, wpk_m, wpk_o, wpk_l;  // float counts
    // c5 (convq.hip, round 4): the two gate groups as the jobs of one launch, conv_o + output gate as another, on 16x16-pixel tiles
    // over split-format operands; conv_last (1x1) stays on the first-generation kernel
    bool c5; size_t c5_wc, c5_wm, c5_wo;   // bytes of its weight packs
    // c5k: the same convolutions K-SPLIT on small grids — every K chunk is a job of the launch and writes fp32 partial sums,
    // pointwise kernels (st_pointwise.hip) add them and apply the gates
    bool c5k; int ks_g, ks_o, nt_o; size_t k_wc[6], k_wm[6], k_wo[6];
};
// conv_o on c5: 64-column N tiles, or 32-column ones when those would leave the chip half empty (B = 128 on 16x16 maps, Ch = 128: 256
// workgroups of four waves = one wave per SIMD; 512 at 32 columns)
static inline int c5f_nt_o(const vpx_stlstm_desc* d) {
    const long long mt = (long long)d->B * ((d->H + 15) / 16) * ((d->W + 15) / 16);
    if (const int f = dev_switch("VPX_C5_NT_O", 0)) return f;   // (developer build only)
    return mt * ((d->Ch + 63) / 64) < 384 ? 2 : 4;
}

// VPX_OPT_EXPERIMENT bit 8 keeps the first-generation forward launches (A/B runs, tests)
// Grid rule (measured, tools/ab_predrnn.py, predrnn-pp inference, ms per forward c5 vs first generation): 16x16 maps B = 8 / 16 / 32 / 64 /
// 128: 23.5 / 23.7 / 24.4 / 30.5 / 53.8 vs 13.7 / 14.1 / 17.1 / 26.9 / 58.5; 32x32 maps (128x128x3, 4 layers, 10 -> 30) B = 4 / 8 / 16:
// 72.5 / 78.1 / 90.9 vs 45.0 / 53.0 / 83.5 — a c5 workgroup runs its whole K (200 steps of 96 MFMAs) on one CU, the first generation
// splits K over workgroups when the pixel tiles do not fill the chip. c5 from 96 pixel tiles of 16x16 on — first pass of round 4. After the
// K loop's diet (convq.hip) the unsplit form wins from 48 tiles on against its own K-split job forms (tools/ab_c5_min.sh, 16x16 maps,
// ms per forward unsplit vs K-split: B = 40 30.2 vs 28.9, 48 30.7 vs 33.2, 56 32.1 vs 34.7, 64 33.5 vs 36.2, 80 47.1 vs 50.4); the backward
// launches keep 96 (training step B = 48 221 vs 211 ms, 64 246 vs 242, 80 309.5 vs 309.1).
static bool c5_shape_ok(const vpx_stlstm_desc* d) {
    return d->k == 5 && d->precision == VPX_PREC_BF16X3 && !(d->Ch & 31) && !(d->Cin & 7) && !d->layer_norm && !(g_experiment & 256);
}
bool c5_fwd_applicable(const vpx_stlstm_desc* d) {
    const long long mt = (long long)d->B * ((d->H + 15) / 16) * ((d->W + 15) / 16);
    return c5_shape_ok(d) && (mt >= dev_switch("VPX_C5_MIN_TILES", 48) || (g_experiment & 1024));   // (bit 10 forces the unsplit form on small grids: tests)
}
// below the bar: K-split jobs (VPX_OPT_EXPERIMENT bit 11 keeps the first-generation launches there)
bool c5k_fwd_applicable(const vpx_stlstm_desc* d) { return c5_shape_ok(d) && !c5_fwd_applicable(d) && !(g_experiment & 2048); }
// K chunks so that the launch has about 448 workgroups, every chunk at least four 8-channel stages (25 steps), at most six chunks
static int c5k_chunks(long long wgs1, int S8) {
    const long long target = dev_switch("VPX_C5K_TARGET", 448);   // workgroups the chunked launch aims at
    int ks = (int)((target + wgs1 / 2) / (wgs1 > 0 ? wgs1 : 1));
    if (ks > S8 / 4) ks = S8 / 4;
    if (ks > 6) ks = 6;
    return ks < 1 ? 1 : ks;
}

int check_st_desc(const vpx_stlstm_desc* d) {
    if (!d) { set_error("stlstm desc is NULL"); return VPX_ERR_ARG; }
    if (d->B < 1 || d->Cin < 1 || d->Ch < 1 || d->H < 1 || d->W < 1) { set_error("stlstm desc: non-positive dimension"); return VPX_ERR_ARG; }
    if (d->k < 1 || !(d->k & 1) || d->k > 7) { set_error("stlstm desc: filter size must be odd and <= 7 (got %d)", d->k); return VPX_ERR_ARG; }
    if (d->layout != VPX_LAYOUT_NHWC && d->layout != VPX_LAYOUT_NCHW) { set_error("stlstm desc: unknown layout %d", d->layout); return VPX_ERR_ARG; }
    if ((d->precision < VPX_PREC_F32 || d->precision > VPX_PREC_BF16)) { set_error("stlstm: precision %d not implemented", d->precision); return VPX_ERR_UNSUPPORTED; }
    return VPX_OK;
}

int st_layout(const vpx_stlstm_desc* d, STLayout& L) {
    L.taps = d->k * d->k;
    L.tiles32 = (d->Ch + 31) / 32;
    L.ng_l = plain_groups(d->Ch, (long long)d->B * ((d->H + TILE_H - 1) / TILE_H) * ((d->W + TILE_W - 1) / TILE_W));
    L.tiles128 = plain_tiles_ng(d->Ch, L.ng_l);
    const int segG[2] = {d->Cin, d->Ch};
    const int segO[2] = {d->Ch, d->Ch};
    L.mw_g = d->layer_norm ? 1 : pick_mw(d->B, d->H, d->W, 2 * L.tiles32, d->precision);
    if (L.mw_g > 1 && !conv_fits_lds(segG, 2, d->k, d->k, 4, d->precision, L.mw_g)) L.mw_g = 1;   // (8-wave form too large for LDS)
    L.nstage_g = build_stages(L.stage_g, &L.chunks_g, segG, 2, L.taps, pick_stage_channels(segG, 2, d->k, d->k, 4, d->precision, L.mw_g), d->precision);
    const long long m_tiles = (long long)d->B * ((d->H + TILE_H - 1) / TILE_H) * ((d->W + TILE_W - 1) / TILE_W);
    // conv_o: fused with the output gate (32 channels per workgroup) when that fills the chip; on small maps a 128-wide
    // K-split plain convolution adds conv_o(mem) into o_pre and a pointwise kernel applies the gate
    L.o_ng = 1; L.o_tiles = L.tiles32; L.o_split = 0;
    if (m_tiles * L.tiles32 < 384 && !d->layer_norm) {
        const int ng = plain_groups(d->Ch);
        ConvStage st[MAX_STAGE];
        int ch = 0;
        const int ns = build_stages(st, &ch, segO, 2, L.taps, pick_stage_channels(segO, 2, d->k, d->k, ng, d->precision), d->precision);
        const int ks = ns > 0 ? pick_ksplit(m_tiles * plain_tiles_ng(d->Ch, ng), ns) : 1;
        if (ks > 1) { L.o_split = ks; L.o_ng = ng; L.o_tiles = plain_tiles_ng(d->Ch, ng); }
    }
    L.mw_o = d->layer_norm ? 1 : pick_mw(d->B, d->H, d->W, L.o_tiles, d->precision);
    if (L.mw_o > 1 && !conv_fits_lds(segO, 2, d->k, d->k, L.o_ng, d->precision, L.mw_o)) L.mw_o = 1;
    if (!conv_fits_lds(segG, 2, d->k, d->k, 4, d->precision, L.mw_g) || !conv_fits_lds(segO, 2, d->k, d->k, L.o_ng, d->precision, L.mw_o)) {
        set_error("stlstm: %dx%d kernel over %d+%d channels does not fit the kernel's LDS stages", d->k, d->k, d->Cin, d->Ch);
        return VPX_ERR_UNSUPPORTED;
    }
    L.nstage_o = build_stages(L.stage_o, &L.chunks_o, segO, 2, L.taps, pick_stage_channels(segO, 2, d->k, d->k, L.o_ng, d->precision, L.mw_o), d->precision);
    L.nstage_l = build_stages(L.stage_l, &L.chunks_l, segO, 2, 1, pick_stage_channels(segO, 2, 1, 1, L.ng_l, d->precision), d->precision);
    if (L.nstage_g < 0 || L.nstage_o < 0 || L.nstage_l < 0) { set_error("stlstm: too many channel stages"); return VPX_ERR_UNSUPPORTED; }
    L.n_state = (size_t)d->B * d->H * d->W * d->Ch;
    L.n_x = (size_t)d->B * d->H * d->W * d->Cin;
    L.wpk_c = packed_weight_bytes(L.tiles32, L.chunks_g, 4, d->precision) / 4;
    L.wpk_m = packed_weight_bytes(L.tiles32, L.chunks_g, 3, d->precision) / 4;
    L.wpk_o = packed_weight_bytes(L.o_tiles, L.chunks_o, L.o_ng, d->precision) / 4;
    L.wpk_l = packed_weight_bytes(L.tiles128, L.chunks_l, L.ng_l, d->precision) / 4;
    L.ksplit_l = pick_ksplit(m_tiles * L.tiles128, L.nstage_l);
    L.c5 = c5_fwd_applicable(d);
    if (L.c5) {
        L.c5_wc = align256(c5_wpk_bytes(d->Cin + d->Ch, d->Ch, 8, 4));
        L.c5_wm = align256(c5_wpk_bytes(d->Cin + d->Ch, d->Ch, 8, 3));
        L.c5_wo = align256(c5_wpk_bytes(2 * d->Ch, d->Ch, c5f_nt_o(d)));
    }
    L.c5k = c5k_fwd_applicable(d);
    if (L.c5k) {
        const long long mt = (long long)d->B * ((d->H + 15) / 16) * ((d->W + 15) / 16);
        L.ks_g = c5k_chunks(mt * ((7 * d->Ch + 127) / 128), (d->Cin + d->Ch) / 8);
        L.nt_o = mt * ((d->Ch + 63) / 64) * 6 < 320 ? 2 : 4;
        if (const int f = dev_switch("VPX_C5_NT_O", 0)) L.nt_o = f;   // (developer build only)
        L.ks_o = c5k_chunks(mt * ((d->Ch + L.nt_o * 16 - 1) / (L.nt_o * 16)), 2 * d->Ch / 8);
        for (int k = 0; k < L.ks_g; ++k) {
            L.k_wc[k] = align256(c5_chunk_wpk_bytes(d->Cin + d->Ch, k, L.ks_g, 4 * d->Ch, 8));
            L.k_wm[k] = align256(c5_chunk_wpk_bytes(d->Cin + d->Ch, k, L.ks_g, 3 * d->Ch, 8));
        }
        for (int k = 0; k < L.ks_o; ++k) L.k_wo[k] = align256(c5_chunk_wpk_bytes(2 * d->Ch, k, L.ks_o, d->Ch, L.nt_o));
    }
    return VPX_OK;
}

ConvPlan base_plan(const vpx_stlstm_desc* d, int k) {
    ConvPlan P{};
    P.prec = d->precision;
    P.B = d->B; P.H = d->H; P.W = d->W; P.kh = k; P.kw = k;
    P.tiles_x = (d->W + TILE_W - 1) / TILE_W;
    P.tiles_y = (d->H + TILE_H - 1) / TILE_H;
    return P;
}

}  // namespace

namespace vpx {
STSplitShadows st_shadows_of(const vpx_stlstm_shadows* p) {
    STSplitShadows s{};
    if (!p) return s;
    for (int i = 0; i < 5; ++i) s.in[i] = reinterpret_cast<const char*>(p->in[i]);
    for (int i = 0; i < 3; ++i) s.out[i] = reinterpret_cast<char*>(p->out[i]);
    s.dg8 = reinterpret_cast<char*>(p->dg8_out);
    s.set = 1;
    return s;
}
}  // namespace vpx

extern "C" {

int vpx_stlstm_uses_split(const vpx_stlstm_desc* d) {
    if (check_st_desc(d) != VPX_OK || d->layout != VPX_LAYOUT_NHWC) return 0;
    return (c5_fwd_applicable(d) || c5k_fwd_applicable(d)) ? 1 : 0;
}

size_t vpx_stlstm_reserve_bytes(const vpx_stlstm_desc* d) {
    STLayout L;
    if (check_st_desc(d) != VPX_OK || st_layout(d, L) != VPX_OK) return 0;
    if (!(d->flags & VPX_FLAG_SAVE_FOR_BWD)) return 0;
    if (d->layer_norm) return stlstm_ln_reserve_bytes(d);
    // gates_c (3Ch) + gates_m (3Ch) + o + tanh(conv_last): 8 state-sized planes
    return 2 * align256(3 * L.n_state * 4) + 2 * align256(L.n_state * 4);
}

size_t vpx_stlstm_workspace_bytes(const vpx_stlstm_desc* d) {
    STLayout L;
    if (check_st_desc(d) != VPX_OK || st_layout(d, L) != VPX_OK) return 0;
    if (d->layer_norm) return stlstm_ln_workspace_bytes(d);
    size_t b = align256(L.wpk_c * 4) + align256(L.wpk_m * 4) + align256(L.wpk_o * 4) + align256(L.wpk_l * 4);
    b += 2 * align256(L.n_state * 4);  // o_pre, lc
    if (L.c5) b += L.c5_wc + L.c5_wm + L.c5_wo + align256(L.n_x * 4) + 4 * align256(L.n_state * 4);   // packs; x, h, m, c_new, m_new in split format
    if (L.c5k) {   // chunk packs; the split copies; partial sums of the gate groups (7Ch per chunk) / of conv_o (Ch per chunk)
        for (int k = 0; k < L.ks_g; ++k) b += L.k_wc[k] + L.k_wm[k];
        for (int k = 0; k < L.ks_o; ++k) b += L.k_wo[k];
        b += align256(L.n_x * 4) + 4 * align256(L.n_state * 4) + align256((size_t)L.ks_g * 7 * L.n_state * 4) + align256((size_t)L.ks_o * L.n_state * 4);
    }
    if (d->layout == VPX_LAYOUT_NCHW) b += align256(L.n_x * 4) + 8 * align256(L.n_state * 4);
    size_t bwd = 0;
    if (d->flags & VPX_FLAG_SAVE_FOR_BWD) bwd = stlstm_bwd_workspace_bytes(d);
    return (b > bwd ? b : bwd) + 256;
}

int vpx_stlstm_step_fwd(const vpx_stlstm_desc* d, const float* x, const float* h, const float* c, const float* m,
                        const float* Wx, const float* Wh, const float* Wm, const float* Wo, const float* Wlast,
                        const float* const* ln, float* h_new, float* c_new, float* m_new, float* delta_c,
                        float* delta_m, void* reserve, size_t reserve_bytes, void* workspace, size_t workspace_bytes,
                        void* stream_) {
    return vpx_stlstm_step_fwd_ex(d, x, h, c, m, Wx, Wh, Wm, Wo, Wlast, ln, h_new, c_new, m_new, delta_c, delta_m, reserve, reserve_bytes,
                                  workspace, workspace_bytes, stream_, nullptr);
}

int vpx_stlstm_step_fwd_ex(const vpx_stlstm_desc* d, const float* x, const float* h, const float* c, const float* m,
                           const float* Wx, const float* Wh, const float* Wm, const float* Wo, const float* Wlast,
                           const float* const* ln, float* h_new, float* c_new, float* m_new, float* delta_c,
                           float* delta_m, void* reserve, size_t reserve_bytes, void* workspace, size_t workspace_bytes,
                           void* stream_, const vpx_stlstm_shadows* shadows) {
    STSplitShadows sh = st_shadows_of(shadows);   // an argument of THIS call: nothing survives it, nothing precedes it
    int rc = check_st_desc(d);
    if (rc != VPX_OK) return rc;
    if (d->layout != VPX_LAYOUT_NHWC) sh = STSplitShadows{};
    STLayout L;
    if ((rc = st_layout(d, L)) != VPX_OK) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !h || !c || !m || !Wx || !Wh || !Wm || !Wo || !Wlast || !h_new || !c_new || !m_new || !delta_c || !delta_m) {
        set_error("vpx_stlstm_step_fwd: NULL tensor argument");
        return VPX_ERR_ARG;
    }
    const bool save = (d->flags & VPX_FLAG_SAVE_FOR_BWD) != 0;
    if (save && (!reserve || reserve_bytes < vpx_stlstm_reserve_bytes(d))) { set_error("vpx_stlstm_step_fwd: reserve too small"); return VPX_ERR_WORKSPACE; }
    if (!workspace || workspace_bytes < vpx_stlstm_workspace_bytes(d)) {
        set_error("vpx_stlstm_step_fwd: workspace too small (%zu < %zu)", workspace_bytes, vpx_stlstm_workspace_bytes(d));
        return VPX_ERR_WORKSPACE;
    }
    const int B = d->B, Cin = d->Cin, Ch = d->Ch, H = d->H, Wd = d->W, k = d->k;
    const size_t HW = (size_t)H * Wd;
    if (d->layer_norm) {  // unfused LayerNorm path (stlstm_ln_api.hip); same layout adaptation around it
        if (!ln) { set_error("vpx_stlstm_step_fwd: layer_norm set but ln is NULL"); return VPX_ERR_ARG; }
        Carver w2(workspace, workspace_bytes);
        const float *xn = x, *hn = h, *cn = c, *mn = m;
        float *outs[5] = {h_new, c_new, m_new, delta_c, delta_m}, *outn[5] = {h_new, c_new, m_new, delta_c, delta_m};
        if (d->layout == VPX_LAYOUT_NCHW) {
            float* bx = w2.take(L.n_x);
            float* st[8];
            for (auto& p : st) p = w2.take(L.n_state);
            VPX_CHECK_CARVE(w2, "vpx_stlstm_step_fwd (LayerNorm, NCHW)");
            VPX_CHECK_HIP(launch_nchw_to_nhwc(x, bx, B, Cin, H, Wd, stream)); xn = bx;
            VPX_CHECK_HIP(launch_nchw_to_nhwc(h, st[0], B, Ch, H, Wd, stream)); hn = st[0];
            VPX_CHECK_HIP(launch_nchw_to_nhwc(c, st[1], B, Ch, H, Wd, stream)); cn = st[1];
            VPX_CHECK_HIP(launch_nchw_to_nhwc(m, st[2], B, Ch, H, Wd, stream)); mn = st[2];
            for (int i = 0; i < 5; ++i) outn[i] = st[3 + i];
        }
        rc = stlstm_ln_fwd(d, xn, hn, cn, mn, Wx, Wh, Wm, Wo, Wlast, ln, outn[0], outn[1], outn[2], outn[3], outn[4], reserve, w2, stream);
        if (rc != VPX_OK) return rc;
        if (d->layout == VPX_LAYOUT_NCHW)
            for (int i = 0; i < 5; ++i) VPX_CHECK_HIP(launch_nhwc_to_nchw(outn[i], outs[i], B, Ch, H, Wd, stream));
        return VPX_OK;
    }
    Carver ws(workspace, workspace_bytes);
    float* wpk_c = ws.take(L.wpk_c);
    float* wpk_m = ws.take(L.wpk_m);
    float* wpk_o = ws.take(L.wpk_o);
    float* wpk_l = ws.take(L.wpk_l);
    float* o_pre = ws.take(L.n_state);
    float* lc = ws.take(L.n_state);
    char *c5wc = nullptr, *c5wm = nullptr, *c5wo = nullptr, *x_sp = nullptr, *h_sp = nullptr, *m_sp = nullptr, *cn_sp = nullptr, *mn_sp = nullptr;
    if (L.c5 && !d->layer_norm) {
        c5wc = (char*)ws.take(L.c5_wc / 4); c5wm = (char*)ws.take(L.c5_wm / 4); c5wo = (char*)ws.take(L.c5_wo / 4);
        x_sp = (char*)ws.take(L.n_x); h_sp = (char*)ws.take(L.n_state); m_sp = (char*)ws.take(L.n_state);
        cn_sp = (char*)ws.take(L.n_state); mn_sp = (char*)ws.take(L.n_state);
    }
    char *kwc[6] = {}, *kwm[6] = {}, *kwo[6] = {};
    float *part_g = nullptr, *part_o = nullptr;
    if (L.c5k && !d->layer_norm) {
        for (int k = 0; k < L.ks_g; ++k) { kwc[k] = (char*)ws.take(L.k_wc[k] / 4); kwm[k] = (char*)ws.take(L.k_wm[k] / 4); }
        for (int k = 0; k < L.ks_o; ++k) kwo[k] = (char*)ws.take(L.k_wo[k] / 4);
        x_sp = (char*)ws.take(L.n_x); h_sp = (char*)ws.take(L.n_state); m_sp = (char*)ws.take(L.n_state);
        cn_sp = (char*)ws.take(L.n_state); mn_sp = (char*)ws.take(L.n_state);
        part_g = ws.take((size_t)L.ks_g * 7 * L.n_state); part_o = ws.take((size_t)L.ks_o * L.n_state);
    }

    VPX_CHECK_CARVE(ws, "vpx_stlstm_step_fwd");
    const float *xn = x, *hn = h, *cn = c, *mn = m;
    float *hO = h_new, *cO = c_new, *mO = m_new, *dcO = delta_c, *dmO = delta_m;
    if (d->layout == VPX_LAYOUT_NCHW) {
        float* bx = ws.take(L.n_x);
        float* st[8];
        for (auto& p : st) p = ws.take(L.n_state);
        VPX_CHECK_CARVE(ws, "vpx_stlstm_step_fwd");
        VPX_CHECK_HIP(launch_nchw_to_nhwc(x, bx, B, Cin, H, Wd, stream)); xn = bx;
        VPX_CHECK_HIP(launch_nchw_to_nhwc(h, st[0], B, Ch, H, Wd, stream)); hn = st[0];
        VPX_CHECK_HIP(launch_nchw_to_nhwc(c, st[1], B, Ch, H, Wd, stream)); cn = st[1];
        VPX_CHECK_HIP(launch_nchw_to_nhwc(m, st[2], B, Ch, H, Wd, stream)); mn = st[2];
        hO = st[3]; cO = st[4]; mO = st[5]; dcO = st[6]; dmO = st[7];
    }

    float *gates_c = nullptr, *gates_m = nullptr, *o_save = nullptr, *tl_save = nullptr;
    if (save) {
        char* r = (char*)reserve;
        gates_c = (float*)r; r += align256(3 * L.n_state * 4);
        gates_m = (float*)r; r += align256(3 * L.n_state * 4);
        o_save = (float*)r; r += align256(L.n_state * 4);
        tl_save = (float*)r;
    }
    const bool packed = (d->flags & VPX_FLAG_WEIGHTS_PACKED) != 0;
    if (L.c5) {
        // ---- round-4 path: operands in the split format, gate groups + conv_o on the 16x16-tile kernel (convq.hip, c5) ----
        const long long npix = (long long)B * (long long)HW;
        // operands a previous call left in the split format come back as shadows; the others are converted here
        if (sh.in[0]) x_sp = const_cast<char*>(sh.in[0]); else VPX_CHECK_HIP(launch_split_convert(xn, x_sp, npix, Cin, stream));
        if (sh.in[1]) h_sp = const_cast<char*>(sh.in[1]); else VPX_CHECK_HIP(launch_split_convert(hn, h_sp, npix, Ch, stream));
        if (sh.in[2]) m_sp = const_cast<char*>(sh.in[2]); else VPX_CHECK_HIP(launch_split_convert(mn, m_sp, npix, Ch, stream));
        if (sh.out[1]) cn_sp = sh.out[1];
        if (sh.out[2]) mn_sp = sh.out[2];
        C5Plan cp{};
        cp.B = B; cp.H = H; cp.W = Wd;
        cp.src[0] = C5Src{x_sp, (long long)HW * Cin * 4, Cin * 4, 0};
        cp.src[1] = C5Src{h_sp, (long long)HW * Ch * 4, Ch * 4, 0};
        cp.src[2] = C5Src{m_sp, (long long)HW * Ch * 4, Ch * 4, 0};
        const long long ws_x = (long long)Cin * L.taps, ws_h = (long long)Ch * L.taps;   // weight strides of one output channel
        for (int grp = 0; grp < 2; ++grp) {   // 0: c group (i,f,g,o_pre) from [x | h]; 1: m group (i',f',g') from [x | m]
            C5Job& j = cp.job[cp.njobs++];
            j = C5Job{};
            j.nrange = 2;
            j.r_src[0] = 0; j.r_c0[0] = 0; j.r_n[0] = Cin;
            j.r_src[1] = grp ? 2 : 1; j.r_c0[1] = 0; j.r_n[1] = Ch;
            j.epi = 1; j.Co = Ch; j.Ch = Ch; j.ng = grp ? 3 : 4; j.fbias = 1.0f;
            j.wpk = grp ? c5wm : c5wc;
            j.e_in0 = grp ? mn : cn;
            j.e_out[0] = grp ? mO : cO; j.e_out[1] = grp ? dmO : dcO; j.e_out[2] = grp ? nullptr : o_pre; j.e_out[3] = grp ? gates_m : gates_c;
            j.e_sp = grp ? mn_sp : cn_sp;
            C5PackRange pr[2];
            // x rows: (i,f,g,o) = blocks 0,1,2,6 of Wx, (i',f',g') = blocks 3,4,5 (predrnn.py:61); recurrent rows: blocks 0.. of Wh / Wm (:62-63)
            if (grp == 0) pr[0] = C5PackRange{Wx, ws_x, (long long)L.taps, 0, {0, Ch, 2 * Ch, 6 * Ch}};
            else pr[0] = C5PackRange{Wx, ws_x, (long long)L.taps, 0, {3 * Ch, 4 * Ch, 5 * Ch, 0}};
            pr[1] = C5PackRange{grp ? Wm : Wh, ws_h, (long long)L.taps, 0, {0, Ch, 2 * Ch, 3 * Ch}};
            if ((rc = c5_prepare_job(j, 8, pr, grp ? 3 : 4, 0, packed, stream))) return rc;
        }
        VPX_CHECK_HIP(launch_c5(cp, 8, stream));
        if (!packed) {   // conv_last's first-generation pack
            PackDesc pl{};
            pl.seg[0] = PackSeg{Wlast, (long long)2 * Ch, 1, 0, Ch};
            pl.seg[1] = PackSeg{Wlast, (long long)2 * Ch, 1, Ch, Ch};
            memcpy(pl.stage, L.stage_l, sizeof(ConvStage) * L.nstage_l);
            pl.nstage = L.nstage_l; pl.chunks_total = L.chunks_l; pl.prec = d->precision; pl.taps = 1;
            fill_plain_pack(pl, Ch, 0, L.ng_l);
            VPX_CHECK_HIP(launch_pack_weights(pl, wpk_l, stream));
        }
    } else if (L.c5k) {
        // ---- small grid: the same two gate groups, K-split into ks_g chunks = 2 * ks_g jobs of one launch writing partial sums ----
        const long long npix = (long long)B * (long long)HW;
        // operands a previous call left in the split format come back as shadows; the others are converted here
        if (sh.in[0]) x_sp = const_cast<char*>(sh.in[0]); else VPX_CHECK_HIP(launch_split_convert(xn, x_sp, npix, Cin, stream));
        if (sh.in[1]) h_sp = const_cast<char*>(sh.in[1]); else VPX_CHECK_HIP(launch_split_convert(hn, h_sp, npix, Ch, stream));
        if (sh.in[2]) m_sp = const_cast<char*>(sh.in[2]); else VPX_CHECK_HIP(launch_split_convert(mn, m_sp, npix, Ch, stream));
        if (sh.out[1]) cn_sp = sh.out[1];
        if (sh.out[2]) mn_sp = sh.out[2];
        C5Plan cp{};
        cp.B = B; cp.H = H; cp.W = Wd;
        cp.src[0] = C5Src{x_sp, (long long)HW * Cin * 4, Cin * 4, 0};
        cp.src[1] = C5Src{h_sp, (long long)HW * Ch * 4, Ch * 4, 0};
        cp.src[2] = C5Src{m_sp, (long long)HW * Ch * 4, Ch * 4, 0};
        const long long ws_x = (long long)Cin * L.taps, ws_h = (long long)Ch * L.taps;   // weight strides of one output channel
        for (int grp = 0; grp < 2; ++grp) {
            C5Job full{};
            full.nrange = 2;
            full.r_src[0] = 0; full.r_c0[0] = 0; full.r_n[0] = Cin;
            full.r_src[1] = grp ? 2 : 1; full.r_c0[1] = 0; full.r_n[1] = Ch;
            full.epi = 0; full.Co = (grp ? 3 : 4) * Ch; full.ld = 7 * Ch; full.accumulate = 0;
            full.out_bstride = (long long)HW * 7 * Ch;
            C5PackRange prf[2];
            if (grp == 0) prf[0] = C5PackRange{Wx, ws_x, (long long)L.taps, 0, {0, Ch, 2 * Ch, 6 * Ch}};
            else prf[0] = C5PackRange{Wx, ws_x, (long long)L.taps, 0, {3 * Ch, 4 * Ch, 5 * Ch, 0}};
            prf[1] = C5PackRange{grp ? Wm : Wh, ws_h, (long long)L.taps, 0, {0, Ch, 2 * Ch, 3 * Ch}};
            for (int kk = 0; kk < L.ks_g; ++kk) {
                C5Job& j = cp.job[cp.njobs++];
                C5PackRange pr[3];
                c5_chunk_job(full, prf, kk, L.ks_g, j, pr);
                j.wpk = grp ? kwm[kk] : kwc[kk];
                j.out = part_g + (size_t)kk * 7 * L.n_state + (grp ? 4 * Ch : 0);
                if ((rc = c5_prepare_job(j, 8, pr, grp ? 3 : 4, 0, packed, stream, 1))) return rc;
            }
        }
        VPX_CHECK_HIP(launch_c5(cp, 8, stream));
        STGatesKSArgs ga{};
        ga.npix = npix; ga.Ch = Ch; ga.ks = L.ks_g; ga.pstride = (long long)(7 * L.n_state); ga.fbias = 1.0f;
        ga.part = part_g; ga.c = cn; ga.m = mn;
        ga.c_new = cO; ga.m_new = mO; ga.delta_c = dcO; ga.delta_m = dmO; ga.o_pre = o_pre; ga.gates_c = gates_c; ga.gates_m = gates_m;
        ga.cn_sp = cn_sp; ga.mn_sp = mn_sp;
        VPX_CHECK_HIP(launch_st_gates_ks(ga, stream));
        if (!packed) {   // conv_last's first-generation pack (used when the streaming form does not apply)
            PackDesc pl{};
            pl.seg[0] = PackSeg{Wlast, (long long)2 * Ch, 1, 0, Ch};
            pl.seg[1] = PackSeg{Wlast, (long long)2 * Ch, 1, Ch, Ch};
            memcpy(pl.stage, L.stage_l, sizeof(ConvStage) * L.nstage_l);
            pl.nstage = L.nstage_l; pl.chunks_total = L.chunks_l; pl.prec = d->precision; pl.taps = 1;
            fill_plain_pack(pl, Ch, 0, L.ng_l);
            VPX_CHECK_HIP(launch_pack_weights(pl, wpk_l, stream));
        }
    } else
    // ---- weight repack (skipped when the caller vouches the workspace still holds it) ----
    if (!(d->flags & VPX_FLAG_WEIGHTS_PACKED)) {
        PackDesc pd{};
        // c group: x rows (i,f,g,o) = blocks 0,1,2,6 of Wx (predrnn.py:61); h rows (i,f,g,o) = blocks 0..3 of Wh (:62)
        pd.seg[0] = PackSeg{Wx, (long long)Cin * L.taps, L.taps, 0, Cin};
        pd.seg[1] = PackSeg{Wh, (long long)Ch * L.taps, L.taps, 0, Ch};
        memcpy(pd.stage, L.stage_g, sizeof(ConvStage) * L.nstage_g);
        pd.nstage = L.nstage_g; pd.chunks_total = L.chunks_g; pd.prec = d->precision; pd.n_tiles = L.tiles32; pd.taps = L.taps; pd.NG = 4;
        const int xr[4] = {0, 1, 2, 6};
        for (int g = 0; g < 4; ++g) { pd.rowbase[0][g] = xr[g] * Ch; pd.rowbase[1][g] = g * Ch; pd.goff[g] = 0; }
        pd.tile_stride = 32; pd.nch = Ch;
        VPX_CHECK_HIP(launch_pack_weights(pd, wpk_c, stream));
        // m group: x rows (i',f',g') = blocks 3,4,5 of Wx; m rows (i,f,g) = blocks 0..2 of Wm (:63)
        pd.seg[1] = PackSeg{Wm, (long long)Ch * L.taps, L.taps, 0, Ch};
        pd.NG = 3;
        for (int g = 0; g < 3; ++g) { pd.rowbase[0][g] = (3 + g) * Ch; pd.rowbase[1][g] = g * Ch; }
        pd.rowbase[0][3] = pd.rowbase[1][3] = -1;
        VPX_CHECK_HIP(launch_pack_weights(pd, wpk_m, stream));
        // conv_o over mem = [c_new | m_new]: Wo [Ch, 2Ch, k, k]
        PackDesc po{};
        po.seg[0] = PackSeg{Wo, (long long)2 * Ch * L.taps, L.taps, 0, Ch};
        po.seg[1] = PackSeg{Wo, (long long)2 * Ch * L.taps, L.taps, Ch, Ch};
        memcpy(po.stage, L.stage_o, sizeof(ConvStage) * L.nstage_o);
        po.nstage = L.nstage_o; po.chunks_total = L.chunks_o; po.prec = d->precision; po.taps = L.taps;
        fill_plain_pack(po, Ch, 0, L.o_ng);  // ng = 1: 32 output channels per N tile (the fused launch's layout)
        VPX_CHECK_HIP(launch_pack_weights(po, wpk_o, stream));
        // conv_last 1x1: Wlast [Ch, 2Ch, 1, 1]
        PackDesc pl{};
        pl.seg[0] = PackSeg{Wlast, (long long)2 * Ch, 1, 0, Ch};
        pl.seg[1] = PackSeg{Wlast, (long long)2 * Ch, 1, Ch, Ch};
        memcpy(pl.stage, L.stage_l, sizeof(ConvStage) * L.nstage_l);
        pl.nstage = L.nstage_l; pl.chunks_total = L.chunks_l; pl.prec = d->precision; pl.taps = 1;
        fill_plain_pack(pl, Ch, 0, L.ng_l);
        VPX_CHECK_HIP(launch_pack_weights(pl, wpk_l, stream));
    }


    // ---- launch 1: c group ----
    if (!L.c5 && !L.c5k) {
        ConvPlan P = base_plan(d, k);
        set_plan_tiles(P, L.mw_g);
        P.nseg = 2;
        P.seg[0] = ConvSeg{xn, (long long)(HW * Cin), Cin, 0};
        P.seg[1] = ConvSeg{hn, (long long)(HW * Ch), Ch, 0};
        P.nstage = L.nstage_g; memcpy(P.stage, L.stage_g, sizeof(ConvStage) * L.nstage_g);
        P.chunks_total = L.chunks_g; P.a_bytes = conv_a_bytes(L.stage_g, L.nstage_g, k, k, L.mw_g); P.wpk = wpk_c;
        STGateArgs ea{Ch, 1.0f, cn, cO, dcO, o_pre, gates_c};
        // ---- launch 2 (merged with 1): m group — the two groups are independent and run as ONE dual launch ----
        ConvPlan Pm = P;
        Pm.seg[1] = ConvSeg{mn, (long long)(HW * Ch), Ch, 0};
        Pm.wpk = wpk_m;
        STGateArgs em{Ch, 1.0f, mn, mO, dmO, nullptr, gates_m};
        VPX_CHECK_HIP(launch_st_gates_dual(P, ea, Pm, em, L.tiles32, stream));
    }
    // ---- launch 3: conv_last(mem) 1x1 -> lc ----
    C1Args c1{};
    c1.x[0] = cO; c1.x[1] = mO; c1.xld[0] = c1.xld[1] = Ch; c1.xc[0] = c1.xc[1] = Ch;
    c1.npix = (long long)B * (long long)HW;
    c1.w = Wlast; c1.w_sn = 2 * Ch; c1.w_sc = 1;
    c1.y[0] = lc; c1.yld[0] = Ch; c1.ysplit = Ch; c1.Co = Ch;
    if ((L.c5 || L.c5k) && cn_sp && mn_sp && !(g_experiment & (1 << 27))) {
        // the gate stage left c_new / m_new in the operand format as well: read THOSE (no fp32 -> (hi, lo) conversion in the kernel; round 6).
        // VPX_OPT_EXPERIMENT bit 27 keeps the fp32 sources (tests, A/B)
        c1.x[0] = reinterpret_cast<const float*>(cn_sp); c1.x[1] = reinterpret_cast<const float*>(mn_sp); c1.x_split = 1;
    }
    if (c1_applicable(c1, d->precision)) {
        VPX_CHECK_HIP(launch_c1(c1, stream));   // streaming form (conv1.hip): no weight pack, no LDS
    } else {
        ConvPlan P = base_plan(d, 1);
        P.nseg = 2;
        P.seg[0] = ConvSeg{cO, (long long)(HW * Ch), Ch, 0};
        P.seg[1] = ConvSeg{mO, (long long)(HW * Ch), Ch, 0};
        if (L.c5 || L.c5k) {   // the gate stage left c_new / m_new in the split format as well: staged without conversion
            P.seg[0] = ConvSeg{reinterpret_cast<const float*>(cn_sp), (long long)(HW * Ch), Ch, 0, 1};
            P.seg[1] = ConvSeg{reinterpret_cast<const float*>(mn_sp), (long long)(HW * Ch), Ch, 0, 1};
        }
        P.nstage = L.nstage_l; memcpy(P.stage, L.stage_l, sizeof(ConvStage) * L.nstage_l);
        P.chunks_total = L.chunks_l; P.a_bytes = conv_a_bytes(L.stage_l, L.nstage_l, 1, 1); P.wpk = wpk_l;
        PlainEpiArgs ea{};
        ea.Co = Ch; ea.split = Ch; ea.out0 = lc; ea.bstride0 = (long long)(HW * Ch); ea.ld0 = Ch; ea.ng = L.ng_l;
        P.ksplit = L.ksplit_l;
        if (P.ksplit > 1) VPX_CHECK_HIP(vpx_memset_async(lc, 0, L.n_state * sizeof(float), stream));
        VPX_CHECK_HIP(launch_conv_plain_f32(P, ea, L.tiles128, stream));
    }
    // ---- launch 4: conv_o(mem) + output gate ----
    if (L.c5) {
        C5Plan cp{};
        cp.B = B; cp.H = H; cp.W = Wd;
        cp.src[0] = C5Src{cn_sp, (long long)HW * Ch * 4, Ch * 4, 0};
        cp.src[1] = C5Src{mn_sp, (long long)HW * Ch * 4, Ch * 4, 0};
        C5Job& j = cp.job[cp.njobs++];
        j = C5Job{};
        j.nrange = 2;
        j.r_src[0] = 0; j.r_c0[0] = 0; j.r_n[0] = Ch;
        j.r_src[1] = 1; j.r_c0[1] = 0; j.r_n[1] = Ch;
        j.epi = 2; j.Co = Ch; j.Ch = Ch; j.wpk = c5wo;
        j.e_in0 = o_pre; j.e_in1 = lc;
        j.e_out[0] = hO; j.e_out[1] = o_save; j.e_out[2] = tl_save;
        j.e_sp = sh.out[0];
        const long long so = (long long)2 * Ch * L.taps;
        C5PackRange pr[2] = {C5PackRange{Wo, so, (long long)L.taps, 0, {0, 0, 0, 0}}, C5PackRange{Wo, so, (long long)L.taps, Ch, {0, 0, 0, 0}}};
        if ((rc = c5_prepare_job(j, c5f_nt_o(d), pr, 0, 0, packed, stream))) return rc;
        VPX_CHECK_HIP(launch_c5(cp, c5f_nt_o(d), stream));
    } else if (L.c5k) {
        C5Plan cp{};
        cp.B = B; cp.H = H; cp.W = Wd;
        cp.src[0] = C5Src{cn_sp, (long long)HW * Ch * 4, Ch * 4, 0};
        cp.src[1] = C5Src{mn_sp, (long long)HW * Ch * 4, Ch * 4, 0};
        C5Job full{};
        full.nrange = 2;
        full.r_src[0] = 0; full.r_c0[0] = 0; full.r_n[0] = Ch;
        full.r_src[1] = 1; full.r_c0[1] = 0; full.r_n[1] = Ch;
        full.epi = 0; full.Co = Ch; full.ld = Ch; full.accumulate = 0; full.out_bstride = (long long)HW * Ch;
        const long long so = (long long)2 * Ch * L.taps;
        const C5PackRange prf[2] = {C5PackRange{Wo, so, (long long)L.taps, 0, {0, 0, 0, 0}}, C5PackRange{Wo, so, (long long)L.taps, Ch, {0, 0, 0, 0}}};
        for (int kk = 0; kk < L.ks_o; ++kk) {
            C5Job& j = cp.job[cp.njobs++];
            C5PackRange pr[3];
            c5_chunk_job(full, prf, kk, L.ks_o, j, pr);
            j.wpk = kwo[kk];
            j.out = part_o + (size_t)kk * L.n_state;
            if ((rc = c5_prepare_job(j, L.nt_o, pr, 0, 0, packed, stream))) return rc;
        }
        VPX_CHECK_HIP(launch_c5(cp, L.nt_o, stream));
        STOutKSArgs oa{};
        oa.n = (long long)L.n_state; oa.ks = L.ks_o; oa.pstride = (long long)L.n_state; oa.part = part_o;
        oa.o_pre = o_pre; oa.lc = lc; oa.h_new = hO; oa.o_save = o_save; oa.tl_save = tl_save; oa.h_sp = sh.out[0]; oa.Ch = Ch;
        VPX_CHECK_HIP(launch_st_out_ks(oa, stream));
    } else {
        ConvPlan P = base_plan(d, k);
        set_plan_tiles(P, L.mw_o);
        P.nseg = 2;
        P.seg[0] = ConvSeg{cO, (long long)(HW * Ch), Ch, 0};
        P.seg[1] = ConvSeg{mO, (long long)(HW * Ch), Ch, 0};
        P.nstage = L.nstage_o; memcpy(P.stage, L.stage_o, sizeof(ConvStage) * L.nstage_o);
        P.chunks_total = L.chunks_o; P.a_bytes = conv_a_bytes(L.stage_o, L.nstage_o, k, k, L.mw_o); P.wpk = wpk_o;
        if (L.o_split > 1) {
            PlainEpiArgs pa{};
            pa.Co = Ch; pa.split = Ch; pa.out0 = o_pre; pa.bstride0 = (long long)(HW * Ch); pa.ld0 = Ch; pa.ng = L.o_ng;
            pa.accumulate = 1;  // o_pre already holds o_x + o_h (c-group launch)
            P.ksplit = L.o_split;
            VPX_CHECK_HIP(launch_conv_plain_f32(P, pa, L.o_tiles, stream));
            VPX_CHECK_HIP(launch_st_ln_out(o_pre, nullptr, lc, hO, o_save, tl_save, (long long)L.n_state, stream));
        } else {
            STOutArgs ea{Ch, o_pre, lc, hO, o_save, tl_save};
            VPX_CHECK_HIP(launch_st_out_f32(P, ea, L.tiles32, stream));
        }
    }
    if (d->layout == VPX_LAYOUT_NCHW) {
        VPX_CHECK_HIP(launch_nhwc_to_nchw(hO, h_new, B, Ch, H, Wd, stream));
        VPX_CHECK_HIP(launch_nhwc_to_nchw(cO, c_new, B, Ch, H, Wd, stream));
        VPX_CHECK_HIP(launch_nhwc_to_nchw(mO, m_new, B, Ch, H, Wd, stream));
        VPX_CHECK_HIP(launch_nhwc_to_nchw(dcO, delta_c, B, Ch, H, Wd, stream));
        VPX_CHECK_HIP(launch_nhwc_to_nchw(dmO, delta_m, B, Ch, H, Wd, stream));
    }
    return VPX_OK;
}

}  // extern "C"
