// stlstm_bwd_api.hip — vpx_stlstm_step_bwd: backward of one ST-LSTM cell step (predrnn.py:57-83), explicit instead of
// autograd. d(pre-activations) of all seven gate blocks live in ONE tensor dG7 [B,HW,7Ch] ordered
// (i,f,g | o | i',f',g') so that Wh's rows are its first 4Ch channels, Wm's rows its last 3Ch, and Wx's row blocks the
// permutation {0,1,2,6,3,4,5}; convolution sources are channel slices of it (ConvSeg.ld).
//   A  pointwise: dh_new -> d(o pre-act) [dG7 block 3], d conv_last
//   B  dgrad conv_o (k x k) + dgrad conv_last (1 x 1, accumulate) -> grads of c_new / m_new through mem
//   C  pointwise: gate groups -> dG7 blocks 0-2, 4-6; dc; direct part of dm
//   D  dgrad: dx (3 segments over Wx), dh (Wh), dm += (Wm)
//   E  wgrad x5 (K-slice slabs + reduce): Wo, Wlast, Wx (row-block map), Wh, Wm
#include "vpx_host.h"

using namespace vpx;

namespace {

struct STBwdLayout {
    int taps, Ch, Cin;
    size_t n_state, n_x, n_g7;
    // dgrad plans: stage tables + chunk counts + N tiles
    struct DG { int nstage, chunks, tiles, ng, ksplit, mw; ConvStage stage[MAX_STAGE]; size_t wpk; } o, l, x, h, m;
    int n_slices;
    size_t slab_floats;
    // stw (wgrad2.hip): the four k x k weight gradients in one launch on split operands — 5x5, bf16x3, channels in 8s
    bool stw; int stw_pairs, stw_ns;
    // c5 (convq.hip): the k x k data gradients on 16x16-pixel tiles over the split-format dG8, two launches (conv_o's adjoint; dx | dh | dm)
    // slots: conv_o -> c, conv_o -> m, dx, dh, dm. On small grids (< 96 pixel tiles) every slot's K is cut into c5_ks chunks that run
    // as separate jobs writing fp32 partial sums (no atomics), added by sum_partials_kernel (st_pointwise.hip)
    bool c5; int c5_ks[5]; size_t c5_wpk[5][6]; size_t c5_part[5];   // bytes of the chunk packs; floats of a slot's partial buffers (0: none)
};
constexpr int C5_NT = 4;   // 64-column N tiles: 6 jobs x 128 pixel tiles of unequal K balance over the chip (see convq.hip)

// VPX_OPT_EXPERIMENT bit 6 keeps the first-generation weight-gradient launches (A/B runs, tests)
bool stw_applicable(const vpx_stlstm_desc* d) {
    return d->k == 5 && d->precision == VPX_PREC_BF16X3 && !(d->Ch & 7) && !(d->Cin & 7) && !d->layer_norm && !(g_experiment & 64);
}

int mk_dg(STBwdLayout::DG& g, const int* segC, int nseg, int k, int n_out, int prec, long long m_tiles) {
    const int taps = k * k;
    g.ng = plain_groups(n_out, m_tiles);
    g.mw = pick_mw_tiles(m_tiles, plain_tiles_ng(n_out, g.ng), prec);  // 8-wave form when the grid stays large
    if (g.mw > 1 && !conv_fits_lds(segC, nseg, k, k, g.ng, prec, g.mw)) g.mw = 1;   // (... and fits LDS)
    if (!conv_fits_lds(segC, nseg, k, k, g.ng, prec, g.mw)) return -1;
    g.nstage = build_stages(g.stage, &g.chunks, segC, nseg, taps, pick_stage_channels(segC, nseg, k, k, g.ng, prec, g.mw), prec);
    if (g.nstage < 0) return -1;
    g.tiles = plain_tiles_ng(n_out, g.ng);
    g.ksplit = pick_ksplit(m_tiles * g.tiles, g.nstage, true);  // 16x16 maps: 64 pixel tiles per launch, K = gates*Ch*k*k is long
    g.wpk = packed_weight_bytes(g.tiles, g.chunks, g.ng, prec) / 4;
    return 0;
}

int st_bwd_layout(const vpx_stlstm_desc* d, STBwdLayout& L) {
    const int Ch = d->Ch, Cin = d->Cin;
    L.taps = d->k * d->k; L.Ch = Ch; L.Cin = Cin;
    L.n_state = (size_t)d->B * d->H * d->W * Ch;
    L.n_x = (size_t)d->B * d->H * d->W * Cin;
    L.n_g7 = L.n_state * (stw_applicable(d) ? 8 : 7);   // the one-launch weight gradient keeps d conv_last as an eighth block
    const int s1[1] = {Ch}, s3[3] = {3 * Ch, Ch, 3 * Ch}, s4[1] = {4 * Ch}, s3m[1] = {3 * Ch};
    const int pr = d->precision;
    const long long mt = (long long)d->B * ((d->W + TILE_W - 1) / TILE_W) * ((d->H + TILE_H - 1) / TILE_H);
    if (mk_dg(L.o, s1, 1, d->k, 2 * Ch, pr, mt) || mk_dg(L.l, s1, 1, 1, 2 * Ch, pr, mt) || mk_dg(L.x, s3, 3, d->k, Cin, pr, mt) ||
        mk_dg(L.h, s4, 1, d->k, Ch, pr, mt) || mk_dg(L.m, s3m, 1, d->k, Ch, pr, mt)) {
        set_error("stlstm bwd: too many channel stages (Ch=%d)", Ch);
        return VPX_ERR_UNSUPPORTED;
    }
    const int tiles = ((d->W + TILE_W - 1) / TILE_W) * ((d->H + TILE_H - 1) / TILE_H);
    long long items = (long long)d->B * tiles;
    L.n_slices = (int)(items < 32 ? items : 32);
    // largest weight-gradient tensor (elements) among Wx, Wh, Wm, Wo, Wlast
    size_t mx = (size_t)7 * Ch * Cin * L.taps;
    if ((size_t)4 * Ch * Ch * L.taps > mx) mx = (size_t)4 * Ch * Ch * L.taps;
    if ((size_t)2 * Ch * Ch * L.taps > mx) mx = (size_t)2 * Ch * Ch * L.taps;
    L.slab_floats = mx * L.n_slices;
    L.stw = stw_applicable(d); L.stw_pairs = 0; L.stw_ns = 0;
    if (L.stw) {
        static thread_local STWArgs sa; static thread_local STWOut so;
        L.stw_pairs = stw_build(sa, so, d->B, d->H, d->W, Cin, Ch);
        if (L.stw_pairs < 1) L.stw = false;
        else {
            L.stw_ns = stw_slices(sa.npairs5, (long long)d->B * ((d->W + 15) / 16) * ((d->H + 3) / 4));
            const size_t need = sa.slab_stride * L.stw_ns;
            if (need > L.slab_floats) L.slab_floats = need;
        }
    }
    // VPX_OPT_EXPERIMENT bit 7 keeps the first-generation data-gradient launches (A/B runs, tests)
    // From 96 pixel tiles of 16x16 on, a job runs its whole K (the grid rule of the forward launches, stlstm_api.hip; measured training
    // step at 16x16 maps, unsplit c5 vs first-generation K-split data gradients: B = 8 130 vs 78 ms, B = 32 165 vs 136 ms, B = 128 361
    // vs ~405 ms); below, the K of every slot is cut into chunks (bit 11 of VPX_OPT_EXPERIMENT keeps the first generation there).
    const long long mt16 = (long long)d->B * ((d->H + 15) / 16) * ((d->W + 15) / 16);
    const bool big = mt16 >= dev_switch("VPX_C5B_MIN_TILES", 96) || (g_experiment & 1024);
    L.c5 = L.stw && !(g_experiment & 128) && (big || !(g_experiment & 2048));
    if (L.c5) {
        const int K[5] = {Ch, Ch, 7 * Ch, 4 * Ch, 3 * Ch}, Co[5] = {Ch, Ch, Cin, Ch, Ch};
        const int ks_small[5] = {4, 4, 6, 3, 3}, ks_mid[5] = {2, 2, 4, 2, 2};
        for (int i = 0; i < 5; ++i) {
            int ks = big ? 1 : (mt16 <= 16 ? ks_small[i] : ks_mid[i]);
            if (ks > K[i] / 32) ks = K[i] / 32 > 0 ? K[i] / 32 : 1;   // at least four 8-channel stages per chunk
            L.c5_ks[i] = ks;
            for (int k = 0; k < ks; ++k) L.c5_wpk[i][k] = align256(c5_chunk_wpk_bytes(K[i], k, ks, Co[i], C5_NT));
            L.c5_part[i] = ks > 1 ? (size_t)ks * (i == 2 ? L.n_x : L.n_state) : 0;
        }
    }
    return VPX_OK;
}

}  // namespace

namespace vpx {
size_t stlstm_bwd_workspace_bytes(const vpx_stlstm_desc* d) {
    STBwdLayout L;
    if (st_bwd_layout(d, L) != VPX_OK) return 0;
    size_t b = align256(L.n_g7 * 4) + 4 * align256(L.n_state * 4);  // dG7, dlc, dcn_conv, dmn_conv, dm scratch
    b += align256(L.o.wpk * 4) + align256(L.l.wpk * 4) + align256(L.x.wpk * 4) + align256(L.h.wpk * 4) + align256(L.m.wpk * 4);
    b += align256(L.slab_floats * 4);
    if (L.stw) b += align256(L.n_x * 4) + 4 * align256(L.n_state * 4);   // split copies: x, h, m, c_new, m_new
    if (L.c5) for (int i = 0; i < 5; ++i) { for (int k = 0; k < L.c5_ks[i]; ++k) b += L.c5_wpk[i][k]; b += align256(L.c5_part[i] * 4); }
    if (d->layout == VPX_LAYOUT_NCHW) b += 2 * align256(L.n_x * 4) + 14 * align256(L.n_state * 4);
    return b;
}
}  // namespace vpx

namespace {

// one weight gradient: dW[n_out, C0 + C1, kh, kw] from dG (channel slice [N] of a [.., ldG] tensor) and up to two sources
int run_wgrad(const vpx_stlstm_desc* d, const STBwdLayout& L, const float* dG, int N, int ldG, const float* src0, int C0,
              const float* src1, int C1, int k, const int* rowblk, int blk, int n_out, float* slabs, float* dW,
              hipStream_t stream) {
    WgradArgs wa{};
    wa.T = 1; wa.B = d->B; wa.H = d->H; wa.W = d->W; wa.HW = d->H * d->W; wa.kh = k; wa.kw = k;
    wa.tiles_x = (d->W + TILE_W - 1) / TILE_W; wa.tiles_y = (d->H + TILE_H - 1) / TILE_H;
    wa.N4 = N; wa.Cin = C0; wa.Ch = C1 > 0 ? C1 : 1; wa.Ct = C0 + C1;
    wa.ldG = ldG; wa.blk = blk; wa.n_out = n_out; wa.prec = d->precision;
    if (rowblk) for (int i = 0; i < 8; ++i) wa.rowblk[i] = rowblk[i];
    wa.dG = dG;
    wa.x = src0; wa.x_bstride = (long long)wa.HW * C0; wa.x_tstride = 0;
    wa.hseq = nullptr; wa.h0 = src1;
    wa.n_ctiles = wgrad_make_ctiles(wa.ct, WG_MAX_CTILES, C0, C1, C0);
    if (wa.n_ctiles < 0) { set_error("stlstm bwd: too many channels for the weight-gradient kernel"); return VPX_ERR_UNSUPPORTED; }
    wa.slabs = slabs;
    const int taps = k * k;
    // K slices: enough for ~1024 workgroups, no more (every slice costs a slab write + a reduce read of the whole dW)
    const int out_tiles = ((N + 63) / 64) * wa.n_ctiles * ((taps + 8) / 9);
    int ns = wgrad_target_wgs() / out_tiles;   // rounded down: whole rounds of workgroups (see convlstm_layout)
    if (ns > L.n_slices) ns = L.n_slices;
    if (ns < 1) ns = 1;
    // (no slab clear: every workgroup of every slice stores its full tile, so each slab element is written once)
    VPX_CHECK_HIP(launch_wgrad(wa, ns, stream));
    VPX_CHECK_HIP(launch_wgrad_reduce(slabs, dW, ns, taps, n_out, wa.Ct, stream));
    return VPX_OK;
}

}  // namespace

extern "C" int vpx_stlstm_step_bwd(const vpx_stlstm_desc* d, const float* x, const float* h, const float* c,
                                   const float* m, const float* c_new, const float* m_new, const float* Wx,
                                   const float* Wh, const float* Wm, const float* Wo, const float* Wlast,
                                   const float* const* ln, const void* reserve, size_t reserve_bytes,
                                   const float* dh_new, const float* dc_new, const float* dm_new, const float* ddelta_c,
                                   const float* ddelta_m, float* dx, float* dh, float* dc, float* dm, float* dWx,
                                   float* dWh, float* dWm, float* dWo, float* dWlast, float* const* dln, void* workspace,
                                   size_t workspace_bytes, void* stream_) {
    return vpx_stlstm_step_bwd_ex(d, x, h, c, m, c_new, m_new, Wx, Wh, Wm, Wo, Wlast, ln, reserve, reserve_bytes, dh_new, dc_new, dm_new,
                                  ddelta_c, ddelta_m, dx, dh, dc, dm, dWx, dWh, dWm, dWo, dWlast, dln, workspace, workspace_bytes, stream_, nullptr);
}

extern "C" int vpx_stlstm_step_bwd_ex(const vpx_stlstm_desc* d, const float* x, const float* h, const float* c,
                                      const float* m, const float* c_new, const float* m_new, const float* Wx,
                                      const float* Wh, const float* Wm, const float* Wo, const float* Wlast,
                                      const float* const* ln, const void* reserve, size_t reserve_bytes,
                                      const float* dh_new, const float* dc_new, const float* dm_new, const float* ddelta_c,
                                      const float* ddelta_m, float* dx, float* dh, float* dc, float* dm, float* dWx,
                                      float* dWh, float* dWm, float* dWo, float* dWlast, float* const* dln, void* workspace,
                                      size_t workspace_bytes, void* stream_, const vpx_stlstm_shadows* shadows) {
    STSplitShadows sh = st_shadows_of(shadows);   // an argument of THIS call
    if (!d) { set_error("stlstm desc is NULL"); return VPX_ERR_ARG; }
    if (d->layout != VPX_LAYOUT_NHWC) {
        // the split-format shadows (and the deferred weight gradient's dG8 slot with them) exist for NHWC callers only: a request for a
        // deferred weight gradient on reference-layout buffers is refused, not dropped (a caller who passed NULL dW pointers for it would
        // get VPX_OK and no weight gradient at all)
        if (shadows && shadows->dg8_out) { set_error("vpx_stlstm_step_bwd_ex: dg8_out (deferred weight gradients) needs VPX_LAYOUT_NHWC"); return VPX_ERR_UNSUPPORTED; }
        sh = STSplitShadows{};
    }
    if ((d->precision < VPX_PREC_F32 || d->precision > VPX_PREC_BF16)) { set_error("stlstm: precision %d not implemented", d->precision); return VPX_ERR_UNSUPPORTED; }
    if (!(d->flags & VPX_FLAG_SAVE_FOR_BWD)) { set_error("vpx_stlstm_step_bwd: desc lacks VPX_FLAG_SAVE_FOR_BWD"); return VPX_ERR_ARG; }
    if (!x || !h || !c || !m || !c_new || !m_new || !Wx || !Wh || !Wm || !Wo || !Wlast || !reserve) {
        set_error("vpx_stlstm_step_bwd: NULL tensor argument");
        return VPX_ERR_ARG;
    }
    STBwdLayout L;
    int rc = st_bwd_layout(d, L);
    if (rc != VPX_OK) return rc;
    if (reserve_bytes < vpx_stlstm_reserve_bytes(d)) { set_error("vpx_stlstm_step_bwd: reserve too small"); return VPX_ERR_WORKSPACE; }
    if (!workspace || workspace_bytes < vpx_stlstm_workspace_bytes(d)) { set_error("vpx_stlstm_step_bwd: workspace too small"); return VPX_ERR_WORKSPACE; }
    hipStream_t stream = (hipStream_t)stream_;
    const int B = d->B, Cin = d->Cin, Ch = d->Ch, H = d->H, Wd = d->W, k = d->k;
    const size_t HW = (size_t)H * Wd;
    vpx_stlstm_desc d2;
    if (d->layer_norm) {
        if (!ln) { set_error("vpx_stlstm_step_bwd: layer_norm set but ln is NULL"); return VPX_ERR_ARG; }
        Carver w2(workspace, workspace_bytes);
        if (d->layout == VPX_LAYOUT_NHWC)
            return stlstm_ln_bwd(d, x, h, c, m, Wx, Wh, Wm, Wo, Wlast, ln, reserve, dh_new, dc_new, dm_new, ddelta_c, ddelta_m,
                                 dx, dh, dc, dm, dWx, dWh, dWm, dWo, dWlast, dln, w2, stream);
        // reference layout (NCHW): activations and incoming gradients are transposed into the workspace, the NHWC routine
        // runs on the copies, the data gradients are transposed back (the LayerNorm parameters and their gradients keep the
        // reference's [C,H,W] layout either way)
        const size_t n_x = (size_t)B * HW * Cin, n_s = (size_t)B * HW * Ch;
        float* bx = w2.take(n_x);
        float* bdx = w2.take(n_x);
        float* st[12];
        for (auto& p : st) p = w2.take(n_s);
        VPX_CHECK_CARVE(w2, "vpx_stlstm_step_bwd (LayerNorm, NCHW)");
        const float* in_nchw[9] = {h, c, m, dh_new, dc_new, dm_new, ddelta_c, ddelta_m, nullptr};
        const float* in_nhwc[9] = {};
        VPX_CHECK_HIP(launch_nchw_to_nhwc(x, bx, B, Cin, H, Wd, stream));
        for (int i = 0; i < 8; ++i)
            if (in_nchw[i]) { VPX_CHECK_HIP(launch_nchw_to_nhwc(in_nchw[i], st[i], B, Ch, H, Wd, stream)); in_nhwc[i] = st[i]; }
        float* dhn = dh ? st[8] : nullptr; float* dcn2 = dc ? st[9] : nullptr; float* dmn2 = dm ? st[10] : nullptr;
        d2 = *d; d2.layout = VPX_LAYOUT_NHWC;
        rc = stlstm_ln_bwd(&d2, bx, in_nhwc[0], in_nhwc[1], in_nhwc[2], Wx, Wh, Wm, Wo, Wlast, ln, reserve, in_nhwc[3], in_nhwc[4],
                           in_nhwc[5], in_nhwc[6], in_nhwc[7], dx ? bdx : nullptr, dhn, dcn2, dmn2, dWx, dWh, dWm, dWo, dWlast, dln, w2, stream);
        if (rc != VPX_OK) return rc;
        if (dx) VPX_CHECK_HIP(launch_nhwc_to_nchw(bdx, dx, B, Cin, H, Wd, stream));
        if (dh) VPX_CHECK_HIP(launch_nhwc_to_nchw(dhn, dh, B, Ch, H, Wd, stream));
        if (dc) VPX_CHECK_HIP(launch_nhwc_to_nchw(dcn2, dc, B, Ch, H, Wd, stream));
        if (dm) VPX_CHECK_HIP(launch_nhwc_to_nchw(dmn2, dm, B, Ch, H, Wd, stream));
        return VPX_OK;
    }

    // VPX_FLAG_WEIGHTS_PACKED: `workspace` is the one a previous backward call of the SAME cell, weights, shape and set of
    // requested data gradients left behind — its five transposed weight packs are reused (PredRNN: 57 cell steps per
    // training step share four cells' weights)
    const bool packed = (d->flags & VPX_FLAG_WEIGHTS_PACKED) != 0;
    Carver ws(workspace, workspace_bytes);
    float* dG7 = ws.take(L.n_g7);
    // deferred weight gradients (vpx_stlstm_shadows::dg8_out): this step's dG8 goes to the caller's slab slot in the split format and the
    // five weight gradients are left to ONE vpx_stlstm_wgrad_batch call over all the steps of the cell
    const bool defer = sh.dg8 != nullptr;
    if (defer && !L.stw) { set_error("vpx_stlstm_step_bwd_ex: dg8_out given, but this descriptor's weight gradients cannot be deferred (vpx_stlstm_defers_wgrad)"); return VPX_ERR_UNSUPPORTED; }
    if (defer) dG7 = reinterpret_cast<float*>(sh.dg8);
    float* dlc = ws.take(L.n_state);
    float* dcn_conv = ws.take(L.n_state);
    float* dmn_conv = ws.take(L.n_state);
    float* dm_scratch = ws.take(L.n_state);
    float* wpk_o = ws.take(L.o.wpk);
    float* wpk_l = ws.take(L.l.wpk);
    float* wpk_x = ws.take(L.x.wpk);
    float* wpk_h = ws.take(L.h.wpk);
    float* wpk_m = ws.take(L.m.wpk);
    float* slabs = ws.take(L.slab_floats);
    char *x_sp = nullptr, *st_sp[4] = {nullptr, nullptr, nullptr, nullptr};
    if (L.stw) {
        x_sp = (char*)ws.take(L.n_x);
        for (auto& q : st_sp) q = (char*)ws.take(L.n_state);
    }
    char* c5w[5][6] = {};
    float* c5p[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    if (L.c5) for (int i = 0; i < 5; ++i) {
        for (int k = 0; k < L.c5_ks[i]; ++k) c5w[i][k] = (char*)ws.take(L.c5_wpk[i][k] / 4);
        if (L.c5_part[i]) c5p[i] = ws.take(L.c5_part[i]);
    }

    VPX_CHECK_CARVE(ws, "vpx_stlstm_step_bwd");
    const float *xn = x, *hn = h, *cn = c, *mn = m, *cnn = c_new, *mnn = m_new;
    const float *g_h = dh_new, *g_c = dc_new, *g_m = dm_new, *g_dc = ddelta_c, *g_dm = ddelta_m;
    float *dxn = dx, *dhn = dh, *dcn = dc, *dmn = dm;
    if (d->layout == VPX_LAYOUT_NCHW) {
        float* bx = ws.take(L.n_x);
        float* bdx = ws.take(L.n_x);
        float* st[14];
        for (auto& p : st) p = ws.take(L.n_state);
        VPX_CHECK_CARVE(ws, "vpx_stlstm_step_bwd");
        VPX_CHECK_HIP(launch_nchw_to_nhwc(x, bx, B, Cin, H, Wd, stream)); xn = bx;
        const float* ins[10] = {h, c, m, c_new, m_new, dh_new, dc_new, dm_new, ddelta_c, ddelta_m};
        const float** outs[10] = {&hn, &cn, &mn, &cnn, &mnn, &g_h, &g_c, &g_m, &g_dc, &g_dm};
        for (int i = 0; i < 10; ++i)
            if (ins[i]) { VPX_CHECK_HIP(launch_nchw_to_nhwc(ins[i], st[i], B, Ch, H, Wd, stream)); *outs[i] = st[i]; }
        if (dx) dxn = bdx;
        if (dh) dhn = st[10];
        if (dc) dcn = st[11];
        if (dm) dmn = st[12];
    }
    if (!dmn) dmn = dm_scratch;

    const char* r = (const char*)reserve;
    const float* gates_c = (const float*)r; r += align256(3 * L.n_state * 4);
    const float* gates_m = (const float*)r; r += align256(3 * L.n_state * 4);
    const float* o_save = (const float*)r; r += align256(L.n_state * 4);
    const float* tl_save = (const float*)r;

    auto plan_for = [&](const STBwdLayout::DG& g, int kk, const float* wpk) {
        ConvPlan P{};
        P.prec = d->precision;
        P.B = B; P.H = H; P.W = Wd; P.kh = kk; P.kw = kk;
        set_plan_tiles(P, g.mw);
        P.nstage = g.nstage; memcpy(P.stage, g.stage, sizeof(ConvStage) * g.nstage);
        P.chunks_total = g.chunks; P.a_bytes = conv_a_bytes(g.stage, g.nstage, kk, kk, g.mw); P.wpk = wpk;
        return P;
    };
    auto pack_plain_T = [&](PackDesc& pd, const STBwdLayout::DG& g, int taps, int n_out) {
        memcpy(pd.stage, g.stage, sizeof(ConvStage) * g.nstage);
        pd.nstage = g.nstage; pd.chunks_total = g.chunks; pd.prec = d->precision; pd.taps = taps;
        fill_plain_pack(pd, n_out, 0, g.ng);
        pd.transposed = 1; pd.flip = 1;
    };

    // K-split data gradients add their partial sums atomically: destinations that are not accumulated into start at zero
    auto split_plan = [&](ConvPlan& P, const STBwdLayout::DG& g, float* o0, float* o1, size_t n, bool accumulate) -> hipError_t {
        P.ksplit = g.ksplit;
        if (g.ksplit > 1 && !accumulate) {
            hipError_t e = vpx_memset_async(o0, 0, n * sizeof(float), stream);
            if (e == hipSuccess && o1) e = vpx_memset_async(o1, 0, n * sizeof(float), stream);
            return e;
        }
        return hipSuccess;
    };
    // The one-launch weight gradient wants dG7 in the split operand format: the two pointwise stages then write that format ONLY
    // (the data-gradient convolutions read it without conversion, ConvSeg.split) — no fp32 dG7, no conversion pass.
    const bool stw = L.stw && (defer || (dWx && dWh && dWm && dWo && dWlast));
    const int g7s = stw ? 1 : 0;
    const int ldG = stw ? 8 * Ch : 7 * Ch;   // stw: dG8 = the seven gate blocks + d conv_last (block 7), all in the split format
    // ---- A: through h_new = o * tanh(conv_last(mem)) ----
    {
        // (stw: d conv_last goes to dG8 block 7 in the split format for the weight gradient AND to the fp32 tensor dlc for the streaming 1x1 adjoint)
        STBwdOutArgs a{(long long)L.n_state, Ch, ldG, 3 * Ch, stw ? 7 * Ch : -1, g_h, o_save, tl_save, dG7, dlc, g7s};
        VPX_CHECK_HIP(launch_st_bwd_out(a, stream));
    }
    // ---- B: grads of mem = [c_new | m_new] through conv_o (k x k) and conv_last (1 x 1) ----
    const bool c5 = L.c5 && stw;
    C5Plan cp{};
    cp.B = B; cp.H = H; cp.W = Wd;
    cp.src[0] = C5Src{reinterpret_cast<const char*>(dG7), (long long)HW * ldG * 4, ldG * 4, 0};
    // rng[i] = {source channel 0, channels, weight row 0}: the job's K = channel ranges of dG8, each multiplying rows of `w`
    struct Pending { float* out; const float* part; long long n; int ks, acc; } pend[5];
    int npend = 0;
    auto c5_job = [&](int slot, int Co_, float* out, int ld_out, int acc, const float* w, long long s_row, int w_col0, int nrange,
                      const int (*rng)[3]) -> int {
        C5Job full{};
        full.nrange = nrange;
        C5PackRange prf[3] = {};
        for (int i = 0; i < nrange; ++i) {
            full.r_src[i] = 0; full.r_c0[i] = rng[i][0]; full.r_n[i] = rng[i][1];
            prf[i] = C5PackRange{w, (long long)L.taps, s_row, rng[i][2], {w_col0, 0, 0, 0}};   // data gradient: column = the weight's input channel
        }
        full.Co = Co_; full.ld = ld_out; full.accumulate = acc; full.out = out; full.out_bstride = (long long)HW * ld_out;
        const int ks = L.c5_ks[slot];
        if (ks == 1) {
            C5Job& j = cp.job[cp.njobs++];
            j = full; j.wpk = c5w[slot][0];
            return c5_prepare_job(j, C5_NT, prf, 0, 1, packed, stream);
        }
        const long long n = (long long)B * (long long)HW * ld_out;
        for (int kk = 0; kk < ks; ++kk) {   // chunk jobs write partial sums; sum_partials_kernel adds them (onto `out` when acc)
            C5Job& j = cp.job[cp.njobs++];
            C5PackRange pr[3];
            c5_chunk_job(full, prf, kk, ks, j, pr);
            j.wpk = c5w[slot][kk]; j.accumulate = 0; j.out = c5p[slot] + (size_t)kk * n;
            const int rcj = c5_prepare_job(j, C5_NT, pr, 0, 1, packed, stream);
            if (rcj) return rcj;
        }
        pend[npend++] = Pending{out, c5p[slot], n, ks, acc};
        return VPX_OK;
    };
    auto c5_flush = [&]() -> int {
        if (cp.njobs) VPX_CHECK_HIP(launch_c5(cp, C5_NT, stream));
        {   // the slots' partial sums in ONE launch (round 5: five 5 us launches per cell step on the small grids)
            float* po[5]; const float* pp[5]; long long pn[5]; int pk[5], pa[5];
            bool vec = true;
            for (int i = 0; i < npend; ++i) { po[i] = pend[i].out; pp[i] = pend[i].part; pn[i] = pend[i].n; pk[i] = pend[i].ks; pa[i] = pend[i].acc; vec = vec && !(pend[i].n & 3); }
            if (vec) VPX_CHECK_HIP(launch_sum_partials_multi(npend, po, pp, pn, pk, pa, stream));
            else for (int i = 0; i < npend; ++i) VPX_CHECK_HIP(launch_sum_partials(pend[i].out, pend[i].part, pend[i].n, pend[i].ks, pend[i].n, pend[i].acc, stream));
        }
        cp.njobs = 0; npend = 0;
        return VPX_OK;
    };
    if (c5) {
        const int ro[1][3] = {{3 * Ch, Ch, 0}};
        if ((rc = c5_job(0, Ch, dcn_conv, Ch, 0, Wo, (long long)2 * Ch * L.taps, 0, 1, ro))) return rc;
        if ((rc = c5_job(1, Ch, dmn_conv, Ch, 0, Wo, (long long)2 * Ch * L.taps, Ch, 1, ro))) return rc;
        if ((rc = c5_flush())) return rc;
    }
    {
        PackDesc pd{};
        pd.seg[0] = PackSeg{Wo, (long long)2 * Ch * L.taps, L.taps, 0, Ch};
        pack_plain_T(pd, L.o, L.taps, 2 * Ch);
        if (!packed && !c5) VPX_CHECK_HIP(launch_pack_weights(pd, wpk_o, stream));
        ConvPlan P = plan_for(L.o, k, wpk_o);
        P.nseg = 1; P.seg[0] = ConvSeg{dG7 + 3 * Ch, (long long)(HW * ldG), Ch, ldG, g7s};
        PlainEpiArgs ea{};
        ea.Co = 2 * Ch; ea.split = Ch; ea.ng = L.o.ng;
        ea.out0 = dcn_conv; ea.bstride0 = (long long)(HW * Ch); ea.ld0 = Ch;
        ea.out1 = dmn_conv; ea.bstride1 = (long long)(HW * Ch); ea.ld1 = Ch;
        if (!c5) {
            VPX_CHECK_HIP(split_plan(P, L.o, dcn_conv, dmn_conv, L.n_state, false));
            VPX_CHECK_HIP(launch_conv_plain_f32(P, ea, L.o.tiles, stream));
        }

        C1Args c1{};   // conv_last's adjoint: [dc_new | dm_new] += Wlast^T d conv_last
        c1.x[0] = dlc; c1.xld[0] = Ch; c1.xc[0] = Ch;
        c1.npix = (long long)B * (long long)HW;
        c1.w = Wlast; c1.w_sn = 1; c1.w_sc = 2 * Ch;
        c1.y[0] = dcn_conv; c1.y[1] = dmn_conv; c1.yld[0] = c1.yld[1] = Ch; c1.ysplit = Ch; c1.Co = 2 * Ch; c1.accumulate = 1;
        if (c1_applicable(c1, d->precision)) {
            VPX_CHECK_HIP(launch_c1(c1, stream));   // streaming form (conv1.hip)
        } else {
        PackDesc pl{};
        pl.seg[0] = PackSeg{Wlast, (long long)2 * Ch, 1, 0, Ch};
        pack_plain_T(pl, L.l, 1, 2 * Ch);
        if (!packed) VPX_CHECK_HIP(launch_pack_weights(pl, wpk_l, stream));
        ConvPlan Q = plan_for(L.l, 1, wpk_l);
        Q.nseg = 1; Q.seg[0] = stw ? ConvSeg{dG7 + 7 * Ch, (long long)(HW * ldG), Ch, ldG, 1} : ConvSeg{dlc, (long long)(HW * Ch), Ch, 0};
        ea.accumulate = 1;
        VPX_CHECK_HIP(split_plan(Q, L.l, dcn_conv, dmn_conv, L.n_state, true));
        VPX_CHECK_HIP(launch_conv_plain_f32(Q, ea, L.l.tiles, stream));
        }
    }
    // ---- C: gate groups ----
    {
        STBwdGateArgs a{};
        a.npix = (long long)B * HW; a.Ch = Ch; a.ldG = ldG;
        a.gates_c = gates_c; a.gates_m = gates_m; a.c = cn; a.m = mn;
        a.dcn_ext = g_c; a.dmn_ext = g_m; a.ddc_ext = g_dc; a.ddm_ext = g_dm;
        a.dcn_conv = dcn_conv; a.dmn_conv = dmn_conv;
        a.dG7 = dG7; a.dc = dcn; a.dm = dmn; a.split = g7s;
        VPX_CHECK_HIP(launch_st_bwd_gates(a, stream));
    }
    // ---- D: data gradients ----
    if (c5) {   // dx | dh | dm as the jobs of ONE launch, longest K first
        cp.njobs = 0;
        if (dxn) {
            const int rx[3][3] = {{0, 3 * Ch, 0}, {3 * Ch, Ch, 6 * Ch}, {4 * Ch, 3 * Ch, 3 * Ch}};   // dG8 blocks (i,f,g | o | i',f',g') <-> Wx row blocks 0-2 | 6 | 3-5
            if ((rc = c5_job(2, Cin, dxn, Cin, 0, Wx, (long long)Cin * L.taps, 0, 3, rx))) return rc;
        }
        if (dhn) {
            const int rh[1][3] = {{0, 4 * Ch, 0}};
            if ((rc = c5_job(3, Ch, dhn, Ch, 0, Wh, (long long)Ch * L.taps, 0, 1, rh))) return rc;
        }
        if (dm) {
            const int rm[1][3] = {{4 * Ch, 3 * Ch, 0}};
            if ((rc = c5_job(4, Ch, dmn, Ch, 1, Wm, (long long)Ch * L.taps, 0, 1, rm))) return rc;   // onto dm_new_total * f' written by stage C
        }
        if ((rc = c5_flush())) return rc;
    }
    if (!c5 && dxn) {
        PackDesc pd{};
        const long long ldo = (long long)Cin * L.taps;
        pd.seg[0] = PackSeg{Wx, ldo, L.taps, 0, 3 * Ch};        // dG7 blocks (i,f,g)     <-> Wx row blocks 0,1,2
        pd.seg[1] = PackSeg{Wx, ldo, L.taps, 6 * Ch, Ch};       // dG7 block  o           <-> Wx row block 6
        pd.seg[2] = PackSeg{Wx, ldo, L.taps, 3 * Ch, 3 * Ch};   // dG7 blocks (i',f',g')  <-> Wx row blocks 3,4,5
        pack_plain_T(pd, L.x, L.taps, Cin);
        if (!packed) VPX_CHECK_HIP(launch_pack_weights(pd, wpk_x, stream));
        ConvPlan P = plan_for(L.x, k, wpk_x);
        P.nseg = 3;
        P.seg[0] = ConvSeg{dG7, (long long)(HW * ldG), 3 * Ch, ldG, g7s};
        P.seg[1] = ConvSeg{dG7 + 3 * Ch, (long long)(HW * ldG), Ch, ldG, g7s};
        P.seg[2] = ConvSeg{dG7 + 4 * Ch, (long long)(HW * ldG), 3 * Ch, ldG, g7s};
        PlainEpiArgs ea{};
        ea.Co = Cin; ea.split = Cin; ea.ng = L.x.ng; ea.out0 = dxn; ea.bstride0 = (long long)(HW * Cin); ea.ld0 = Cin;
        VPX_CHECK_HIP(split_plan(P, L.x, dxn, nullptr, L.n_x, false));
        VPX_CHECK_HIP(launch_conv_plain_f32(P, ea, L.x.tiles, stream));
    }
    if (!c5 && dhn) {
        PackDesc pd{};
        pd.seg[0] = PackSeg{Wh, (long long)Ch * L.taps, L.taps, 0, 4 * Ch};
        pack_plain_T(pd, L.h, L.taps, Ch);
        if (!packed) VPX_CHECK_HIP(launch_pack_weights(pd, wpk_h, stream));
        ConvPlan P = plan_for(L.h, k, wpk_h);
        P.nseg = 1; P.seg[0] = ConvSeg{dG7, (long long)(HW * ldG), 4 * Ch, ldG, g7s};
        PlainEpiArgs ea{};
        ea.Co = Ch; ea.split = Ch; ea.ng = L.h.ng; ea.out0 = dhn; ea.bstride0 = (long long)(HW * Ch); ea.ld0 = Ch;
        VPX_CHECK_HIP(split_plan(P, L.h, dhn, nullptr, L.n_state, false));
        VPX_CHECK_HIP(launch_conv_plain_f32(P, ea, L.h.tiles, stream));
    }
    if (!c5 && dm) {
        PackDesc pd{};
        pd.seg[0] = PackSeg{Wm, (long long)Ch * L.taps, L.taps, 0, 3 * Ch};
        pack_plain_T(pd, L.m, L.taps, Ch);
        if (!packed) VPX_CHECK_HIP(launch_pack_weights(pd, wpk_m, stream));
        ConvPlan P = plan_for(L.m, k, wpk_m);
        P.nseg = 1; P.seg[0] = ConvSeg{dG7 + 4 * Ch, (long long)(HW * ldG), 3 * Ch, ldG, g7s};
        PlainEpiArgs ea{};
        ea.Co = Ch; ea.split = Ch; ea.ng = L.m.ng; ea.out0 = dmn; ea.bstride0 = (long long)(HW * Ch); ea.ld0 = Ch;
        ea.accumulate = 1;  // onto dm_new_total * f' written by stage C
        VPX_CHECK_HIP(split_plan(P, L.m, dmn, nullptr, L.n_state, true));
        VPX_CHECK_HIP(launch_conv_plain_f32(P, ea, L.m.tiles, stream));
    }
    // ---- E: weight gradients ----
    if (stw && !defer) {
        // all four k x k tensors in one launch (wgrad2.hip, stw): operands once more in the split format
        static thread_local STWArgs sa; static thread_local STWOut so;
        if (stw_build(sa, so, B, H, Wd, Cin, Ch) != L.stw_pairs) { set_error("stlstm bwd: pair table changed"); return VPX_ERR_ARG; }
        const long long npix = (long long)B * (long long)HW;
        if (sh.in[0]) x_sp = const_cast<char*>(sh.in[0]); else VPX_CHECK_HIP(launch_split_convert(xn, x_sp, npix, Cin, stream));
        const float* st_src[4] = {hn, mn, cnn, mnn};
        for (int i = 0; i < 4; ++i) {   // the forward's split shadows of h, m, c_new, m_new where the caller kept them
            if (sh.in[1 + i]) st_sp[i] = const_cast<char*>(sh.in[1 + i]);
            else VPX_CHECK_HIP(launch_split_convert(st_src[i], st_sp[i], npix, Ch, stream));
        }
        sa.g_sp = reinterpret_cast<const char*>(dG7);   // written in the split format by stages A and C
        sa.src[0] = STWSrc{x_sp, Cin};
        for (int i = 0; i < 4; ++i) sa.src[1 + i] = STWSrc{st_sp[i], Ch};
        sa.n_slices = L.stw_ns; sa.slabs = slabs;
        so.dW[0] = dWx; so.dW[1] = dWh; so.dW[2] = dWm; so.dW[3] = dWo; so.dW[4] = dWlast;
        VPX_CHECK_HIP(launch_stw(sa, so, stream));
    }
    if (!stw && dWo && (rc = run_wgrad(d, L, dG7 + 3 * Ch, Ch, ldG, cnn, Ch, mnn, Ch, k, nullptr, 0, Ch, slabs, dWo, stream))) return rc;
    if (!stw && dWlast && (rc = run_wgrad(d, L, dlc, Ch, Ch, cnn, Ch, mnn, Ch, 1, nullptr, 0, Ch, slabs, dWlast, stream))) return rc;
    if (!stw && dWx) {
        const int rowblk[8] = {0, 1, 2, 6, 3, 4, 5, 0};
        if ((rc = run_wgrad(d, L, dG7, 7 * Ch, ldG, xn, Cin, nullptr, 0, k, rowblk, Ch, 7 * Ch, slabs, dWx, stream))) return rc;
    }
    if (!stw && dWh && (rc = run_wgrad(d, L, dG7, 4 * Ch, ldG, hn, Ch, nullptr, 0, k, nullptr, 0, 4 * Ch, slabs, dWh, stream))) return rc;
    if (!stw && dWm && (rc = run_wgrad(d, L, dG7 + 4 * Ch, 3 * Ch, ldG, mn, Ch, nullptr, 0, k, nullptr, 0, 3 * Ch, slabs, dWm, stream))) return rc;

    if (d->layout == VPX_LAYOUT_NCHW) {
        if (dx) VPX_CHECK_HIP(launch_nhwc_to_nchw(dxn, dx, B, Cin, H, Wd, stream));
        if (dh) VPX_CHECK_HIP(launch_nhwc_to_nchw(dhn, dh, B, Ch, H, Wd, stream));
        if (dc) VPX_CHECK_HIP(launch_nhwc_to_nchw(dcn, dc, B, Ch, H, Wd, stream));
        if (dm) VPX_CHECK_HIP(launch_nhwc_to_nchw(dmn, dm, B, Ch, H, Wd, stream));
    }
    return VPX_OK;
}

// ---- deferred weight gradients: all the steps of one cell in ONE launch ---------------------------------------------------------
// The one-launch kernel (stw, wgrad2.hip) walks items = (image, 4 x 16-pixel tile); a step's images are just more images. With every
// operand of the T steps in a dense slab [T][B][HW][C] (split format) the batch is the same launch over T * B images: one slab write
// and one slice reduction per cell and training step instead of one per cell STEP (at B = 2..4 per GPU those two were a third of the
// kernel's time, and the per-step results cost autograd one accumulation launch per weight tensor and step).
extern "C" int vpx_stlstm_defers_wgrad(const vpx_stlstm_desc* d) {
    if (!d || d->layout != VPX_LAYOUT_NHWC || d->B < 1 || d->H < 1 || d->W < 1 || d->Cin < 1 || d->Ch < 1) return 0;
    if (!stw_applicable(d)) return 0;
    static thread_local STWArgs sa; static thread_local STWOut so;
    return stw_build(sa, so, d->B, d->H, d->W, d->Cin, d->Ch) >= 1 ? 1 : 0;
}

extern "C" size_t vpx_stlstm_wgrad_batch_workspace_bytes(const vpx_stlstm_desc* d) {
    if (!vpx_stlstm_defers_wgrad(d)) return 0;
    static thread_local STWArgs sa; static thread_local STWOut so;
    if (stw_build(sa, so, d->B, d->H, d->W, d->Cin, d->Ch) < 1) return 0;
    const int ns = stw_slices(sa.npairs5, (long long)d->B * ((d->W + 15) / 16) * ((d->H + 3) / 4));
    return align256(sa.slab_stride * (size_t)ns * 4) + 512;
}

extern "C" int vpx_stlstm_wgrad_batch(const vpx_stlstm_desc* d, const void* dg8_split, const void* const* src5_split, float* dWx, float* dWh,
                                      float* dWm, float* dWo, float* dWlast, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!vpx_stlstm_defers_wgrad(d)) { set_error("vpx_stlstm_wgrad_batch: not available for this descriptor (vpx_stlstm_defers_wgrad)"); return VPX_ERR_UNSUPPORTED; }
    if (!dg8_split || !src5_split || !dWx || !dWh || !dWm || !dWo || !dWlast) { set_error("vpx_stlstm_wgrad_batch: NULL tensor argument"); return VPX_ERR_ARG; }
    for (int i = 0; i < 5; ++i) if (!src5_split[i]) { set_error("vpx_stlstm_wgrad_batch: source %d is NULL", i); return VPX_ERR_ARG; }
    if ((long long)d->B * ((d->W + 15) / 16) * ((d->H + 3) / 4) > 0x7fffffffll) { set_error("vpx_stlstm_wgrad_batch: too many items"); return VPX_ERR_UNSUPPORTED; }
    const size_t need = vpx_stlstm_wgrad_batch_workspace_bytes(d);
    if (!workspace || workspace_bytes < need) { set_error("vpx_stlstm_wgrad_batch: workspace too small"); return VPX_ERR_WORKSPACE; }
    static thread_local STWArgs sa; static thread_local STWOut so;
    if (stw_build(sa, so, d->B, d->H, d->W, d->Cin, d->Ch) < 1) { set_error("vpx_stlstm_wgrad_batch: pair table"); return VPX_ERR_UNSUPPORTED; }
    const int ns = stw_slices(sa.npairs5, (long long)d->B * ((d->W + 15) / 16) * ((d->H + 3) / 4));
    Carver ws(workspace, workspace_bytes);
    float* slabs = ws.take(sa.slab_stride * (size_t)ns);
    VPX_CHECK_CARVE(ws, "vpx_stlstm_wgrad_batch");
    sa.g_sp = reinterpret_cast<const char*>(dg8_split);
    sa.src[0] = STWSrc{reinterpret_cast<const char*>(src5_split[0]), d->Cin};
    for (int i = 1; i < 5; ++i) sa.src[i] = STWSrc{reinterpret_cast<const char*>(src5_split[i]), d->Ch};
    sa.n_slices = ns; sa.slabs = slabs;
    so.dW[0] = dWx; so.dW[1] = dWh; so.dW[2] = dWm; so.dW[3] = dWo; so.dW[4] = dWlast;
    VPX_CHECK_HIP(launch_stw(sa, so, (hipStream_t)stream_));
    return VPX_OK;
}
