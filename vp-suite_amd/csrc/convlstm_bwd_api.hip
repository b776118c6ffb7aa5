// convlstm_bwd_api.hip — vpx_convlstm_seq_bwd: BPTT through the fused ConvLSTM sequence (include/vpx.h).
// Per step (reverse time): gate-backward (pointwise) -> data-gradient conv (same implicit-GEMM kernel as the forward,
// weights packed transposed + tap-flipped); after the loop one weight-gradient launch over all (t, b) images, a
// K-slice reduction into the reference's OIHW layout, and a column sum for the bias gradient.
#include "vpx_host.h"

using namespace vpx;

extern "C" int vpx_convlstm_seq_bwd(const vpx_convlstm_desc* d, const float* x, const float* h0, const float* c0,
                                    const float* W, const float* Wci, const float* Wcf, const float* Wco,
                                    const float* out, const void* reserve, size_t reserve_bytes, const float* dout,
                                    const float* dhT, const float* dcT, float* dx, float* dh0, float* dc0, float* dW,
                                    float* db, float* dWci, float* dWcf, float* dWco, void* workspace,
                                    size_t workspace_bytes, void* stream_) {
    int rc = check_convlstm_desc(d);
    if (rc != VPX_OK) return rc;
    ConvLSTMLayout L;
    if ((rc = convlstm_layout(d, L)) != VPX_OK) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    if (!(d->flags & VPX_FLAG_SAVE_FOR_BWD)) { set_error("vpx_convlstm_seq_bwd: desc lacks VPX_FLAG_SAVE_FOR_BWD"); return VPX_ERR_ARG; }
    if (!W || !out || !reserve) { set_error("vpx_convlstm_seq_bwd: W, out and reserve must not be NULL"); return VPX_ERR_ARG; }
    if (reserve_bytes < vpx_convlstm_reserve_bytes(d)) { set_error("vpx_convlstm_seq_bwd: reserve too small"); return VPX_ERR_WORKSPACE; }
    if (!workspace || workspace_bytes < vpx_convlstm_workspace_bytes(d)) {
        set_error("vpx_convlstm_seq_bwd: workspace too small (%zu < %zu)", workspace_bytes, vpx_convlstm_workspace_bytes(d));
        return VPX_ERR_WORKSPACE;
    }
    const bool peep = Wci || Wcf || Wco;
    if (peep && !(Wci && Wcf && Wco)) { set_error("vpx_convlstm_seq_bwd: peephole tensors must be given together"); return VPX_ERR_ARG; }
    const bool dpeep = dWci || dWcf || dWco;
    if (dpeep && !(dWci && dWcf && dWco && peep)) { set_error("vpx_convlstm_seq_bwd: peephole gradients must be requested together"); return VPX_ERR_ARG; }

    const int B = d->B, T = d->T, Cin = d->Cin, Ch = d->Ch, H = d->H, Wd = d->W;
    const int N4 = 4 * Ch, Ct = Cin + Ch;
    const size_t HW = (size_t)H * Wd;
    Carver ws(workspace, workspace_bytes);
    float* wpk = ws.take(packed_weight_bytes(L.d_tiles_full, L.d_chunks, 4, d->precision, L.d_qpc) / sizeof(float));
    float* dG_all = ws.take((size_t)T * L.n_state * 4);
    float* dh_buf = ws.take(L.n_state);
    float* dc_buf = ws.take(L.n_state);
    float* slabs = ws.take(L.slab_floats);
    // peephole gradients: a sum over the batch. With several batch slices every slice accumulates (over the time steps, single
    // owner per element) into its own partial tensor; launch_peep_reduce adds them in a fixed order after the loop
    const int gb_slices = gate_bwd_slices((int)HW, Ch, B, false);
    const int gb_blocks = gate_bwd_blocks((int)HW, Ch) * gb_slices;  // partial rows per step
    float* db_part = ws.take((size_t)T * GATE_BWD_MAX_SLICES * gate_bwd_blocks((int)HW, Ch) * N4);
    float* db_part2 = ws.take((size_t)COLSUM_BLOCKS * N4);
    float* peep_part[3];
    for (auto& pp_ : peep_part) pp_ = ws.take((size_t)GATE_BWD_MAX_SLICES * L.n_peep);
    // forward on the second-generation cell: the gate-backward kernel also writes dG in split operand format and the data
    // gradient runs on the cell2 main loop with a plain epilogue (conv2); VPX_CONV2_DGRAD=0 keeps the first-generation kernel
    static int c2d_env = -1;
    if (c2d_env < 0) c2d_env = dev_switch("VPX_CONV2_DGRAD", 1);
    const bool c2d = L.v2 && c2d_env != 0;
    char* dG_sp_all = nullptr; char* wpk2 = nullptr;
    if (L.v2) {
        dG_sp_all = (char*)ws.take((size_t)T * L.n_state * 4);
        wpk2 = (char*)ws.take((cell2_packed_bytes(conv2_tiles(Ct), 3 * (N4 / 16)) + 16384 * conv2_tiles(Ct)) / sizeof(float));
    }
    const bool wsp = wgrad2_wsp(d, L) && dW != nullptr;   // forward on another kernel, weight gradient on wgrad2 all the same
    char *xsp_w = nullptr, *hsp_w = nullptr, *h0sp_w = nullptr;
    if (wgrad2_wsp(d, L)) {
        dG_sp_all = (char*)ws.take((size_t)T * L.n_state * 4);
        xsp_w = (char*)ws.take(L.n_x);
        hsp_w = (char*)ws.take(L.n_out);
        h0sp_w = (char*)ws.take(L.n_state);
    }

    VPX_CHECK_CARVE(ws, "vpx_convlstm_seq_bwd");
    // ---- layout adaptation ----
    const float *xn = x, *h0n = h0, *c0n = c0, *outn = out, *doutn = dout, *dhTn = dhT, *dcTn = dcT;
    const float *wci = Wci, *wcf = Wcf, *wco = Wco;
    float *dxn = dx, *dh0n = dh0, *dc0n = dc0, *dwci = dWci, *dwcf = dWcf, *dwco = dWco;
    if (d->layout == VPX_LAYOUT_NCHW) {
        float* bx = ws.take(L.n_x);
        float* bdx = ws.take(L.n_x);
        float* bout = ws.take(L.n_out);
        float* bdout = ws.take(L.n_out);
        float* st[6];
        for (auto& p : st) p = ws.take(L.n_state);
        float* pp[6];
        for (auto& p : pp) p = ws.take(L.n_peep);
        VPX_CHECK_CARVE(ws, "vpx_convlstm_seq_bwd");
        if (x) { VPX_CHECK_HIP(launch_nchw_to_nhwc(x, bx, B * T, Cin, H, Wd, stream)); xn = bx; }
        VPX_CHECK_HIP(launch_nchw_to_nhwc(out, bout, B * T, Ch, H, Wd, stream)); outn = bout;
        if (dout) { VPX_CHECK_HIP(launch_nchw_to_nhwc(dout, bdout, B * T, Ch, H, Wd, stream)); doutn = bdout; }
        if (h0) { VPX_CHECK_HIP(launch_nchw_to_nhwc(h0, st[0], B, Ch, H, Wd, stream)); h0n = st[0]; }
        if (c0) { VPX_CHECK_HIP(launch_nchw_to_nhwc(c0, st[1], B, Ch, H, Wd, stream)); c0n = st[1]; }
        if (dhT) { VPX_CHECK_HIP(launch_nchw_to_nhwc(dhT, st[2], B, Ch, H, Wd, stream)); dhTn = st[2]; }
        if (dcT) { VPX_CHECK_HIP(launch_nchw_to_nhwc(dcT, st[3], B, Ch, H, Wd, stream)); dcTn = st[3]; }
        if (dh0) dh0n = st[4];
        if (dc0) dc0n = st[5];
        if (dx) dxn = bdx;
        if (peep) {
            VPX_CHECK_HIP(launch_nchw_to_nhwc(Wci, pp[0], 1, Ch, H, Wd, stream));
            VPX_CHECK_HIP(launch_nchw_to_nhwc(Wcf, pp[1], 1, Ch, H, Wd, stream));
            VPX_CHECK_HIP(launch_nchw_to_nhwc(Wco, pp[2], 1, Ch, H, Wd, stream));
            wci = pp[0]; wcf = pp[1]; wco = pp[2];
        }
        if (dpeep) { dwci = pp[3]; dwcf = pp[4]; dwco = pp[5]; }
    }

    const float* gates_all = (const float*)reserve;
    const float* cs_all = (const float*)((const char*)reserve + align256((size_t)T * L.n_state * 4 * sizeof(float)));
    int gp[4];
    gate_positions(d->gate_order, gp);

    const bool peep_sliced = dpeep && gb_slices > 1;
    if (peep_sliced) {
        for (auto pp_ : peep_part) VPX_CHECK_HIP(vpx_memset_async(pp_, 0, (size_t)gb_slices * L.n_peep * sizeof(float), stream));
    } else if (dpeep) {
        VPX_CHECK_HIP(vpx_memset_async(dwci, 0, L.n_peep * sizeof(float), stream));
        VPX_CHECK_HIP(vpx_memset_async(dwcf, 0, L.n_peep * sizeof(float), stream));
        VPX_CHECK_HIP(vpx_memset_async(dwco, 0, L.n_peep * sizeof(float), stream));
    }

    // ---- data-gradient weights: contraction over the 4Ch gate rows, outputs over [x | h] (or only h) columns ----
    const bool need_dx = dxn && xn;
    const int col_start = need_dx ? 0 : Cin;
    const int n_out = need_dx ? Ct : Ch;
    const int d_tiles = plain_tiles(n_out);
    const int qform = mfma_shape() == 1 ? 1 : 0;   // conv2's epilogue handles ragged tiles in both forms
    if (c2d) {
        Conv2Pack pk{W, (long long)L.taps, (long long)Ct * L.taps, n_out, col_start, conv2_tiles(n_out), conv2_gpt(n_out),
                     qform ? cell2_qchunks(N4 / 16) : 3 * (N4 / 16), 1, qform, 0};
        VPX_CHECK_HIP(launch_conv2_pack(pk, wpk2, stream));
    } else {
        PackDesc pd{};
        pd.seg[0] = PackSeg{W, (long long)Ct * L.taps, L.taps, 0, N4};
        memcpy(pd.stage, L.d_stage, sizeof(ConvStage) * L.d_nstage);
        pd.nstage = L.d_nstage; pd.chunks_total = L.d_chunks; pd.prec = d->precision; pd.taps = L.taps; pd.qpc = L.d_qpc;
        fill_plain_pack(pd, n_out, col_start);
        pd.transposed = 1; pd.flip = 1;
        VPX_CHECK_HIP(launch_pack_weights(pd, wpk, stream));
    }

    // fp32 dG is only written for consumers that still read it (first-generation data- / weight-gradient kernels)
    bool wgrad_takes_sp = false;
    if (c2d && d->layout == VPX_LAYOUT_NHWC) {
        WgradArgs probe{};
        probe.T = T; probe.B = B; probe.H = H; probe.W = Wd; probe.kh = d->kh; probe.kw = d->kw; probe.N4 = N4; probe.Cin = Cin; probe.Ch = Ch;
        probe.prec = d->precision; probe.a_split = 1; probe.g_sp = dG_sp_all;
        wgrad_takes_sp = wgrad2_applicable(probe);
    }
    const bool need_dG_f32 = !c2d || (dW && !wgrad_takes_sp);
    for (int t = T - 1; t >= 0; --t) {
        GateBwdArgs ga{};
        ga.B = B; ga.HW = (int)HW; ga.Ch = Ch;
        memcpy(ga.gate_pos, gp, sizeof(gp));
        ga.gates = gates_all + (size_t)t * L.n_state * 4;
        ga.c_t = cs_all + (size_t)t * L.n_state;
        ga.c_prev = (t > 0) ? cs_all + (size_t)(t - 1) * L.n_state : c0n;
        ga.dh_in = (t == T - 1) ? dhTn : dh_buf;
        ga.dout = doutn ? doutn + (size_t)t * HW * Ch : nullptr;
        ga.dout_bstride = (long long)((size_t)T * HW * Ch);
        ga.dc_in = (t == T - 1) ? dcTn : dc_buf;
        ga.dc_out = (t == 0 && dc0n) ? dc0n : dc_buf;
        ga.wci = wci; ga.wcf = wcf; ga.wco = wco;
        ga.dwci = dpeep ? dwci : nullptr; ga.dwcf = dpeep ? dwcf : nullptr; ga.dwco = dpeep ? dwco : nullptr;
        if (peep_sliced) { ga.dwci = peep_part[0]; ga.dwcf = peep_part[1]; ga.dwco = peep_part[2]; ga.peep_slice_stride = (long long)L.n_peep; }
        ga.dG = need_dG_f32 ? dG_all + (size_t)t * L.n_state * 4 : nullptr;
        ga.dG_sp = (c2d || wsp) ? dG_sp_all + (size_t)t * L.n_state * 16 : nullptr;
        ga.db_partial = db ? db_part + (size_t)t * gb_blocks * N4 : nullptr;
        VPX_CHECK_HIP(launch_gate_bwd(ga, stream));

        float* dh_target = (t > 0) ? dh_buf : dh0n;  // at t = 0 the recurrent gradient is dL/dh0 (if requested)
        if ((need_dx || dh_target) && c2d) {
            Conv2Args ca{};
            ca.B = B; ca.H = H; ca.W = Wd; ca.C = N4; ca.Co = n_out; ca.split = need_dx ? Cin : 0;
            ca.src_sp = ga.dG_sp; ca.src_bstride = (long long)(HW * N4 * 4);
            ca.wpk = wpk2; ca.qform = qform;
            ca.out0 = need_dx ? dxn + (size_t)t * HW * Cin : nullptr;
            ca.bstride0 = (long long)((size_t)T * HW * Cin); ca.ld0 = Cin;
            ca.out1 = dh_target; ca.bstride1 = (long long)(HW * Ch); ca.ld1 = Ch;
            VPX_CHECK_HIP(launch_conv2(ca, stream));
        } else if (need_dx || dh_target) {
            ConvPlan P{};
            P.B = B; P.H = H; P.W = Wd; P.kh = d->kh; P.kw = d->kw;
            set_plan_tiles(P, L.d_mw);
            P.nseg = 1;
            P.seg[0] = ConvSeg{ga.dG, (long long)(HW * N4), N4, 0};
            P.nstage = L.d_nstage;
            memcpy(P.stage, L.d_stage, sizeof(ConvStage) * L.d_nstage);
            P.chunks_total = L.d_chunks; P.prec = d->precision; P.qpc = L.d_qpc;
            P.a_bytes = conv_a_bytes(L.d_stage, L.d_nstage, d->kh, d->kw, L.d_mw);
            P.wpk = wpk;
            PlainEpiArgs ea{};
            ea.Co = n_out; ea.ng = plain_groups(n_out);
            ea.split = need_dx ? Cin : 0;
            ea.out0 = need_dx ? dxn + (size_t)t * HW * Cin : nullptr;
            ea.bstride0 = (long long)((size_t)T * HW * Cin); ea.ld0 = Cin;
            ea.out1 = dh_target; ea.bstride1 = (long long)(HW * Ch); ea.ld1 = Ch;
            VPX_CHECK_HIP(launch_conv_plain_f32(P, ea, d_tiles, stream));
        }
    }
    if (dxn && !xn) { set_error("vpx_convlstm_seq_bwd: dx requested but x is NULL"); return VPX_ERR_ARG; }
    if (peep_sliced)
        VPX_CHECK_HIP(launch_peep_reduce(peep_part[0], peep_part[1], peep_part[2], dwci, dwcf, dwco, gb_slices, (long long)L.n_peep, stream));

    // ---- weight gradient over all (t, b) images ----
    if (dW) {
        WgradArgs wa{};
        wa.T = T; wa.B = B; wa.H = H; wa.W = Wd; wa.HW = (int)HW; wa.kh = d->kh; wa.kw = d->kw;
        wa.tiles_x = (Wd + TILE_W - 1) / TILE_W; wa.tiles_y = (H + TILE_H - 1) / TILE_H;
        wa.N4 = N4; wa.Cin = Cin; wa.Ch = Ch; wa.Ct = Ct; wa.prec = d->precision;
        wa.dG = dG_all;
        wa.x = xn; wa.x_bstride = (long long)((size_t)T * HW * Cin); wa.x_tstride = (long long)(HW * Cin);
        wa.hseq = outn; wa.h_bstride = (long long)((size_t)T * HW * Ch); wa.h_tstride = (long long)(HW * Ch);
        wa.h0 = h0n;
        wa.n_ctiles = wgrad_make_ctiles(wa.ct, WG_MAX_CTILES, xn ? Cin : 0, Ch, Cin);  // no input tensor: its columns stay zero
        if (L.v2 && d->layout == VPX_LAYOUT_NHWC) {
            // the forward ran on the second-generation cell: x, h_0 and h_t sit in the reserve in split operand format
            const char* r = (const char*)reserve + align256((size_t)T * L.n_state * 4 * sizeof(float)) + align256((size_t)T * L.n_state * sizeof(float));
            wa.a_split = 1;
            wa.x_sp = xn ? r : nullptr; wa.x_sp_bstride = (long long)((size_t)T * HW * Cin * 4); wa.x_sp_tstride = (long long)(HW * Cin * 4);
            r += align256(L.n_x * 4);
            wa.h0_sp = h0n ? r : nullptr;
            r += align256(L.n_state * 4);
            wa.h_sp = r; wa.h_sp_bstride = (long long)(HW * Ch * 4); wa.h_sp_tstride = (long long)(L.n_state * 4);
        }
        if (wsp) {   // operands in split form, laid out like x / out ([B][T][HW][C]); conversions are one pass each over small tensors
            if (xn) VPX_CHECK_HIP(launch_split_convert(xn, xsp_w, (long long)B * T * (long long)HW, Cin, stream));
            VPX_CHECK_HIP(launch_split_convert(outn, hsp_w, (long long)B * T * (long long)HW, Ch, stream));
            if (h0n) VPX_CHECK_HIP(launch_split_convert(h0n, h0sp_w, (long long)B * (long long)HW, Ch, stream));
            wa.a_split = 1;
            wa.x_sp = xn ? xsp_w : nullptr; wa.x_sp_bstride = (long long)((size_t)T * HW * Cin * 4); wa.x_sp_tstride = (long long)(HW * Cin * 4);
            wa.h0_sp = h0n ? h0sp_w : nullptr;
            wa.h_sp = hsp_w; wa.h_sp_bstride = (long long)((size_t)T * HW * Ch * 4); wa.h_sp_tstride = (long long)(HW * Ch * 4);
        }
        wa.slabs = slabs;
        // every launched tile stores all of its slab elements; only the skipped x columns need a clear
        if (!xn) VPX_CHECK_HIP(vpx_memset_async(slabs, 0, L.slab_floats * sizeof(float), stream));
        wa.g_sp = (c2d || wsp) ? dG_sp_all : nullptr;
        int ns_used = L.n_slices, tail_col0 = Ct, tail_slices = L.n_slices;
        if (L.n_slices2 > 0 && wgrad2_applicable(wa)) VPX_CHECK_HIP(launch_wgrad2(wa, L.n_slices2, &ns_used, &tail_col0, &tail_slices, stream));
        else VPX_CHECK_HIP(launch_wgrad(wa, L.n_slices, stream));
        VPX_CHECK_HIP(launch_wgrad_reduce_tail(slabs, dW, ns_used, L.taps, N4, Ct, tail_col0, tail_slices, stream));
    }
    if (db)  // block partials from the gate-backward kernel, summed in a fixed order
        VPX_CHECK_HIP(launch_colsum(db_part, nullptr, 0.f, nullptr, db, db_part2, (long long)T * gb_blocks, N4, stream));
    if (d->layout == VPX_LAYOUT_NCHW) {
        if (dx) VPX_CHECK_HIP(launch_nhwc_to_nchw(dxn, dx, B * T, Cin, H, Wd, stream));
        if (dh0) VPX_CHECK_HIP(launch_nhwc_to_nchw(dh0n, dh0, B, Ch, H, Wd, stream));
        if (dc0) VPX_CHECK_HIP(launch_nhwc_to_nchw(dc0n, dc0, B, Ch, H, Wd, stream));
        if (dpeep) {
            VPX_CHECK_HIP(launch_nhwc_to_nchw(dwci, dWci, 1, Ch, H, Wd, stream));
            VPX_CHECK_HIP(launch_nhwc_to_nchw(dwcf, dWcf, 1, Ch, H, Wd, stream));
            VPX_CHECK_HIP(launch_nhwc_to_nchw(dwco, dWco, 1, Ch, H, Wd, stream));
        }
    }
    return VPX_OK;
}
