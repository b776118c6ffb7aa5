// vpx_host.h — host-side helpers shared by the extern "C" translation units.
#pragma once
#include <stdlib.h>
#include <string.h>

#include "vpx_internal.h"

namespace vpx {

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct Carver {  // bump allocator over the caller's workspace
    char* base;
    size_t off, cap;
    float* take(size_t nfloat) {
        float* p = reinterpret_cast<float*>(base + off);
        off += align256(nfloat * sizeof(float));
        return p;
    }
};

#define VPX_CHECK_HIP(expr)                                                                   \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) {                                                              \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return VPX_ERR_LAUNCH;                                                            \
        }                                                                                     \
    } while (0)

static inline void gate_positions(int gate_order, int gp[4]) {
    gp[0] = 0;
    gp[1] = 1;
    if (gate_order == VPX_GATE_IFGO) { gp[2] = 2; gp[3] = 3; }  // logical g at chunk 2, o at chunk 3
    else { gp[2] = 3; gp[3] = 2; }                             // ndrplz: o at chunk 2, g at chunk 3
}

static inline int check_convlstm_desc(const vpx_convlstm_desc* d) {
    if (!d) { set_error("desc is NULL"); return VPX_ERR_ARG; }
    if (d->B < 1 || d->T < 1 || d->Cin < 1 || d->Ch < 1 || d->H < 1 || d->W < 1) {
        set_error("convlstm desc: non-positive dimension (B=%d T=%d Cin=%d Ch=%d H=%d W=%d)", d->B, d->T, d->Cin,
                  d->Ch, d->H, d->W);
        return VPX_ERR_ARG;
    }
    if (d->kh < 1 || d->kw < 1 || !(d->kh & 1) || !(d->kw & 1) || d->kh > 7 || d->kw > 7) {
        set_error("convlstm desc: kernel size must be odd and <= 7 (got %dx%d)", d->kh, d->kw);
        return VPX_ERR_ARG;
    }
    if (d->gate_order != VPX_GATE_IFGO && d->gate_order != VPX_GATE_IFOG) {
        set_error("convlstm desc: unknown gate_order %d", d->gate_order);
        return VPX_ERR_ARG;
    }
    if (d->layout != VPX_LAYOUT_NHWC && d->layout != VPX_LAYOUT_NCHW) {
        set_error("convlstm desc: unknown layout %d", d->layout);
        return VPX_ERR_ARG;
    }
    if ((d->precision < VPX_PREC_F32 || d->precision > VPX_PREC_BF16)) {
        set_error("convlstm desc: precision %d not implemented (VPX_PREC_F32 and VPX_PREC_BF16X3 are)", d->precision);
        return VPX_ERR_UNSUPPORTED;
    }
    return VPX_OK;
}

struct ConvLSTMLayout {  // derived sizes shared by workspace query, fwd and bwd
    int taps, n_tiles, nstage, chunks_total;
    int mw;                        // forward cell kernel: 32-pixel row tiles per wave
    ConvStage stage[MAX_STAGE];
    size_t n_state, n_x, n_out, n_peep;
    // backward
    int d_nstage, d_chunks;        // data-gradient conv: K stages over the 4Ch gate axis
    ConvStage d_stage[MAX_STAGE];
    int d_tiles_full, d_tiles_h;   // N tiles when producing [dx | dh] resp. only dh
    int n_ctiles;                  // weight-gradient: 64-channel slices of [x | h]
    WgradCTile ct[16];
    int n_slices;                  // weight-gradient K slices
    size_t slab_floats;
};

static inline int convlstm_layout(const vpx_convlstm_desc* d, ConvLSTMLayout& L) {
    L.taps = d->kh * d->kw;
    L.n_tiles = (d->Ch + 31) / 32;
    const int segC[2] = {d->Cin, d->Ch};
    L.mw = pick_mw(d->B, d->H, d->W, L.n_tiles, d->precision);
    L.nstage = build_stages(L.stage, &L.chunks_total, segC, 2, L.taps, pick_stage_channels(segC, 2, d->kh, d->kw, 4, d->precision, L.mw), d->precision);
    if (L.nstage < 0) { set_error("convlstm: too many channel stages (Cin=%d Ch=%d)", d->Cin, d->Ch); return VPX_ERR_UNSUPPORTED; }
    L.n_state = (size_t)d->B * d->H * d->W * d->Ch;
    L.n_x = (size_t)d->B * d->T * d->H * d->W * d->Cin;
    L.n_out = (size_t)d->B * d->T * d->H * d->W * d->Ch;
    L.n_peep = (size_t)d->H * d->W * d->Ch;
    // ---- backward sizing ----
    const int N4 = 4 * d->Ch, Ct = d->Cin + d->Ch;
    const int segD[1] = {N4};
    L.d_nstage = build_stages(L.d_stage, &L.d_chunks, segD, 1, L.taps, pick_stage_channels(segD, 1, d->kh, d->kw, 4, d->precision), d->precision);
    if (L.d_nstage < 0) { set_error("convlstm: too many channel stages in the data-gradient conv (Ch=%d)", d->Ch); return VPX_ERR_UNSUPPORTED; }
    L.d_tiles_full = plain_tiles(Ct);
    L.d_tiles_h = plain_tiles(d->Ch);
    L.n_ctiles = 0;
    for (int c0 = 0; c0 < d->Cin; c0 += 64) {
        if (L.n_ctiles >= 16) { set_error("convlstm: too many channels for the weight-gradient kernel"); return VPX_ERR_UNSUPPORTED; }
        L.ct[L.n_ctiles++] = WgradCTile{0, c0, (d->Cin - c0 < 64) ? d->Cin - c0 : 64, c0};
    }
    for (int c0 = 0; c0 < d->Ch; c0 += 64) {
        if (L.n_ctiles >= 16) { set_error("convlstm: too many channels for the weight-gradient kernel"); return VPX_ERR_UNSUPPORTED; }
        L.ct[L.n_ctiles++] = WgradCTile{1, c0, (d->Ch - c0 < 64) ? d->Ch - c0 : 64, d->Cin + c0};
    }
    {
        const int tiles = ((d->W + TILE_W - 1) / TILE_W) * ((d->H + TILE_H - 1) / TILE_H);
        const long long items = (long long)d->T * d->B * tiles;
        const int out_tiles = ((N4 + 63) / 64) * L.n_ctiles * ((L.taps + 8) / 9);
        long long ns = (1024 + out_tiles - 1) / out_tiles;
        if (ns > items) ns = items;
        if (ns > 256) ns = 256;
        if (ns < 1) ns = 1;
        L.n_slices = (int)ns;
        L.slab_floats = (size_t)L.n_slices * L.taps * N4 * Ct;
    }
    return VPX_OK;
}


static inline size_t convlstm_bwd_workspace_bytes(const vpx_convlstm_desc* d, const ConvLSTMLayout& L) {
    size_t b = align256(packed_weight_bytes(L.d_tiles_full, L.d_chunks, 4, d->precision));  // upper bound over both tilings
    b += align256((size_t)d->T * L.n_state * 4 * sizeof(float));  // dG, all steps
    b += 2 * align256(L.n_state * sizeof(float));                 // dh, dc carries
    b += align256(L.slab_floats * sizeof(float));
    b += align256((size_t)d->T * gate_bwd_blocks(d->H * d->W, d->Ch) * 4 * d->Ch * sizeof(float));  // bias-gradient partials
    if (d->layout == VPX_LAYOUT_NCHW) {
        // staged copies of x, out, dout, dx + states (h0,c0,dhT,dcT,dh0,dc0) + 6 peephole-sized buffers
        b += 2 * align256(L.n_x * 4) + 2 * align256(L.n_out * 4) + 6 * align256(L.n_state * 4) + 6 * align256(L.n_peep * 4);
    }
    return b;
}

// backward workspace of one ST-LSTM step (stlstm_bwd_api.hip); 0 until that file provides a real figure
size_t stlstm_bwd_workspace_bytes(const vpx_stlstm_desc* d);

}  // namespace vpx
