// vpx_host.h — host-side helpers shared by the extern "C" translation units.
#pragma once
#include <stdlib.h>
#include <string.h>

#include "vpx_internal.h"

namespace vpx {


// (Carver — the bump allocator over the caller's workspace — lives in vpx_internal.h: the launchers check their writes against it)

#define VPX_CHECK_HIP(expr)                                                                   \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) {                                                              \
            if (ws_violation()[0]) {   /* a launcher refused to write past its workspace slot */ \
                set_error("%s (%s:%d)", ws_violation(), __FILE__, __LINE__);                   \
                ws_violation_clear();   /* consumed: a later, unrelated HIP error must not report it again */ \
                return VPX_ERR_WORKSPACE;                                                     \
            }                                                                                 \
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

// Three k-steps per weight chunk when that removes the half-empty chunk at the end of every stage (the tap count is a
// multiple of 3, e.g. 3x3: 9 k-steps per 16-channel stage = 3 chunks instead of 5) and the larger weight buffers do not
// lower the number of resident workgroups. bf16 modes only. VPX_QPC=2/3 forces it.
static inline int pick_qpc(const int* segC, int nseg, int kh, int kw, int ng, int prec, int mw) {
    static int forced = -1;
    if (forced < 0) forced = dev_switch("VPX_QPC", 0);
    if (prec == VPX_PREC_F32 || (kh * kw) % 3 != 0 || mw > 2) return 2;
    if (forced == 2 || forced == 3) return forced;
    auto residency = [&](int qpc) {
        const int cs = pick_stage_channels(segC, nseg, kh, kw, ng, prec, mw, 1, qpc);
        const int npos = (TILE_H * mw + kh - 1) * (TILE_W + kw - 1);
        const int lds = npos * (cs * 4 + 16) + 2 * ng * 32 * ((prec == VPX_PREC_F32 ? 8 : 16) * qpc * 4 + 16);
        int wg = (160 * 1024) / lds;
        const int cap = mw == 2 ? 2 : 3;
        return wg > cap ? cap : wg;
    };
    return residency(3) >= residency(2) ? 3 : 2;
}

// workgroups a weight-gradient launch aims at (K slices x output tiles); VPX_WGRAD_WGS overrides (experiments)
static inline int wgrad_target_wgs() {
    static int v = -1;
    if (v < 0) v = dev_switch("VPX_WGRAD_WGS", 1024);
    return v;
}

// Second-generation fused cell (cell2.hip) applies to: bf16x3, 3x3, channel counts in whole 16-channel stages, maps
// taller than half a 32x16 tile, and a launch of at least 128 workgroups (half the CUs: measured crossover, B=16 +11 %, B=32
// +3 % against a bar of 256; below, the first-generation kernel's 128-pixel tiles fill more CUs). VPX_CELL2=0 disables it, =2 forces it
// wherever the shape allows (experiments).
extern int g_cell2_mode;  // vpx_api.hip: -1 = not yet read from the environment
static inline int cell2_mode() {
    if (g_cell2_mode < 0) g_cell2_mode = dev_switch("VPX_CELL2", 1);
    return g_cell2_mode;
}
static inline bool cell2_applicable(const vpx_convlstm_desc* d) {
    if (cell2_mode() == 0) return false;
    if (d->kh != 3 || d->kw != 3) return false;
    if (d->precision == VPX_PREC_BF16) {
        // plain bf16 (BASELINE configs[1]'s literal dtype): inference only, on the half tile of the 16x16x32 form (cell2_kernel_q<.., 4, true>);
        // a call that saves for the backward pass stays on the first-generation kernel, whose BPTT kernels know this mode
        if ((d->flags & VPX_FLAG_SAVE_FOR_BWD) || mfma_shape() != 1 || (d->H & 15) || (d->W & 15) || (d->Ch & 31)) return false;
    } else if (d->precision != VPX_PREC_BF16X3) return false;
    if ((d->Cin & 15) || (d->Ch & 15) || (d->Cin + d->Ch) / 16 > MAX_STAGE) return false;
    // 16-row maps: on the half tile (q form) only. Measured at B=128, (96,96,16x16): 128 -> 106 us per step against the
    // first-generation kernel (384 workgroups of four waves); VPX_CELL2_H16=0 keeps them there
    static int h16 = -1;
    if (h16 < 0) h16 = dev_switch("VPX_CELL2_H16", 1);
    if (d->H < 16 || (d->H == 16 && !(h16 && mfma_shape() == 1 && (d->W & 15) == 0 && (d->Ch & 31) == 0))) return false;
    if (cell2_mode() == 2) return true;
    const long long wgs = (long long)d->B * ((d->H + 31) / 32) * ((d->W + 15) / 16) * ((d->Ch + 31) / 32);
    static int min_wgs = -1;   // VPX_CELL2_MIN_WGS: experiment override of the bar below
    // 64 (= 128 half tiles of four waves, cell2_kernel_q<.., 4>): the B=4 steps of the 64x64 blocks, whose first-generation launch
    // has exactly 256 workgroups, run 48 -> 38 us there (B=4 step 2.13 -> 1.93 ms; B=4 at 3x128x128 6.71 -> 5.89 ms)
    if (min_wgs < 0) min_wgs = dev_switch("VPX_CELL2_MIN_WGS", 64);
    return wgs >= min_wgs;
}

// 16x16x32 form of the second-generation kernels (vpx_set_option(VPX_OPT_MFMA_SHAPE)): the fused cell's q-form epilogue is the
// vectorised one only — tiles inside the image, whole 32-channel tiles
#define VPX_MFMA_SHAPE_DEFAULT 1   // measured (tools/ab_shape.py, B=128, one process, interleaved): 1.05-1.11x per fused step, every block shape
static inline bool cell2_q_applicable(const vpx_convlstm_desc* d) {
    return mfma_shape() == 1 && (d->H & 15) == 0 && (d->W & 15) == 0 && (d->Ch & 31) == 0;   // (H % 32 == 16: the half tile only)
}

struct ConvLSTMLayout {  // derived sizes shared by workspace query, fwd and bwd
    int taps, n_tiles, nstage, chunks_total;
    int v2;                        // 1: the forward steps run on cell2_kernel (pre-split operands)
    int v3;                        // 1: small grid, the forward steps run on cell3_kernel (8-channel slices, hoisted input projection)
    int mw;                        // forward cell kernel: 32-pixel row tiles per wave
    int qpc, d_qpc;                // k-steps per weight chunk of the forward cell / data-gradient launches (2 or 3)
    ConvStage stage[MAX_STAGE];
    // small maps: the step as a K-split plain convolution into a gate buffer + a pointwise gate kernel (0 = fused launch)
    int split, s_ng, s_tiles, s_nstage, s_chunks;
    ConvStage s_stage[MAX_STAGE];
    // ... with the input projection hoisted out of the recurrence (split path, T > 1): W_x * x_t for ALL steps is ONE launch
    // over B*T images into the gate buffers; a step then only contracts h_{t-1} (K = Ch*taps instead of (Cin+Ch)*taps on
    // the serial path) and accumulates into its slice — which also saves the per-step clear of the gate buffer
    int hoist, hx_nstage, hx_chunks, hh_nstage, hh_chunks, hh_split;
    ConvStage hx_stage[MAX_STAGE], hh_stage[MAX_STAGE];
    size_t n_state, n_x, n_out, n_peep;
    // backward
    int d_nstage, d_chunks, d_mw;  // data-gradient conv: K stages over the 4Ch gate axis; rows-per-workgroup variant
    ConvStage d_stage[MAX_STAGE];
    int d_tiles_full, d_tiles_h;   // N tiles when producing [dx | dh] resp. only dh
    int n_ctiles;                  // weight-gradient: 64-channel slices of [x | h]
    WgradCTile ct[WG_MAX_CTILES];
    int n_slices;                  // weight-gradient K slices (first-generation kernels)
    int n_slices2;                 // ... of wgrad2.hip where it can take the launch (0 = never)
    size_t slab_floats;
};

// Blocks whose FORWARD did not run on the operand-format kernel (16x16 maps, small grids) still take the wgrad2 weight gradient:
// the backward converts x, the output sequence and h0 to the split format itself (small tensors there) and the gate-backward
// kernel writes dG in both forms. VPX_WGRAD2_WSP=0 disables.
static inline bool wgrad2_wsp(const vpx_convlstm_desc* d, const ConvLSTMLayout& L) {
    static int env = -1;
    if (env < 0) env = dev_switch("VPX_WGRAD2_WSP", 1);
    return env && !L.v2 && d->precision == VPX_PREC_BF16X3 && d->kh == 3 && d->kw == 3 && (d->Cin & 7) == 0 && (d->Ch & 7) == 0;
}

static inline int convlstm_layout(const vpx_convlstm_desc* d, ConvLSTMLayout& L) {
    L.taps = d->kh * d->kw;
    L.n_tiles = (d->Ch + 31) / 32;
    const int segC[2] = {d->Cin, d->Ch};
    L.mw = pick_mw(d->B, d->H, d->W, L.n_tiles, d->precision);
    L.qpc = pick_qpc(segC, 2, d->kh, d->kw, 4, d->precision, L.mw);
    if (L.mw > 1 && !conv_fits_lds(segC, 2, d->kh, d->kw, 4, d->precision, L.mw, 1, L.qpc)) {   // (8-wave form too large for LDS: 4-wave form)
        L.mw = 1;
        L.qpc = pick_qpc(segC, 2, d->kh, d->kw, 4, d->precision, 1);
    }
    if (!conv_fits_lds(segC, 2, d->kh, d->kw, 4, d->precision, L.mw, 1, L.qpc)) {
        set_error("convlstm: %dx%d kernel over %d+%d channels does not fit the cell kernel's LDS stages", d->kh, d->kw, d->Cin, d->Ch);
        return VPX_ERR_UNSUPPORTED;
    }
    L.nstage = build_stages(L.stage, &L.chunks_total, segC, 2, L.taps, pick_stage_channels(segC, 2, d->kh, d->kw, 4, d->precision, L.mw, 1, L.qpc), d->precision, L.qpc);
    if (L.nstage < 0) { set_error("convlstm: too many channel stages (Cin=%d Ch=%d)", d->Cin, d->Ch); return VPX_ERR_UNSUPPORTED; }
    L.split = 0;
    L.hoist = 0;
    L.v3 = 0;
    {
        const long long m_tiles = (long long)d->B * ((d->H + TILE_H - 1) / TILE_H) * ((d->W + TILE_W - 1) / TILE_W);
        // the fused launch would leave CUs idle (small batches / maps). Measured on MI355X, bf16x3, (96,96,16x16):
        // B=4 (24 workgroups fused) 20 -> 39 TF with the split; B=32 (192 workgroups) 155 -> 164 TF with 2 splits
        static int bar = -1;  // VPX_SPLIT_BAR: experiment override of the workgroup-count bar below
        if (bar < 0) bar = dev_switch("VPX_SPLIT_BAR", 256);
        static int bar3 = -1;  // VPX_CELL3_BAR: workgroup count of the fused launch below which cell3.hip takes the step (where it applies)
        if (bar3 < 0) bar3 = dev_switch("VPX_CELL3_BAR", 256);
        const bool force2 = cell2_mode() == 2 && cell2_applicable(d);   // "wherever the shape allows": also on small grids (tests, A/B)
        const bool want3 = !force2 && m_tiles * L.n_tiles < bar3 && cell3_applicable(d);
        if (!force2 && (m_tiles * L.n_tiles < bar || want3)) {
            const int ng = plain_groups(4 * d->Ch);
            const int tiles = plain_tiles_ng(4 * d->Ch, ng);
            // grids this small never have more than ~1 workgroup per CU, so LDS residency is no argument for small stages:
            // 32-channel stages halve the stage switches (measured (96,96,16x16) B=32: 164 -> 174 TF)
            int cs = pick_stage_channels(segC, 2, d->kh, d->kw, ng, d->precision);
            if (cs < 32) cs = 32;
            L.s_nstage = build_stages(L.s_stage, &L.s_chunks, segC, 2, L.taps, cs, d->precision);
            const int ks = (L.s_nstage > 0 && m_tiles * L.n_tiles < bar) ? pick_ksplit(m_tiles * tiles, L.s_nstage) : 1;
            if (ks > 1) { L.split = ks; L.s_ng = ng; L.s_tiles = tiles; }
            L.hoist = 0;
            static int hoist_on = -1;  // VPX_HOIST=0 disables (experiments)
            if (hoist_on < 0) hoist_on = dev_switch("VPX_HOIST", 1);
            if (L.split && d->T > 1 && hoist_on) {
                const int sx[1] = {d->Cin}, sh[1] = {d->Ch};
                L.hx_nstage = build_stages(L.hx_stage, &L.hx_chunks, sx, 1, L.taps, pick_stage_channels(sx, 1, d->kh, d->kw, ng, d->precision), d->precision);
                L.hh_nstage = build_stages(L.hh_stage, &L.hh_chunks, sh, 1, L.taps, cs, d->precision);
                if (L.hx_nstage > 0 && L.hh_nstage > 0) {
                    L.hh_split = pick_ksplit(m_tiles * tiles, L.hh_nstage);
                    L.hoist = 1;
                }
            }
            // third form (cell3.hip): no K split, no atomics — preferred wherever its shape limits allow
            if (want3) {
                const int sx[1] = {d->Cin};
                L.hx_nstage = build_stages(L.hx_stage, &L.hx_chunks, sx, 1, L.taps, pick_stage_channels(sx, 1, d->kh, d->kw, ng, d->precision), d->precision);
                if (L.hx_nstage > 0) { L.v3 = 1; L.split = 0; L.hoist = 0; L.s_ng = ng; L.s_tiles = tiles; }
            }
        }
    }
    L.v2 = (!L.split && !L.v3 && cell2_applicable(d)) ? 1 : 0;
    L.n_state = (size_t)d->B * d->H * d->W * d->Ch;
    L.n_x = (size_t)d->B * d->T * d->H * d->W * d->Cin;
    L.n_out = (size_t)d->B * d->T * d->H * d->W * d->Ch;
    L.n_peep = (size_t)d->H * d->W * d->Ch;
    // ---- backward sizing ----
    const int N4 = 4 * d->Ch, Ct = d->Cin + d->Ch;
    const int segD[1] = {N4};
    L.d_mw = pick_mw(d->B, d->H, d->W, plain_tiles(Ct), d->precision);
    L.d_qpc = pick_qpc(segD, 1, d->kh, d->kw, 4, d->precision, L.d_mw);
    if (L.d_mw > 1 && !conv_fits_lds(segD, 1, d->kh, d->kw, 4, d->precision, L.d_mw, 1, L.d_qpc)) {
        L.d_mw = 1;
        L.d_qpc = pick_qpc(segD, 1, d->kh, d->kw, 4, d->precision, 1);
    }
    if ((d->flags & VPX_FLAG_SAVE_FOR_BWD) && !conv_fits_lds(segD, 1, d->kh, d->kw, 4, d->precision, L.d_mw, 1, L.d_qpc)) {
        set_error("convlstm: the %dx%d data gradient over %d gate channels does not fit the kernel's LDS stages", d->kh, d->kw, N4);
        return VPX_ERR_UNSUPPORTED;
    }
    L.d_nstage = build_stages(L.d_stage, &L.d_chunks, segD, 1, L.taps, pick_stage_channels(segD, 1, d->kh, d->kw, 4, d->precision, L.d_mw, 1, L.d_qpc), d->precision, L.d_qpc);
    if (L.d_nstage < 0) { set_error("convlstm: too many channel stages in the data-gradient conv (Ch=%d)", d->Ch); return VPX_ERR_UNSUPPORTED; }
    L.d_tiles_full = plain_tiles(Ct);
    L.d_tiles_h = plain_tiles(d->Ch);
    L.n_ctiles = wgrad_make_ctiles(L.ct, WG_MAX_CTILES, d->Cin, d->Ch, d->Cin);
    if (L.n_ctiles < 0) { set_error("convlstm: too many channels for the weight-gradient kernel"); return VPX_ERR_UNSUPPORTED; }
    {
        const int tiles = ((d->W + TILE_W - 1) / TILE_W) * ((d->H + TILE_H - 1) / TILE_H);
        const long long items = (long long)d->T * d->B * tiles;
        const int out_tiles = ((N4 + 63) / 64) * L.n_ctiles * ((L.taps + 8) / 9);
        // rounded DOWN: the launch runs one workgroup per CU in rounds of 256, and 1032 workgroups (86 slices x 12 tiles) took
        // five rounds where 1008 take four (measured: training step 144.8 -> 140.2 ms)
        long long ns = wgrad_target_wgs() / out_tiles;
        if (ns > items) ns = items;
        if (ns > 256) ns = 256;
        if (ns < 1) ns = 1;
        L.n_slices = (int)ns;
        // wgrad2.hip (128-row tiles, its own slice rule): the slab space covers whichever kernel takes the launch; the
        // first-generation count above stays what launch_wgrad is handed when it does
        L.n_slices2 = 0;
        if (L.v2 || wgrad2_wsp(d, L)) {
            const bool half_tail = L.ct[L.n_ctiles - 1].h[1].cn == 0;
            long long ns2 = wgrad2_slices(wgrad2_target_wgs(), (N4 + 127) / 128, L.n_ctiles, half_tail);
            if (ns2 > items) ns2 = items;
            if (ns2 < 1) ns2 = 1;
            L.n_slices2 = (int)ns2;
        }
        L.slab_floats = (size_t)(L.n_slices > L.n_slices2 ? L.n_slices : L.n_slices2) * L.taps * N4 * Ct;
    }
    return VPX_OK;
}


static inline size_t convlstm_bwd_workspace_bytes(const vpx_convlstm_desc* d, const ConvLSTMLayout& L) {
    size_t b = align256(packed_weight_bytes(L.d_tiles_full, L.d_chunks, 4, d->precision, L.d_qpc));  // upper bound over both tilings
    b += align256((size_t)d->T * L.n_state * 4 * sizeof(float));  // dG, all steps
    b += 2 * align256(L.n_state * sizeof(float));                 // dh, dc carries
    b += align256(L.slab_floats * sizeof(float));
    b += align256((size_t)d->T * GATE_BWD_MAX_SLICES * gate_bwd_blocks(d->H * d->W, d->Ch) * 4 * d->Ch * sizeof(float));  // bias-gradient partials
    b += align256((size_t)COLSUM_BLOCKS * 4 * d->Ch * sizeof(float));                                // ... and their second level
    b += 3 * align256((size_t)GATE_BWD_MAX_SLICES * L.n_peep * sizeof(float));                       // per-batch-slice peephole-gradient partials
    if (L.v2)  // dG of all steps in split operand format + the conv2 weight pack of the data gradient
        b += align256((size_t)d->T * L.n_state * 16) + align256(cell2_packed_bytes(conv2_tiles(d->Cin + d->Ch), 3 * (4 * d->Ch / 16)) + 16384 * conv2_tiles(d->Cin + d->Ch));
    if (wgrad2_wsp(d, L))  // split dG of all steps + split copies of x, the output sequence and h0
        b += align256((size_t)d->T * L.n_state * 16) + align256(L.n_x * 4) + align256(L.n_out * 4) + align256(L.n_state * 4);
    if (d->layout == VPX_LAYOUT_NCHW) {
        // staged copies of x, out, dout, dx + states (h0,c0,dhT,dcT,dh0,dc0) + 6 peephole-sized buffers
        b += 2 * align256(L.n_x * 4) + 2 * align256(L.n_out * 4) + 6 * align256(L.n_state * 4) + 6 * align256(L.n_peep * 4);
    }
    return b;
}

// ---- plain stride-1 'same' convolution / weight gradient on NHWC tensors (shared by conv_api, stlstm LN path) ----
struct ConvGeo { int N, H, W; };

// y (+)= conv(src; w) with Co outputs; `transposed`: contraction over w's O axis (data gradient). Returns packed floats used.
static inline int plain_conv(hipStream_t stream, int prec, ConvGeo g, const float* src, int C, int ld, const float* w, long long ld_o,
               int ld_i, int kh, int kw, int Co, bool transposed, const float* bias, float* out, int out_ld,
               bool accumulate, float* wpk, float leaky = 0.0f, bool weights_packed = false) {
    ConvPlan P{};
    int chunks = 0;
    const int segC[1] = {C};
    P.prec = prec;
    const long long m_tiles = (long long)g.N * ((g.H + TILE_H - 1) / TILE_H) * ((g.W + TILE_W - 1) / TILE_W);
    const int ng = plain_groups(Co, m_tiles);
    P.nstage = build_stages(P.stage, &chunks, segC, 1, kh * kw, pick_stage_channels(segC, 1, kh, kw, ng, prec), prec);
    if (P.nstage < 0) { set_error("conv: too many channel stages (C=%d)", C); return VPX_ERR_UNSUPPORTED; }
    PackDesc pd{};
    pd.seg[0] = PackSeg{w, ld_o, ld_i, 0, C};
    memcpy(pd.stage, P.stage, sizeof(ConvStage) * P.nstage);
    pd.nstage = P.nstage; pd.chunks_total = chunks; pd.prec = prec; pd.taps = kh * kw;
    fill_plain_pack(pd, Co, 0, ng);
    pd.transposed = transposed ? 1 : 0; pd.flip = transposed ? 1 : 0;
    if (!weights_packed) VPX_CHECK_HIP(launch_pack_weights(pd, wpk, stream));   // (else: wpk still holds this layer's pack of an earlier call)
    P.B = g.N; P.H = g.H; P.W = g.W; P.kh = kh; P.kw = kw;
    set_plan_tiles(P, 1);
    P.nseg = 1;
    P.seg[0] = ConvSeg{src, (long long)g.H * g.W * ld, C, ld};
    P.chunks_total = chunks;
    P.a_bytes = conv_a_bytes(P.stage, P.nstage, kh, kw);
    P.wpk = wpk;
    PlainEpiArgs ea{};
    ea.bias = bias; ea.Co = Co; ea.split = Co; ea.ng = ng;
    ea.out0 = out; ea.bstride0 = (long long)g.H * g.W * out_ld; ea.ld0 = out_ld;
    ea.accumulate = accumulate ? 1 : 0;
    ea.leaky = leaky;
    // small maps: split K over workgroups (atomic partial sums) — needs a dense or already-initialised destination
    P.ksplit = leaky != 0.0f ? 1 : pick_ksplit(m_tiles * pd.n_tiles, P.nstage);   // partial sums cannot be activated
    if (P.ksplit > 1 && !accumulate) {
        if (out_ld == Co) VPX_CHECK_HIP(vpx_memset_async(out, 0, (size_t)g.N * g.H * g.W * Co * sizeof(float), stream));
        else P.ksplit = 1;
    }
    VPX_CHECK_HIP(launch_conv_plain_f32(P, ea, pd.n_tiles, stream));
    return VPX_OK;
}

// floats of the weight pack plain_conv() writes for this problem (the launch's own rule: N tiling from plain_groups, stage size from
// that tiling) — what a caller that knows the geometry carves
static inline size_t plain_conv_pack_floats(int prec, ConvGeo g, int C, int kh, int kw, int Co) {
    const long long m_tiles = (long long)g.N * ((g.H + TILE_H - 1) / TILE_H) * ((g.W + TILE_W - 1) / TILE_W);
    const int ng = plain_groups(Co, m_tiles);
    ConvStage st[MAX_STAGE];
    int chunks = 0;
    const int segC[1] = {C};
    if (build_stages(st, &chunks, segC, 1, kh * kw, pick_stage_channels(segC, 1, kh, kw, ng, prec), prec) < 0) return 0;
    return packed_weight_bytes(plain_tiles_ng(Co, ng), chunks, ng, prec) / 4;
}

static inline size_t plain_conv_wpk_floats(int C, int Co, int kh, int kw) {
    // upper bound over operand modes and over EVERY N tiling (ng = 1..4, each with the stage size and the tile count it selects):
    // holds whatever plain_groups() picks for the launch's geometry
    size_t best = 0;
    ConvStage st[MAX_STAGE];
    for (int prec = VPX_PREC_F32; prec <= VPX_PREC_BF16; ++prec)
        for (int ng = 1; ng <= 4; ++ng) {
            int chunks = 0;
            const int segC[1] = {C};
            if (build_stages(st, &chunks, segC, 1, kh * kw, pick_stage_channels(segC, 1, kh, kw, ng, prec), prec) < 0) return 0;
            const size_t b = packed_weight_bytes(plain_tiles_ng(Co, ng), chunks, ng, prec) / 4;
            if (b > best) best = b;
        }
    return best;
}

// upper bound of the K slices of a plain weight gradient (sizes the slab workspace)
static inline int wgrad_slices(int N, int H, int W) {
    const long long items = (long long)N * ((W + TILE_W - 1) / TILE_W) * ((H + TILE_H - 1) / TILE_H);
    return (int)(items < 32 ? items : 32);
}
// A layer whose whole dW is ONE output tile (the decoupling adapter: Ch x Ch, 1x1) has as many workgroups as slices: 32 left 7/8 of the chip
// idle (PredRNN-V2 training, B = 128: 67 launches of 180 us per step for 33 MB of operands each). Up to 256 slices there — a slab is Ch x Ch.
static inline int wgrad_slices_1x1(int N, int H, int W) {
    const long long items = (long long)N * ((W + TILE_W - 1) / TILE_W) * ((H + TILE_H - 1) / TILE_H);
    return (int)(items < 256 ? items : 256);
}
// ... and the cap of a general stride-1 layer: 256 where all of dW is one output tile (<= 64 rows x 64 channels, up to 3x3), else 32
static inline int wgrad_slices_for(int N, int H, int W, int Co, int Ci, int kh, int kw) {
    return (Co <= 64 && Ci <= 64 && kh * kw <= 9) ? wgrad_slices_1x1(N, H, W) : wgrad_slices(N, H, W);
}
// slices actually launched: enough for ~1024 workgroups (each slice costs a slab write + a reduce read of all of dW)
static inline int wgrad_pick_slices(int cap, int rows, int n_ctiles, int taps) {
    const int out_tiles = ((rows + 63) / 64) * n_ctiles * ((taps + 8) / 9);
    int ns = wgrad_target_wgs() / out_tiles;   // rounded down: whole rounds of 256 workgroups (see convlstm_layout)
    if (ns > cap) ns = cap;
    return ns < 1 ? 1 : ns;
}

// dw[Co, C, kh, kw] (+)= wgrad(dy [N,HW,Co], x [N,HW,C])
// x2: a second operand pair in the same launch — dw = wgrad(dy, x) + wgrad(dy + N*HW*Co, x2) (the two pairs become
// "time steps" 0 and 1 of one item walk; x2 may live anywhere, its distance from x is just the step stride)
static inline int plain_wgrad(hipStream_t stream, int prec, ConvGeo g, const float* dy, int Co, const float* x, int C, int kh, int kw,
                float* slabs, float* dw, const float* x2 = nullptr, int slice_cap = 0) {   // slice_cap: what `slabs` was sized for (0: wgrad_slices)
    WgradArgs wa{};
    wa.T = x2 ? 2 : 1; wa.x_tstride = x2 ? (long long)(x2 - x) : 0;
    wa.B = g.N; wa.H = g.H; wa.W = g.W; wa.HW = g.H * g.W; wa.kh = kh; wa.kw = kw;
    wa.tiles_x = (g.W + TILE_W - 1) / TILE_W; wa.tiles_y = (g.H + TILE_H - 1) / TILE_H;
    wa.N4 = Co; wa.Cin = C; wa.Ch = 1; wa.Ct = C; wa.ldG = Co; wa.n_out = Co; wa.prec = prec;
    wa.dG = dy; wa.x = x; wa.x_bstride = (long long)wa.HW * C;
    wa.n_ctiles = wgrad_make_ctiles(wa.ct, WG_MAX_CTILES, C, 0, 0);
    if (wa.n_ctiles < 0) { set_error("conv wgrad: too many input channels (%d)", C); return VPX_ERR_UNSUPPORTED; }
    wa.slabs = slabs;
    const int ns = wgrad_pick_slices(slice_cap > 0 ? slice_cap : wgrad_slices(g.N, g.H, g.W), Co, wa.n_ctiles, kh * kw);
    VPX_CHECK_HIP(launch_wgrad(wa, ns, stream));
    VPX_CHECK_HIP(launch_wgrad_reduce(slabs, dw, ns, kh * kw, Co, C, stream));
    return VPX_OK;
}


// backward workspace of one ST-LSTM step (stlstm_bwd_api.hip)
size_t stlstm_bwd_workspace_bytes(const vpx_stlstm_desc* d);
// LayerNorm variant (stlstm_ln_api.hip)
size_t stlstm_ln_reserve_bytes(const vpx_stlstm_desc* d);
size_t stlstm_ln_workspace_bytes(const vpx_stlstm_desc* d);
int stlstm_ln_fwd(const vpx_stlstm_desc* d, const float* x, const float* h, const float* c, const float* m,
                  const float* Wx, const float* Wh, const float* Wm, const float* Wo, const float* Wlast,
                  const float* const* ln, float* h_new, float* c_new, float* m_new, float* delta_c, float* delta_m,
                  void* reserve, Carver& ws, hipStream_t stream);
int stlstm_ln_bwd(const vpx_stlstm_desc* d, const float* x, const float* h, const float* c, const float* m,
                  const float* Wx, const float* Wh, const float* Wm, const float* Wo, const float* Wlast,
                  const float* const* ln, const void* reserve, const float* dh_new, const float* dc_new,
                  const float* dm_new, const float* ddc, const float* ddm, float* dx, float* dh, float* dc, float* dm,
                  float* dWx, float* dWh, float* dWm, float* dWo, float* dWlast, float* const* dln, Carver& ws,
                  hipStream_t stream);

}  // namespace vpx
