// vpx_api.hip — the extern "C" boundary of libvpx_hip.so (declared in include/vpx.h).
// Host-side orchestration only: plan building, workspace carving, per-timestep launches on the caller's stream.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "vpx_host.h"

namespace vpx {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- workspace accounting (vpx_internal.h) ----
static thread_local Carver* g_carver = nullptr;
static thread_local char g_ws_violation[256] = "";

Carver::Carver(void* workspace, size_t bytes)
    : base((char*)workspace), off((256 - ((uintptr_t)workspace & 255)) & 255), cap(bytes), nslot(0), over(false), prev(g_carver) {
    if (off > cap) over = true;
    if (!prev) c5_drop_pending_packs();   // (outermost carver of a call: nothing queued by an earlier call survives)
    g_carver = this;
    g_ws_violation[0] = 0;
}
Carver::~Carver() { g_carver = prev; }

const char* ws_violation() { return g_ws_violation; }
void ws_violation_clear() { g_ws_violation[0] = 0; }

bool ws_write_ok(const void* dst_, size_t bytes, const char* what) {
    const char* dst = (const char*)dst_;
    for (const Carver* c = g_carver; c; c = c->prev) {
        if (dst < c->base || dst >= c->base + c->cap) continue;   // not in this call's workspace: the caller's own tensor
        const char* end = c->base + c->cap;
        size_t room = (size_t)(end - dst);
        for (int i = 0; i < c->nslot; ++i)
            if (dst >= c->slot[i].p && dst < c->slot[i].p + c->slot[i].n) { room = (size_t)(c->slot[i].p + c->slot[i].n - dst); break; }
        if (bytes <= room) return true;
        snprintf(g_ws_violation, sizeof(g_ws_violation), "%s: %zu bytes do not fit the %zu bytes carved for them (workspace sizing rule and launch disagree)",
                 what, bytes, room);
        return false;
    }
    return true;
}

int g_dry_run = 0;
int g_deterministic = 0;
int g_cell3_mode = -1;
int g_cell2_mode = -1;
int g_experiment = 0;
int g_mfma_shape = -1;
int mfma_shape() {
    if (g_mfma_shape < 0) g_mfma_shape = dev_switch("VPX_MFMA_SHAPE", VPX_MFMA_SHAPE_DEFAULT) ? 1 : 0;
    return g_mfma_shape;
}

}  // namespace vpx

using namespace vpx;

extern "C" {

int vpx_version(void) { return VPX_VERSION; }
static int g_option_epoch = 0;
int vpx_option_epoch(void) { return g_option_epoch; }
int vpx_set_deterministic(int on) { const int prev = vpx::g_deterministic; vpx::g_deterministic = on ? 1 : 0; ++g_option_epoch; return prev; }
const char* vpx_last_error(void) { return g_err; }
int vpx_set_option(int option, int value) {
    ++g_option_epoch;
    if (option == VPX_OPT_CELL2) {
        const int prev = cell2_mode();
        vpx::g_cell2_mode = value < 0 ? 0 : (value > 2 ? 2 : value);
        return prev;
    }
    if (option == VPX_OPT_EXPERIMENT) {
        const int prev = vpx::g_experiment;
        vpx::g_experiment = value;
        return prev;
    }
    if (option == VPX_OPT_MFMA_SHAPE) {
        const int prev = mfma_shape();
        vpx::g_mfma_shape = value ? 1 : 0;
        return prev;
    }
    if (option == VPX_OPT_DRY_RUN) {
        const int prev = vpx::g_dry_run;
        vpx::g_dry_run = value ? 1 : 0;
        return prev;
    }
    if (option == VPX_OPT_CELL3) {
        const int prev = cell3_mode();
        vpx::g_cell3_mode = value ? 1 : 0;
        return prev;
    }
    set_error("vpx_set_option: unknown option %d", option);
    return VPX_ERR_ARG;
}

int vpx_nchw_to_nhwc(const float* src, float* dst, int N, int C, int H, int W, void* stream) {
    if (!src || !dst || N < 1 || C < 1 || H < 1 || W < 1) { set_error("vpx_nchw_to_nhwc: bad argument"); return VPX_ERR_ARG; }
    VPX_CHECK_HIP(launch_nchw_to_nhwc(src, dst, N, C, H, W, (hipStream_t)stream));
    return VPX_OK;
}
int vpx_nhwc_to_nchw(const float* src, float* dst, int N, int C, int H, int W, void* stream) {
    if (!src || !dst || N < 1 || C < 1 || H < 1 || W < 1) { set_error("vpx_nhwc_to_nchw: bad argument"); return VPX_ERR_ARG; }
    VPX_CHECK_HIP(launch_nhwc_to_nchw(src, dst, N, C, H, W, (hipStream_t)stream));
    return VPX_OK;
}

/* ------------------------------------------------------------------------------------------------------------- */
size_t vpx_convlstm_reserve_bytes(const vpx_convlstm_desc* d) {
    ConvLSTMLayout L;
    if (check_convlstm_desc(d) != VPX_OK || convlstm_layout(d, L) != VPX_OK) return 0;
    if (!(d->flags & VPX_FLAG_SAVE_FOR_BWD)) return 0;
    // gates [T][B,H,W,4Ch] + cell states [T][B,H,W,Ch]
    size_t b = align256((size_t)d->T * L.n_state * 4 * sizeof(float)) + align256((size_t)d->T * L.n_state * sizeof(float));
    // second-generation cell: the operands in split format stay for the weight gradient (x all frames, h_0, h_1 .. h_T)
    if (L.v2) b += align256(L.n_x * 4) + align256(L.n_state * 4) + align256((size_t)d->T * L.n_state * 4);
    return b;
}

// one cell2 weight pack: an upper bound over both MFMA forms (the q form rounds an odd stage count up by half a stage)
// c3 (convq.hip: c5_kernel<NT, 3>, round 4): the fused step on 16x16-pixel tiles x (4 gates x NT * 4 channels) with 8-channel stages, for the
// launches in which the half tile of cell2_kernel_q leaves most of the chip empty (B = 4 on 64x64 maps: 128 workgroups of four waves,
// one wave per SIMD on half the CUs, each running 36 steps of 96 MFMAs plus a 25 k-cycle epilogue alone). Halving / quartering the N
// tile doubles / quadruples the waves and shortens each wave's critical path. Inference only (it does not write the saved gates).
// VPX_OPT_EXPERIMENT bit 12 keeps the half tile there (A/B runs, tests).
static int c3_nt(const vpx_convlstm_desc* d, const ConvLSTMLayout& L) {   // 0: not applicable; else column tiles per N tile (4 | 2)
    if (!L.v2 || !cell2_q_applicable(d) || d->precision != VPX_PREC_BF16X3 || (d->flags & VPX_FLAG_SAVE_FOR_BWD) || (g_experiment & 4096)) return 0;
    if ((d->Cin & 7) || (d->Ch & 15)) return 0;
    const long long mt = (long long)d->B * ((d->H + 15) / 16) * ((d->W + 15) / 16);
    // Measured (round 4, B = 4, 64x64 maps, Ch = 64: 128 half-tile workgroups of 39 us). First pass: 32-column tiles (512 workgroups) 34 us,
    // 64-column tiles (256) 36 us — the narrow tiles were bound by their K loop's bookkeeping. With that gone (convq.hip, "the K loop's
    // diet") the 64-column tiles win: cell (64,64,64^2) B = 4 0.309 vs 0.336 ms per 10 steps, inference step 1.60 vs 1.63 ms
    // (tools/ab_c3_rule.sh) — half the stage copies per MFMA. With 192 half-tile workgroups (64x64 maps, Ch = 96: configs[3]'s shard) the
    // 64-column c3 tiles are 1.4 % ahead on the whole model (5.70 vs 5.78 ms per step, the (64,96,64^2) cell alone 2 %), equal at B = 8 / 16 on
    // 64x64 maps with Ch = 64 (256 / 512 half-tile workgroups) — and behind once the 128x128 maps' 512 workgroups join (5.88 vs 5.72):
    // c3 up to 256 half-tile workgroups (128 in the first pass of the round, for the 32-column tiles).
    if (mt * L.n_tiles > dev_switch("VPX_C3_MAX", 256)) return 0;
    if (const int f = dev_switch("VPX_C3_NT", 0)) return f;   // (developer build only)
    return (g_experiment & 8192) ? 2 : 4;   // VPX_OPT_EXPERIMENT bit 13: the 32-column tiles (tests, A/B runs)
}

static size_t cell2_wpk_bytes(const vpx_convlstm_desc* d, const ConvLSTMLayout& L) {
    const int S = (d->Cin + d->Ch) / 16;
    const size_t a = cell2_packed_bytes(L.n_tiles, 3 * S), b = cell2_packed_bytes_q(L.n_tiles, S);
    size_t m = a > b ? a : b;
    for (int nt = 2; nt <= 4; nt += 2) { const size_t c = c5_wpk_bytes(d->Cin + d->Ch, d->Ch, nt, 4, 3); if (c > m) m = c; }
    return m;
}

static size_t convlstm_wpk_bytes(const vpx_convlstm_desc* d, const ConvLSTMLayout& L) {
    return L.split ? packed_weight_bytes(L.s_tiles, L.s_chunks, L.s_ng, d->precision)
                   : packed_weight_bytes(L.n_tiles, L.chunks_total, 4, d->precision, L.qpc);
}

static bool hoist_q_problem(const vpx_convlstm_desc* d, ConvQProblem& pr);

static bool second_generation_inference(const vpx_convlstm_desc* d, const ConvLSTMLayout& L) {
    return L.v2 && d->layout == VPX_LAYOUT_NHWC && !(d->flags & VPX_FLAG_SAVE_FOR_BWD);
}

int vpx_convlstm_takes_split_input(const vpx_convlstm_desc* d) {
    ConvLSTMLayout L;
    if (check_convlstm_desc(d) != VPX_OK || convlstm_layout(d, L) != VPX_OK) return 0;
    if (second_generation_inference(d, L)) return 1;
    // small grids (cell3): x only feeds the hoisted projection W_x (*) x_t of all frames; where that runs on the schedule-driven kernel
    // it reads the operand format — the producing stage then writes it directly (no fp32 copy, no conversion launch)
    if (L.v3 && d->layout == VPX_LAYOUT_NHWC && !(d->flags & VPX_FLAG_SAVE_FOR_BWD)) {
        static thread_local ConvQProblem pr;
        return hoist_q_problem(d, pr) && convq_wpk_bytes(pr) ? 1 : 0;
    }
    return 0;
}

int vpx_split_convert(const float* x, void* x_split, long long n_pixels, int C, void* stream) {
    if (!x || !x_split || n_pixels < 0 || C < 8 || (C & 7)) { set_error("vpx_split_convert: NULL tensor or channel count %d not a multiple of 8", C); return VPX_ERR_ARG; }
    VPX_CHECK_HIP(launch_split_convert(x, x_split, n_pixels, C, (hipStream_t)stream));
    return VPX_OK;
}

int vpx_convlstm_writes_split_output(const vpx_convlstm_desc* d) {   // the second-generation cell writes h_t in operand format anyway
    ConvLSTMLayout L;
    if (check_convlstm_desc(d) != VPX_OK || convlstm_layout(d, L) != VPX_OK) return 0;
    if (second_generation_inference(d, L)) return 1;
    return (L.v3 && d->layout == VPX_LAYOUT_NHWC && !(d->flags & VPX_FLAG_SAVE_FOR_BWD)) ? 1 : 0;   // (cell3 keeps h_t in operand format for its own recurrence)
}

}  // extern "C"

// The hoisted input projection W_x (*) x_t of all B*T frames (small-grid paths) as ONE launch of the schedule-driven K = 32 kernel
// (convq.hip): a plain 3x3 'same' convolution whose output channel is the reference row of W (gate-major), on operand-format input.
// 40 frames 64 -> 4x96 channels at 32x32: 84 -> ~25 us against the first-generation launch. VPX_HOIST_Q=0 keeps the latter.
static bool hoist_q_problem(const vpx_convlstm_desc* d, ConvQProblem& pr) {
    static int on = -1;
    if (on < 0) on = dev_switch("VPX_HOIST_Q", 1);
    memset(&pr, 0, sizeof(pr));
    if (g_experiment & 32) return false;   // VPX_OPT_EXPERIMENT bit 5: the first-generation launch (tests, A/B)
    if (!on || d->precision != VPX_PREC_BF16X3 || d->kh != 3 || d->kw != 3 || (d->Cin & 15) || d->Cin < 16 || d->layout != VPX_LAYOUT_NHWC) return false;
    pr.N = d->B * d->T; pr.H = d->H; pr.W = d->W; pr.halo = 2;
    pr.nseg = 1;
    CQSeg& sg = pr.seg[0];
    sg.nT = 1; sg.bstride = (long long)d->H * d->W * d->Cin * 4;
    sg.rowpitch = d->W * d->Cin * 4; sg.colpitch = d->Cin * 4; sg.org = 0; sg.Hs = d->H; sg.Ws = d->W; sg.nstage = d->Cin / 16; sg.c0 = 0;
    pr.ngs = 1;
    ConvQGroupSet& gs = pr.gs[0];
    gs.nt0 = 0; gs.ntn = 8; gs.nterm = 0;
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) gs.term[gs.nterm++] = ConvQTerm{0, ky - 1, kx - 1, ky * 3 + kx};
    pr.periodic = 1;
    pr.s_oc = (long long)(d->Cin + d->Ch) * 9; pr.s_ic = 9;
    pr.Co = 4 * d->Ch; pr.col0 = 0; pr.phases = 0;
    return convq_wpk_bytes(pr) != 0;
}

extern "C" {

size_t vpx_convlstm_workspace_bytes(const vpx_convlstm_desc* d) {
    ConvLSTMLayout L;
    if (check_convlstm_desc(d) != VPX_OK || convlstm_layout(d, L) != VPX_OK) return 0;
    // forward: packed weights + cell scratch (+ NCHW staging)
    size_t fwd = align256(convlstm_wpk_bytes(d, L)) + align256(L.n_state * sizeof(float));
    if (L.split) fwd += align256(4 * L.n_state * sizeof(float));  // gate pre-activations of one step
    if (L.hoist)  // input projection of all steps [B,T,HW,4Ch] + the two weight packs (x columns, h columns)
        fwd += align256(4 * L.n_out * sizeof(float)) + align256(packed_weight_bytes(L.s_tiles, L.hx_chunks, L.s_ng, d->precision)) +
               align256(packed_weight_bytes(L.s_tiles, L.hh_chunks, L.s_ng, d->precision));
    if (L.v3)  // input projection of all steps + its weight pack + the slice-major recurrent weights + split h0 and a two-slot ring of h_t
        fwd += align256(4 * L.n_out * sizeof(float)) + align256(packed_weight_bytes(L.s_tiles, L.hx_chunks, L.s_ng, d->precision)) +
               align256(cell3_packed_bytes(d->Ch)) + 3 * align256(L.n_state * 4);
    if (L.v2)  // packed weights of cell2 (q form: up to two packs, one per set of present operands) + split copies of x, h0 and a two-slot ring of h_t
        fwd += 2 * align256(cell2_wpk_bytes(d, L)) + align256(L.n_x * 4) + 3 * align256(L.n_state * 4);
    if (L.hoist || L.v3) {  // the hoisted projection on convq: its weight pack + x in operand format
        static thread_local ConvQProblem hq;
        if (hoist_q_problem(d, hq)) fwd += align256(convq_wpk_bytes(hq)) + align256(L.n_x * 4);
    }
    if (d->layout == VPX_LAYOUT_NCHW)
        fwd += align256(L.n_x * 4) + align256(L.n_out * 4) + 4 * align256(L.n_state * 4) + 3 * align256(L.n_peep * 4);
    // backward (only with SAVE_FOR_BWD): packed dgrad weights + dG for all steps + dh/dc carries + wgrad K-slice slabs
    size_t bwd = 0;
    if (d->flags & VPX_FLAG_SAVE_FOR_BWD)
        bwd = convlstm_bwd_workspace_bytes(d, L);
    return (fwd > bwd ? fwd : bwd) + 256;
}

int vpx_convlstm_seq_fwd(const vpx_convlstm_desc* d, const float* x, const float* h0, const float* c0,
                         const float* W, const float* bias, const float* Wci, const float* Wcf, const float* Wco,
                         float* out, float* hT, float* cT, void* reserve, size_t reserve_bytes, void* workspace,
                         size_t workspace_bytes, void* stream_) {
    int rc = check_convlstm_desc(d);
    if (rc != VPX_OK) return rc;
    ConvLSTMLayout L;
    if ((rc = convlstm_layout(d, L)) != VPX_OK) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    if (!W || !out) { set_error("vpx_convlstm_seq_fwd: W and out must not be NULL"); return VPX_ERR_ARG; }
    const bool peep = Wci || Wcf || Wco;
    if (peep && !(Wci && Wcf && Wco)) { set_error("vpx_convlstm_seq_fwd: peephole tensors must be given together"); return VPX_ERR_ARG; }
    const bool save = (d->flags & VPX_FLAG_SAVE_FOR_BWD) != 0;
    const bool x_split = (d->flags & VPX_FLAG_X_SPLIT) != 0;
    // the caller's workspace still holds every weight pack of an earlier call with the same descriptor, weights and operand set
    const bool wp = (d->flags & VPX_FLAG_WEIGHTS_PACKED) != 0;
    const bool out_split = (d->flags & VPX_FLAG_OUT_SPLIT) != 0;
    if (out_split && !vpx_convlstm_writes_split_output(d)) {
        set_error("vpx_convlstm_seq_fwd: VPX_FLAG_OUT_SPLIT given, but this descriptor's forward writes fp32 only (vpx_convlstm_writes_split_output)");
        return VPX_ERR_ARG;
    }
    if (x_split && !vpx_convlstm_takes_split_input(d)) {
        set_error("vpx_convlstm_seq_fwd: VPX_FLAG_X_SPLIT given, but this descriptor's forward takes fp32 input (vpx_convlstm_takes_split_input)");
        return VPX_ERR_ARG;
    }
    if (save && (!reserve || reserve_bytes < vpx_convlstm_reserve_bytes(d))) {
        set_error("vpx_convlstm_seq_fwd: reserve too small (%zu < %zu)", reserve_bytes, vpx_convlstm_reserve_bytes(d));
        return VPX_ERR_WORKSPACE;
    }
    if (!workspace || workspace_bytes < vpx_convlstm_workspace_bytes(d)) {
        set_error("vpx_convlstm_seq_fwd: workspace too small (%zu < %zu)", workspace_bytes, vpx_convlstm_workspace_bytes(d));
        return VPX_ERR_WORKSPACE;
    }
    const int B = d->B, T = d->T, Cin = d->Cin, Ch = d->Ch, H = d->H, Wd = d->W;
    const size_t HW = (size_t)H * Wd;
    Carver ws(workspace, workspace_bytes);
    float* wpk = ws.take(convlstm_wpk_bytes(d, L) / sizeof(float));
    float* c_scratch = ws.take(L.n_state);
    float* pre_scratch = L.split ? ws.take(4 * L.n_state) : nullptr;
    const bool hoist = L.hoist && x != nullptr;
    float *pre_all = nullptr, *wpk_hx = nullptr, *wpk_hh = nullptr;
    if (hoist) {
        pre_all = ws.take(4 * L.n_out);
        wpk_hx = ws.take(packed_weight_bytes(L.s_tiles, L.hx_chunks, L.s_ng, d->precision) / sizeof(float));
        wpk_hh = ws.take(packed_weight_bytes(L.s_tiles, L.hh_chunks, L.s_ng, d->precision) / sizeof(float));
    }
    char *wpk3 = nullptr, *h0_sp3 = nullptr, *h_ring3[2] = {nullptr, nullptr};
    if (L.v3) {
        pre_all = ws.take(4 * L.n_out);
        wpk_hx = ws.take(packed_weight_bytes(L.s_tiles, L.hx_chunks, L.s_ng, d->precision) / sizeof(float));
        wpk3 = (char*)ws.take(cell3_packed_bytes(Ch) / sizeof(float));
        h0_sp3 = (char*)ws.take(L.n_state);
        h_ring3[0] = (char*)ws.take(L.n_state);
        h_ring3[1] = (char*)ws.take(L.n_state);
    }
    static thread_local ConvQProblem hq;
    char *wpk_hq = nullptr, *x_sp_hq = nullptr;
    const bool hoist_q = (L.hoist || L.v3) && hoist_q_problem(d, hq);
    if (hoist_q) {
        wpk_hq = (char*)ws.take(align256(convq_wpk_bytes(hq)) / sizeof(float));
        x_sp_hq = (char*)ws.take(L.n_x);
    }
    char *wpk2 = nullptr, *wpk2b = nullptr, *x_sp = nullptr, *h0_sp = nullptr, *h_ring[2] = {nullptr, nullptr}, *h_sp_all = nullptr;
    if (L.v2) {
        wpk2 = (char*)ws.take(cell2_wpk_bytes(d, L) / sizeof(float));
        wpk2b = (char*)ws.take(cell2_wpk_bytes(d, L) / sizeof(float));
        x_sp = (char*)ws.take(L.n_x);
        h0_sp = (char*)ws.take(L.n_state);
        h_ring[0] = (char*)ws.take(L.n_state);
        h_ring[1] = (char*)ws.take(L.n_state);
        if (save) {  // keep the split operands for the backward's weight gradient: they live in the reserve instead
            char* r = (char*)reserve + align256((size_t)T * L.n_state * 4 * sizeof(float)) + align256((size_t)T * L.n_state * sizeof(float));
            x_sp = r; r += align256(L.n_x * 4);
            h0_sp = r; r += align256(L.n_state * 4);
            h_sp_all = r;
        }
    }

    VPX_CHECK_CARVE(ws, "vpx_convlstm_seq_fwd");
    // ---- layout adaptation (reference NCHW -> native NHWC) ----
    const float *xn = x, *h0n = h0, *c0n = c0, *wci = Wci, *wcf = Wcf, *wco = Wco;
    float *outn = out, *hTn = hT, *cTn = cT;
    if (d->layout == VPX_LAYOUT_NCHW) {
        float* bx = ws.take(L.n_x);
        outn = ws.take(L.n_out);
        float* bh0 = ws.take(L.n_state);
        float* bc0 = ws.take(L.n_state);
        float* bhT = ws.take(L.n_state);
        float* bcT = ws.take(L.n_state);
        float* p0 = ws.take(L.n_peep);
        float* p1 = ws.take(L.n_peep);
        float* p2 = ws.take(L.n_peep);
        VPX_CHECK_CARVE(ws, "vpx_convlstm_seq_fwd");
        if (x) { VPX_CHECK_HIP(launch_nchw_to_nhwc(x, bx, B * T, Cin, H, Wd, stream)); xn = bx; }
        if (h0) { VPX_CHECK_HIP(launch_nchw_to_nhwc(h0, bh0, B, Ch, H, Wd, stream)); h0n = bh0; }
        if (c0) { VPX_CHECK_HIP(launch_nchw_to_nhwc(c0, bc0, B, Ch, H, Wd, stream)); c0n = bc0; }
        if (peep) {
            VPX_CHECK_HIP(launch_nchw_to_nhwc(Wci, p0, 1, Ch, H, Wd, stream));
            VPX_CHECK_HIP(launch_nchw_to_nhwc(Wcf, p1, 1, Ch, H, Wd, stream));
            VPX_CHECK_HIP(launch_nchw_to_nhwc(Wco, p2, 1, Ch, H, Wd, stream));
            wci = p0; wcf = p1; wco = p2;
        }
        hTn = hT ? bhT : nullptr;
        cTn = cT ? bcT : nullptr;
    }

    // ---- weight repack: OIHW [4Ch, Cin+Ch, kh, kw] -> per-tile K-chunk stream ----
    int gp[4];
    gate_positions(d->gate_order, gp);
    PackDesc pd{};
    const long long ld_o = (long long)(Cin + Ch) * L.taps;
    pd.seg[0] = PackSeg{W, ld_o, L.taps, 0, Cin};
    pd.seg[1] = PackSeg{W, ld_o, L.taps, Cin, Ch};
    pd.prec = d->precision;
    pd.taps = L.taps;
    pd.transposed = 0;
    pd.flip = 0;
    // q form (16x16x32 MFMAs): the K = 32 steps pair taps over the sequence of PRESENT stages, so steps with different operand
    // sets (t = 0 without an initial state: x only; no input tensor: h only) read different packs — at most two per call
    const int qform = (L.v2 && cell2_q_applicable(d)) ? 1 : 0;
    const int c3nt = c3_nt(d, L);   // > 0: the small-grid 3x3 form on the c5 machinery takes the steps (its packs replace the q packs)
    const int combo0 = (xn ? 1 : 0) | (h0n ? 2 : 0), combo1 = (xn ? 1 : 0) | 2;   // operand sets of step 0 / of steps t >= 1
    auto pack_of = [&](int combo) -> char* { return combo == combo1 ? wpk2 : wpk2b; };
    if (L.v2) {
        Cell2Pack pk{};
        pk.w = W; pk.Ch = Ch; pk.Ct = Cin + Ch; pk.n_tiles = L.n_tiles;
        memcpy(pk.gate_pos, gp, sizeof(gp));
        if (!qform) {
            pk.chunks_total = 3 * ((Cin + Ch) / 16);
            for (int s = 0; s < (Cin + Ch) / 16; ++s) pk.stage_col[s] = 16 * s;  // x stages first, then h: columns of [x | h] in order
            if (!wp) VPX_CHECK_HIP(launch_cell2_pack(pk, wpk2, stream));
        } else {
            for (int pass = 0; pass < 2; ++pass) {
                const int combo = pass ? combo0 : combo1;
                // combo1 serves steps t >= 1 (and step 0 when it has the same operands); combo0 only when it differs and is not empty
                if (pass == 0 ? (T < 2 && combo0 != combo1) : (combo0 == combo1 || combo0 == 0)) continue;
                pk.qform = 1; pk.S = 0;
                if (combo & 1) for (int s = 0; s < Cin / 16; ++s) pk.stage_col[pk.S++] = 16 * s;
                if (combo & 2) for (int s = 0; s < Ch / 16; ++s) pk.stage_col[pk.S++] = Cin + 16 * s;
                pk.chunks_total = cell2_qchunks(pk.S);
                if (!wp && !c3nt) VPX_CHECK_HIP(launch_cell2_pack(pk, pack_of(combo), stream));
                if (!wp && c3nt) {   // the same operand sets, packed for c5_kernel<NT, 3>
                    C5Job j{};
                    C5PackRange pr[2];
                    const long long so = (long long)(Cin + Ch) * 9;
                    if (combo & 1) { j.r_n[j.nrange] = Cin; pr[j.nrange] = C5PackRange{W, so, 9, 0, {gp[0] * Ch, gp[1] * Ch, gp[2] * Ch, gp[3] * Ch}}; ++j.nrange; }
                    if (combo & 2) { j.r_n[j.nrange] = Ch; pr[j.nrange] = C5PackRange{W, so, 9, Cin, {gp[0] * Ch, gp[1] * Ch, gp[2] * Ch, gp[3] * Ch}}; ++j.nrange; }
                    j.Co = Ch; j.wpk = pack_of(combo);
                    int rcp = c5_prepare_job(j, c3nt, pr, 4, 0, false, stream, 0, 3);
                    if (rcp) return rcp;
                }
            }
        }
        // operands of the steps in split form: the whole input sequence and the initial hidden state, once
        if (xn && x_split) x_sp = const_cast<char*>(reinterpret_cast<const char*>(xn));   // the producer already wrote operands
        else if (xn) VPX_CHECK_HIP(launch_split_convert(xn, x_sp, (long long)B * T * (long long)HW, Cin, stream));
        if (h0n) VPX_CHECK_HIP(launch_split_convert(h0n, h0_sp, (long long)B * (long long)HW, Ch, stream));
    } else if (L.split) {  // plain layout: output channel n = reference row n of W (gate-major), s_ng * 32 rows per N tile
        memcpy(pd.stage, L.s_stage, sizeof(ConvStage) * L.s_nstage);
        pd.nstage = L.s_nstage;
        pd.chunks_total = L.s_chunks;
        fill_plain_pack(pd, 4 * Ch, 0, L.s_ng);
    } else {
        memcpy(pd.stage, L.stage, sizeof(ConvStage) * L.nstage);
        pd.nstage = L.nstage;
        pd.chunks_total = L.chunks_total;
        pd.qpc = L.qpc;
        pd.n_tiles = L.n_tiles;
        pd.NG = 4;
        for (int g = 0; g < 4; ++g) { pd.rowbase[0][g] = pd.rowbase[1][g] = gp[g] * Ch; pd.goff[g] = 0; }
        pd.tile_stride = 32;
        pd.nch = Ch;
    }
    if (!L.v2 && !L.v3 && !wp) VPX_CHECK_HIP(launch_pack_weights(pd, wpk, stream));
    if (L.v3) {
        Cell3Pack pk{W, Cin, Ch, Cin + Ch, Ch / 8, 9 * Ch / 16, {gp[0], gp[1], gp[2], gp[3]}};
        if (!wp) VPX_CHECK_HIP(launch_cell3_pack(pk, wpk3, stream));
        if (h0n) VPX_CHECK_HIP(launch_split_convert(h0n, h0_sp3, (long long)B * (long long)HW, Ch, stream));
    }

    const bool hoist_x = hoist || (L.v3 && xn != nullptr);
    if (hoist_x) {
        // two packs (x columns / h columns of W, reference row order), then W_x * x for all B*T frames in one launch
        PackDesc px{}, ph{};
        px.seg[0] = PackSeg{W, ld_o, L.taps, 0, Cin};
        memcpy(px.stage, L.hx_stage, sizeof(ConvStage) * L.hx_nstage);
        px.nstage = L.hx_nstage; px.chunks_total = L.hx_chunks; px.prec = d->precision; px.taps = L.taps;
        fill_plain_pack(px, 4 * Ch, 0, L.s_ng);
        if (!wp && !hoist_q) VPX_CHECK_HIP(launch_pack_weights(px, wpk_hx, stream));
        if (hoist) {
            ph.seg[0] = PackSeg{W, ld_o, L.taps, Cin, Ch};
            memcpy(ph.stage, L.hh_stage, sizeof(ConvStage) * L.hh_nstage);
            ph.nstage = L.hh_nstage; ph.chunks_total = L.hh_chunks; ph.prec = d->precision; ph.taps = L.taps;
            fill_plain_pack(ph, 4 * Ch, 0, L.s_ng);
            if (!wp) VPX_CHECK_HIP(launch_pack_weights(ph, wpk_hh, stream));
        }
        if (hoist_q) {
            const char* xs = reinterpret_cast<const char*>(xn);
            if (!x_split) { VPX_CHECK_HIP(launch_split_convert(xn, x_sp_hq, (long long)B * T * (long long)HW, Cin, stream)); xs = x_sp_hq; }
            hq.seg[0].sp = xs;
            hq.w = W;
            ConvQEpiArgs ea{};
            ea.Co = 4 * Ch; ea.split = 4 * Ch;
            ea.oys = 1; ea.oxs = 1; ea.Hmem = H; ea.Wmem = Wd;
            ea.out0 = pre_all; ea.bstride0 = (long long)(HW * 4 * Ch); ea.ld0 = 4 * Ch;
            if ((rc = convq_run(hq, ea, wpk_hq, wp, stream)) != VPX_OK) return rc;
        } else {
            ConvPlan PX{};
            PX.B = B * T; PX.H = H; PX.W = Wd; PX.kh = d->kh; PX.kw = d->kw;
            set_plan_tiles(PX, 1);
            PX.nseg = 1;
            PX.seg[0] = ConvSeg{xn, (long long)(HW * Cin), Cin, 0};   // x is [B][T][HW][Cin]: B*T dense images
            PX.nstage = L.hx_nstage;
            memcpy(PX.stage, L.hx_stage, sizeof(ConvStage) * L.hx_nstage);
            PX.chunks_total = L.hx_chunks; PX.prec = d->precision;
            PX.a_bytes = conv_a_bytes(L.hx_stage, L.hx_nstage, d->kh, d->kw, 1);
            PX.wpk = wpk_hx;
            PlainEpiArgs pa{};
            pa.Co = 4 * Ch; pa.split = 4 * Ch; pa.ng = L.s_ng;
            pa.out0 = pre_all; pa.bstride0 = (long long)(HW * 4 * Ch); pa.ld0 = 4 * Ch;
            VPX_CHECK_HIP(launch_conv_plain_f32(PX, pa, L.s_tiles, stream));
        }
    }

    float* gates_all = nullptr;
    float* cs_all = nullptr;
    if (save) {
        gates_all = (float*)reserve;
        cs_all = (float*)((char*)reserve + align256((size_t)T * L.n_state * 4 * sizeof(float)));
    }

    // ---- time loop: one fused conv + gate + state-update launch per step ----
    for (int t = 0; t < T; ++t) {
        ConvPlan P{};
        P.B = B; P.H = H; P.W = Wd; P.kh = d->kh; P.kw = d->kw;
        set_plan_tiles(P, L.split ? 1 : L.mw);
        P.nseg = 2;
        P.seg[0] = ConvSeg{xn ? xn + (size_t)t * HW * Cin : nullptr, (long long)((size_t)T * HW * Cin), Cin, 0};
        const float* hprev = (t == 0) ? h0n : outn + (size_t)(t - 1) * HW * Ch;
        const long long hprev_bs = (long long)((t == 0) ? HW * Ch : (size_t)T * HW * Ch);
        P.seg[1] = ConvSeg{hprev, hprev_bs, Ch, 0};
        const ConvStage* stages = L.split ? L.s_stage : L.stage;
        const int nstages = L.split ? L.s_nstage : L.nstage;
        P.nstage = 0;
        for (int s = 0; s < nstages; ++s) {
            const float* src = P.seg[stages[s].seg].ptr;
            if (src) P.stage[P.nstage++] = stages[s];  // absent source == all-zero operand: its K range is skipped
        }
        P.chunks_total = L.split ? L.s_chunks : L.chunks_total; P.prec = d->precision;
        P.qpc = L.split ? 0 : L.qpc;
        P.a_bytes = conv_a_bytes(stages, nstages, d->kh, d->kw, L.split ? 1 : L.mw);
        P.wpk = wpk;

        ConvLSTMStepArgs ea{};
        ea.bias = bias;
        memcpy(ea.gate_pos, gp, sizeof(gp));
        ea.Ch = Ch;
        if (save) {
            ea.c_in = (t == 0) ? c0n : cs_all + (size_t)(t - 1) * L.n_state;
            ea.c_out = cs_all + (size_t)t * L.n_state;
            ea.gates = gates_all + (size_t)t * L.n_state * 4;
        } else {
            ea.c_in = (t == 0) ? c0n : c_scratch;
            ea.c_out = (t == T - 1 && cTn) ? cTn : c_scratch;  // the last step writes c_T where the caller wants it
            ea.gates = nullptr;
        }
        ea.wci = wci; ea.wcf = wcf; ea.wco = wco;
        ea.h_out = outn + (size_t)t * HW * Ch;
        ea.h_bstride = (long long)((size_t)T * HW * Ch);
        if (L.v3) {
            Cell3Args c3{};
            c3.B = B; c3.H = H; c3.W = Wd;
            c3.h_sp = (t == 0) ? (h0n ? h0_sp3 : nullptr) : h_ring3[(t - 1) & 1];
            c3.h_bstride = (long long)(HW * Ch * 4);
            c3.wpk = wpk3;
            c3.pre = xn ? pre_all + (size_t)t * HW * 4 * Ch : nullptr;
            c3.pre_bstride = (long long)((size_t)T * HW * 4 * Ch);
            c3.h_sp_out = (t + 1 < T) ? h_ring3[t & 1] : nullptr;
            c3.h_sp_out_bstride = (long long)(HW * Ch * 4);
            if (out_split) {
                // the caller's `out` [B][T][HW][Ch] takes the steps' operand-format h_t (it is also where step t + 1 reads h_t); no fp32
                // sequence — the last step still hands h_T out in fp32 where the caller wants it
                const long long seq_bs = (long long)((size_t)T * HW * Ch * 4);
                char* const out_sp = reinterpret_cast<char*>(outn);
                if (t > 0) { c3.h_sp = out_sp + (size_t)(t - 1) * HW * Ch * 4; c3.h_bstride = seq_bs; }
                c3.h_sp_out = out_sp + (size_t)t * HW * Ch * 4;
                c3.h_sp_out_bstride = seq_bs;
                ea.h_out = (t == T - 1) ? hTn : nullptr;
                ea.h_bstride = (long long)(HW * Ch);
            }
            c3.ea = ea;
            VPX_CHECK_HIP(launch_cell3(c3, stream));
        } else if (L.v2) {
            Cell2Plan P2{};
            P2.B = B; P2.H = H; P2.W = Wd; P2.tiles_x = (Wd + 15) / 16; P2.tiles_y = (H + 31) / 32; P2.n_tiles = L.n_tiles;
            P2.chunks_total = 3 * ((Cin + Ch) / 16);
            P2.wpk = wpk2;
            P2.qform = qform;
            P2.plain = d->precision == VPX_PREC_BF16 ? 1 : 0;
            // h_t in operand format: the reserve (training), the caller's `out` buffer laid out [B][T][HW][Ch] (OUT_SPLIT), or a two-slot ring
            const long long hsp_bs = out_split ? (long long)((size_t)T * HW * Ch * 4) : (long long)(HW * Ch * 4);
            auto h_slot = [&](int tt) -> char* {
                if (out_split) return reinterpret_cast<char*>(outn) + (size_t)tt * HW * Ch * 4;
                return h_sp_all ? h_sp_all + (size_t)tt * L.n_state * 4 : h_ring[tt & 1];
            };
            if (out_split) {   // no fp32 sequence; the last step still hands h_T out in fp32 where the caller wants it
                ea.h_out = (t == T - 1) ? hTn : nullptr;
                ea.h_bstride = (long long)(HW * Ch);
            }
            const char* hprev_sp = (t == 0) ? (h0n ? h0_sp : nullptr) : h_slot(t - 1);
            P2.seg[0] = Cell2Seg{xn ? x_sp + (size_t)t * HW * Cin * 4 : nullptr, (long long)((size_t)T * HW * Cin * 4), Cin, 0};
            P2.seg[1] = Cell2Seg{hprev_sp, (t == 0) ? (long long)(HW * Ch * 4) : hsp_bs, Ch, 0};
            P2.nx = xn ? Cin / 16 : 0;
            P2.nh = hprev_sp ? Ch / 16 : 0;
            P2.hs_off = Cin / 16;
            if (qform) {   // the pack of this step's operand set
                P2.chunks_total = cell2_qchunks(P2.nx + P2.nh);
                P2.wpk = pack_of((P2.nx ? 1 : 0) | (P2.nh ? 2 : 0));
            }
            // the split copy of h_t feeds step t+1 only: the last step does not need it
            char* const hsp_t = (t + 1 < T || out_split) ? h_slot(t) : nullptr;
            if (c3nt) {
                C5Plan cp{};
                cp.B = B; cp.H = H; cp.W = Wd; cp.ks = 3;
                cp.src[0] = C5Src{P2.seg[0].sp, P2.seg[0].bstride, Cin * 4, 0};
                cp.src[1] = C5Src{P2.seg[1].sp, P2.seg[1].bstride, Ch * 4, 0};
                C5Job& j = cp.job[cp.njobs++];
                j = C5Job{};
                if (P2.nx) { j.r_src[j.nrange] = 0; j.r_n[j.nrange] = Cin; ++j.nrange; }
                if (P2.nh) { j.r_src[j.nrange] = 1; j.r_n[j.nrange] = Ch; ++j.nrange; }
                j.epi = 3; j.Co = Ch; j.Ch = Ch; j.wpk = P2.wpk;
                j.e_in0 = ea.c_in; j.e_in1 = ea.wci; j.e_in2 = ea.wcf; j.e_in3 = ea.wco; j.bias = ea.bias;
                for (int g = 0; g < 4; ++g) j.gate_pos[g] = ea.gate_pos[g];
                j.e_out[0] = ea.h_out; j.e_out[1] = ea.c_out; j.h_bstride = ea.h_bstride;
                j.e_sp = hsp_t; j.sp_bstride = hsp_bs;
                const int rcj = c5_prepare_job(j, c3nt, nullptr, 4, 0, true, stream, 0, 3);   // (derived fields only: the pack is in place)
                if (rcj) return rcj;
                VPX_CHECK_HIP(launch_c5(cp, c3nt, stream));
            } else
            VPX_CHECK_HIP(launch_cell2(P2, ea, hsp_t, hsp_bs, stream));
        } else if (hoist) {
            // the step contracts only h_{t-1} and accumulates (atomics when K is split) into its slice of the hoisted input
            // projection; the pointwise kernel reads that slice (batch stride T*HW*4Ch) and writes gates / c / h
            float* pre_t = pre_all + (size_t)t * HW * 4 * Ch;
            const long long pre_bs = (long long)((size_t)T * HW * 4 * Ch);
            if (hprev) {
                ConvPlan PH{};
                PH.B = B; PH.H = H; PH.W = Wd; PH.kh = d->kh; PH.kw = d->kw;
                set_plan_tiles(PH, 1);
                PH.nseg = 1;
                PH.seg[0] = ConvSeg{hprev, hprev_bs, Ch, 0};
                PH.nstage = L.hh_nstage;
                memcpy(PH.stage, L.hh_stage, sizeof(ConvStage) * L.hh_nstage);
                PH.chunks_total = L.hh_chunks; PH.prec = d->precision;
                PH.a_bytes = conv_a_bytes(L.hh_stage, L.hh_nstage, d->kh, d->kw, 1);
                PH.wpk = wpk_hh;
                PH.ksplit = L.hh_split;
                PlainEpiArgs pa{};
                pa.Co = 4 * Ch; pa.split = 4 * Ch; pa.ng = L.s_ng;
                pa.out0 = pre_t; pa.bstride0 = pre_bs; pa.ld0 = 4 * Ch;
                pa.accumulate = 1;
                VPX_CHECK_HIP(launch_conv_plain_f32(PH, pa, L.s_tiles, stream));
            }
            VPX_CHECK_HIP(launch_convlstm_pointwise(ea, pre_t, B, (long long)HW, stream, pre_bs));
        } else if (L.split) {
            // pre-activations of all four gates by a K-split plain convolution (atomic partial sums), then the gates
            float* pre = ea.gates ? ea.gates : pre_scratch;  // with SAVE_FOR_BWD the reserve slot doubles as scratch
            VPX_CHECK_HIP(vpx_memset_async(pre, 0, 4 * L.n_state * sizeof(float), stream));
            PlainEpiArgs pa{};
            pa.Co = 4 * Ch; pa.split = 4 * Ch; pa.ng = L.s_ng;
            pa.out0 = pre; pa.bstride0 = (long long)(HW * 4 * Ch); pa.ld0 = 4 * Ch;
            P.ksplit = L.split;
            if (P.nstage == 0) P.ksplit = 0;
            if (P.nstage > 0) VPX_CHECK_HIP(launch_conv_plain_f32(P, pa, L.s_tiles, stream));
            VPX_CHECK_HIP(launch_convlstm_pointwise(ea, pre, B, (long long)HW, stream));
        } else {
            VPX_CHECK_HIP(launch_convlstm_step_f32(P, ea, L.n_tiles, stream));
        }
    }

    // ---- final states ----
    if (cTn && save)
        VPX_CHECK_HIP(vpx_memcpy_async(cTn, cs_all + (size_t)(T - 1) * L.n_state, L.n_state * sizeof(float), hipMemcpyDeviceToDevice, stream));
    if (hTn && !out_split)
        VPX_CHECK_HIP(vpx_memcpy2d_async(hTn, HW * Ch * sizeof(float), outn + (size_t)(T - 1) * HW * Ch,
                                       (size_t)T * HW * Ch * sizeof(float), HW * Ch * sizeof(float), B,
                                       hipMemcpyDeviceToDevice, stream));
    if (d->layout == VPX_LAYOUT_NCHW) {
        VPX_CHECK_HIP(launch_nhwc_to_nchw(outn, out, B * T, Ch, H, Wd, stream));
        if (hT) VPX_CHECK_HIP(launch_nhwc_to_nchw(hTn, hT, B, Ch, H, Wd, stream));
        if (cT) VPX_CHECK_HIP(launch_nhwc_to_nchw(cTn, cT, B, Ch, H, Wd, stream));
    }
    return VPX_OK;
}

/* ---- plain conv ------------------------------------------------------------------------------------------------ */
size_t vpx_conv2d_workspace_bytes(int Ci, int Co, int kh, int kw) {
    if (Ci < 1 || Co < 1 || kh < 1 || kw < 1) return 0;
    ConvStage st[MAX_STAGE];
    int chunks = 0;
    const int segC[1] = {Ci};
    // one query serves vpx_conv2d_nhwc_fwd (stage size of the 4-group tile) and vpx_conv2d_nhwc_fwd_ex (plain_conv: stage size of
    // the tiling it picks) in every operand mode: the larger of the two packs. (Round 4 sized only the first — the 5x5 layers
    // with <= 32 outputs of the TrajGRU flow generator then packed 4 % more than the workspace held.)
    size_t best = plain_conv_wpk_floats(Ci, Co, kh, kw) * 4;
    for (int prec = VPX_PREC_F32; prec <= VPX_PREC_BF16; ++prec) {
        if (build_stages(st, &chunks, segC, 1, kh * kw, pick_stage_channels(segC, 1, kh, kw, 4, prec), prec) < 0) return 0;
        const size_t b = packed_weight_bytes(plain_tiles(Co), chunks, plain_groups(Co), prec);
        if (b > best) best = b;
    }
    return align256(best) + 256;
}

int vpx_conv2d_nhwc_fwd(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Ci,
                        int Co, int kh, int kw, int precision, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!x || !w || !y || N < 1 || H < 1 || W < 1 || Ci < 1 || Co < 1 || !(kh & 1) || !(kw & 1) || kh > 7 || kw > 7) {
        set_error("vpx_conv2d_nhwc_fwd: bad argument");
        return VPX_ERR_ARG;
    }
    if ((precision < VPX_PREC_F32 || precision > VPX_PREC_BF16)) { set_error("vpx_conv2d_nhwc_fwd: precision %d not implemented", precision); return VPX_ERR_UNSUPPORTED; }
    hipStream_t stream = (hipStream_t)stream_;
    ConvPlan P{};
    int chunks = 0;
    const int segC[1] = {Ci};
    P.nstage = build_stages(P.stage, &chunks, segC, 1, kh * kw, pick_stage_channels(segC, 1, kh, kw, 4, precision), precision);
    if (P.nstage < 0) { set_error("vpx_conv2d_nhwc_fwd: too many channel stages"); return VPX_ERR_UNSUPPORTED; }
    const int n_tiles = plain_tiles(Co);
    if (!workspace || workspace_bytes < vpx_conv2d_workspace_bytes(Ci, Co, kh, kw)) {
        set_error("vpx_conv2d_nhwc_fwd: workspace too small");
        return VPX_ERR_WORKSPACE;
    }
    Carver ws(workspace, workspace_bytes);
    float* wpk = ws.take(packed_weight_bytes(n_tiles, chunks, plain_groups(Co), precision) / sizeof(float));
    VPX_CHECK_CARVE(ws, "vpx_conv2d_nhwc_fwd");
    PackDesc pd{};
    pd.seg[0] = PackSeg{w, (long long)Ci * kh * kw, kh * kw, 0, Ci};
    memcpy(pd.stage, P.stage, sizeof(ConvStage) * P.nstage);
    pd.nstage = P.nstage; pd.chunks_total = chunks; pd.prec = precision; pd.taps = kh * kw;
    fill_plain_pack(pd, Co, 0);
    VPX_CHECK_HIP(launch_pack_weights(pd, wpk, stream));
    P.B = N; P.H = H; P.W = W; P.kh = kh; P.kw = kw;
    P.tiles_x = (W + TILE_W - 1) / TILE_W; P.tiles_y = (H + TILE_H - 1) / TILE_H;
    P.nseg = 1;
    P.seg[0] = ConvSeg{x, (long long)H * W * Ci, Ci, 0};
    P.chunks_total = chunks; P.prec = precision;
    P.a_bytes = conv_a_bytes(P.stage, P.nstage, kh, kw);
    P.wpk = wpk;
    PlainEpiArgs ea{};
    ea.bias = bias; ea.Co = Co; ea.split = Co; ea.ng = plain_groups(Co);
    ea.out0 = y; ea.bstride0 = (long long)H * W * Co; ea.ld0 = Co;
    VPX_CHECK_HIP(launch_conv_plain_f32(P, ea, n_tiles, stream));
    return VPX_OK;
}

}  // extern "C"
