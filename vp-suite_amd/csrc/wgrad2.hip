// wgrad2.hip — weight gradient of the ConvLSTM block with BOTH operands pre-split (round 2): dW[n][c][tap] =
// sum over (t, b, pixel) of dG[t,b,pixel][n] * [x_t | h_{t-1}][pixel + tap][c]  (autograd of conv_lstm_hzzone.py:59-61).
// Same contraction, operand split and slab / K-slice scheme as wgrad_tg_kernel (lstm_bwd.hip); what changed and why:
//   * dG arrives in split operand format from the gate-backward kernel, the activations from the cell2 forward: every
//     operand byte goes HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4). No split VALU work, no staging registers.
//   * a wave owns 64 gate rows x 32 channels x its tap group (wgrad_tg: 32 x 32): the activation fragments of a tap are
//     read once for two row blocks, 14 fragment reads per 30 MFMAs instead of 12 per 15. The tap-group kernel kept the
//     LDS read pipe ~80 % busy at full matrix rate (PMC: MFMA busy 36 %); here it is 47 %.
//   * workgroup tile 128 rows x 64 channels x 9 taps, items of 4 x 16 pixels (two item buffers of 59 KiB), one barrier per item.
// LDS image of an item: dG planes [row half wn][hi | lo][64 px][64 rows] (128-byte pixel rows, wg_aswz swizzle), then the
// activation halo planes [hi | lo][6 x 18 positions][64 channels]. Fragments are transposing reads (ds_read_b64_tr_b16).
#include <stdlib.h>

#include "vpx_internal.h"

namespace vpx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int W2_TH = 4;                          // tile rows per item
constexpr int W2_NPX = W2_TH * 16;                // 64 pixels = 4 k-steps of 16
constexpr int W2_HALO_W = 18;
constexpr int W2_NPOS = (W2_TH + 2) * W2_HALO_W;  // 108 halo positions
constexpr int W2_GPL = W2_NPX * 128;              // one dG plane: 8 KiB
constexpr int W2_A0 = 4 * W2_GPL;                 // activation planes start here
constexpr int W2_APL = W2_NPOS * 128;             // 13824 B
constexpr int W2_BUF = W2_A0 + 2 * W2_APL;        // 60416 B per item
constexpr int W2_LDS = 2 * W2_BUF;                // 120832 B
constexpr int W2_APIECES = 2 * W2_NPOS * 8;       // 1728 16-byte pieces of the activation planes

__device__ const float w2_zero16[4] __attribute__((aligned(16))) = {0.f, 0.f, 0.f, 0.f};

// 128-byte rows. 32x32x16 form: row bit 1 swaps the 64-byte halves (a 32-lane half reads 4 consecutive rows at two column blocks).
// 16x16x32 form (QF): row bit 2 additionally swaps the 32-byte quarters — there a 32-lane half reads EIGHT consecutive rows at one
// 16-column block (32 bytes), and (row bit 0 -> the 128-byte row parity, bit 1, bit 2) then spread them over the 8 x 8 banks.
template <bool QF>
__device__ __forceinline__ int w2_swz(const int off) { return QF ? (off ^ ((off >> 2) & 0x40) ^ ((off >> 4) & 0x20)) : (off ^ ((off >> 2) & 0x40)); }

template <int STEP2 = 512>
__device__ __forceinline__ bf16x8 w2_frag(const char* base) {
    // two transposing reads: pixels +0..3 and +4..7 of this lane half's 8-pixel group (128 bytes per pixel row); 16x16x32 form:
    // pixels +0..3 and +8..11 (STEP2 = 1024) — any k <-> pixel map works as long as both operands use the same one
    typedef bf16x4 __attribute__((address_space(3))) * lds_v4;
    const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(base));
    const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(base + STEP2));
    return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

__device__ __forceinline__ void w2_dma16(const char* g, char* lds_wave_base) {
    // 64 lanes x 16 B -> lds_wave_base + 16 * lane. Inline asm: see c2_dma16 (cell2.hip) — a compiler-visible LDS-DMA
    // degrades every later s_waitcnt of the kernel to zero.
    const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_wave_base;
    asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(g), "{m0}"(__builtin_amdgcn_readfirstlane(lds)) : "memory");
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef VPX_DEV_SWITCHES
// developer build: per WORKGROUP start / end of the item loop (s_memtime), HW_ID | XCC_ID << 32, and what it worked on (stw_kernel: pass | slice << 8 |
// pair << 24; wgrad2_kernel: 4 | half-tail << 3 | slice << 8 | tile << 24) — how a launch's workgroups were dealt to XCDs and CUs and how long each
// ran (tools/trace_stw.py; vpx_dbg_stw_trace reads it back). Only times of ONE CU may be compared: the counter's base differs between shader engines.
__device__ unsigned long long stw_trace[8192 * 4];
#endif

// QF: the same kernel on v_mfma_f32_16x16x32_bf16 (vpx_set_option(VPX_OPT_MFMA_SHAPE, 1)): a K = 32 step is two tile rows of the
// item, a wave's 64 gate rows x 32 channels x tap group are 4 x 2 accumulator tiles of 16x16 per tap (the same 160 registers),
// the same fragment bytes per MFMA cycle.
// GLUE (round 6, QF only): the weight gradient of a stage-glue layer's stride residue (conv_api.hip, strided_wgrad) on the same machinery —
// kh x kw <= 3 x 3 taps (a 4x4 stride-2 layer's residues have 2 x 2, a 3x3 stride-2 layer's 2 x 2 / 2 x 1 / 1 x 2), the tap (0,0) at
// (org_y, org_x) instead of (-1,-1), and the activation operand a SUB-IMAGE of a larger one: position (y, x) of the walk reads pixel
// (a_sy * y + a_oy, a_sx * x + a_ox) of an image a_Wfull pixels wide. All of it is linear in the position, so a copy is still a per-item
// scalar base + a per-thread offset fixed for the launch. T = 1, x segment only. Row tiles past N4 are skipped (96 rows: 6 of 8 tiles).
template <bool QF, bool GLUE = false>
__global__ __launch_bounds__(512, 2) void wgrad2_kernel(const WgradArgs a) {
    static_assert(QF || !GLUE, "the glue form exists on the 16x16x32 shape only");
    constexpr int TA = 5, TB = 4;   // taps of group 0 / group 1 (GLUE: ceil(ntaps / 2) and the rest, at run time)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, hh = lane >> 5;
    const int tg = wave >> 2, wn = (wave >> 1) & 1, wc = wave & 1;
    // XCD-aware block decode (the rule of wg_block, lstm_bwd.hip): contiguous ranges of the logical sequence per XCD. The
    // sequence lists the FULL column tiles of every slice first (grid_slices slices), then the half-empty last column tile
    // (w2_nh row tiles of it per slice) on FEWER slices (w2_ns_half): its items are cheaper (k-steps split over the wave pair,
    // below), so each of its workgroups takes more of them and every workgroup of the launch runs equally long.
    int bx, slice, ns;
    {
        const int nf = a.grid_x - a.w2_nh;                       // full tiles per slice
        const long long n_full = (long long)nf * a.grid_slices;
        const long long total = n_full + (long long)a.w2_nh * a.w2_ns_half;
        const long long per_xcd = (total + 7) / 8;
        const unsigned L = blockIdx.x;
        const long long v = (long long)(L & 7) * per_xcd + (L >> 3);
        if ((long long)(L >> 3) >= per_xcd || v >= total) return;
        int fi;
        if (v < n_full) { slice = (int)(v / nf); fi = (int)(v - (long long)slice * nf); ns = a.grid_slices; }
        else { const long long u = v - n_full; slice = (int)(u / a.w2_nh); fi = -1 - (int)(u - (long long)slice * a.w2_nh); ns = a.w2_ns_half; }
        const int n_ctf = a.w2_nh ? a.n_ctiles - 1 : a.n_ctiles; // column tiles with both halves in use
        // bx = row tile * n_ctiles + column tile, as before
        bx = fi >= 0 ? (fi / n_ctf) * a.n_ctiles + fi % n_ctf : (-1 - fi) * a.n_ctiles + (a.n_ctiles - 1);
        slice = __builtin_amdgcn_readfirstlane(slice); bx = __builtin_amdgcn_readfirstlane(bx); ns = __builtin_amdgcn_readfirstlane(ns);
    }
#ifdef VPX_DEV_SWITCHES
    if (threadIdx.x == 0 && blockIdx.x < 8192) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        stw_trace[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime();
        stw_trace[blockIdx.x * 4 + 2] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
        stw_trace[blockIdx.x * 4 + 3] = 4ull | (ns != a.grid_slices ? 8ull : 0ull) | ((unsigned long long)slice << 8) | ((unsigned long long)bx << 24);
    }
#endif
    const int n_ct = a.n_ctiles;
    const int ct_id = __builtin_amdgcn_readfirstlane(bx % n_ct);
    const WgradCHalf ch0 = a.ct[ct_id].h[0], ch1 = a.ct[ct_id].h[1];
    const int n0 = __builtin_amdgcn_readfirstlane((bx / n_ct) * 128);
    const int g_kw = GLUE ? a.kw : 3, g_ntaps = GLUE ? a.kh * a.kw : 9, g_ta = GLUE ? (g_ntaps + 1) / 2 : TA;
    const int tap0 = tg ? g_ta : 0;
    const int my_nt = tg ? g_ntaps - g_ta : g_ta;   // taps of this wave's group

    f32x16 acc[QF ? 1 : 2][QF ? 1 : TA];     // 32x32x16 form: [row block of 32][tap]
    f32x4 accq[QF ? 4 : 1][2][QF ? TA : 1];  // 16x16x32 form: [row tile of 16][channel tile of 16][tap]
#pragma unroll
    for (int nb = 0; nb < (QF ? 1 : 2); ++nb)
#pragma unroll
        for (int t = 0; t < (QF ? 1 : TA); ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][t][r] = 0.0f;
#pragma unroll
    for (int rt = 0; rt < (QF ? 4 : 1); ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int t = 0; t < (QF ? TA : 1); ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) accq[rt][ct][t][r] = 0.0f;

    // fragment addressing (transposing reads; same lane map as wgrad_tg_kernel)
    const int L16 = lane & 15, q = L16 >> 2, p = L16 & 3, half16 = (lane >> 4) & 1;
    const int kgq = lane >> 4;   // QF: k group = tile row of the K = 32 pair (kgq >> 1) and which 4 + 4 of its pixels (kgq & 1)
    int g_lane[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) g_lane[nb] = w2_swz<false>(((8 * hh + q) * 64 + nb * 32 + 16 * half16 + 4 * p) * 2);
    // QF: rows 16 * (kgq >> 1) + 4 * (kgq & 1) + q (+ 8 in the second read), row tile rt at columns rt * 16: + (rt * 32) under the XOR
    const int g_laneq = w2_swz<true>(((16 * (kgq >> 1) + 4 * (kgq & 1) + q) * 64 + 4 * p) * 2);
    // A column tile whose second 32-channel half is empty (80 = 16 + 32 + 32 channels -> the last tile holds one half) would leave
    // the wc = 1 waves multiplying zeros: there, both waves of a pair take the SAME channels and split the k-steps of every item
    // between them (wc = 0: rows 0-1, wc = 1: rows 2-3 of the 4 x 16 item); the pair's accumulators meet in LDS after the loop.
    const bool ksplit = ch1.cn == 0;
    const int s_lo = ksplit ? 2 * wc : 0, s_hi = ksplit ? 2 * wc + 2 : W2_TH;
    const int a_lane = QF ? (((kgq >> 1) * W2_HALO_W + 4 * (kgq & 1) + q) * 64 + (ksplit ? 0 : wc) * 32 + 4 * p) * 2
                          : ((8 * hh + q) * 64 + (ksplit ? 0 : wc) * 32 + 16 * half16 + 4 * p) * 2;
    int tapoff[TA];
#pragma unroll
    for (int t = 0; t < TA; ++t) {
        const int tp = tap0 + t;
        const int dy = tp / g_kw, dx = tp - dy * g_kw;
        tapoff[t] = (dy * W2_HALO_W + dx) * 128;
    }

    // ---- item walk: slice, slice + n_slices, ... as a mixed-radix counter (tx, ty, b, t) ----
    const int tiles_x = (a.W + 15) / 16, tiles_y = (a.H + W2_TH - 1) / W2_TH, tiles = tiles_x * tiles_y;
    const int d_tx = __builtin_amdgcn_readfirstlane(ns % tiles_x), d_ty = __builtin_amdgcn_readfirstlane((ns / tiles_x) % tiles_y);
    const int d_b = __builtin_amdgcn_readfirstlane((ns / tiles) % a.B), d_t = __builtin_amdgcn_readfirstlane(ns / (tiles * a.B));
    struct Item { int tx, ty, b, t; };
    auto advance = [&](Item& it) {
        it.tx += d_tx; if (it.tx >= tiles_x) { it.tx -= tiles_x; ++it.ty; }
        it.ty += d_ty; if (it.ty >= tiles_y) { it.ty -= tiles_y; ++it.b; }
        it.b += d_b;   if (it.b >= a.B) { it.b -= a.B; ++it.t; }
        it.t += d_t;
    };
    // a tile whose present halves all read h sees nothing at t = 0 without an initial state: start at the first item with t > 0
    const bool skip_t0 = !a.h0_sp && (ch0.cn == 0 || ch0.seg == 1) && (ch1.cn == 0 || ch1.seg == 1);
    Item cur;
    {
        int w = slice;
        const int first = skip_t0 ? a.B * tiles : 0;
        if (w < first) w += (first - w + ns - 1) / ns * ns;
        const int tile = w % tiles, tb = w / tiles;
        cur.ty = __builtin_amdgcn_readfirstlane(tile / tiles_x);
        cur.tx = __builtin_amdgcn_readfirstlane(tile - cur.ty * tiles_x);
        cur.b = __builtin_amdgcn_readfirstlane(tb % a.B);
        cur.t = __builtin_amdgcn_readfirstlane(tb / a.B);
    }

    // ---- this thread's DMA pieces. dG: one 16-byte slot of each of the four planes (same pixel, same slot); the XOR swizzle of
    //      the image is applied through the CHOICE of source (physical slot -> logical slot, the swizzle is an involution) ----
    const int g_px = tid >> 3;
    const int g_sl = (w2_swz<QF>(g_px * 128 + (tid & 7) * 16) & 127) >> 4;     // logical slot: rows g_sl*8 .. +7 of the plane's 64
    const int g_off = ((n0 + g_sl * 8) >> 3) * 32;                         // byte offset inside the dG pixel row (plane u adds (u>>1)*256 + (u&1)*16)
    const bool g_ok0 = n0 + g_sl * 8 < a.N4, g_ok1 = n0 + 64 + g_sl * 8 < a.N4;
    const unsigned g_prow = (unsigned)a.N4 * 4u;
    // Addresses = a per-item SCALAR base (image of (t, b) + the tile's origin) + a per-thread 32-bit offset fixed for the whole
    // launch: a copy costs two adds and a select per piece (the first version redid a 64-bit multiply chain per piece and item,
    // ~250 vector instructions per item and wave next to its 120 MFMAs).
    const unsigned g_toff = (unsigned)((g_px >> 4) * a.W + (g_px & 15)) * g_prow + (unsigned)g_off;
    // activations: pieces tid + 512 u of [hi plane | lo plane], 8 pieces per halo position
    int pc_hyx[4];        // (row << 16) | column of the halo position, -1: none
    unsigned pc_toff[4];  // (hy * W + hx) * bytes-per-pixel of its half + the byte offset of its 8 channels (hi or lo)
    bool pc_h1[4];        // the piece belongs to the tile's second half
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int piece = tid + 512 * u;
        const int plane = piece >= W2_NPOS * 8 ? 1 : 0;
        const int qq = piece - plane * W2_NPOS * 8;
        const int pos = qq >> 3;
        const int sl = (w2_swz<QF>(pos * 128 + (qq & 7) * 16) & 127) >> 4;     // logical slot = 8 channels of the 64-channel row
        const int hy = pos / W2_HALO_W, hx = pos - hy * W2_HALO_W;
        const WgradCHalf hf = (sl >> 2) ? ch1 : ch0;
        const bool ok = piece < W2_APIECES && (sl & 3) * 8 < hf.cn && (!GLUE || (hy < W2_TH + a.kh - 1 && hx < 16 + a.kw - 1));
        const unsigned prow_h = (unsigned)(hf.seg == 0 ? a.Cin : a.Ch) * 4u;
        pc_hyx[u] = ok ? ((hy << 16) | hx) : -1;
        pc_toff[u] = (GLUE ? (unsigned)(hy * a.a_sy * a.a_Wfull + hx * a.a_sx) : (unsigned)(hy * a.W + hx)) * prow_h +
                     (unsigned)(((hf.c0 + (sl & 3) * 8) >> 3) * 32 + plane * 16);
        pc_h1[u] = (sl >> 2) != 0;
    }
    const unsigned prow0 = (unsigned)(ch0.seg == 0 ? a.Cin : a.Ch) * 4u, prow1 = (unsigned)(ch1.seg == 0 ? a.Cin : a.Ch) * 4u;
    auto dma_item = [&](const Item& it, char* buf) {
        const int y0 = it.ty * W2_TH, x0 = it.tx * 16;
        {
            const bool pix_ok = (y0 + (g_px >> 4) < a.H) & (x0 + (g_px & 15) < a.W);   // (bitwise on purpose, here and below: selects, not divergent branches)
            const char* const gbase = a.g_sp + (((size_t)it.t * a.B + it.b) * a.HW + (size_t)(y0 * a.W + x0)) * g_prow;   // scalar
            const char* row = gbase + g_toff;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool ok = pix_ok & ((u >> 1) ? g_ok1 : g_ok0);
                const char* src = ok ? row + (u >> 1) * 256 + (u & 1) * 16 : reinterpret_cast<const char*>(w2_zero16);
                w2_dma16(src, buf + u * W2_GPL + wave * 1024);
            }
        }
        // per half (scalar): the (t, b) image of its split tensor shifted to the halo origin (y0 - 1, x0 - 1) — only dereferenced
        // for positions inside the image; null = the half stages zeros (unused half, or no hidden state at t = 0)
        const char* base[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const WgradCHalf hf = h ? ch1 : ch0;
            base[h] = nullptr;
            if (hf.cn == 0) continue;
            if (hf.seg == 0) base[h] = a.x_sp + (size_t)it.b * a.x_sp_bstride + (size_t)it.t * a.x_sp_tstride;
            else if (it.t > 0) base[h] = a.h_sp + (size_t)(it.t - 1) * a.h_sp_tstride + (size_t)it.b * a.h_sp_bstride;
            else if (a.h0_sp) base[h] = a.h0_sp + (size_t)it.b * a.HW * a.Ch * 4;
        }
        // halo origin in the operand's own grid (GLUE: the sub-image's), and its pixel index in memory
        const int oy = GLUE ? y0 + a.org_y : y0 - 1, ox = GLUE ? x0 + a.org_x : x0 - 1;
        const int lim_y = GLUE ? a.a_Hs : a.H, lim_x = GLUE ? a.a_Ws : a.W;
        const long long org = GLUE ? (long long)(oy * a.a_sy + a.a_oy) * a.a_Wfull + (ox * a.a_sx + a.a_ox) : (long long)oy * a.W + ox;
        const bool ok0 = base[0] != nullptr, ok1 = base[1] != nullptr;
        const char* const ab0 = ok0 ? base[0] + org * (long long)prow0 : nullptr;
        const char* const ab1 = ok1 ? base[1] + org * (long long)prow1 : nullptr;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int gy = oy + (pc_hyx[u] >> 16), gx = ox + (pc_hyx[u] & 0xffff);
            const bool ok = (pc_hyx[u] >= 0) & ((unsigned)gy < (unsigned)lim_y) & ((unsigned)gx < (unsigned)lim_x) & (pc_h1[u] ? ok1 : ok0);
            const char* src = ok ? (pc_h1[u] ? ab1 : ab0) + pc_toff[u] : reinterpret_cast<const char*>(w2_zero16);
            if (512 * u + wave * 64 < W2_APIECES)   // (wave-uniform: 1728 = 27 waves' worth of pieces)
                w2_dma16(src, buf + W2_A0 + (512 * u + wave * 64) * 16);
        }
    };

    // ---- one item: 4 k-steps of 16 pixels; per k-step the dG fragments of both row blocks, then the taps in batches ----
    auto tap_batch = [&](const char* buf, const bf16x8 (&gh)[2], const bf16x8 (&gl)[2], int arow, auto t0_c, auto nb_c) {
        constexpr int T0 = decltype(t0_c)::value, NB = decltype(nb_c)::value;
        bf16x8 ah[NB], al[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int aoff = w2_swz<false>(arow + tapoff[T0 + j]);
            ah[j] = w2_frag(buf + W2_A0 + aoff);
            al[j] = w2_frag(buf + W2_A0 + W2_APL + aoff);
        }
        // the three terms term-major: consecutive MFMAs accumulate into different tiles
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[nb][T0 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gl[nb], ah[j], acc[nb][T0 + j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[nb][T0 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh[nb], al[j], acc[nb][T0 + j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[nb][T0 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh[nb], ah[j], acc[nb][T0 + j], 0, 0, 0);
    };
    // GLUE: 16-row tiles of this wave's 64 rows that lie inside the layer's rows
    const int rt_lim = GLUE ? (a.N4 - n0 - wn * 64 + 15) / 16 : 4;
    auto multiply = [&](const char* buf) {
        const char* gb = buf + wn * 2 * W2_GPL;
        if constexpr (QF) {
            // K = 32 steps: tile rows (0,1) and (2,3) of the item; ksplit: one step per wave of the pair
#pragma unroll 1
            for (int s2 = (ksplit ? wc : 0); s2 < (ksplit ? wc + 1 : 2); ++s2) {
                bf16x8 gh[4], gl[4];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    gh[rt] = w2_frag<1024>(gb + (g_laneq ^ (rt * 32)) + s2 * 4096);
                    gl[rt] = w2_frag<1024>(gb + W2_GPL + (g_laneq ^ (rt * 32)) + s2 * 4096);
                }
                const int arow = a_lane + s2 * 2 * W2_HALO_W * 128;
#pragma unroll
                for (int t = 0; t < TA; ++t) {
                    if (GLUE ? t >= my_nt : (t == TA - 1 && tg != 0)) break;
                    bf16x8 ah[2], al[2];
                    const int aoff = w2_swz<true>(arow + tapoff[t]);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        ah[ct] = w2_frag<1024>(buf + W2_A0 + (aoff ^ (ct * 32)));
                        al[ct] = w2_frag<1024>(buf + W2_A0 + W2_APL + (aoff ^ (ct * 32)));
                    }
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) {
                        if (GLUE && rt >= rt_lim) break;   // (wave-uniform: row tiles past the layer's rows hold zeros)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) accq[rt][ct][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gl[rt], ah[ct], accq[rt][ct][t], 0, 0, 0);
                    }
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) {
                        if (GLUE && rt >= rt_lim) break;
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) accq[rt][ct][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh[rt], al[ct], accq[rt][ct][t], 0, 0, 0);
                    }
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) {
                        if (GLUE && rt >= rt_lim) break;
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) accq[rt][ct][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh[rt], ah[ct], accq[rt][ct][t], 0, 0, 0);
                    }
                }
            }
        } else {
#pragma unroll 1
        for (int s = s_lo; s < s_hi; ++s) {
            bf16x8 gh[2], gl[2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                gh[nb] = w2_frag(gb + g_lane[nb] + s * 2048);
                gl[nb] = w2_frag(gb + W2_GPL + g_lane[nb] + s * 2048);
            }
            const int arow = a_lane + s * W2_HALO_W * 128;
            tap_batch(buf, gh, gl, arow, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
            tap_batch(buf, gh, gl, arow, std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
            if (tg == 0) tap_batch(buf, gh, gl, arow, std::integral_constant<int, 4>{}, std::integral_constant<int, 1>{});
        }
        }
    };

    Item nxt = cur;
    advance(nxt);
    if (cur.t < a.T) dma_item(cur, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int bsel = 0;
    while (cur.t < a.T) {
        char* bcur = smem + bsel * W2_BUF;
        char* bnxt = smem + (bsel ^ 1) * W2_BUF;
        // the other buffer was multiplied in the previous iteration and every wave has passed that iteration's barrier:
        // item i+1 is copied over it while item i is multiplied (a whole item of MFMA time to land)
        if (nxt.t < a.T) dma_item(nxt, bnxt);
        multiply(bcur);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        cur = nxt; advance(nxt); bsel ^= 1;
    }
#ifdef VPX_DEV_SWITCHES
    if (threadIdx.x == 0 && blockIdx.x < 8192) stw_trace[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime();
#endif

    if (ksplit) {
        // pair reduction (once per workgroup): the wc = 1 wave of each (tap group, row half) pair parks its accumulators in LDS,
        // the wc = 0 wave adds them; one row block (5 tiles x 16 registers x 64 lanes = 20 KiB per pair) at a time
        float* red = reinterpret_cast<float*>(smem) + (wave >> 1) * (TA * 16 * 64);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            __syncthreads();
            if (wc == 1) {
                if constexpr (QF) {
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                            for (int t = 0; t < TA; ++t)
#pragma unroll
                                for (int r = 0; r < 4; ++r) red[(((rr * 2 + ct) * TA + t) * 4 + r) * 64 + lane] = accq[2 * nb + rr][ct][t][r];
                } else {
#pragma unroll
                for (int t = 0; t < TA; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[(t * 16 + r) * 64 + lane] = acc[nb][t][r];
                }
            }
            __syncthreads();
            if (wc == 0) {
                if constexpr (QF) {
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                            for (int t = 0; t < TA; ++t)
#pragma unroll
                                for (int r = 0; r < 4; ++r) accq[2 * nb + rr][ct][t][r] += red[(((rr * 2 + ct) * TA + t) * 4 + r) * 64 + lane];
                } else {
#pragma unroll
                for (int t = 0; t < TA; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[nb][t][r] += red[(t * 16 + r) * 64 + lane];
                }
            }
        }
    }
    const WgradCHalf oh = wc ? ch1 : ch0;
    float* slab = a.slabs + (size_t)slice * g_ntaps * a.N4 * a.Ct;
    if constexpr (QF) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int cw = ct * 16 + (lane & 15);
                const int col = oh.cglobal + cw;
                const bool col_ok = cw < oh.cn;
#pragma unroll
                for (int t = 0; t < TA; ++t) {
                    if (GLUE ? t >= my_nt : (tg == 1 && t >= TB)) break;
                    float* st = slab + (size_t)(tap0 + t) * a.N4 * a.Ct;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = n0 + wn * 64 + rt * 16 + 4 * (lane >> 4) + r;
                        if (n < a.N4 && col_ok) st[(size_t)n * a.Ct + col] = accq[rt][ct][t][r];
                    }
                }
            }
    } else {
    const int col = oh.cglobal + i;
    const bool col_ok = i < oh.cn;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int t = 0; t < TA; ++t) {
            if (tg == 1 && t >= TB) break;
            float* st = slab + (size_t)(tap0 + t) * a.N4 * a.Ct;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 64 + nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (n < a.N4 && col_ok) st[(size_t)n * a.Ct + col] = acc[nb][t][r];
            }
        }
    }
}

// Applies when launch_wgrad would pick the pre-split tap-group kernel AND dG is available in split format: 3x3, bf16x3.
bool wgrad2_applicable(const WgradArgs& a) {
    static int env = -1;   // VPX_WGRAD2=0: keep wgrad_tg_kernel (experiments)
    if (env < 0) env = dev_switch("VPX_WGRAD2", 1);
    if (!env || !a.a_split || !a.g_sp || a.kh != 3 || a.kw != 3 || a.prec != VPX_PREC_BF16X3) return false;
    if (a.a_sub || a.use_org || a.blk || (a.n_out && a.n_out != a.N4) || (a.N4 & 7) || (a.Cin & 7) || (a.Ch & 7)) return false;
    const long long items = (long long)a.T * a.B * ((a.W + 15) / 16) * ((a.H + W2_TH - 1) / W2_TH);
    return items + 4096 < (1ll << 31);
}

int wgrad2_target_wgs() {
    static int target = -1;
    if (target < 0) target = dev_switch("VPX_WGRAD2_WGS", 512);
    return target;
}

// slices: whole rounds of 256 one-per-CU workgroups (two rounds by default, VPX_WGRAD2_WGS overrides), at most max_slices
hipError_t launch_wgrad2(const WgradArgs& a_in, int max_slices, int* used_slices, int* tail_col0, int* tail_slices, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = vpx_func_attr(reinterpret_cast<const void*>(&wgrad2_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS);
        if (e != hipSuccess) return e;
        e = vpx_func_attr(reinterpret_cast<const void*>(&wgrad2_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS);
        if (e != hipSuccess) return e;
        attr_set = !g_dry_run;
    }
    const int target = wgrad2_target_wgs();
    WgradArgs a = a_in;
    const int rows = (a.N4 + 127) / 128;
    a.grid_x = rows * a.n_ctiles;
    const bool half_tail = a.ct[a.n_ctiles - 1].h[1].cn == 0;
    a.w2_nh = half_tail ? rows : 0;
    const int nf = a.grid_x - a.w2_nh;
    // Full tiles take ns slices, the half-empty tail tiles 5/8 ns: an item of a tail tile costs 0.62 of a full one (measured: the
    // MFMA work halves, the copy and the barrier do not), so with 8/5 of the items its workgroups run as long as the others —
    // ns * nf + 5/8 ns * nh workgroups of equal length, at most `target`.
    int ns = wgrad2_slices(target, rows, a.n_ctiles, half_tail);
    if (ns > max_slices) ns = max_slices;
    const long long items = (long long)a.T * a.B * ((a.W + 15) / 16) * ((a.H + W2_TH - 1) / W2_TH);
    if (ns > items) ns = (int)items;
    if (half_tail && ns >= 8) ns &= ~7;
    if (ns < 1) ns = 1;
    a.grid_slices = ns;
    a.w2_ns_half = half_tail ? (ns >= 8 ? ns / 8 * 5 : ns) : 0;
    *used_slices = ns;
    *tail_col0 = half_tail ? a.ct[a.n_ctiles - 1].h[0].cglobal : a.Ct;
    *tail_slices = half_tail ? a.w2_ns_half : ns;
    const long long total = (long long)nf * ns + (long long)a.w2_nh * a.w2_ns_half;
    if (!ws_write_ok(a.slabs, (size_t)ns * 9 * a.N4 * a.Ct * sizeof(float), "weight-gradient slabs (wgrad2_kernel)")) return hipErrorInvalidValue;
    if (mfma_shape() == 1) VPX_LAUNCH(wgrad2_kernel<true>, dim3((unsigned)(8 * ((total + 7) / 8))), dim3(512), W2_LDS, s, a);
    else VPX_LAUNCH(wgrad2_kernel<false>, dim3((unsigned)(8 * ((total + 7) / 8))), dim3(512), W2_LDS, s, a);
    return vpx_hip_last_error();
}

// ---- glue form (wgrad2_kernel<true, true>): one stride residue of a stage-glue layer's weight gradient, both operands pre-split ----
bool wgrad2g_applicable(const WgradArgs& a) {
    if (g_experiment & (1 << 29)) return false;   // VPX_OPT_EXPERIMENT bit 29: the tap-group kernel on fp32 operands (A/B runs, tests)
    if (mfma_shape() != 1 || a.prec != VPX_PREC_BF16X3 || !a.g_sp || !a.x_sp || a.T != 1) return false;
    const int taps = a.kh * a.kw;
    if (taps < 2 || a.kh > 3 || a.kw > 3) return false;
    if (a.blk || (a.n_out && a.n_out != a.N4) || (a.N4 & 7) || (a.Cin & 7) || a.Ct != a.Cin) return false;
    for (int c = 0; c < a.n_ctiles; ++c)
        for (int h = 0; h < 2; ++h) if (a.ct[c].h[h].cn && a.ct[c].h[h].seg != 0) return false;
    // 32-bit per-thread offsets: a halo position's byte offset inside the operand image
    const long long span = ((long long)(W2_TH + 2) * (a.a_sub ? a.a_sy : 1) * (a.a_sub ? a.a_Wfull : a.W) + 18ll * (a.a_sub ? a.a_sx : 1)) * a.Cin * 4;
    if (span >= (1ll << 31) || (long long)a.HW * a.N4 * 4 >= (1ll << 31)) return false;
    const long long items = (long long)a.B * ((a.W + 15) / 16) * ((a.H + W2_TH - 1) / W2_TH);
    return items + 4096 < (1ll << 31);
}

// slabs [used_slices][kh * kw][N4][Ct]; every column tile on the same slices (the caller's reduce maps taps, it has no tail form)
hipError_t launch_wgrad2g(const WgradArgs& a_in, int max_slices, int* used_slices, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = vpx_func_attr(reinterpret_cast<const void*>(&wgrad2_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS);
        if (e != hipSuccess) return e;
        attr_set = !g_dry_run;
    }
    WgradArgs a = a_in;
    if (!a.a_sub) {   // plain stride-1 layer: the "sub-image" is the image
        a.a_sy = a.a_sx = 1; a.a_oy = a.a_ox = 0; a.a_Hs = a.H; a.a_Ws = a.W; a.a_Wfull = a.W;
        if (!a.use_org) { a.org_y = -(a.kh / 2); a.org_x = -(a.kw / 2); }
    }
    const int rows = (a.N4 + 127) / 128;
    a.grid_x = rows * a.n_ctiles;
    const bool half_tail = a.ct[a.n_ctiles - 1].h[1].cn == 0;
    a.w2_nh = half_tail ? rows : 0;
    int ns = wgrad2_target_wgs() / a.grid_x;
    if (ns > max_slices) ns = max_slices;
    const long long items = (long long)a.B * ((a.W + 15) / 16) * ((a.H + W2_TH - 1) / W2_TH);
    if (ns > items) ns = (int)items;
    if (ns < 1) ns = 1;
    a.grid_slices = ns;
    a.w2_ns_half = half_tail ? ns : 0;
    *used_slices = ns;
    const long long total = (long long)a.grid_x * ns;
    if (!ws_write_ok(a.slabs, (size_t)ns * a.kh * a.kw * a.N4 * a.Ct * sizeof(float), "weight-gradient slabs (wgrad2_kernel, glue form)")) return hipErrorInvalidValue;
    VPX_LAUNCH((wgrad2_kernel<true, true>), dim3((unsigned)(8 * ((total + 7) / 8))), dim3(512), W2_LDS, s, a);
    return vpx_hip_last_error();
}


// =====================================================================================================================
// stw: the four 5x5 weight gradients of ONE ST-LSTM cell step (autograd of predrnn.py:57-83: conv_x, conv_h, conv_m, conv_o)
// in ONE launch on the wgrad2 machinery (round 4). dW*[n][c][tap] = sum over (b, pixel) dG7[b,pixel][n] * src[pixel + tap][c]
// with dG7 [B,HW,7Ch] = d(pre-activations) ordered (i,f,g | o | i',f',g') and the sources x, h, m, c_new, m_new — all six
// tensors in the split operand format, staged by LDS-DMA. What differs from wgrad2_kernel<true>:
//   * 5x5: the halo tile of a 4 x 16-pixel item is 8 x 20 positions (two item buffers of 72 KiB); the 25 taps are three
//     PASSES of a workgroup's (128 rows x 64 channels) tile — tap rows {0,1}, {2,3}, {4}: a wave's tap group is one tap row
//     (5 accumulator tiles = the same 160 registers). In pass 2 both tap groups take tap row 4 and split the K = 32 steps
//     of every item between them; the pair's accumulators meet in LDS after the loop (fixed order).
//   * the work list is a table of PAIRS (128-row tile of dG7, 64-column tile of one source) instead of a dense row x column
//     grid: x pairs with all rows, h with the first 4Ch, m with the last 3Ch, [c_new | m_new] with the o block — one launch
//     of npairs x 3 passes x K slices workgroups of equal shape instead of eight launches of the first-generation kernel
//     (which re-staged fp32 operands through registers, one item buffer, taps 9 + 9 + 7).
//   * every pair's slab block is [tap][128][64]; stw_reduce_kernel sums the K slices and scatters into the four OIHW tensors
//     (the row-block permutation of Wx is a per-tensor block map there).
// =====================================================================================================================
constexpr int W5_HALO_W = 20;
constexpr int W5_NPOS = (W2_TH + 4) * W5_HALO_W;  // 160 halo positions
constexpr int W5_APL = W5_NPOS * 128;             // 20480 B per plane
constexpr int W5_BUF = W2_A0 + 2 * W5_APL;        // 73728 B per item
constexpr int W5_LDS = 2 * W5_BUF;                // 147456 B
constexpr int W5_APIECES = 2 * W5_NPOS * 8;       // 2560 pieces = 5 per thread exactly
constexpr int W5_BLOCK = 25 * 128 * 64;           // floats of a pair's slab block

__global__ __launch_bounds__(512, 2) void stw_kernel(const STWArgs a) {
    constexpr int TA = 5;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tg = wave >> 2, wn = (wave >> 1) & 1, wc = wave & 1;
    int slice, pair_i, pass;
    {
        const int per_slice = a.npairs5 * 3 + (a.npairs - a.npairs5);
        const long long total = (long long)per_slice * a.n_slices;
        const long long per_xcd = (total + 7) / 8;
        const unsigned L = blockIdx.x;
        if ((a.n_slices & 7) == 0) {
            // Whole slices per XCD (round 5): XCD x takes slices x, x + 8, ... — every XCD gets the same mix of long (tap rows {0,1}, {2,3}),
            // short (tap row 4: half the work) and centre-tap workgroups, and all its workgroups walk the same pixels (one L2). In dispatch
            // order: the long ones of all its slices, then the short ones, then the 1x1 tensor's. The contiguous-range rule below handed
            // XCD 0 long workgroups only and its neighbour mostly short ones: the launch took two long rounds at 70 % of the CUs' time.
            const int i = (int)(L >> 3), spx = a.n_slices >> 3;
            const int nl = 2 * a.npairs5, nsh = a.npairs5, nc = a.npairs - a.npairs5;
            if (i >= spx * per_slice) return;
            int sl;
            if (i < spx * nl) { sl = i / nl; const int r = i - sl * nl; pair_i = r >> 1; pass = r & 1; }
            else if (i < spx * (nl + nsh)) { const int j = i - spx * nl; sl = j / nsh; pair_i = j - sl * nsh; pass = 2; }
            else { const int j = i - spx * (nl + nsh); sl = j / nc; pair_i = a.npairs5 + (j - sl * nc); pass = 3; }
            slice = (int)(L & 7) + 8 * sl;
        } else {
        const long long v = (long long)(L & 7) * per_xcd + (L >> 3);
        if ((long long)(L >> 3) >= per_xcd || v >= total) return;
        slice = (int)(v / per_slice);
        const int rem = (int)(v - (long long)slice * per_slice);
        // the long passes (tap rows {0,1}, {2,3}) of every pair first, the short pass last
        // (pairs [npairs5, npairs) are the 1x1 tensor's: one pass, centre tap only)
        if (rem < 2 * a.npairs5) { pair_i = rem >> 1; pass = rem & 1; }
        else if (rem < 3 * a.npairs5) { pair_i = rem - 2 * a.npairs5; pass = 2; }
        else { pair_i = rem - 2 * a.npairs5; pass = 3; }
        }
        slice = __builtin_amdgcn_readfirstlane(slice); pair_i = __builtin_amdgcn_readfirstlane(pair_i); pass = __builtin_amdgcn_readfirstlane(pass);
    }
#ifdef VPX_DEV_SWITCHES
    if (threadIdx.x == 0 && blockIdx.x < 8192) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        stw_trace[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime();
        stw_trace[blockIdx.x * 4 + 2] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
        stw_trace[blockIdx.x * 4 + 3] = (unsigned long long)pass | ((unsigned long long)slice << 8) | ((unsigned long long)pair_i << 24);
    }
#endif
    const STWPair pr = a.pair[pair_i];
    const STWHalf ch0 = pr.h[0], ch1 = pr.h[1];
    const int n0 = pr.n0;
    const bool ksplit = ch1.cn == 0;                 // half-empty column tile: the wave pair (wc) shares the channels and splits K
    const bool centre = pass == 3;                   // conv_last (1x1): tap (2, 2) only
    const bool tsplit = pass >= 2 && !ksplit;        // tap row 4 (or the centre tap): the tap groups share it and split K
    const int trow = centre ? 2 : (tsplit ? 4 : 2 * pass + tg);   // this wave's tap row
    const bool active = centre ? (tsplit || tg == 0) : trow < 5;

    f32x4 accq[4][2][TA];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int t = 0; t < TA; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) accq[rt][ct][t][r] = 0.0f;

    const int L16 = lane & 15, q = L16 >> 2, p = L16 & 3, kgq = lane >> 4;
    const int g_laneq = w2_swz<true>(((16 * (kgq >> 1) + 4 * (kgq & 1) + q) * 64 + 4 * p) * 2);
    const int a_lane = (((kgq >> 1) * W5_HALO_W + 4 * (kgq & 1) + q) * 64 + (ksplit ? 0 : wc) * 32 + 4 * p) * 2;
    const int taprow_off = trow * W5_HALO_W * 128;

    const int tiles_x = (a.W + 15) / 16, tiles_y = (a.H + W2_TH - 1) / W2_TH, tiles = tiles_x * tiles_y;
    const int n_items = a.B * tiles;

    // ---- DMA pieces (as wgrad2_kernel): dG7 one slot of each of the four planes; activations five pieces per thread ----
    const int g_px = tid >> 3;
    const int g_sl = (w2_swz<true>(g_px * 128 + (tid & 7) * 16) & 127) >> 4;
    const bool g_ok0 = n0 + g_sl * 8 < a.N7, g_ok1 = n0 + 64 + g_sl * 8 < a.N7;
    const unsigned g_prow = (unsigned)a.N7 * 4u;
    const unsigned g_toff = (unsigned)((g_px >> 4) * a.W + (g_px & 15)) * g_prow + (unsigned)(n0 + g_sl * 8) * 4u;
    const unsigned prow0 = (unsigned)a.src[ch0.src].C * 4u, prow1 = (unsigned)a.src[ch1.src].C * 4u;
    int pc_hyx[5];
    unsigned pc_toff[5];
    bool pc_h1[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const int piece = tid + 512 * u;
        const int plane = piece >= W5_NPOS * 8 ? 1 : 0;
        const int qq = piece - plane * W5_NPOS * 8;
        const int pos = qq >> 3;
        const int sl = (w2_swz<true>(pos * 128 + (qq & 7) * 16) & 127) >> 4;
        const int hy = pos / W5_HALO_W, hx = pos - hy * W5_HALO_W;
        const bool h1 = (sl >> 2) != 0;
        const STWHalf hf = h1 ? ch1 : ch0;
        const bool ok = (sl & 3) * 8 < hf.cn;
        pc_hyx[u] = ok ? ((hy << 16) | hx) : -1;
        pc_toff[u] = (unsigned)(hy * a.W + hx) * (h1 ? prow1 : prow0) + (unsigned)(hf.c0 + (sl & 3) * 8) * 4u + (unsigned)plane * 16u;
        pc_h1[u] = h1;
    }
    // An item = (sample b, tile row ty, tile column tx); a workgroup's items lie n_slices apart: the next one is found by carrying, not by
    // dividing (two scalar divisions per item were a fifth of the 250 instructions a wave spent requesting an item's copies — as long as
    // a quarter of its products, in-kernel stamps), and the two sources' base addresses stay in vector registers (the kernel is at the
    // SGPR limit: as scalars they were reloaded from the kernel arguments for every item, s_waitcnt lgkmcnt(0) included).
    struct SItem { int w, b, ty, tx; };
    const int it_db = a.n_slices / tiles, it_dty = (a.n_slices % tiles) / tiles_x, it_dtx = (a.n_slices % tiles) % tiles_x;
    auto advance = [&](SItem& it) {
        it.w += a.n_slices;
        it.tx += it_dtx; if (it.tx >= tiles_x) { it.tx -= tiles_x; ++it.ty; }
        it.ty += it_dty; if (it.ty >= tiles_y) { it.ty -= tiles_y; ++it.b; }
        it.b += it_db;
    };
    auto keep_in_vgprs = [](const char* p) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(p);
        unsigned lo, hi;
        asm volatile("v_mov_b32 %0, %1" : "=v"(lo) : "v"((unsigned)v));
        asm volatile("v_mov_b32 %0, %1" : "=v"(hi) : "v"((unsigned)(v >> 32)));
        return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
    };
    const char* const sp0 = keep_in_vgprs(a.src[ch0.src].sp);
    const char* const sp1 = keep_in_vgprs(a.src[ch1.src].sp);
    auto dma_item = [&](const SItem& it, char* buf) {
        const int b = it.b;
        const int y0 = it.ty * W2_TH, x0 = it.tx * 16;
        {
            const bool pix_ok = (y0 + (g_px >> 4) < a.H) & (x0 + (g_px & 15) < a.W);
            const char* const gbase = a.g_sp + ((size_t)b * a.HW + (size_t)(y0 * a.W + x0)) * g_prow;
            const char* row = gbase + g_toff;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool ok = pix_ok & ((u >> 1) ? g_ok1 : g_ok0);
                const char* src = ok ? row + (u >> 1) * 256 + (u & 1) * 16 : reinterpret_cast<const char*>(w2_zero16);
                w2_dma16(src, buf + u * W2_GPL + wave * 1024);
            }
        }
        const long long org = (long long)b * a.HW + (long long)(y0 - 2) * a.W + (x0 - 2);   // pixels from the tensor's first to the halo origin
        const char* const ab0 = sp0 + org * (long long)prow0;
        const char* const ab1 = sp1 + org * (long long)prow1;
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int gy = y0 - 2 + (pc_hyx[u] >> 16), gx = x0 - 2 + (pc_hyx[u] & 0xffff);
            const bool ok = (pc_hyx[u] >= 0) & ((unsigned)gy < (unsigned)a.H) & ((unsigned)gx < (unsigned)a.W);
            const char* src = ok ? (pc_h1[u] ? ab1 : ab0) + pc_toff[u] : reinterpret_cast<const char*>(w2_zero16);
            w2_dma16(src, buf + W2_A0 + (512 * u + wave * 64) * 16);
        }
    };

    const int s_lo = ksplit ? wc : (tsplit ? tg : 0), s_hi = (ksplit || tsplit) ? s_lo + 1 : 2;
    auto multiply = [&](const char* buf) {
        const char* gb = buf + wn * 2 * W2_GPL;
#pragma unroll 1
        for (int s2 = s_lo; s2 < s_hi; ++s2) {
            bf16x8 gh[4], gl[4];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                gh[rt] = w2_frag<1024>(gb + (g_laneq ^ (rt * 32)) + s2 * 4096);
                gl[rt] = w2_frag<1024>(gb + W2_GPL + (g_laneq ^ (rt * 32)) + s2 * 4096);
            }
            const int arow = a_lane + s2 * 2 * W5_HALO_W * 128 + taprow_off;
#pragma unroll
            for (int t = 0; t < TA; ++t) {
                if (centre && t != 2) continue;
                bf16x8 ah[2], al[2];
                const int aoff = w2_swz<true>(arow + t * 128);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    ah[ct] = w2_frag<1024>(buf + W2_A0 + (aoff ^ (ct * 32)));
                    al[ct] = w2_frag<1024>(buf + W2_A0 + W5_APL + (aoff ^ (ct * 32)));
                }
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) accq[rt][ct][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gl[rt], ah[ct], accq[rt][ct][t], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) accq[rt][ct][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh[rt], al[ct], accq[rt][ct][t], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) accq[rt][ct][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh[rt], ah[ct], accq[rt][ct][t], 0, 0, 0);
            }
        }
    };

    SItem cur;
    {
        cur.w = slice; cur.b = slice / tiles;
        const int tile = slice - cur.b * tiles;
        cur.ty = tile / tiles_x; cur.tx = tile - cur.ty * tiles_x;
        cur.b = __builtin_amdgcn_readfirstlane(cur.b); cur.ty = __builtin_amdgcn_readfirstlane(cur.ty); cur.tx = __builtin_amdgcn_readfirstlane(cur.tx);
    }
    SItem nxt = cur;
    advance(nxt);
    if (cur.w < n_items) dma_item(cur, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int bsel = 0;
#ifdef VPX_DEV_SWITCHES
    // developer timing stamps: shader cycles of one workgroup's waves spent requesting copies / multiplying / at the item's sync point
    const bool stamp = a.stamps != nullptr && (int)blockIdx.x == a.stamp_block;
    unsigned long long t_dma = 0, t_mul = 0, t_sync = 0, n_it = 0;
#endif
    while (cur.w < n_items) {
        char* bcur = smem + bsel * W5_BUF;
        char* bnxt = smem + (bsel ^ 1) * W5_BUF;
#ifdef VPX_DEV_SWITCHES
        const unsigned long long t0 = stamp ? __builtin_amdgcn_s_memtime() : 0;
#endif
        if (nxt.w < n_items) dma_item(nxt, bnxt);
#ifdef VPX_DEV_SWITCHES
        const unsigned long long t1 = stamp ? __builtin_amdgcn_s_memtime() : 0;
#endif
        if (active) multiply(bcur);
#ifdef VPX_DEV_SWITCHES
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t2 = stamp ? __builtin_amdgcn_s_memtime() : 0;
#endif
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef VPX_DEV_SWITCHES
        if (stamp) { const unsigned long long t3 = __builtin_amdgcn_s_memtime(); t_dma += t1 - t0; t_mul += t2 - t1; t_sync += t3 - t2; ++n_it; }
#endif
        cur = nxt; advance(nxt); bsel ^= 1;
    }
#ifdef VPX_DEV_SWITCHES
    if (threadIdx.x == 0 && blockIdx.x < 8192) stw_trace[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime();
    if (stamp && lane == 0) {
        a.stamps[wave * 8 + 0] = t_dma; a.stamps[wave * 8 + 1] = t_mul; a.stamps[wave * 8 + 2] = t_sync; a.stamps[wave * 8 + 3] = n_it;
        a.stamps[wave * 8 + 4] = (unsigned long long)pass; a.stamps[wave * 8 + 5] = (unsigned long long)pair_i;
    }
#endif

    if (ksplit || tsplit) {
        // pair reduction: the second wave of each pair parks its accumulators in LDS, the first adds them (two row tiles at a time:
        // 4 pairs x 20 KiB). ksplit: pairs (wave, wave ^ 1); tsplit: pairs (wave, wave ^ 4).
        const bool parks = ksplit ? wc == 1 : tg == 1;
        float* red = reinterpret_cast<float*>(smem) + (ksplit ? (wave >> 1) : (wave & 3)) * (TA * 16 * 64);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            __syncthreads();
            if (parks && active) {
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int t = 0; t < TA; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) red[(((rr * 2 + ct) * TA + t) * 4 + r) * 64 + lane] = accq[2 * nb + rr][ct][t][r];
            }
            __syncthreads();
            if (!parks && active) {
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int t = 0; t < TA; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) accq[2 * nb + rr][ct][t][r] += red[(((rr * 2 + ct) * TA + t) * 4 + r) * 64 + lane];
            }
        }
        if (parks) return;
    }
    if (!active) return;
    // slab block of the pair: [tap][128 rows][64 columns]
    float* blk = a.slabs + (size_t)slice * a.slab_stride + (size_t)pair_i * W5_BLOCK + (size_t)(trow * 5) * (128 * 64);
    const int wcol = ksplit ? 0 : wc * 32;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int col = wcol + ct * 16 + (lane & 15);
#pragma unroll
            for (int t = 0; t < TA; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int nl = wn * 64 + rt * 16 + 4 * (lane >> 4) + r;
                    blk[(size_t)t * (128 * 64) + nl * 64 + col] = accq[rt][ct][t][r];
                }
        }
}

// dW_k[row][col][tap] = sum over slices of the pair blocks; row = blockmap_k[n / Ch] * Ch + n % Ch for dG7 channel n
__global__ __launch_bounds__(256) void stw_reduce_kernel(const STWArgs a, const STWOut o) {
    // four consecutive columns per thread: 16-byte loads, all slices of the four in flight together (one element per thread with four
    // 4-byte loads in flight: 70 us for 146 MB; this form 57 us; a thread walking the 25 taps of its element — 25 consecutive floats per
    // thread, 100 bytes apart across the lanes of every store — 120 us)
    const long long total4 = (long long)a.npairs * (W5_BLOCK / 4);
    const long long e4 = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (e4 >= total4) return;
    const long long e = e4 * 4;
    const int c = (int)(e & 63), nl = (int)((e >> 6) & 127);
    const int tap = (int)((e >> 13) % 25), pi = (int)((e >> 13) / 25);
    const STWPair pr = a.pair[pi];
    const STWHalf hf = pr.h[c >> 5];
    const int n = pr.n0 + nl;
    if ((c & 31) >= hf.cn || n >= a.N7) return;      // (cn is a multiple of 8: the four columns are in or out together)
    const int db = o.blockmap[pr.tensor][n / a.Ch];
    if (db < 0) return;
    const bool one = o.ntaps[pr.tensor] == 1;        // the 1x1 tensor: only the centre tap of its blocks was written
    if (one && tap != 12) return;
    f32x4 p[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) p[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* src = a.slabs + e;
    int s = 0;
    for (; s + 4 <= a.n_slices; s += 4) {
        f32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const f32x4*>(src + (size_t)(s + k) * a.slab_stride);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) p[k][i] += v[k][i];
    }
    for (int k = 0; s < a.n_slices; ++s, ++k) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)s * a.slab_stride);
#pragma unroll
        for (int i = 0; i < 4; ++i) p[k][i] += v[i];
    }
    const int row = db * a.Ch + n % a.Ch, col = hf.cglobal + (c & 31);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float r = (p[0][i] + p[1][i]) + (p[2][i] + p[3][i]);   // (the order of the one-element version: bit-identical)
        o.dW[pr.tensor][one ? (size_t)row * o.Ct[pr.tensor] + col + i : ((size_t)row * o.Ct[pr.tensor] + col + i) * 25 + tap] = r;
    }
}

// pairs of one cell step. The row operand is dG8 [B,HW,8Ch] = the seven gate blocks (i,f,g | o | i',f',g') + d conv_last as block 7.
// Sources: 0 x (Cin), 1 h, 2 m, 3 c_new, 4 m_new (Ch each); tensors: 0 Wx, 1 Wh, 2 Wm, 3 Wo (k x k), 4 Wlast (1 x 1, pairs listed last).
int stw_build(STWArgs& a, STWOut& o, int B, int H, int W, int Cin, int Ch) {
    a.B = B; a.H = H; a.W = W; a.HW = H * W; a.Ch = Ch; a.N7 = 8 * Ch;
    a.npairs = 0; a.npairs5 = 0;
    static const signed char maps[5][8] = {{0, 1, 2, 6, 3, 4, 5, -1}, {0, 1, 2, 3, -1, -1, -1, -1}, {-1, -1, -1, -1, 0, 1, 2, -1},
                                           {-1, -1, -1, 0, -1, -1, -1, -1}, {-1, -1, -1, -1, -1, -1, -1, 0}};
    for (int k = 0; k < 5; ++k) for (int i = 0; i < 8; ++i) o.blockmap[k][i] = maps[k][i];
    o.Ct[0] = Cin; o.Ct[1] = Ch; o.Ct[2] = Ch; o.Ct[3] = 2 * Ch; o.Ct[4] = 2 * Ch;
    for (int k = 0; k < 5; ++k) o.ntaps[k] = k == 4 ? 1 : 25;
    struct Col { int tensor, src, C, cg; };
    const Col cols[7] = {{0, 0, Cin, 0}, {1, 1, Ch, 0}, {2, 2, Ch, 0}, {3, 3, Ch, 0}, {3, 4, Ch, Ch}, {4, 3, Ch, 0}, {4, 4, Ch, Ch}};
    for (int kpass = 0; kpass < 2; ++kpass)   // the k x k tensors' pairs first, then the 1 x 1 tensor's
    for (int n0 = 0; n0 < 8 * Ch; n0 += 128) {
        const int n1 = n0 + 128 < 8 * Ch ? n0 + 128 : 8 * Ch;
        for (int k = kpass ? 4 : 0; k < (kpass ? 5 : 4); ++k) {
            bool any = false;   // does the row tile hold a row of tensor k?
            for (int blk = n0 / Ch; blk <= (n1 - 1) / Ch; ++blk) any = any || maps[k][blk] >= 0;
            if (!any) continue;
            // the tensor's columns in 32-channel halves, two per pair (the two sources of Wo continue one another)
            STWHalf hv[64]; int nh = 0;
            for (const Col& cc : cols) {
                if (cc.tensor != k) continue;
                for (int c0 = 0; c0 < cc.C; c0 += 32) { if (nh >= 64) return -1; hv[nh++] = STWHalf{cc.src, c0, cc.C - c0 < 32 ? cc.C - c0 : 32, cc.cg + c0}; }
            }
            for (int i = 0; i < nh; i += 2) {
                if (a.npairs >= STW_MAX_PAIRS) return -1;
                STWPair& pr = a.pair[a.npairs++];
                pr.n0 = n0; pr.tensor = k; pr.h[0] = hv[i];
                pr.h[1] = i + 1 < nh ? hv[i + 1] : STWHalf{hv[i].src, 0, 0, 0};
                if (!kpass) a.npairs5 = a.npairs;
            }
        }
    }
    a.slab_stride = (size_t)a.npairs * W5_BLOCK;
    return a.npairs;
}

// K slices: about two rounds of one-per-CU workgroups (npairs x 3 workgroups per slice), every slice at least 8 items
int stw_slices(int npairs, long long items) {   // npairs: the k x k tensors' pairs (three workgroups each; the 1 x 1 pairs add one short one)
    int ns = (480 + npairs * 3 / 2) / (npairs * 3);
    // round 5: whole slices per XCD (stw_kernel's block decode) wherever a slice keeps >= 32 items — 8, 16 or 32 of them as the item list grows
    // (the deferred weight gradients of a whole pass: 19 456 items at B = 128): finer workgroups fill the last dispatch round of every XCD, a
    // slab costs 0.8 MB per pair. Measured, PredRNN-V2 training step (A/B builds -DVPX_STW_NS_FIXED=n, one box): B = 128 round-4 rule (4 slices,
    // contiguous ranges) 307.8 / 308.8 ms, 8 slices 297.8, 16 291.0, 32 288.1 / 287.8, 48 286.5; configs[4]'s shard (2 496 items) 82.5 -> 80.1 / 79.6 / 80.1.
    if (items >= 32 * 256) ns = 32;
    else if (items >= 16 * 128) ns = 16;
    else if (items >= 8 * 32) ns = 8;
#ifdef VPX_DEV_SWITCHES
    ns = dev_switch("VPX_STW_NS", ns);
#endif
#ifdef VPX_STW_NS_FIXED
    ns = VPX_STW_NS_FIXED ? VPX_STW_NS_FIXED : (480 + npairs * 3 / 2) / (npairs * 3);   // A/B builds (0: the round-4 rule)
#endif
    if (ns > items / 8) ns = (int)(items / 8);
    return ns < 1 ? 1 : ns;
}

#ifdef VPX_DEV_SWITCHES
static unsigned long long* g_stw_stamps = nullptr;
static int g_stw_stamp_block = 0;
extern "C" int vpx_dbg_stw_stamps(unsigned long long* dev_buf, int block) { g_stw_stamps = dev_buf; g_stw_stamp_block = block; return 0; }
extern "C" int vpx_dbg_stw_trace(unsigned long long* out32768) {
    return (int)hipMemcpyFromSymbol(out32768, HIP_SYMBOL(vpx::stw_trace), sizeof(unsigned long long) * 8192 * 4);
}
#endif

hipError_t launch_stw(const STWArgs& a_in, const STWOut& o, hipStream_t s) {
    STWArgs a = a_in;
#ifdef VPX_DEV_SWITCHES
    a.stamps = g_stw_stamps; a.stamp_block = g_stw_stamp_block;
#else
    a.stamps = nullptr; a.stamp_block = 0;
#endif
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = vpx_func_attr(reinterpret_cast<const void*>(&stw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, W5_LDS);
        if (e != hipSuccess) return e;
        attr_set = !g_dry_run;
    }
    const long long total = (long long)(a.npairs5 * 3 + (a.npairs - a.npairs5)) * a.n_slices;
    if (!ws_write_ok(a.slabs, (size_t)a.n_slices * a.slab_stride * sizeof(float), "weight-gradient slabs (stw_kernel)")) return hipErrorInvalidValue;
    VPX_LAUNCH(stw_kernel, dim3((unsigned)(8 * ((total + 7) / 8))), dim3(512), W5_LDS, s, a);
    hipError_t e = vpx_hip_last_error();
    if (e != hipSuccess) return e;
    const long long n = (long long)a.npairs * (W5_BLOCK / 4);
    VPX_LAUNCH(stw_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, o);
    return vpx_hip_last_error();
}

}  // namespace vpx
