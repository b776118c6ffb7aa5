// convq.hip — a schedule-driven convolution on the second-generation main loop (split-bf16 operands, all staging by LDS-DMA,
// v_mfma_f32_16x16x32_bf16): the EF stage glue of convlstm-shi (ef_blocks.py:15-49, layer table ef_conv_lstm.py:36-65) — stride-2
// convolutions, stride-2 transposed convolutions, the stride-1 transposed 3x3 — and their adjoints (data gradients).
//
// Everything such a layer needs is a list of TERMS: "tap (da, db) of source sub-image s contributes to output group g with
// weight tap w". A stride-2 convolution reads the four sub-images x[2i + sy, 2j + sx] of its input (addressed in place:
// the LDS-DMA source address carries the pixel stride), each with its own taps; a stride-2 transposed convolution computes its
// four output phases as four column groups of ONE N tile over the same input halo tile (like the four gates of the cell).
// The host pairs the taps of each 16-channel stage into K = 32 steps (any two taps of a stage, or the last tap of one stage with
// the first of the next: a lane's fragment address is base(lane) + (lane half ? offB : offA)), lays the steps out in weight
// chunks of 128 columns x 32 k and hands the kernel a table of at most 64 steps that covers all stages (or two / four stages,
// repeated, when every stage has the same taps: 3x3 and 5x5 'same' convolutions). The kernel keeps the table in one VGPR pair
// (lane i = step i, v_readlane: no memory access inside the loop) and runs cell2_kernel_q's pipeline: two activation stage
// buffers, a ring of three weight chunks, one sync point per chunk (counted vmcnt -> barrier -> next copies, spread behind the
// MFMAs), fragments of the next column tile / step read ahead of the MFMAs.
#include <stdlib.h>
#include <string.h>

#include "cell2_dev.h"
#include "vpx_host.h"

namespace vpx {

// ---- schedule entry (64 bits) --------------------------------------------------------------------------------------
// [0,17) offA  [17,34) offB  [34,42) mask of 16-column tiles  [42,45) bcol0  [45] first step of a chunk  [46] stage copy at this
// chunk's sync  [47] ... issued BEFORE the weight chunk  [48] late: this step's activation fragments are read after its sync
// [49,56) stage to copy (relative to the pass)  [56,63) stage of tap A (relative)
static inline unsigned long long cq_entry(unsigned offA, unsigned offB, unsigned mask, unsigned bcol0, unsigned newchunk, unsigned issue,
                                          unsigned afirst, unsigned late, unsigned istage, unsigned stA) {
    return (unsigned long long)offA | ((unsigned long long)offB << 17) | ((unsigned long long)mask << 34) | ((unsigned long long)bcol0 << 42) |
           ((unsigned long long)newchunk << 45) | ((unsigned long long)issue << 46) | ((unsigned long long)afirst << 47) |
           ((unsigned long long)late << 48) | ((unsigned long long)istage << 49) | ((unsigned long long)stA << 56);
}
struct CQDec { int offA, offB, mask, bcol0, newchunk, issue, afirst, late, istage, stA; };
__host__ __device__ __forceinline__ CQDec cq_decode(unsigned lo, unsigned hi) {
    CQDec d;
    d.offA = (int)(lo & 0x1ffff);
    d.offB = (int)((lo >> 17) | ((hi & 3u) << 15));
    d.mask = (int)((hi >> 2) & 0xff);
    d.bcol0 = (int)((hi >> 10) & 7);
    d.newchunk = (int)((hi >> 13) & 1);
    d.issue = (int)((hi >> 14) & 1);
    d.afirst = (int)((hi >> 15) & 1);
    d.late = (int)((hi >> 16) & 1);
    d.istage = (int)((hi >> 17) & 0x7f);
    d.stA = (int)((hi >> 24) & 0x7f);
    return d;
}

// ---- epilogue: bias, LeakyReLU, two fp32 destinations split by channel or one destination (also) in split operand format,
//      optional output-phase mapping (group g of the N tile = output phase (g >> 1, g & 1) of a stride-2 transposed convolution)
struct ConvQEpi {
    ConvQEpiArgs a;

    // phase_of_wave >= 0 (half-tile phase form, see convq_kernel): the wave holds ONE output phase for all 16 tile rows — LDS group g
    // is then the row group (rows 4 * g + prow ..) and every group belongs to phase `phase_of_wave`; else group g = phase / channel group
    __device__ __forceinline__ void store_sub(const float* ldsf, int lane, int b, int y0, int x0, int n_tile, int ngr, int prow, int H, int W,
                                              int phase_of_wave = -1) const {
        const int cg = lane & 7, p4 = lane >> 3;
        const bool v4 = ((a.Co | a.split | a.ld0 | a.ld1) & 3) == 0;
        float* const o0 = a.out0 ? a.out0 + (size_t)b * a.bstride0 : nullptr;
        float* const o1 = a.out1 ? a.out1 + (size_t)b * a.bstride1 : nullptr;
        char* const sp = a.sp_out ? a.sp_out + (size_t)b * a.sp_bstride : nullptr;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g >= ngr) continue;
            const int c = (a.phases ? n_tile : n_tile * a.gpt + g) * 32 + cg * 4;
            if (c >= a.Co) continue;
            const int ph = phase_of_wave >= 0 ? phase_of_wave : g;
            const int py = a.phases ? (ph >> 1) : 0, px = a.phases ? (ph & 1) : 0;
            const int rbase = phase_of_wave >= 0 ? 4 * g + prow : prow;
            const bool first = c < a.split;
            float* const ob = first ? o0 : o1;
            const unsigned ld = (unsigned)(first ? a.ld0 : a.ld1);
            const int cc = first ? c : c - a.split;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ip = k * 8 + p4;
                const int y = y0 + rbase + (ip >> 4), x = x0 + (ip & 15);
                f32x4 v = *reinterpret_cast<const f32x4*>(ldsf + g * 1024 + ip * 32 + cg * 4);
                const int my = y * a.oys + a.oyo + py, mx = x * a.oxs + a.oxo + px;
                if (y >= H || x >= W || my >= a.Hmem || mx >= a.Wmem) continue;
                const unsigned mp = (unsigned)(my * a.Wmem + mx);
                if (v4) {
                    if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + c);
                    if (a.accumulate && ob) v += *reinterpret_cast<const f32x4*>(ob + (size_t)mp * ld + cc);
                    if (a.leaky != 0.0f) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.0f ? v[q] : v[q] * a.leaky;
                    }
                    if (ob) *reinterpret_cast<f32x4*>(ob + (size_t)mp * ld + cc) = v;
                    if (sp && first) {
                        unsigned h[4], l[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) c2_split(v[q], h[q], l[q]);
                        char* dst = sp + (size_t)mp * ((unsigned)a.Co * 4u) + (unsigned)((c >> 3) * 32 + (c & 7) * 2);
                        *reinterpret_cast<uint2*>(dst) = uint2{h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
                        *reinterpret_cast<uint2*>(dst + 16) = uint2{l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int cq = c + q;
                        if (cq >= a.Co) continue;
                        const bool f1 = cq < a.split;
                        float* const oq = f1 ? o0 : o1;
                        if (!oq) continue;
                        const size_t eq = (size_t)mp * (unsigned)(f1 ? a.ld0 : a.ld1) + (f1 ? cq : cq - a.split);
                        float val = v[q] + (a.bias ? a.bias[cq] : 0.f);
                        if (a.accumulate) val += oq[eq];
                        if (a.leaky != 0.0f) val = val > 0.0f ? val : val * a.leaky;
                        oq[eq] = val;
                    }
                }
            }
        }
    }

    template <bool REMAP = false>
    __device__ __forceinline__ void finish16(const f32x4 (&acc)[4][8], char* smem, int wave, int lane, int b, int y0, int x0, int n_tile,
                                             int ngr, int H, int W) const {
        c2_barrier();
        float* ldsf = reinterpret_cast<float*>(smem + wave * 16384);
        const int c16 = lane & 15, q4 = lane >> 4;
#pragma unroll
        for (int mp = 0; mp < 2; ++mp) {
#pragma unroll
            for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                for (int nt = 0; nt < 8; ++nt)
                    if ((nt >> 1) < ngr)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            ldsf[(nt >> 1) * 1024 + (mm * 16 + 4 * q4 + r) * 32 + (nt & 1) * 16 + c16] = acc[2 * mp + mm][nt][r];
            if (REMAP) store_sub(ldsf, lane, b, y0, x0, n_tile, ngr, 2 * mp, H, W, wave);
            else store_sub(ldsf, lane, b, y0, x0, n_tile, ngr, 4 * wave + 2 * mp, H, W);
        }
    }
};

// ---- the kernel ----------------------------------------------------------------------------------------------------
// NTS = 16-column tiles per step (8: a step fills the whole N tile; 4 / 2 / 1: 2 / 4 / 8 steps sit side by side in one weight
// chunk — few output channels). PHASE: the (up to) four steps of a chunk accumulate into DIFFERENT accumulator tiles (output
// phases of a stride-2 transposed convolution, NTS = 2); otherwise every step accumulates into tiles [0, NTS).
// The loop body is one weight chunk = 8 column tiles with static tile / register indices; what is read at run time per step
// is only where its two taps sit (offA / offB) and whether it exists.
// NW = 8: 32x16-pixel tile, one workgroup per CU, ring of three weight chunks. NW = 4: the half tile (16x16 pixels, 256 threads,
// 80 KiB of LDS) — two workgroups per CU, so that one's stage-copy waits and its epilogue run under the other's MFMAs; ring of TWO
// weight chunks: chunk c+1 is requested at the sync of chunk c and awaited at the sync of chunk c+1 (its first column tile's
// fragments are read after that sync instead of ahead of it).
constexpr int convq_ppos(int halo, int nw) { return nw == 8 ? (halo == 2 ? 640 : 768) : (halo == 2 ? 384 : 448); }

template <int NTS, bool PHASE, int HALO, int NW>
__global__ __launch_bounds__(64 * NW, 2) void convq_kernel(const ConvQPlan P, const ConvQEpi epi) {
    constexpr int HW_ = 16 + HALO;                 // halo tile width
    constexpr int NT = 64 * NW, TH = 4 * NW;       // threads, tile rows
    constexpr int NPOS = (TH + HALO) * HW_;        // halo positions of the tile
    constexpr int PPOS = convq_ppos(HALO, NW);     // padded: 4 planes = NP pieces per thread
    constexpr int PLANE = PPOS * 16, ABUF = 4 * PLANE, NP = 4 * PPOS / NT;
    constexpr int NSUB = 8 / NTS;                  // steps per chunk
    constexpr int WSLOTS = NW == 8 ? 3 : 2;
    constexpr int WP = CQ_WCHUNK / (NT * 16);      // weight-chunk DMAs per thread: 2 | 4
    static_assert(NP == 5 || NP == 6, "stage copy: 5 or 6 pieces per thread");
    // Half-tile phase form: wave w owns output phase w for ALL 16 tile rows (its step of every chunk) instead of four rows of all
    // four phases. A chunk then costs a wave 4 weight-fragment reads instead of 16 (its two column tiles serve the four row groups):
    // 36 instead of 48 ds_read_b128 per 96 MFMAs — the phase form is LDS-read-bound. Same products, same order.
    constexpr bool REMAP = PHASE && NW == 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef VPX_DEV_SWITCHES
    const int dbg_bits = P.dbg;   // timing ablations of the developer build (VPX_CQ_DBG)
#else
    constexpr int dbg_bits = 0;   // (the product carries none of their tests: seven run-time tests per chunk of 96 MFMAs)
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kg = lane >> 4;

    const unsigned L = blockIdx.x;
    const long long total = (long long)P.grid_m * P.n_tiles;
    const long long per_xcd = (total + 7) / 8;
    const long long sidx = (long long)(L & 7) * per_xcd + (L >> 3);
    if ((long long)(L >> 3) >= per_xcd || sidx >= total) return;
    int mt = (int)(sidx / P.n_tiles);
    const int n_tile = (int)(sidx - (long long)mt * P.n_tiles);
    const int tx = mt % P.tiles_x;
    mt /= P.tiles_x;
    const int ty = mt % P.tiles_y;
    const int b = mt / P.tiles_y;
    const int x0 = tx * 16, y0 = ty * TH;
    int ngr = P.n_groups - n_tile * P.gpt;   // 32-column groups of this N tile that hold outputs
    if (ngr > P.gpt) ngr = P.gpt;
    const int tmask = (1 << (2 * ngr)) - 1;  // accumulator tiles of those groups: the MFMAs of the others are skipped

    char* const Abuf = smem;
    char* const Wbuf = smem + 2 * ABUF;

    // the schedule: lane i keeps entry i
    const unsigned long long my_entry = P.sched[lane];
    const int e_lo = (int)(unsigned)my_entry, e_hi = (int)(unsigned)(my_entry >> 32);
    auto entry = [&](int i) { return cq_decode((unsigned)__builtin_amdgcn_readlane(e_lo, i), (unsigned)__builtin_amdgcn_readlane(e_hi, i)); };
    const int e_off = (int)P.offs[lane];   // (offA / 16) | on << 15 | (offB / 16) << 16

    // this thread's pieces of a stage copy: byte offset from the tile's halo origin in the source (sub-)image + the piece's
    // 16 bytes inside the stage's 64-byte channel block, and per segment whether the position lies inside its (sub-)image
    // (all segments of a layer share the row / column pitch; their extents differ by a pixel at most: odd-sized stride-2 inputs)
    int pofs[NP], pval[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int piece = tid + NT * u;
        const int plane = piece / PPOS, pos = piece - plane * PPOS;
        const int hy = pos / HW_, hx = pos - hy * HW_;
        const int gy = y0 + P.oy + hy, gx = x0 + P.ox + hx;
        pofs[u] = hy * P.seg[0].rowpitch + hx * P.seg[0].colpitch + (plane & 1) * 32 + (plane >> 1) * 16;   // plane = part*2 + khalf
        int v = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < P.nseg && pos < NPOS && (unsigned)gy < (unsigned)P.seg[i].Hs && (unsigned)gx < (unsigned)P.seg[i].Ws) v |= 1 << i;
        pval[u] = v;
    }
    const int dma_off = (wave * 64) * 16;
    const char* const wtile = P.wpk + (size_t)n_tile * P.nchunk_total * CQ_WCHUNK + tid * 16;
    const int S = P.S;
    const long long tile_org = (long long)(y0 + P.oy) * P.seg[0].rowpitch + (long long)(x0 + P.ox) * P.seg[0].colpitch;

    // source of a stage (scalars): segment lookup + the image's base shifted to the tile's halo origin; stages >= S read zeros
    struct Src { const char* base; int seg; };
    auto stage_src = [&](int s) -> Src {
        int first = 0, si = 0;
        bool go = true;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (go && i + 1 < P.nseg && s >= first + P.seg[i].nstage) { first += P.seg[i].nstage; si = i + 1; }
            else go = false;
        }
        const CQSeg sg = P.seg[si];
        const int bb = b / sg.nT, tt = b - bb * sg.nT;
        Src r;
        r.base = s < S ? sg.sp + (size_t)bb * sg.bstride + (size_t)tt * sg.tstride + sg.org + (size_t)(sg.c0 + 16 * (s - first)) * 4 + tile_org : nullptr;
        r.seg = si;
        return r;
    };
    auto issue_A1 = [&](const Src& sc, int buf, int u) {
        const bool ok = (sc.base != nullptr) & (((pval[u] >> sc.seg) & 1) != 0);
        const char* src = ok ? sc.base + pofs[u] : reinterpret_cast<const char*>(c2_zero16);
        c2_dma16(src, Abuf + buf * ABUF + dma_off + u * (NT * 16));
    };
    auto issue_W1 = [&](int chunk, int slot, int u) {   // half u of a weight chunk
#pragma unroll
        for (int w = 0; w < WP / 2; ++w)
            c2_dma16(wtile + (size_t)chunk * CQ_WCHUNK + u * 8192 + w * (NT * 16), Wbuf + slot * CQ_WCHUNK + dma_off + u * 8192 + w * (NT * 16));
    };

    f32x4 acc[4][8];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[m][nt][r] = 0.0f;

    // lane bases: activation fragments (buffer 0, halo position (0,0), tile row 4 * wave, hi plane); weight fragments
    const int a_lane = (kg & 1) * PLANE + ((REMAP ? 0 : 4 * wave) * HW_ + r16) * 16;
    const int w_lane = kg * 2048 + r16 * 16 + (REMAP ? wave * 512 : 0);   // REMAP: the wave's two column tiles (its phase)
    const int tsel_shift = (kg >> 1) ? 16 : 0;   // which half of a packed offset pair this lane's k group reads
    // fragment address of a step for this lane: a_lane + 16 * (its half of the step's packed offsets)
    auto step_va = [&](int packed) { return a_lane + ((int)(((unsigned)packed >> tsel_shift) & 0x1fffu) << 4); };
    // activation fragments: with several steps per chunk (NSUB >= 2: 12 or 24 MFMAs per step) two sets — the next step's rows are
    // requested at the START of the current step; with one step per chunk (96 MFMAs) one set, refilled row by row behind the
    // last column tile's MFMAs (as cell2_kernel_q)
    constexpr int NSET = NSUB >= 2 ? 2 : 1;
    bf16x8 ah[NSET][4], al[NSET][4], bh[2], bl[2];
    auto load_A1 = [&](int set, int va, int m) {
        const char* a = smem + va + m * (HW_ * 16);
        ah[set][m] = *reinterpret_cast<const bf16x8*>(a);
        al[set][m] = *reinterpret_cast<const bf16x8*>(a + 2 * PLANE);
    };
    auto load_B = [&](int slot, int t) {   // weight fragments of chunk tile t -> set t & 1
        const char* w = Wbuf + slot * CQ_WCHUNK + w_lane + t * 256;
        bh[t & 1] = *reinterpret_cast<const bf16x8*>(w);
        bl[t & 1] = *reinterpret_cast<const bf16x8*>(w + 8192);
    };

    if (S > 0 && P.nsub > 0) {
        // ---- prologue: stage 0 (and stage 1 where the schedule expects it to be under way), chunks 0 and 1 ----
        Src aq_src = stage_src(0);
        if (dbg_bits & 32) return;
        if (!(dbg_bits & 16)) {
#pragma unroll
        for (int u = 0; u < NP; ++u) issue_A1(aq_src, 0, u);
        }
        if (P.pro_stage1) {
            const Src s1 = stage_src(1);
#pragma unroll
            for (int u = 0; u < NP; ++u) issue_A1(s1, 1, u);
        }
        issue_W1(0, 0, 0); issue_W1(0, 0, 1);
        if (P.nchunk_total > 1) { issue_W1(1, 1, 0); issue_W1(1, 1, 1); }
        C2_WAIT_VM(0);
        c2_barrier();
        if (dbg_bits & 64) return;

        int idx = 0, base = 0, c = 0;
        bool a_pending = false;                 // a stage copy issued AFTER the last weight chunk may still fly at the next sync
        CQDec cur = entry(0);                   // first step of the chunk (carries the chunk's events)
        int cur_off = __builtin_amdgcn_readlane(e_off, REMAP ? wave : 0);   // REMAP: the offsets of this wave's own step of the chunk
        {
            const int va = step_va(cur_off);
#pragma unroll
            for (int m = 0; m < 4; ++m) load_A1(0, va, m);
            load_B(0, 0);
            if (REMAP) load_B(0, 1);
        }
        while (true) {
            // ---- sync point of chunk c ----
            if (c > 0) {
                if (a_pending) { if (NP == 5) C2_WAIT_VM(5); else C2_WAIT_VM(6); } else C2_WAIT_VM(0);
                c2_barrier();
                if (WSLOTS == 2) load_B(c & 1, 0);   // ring of two: this chunk landed with this sync
                if (REMAP) load_B(c & 1, 1);
            }
            const int slot = c % WSLOTS, nslot = (c + 1) % WSLOTS;
            // ring of three: chunk c+2 into the slot chunk c-1 has left; ring of two: chunk c+1 (chunk 1 went out with the prologue)
            const int wq_c = WSLOTS == 3 ? c + 2 : (c > 0 ? c + 1 : 1 << 30);
            const int wq = (wq_c < P.nchunk_total && !(dbg_bits & 4)) ? wq_c : -1, wq_slot = wq_c % WSLOTS;
            bool aq = false, aq_first = false; int aq_buf = 0;
            if (cur.issue && !(dbg_bits & 2)) {
                const int st = base + cur.istage;
                if (st <= S) {           // st == S: zero fill (a cross step may read that buffer against zero weights)
                    aq_src = stage_src(st);
                    aq = true; aq_buf = st & 1; aq_first = cur.afirst != 0;
                }
            }
            a_pending = aq && !aq_first;
            if (cur.late) {              // a stage that landed with this very sync: its fragments could not be read ahead
                const int va = step_va(cur_off);
#pragma unroll
                for (int m = 0; m < 4; ++m) load_A1(0, va, m);
            }
            // next chunk's first entry (the schedule wraps around with the stage base advanced)
            int nidx = idx + NSUB, nbase = base;
            if (nidx >= P.nsub) { nidx = 0; nbase = base + P.SP; }
            const CQDec nxt = entry(nidx);
            const int nxt_off = __builtin_amdgcn_readlane(e_off, REMAP ? nidx + wave : nidx);
            const bool more = nbase + nxt.stA < S && c + 1 < P.nchunk_total;
#pragma unroll
            for (int j = 0; j < NSUB; ++j) {
                // step j of the chunk; the step after it (for the read-ahead of its activation rows)
                // REMAP: j is the row group; the step is the wave's own throughout the chunk
                const int oj = (REMAP || j == 0) ? cur_off : __builtin_amdgcn_readlane(e_off, idx + j);
                const int on_ = j + 1 < NSUB ? (REMAP ? cur_off : __builtin_amdgcn_readlane(e_off, idx + j + 1)) : nxt_off;
                const int n_va = step_va(on_) + (REMAP && j + 1 < NSUB ? (j + 1) * 4 * HW_ * 16 : 0);
                const bool on = (oj & 0x8000) != 0;
                constexpr int SET_SHIFT = 0;
                const int cs = NSET == 2 ? (j & 1) : 0, ns = NSET == 2 ? ((j + 1) & 1) : 0;   // this step's / the next step's fragment set
                (void)SET_SHIFT;
                if (NSET == 2) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) load_A1(ns, n_va, m);
                }
#pragma unroll
                for (int k = 0; k < NTS; ++k) {
                    const int t = j * NTS + k;           // tile inside the weight chunk (REMAP: position in time only)
                    const int at = PHASE ? t : k;        // accumulator tile
                    const int bt = REMAP ? k : t;        // weight-fragment set
                    if (!REMAP) { if (t < 7) load_B(slot, t + 1); else if (WSLOTS == 3) load_B(nslot, 0); }
                    const bool go = on && ((tmask >> (REMAP ? 2 * wave + k : at)) & 1) && !(dbg_bits & 1);
                    if (go) {
                        __builtin_amdgcn_s_setprio(1);
#pragma unroll
                        for (int m = 0; m < 4; ++m) {
                            f32x4 cc = acc[m][at];
                            cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[cs][m], bh[bt & 1], cc, 0, 0, 0);
                            cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[cs][m], bl[bt & 1], cc, 0, 0, 0);
                            cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[cs][m], bh[bt & 1], cc, 0, 0, 0);
                            acc[m][at] = cc;
                            if (NSET == 1 && k == NTS - 1) load_A1(0, n_va, m);   // the step's last tile frees row m: the next step's fragments
                        }
                        __builtin_amdgcn_s_setprio(0);
                    } else if (NSET == 1 && k == NTS - 1) {
#pragma unroll
                        for (int m = 0; m < 4; ++m) load_A1(0, n_va, m);
                    }
                    // ---- this sync's copies, behind the MFMAs of tiles 0..3: weight chunk, then the stage (or the stage first) ----
                    if (t == 0) {
                        if (aq && aq_first) { issue_A1(aq_src, aq_buf, 0); issue_A1(aq_src, aq_buf, 1); }
                        else if (wq >= 0) { issue_W1(wq, wq_slot, 0); issue_W1(wq, wq_slot, 1); }
                    }
                    if (t == 1 && aq) {
                        if (aq_first) { issue_A1(aq_src, aq_buf, 2); issue_A1(aq_src, aq_buf, 3); }
                        else { issue_A1(aq_src, aq_buf, 0); issue_A1(aq_src, aq_buf, 1); }
                    }
                    if (t == 2 && aq) {
                        if (aq_first) { issue_A1(aq_src, aq_buf, 4); if (NP > 5) issue_A1(aq_src, aq_buf, NP - 1); }
                        else { issue_A1(aq_src, aq_buf, 2); issue_A1(aq_src, aq_buf, 3); }
                    }
                    if (t == 3) {
                        if (aq && !aq_first) { issue_A1(aq_src, aq_buf, 4); if (NP > 5) issue_A1(aq_src, aq_buf, NP - 1); }
                        if (aq && aq_first && wq >= 0) { issue_W1(wq, wq_slot, 0); issue_W1(wq, wq_slot, 1); }
                    }
                }
            }
            if (!more) break;
            idx = nidx; base = nbase; cur = nxt; cur_off = nxt_off; ++c;
        }
        C2_WAIT_VM(0);                            // no copy may land in the epilogue's transposition space
    }
    if (!(dbg_bits & 8)) epi.template finish16<REMAP>(acc, smem, wave, lane, b, y0, x0, n_tile, ngr, P.H, P.W);
}

// ---- weight pack: [n_tile][chunk][part][k group][n][8 bf16]; the table says, per chunk of a pass and 16-column tile, which
//      (stage, weight tap) the two lane halves multiply ----
struct ConvQPackArgs {
    const float* w; long long s_oc, s_ic;
    int Co, col0, n_tiles, gpt, phases, colw;   // colw: columns of one step (128 unless several steps sit side by side in a chunk)
    int S, SP, nchunk_pass, nchunk_total;
    int nseg; int seg_nstage[4], seg_wc0[4];
    signed char tab[64][8][4];   // [chunk of a pass][16-column tile][stage A (relative), weight tap A, stage B, weight tap B]; tap < 0: zeros
};

__global__ void convq_pack_kernel(const ConvQPackArgs pk, char* __restrict__ dst) {
    const long long total = (long long)pk.n_tiles * pk.nchunk_total * (CQ_WCHUNK / 2);  // bf16 elements
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(e & 7);
        long long r = e >> 3;
        const int n = (int)(r & 127); r >>= 7;
        const int kg = (int)(r & 3); r >>= 2;
        const int part = (int)(r & 1); r >>= 1;
        const int q = (int)(r % pk.nchunk_total);
        const int n_tile = (int)(r / pk.nchunk_total);
        const int tsel = kg >> 1, khalf = kg & 1;
        const int pass = q / pk.nchunk_pass, cq = q - pass * pk.nchunk_pass;
        const signed char* t = pk.tab[cq][n >> 4];
        const int stage = pass * pk.SP + t[2 * tsel], tap = t[2 * tsel + 1];
        float v = 0.0f;
        if (tap >= 0 && stage < pk.S) {
            const int nn = n % pk.colw;   // column inside its step
            const int g = nn >> 5, j = nn & 31;
            const int oc = pk.phases ? n_tile * 32 + j : (n_tile * pk.gpt + g) * 32 + j;
            if (oc < pk.Co && (pk.phases || g < pk.gpt)) {
                int first = 0, si = 0;
                bool go = true;
                for (int k = 0; k < 3; ++k) {
                    if (go && k + 1 < pk.nseg && stage >= first + pk.seg_nstage[k]) { first += pk.seg_nstage[k]; si = k + 1; }
                    else go = false;
                }
                const int ic = pk.seg_wc0[si] + 16 * (stage - first) + khalf * 8 + i;
                v = pk.w[(long long)ic * pk.s_ic + (long long)(pk.col0 + oc) * pk.s_oc + tap];
            }
        }
        unsigned hi, lo;
        c2_split(v, hi, lo);
        reinterpret_cast<unsigned short*>(dst)[e] = (unsigned short)(part ? lo : hi);
    }
}

// ---- host: schedule construction ----------------------------------------------------------------------------------
namespace {

struct QTap { int stage, off, wtap; };       // off: byte offset of the tap inside its stage buffer (without the buffer's own offset)
struct QSub { int chunk, pos; QTap a, b; bool has_b; };   // pos: which of the chunk's 8 / NTS steps

struct QBuild {
    ConvQPlan P;
    ConvQPackArgs pk;
    int halo, NTS, nw;
};

// waves per workgroup: 4 = the half tile (two workgroups per CU), 3x3-halo layers only; VPX_CONVQ_NW=8 / experiment bit 4 keep the 32x16 tile
int convq_pick_nw(const ConvQProblem& pr) {
    static int env = -1;
    if (env < 0) env = dev_switch("VPX_CONVQ_NW", 4);
    if (g_experiment & 16) return 8;   // VPX_OPT_EXPERIMENT bit 4: the 32x16 tile (A/B runs, tests)
    return (pr.halo == 2 && env == 4) ? 4 : 8;
}

int seg_of_stage(const ConvQProblem& pr, int s) {
    int first = 0;
    for (int i = 0; i < pr.nseg; ++i) { if (s < first + pr.seg[i].nstage) return i; first += pr.seg[i].nstage; }
    return pr.nseg - 1;
}

int convq_build_impl(const ConvQProblem& pr, QBuild& out, int max_cpack = 8) {
    if (pr.halo != 2 && pr.halo != 4) { set_error("convq: halo %d", pr.halo); return VPX_ERR_ARG; }
    const int nw = convq_pick_nw(pr);
    const int HW_ = 16 + pr.halo, PPOS = convq_ppos(pr.halo, nw), ABUF = 4 * PPOS * 16, org = pr.halo / 2;
    int S = 0;
    for (int i = 0; i < pr.nseg; ++i) S += pr.seg[i].nstage;
    if (S < 1 || pr.nseg < 1 || pr.nseg > 4 || pr.ngs < 1 || pr.ngs > 4) { set_error("convq: bad problem (S=%d nseg=%d ngs=%d)", S, pr.nseg, pr.ngs); return VPX_ERR_ARG; }
    for (int i = 1; i < pr.nseg; ++i)
        if (pr.seg[i].rowpitch != pr.seg[0].rowpitch || pr.seg[i].colpitch != pr.seg[0].colpitch) { set_error("convq: segments must share their pixel pitches"); return VPX_ERR_ARG; }
    if (pr.phases ? pr.ngs > 4 : pr.ngs != 1) { set_error("convq: several group sets need the phase mapping"); return VPX_ERR_ARG; }
    const int n_tiles = pr.phases ? (pr.Co + 31) / 32 : ((pr.Co + 31) / 32 + 3) / 4;
    const int n_groups = pr.phases ? 4 * n_tiles : (pr.Co + 31) / 32;   // (phases: every N tile holds the four phases of its 32 channels)
    const int gpt = pr.phases ? 4 : (n_groups + n_tiles - 1) / n_tiles;
    // NTS = 16-column tiles of a step, rounded up to 1 / 2 / 4 / 8: 8 / NTS steps share one weight chunk (phases: the four phases'
    // steps; else consecutive steps of a layer with few output channels)
    int NTS = 8;
    if (pr.phases) NTS = 2;
    else if (n_tiles == 1) {
        const int ntn = ((pr.Co < 128 ? pr.Co : 128) + 15) / 16;
        NTS = ntn <= 1 ? 1 : (ntn <= 2 ? 2 : (ntn <= 4 ? 4 : 8));
        if (8 / NTS > max_cpack) NTS = 8 / max_cpack;
    }
    const int cpack = pr.phases ? 1 : 8 / NTS, NSUB = 8 / NTS;
    const bool all_stages = pr.periodic != 0;   // the one group set's terms apply to every stage
    const int nterm0 = pr.gs[0].nterm;
    // a short K loop is laid out in full; a long one with identical stages as a pass over two stages that repeats
    bool periodic = all_stages;
    if (periodic && (((S * nterm0 + 1) / 2 + cpack - 1) / cpack) * NSUB <= 64) periodic = false;
    if (periodic && cpack > 1) return convq_build_impl(pr, out, 1);   // (a pass of 2 stages does not divide into side-by-side steps)
    const int SP = periodic ? 2 : S;
    const int NSV = periodic ? 3 * SP : S;   // stages laid out (periodic: three passes, the middle one is the steady state)

    static thread_local QSub subs[4096];
    int nsub = 0, nchunk = 0;
    auto tap_of = [&](int stage, const ConvQTerm& t) {
        QTap q; q.stage = stage; q.off = ((t.da + org) * HW_ + (t.db + org)) * 16; q.wtap = t.wtap; return q;
    };
    if (!pr.phases) {
        // one group set: the taps of all stages in one line, paired in order (the last tap of a stage with the first of the next)
        static thread_local QTap line[8192];
        int n = 0;
        const ConvQGroupSet& gs = pr.gs[0];
        for (int s = 0; s < NSV; ++s)
            for (int k = 0; k < gs.nterm; ++k) {
                if (!all_stages && gs.term[k].seg != seg_of_stage(pr, s)) continue;
                if (n >= 8192) { set_error("convq: too many taps"); return VPX_ERR_UNSUPPORTED; }
                line[n++] = tap_of(s, gs.term[k]);
            }
        const int pairs = (n + 1) / 2;
        nchunk = (pairs + cpack - 1) / cpack;
        for (int p = 0; p < pairs; ++p) {
            if (nsub >= 4096) { set_error("convq: schedule too long"); return VPX_ERR_UNSUPPORTED; }
            QSub& q = subs[nsub++];
            q.chunk = p / cpack; q.pos = p % cpack;
            q.a = line[2 * p]; q.has_b = 2 * p + 1 < n; q.b = q.has_b ? line[2 * p + 1] : line[2 * p];
        }
    } else {
        // output phases: stage by stage, each phase pairs its own taps of that stage; a chunk holds the k-th pair of every phase
        for (int s = 0; s < NSV; ++s) {
            int maxp = 0;
            for (int g = 0; g < pr.ngs; ++g) {
                int cnt = 0;
                for (int k = 0; k < pr.gs[g].nterm; ++k) if (pr.gs[g].term[k].seg == seg_of_stage(pr, s)) ++cnt;
                if ((cnt + 1) / 2 > maxp) maxp = (cnt + 1) / 2;
            }
            for (int p = 0; p < maxp; ++p) {
                for (int g = 0; g < pr.ngs; ++g) {
                    QTap tp[2]; int cnt = 0, seen = 0;
                    for (int k = 0; k < pr.gs[g].nterm; ++k) {
                        if (pr.gs[g].term[k].seg != seg_of_stage(pr, s)) continue;
                        if (seen >= 2 * p && seen < 2 * p + 2) tp[cnt++] = tap_of(s, pr.gs[g].term[k]);
                        ++seen;
                    }
                    if (!cnt) continue;
                    if (nsub >= 4096) { set_error("convq: schedule too long"); return VPX_ERR_UNSUPPORTED; }
                    QSub& q = subs[nsub++];
                    q.chunk = nchunk + p; q.pos = pr.gs[g].nt0 / 2;
                    q.a = tp[0]; q.has_b = cnt > 1; q.b = q.has_b ? tp[1] : tp[0];
                }
            }
            nchunk += maxp;
        }
    }
    if (nchunk < 1) { set_error("convq: empty schedule"); return VPX_ERR_ARG; }

    // per stage: first / last chunk that reads it; then the sync at which its copy is issued (stage t >= 1; stage 0: prologue)
    static thread_local int fu[8192], lu[8192], ev_stage[8192], ev_first[8192], late_chunk[8192];
    if (NSV > 8192 || nchunk > 8192) { set_error("convq: problem too large"); return VPX_ERR_UNSUPPORTED; }
    for (int s = 0; s < NSV; ++s) { fu[s] = 1 << 30; lu[s] = -1; }
    for (int i = 0; i < nsub; ++i)
        for (int h = 0; h < 2; ++h) {
            const int st = h ? subs[i].b.stage : subs[i].a.stage;
            if (subs[i].chunk < fu[st]) fu[st] = subs[i].chunk;
            if (subs[i].chunk > lu[st]) lu[st] = subs[i].chunk;
        }
    for (int k = 0; k < nchunk; ++k) { ev_stage[k] = -1; ev_first[k] = 0; late_chunk[k] = 0; }
    for (int t = 1; t < NSV; ++t) {
        if (lu[t] < 0) continue;   // a stage nobody reads (cannot happen with well-formed terms)
        const int lo = t >= 2 ? lu[t - 2] + 1 : 0;
        // copy issued at sync ci after the weight chunk: landed for reads after sync ci + 2; before it: after sync ci + 1. Reads of
        // chunk fu's first step are issued at the end of chunk fu - 1 — unless that chunk is marked late (read after its own sync).
        // Order of preference: (0) after the chunk, read ahead; (1) after the chunk, first chunk late — the copy still has two
        // chunks of time; (2) before the chunk (must land within ONE chunk: the sync may stall); (3) before the chunk and late.
        int ci = -1, afirst = 0, late = 0;
        static const int m_hi[4] = {3, 2, 2, 1}, m_af[4] = {0, 0, 1, 1}, m_late[4] = {0, 1, 0, 1};
        static int mode_mask = -1;   // VPX_CONVQ_MODES: bit m allows mode m (experiments)
        if (mode_mask < 0) mode_mask = dev_switch("VPX_CONVQ_MODES", 15);
        for (int mode = 0; mode < 4 && ci < 0; ++mode) {
            if (!((mode_mask >> mode) & 1)) continue;
            const int hi = fu[t] - m_hi[mode];
            for (int k = lo; k <= hi; ++k)
                if (k >= 0 && ev_stage[k] < 0) { ci = k; afirst = m_af[mode]; late = m_late[mode]; break; }
        }
        if (ci < 0) {
            if (cpack > 1) return convq_build_impl(pr, out, 1);   // short stages: one step per weight chunk gives every stage its chunk boundaries
            set_error("convq: stage %d cannot be double-buffered (first use chunk %d, buffer free from chunk %d)", t, fu[t], lo);
            return VPX_ERR_UNSUPPORTED;
        }
        ev_stage[ci] = t; ev_first[ci] = afirst;
        if (late) late_chunk[fu[t]] = 1;
    }

    // extract the pass the kernel repeats: NSUB entries per chunk (missing steps: mask 0)
    const int nchunk_pass = periodic ? nchunk / 3 : nchunk;
    if (periodic && nchunk % 3) { set_error("convq: periodic schedule does not divide into passes"); return VPX_ERR_UNSUPPORTED; }
    const int c0 = periodic ? nchunk_pass : 0, s0 = periodic ? SP : 0;
    if (nchunk_pass * NSUB > 64) {
        if (cpack > 1) return convq_build_impl(pr, out, cpack / 2);
        set_error("convq: %d steps per pass (max 64)", nchunk_pass * NSUB);
        return VPX_ERR_UNSUPPORTED;
    }
    QBuild& B = out;
    memset(&B.P, 0, sizeof(B.P));
    memset(&B.pk, 0, sizeof(B.pk));
    for (int k = 0; k < 64; ++k) for (int t = 0; t < 8; ++t) { B.pk.tab[k][t][0] = 0; B.pk.tab[k][t][1] = -1; B.pk.tab[k][t][2] = 0; B.pk.tab[k][t][3] = -1; }
    bool rel1 = false, relSP1 = false;
    // default entries: no step; a chunk's first entry still carries its events and a stage index for the termination test
    static thread_local int first_stage[8192];
    for (int k = 0; k < nchunk; ++k) first_stage[k] = 1 << 30;
    for (int i = 0; i < nsub; ++i) if (subs[i].a.stage < first_stage[subs[i].chunk]) first_stage[subs[i].chunk] = subs[i].a.stage;
    for (int ck = 0; ck < nchunk_pass; ++ck) {
        unsigned issue = 0, afirst = 0, istage = 0;
        const int gc = c0 + ck;
        if (ev_stage[gc] >= 0) {
            issue = 1; afirst = (unsigned)ev_first[gc]; istage = (unsigned)(ev_stage[gc] - s0);
            if (istage == 1) rel1 = true;
            if ((int)istage == SP + 1) relSP1 = true;
            if (istage > 127) { set_error("convq: stage index out of range"); return VPX_ERR_UNSUPPORTED; }
        }
        const int fs = first_stage[gc] - s0;
        if (fs < 0 || fs > 127) { set_error("convq: stage index out of range"); return VPX_ERR_UNSUPPORTED; }
        for (int j = 0; j < NSUB; ++j)
            B.P.sched[ck * NSUB + j] = cq_entry(0, 0, 0, 0, j == 0, j == 0 ? issue : 0, j == 0 ? afirst : 0,
                                                j == 0 ? (unsigned)late_chunk[gc] : 0, j == 0 ? istage : 0, (unsigned)fs);
    }
    for (int i = 0; i < nsub; ++i) {
        const QSub& q = subs[i];
        if (q.chunk < c0 || q.chunk >= c0 + nchunk_pass) continue;
        const int ck = q.chunk - c0;
        const int stA = q.a.stage - s0, stB = q.b.stage - s0;
        if (stA < 0 || stB < 0 || stA > 127 || stB > 127) { set_error("convq: stage index out of range"); return VPX_ERR_UNSUPPORTED; }
        const unsigned offA = (unsigned)((q.a.stage & 1) * ABUF + q.a.off), offB = (unsigned)((q.b.stage & 1) * ABUF + q.b.off);
        unsigned long long& e = B.P.sched[ck * NSUB + q.pos];
        const CQDec d = cq_decode((unsigned)e, (unsigned)(e >> 32));
        e = cq_entry(offA, offB, 0xff, 0, (unsigned)d.newchunk, (unsigned)d.issue, (unsigned)d.afirst, (unsigned)d.late, (unsigned)d.istage, (unsigned)d.stA);
        B.P.offs[ck * NSUB + q.pos] = (offA >> 4) | 0x8000u | ((offB >> 4) << 16);
        for (int t = 0; t < NTS; ++t) {
            signed char* row = B.pk.tab[ck][q.pos * NTS + t];
            row[0] = (signed char)stA; row[1] = (signed char)q.a.wtap;
            row[2] = (signed char)stB; row[3] = (signed char)(q.has_b ? q.b.wtap : -1);
        }
    }
    // total chunks over the S real stages
    int nchunk_total;
    if (periodic) {
        const int pairs = (S * nterm0 + 1) / 2;
        nchunk_total = (pairs + cpack - 1) / cpack;
    } else {
        nchunk_total = nchunk;
    }
    ConvQPlan& P = B.P;
    P.B = pr.N; P.H = pr.H; P.W = pr.W;
    P.tiles_x = (pr.W + 15) / 16; P.tiles_y = (pr.H + 4 * nw - 1) / (4 * nw); P.n_tiles = n_tiles; P.grid_m = pr.N * P.tiles_x * P.tiles_y;
    P.n_groups = n_groups; P.gpt = gpt;
    P.S = S; P.SP = SP; P.nsub = nchunk_pass * NSUB; P.nchunk_total = nchunk_total;
    P.pro_stage1 = (periodic && relSP1 && !rel1) ? 1 : 0;
    P.oy = -org; P.ox = -org; P.nseg = pr.nseg;
    for (int i = 0; i < pr.nseg; ++i) P.seg[i] = pr.seg[i];
    ConvQPackArgs& pk = B.pk;
    pk.w = pr.w; pk.s_oc = pr.s_oc; pk.s_ic = pr.s_ic;
    pk.Co = pr.Co; pk.col0 = pr.col0; pk.n_tiles = n_tiles; pk.gpt = gpt; pk.phases = pr.phases;
    pk.colw = pr.phases ? 128 : 16 * NTS;
    pk.S = S; pk.SP = SP; pk.nchunk_pass = nchunk_pass; pk.nchunk_total = nchunk_total;
    pk.nseg = pr.nseg;
    for (int i = 0; i < pr.nseg; ++i) { pk.seg_nstage[i] = pr.seg[i].nstage; pk.seg_wc0[i] = pr.seg_wc0[i]; }
    B.halo = pr.halo; B.NTS = NTS; B.nw = nw;
    return VPX_OK;
}

// The schedule depends on the layer's geometry only, not on its tensors: a forward pass asks for the same few layers every step
// (three times per call: kernel choice, workspace size, launch), and at small batches the step is host-bound — keep the last builds.
int convq_build(const ConvQProblem& pr, QBuild& out) {
    struct Entry { ConvQProblem key; QBuild val; int rc, nw; bool used; };
    constexpr int NCACHE = 16;
    static thread_local Entry* cache = nullptr;
    static thread_local int next = 0;
    if (!cache) { cache = new Entry[NCACHE]; for (int i = 0; i < NCACHE; ++i) cache[i].used = false; }
    static thread_local ConvQProblem key;
    key = pr;
    key.w = nullptr;
    for (int i = 0; i < 4; ++i) { key.seg[i].sp = nullptr; key.seg[i].bstride = 0; key.seg[i].tstride = 0; key.seg[i].nT = 0; }
    for (int i = pr.nseg; i < 4; ++i) { memset(&key.seg[i], 0, sizeof(CQSeg)); key.seg_wc0[i] = 0; }
    for (int g = 0; g < 4; ++g) {
        if (g >= pr.ngs) { memset(&key.gs[g], 0, sizeof(ConvQGroupSet)); continue; }
        for (int k = pr.gs[g].nterm; k < 32; ++k) memset(&key.gs[g].term[k], 0, sizeof(ConvQTerm));
    }
    const int nw = convq_pick_nw(pr);   // (an experiment switch may change the tile form between calls)
    for (int i = 0; i < NCACHE; ++i) {
        if (!cache[i].used || cache[i].nw != nw || memcmp(&cache[i].key, &key, sizeof(ConvQProblem)) != 0) continue;
        if (cache[i].rc != VPX_OK) return convq_build_impl(pr, out);   // (sets the error text again)
        out = cache[i].val;
        for (int k = 0; k < pr.nseg; ++k) out.P.seg[k] = pr.seg[k];
        out.pk.w = pr.w;
        return VPX_OK;
    }
    const int rc = convq_build_impl(pr, out);
    Entry& e = cache[next];
    next = (next + 1) % NCACHE;
    e.key = key; e.rc = rc; e.nw = nw; e.used = true;
    if (rc == VPX_OK) e.val = out;
    return rc;
}

template <int NTS, bool PHASE, int HALO, int NW>
hipError_t launch_convq_t(const ConvQPlan& P, const ConvQEpi& epi, hipStream_t s) {
    constexpr int LDS = 2 * 4 * convq_ppos(HALO, NW) * 16 + (NW == 8 ? 3 : 2) * CQ_WCHUNK;   // 128 | 80 KiB (3x3 halo)
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = vpx_func_attr(reinterpret_cast<const void*>(&convq_kernel<NTS, PHASE, HALO, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        attr_set = !g_dry_run;
    }
    const long long per_xcd = ((long long)P.grid_m * P.n_tiles + 7) / 8;
    VPX_LAUNCH((convq_kernel<NTS, PHASE, HALO, NW>), dim3((unsigned)(per_xcd * 8)), dim3(64 * NW), LDS, s, P, epi);
    return vpx_hip_last_error();
}
template <int NTS, bool PHASE>
hipError_t launch_convq_h2(const ConvQPlan& P, const ConvQEpi& epi, int nw, hipStream_t s) {
    return nw == 4 ? launch_convq_t<NTS, PHASE, 2, 4>(P, epi, s) : launch_convq_t<NTS, PHASE, 2, 8>(P, epi, s);
}

}  // namespace

size_t convq_wpk_bytes(const ConvQProblem& pr) {
    static thread_local QBuild b;
    if (convq_build(pr, b) != VPX_OK) return 0;
    return (size_t)b.P.n_tiles * b.P.nchunk_total * CQ_WCHUNK;
}

int convq_run(const ConvQProblem& pr, const ConvQEpiArgs& ea_in, char* wpk, bool weights_packed, hipStream_t s) {
    static thread_local QBuild b;
    int rc = convq_build(pr, b);
    if (rc != VPX_OK) return rc;
    if (!weights_packed) {
        const long long total = (long long)b.pk.n_tiles * b.pk.nchunk_total * (CQ_WCHUNK / 2);
        if (!ws_write_ok(wpk, (size_t)total * 2, "weight pack (convq_pack_kernel)")) { set_error("%s", ws_violation()); return VPX_ERR_WORKSPACE; }
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        VPX_LAUNCH(convq_pack_kernel, dim3(blocks), dim3(256), 0, s, b.pk, wpk);
        VPX_CHECK_HIP(vpx_hip_last_error());
    }
    b.P.wpk = wpk;
    { static int dbg = -1; if (dbg < 0) dbg = dev_switch("VPX_CQ_DBG", 0); b.P.dbg = dbg; }
    ConvQEpi epi{ea_in};
    epi.a.gpt = b.P.gpt;
    epi.a.phases = pr.phases;
    hipError_t e;
    if (b.halo == 4) e = launch_convq_t<8, false, 4, 8>(b.P, epi, s);
    else if (pr.phases) e = launch_convq_h2<2, true>(b.P, epi, b.nw, s);
    else if (b.NTS == 8) e = launch_convq_h2<8, false>(b.P, epi, b.nw, s);
    else if (b.NTS == 4) e = launch_convq_h2<4, false>(b.P, epi, b.nw, s);
    else if (b.NTS == 2) e = launch_convq_h2<2, false>(b.P, epi, b.nw, s);
    else e = launch_convq_h2<1, false>(b.P, epi, b.nw, s);
    VPX_CHECK_HIP(e);
    return VPX_OK;
}


// =====================================================================================================================
// c5: 5x5 'same' convolutions of the ST-LSTM step on 16x16-pixel tiles (round 4) — PredRNN's maps are 16x16 (MovingMNIST / 4)
// or 32x32 (128x128 / 4): the 32-row tile of convq_kernel<.., 4, 8> is half empty there and its one workgroup per CU has nothing
// to run under its prologue / epilogue. What makes a 20x20-position halo tile fit two workgroups per CU:
//   * a K stage is EIGHT channels (one 32-byte group of the split format: 16 B hi | 16 B lo per pixel), not sixteen: a stage buffer
//     is 2 planes x 400 positions x 16 B (padded to 512 positions = 4 LDS-DMA pieces per thread exactly): 16 KiB, two of them;
//   * a K = 32 step multiplies FOUR (stage, tap) slots — the lane's k group (lane >> 4) selects the slot, its 8 k values are the
//     slot's 8 channels. Slots run tap-major through the stages, 100 slots = 25 steps per PERIOD of four stages; at most two
//     stages are live at a time (stage j of a period is read in steps floor(25 j / 4) .. floor((25 j + 24) / 4)), so two buffers
//     suffice: stage j + 1 is requested at period step 0 / 7 / 13 / 19, three steps or more before its first fragment read and
//     after the last read of the stage whose buffer it takes;
//   * weights: one 16 KiB (8 KiB for 64-column tiles) chunk per step, [half][part][k group][column][8 bf16], ring of three as in
//     cell2_kernel_q<.., 8>: sync point S_q before the 6th column tile of step q (counted vmcnt -> barrier -> copy of chunk q + 2).
// LDS: 32 + 48 = 80 KiB (NT = 8) / 64 KiB (NT = 4; the epilogue's transposition space) -> two workgroups per CU.
// One launch runs a TABLE of jobs over the same split-format source (the ST-LSTM backward's dG8): job = (channel ranges of the
// source = its K, packed weights, destination, 128- or 64-column N tiles). The jobs of a launch differ in K (dx: 7Ch, dh: 4Ch,
// dm: 3Ch channels): every XCD gets every job's tiles of ITS pixel tiles, longest job first, the N tiles of a pixel tile
// adjacent in the XCD's dispatch order (they share the stage in its L2).
// =====================================================================================================================
// KS = 5 (the ST-LSTM step) or 3 (the ConvLSTM step on small grids, round 4: "c3"). 3x3: 18x18 halo positions (planes padded to 384 = 3
// pieces per thread), nine slots per stage = nine steps per period of four stages; a stage lives 2-3 steps only, so the period's four
// stages have a buffer EACH (4 x 12 KiB): stage j of the NEXT period is requested at period step 3 / 5 / 7 / (next) 0, right after the
// last read of stage j of this one, six steps before its first read.
template <int NT, int KS = 5> struct C5Geom {
    static constexpr int HWP = 16 + KS - 1;               // halo tile width / height: 20 | 18
    static constexpr int NPOS = HWP * HWP;                // 400 | 324
    static constexpr int PLP = KS == 5 ? 512 : 384;       // padded positions of a plane
    static constexpr int PLANE = PLP * 16;                // one plane (hi or lo) of a stage buffer
    static constexpr int ABUF = 2 * PLANE;                // 16 | 12 KiB
    static constexpr int NPC = 2 * PLP / 256;             // stage-copy pieces per thread: 4 | 3
    static constexpr int NBUF = KS == 5 ? 2 : 4;
    static constexpr int SPS = KS * KS;                   // slots per stage = steps per period of four stages
    static constexpr int WCH = NT * 2048;                 // weight chunk of one K = 32 step: 16 | 8 | 4 KiB
    static constexpr int WP = WCH / (256 * 16);           // DMAs per thread and chunk: 4 | 2 | 1
    static constexpr int HCOLS = NT >= 4 ? 64 : NT * 16;  // columns of a chunk half ([half][part][k group][HCOLS][16 B])
    static constexpr int KGS = HCOLS * 16, PARTS = 4 * KGS;
    // Weight ring: RD chunks, chunk q + RD - 1 is requested at the sync point of step q. Three for the 128-column tiles (96 MFMAs per wave
    // and step: one step of MFMA time covers the copy's latency); the narrow tiles run 48 / 24 MFMAs per step — less than a copy takes to
    // land (measured: c5_kernel<2, 3> 34 us for 36 steps = 0.9 us per step, the L2 latency, against 14 k cycles of MFMAs) — and request
    // three / four steps ahead.
    // The period (nine / 25 steps) is unrolled (below); a ring of three (3x3) / five (5x5, narrow tiles) makes the slot of every step a constant.
    static constexpr int RD = (NT == 8 || KS == 3) ? 3 : 5;
    static constexpr bool STATIC_SLOT = SPS % RD == 0;
    static constexpr int PD = RD - 1;                     // prefetch distance in steps
    static constexpr int WAITN = (PD - 2) * WP;           // copies that may still fly at a sync point (the chunks after q + 1)
    static constexpr int LDS = NBUF * ABUF + RD * WCH >= 65536 ? NBUF * ABUF + RD * WCH : 65536;   // >= 64 KiB: 4 waves x 16 KiB in the epilogue
};
template <int N> __device__ __forceinline__ void c5_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

__device__ __forceinline__ void c5_chan_of_stage(const C5Job& j, int s8, int& src, int& chan) {   // source and first channel of the job's stage s8
    int c = 8 * s8;
    if (c < j.r_n[0]) { src = j.r_src[0]; chan = j.r_c0[0] + c; return; }
    c -= j.r_n[0];
    if (c < j.r_n[1]) { src = j.r_src[1]; chan = j.r_c0[1] + c; return; }
    src = j.r_src[2]; chan = j.r_c0[2] + (c - j.r_n[1]);
}

// ---- ST-LSTM forward epilogues on the c5 tile: the wave's 4 tile rows x (NG gates x 32 channels) go through its private 16 KiB of
//      LDS so that a lane owns four channels of a pixel for every gate (16-byte global accesses) ----
__device__ __forceinline__ void c5_store_split4(char* sp, size_t pix, int Ch, int c, const f32x4& v) {
    unsigned h[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) c2_split(v[q], h[q], l[q]);
    char* dst = sp + pix * ((size_t)Ch * 4u) + (unsigned)((c >> 3) * 32 + (c & 7) * 2);
    *reinterpret_cast<uint2*>(dst) = uint2{h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
    *reinterpret_cast<uint2*>(dst + 16) = uint2{l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
}

// gates (predrnn.py:61-77): groups (i, f, g[, o_pre]) of 32 channels -> s_new = f * s_in + i * g, delta = i * g
__device__ __forceinline__ void c5_finish_gates(const f32x4 (&acc)[4][8], char* smem, int wave, int lane, int b, int y0, int x0, int n_tile,
                                                const C5Job& J, int H, int W) {
    c2_barrier();
    float* ldsf = reinterpret_cast<float*>(smem + wave * 16384);
    const int c16 = lane & 15, q4 = lane >> 4, cg = lane & 7, p4 = lane >> 3;
    const int ch = n_tile * 32 + cg * 4, Ch = J.Ch, ng = J.ng;
#pragma unroll
    for (int mp = 0; mp < 2; ++mp) {
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int nt = 0; nt < 8; ++nt)
                if ((nt >> 1) < ng)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        ldsf[(nt >> 1) * 1024 + (mm * 16 + 4 * q4 + r) * 32 + (nt & 1) * 16 + c16] = acc[2 * mp + mm][nt][r];
        if (ch < Ch) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ip = k * 8 + p4;
                const int y = y0 + 4 * wave + 2 * mp + (ip >> 4), x = x0 + (ip & 15);
                if (y >= H || x >= W) continue;
                const size_t pix = ((size_t)b * H + y) * W + x, sidx = pix * Ch + ch;
                const f32x4 vi = *reinterpret_cast<const f32x4*>(ldsf + ip * 32 + cg * 4);
                const f32x4 vf = *reinterpret_cast<const f32x4*>(ldsf + 1024 + ip * 32 + cg * 4);
                const f32x4 vg = *reinterpret_cast<const f32x4*>(ldsf + 2048 + ip * 32 + cg * 4);
                const f32x4 sin = *reinterpret_cast<const f32x4*>(J.e_in0 + sidx);
                f32x4 gi, gf, gg, dl, sn;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    gi[q] = sigmoid_f(vi[q]); gf[q] = sigmoid_f(vf[q] + J.fbias); gg[q] = tanh_f(vg[q]);
                    dl[q] = gi[q] * gg[q];
                    sn[q] = gf[q] * sin[q] + dl[q];
                }
                *reinterpret_cast<f32x4*>(J.e_out[0] + sidx) = sn;
                *reinterpret_cast<f32x4*>(J.e_out[1] + sidx) = dl;
                if (ng == 4) *reinterpret_cast<f32x4*>(J.e_out[2] + sidx) = *reinterpret_cast<const f32x4*>(ldsf + 3072 + ip * 32 + cg * 4);
                if (J.e_out[3]) {
                    float* gs = J.e_out[3] + pix * 3 * Ch + ch;
                    *reinterpret_cast<f32x4*>(gs) = gi;
                    *reinterpret_cast<f32x4*>(gs + Ch) = gf;
                    *reinterpret_cast<f32x4*>(gs + 2 * Ch) = gg;
                }
                if (J.e_sp) c5_store_split4(J.e_sp, pix, Ch, ch, sn);
            }
        }
    }
}

// output gate (predrnn.py:80-81): acc = conv_o(mem) -> h_new = sigmoid(o_pre + acc) * tanh(conv_last(mem)); 64-column tiles (groups 0, 1)
__device__ __forceinline__ void c5_finish_out(const f32x4 (&acc)[4][8], char* smem, int wave, int lane, int b, int y0, int x0, int n_tile,
                                              int gpt, const C5Job& J, int H, int W) {
    c2_barrier();
    float* ldsf = reinterpret_cast<float*>(smem + wave * 16384);
    const int c16 = lane & 15, q4 = lane >> 4, cg = lane & 7, p4 = lane >> 3;
    const int Ch = J.Ch;
#pragma unroll
    for (int mp = 0; mp < 2; ++mp) {
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int nt = 0; nt < 8; ++nt)
                if ((nt >> 1) < gpt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        ldsf[(nt >> 1) * 1024 + (mm * 16 + 4 * q4 + r) * 32 + (nt & 1) * 16 + c16] = acc[2 * mp + mm][nt][r];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = (n_tile * gpt + g) * 32 + cg * 4;
            if (g >= gpt || ch >= Ch) continue;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ip = k * 8 + p4;
                const int y = y0 + 4 * wave + 2 * mp + (ip >> 4), x = x0 + (ip & 15);
                if (y >= H || x >= W) continue;
                const size_t pix = ((size_t)b * H + y) * W + x, sidx = pix * Ch + ch;
                const f32x4 v = *reinterpret_cast<const f32x4*>(ldsf + g * 1024 + ip * 32 + cg * 4);
                const f32x4 op = *reinterpret_cast<const f32x4*>(J.e_in0 + sidx), lc = *reinterpret_cast<const f32x4*>(J.e_in1 + sidx);
                f32x4 o, tl, hn;
#pragma unroll
                for (int q = 0; q < 4; ++q) { o[q] = sigmoid_f(op[q] + v[q]); tl[q] = tanh_f(lc[q]); hn[q] = o[q] * tl[q]; }
                *reinterpret_cast<f32x4*>(J.e_out[0] + sidx) = hn;
                if (J.e_out[1]) { *reinterpret_cast<f32x4*>(J.e_out[1] + sidx) = o; *reinterpret_cast<f32x4*>(J.e_out[2] + sidx) = tl; }
                if (J.e_sp) c5_store_split4(J.e_sp, pix, Ch, ch, hn);
            }
        }
    }
}

// ConvLSTM step on a gate-interleaved N tile of GC = NT * 4 channels (columns g * GC + j, gates i, f, g, o): the wave's 64 pixels x NT * 16
// columns go through its private LDS as [pixel][column]; a lane then owns four channels of a pixel for all four gates.
// The epilogue's global operands (c_in, the three peepholes) do not depend on the K loop: c5_clstm_prefetch requests them BEFORE it, so that
// their latency (1.5-2 k cycles of a 6 k cycle epilogue on the small grids this form serves) passes under the MFMAs.
template <int NT> struct ClstmPre {
    static constexpr int NPASS = (NT * 16 / 4 / 4);   // = 64 / PPP
    f32x4 cp[NPASS], wi[NPASS], wf[NPASS], wo[NPASS];
};
template <int NT>
__device__ __forceinline__ void c5_clstm_prefetch(ClstmPre<NT>& pre, int wave, int lane, int b, int y0, int x0, int n_tile, const C5Job& J, int H, int W) {
    constexpr int NC = NT * 16, GC = NC / 4, LPP = GC / 4, PPP = 64 / LPP;
    const int cgi = lane % LPP, pl = lane / LPP;
    const int ch = n_tile * GC + cgi * 4, Ch = J.Ch;
#pragma unroll
    for (int pass = 0; pass < 64 / PPP; ++pass) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        pre.cp[pass] = z; pre.wi[pass] = z; pre.wf[pass] = z; pre.wo[pass] = z;
        const int pix = pass * PPP + pl;
        const int y = y0 + 4 * wave + (pix >> 4), x = x0 + (pix & 15);
        if (ch >= Ch || y >= H || x >= W) continue;
        const size_t e = ((size_t)y * W + x) * Ch + ch, sidx = ((size_t)b * H * W) * Ch + e;
        if (J.e_in0) pre.cp[pass] = *reinterpret_cast<const f32x4*>(J.e_in0 + sidx);
        if (J.e_in1) { pre.wi[pass] = *reinterpret_cast<const f32x4*>(J.e_in1 + e); pre.wf[pass] = *reinterpret_cast<const f32x4*>(J.e_in2 + e); }
        if (J.e_in3) pre.wo[pass] = *reinterpret_cast<const f32x4*>(J.e_in3 + e);
    }
}
template <int NT>
__device__ __forceinline__ void c5_finish_clstm(const f32x4 (&acc)[4][NT], const ClstmPre<NT>& pre, char* smem, int wave, int lane, int b, int y0, int x0,
                                                int n_tile, const C5Job& J, int H, int W) {
    static_assert(NT <= 4, "64 pixels x NT * 16 columns must fit the wave's 16 KiB");
    constexpr int NC = NT * 16, GC = NC / 4, LPP = GC / 4, PPP = 64 / LPP;   // lanes per pixel, pixels per pass
    static_assert(ClstmPre<NT>::NPASS == 64 / PPP, "");
    c2_barrier();
    float* ldsf = reinterpret_cast<float*>(smem + wave * 16384);
    const int c16 = lane & 15, q4 = lane >> 4;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ldsf[(m * 16 + 4 * q4 + r) * NC + nt * 16 + c16] = acc[m][nt][r];
    const int cgi = lane % LPP, pl = lane / LPP;
    const int ch = n_tile * GC + cgi * 4, Ch = J.Ch;
    if (ch >= Ch) return;
    f32x4 bi = {0.f, 0.f, 0.f, 0.f}, bf = bi, bg = bi, bo = bi;
    if (J.bias) {
        bi = *reinterpret_cast<const f32x4*>(J.bias + J.gate_pos[0] * Ch + ch);
        bf = *reinterpret_cast<const f32x4*>(J.bias + J.gate_pos[1] * Ch + ch);
        bg = *reinterpret_cast<const f32x4*>(J.bias + J.gate_pos[2] * Ch + ch);
        bo = *reinterpret_cast<const f32x4*>(J.bias + J.gate_pos[3] * Ch + ch);
    }
    float* const hout = J.e_out[0] ? J.e_out[0] + (size_t)b * J.h_bstride : nullptr;
    char* const hsp = J.e_sp ? J.e_sp + (size_t)b * J.sp_bstride : nullptr;
#pragma unroll
    for (int pass = 0; pass < 64 / PPP; ++pass) {
        const int pix = pass * PPP + pl;
        const int y = y0 + 4 * wave + (pix >> 4), x = x0 + (pix & 15);
        if (y >= H || x >= W) continue;
        const size_t hw = (size_t)y * W + x, e = hw * Ch + ch, sidx = ((size_t)b * H * W) * Ch + e;
        const float* row = ldsf + pix * NC + cgi * 4;
        const f32x4 vi = *reinterpret_cast<const f32x4*>(row), vf = *reinterpret_cast<const f32x4*>(row + GC);
        const f32x4 vg = *reinterpret_cast<const f32x4*>(row + 2 * GC), vo = *reinterpret_cast<const f32x4*>(row + 3 * GC);
        const f32x4 cp = pre.cp[pass], wi = pre.wi[pass], wf = pre.wf[pass], wo = pre.wo[pass];
        f32x4 cn, hn;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float i_ = sigmoid_f(vi[q] + bi[q] + wi[q] * cp[q]), f_ = sigmoid_f(vf[q] + bf[q] + wf[q] * cp[q]);
            const float g_ = tanh_f(vg[q] + bg[q]);
            cn[q] = lstm_c(f_, cp[q], i_, g_);
            const float o_ = sigmoid_f(vo[q] + bo[q] + wo[q] * cn[q]);
            hn[q] = o_ * tanh_f(cn[q]);
        }
        *reinterpret_cast<f32x4*>(J.e_out[1] + sidx) = cn;
        if (hout) *reinterpret_cast<f32x4*>(hout + e) = hn;
        if (hsp) c5_store_split4(hsp, hw, Ch, ch, hn);
    }
}

template <int NT, int KS = 5>
__global__ __launch_bounds__(256, 2) void c5_kernel(const C5Plan P) {
    using G = C5Geom<NT, KS>;
    constexpr int C5_ABUF = G::ABUF, C5_PLANE = G::PLANE, HWP = G::HWP, ORG = KS / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kg = lane >> 4;

    // ---- which (job, pixel tile, N tile): XCD x owns the pixel tiles m = x, x + 8, ...; its list = job 0's tiles, job 1's, ... ----
    const int xcd = blockIdx.x & 7;
    int i = __builtin_amdgcn_readfirstlane((int)(blockIdx.x >> 3));
    const int Mx = (P.m_tiles + 7) / 8;
    // (static indices into the kernel-argument table: a run-time index would move the whole table to scratch memory)
    C5Job J = P.job[0];
    bool found = false;
#pragma unroll
    for (int k = 0; k < C5_MAX_JOBS; ++k) {
        if (!found && k < P.njobs) {
            const int cnt = Mx * P.job[k].n_tiles;
            if (i < cnt) { J = P.job[k]; found = true; }
            else i -= cnt;
        }
    }
    if (!found) return;
    const int mt = (P.order ? i % Mx : i / J.n_tiles) * 8 + xcd, n_tile = P.order ? i / Mx : i % J.n_tiles;
    if (mt >= P.m_tiles) return;
    const int tpi = P.tiles_x * P.tiles_y;
    const int b = mt / tpi, tr = mt - b * tpi;
    const int ty = tr / P.tiles_x, tx = tr - ty * P.tiles_x;
    const int y0 = ty * 16, x0 = tx * 16;

    char* const Abuf = smem;
    char* const Wbuf = smem + G::NBUF * C5_ABUF;
    const int dma_off = wave * 1024;

    // stage copy: NPC pieces per thread = [plane][PLP positions]
    int pixoff[G::NPC], choff[G::NPC];
#pragma unroll
    for (int u = 0; u < G::NPC; ++u) {
        const int piece = tid + 256 * u;
        const int plane = piece / G::PLP, pos = piece - plane * G::PLP;
        const int hy = pos / HWP, hx = pos - hy * HWP;
        const int gy = y0 - ORG + hy, gx = x0 - ORG + hx;
        pixoff[u] = (pos < G::NPOS && (unsigned)gy < (unsigned)P.H && (unsigned)gx < (unsigned)P.W) ? gy * P.W + gx : -1;
        choff[u] = plane * 16;
    }
    // the job's (up to three) channel ranges: base of this sample's first channel and pixel pitch, resolved ONCE (a look-up of P.src inside
    // the K loop is a scalar load from the kernel arguments: s_waitcnt lgkmcnt(0) with every fragment read of the step in flight)
    // (kept in VECTOR registers on purpose: the kernel sits at the SGPR limit — as scalars these nine values spill to scratch memory,
    //  whose reloads wait on vmcnt(0) with the copies in flight; as an indexed array they live there outright)
    auto in_vgpr = [](unsigned x) { unsigned r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(x)); return r; };
    auto range_of = [&](const char* p0, long long bs, int n, unsigned long long& base) {
        const unsigned long long v = (p0 != nullptr && n > 0) ? reinterpret_cast<unsigned long long>(p0 + (size_t)b * bs) : 0ull;
        base = ((unsigned long long)in_vgpr((unsigned)(v >> 32)) << 32) | in_vgpr((unsigned)v);
    };
    unsigned long long rb0, rb1, rb2;
    range_of(J.r_p[0], J.r_bs[0], J.r_n[0], rb0);
    range_of(J.r_p[1], J.r_bs[1], J.r_n[1], rb1);
    range_of(J.r_p[2], J.r_bs[2], J.r_n[2], rb2);
    const unsigned rp0 = in_vgpr((unsigned)J.r_prow[0]), rp1 = in_vgpr((unsigned)J.r_prow[1]), rp2 = in_vgpr((unsigned)J.r_prow[2]);
    const int rn0 = J.r_n[0], rn1 = J.r_n[1];
    auto issue_A = [&](int s8, int buf) {   // stages past the job's K are filled with zeros (their weights are zeros too)
        const char* base = nullptr;
        unsigned prow = 0;
        if (s8 < J.S8) {
            const int c = 8 * s8;
            const bool in0 = c < rn0, in1 = c - rn0 < rn1;
            // (masks, not conditionals: a conditional between captured values becomes the selection of an ADDRESS inside the closure — a
            //  run-time index that keeps the closure, and with it every local it refers to, in scratch memory)
            const unsigned long long m0 = in0 ? ~0ull : 0ull, m1 = (!in0 && in1) ? ~0ull : 0ull, m2 = ~(m0 | m1);
            const unsigned long long rb = (rb0 & m0) | (rb1 & m1) | (rb2 & m2);
            const int cc = c - (in0 ? 0 : rn0) - ((in0 || in1) ? 0 : rn1);
            base = rb != 0 ? reinterpret_cast<const char*>(rb + (unsigned long long)cc * 4u) : nullptr;
            prow = (rp0 & (unsigned)m0) | (rp1 & (unsigned)m1) | (rp2 & (unsigned)m2);
        }
#pragma unroll
        for (int u = 0; u < G::NPC; ++u) {
            const char* src = (base != nullptr && pixoff[u] >= 0) ? base + (size_t)((unsigned)pixoff[u] * (unsigned long long)prow) + choff[u]
                                                                   : reinterpret_cast<const char*>(c2_zero16);
            c2_dma16(src, Abuf + buf * C5_ABUF + dma_off + u * 4096);
        }
    };
    const char* const wtile = J.wpk + (size_t)n_tile * J.Q * G::WCH + tid * 16;
    auto issue_W = [&](int q, int slot) {
#pragma unroll
        for (int w = 0; w < G::WP; ++w) c2_dma16(wtile + (size_t)q * G::WCH + w * 4096, Wbuf + slot * G::WCH + dma_off + w * 4096);
    };

    f32x4 acc[4][NT];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[m][nt][r] = 0.0f;

    // fragments. A: row m of the wave's four tile rows, this lane's slot of period step p (k group kg): slot s = 4p + kg,
    // stage j = s / 25 (buffer j & 1), tap t = s % 25 = (dy, dx)
    const int a_lane = ((4 * wave) * HWP + r16) * 16;
    auto a_off = [&](int p) {
        const int sl = 4 * p + kg;
        if constexpr (KS == 5) {
            const int j = (sl * 41) >> 10, t = sl - 25 * j;
            const int dy = (t * 13) >> 6, dx = t - 5 * dy;
            return a_lane + (j & 1) * C5_ABUF + (dy * HWP + dx) * 16;
        } else {
            const int j = (sl * 57) >> 9, t = sl - 9 * j;          // sl / 9 for sl < 36
            const int dy = (t * 11) >> 5, dx = t - 3 * dy;         // t / 3 for t < 9
            return a_lane + j * C5_ABUF + (dy * HWP + dx) * 16;
        }
    };
    bf16x8 ah[4], al[4], bh[2], bl[2];
    auto load_A1 = [&](int off, int m) {
        const char* a = smem + off + m * (HWP * 16);
        ah[m] = *reinterpret_cast<const bf16x8*>(a);
        al[m] = *reinterpret_cast<const bf16x8*>(a + C5_PLANE);
    };
    // B: chunk [half = nt >> 2][part][k group][64 columns][16 B] (NT = 4: one half)
    const int w_lane = G::NBUF * C5_ABUF + kg * G::KGS + r16 * 16;
    auto load_B = [&](int slot, int nt) {
        const char* w = smem + w_lane + slot * G::WCH + (nt >> 2) * 8192 + (nt & 3) * 256;
        bh[nt & 1] = *reinterpret_cast<const bf16x8*>(w);
        bl[nt & 1] = *reinterpret_cast<const bf16x8*>(w + G::PARTS);
    };

    const int Q = J.Q;
    const int nt_active = J.nt_active;
#if defined(VPX_DEV_SWITCHES) && defined(VPX_C5_ABL)   // (its branches cost the loop a third: a build of its own, hipcc -DVPX_C5_ABL)
    const int abl = P.ablate;   // 1: no stage copies, 2: no weight copies, 4: no sync points, 8: no MFMAs, 16: no A reads, 32: no B reads (K loop only)
#else
    constexpr int abl = 0;
#endif
    // developer timing stamps (P.stamps != nullptr only from tools/): shader clock of one workgroup's waves at the phase boundaries
    const bool stamp = P.stamps != nullptr && (int)blockIdx.x == P.stamp_block && lane == 0;
    if (stamp) P.stamps[wave * 8 + 0] = __builtin_amdgcn_s_memtime();
    bool flies = false;                         // a stage copy was issued in the previous step (it may still fly at this step's sync)
    ClstmPre<(KS == 3 ? NT : 1)> clstm_pre;
    if constexpr (KS == 3) {
        // (ahead of every copy: vmcnt counts in order, the counted waits below name the NEWEST requests that may still fly)
        if (J.epi == 3) c5_clstm_prefetch<NT>(clstm_pre, wave, lane, b, y0, x0, n_tile, J, P.H, P.W);
    }
    if (Q > 0) {
        issue_A(0, 0);
        issue_W(0, 0);
        if constexpr (KS == 3) {
            // the first step needs stage 0 and chunk 0 only: chunk 1 and stage 1 fly behind them (stage 1 is first read for step 2), stages
            // 2 and 3 are requested by steps 0 and 1 of the first period (first read for steps 4 and 6)
            static_assert(G::PD == 2, "");
            if (Q >= 2) { issue_W(1, 1); issue_A(1, 1); c5_wait_vm<G::WP + G::NPC>(); flies = true; }
            else C2_WAIT_VM(0);
        } else if (Q >= G::PD) {   // chunks 1 .. PD - 1 go out too; everything before them has landed when only they still fly
#pragma unroll
            for (int c = 1; c < G::PD; ++c) issue_W(c, c);
            c5_wait_vm<(G::PD - 1) * G::WP>();
        } else {
            for (int c = 1; c < Q; ++c) issue_W(c, c);
            C2_WAIT_VM(0);
        }
        c2_barrier();
        const int o0 = a_off(0);
#pragma unroll
        for (int m = 0; m < 4; ++m) load_A1(o0, m);
        load_B(0, 0);
    }
    constexpr int SYNC_NT = NT == 8 ? 5 : (NT == 4 ? 2 : 1);    // the sync point sits before this column tile of every step
    constexpr int STAGE_NT = SYNC_NT + 1 < NT ? SYNC_NT + 1 : NT - 1;   // the stage copy follows the weight copy (same tile when there is no later one)
    if (stamp) P.stamps[wave * 8 + 1] = __builtin_amdgcn_s_memtime();
    // A wave issues one instruction per four cycles: a step of the narrow tiles holds 24 or 48 MFMAs (384 / 768 cycles of the pipe), and
    // the step's scalar bookkeeping (which stage to request, the fragment offsets: three integer divisions per lane, the ring slot) stood
    // beside them with a hundred instructions of its own — with every copy, read, MFMA and sync point removed the loop still took 950
    // cycles per step (developer build, VPX_C5_ABLATE = 63). 3x3: the period's nine steps are unrolled — the stage schedule and the ring
    // slot (9 = 3 x RD) become constants, the fragment offsets of the nine steps nine registers.
    constexpr int UNR = G::SPS;
    int aoffs[G::SPS];
#pragma unroll
    for (int p = 0; p < G::SPS; ++p) aoffs[p] = a_off(p);
    int q = 0, slot_rt = 0;                     // global step, its ring slot (q % RD)
#pragma unroll 1
    for (int P0 = 0; q < Q; P0 += 4) {          // period: stages P0 .. P0 + 3
#pragma unroll UNR
        for (int p = 0; p < G::SPS; ++p) {
            if (q >= Q) continue;   // (not a loop exit: a loop with sync points unrolls only with an exact trip count)
            const int slot = G::STATIC_SLOT ? p % G::RD : slot_rt;
            const int nslot = slot == G::RD - 1 ? 0 : slot + 1;
            // stage requested at this step (after its sync point). 5x5: period steps 0 / 7 / 13 / 19 -> stages P0 + 1 / + 2 / + 3 / + 4;
            // 3x3: steps 3 / 5 / 7 -> stages P0 + 4 / + 5 / + 6 (the next period's first three), step 0 -> stage P0 + 3; the first period's
            // stages 2 and 3 at its steps 0 and 1 (the prologue requests stages 0 and 1 only)
            int want;
            if constexpr (KS == 5) want = p == 0 ? P0 + 1 : (p == 7 ? P0 + 2 : (p == 13 ? P0 + 3 : (p == 19 ? P0 + 4 : -1)));
            else want = p == 3 ? P0 + 4 : (p == 5 ? P0 + 5 : (p == 7 ? P0 + 6 : (p == 0 ? (P0 > 0 ? P0 + 3 : 2) : ((p == 1 && P0 == 0) ? 3 : -1))));
            const bool issue = want >= 0 && want <= J.S8;   // (== S8: zero fill of the buffer a partial last step still reads)
            const int np = p == G::SPS - 1 ? 0 : p + 1;
            const int n_off = aoffs[np];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (nt == SYNC_NT) {
                    // ---- sync point S_q: chunk q + 1 has landed (the stage requested one step ago may still fly) ----
                    if (abl & 4) {
                    } else if constexpr (G::PD == 2) {
                        if (flies) c5_wait_vm<G::NPC>(); else C2_WAIT_VM(0);
                    } else {
                        // chunks q + 2 .. q + PD - 1 may fly (a stage copy among them lands early: in-order completion); at the tail, where
                        // no further chunk is behind q + 1, everything must have landed
                        if (q + G::PD <= Q) c5_wait_vm<G::WAITN>(); else C2_WAIT_VM(0);
                    }
                    if (!(abl & 4)) c2_barrier();
                }
                if (abl & 32) {
                } else if (nt < NT - 1) load_B(slot, nt + 1);
                else load_B(nslot, 0);
                __builtin_amdgcn_s_setprio(1);
                // (a gate job of three gates holds six column tiles of the eight: it skips the MFMAs of the last two. Only those two carry
                //  the test — a branch around every MFMA group cuts the step into basic blocks the scheduler cannot interleave.)
                const bool go = (NT < 8 || nt < 6) ? true : nt < nt_active;
                if (nt < NT - 1) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        if (go && !(abl & 8)) {
                            f32x4 c = acc[m][nt];
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[m], bh[nt & 1], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], bl[nt & 1], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], bh[nt & 1], c, 0, 0, 0);
                            acc[m][nt] = c;
                        }
                    }
                } else {
                    // the step's last tile, in three rounds over the four rows: the lo fragments are free after the first round and the next
                    // step's are requested there — eight MFMAs ahead of their first use (the scheduler left to itself requests all eight
                    // fragments behind the tenth MFMA and the next step opens with a wait); the hi fragments follow row by row in the third
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        if (go && !(abl & 8)) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[m], bh[nt & 1], acc[m][nt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(abl & 16)) {
#pragma unroll
                        for (int m = 0; m < 4; ++m) al[m] = *reinterpret_cast<const bf16x8*>(smem + n_off + m * (HWP * 16) + C5_PLANE);
                    }
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        if (go && !(abl & 8)) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], bl[nt & 1], acc[m][nt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        if (go && !(abl & 8)) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], bh[nt & 1], acc[m][nt], 0, 0, 0);
                        if (!(abl & 16)) ah[m] = *reinterpret_cast<const bf16x8*>(smem + n_off + m * (HWP * 16));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_s_setprio(0);
                if (nt == SYNC_NT) { if (q + G::PD < Q && !(abl & 2)) issue_W(q + G::PD, slot == 0 ? G::RD - 1 : slot - 1); }
                if (nt == STAGE_NT && issue && !(abl & 1)) issue_A(want, KS == 5 ? (want & 1) : (want & 3));
            }
            flies = issue;
            slot_rt = nslot;
            ++q;
        }
    }
    C2_WAIT_VM(0);
    if (stamp) P.stamps[wave * 8 + 2] = __builtin_amdgcn_s_memtime();
    // ---- epilogue: ConvQEpi (fp32 destination, optional accumulate), the wave's four tile rows ----
    ConvQEpi epi{};
    epi.a.Co = J.Co; epi.a.split = J.Co; epi.a.gpt = NT / 2; epi.a.phases = 0; epi.a.accumulate = J.accumulate;
    epi.a.oys = 1; epi.a.oxs = 1; epi.a.oyo = 0; epi.a.oxo = 0; epi.a.Hmem = P.H; epi.a.Wmem = P.W;
    epi.a.out0 = J.out; epi.a.bstride0 = J.out_bstride; epi.a.ld0 = J.ld;
    int ngr = (J.Co + 31) / 32 - n_tile * (NT / 2);
    if (ngr > NT / 2) ngr = NT / 2;
    if constexpr (KS == 3) {
        if (J.epi == 3) {
            c5_finish_clstm<NT>(acc, clstm_pre, smem, wave, lane, b, y0, x0, n_tile, J, P.H, P.W);
            if (stamp) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); P.stamps[wave * 8 + 3] = __builtin_amdgcn_s_memtime(); }
            return;
        }
    }
    if constexpr (NT == 8) {
        if (J.epi == 1) c5_finish_gates(acc, smem, wave, lane, b, y0, x0, n_tile, J, P.H, P.W);
        else if (J.epi == 2) c5_finish_out(acc, smem, wave, lane, b, y0, x0, n_tile, 4, J, P.H, P.W);
        else epi.finish16(acc, smem, wave, lane, b, y0, x0, n_tile, ngr, P.H, P.W);
    } else {
        f32x4 acc8[4][8];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
                if constexpr (NT == 4) acc8[m][nt] = nt < 4 ? acc[m][nt] : f32x4{0.f, 0.f, 0.f, 0.f};
                else acc8[m][nt] = nt < 2 ? acc[m][nt < 2 ? nt : 0] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        if (J.epi == 2) c5_finish_out(acc8, smem, wave, lane, b, y0, x0, n_tile, NT / 2, J, P.H, P.W);
        else epi.finish16(acc8, smem, wave, lane, b, y0, x0, n_tile, ngr, P.H, P.W);
    }
}

// weight pack of one job: [n_tile][step q][(half)][part][k group][column][8 bf16]. Element (column oc, source channel c of range r,
// tap t) = w_r[ocidx * s_oc_r + (c0_r + c) * s_c_r + tap'] with tap' = 24 - t for a data gradient (flip) and ocidx = col0_r + oc, or —
// gate-interleaved N tiles (forward): column n of N tile nt is gate n / 32, channel nt * 32 + n % 32: ocidx = gate0_r[n / 32] + that.
struct C5PackArgs {
    C5PackRange rg[3];
    int nrange, r_n[3];
    int flip, NT, gates;       // gates: number of gate groups of an N tile (0 = plain column order)
    int gate_major;            // gates > 0: plain column order oc = gate * Co + channel instead of gate-interleaved N tiles
    int sps, gc;               // slots per stage (taps: 25 | 9); channels per gate of a gate-interleaved N tile (32; the ConvLSTM forms: NT * 4)
    int S8, Q, Co, n_tiles;    // Co: plain: output channels; gates: channels per gate (Ch)
};
// One launch packs ALL the jobs a library call has prepared since its last c5 launch (round 5: the 128x128x3 shard spent 5 % of its
// training step in 246 pack launches of ~20 us — one per cell, direction and K chunk; they now share a launch per c5 launch).
constexpr int C5_MAX_PACKS = 16;
struct C5PackBatch { int n, _p; C5PackArgs a[C5_MAX_PACKS]; char* dst[C5_MAX_PACKS]; };
static_assert(sizeof(C5PackBatch) <= 4000, "C5PackBatch travels as a kernel argument");
__global__ void c5_pack_kernel(const C5PackBatch pb) {
    const C5PackArgs& pk = pb.a[blockIdx.y];
    char* const __restrict__ dst = pb.dst[blockIdx.y];
    const int wch2 = pk.NT * 1024;   // bf16 elements of a chunk
    const long long total = (long long)pk.n_tiles * pk.Q * wch2;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(e & 7);
        long long r = e >> 3;
        int n;
        if (pk.NT >= 4) { n = (int)(r & 63); r >>= 6; } else { n = (int)(r & 31); r >>= 5; }
        const int kg = (int)(r & 3); r >>= 2;
        const int part = (int)(r & 1); r >>= 1;
        if (pk.NT == 8) { n += (int)(r & 1) * 64; r >>= 1; }
        const int q = (int)(r % pk.Q);
        const int n_tile = (int)(r / pk.Q);
        const int p = q % pk.sps, sl = 4 * p + kg;
        const int j = sl / pk.sps, t = sl - pk.sps * j;
        const int s8 = 4 * (q / pk.sps) + j;
        float v = 0.0f;
        if (s8 < pk.S8) {
            int c = 8 * s8 + i, ri = 0;
            if (c >= pk.r_n[0]) { c -= pk.r_n[0]; ri = 1; if (c >= pk.r_n[1]) { c -= pk.r_n[1]; ri = 2; } }
            const C5PackRange rg = pk.rg[ri];
            long long ocidx = -1;
            if (pk.gates && pk.gate_major) {
                const int oc = n_tile * (pk.NT * 16) + n, g = oc / pk.Co;
                if (g < pk.gates) ocidx = rg.gate0[g] + (oc - g * pk.Co);
            } else if (pk.gates) {
                const int g = n / pk.gc, chn = n_tile * pk.gc + (n - g * pk.gc);
                if (g < pk.gates && chn < pk.Co) ocidx = rg.gate0[g] + chn;
            } else {
                const int oc = n_tile * (pk.NT * 16) + n;
                if (oc < pk.Co) ocidx = rg.gate0[0] + oc;
            }
            if (ocidx >= 0) v = rg.w[ocidx * rg.s_oc + (long long)(rg.c0 + c) * rg.s_c + (pk.flip ? pk.sps - 1 - t : t)];
        }
        unsigned hi, lo;
        c2_split(v, hi, lo);
        reinterpret_cast<unsigned short*>(dst)[e] = (unsigned short)(part ? lo : hi);
    }
}

size_t c5_wpk_bytes(int K, int Co, int NT, int gates, int ks) {
    const int gc = ks == 3 ? NT * 4 : 32;   // channels per gate of a gate-interleaved N tile
    const int S8 = K / 8, Q = (ks * ks * S8 + 3) / 4, n_tiles = gates ? (Co + gc - 1) / gc : (Co + NT * 16 - 1) / (NT * 16);
    return (size_t)n_tiles * Q * NT * 2048;
}

// fills the job's derived fields (S8, Q, n_tiles, nt_active) and packs its weights into job.wpk unless `packed`.
// gates = 0: plain column order, Co output channels; gates = 3 | 4: gate-interleaved N tiles of 32 channels (NT = 8), Co = channels per gate
// packs prepared and not yet launched (this thread's running library call); flushed by the next launch_c5, dropped when a new call
// carves its workspace (a call that failed between prepare and launch must not leave its pointers behind)
static thread_local C5PackBatch g_c5_pending = {};
static thread_local long long g_c5_pending_max = 0;
void c5_drop_pending_packs() { g_c5_pending.n = 0; g_c5_pending_max = 0; }
hipError_t c5_flush_packs(hipStream_t s) {
    if (g_c5_pending.n == 0) return hipSuccess;
    int blocks = (int)((g_c5_pending_max + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    VPX_LAUNCH(c5_pack_kernel, dim3(blocks, g_c5_pending.n), dim3(256), 0, s, g_c5_pending);
    c5_drop_pending_packs();
    return vpx_hip_last_error();
}

int c5_prepare_job(C5Job& j, int NT, const C5PackRange* rg, int gates, int flip, bool packed, hipStream_t s, int gate_major, int ks) {
    const int sps = ks * ks, gc = ks == 3 ? NT * 4 : 32;
    int K = 0;
    for (int i = 0; i < 3; ++i) {
        if (i >= j.nrange) { j.r_n[i] = 0; j.r_c0[i] = 0; j.r_src[i] = 0; }
        K += j.r_n[i];
        if ((j.r_n[i] | j.r_c0[i]) & 7) { set_error("c5: channel ranges in 8s"); return VPX_ERR_ARG; }
    }
    if (gates && !gate_major && ks == 5 && NT != 8) { set_error("c5: gate-interleaved 5x5 tiles need NT = 8"); return VPX_ERR_ARG; }
    j.S8 = K / 8; j.Q = (sps * j.S8 + 3) / 4;
    // gate_major: j.Co = all columns (gates x channels per gate), plain epilogue; else gates > 0: j.Co = channels per gate
    j.n_tiles = (gates && !gate_major) ? (j.Co + gc - 1) / gc : (j.Co + NT * 16 - 1) / (NT * 16);
    j.nt_active = (gates && !gate_major && ks == 5) ? 2 * gates : NT;
    if (!packed) {
        C5PackArgs pk{};
        pk.nrange = j.nrange; pk.flip = flip; pk.NT = NT; pk.gates = gates; pk.gate_major = gate_major; pk.sps = sps; pk.gc = gc;
        for (int i = 0; i < 3; ++i) { pk.r_n[i] = j.r_n[i]; if (i < j.nrange) pk.rg[i] = rg[i]; }
        pk.S8 = j.S8; pk.Q = j.Q; pk.Co = (gates && gate_major) ? j.Co / gates : j.Co; pk.n_tiles = j.n_tiles;
        const long long total = (long long)j.n_tiles * j.Q * NT * 1024;
        if (!ws_write_ok(j.wpk, (size_t)total * 2, "weight pack (c5_pack_kernel)")) { set_error("%s", ws_violation()); return VPX_ERR_WORKSPACE; }
        if (g_c5_pending.n == C5_MAX_PACKS) VPX_CHECK_HIP(c5_flush_packs(s));
        g_c5_pending.a[g_c5_pending.n] = pk;
        g_c5_pending.dst[g_c5_pending.n] = const_cast<char*>(j.wpk);
        ++g_c5_pending.n;
        if (total > g_c5_pending_max) g_c5_pending_max = total;
    }
    return VPX_OK;
}

// chunk k of ks of a job's K (whole 8-channel stages): the chunk's channel ranges and the matching weight ranges
void c5_chunk_job(const C5Job& full, const C5PackRange* prf, int k, int ks, C5Job& j, C5PackRange* pr) {
    const int S8 = (full.r_n[0] + full.r_n[1] + full.r_n[2]) / 8;
    const int c_lo = 8 * (int)((long long)S8 * k / ks), c_hi = 8 * (int)((long long)S8 * (k + 1) / ks);
    j = full;
    j.nrange = 0;
    for (int i = 0; i < 3; ++i) { j.r_src[i] = 0; j.r_c0[i] = 0; j.r_n[i] = 0; }
    int base = 0;
    for (int i = 0; i < full.nrange; ++i) {
        const int lo = c_lo > base ? c_lo : base, hi = c_hi < base + full.r_n[i] ? c_hi : base + full.r_n[i];
        if (lo < hi) {
            const int q = j.nrange++;
            j.r_src[q] = full.r_src[i]; j.r_c0[q] = full.r_c0[i] + (lo - base); j.r_n[q] = hi - lo;
            pr[q] = prf[i]; pr[q].c0 += lo - base;
        }
        base += full.r_n[i];
    }
}
// bytes of chunk k's pack
size_t c5_chunk_wpk_bytes(int K, int k, int ks, int cols, int NT) {   // (5x5 jobs)
    const int S8 = K / 8;
    const int s8 = (int)((long long)S8 * (k + 1) / ks) - (int)((long long)S8 * k / ks);
    return (size_t)((cols + NT * 16 - 1) / (NT * 16)) * ((25 * s8 + 3) / 4) * NT * 2048;
}

template <int NT, int KS>
static hipError_t launch_c5_t(const C5Plan& P, unsigned grid, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = vpx_func_attr(reinterpret_cast<const void*>(&c5_kernel<NT, KS>), hipFuncAttributeMaxDynamicSharedMemorySize, C5Geom<NT, KS>::LDS);
        if (e != hipSuccess) return e;
        attr_set = !g_dry_run;
    }
    constexpr int lds = C5Geom<NT, KS>::LDS;
    VPX_LAUNCH((c5_kernel<NT, KS>), dim3(grid), dim3(256), lds, s, P);
    return vpx_hip_last_error();
}

#ifdef VPX_DEV_SWITCHES
static unsigned long long* g_c5_stamps = nullptr;
static int g_c5_stamp_block = 0;
extern "C" int vpx_dbg_c5_stamps(unsigned long long* dev_buf, int block) { g_c5_stamps = dev_buf; g_c5_stamp_block = block; return 0; }
#endif

hipError_t launch_c5(const C5Plan& P_in, int NT, hipStream_t s) {
    { const hipError_t ep = c5_flush_packs(s); if (ep != hipSuccess) return ep; }   // the packs this launch (and later ones of the call) read
    C5Plan P = P_in;
    // Block order inside an XCD. The 128-column launches (the forward gate groups: 14 N tiles per pixel tile, 23 MB of packed weights
    // against 4 MB of L2) run the pixel tiles of an N tile next to each other — the XCD's 16 pixel tiles then stream one N tile's weights
    // together: FETCH_SIZE 211 -> 170 MB per launch at B = 128, same time; the narrow launches keep the N tiles of a pixel tile together
    // (conv_o: 49 MB that way, 55 MB the other).
    P.order = NT == 8 ? 1 : 0;
#ifdef VPX_DEV_SWITCHES
    P.stamps = g_c5_stamps; P.stamp_block = g_c5_stamp_block;
    P.ablate = dev_switch("VPX_C5_ABLATE", 0);
    { const int o = dev_switch("VPX_C5_ORDER", -1); if (o >= 0) P.order = (o >> (NT == 8 ? 0 : (NT == 4 ? 1 : 2))) & 1; }   // bit 0 / 1 / 2: 128- / 64- / 32-column launches
#else
    P.ablate = 0;
#endif
    for (int j = 0; j < P.njobs; ++j)
        for (int r = 0; r < 3; ++r) {
            C5Job& J = P.job[j];
            const C5Src& sc = P.src[(J.r_n[r] > 0 && J.r_src[r] >= 0 && J.r_src[r] < 4) ? J.r_src[r] : 0];
            const bool live = J.r_n[r] > 0 && sc.p != nullptr;
            J.r_p[r] = live ? sc.p + (size_t)J.r_c0[r] * 4 : nullptr;
            J.r_bs[r] = live ? sc.bstride : 0;
            J.r_prow[r] = live ? sc.prow : 0;
        }
    P.tiles_x = (P.W + 15) / 16; P.tiles_y = (P.H + 15) / 16; P.m_tiles = P.B * P.tiles_x * P.tiles_y;
    const int Mx = (P.m_tiles + 7) / 8;
    long long per_xcd = 0;
    for (int j = 0; j < P.njobs; ++j) per_xcd += (long long)Mx * P.job[j].n_tiles;
    if (per_xcd < 1) return hipSuccess;
    const unsigned grid = (unsigned)(per_xcd * 8);
    if (P.ks == 3) {   // the ConvLSTM step on small grids
        if (NT == 4) return launch_c5_t<4, 3>(P, grid, s);
        if (NT == 2) return launch_c5_t<2, 3>(P, grid, s);
        return hipErrorInvalidValue;
    }
    if (NT == 8) return launch_c5_t<8, 5>(P, grid, s);
    if (NT == 4) return launch_c5_t<4, 5>(P, grid, s);
    return launch_c5_t<2, 5>(P, grid, s);
}

}  // namespace vpx
