// cell2x.hip — the fused ConvLSTM step with FOUR waves per SIMD (round 6): cell2_kernel_q's half tile (16x16 pixels x 4 gates x 32
// channels, K = 32 steps pairing two taps of a 16-channel stage, same stage image, same weight pack, same ring of two chunks that
// turns over in halves) worked by EIGHT waves instead of four — wave w owns tile rows 4 (w & 3) .. + 3 and the N half w >> 2: the
// four column tiles {h, 2 + h, 4 + h, 6 + h} = the four gates of channels 16 h .. 16 h + 15 of the N tile. Restates
// conv_lstm_hzzone.py:59-68 / conv_lstm_ndrplz.py:31-41 exactly as cell2_kernel_q does: same operand split, same products in the same
// order per output element — bit-identical outputs (tests/test_gpu_cell2.py::test_x_form_*).
//
// Why (profiles/r05_cell2q_phase_trace.txt): on the four-wave half tile a SIMD holds two waves, one per workgroup;
// half of the time one of them is in its epilogue / prologue and the other issues an MFMA every 23.4 cycles instead of every 16 —
// its own copy requests (5.3 LDS-DMA pieces per step, ~100 cycles of issue each, in order with the MFMAs behind them) and its sync
// point stand in its way and nobody fills the pipe meanwhile. A wave tile of 64 pixels x 64 columns is 64 accumulator registers:
// the kernel fits 128 registers, so a SIMD holds FOUR waves (two per workgroup, two workgroups per CU as before):
//   * while one workgroup is in its epilogue the other still has two waves per SIMD in the main loop (the rate measured for "both
//     workgroups in the loop": 18.8 cycles per MFMA pair, 85 % of the pipe), and with both in the loop four;
//   * eight waves share a workgroup's copies: 2.7 pieces per wave and step instead of 5.3 (bf16x3), 1.2-1.4 instead of 2.7 (plain);
//   * the epilogue's transcendental work of a tile is spread over twice the waves.
// Price: every wave reads all four rows' activation fragments for half the columns — LDS fragment bytes per MFMA rise by a third
// (A 8 + B 8 reads per 48 MFMAs against 8 + 16 per 96): 85 B/clk/CU of the LDS's 256 at full MFMA rate.
//
// LDS (80 KiB, two workgroups per CU): two 24 KiB stage buffers + two 16 KiB weight chunks; the epilogue reuses it as 8 x 8 KiB of
// transposition space (two passes of two tile rows per wave).
#include "cell2_dev.h"

namespace vpx {

struct CellXEpi {
    ConvLSTMStepArgs a;
    char* h_sp;               // split h_t [B][HW][Ch] (or null)
    long long h_sp_bstride;   // bytes between batch items
};

// LDS-DMA with a wave-uniform 64-bit base (SGPR pair) and a 32-bit per-lane byte offset: no 64-bit vector address arithmetic, one
// VGPR per request (c2_dma16's notes on the inline asm and M0 apply). `lds` = LDS byte address of the wave's 1 KiB (an integer: casting a
// generic pointer to the LDS address space per request costs a null test of three scalar instructions each time).
__device__ __forceinline__ unsigned long long cx_uniform(const char* p) {
    // (the pointer IS wave-uniform; readfirstlane makes the compiler's divergence analysis agree, else it hands the asm a VGPR pair)
    const unsigned long long sb = reinterpret_cast<unsigned long long>(p);
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(sb >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)sb);
}
__device__ __forceinline__ void cx_dma16(const char* sbase, unsigned voff, unsigned lds) {
    asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(cx_uniform(sbase)), "{m0}"(__builtin_amdgcn_readfirstlane(lds)) : "memory");
}
// ... with a lane mask applied INSIDE the asm: the instruction is always issued (an all-zero mask included), so the wave's vmcnt
// sequence never depends on the data — behind a compiler-made `if (lane valid)` a wave whose 64 lanes are all outside the image skips
// the request (s_cbranch_execz), and every counted wait after it then allows one OLDER request to be in flight (round 6: the first form
// of this kernel did exactly that on tiles at the image's lower edge — intermittently wrong results at many tiles per CU)
__device__ __forceinline__ void cx_dma16_masked(const char* sbase, unsigned voff, unsigned lds, unsigned long long lanes) {
    unsigned long long keep;
    asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, %0"
                 : "=&s"(keep) : "v"(voff), "s"(cx_uniform(sbase)), "{m0}"(__builtin_amdgcn_readfirstlane(lds)), "s"(lanes) : "memory", "scc");
}
#ifdef VPX_ABLATE
// developer build only (make ablate): per-workgroup s_memtime stamps of wave 0 — start, first MFMA, loop end, end — and HW_ID | XCC_ID << 32
// (tools/trace_cell2x.py); vpx_dbg_cell2x_trace() reads them back. Never compiled into the product library.
__device__ unsigned long long cx_trace[8192 * 8];
#define CX_TRACE(k) do { if (wave == 0 && lane == 0 && L < 8192) cx_trace[L * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CX_TRACE(k) do { } while (0)
#endif
template <int N> __device__ __forceinline__ void cx_wait_vm_lgkm() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(N) : "memory"); }

// PLAIN: VPX_PREC_BF16 — hi parts only (one MFMA per product); the lo planes of a stage and the lo half of every weight piece are
// neither copied nor read. Its copies are split between the wave groups A (waves 0-3) and B (waves 4-7): a stage's hi planes are
// 768 pieces (A two per thread, B one), a half chunk's hi part 256 pieces (half 0: group A, half 1: group B).
// SPLIT — how the waves share the tile:
//   0  eight waves, COLUMN split: wave w owns tile rows 4 (w & 3) .. + 3 and the N half w >> 2 (four column tiles = four gates x 16 channels)
//   1  eight waves, ROW split: wave w owns tile rows 2 w, 2 w + 1 and all eight column tiles (A 2 x 2 + B 2 x 2 fragment registers instead
//      of 4 x 2 + 2 x 2: what lets the bf16x3 form fit 128 registers without spilling in the loop; price: every wave reads the whole
//      weight chunk, A 4 + B 16 fragment reads per 48 MFMAs)
// (A third form — four waves on cell2_kernel_q's own wave tile over a compact 40 KiB LDS image at 168 registers, THREE workgroups per CU,
//  plain bf16 only — was built and measured in round 6 and removed: three chains per CU took exactly the CU time of two)
template <bool PLAIN, int SPLIT>
__global__ __launch_bounds__(512, 4) void cell2_kernel_x(const Cell2Plan P, const CellXEpi E) {
    static_assert(SPLIT == 0 || SPLIT == 1, "column split | row split");
    constexpr bool MSPLIT = SPLIT != 0;            // the wave holds all eight column tiles (row-wise epilogue passes, full 128-byte lines)
    constexpr int NTH = 512;                       // threads (eight waves)
    constexpr int RW = SPLIT == 1 ? 2 : 4;         // tile rows per wave
    constexpr int NTW = MSPLIT ? 8 : 4;            // column tiles per wave
    constexpr int SYNC_LT = NTW / 2 - 1;           // the sync point sits before the weight read of the wave's first tile of chunk half 1
    using G = CQGeom<4>;   // the half tile's stage image: 18x18 halo positions, planes padded to 384 -> 24 KiB per stage
    constexpr int ABUFX = G::ABUF;                 // bytes of a stage buffer
    constexpr int WHALF = 8192;                    // bytes of a chunk half in LDS ([part][k group][64 columns][16 B])
    constexpr int WSLOT = 2 * WHALF;               // bytes of a ring slot
    constexpr bool GROUPS = PLAIN;                 // plain: the copies are split between the wave groups A (waves 0-3) and B (4-7)
    constexpr int NPC = GROUPS ? 2 : 3;            // stage-copy pieces per thread (GROUPS: piece 1 exists for group A only)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mrow0 = SPLIT == 1 ? 2 * wave : 4 * (wave & 3);   // first tile row
    const int nh = SPLIT == 0 ? wave >> 2 : 0;                  // N half (column split)
    const bool grpA = wave < 4;
    const int r16 = lane & 15, kg = lane >> 4;

    // XCD-aware tile decode (cell2_kernel_q's rule: N tile fastest, contiguous ranges per XCD)
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
    const int x0 = tx * 16, y0 = ty * 16;

    char* const Abuf = smem;
#ifdef VPX_ABLATE
    // TIMING-ONLY ablations of the developer build (garbage results): experiment bits 20 no waits for copies at the sync points, 21 no
    // barrier at the sync points, 22 no weight copies, 23 no stage copies, 24 no epilogue, 25 no fragment reads
    const int ab = P._q;
#else
    constexpr int ab = 0;
#endif
#ifdef VPX_ABLATE
    if (wave == 0 && lane == 0 && L < 8192) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        cx_trace[L * 8 + 4] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
    }
#endif
    CX_TRACE(0);

    // this thread's pieces of a stage copy: piece = tid + 512 u -> (plane = part * 2 + channel half, halo position). The plane of a piece
    // is wave-uniform (a plane is 384 = 6 x 64 pieces): its byte offset inside a split pixel row stays in a scalar register. Positions
    // outside the image are zeroed ONCE here, in both stage buffers, and their lanes sit out every copy (the DMA writes LDS per lane;
    // inimg[u] = the lanes inside the image, a scalar pair). The pixel offset of a piece is RECOMPUTED at each of its two uses per period
    // (a dozen vector instructions per 432 MFMAs) instead of living in a register through the loop: at 128 registers per wave the
    // compiler spilled exactly these values, and a reload from scratch memory waits on vmcnt(0) in the middle of the copy requests.
    const unsigned smem0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;   // LDS byte address of the dynamic segment
    unsigned long long inimg[NPC];
    int choff[NPC];
    auto piece_pix = [&](int u, bool& ok) {
        int t = tid;
        asm volatile("" : "+v"(t));   // (opaque: keeps the loop-invariant arithmetic below from being hoisted back into registers)
        const int plane = (wave * 64 + NTH * u) / G::PLANE_POS;   // wave-uniform
        const int pos = t + NTH * u - plane * G::PLANE_POS;
        const int hy = pos / C2_HALO_W, hx = pos - hy * C2_HALO_W;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        ok = pos < G::NPOS && gy >= 0 && gy < P.H && gx >= 0 && gx < P.W;
        return ok ? gy * P.W + gx : 0;
    };
#pragma unroll
    for (int u = 0; u < NPC; ++u) {
        const int plane = (wave * 64 + NTH * u) / G::PLANE_POS;
        bool ok;
        (void)piece_pix(u, ok);
        inimg[u] = __builtin_amdgcn_ballot_w64(ok);
        choff[u] = __builtin_amdgcn_readfirstlane((plane & 1) * 32 + (plane >> 1) * 16);   // pixel row: [8-channel group][hi 16 B | lo 16 B]
        if (!ok && (!GROUPS || u == 0 || grpA)) {
            const uint4 z = {0u, 0u, 0u, 0u};
            *reinterpret_cast<uint4*>(Abuf + (tid + NTH * u) * 16) = z;
            *reinterpret_cast<uint4*>(Abuf + ABUFX + (tid + NTH * u) * 16) = z;
        }
    }
    const unsigned dma_lds = smem0 + wave * 1024;   // LDS address of this wave's 64 pieces inside a 512-piece pass (buffer 0, pass 0)
    // weight pieces: a half chunk is [part][k group][64 columns][16 B] = 512 pieces (PLAIN: its hi part = the first 256, copied by one group)
    const char* const wtile = P.wpk + (size_t)n_tile * P.chunks_total * CQ_WCHUNK;   // wave-uniform
    const unsigned wvoff = (unsigned)((PLAIN ? (tid & 255) : tid) * 16);
    const unsigned wdma_lds = smem0 + 2 * ABUFX + (PLAIN ? (wave & 3) : wave) * 1024;

    const int nx = P.nx, S = P.nx + P.nh, Q = (9 * S + 1) / 2;
    const char* const xb = P.seg[0].sp + (size_t)b * P.seg[0].bstride;
    const char* const hb = P.seg[1].sp + (size_t)b * P.seg[1].bstride;
    const unsigned xrow = (unsigned)P.seg[0].C * 4u, hrow = (unsigned)P.seg[1].C * 4u;
    auto issue_A1 = [&](int s, int buf, int u) {
        const bool isx = s < nx;
        const char* base = isx ? xb + s * 64 : hb + (s - nx) * 64;   // 16 channels = 64 bytes of a split pixel row (wave-uniform)
        const unsigned prow = isx ? xrow : hrow;
        bool ok;
        const unsigned pix = (unsigned)piece_pix(u, ok);
        cx_dma16_masked(base + (unsigned)choff[u], pix * prow, dma_lds + buf * ABUFX + u * (NTH * 16), inimg[u]);
    };
    auto issue_A = [&](int s, int buf) {   // this wave's share of a stage copy
        issue_A1(s, buf, 0);
        if constexpr (GROUPS) { if (grpA) issue_A1(s, buf, 1); }
        else { issue_A1(s, buf, 1); issue_A1(s, buf, 2); }
    };
    auto issue_Wh = [&](int chunk, int slot, int half) {   // one 8 KiB half of a weight chunk (PLAIN: its 4 KiB hi part, by the half's group)
        if constexpr (GROUPS) { if (grpA != (half == 0)) return; }
        cx_dma16(wtile + (size_t)chunk * CQ_WCHUNK + half * 8192, wvoff, wdma_lds + slot * WSLOT + half * WHALF);
    };

    f32x4 acc[RW][NTW];   // [tile row m][local column tile]: global tile 2 lt + nh (column split: lt = gate) | lt (row split: gate lt >> 1)
#pragma unroll
    for (int m = 0; m < RW; ++m)
#pragma unroll
        for (int g = 0; g < NTW; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[m][g][r] = 0.0f;

    // fragment lane bases (cell2_kernel_q's, with the wave's row group and N half)
    const int a_lane = (kg & 1) * G::PLANE + (mrow0 * C2_HALO_W + r16) * 16;
    const int base1 = a_lane + (kg >> 1) * 16;                           // tap B one slot right of tap A
    const int base16 = a_lane + (kg >> 1) * 256;                         // tap B = (dy + 1, dx - 2): 16 slots further
    const int baseX = a_lane + (kg >> 1) * (ABUFX - cq_slot(8) * 16);    // cross step: tap 0 of the odd stage (buffer 1)
    int wb0 = 2 * ABUFX + kg * 1024 + r16 * 16 + nh * 256, wb1 = wb0 + WSLOT;
    bf16x8 ah[RW], al[RW], bh[2], bl[2];
    auto load_A1 = [&](int p, int m, int bx) {
        const int base = cq_kind(p) == 0 ? base1 : (cq_kind(p) == 1 ? base16 : bx);
        const char* a = smem + base + cq_aoff(p, ABUFX) + m * (C2_HALO_W * 16);
        ah[m] = *reinterpret_cast<const bf16x8*>(a);
        if constexpr (!PLAIN) al[m] = *reinterpret_cast<const bf16x8*>(a + 2 * G::PLANE);
    };
    auto load_B = [&](int p, int lt) {   // global tile nt = 2 lt + nh | lt: chunk half nt >> 2, 16-column group nt & 3 (nh * 256 sits in wb0 / wb1)
        const char* w = smem + ((p & 1) ? wb1 : wb0) + (MSPLIT ? (lt >> 2) * WHALF + (lt & 3) * 256 : (lt >> 1) * WHALF + (lt & 1) * 512);
        bh[lt & 1] = *reinterpret_cast<const bf16x8*>(w);
        if constexpr (!PLAIN) bl[lt & 1] = *reinterpret_cast<const bf16x8*>(w + 4096);
    };

    if (S > 0) {
        issue_A(0, 0);
        issue_Wh(0, 0, 0); issue_Wh(0, 0, 1);
        if (Q > 1) {
            issue_Wh(1, 1, 0);   // (group A in the plain form, every thread otherwise: the one request that may still fly)
            if constexpr (GROUPS) { if (grpA) cx_wait_vm_lgkm<1>(); else cx_wait_vm_lgkm<0>(); }
            else cx_wait_vm_lgkm<1>();
        } else cx_wait_vm_lgkm<0>();   // (lgkmcnt: the zeroing stores above)
        c2_barrier();
#pragma unroll
        for (int m = 0; m < RW; ++m) load_A1(0, m, a_lane);
        load_B(0, 0);
    }
    CX_TRACE(1);
    for (int s0 = 0; s0 < S; s0 += 2) {
        const bool odd = s0 + 1 < S;         // the period's odd stage exists
        const bool more = s0 + 2 < S;        // another period follows
        const int q0 = (s0 >> 1) * 9;
        const int par = (s0 >> 1) & 1;       // ring slot of period step 0
        const int bx = odd ? baseX : a_lane;  // without an odd stage the cross step's second half multiplies zero weights: read valid data
#pragma unroll
        for (int p = 0; p < 9; ++p) {
            if (p < 5 || odd) {
                const int q = q0 + p;
#pragma unroll
                for (int lt = 0; lt < NTW; ++lt) {
                    if (lt == SYNC_LT) {
                        // ---- sync point X_q, before this wave's first read of half 1 of chunk q: the fragments of its half-0 tiles are in
                        //      registers (lgkmcnt) before that half is given away; the stage copy issued one step ago may still fly ----
                        const bool stage_flies = (p == 1 && odd) || (p == 6 && more);
                        if (ab & (1 << 20)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        else if (stage_flies) {
                            if constexpr (GROUPS) { if (grpA) cx_wait_vm_lgkm<2>(); else cx_wait_vm_lgkm<1>(); }
                            else cx_wait_vm_lgkm<3>();
                        } else cx_wait_vm_lgkm<0>();
                        if (!(ab & (1 << 21))) c2_barrier();
                    }
                    // ---- weight fragments of the next column tile (after the last step: bytes nobody uses, cheaper than a branch) ----
                    if (lt < NTW - 1) load_B(p, lt + 1);
                    else load_B(p + 1, 0);
                    // ---- the MFMAs of column tile lt: three (plain: one) per tile row ----
                    __builtin_amdgcn_s_setprio(1);
                    // (the order of a product's three MFMAs — which operand two consecutive ones share — measured as nothing on the
                    //  power-bound loop: profiles/r06_cell2x_ab.txt, "order")
#pragma unroll
                    for (int m = 0; m < RW; ++m) {
                        f32x4 c = acc[m][lt];
                        if constexpr (!PLAIN) {
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[m], bh[lt & 1], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], bl[lt & 1], c, 0, 0, 0);
                        }
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], bh[lt & 1], c, 0, 0, 0);
                        acc[m][lt] = c;
                        if (lt == NTW - 1) load_A1(p == 8 ? 0 : p + 1, m, bx);   // the last column tile frees row m's fragments
                    }
                    __builtin_amdgcn_s_setprio(0);
                    // ---- this sync point's copies, behind the MFMAs of the following column tiles: the weights first, then the stage ----
                    if (lt == SYNC_LT) {
                        const int sl = (p & 1) ^ par;   // ring slot of chunk q (and q + 2)
                        if (!(ab & (1 << 22))) {
                            if (q + 1 < Q) issue_Wh(q + 1, sl ^ 1, 1);
                            if (q + 2 < Q) issue_Wh(q + 2, sl, 0);
                        }
                    }
                    if (lt == SYNC_LT + 1) {
                        if (!(ab & (1 << 23))) {
                            if (p == 0 && odd) issue_A(s0 + 1, 1);
                            if (p == 5 && more) issue_A(s0 + 2, 0);
                        }
                    }
                }
            }
        }
        { const int t = wb0; wb0 = wb1; wb1 = t; }
    }

    // ---- epilogue (conv_lstm_hzzone.py:62-68): the wave's 64 pixels x (4 gates x 16 channels) go through its private 8 KiB of LDS in
    //      two passes of two tile rows, image [gate][pixel 32][16 channels]; a lane then owns FOUR channels of a pixel for all four
    //      gates: 16-byte global accesses, four lanes = one pixel's 64-byte channel run ----
    const ConvLSTMStepArgs& a = E.a;
    CX_TRACE(2);
#ifdef VPX_ABLATE
    if (ab & (1 << 24)) {   // timing only: no epilogue (one never-taken store keeps the accumulators alive)
        float t = 0.f;
        for (int m = 0; m < RW; ++m) for (int nt = 0; nt < NTW; ++nt) for (int r = 0; r < 4; ++r) t += acc[m][nt][r];
        if (t == 1.2345e-30f) a.c_out[0] = t;
        CX_TRACE(3);
        return;
    }
#endif
    float* const lx = reinterpret_cast<float*>(smem + wave * 8192);
    // lane -> (four channels cg, pixel pp) of a pass: row split — a pass is ONE tile row x 32 channels (8 lanes = a pixel's whole 128-byte
    // channel run: every global access a full line); column split — two tile rows x the wave's 16 channels (4 lanes = 64 bytes)
    const int cg = MSPLIT ? (lane & 7) : (lane & 3), pp = MSPLIT ? (lane >> 3) : (lane >> 2);
    constexpr int PCH = MSPLIT ? 32 : 16;          // channels of a pass
    constexpr int PPI = MSPLIT ? 8 : 16;           // pixels per iteration of a pass (64 lanes / lanes per pixel)
    const unsigned Ch = (unsigned)a.Ch;
    const unsigned ch = (unsigned)(n_tile * 32 + (MSPLIT ? 0 : nh * 16) + cg * 4);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const size_t img = (size_t)b * P.H * P.W;
    const float* const cin_b = a.c_in ? a.c_in + img * Ch : nullptr;
    float* const cout_b = a.c_out + img * Ch;
    float* const hout_b = a.h_out ? a.h_out + (size_t)b * a.h_bstride : nullptr;
    float* const g0 = a.gates ? a.gates + img * 4 * Ch : nullptr;
    char* const hsp_b = E.h_sp ? E.h_sp + (size_t)b * E.h_sp_bstride : nullptr;
    // Four iterations k = 2 * pass + it of 16 (row split: 8) pixels x 4 channels per lane. The global operands of iteration k + 1 (cell state,
    // three peepholes: 64 bytes per lane) are requested BEFORE the arithmetic of iteration k, those of iteration 0 before the barrier and the
    // first LDS round trip: issued right before their use (the first form of this epilogue) every iteration exposed a whole HBM / L2 round
    // trip — 21-22 k cycles per tile of which the arithmetic is 7 k (developer build stamps, profiles/r06_cell2x_trace.txt).
    struct EIn { unsigned eo; f32x4 cp, wi, wf, wo; };
    auto eload = [&](int k, EIn& v) {
        const int ps = k >> 1, it = k & 1;
        const int pix = MSPLIT ? (y0 + mrow0 + ps) * P.W + x0 + it * 8 + pp : (y0 + mrow0 + 2 * ps + it) * P.W + x0 + pp;
        v.eo = __umul24((unsigned)pix, Ch) + ch;
        v.cp = cin_b ? *reinterpret_cast<const f32x4*>(cin_b + v.eo) : zero;
        v.wi = a.wci ? *reinterpret_cast<const f32x4*>(a.wci + v.eo) : zero;
        v.wf = a.wci ? *reinterpret_cast<const f32x4*>(a.wcf + v.eo) : zero;
        v.wo = a.wco ? *reinterpret_cast<const f32x4*>(a.wco + v.eo) : zero;
    };
    // pass ps: column split — tile rows 2 ps, 2 ps + 1 of the wave's four, channel half nh; row split — tile row ps of the wave's two, all 32
    // channels. LDS image [gate][pixel][PCH channels] = 8 KiB either way. Bias: a lane's accumulator column is ONE channel of every gate —
    // added on the way into LDS.
    auto put = [&](int ps) {
        if constexpr (MSPLIT) {
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
                const float bq = a.bias ? a.bias[a.gate_pos[nt >> 1] * Ch + (unsigned)(n_tile * 32 + (nt & 1) * 16 + r16)] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) lx[(nt >> 1) * 512 + (4 * kg + r) * 32 + (nt & 1) * 16 + r16] = acc[ps][nt][r] + bq;
            }
        } else {
            float bq[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.bias) {
#pragma unroll
                for (int g = 0; g < 4; ++g) bq[g] = a.bias[a.gate_pos[g] * Ch + (unsigned)(n_tile * 32 + nh * 16 + r16)];
            }
#pragma unroll
            for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) lx[g * 512 + (mm * 16 + 4 * kg + r) * 16 + r16] = acc[2 * ps + mm][g][r] + bq[g];
        }
    };
    auto emath = [&](int k, const EIn& v) {
        const int it = k & 1;
        const unsigned eo = v.eo;
        const float* row = lx + (it * PPI + pp) * PCH + cg * 4;
        const f32x4 ai = *reinterpret_cast<const f32x4*>(row);
        const f32x4 af = *reinterpret_cast<const f32x4*>(row + 512);
        const f32x4 ag = *reinterpret_cast<const f32x4*>(row + 1024);
        const f32x4 ao = *reinterpret_cast<const f32x4*>(row + 1536);
        f32x4 i4, f4, g4, o4, cn, hn;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float c0 = v.cp[e];
            i4[e] = sigmoid_f(ai[e] + v.wi[e] * c0);   // (ai .. ao carry their bias)
            f4[e] = sigmoid_f(af[e] + v.wf[e] * c0);
            g4[e] = tanh_f(ag[e]);
            cn[e] = lstm_c(f4[e], c0, i4[e], g4[e]);
            o4[e] = sigmoid_f(ao[e] + v.wo[e] * cn[e]);
            hn[e] = o4[e] * tanh_f(cn[e]);
        }
        *reinterpret_cast<f32x4*>(cout_b + eo) = cn;
        if (hout_b) *reinterpret_cast<f32x4*>(hout_b + eo) = hn;   // (null: the consumer reads the split copy below — VPX_FLAG_OUT_SPLIT)
        if (g0) {
            const unsigned go = 4u * (eo - ch) + ch;
            *reinterpret_cast<f32x4*>(g0 + go) = i4;
            *reinterpret_cast<f32x4*>(g0 + go + Ch) = f4;
            *reinterpret_cast<f32x4*>(g0 + go + 2 * Ch) = g4;
            *reinterpret_cast<f32x4*>(g0 + go + 3 * Ch) = o4;
        }
        if (hsp_b) {
            // the lane pair (cg even, cg odd) holds one 8-channel group = 32 bytes [8 hi | 8 lo]: the even lane stores the 16 hi bytes,
            // the odd lane the 16 lo bytes (Cell2Epi::vec_math's exchange: quad_perm [1,0,3,2])
            unsigned h[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) c2_split(hn[e], h[e], l[e]);
            const bool oddl = (cg & 1) != 0;
            const unsigned h0 = h[0] | (h[1] << 16), h1 = h[2] | (h[3] << 16), l0 = l[0] | (l[1] << 16), l1 = l[2] | (l[3] << 16);
            const unsigned s0 = oddl ? h0 : l0, s1 = oddl ? h1 : l1;   // what the partner stores of mine
            const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s0, 0xB1, 0xF, 0xF, true);
            const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s1, 0xB1, 0xF, 0xF, true);
            char* dst = hsp_b + 4u * (eo - ch) + (ch >> 3) * 32 + (oddl ? 16 : 0);
            *reinterpret_cast<uint4*>(dst) = oddl ? uint4{r0, r1, l0, l1} : uint4{h0, h1, r0, r1};
        }
    };
    constexpr int NPASS = MSPLIT ? RW : 2;   // passes: one tile row each (all 32 channels) | two tile rows each (the wave's 16 channels)
    EIn v[2];
    eload(0, v[0]);
    c2_barrier();   // every wave has read its last fragments: the staging buffers become the transposition space
    // (LDS operations of one wave execute in order: the reads of emath see put's writes without a barrier, and the next put's writes come
    //  after the reads of the pass before it)
#pragma unroll
    for (int k = 0; k < 2 * NPASS; ++k) {
        if ((k & 1) == 0) put(k >> 1);
        if (k + 1 < 2 * NPASS) eload(k + 1, v[(k + 1) & 1]);
        emath(k, v[k & 1]);
    }
#ifdef VPX_ABLATE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (how long the epilogue's stores take to complete)
#endif
    CX_TRACE(3);
}

// the fused cell step on the eight-wave half tile; the caller (launch_cell2, cell2.hip) has checked that the q form applies (maps in
// whole 16x16 tiles, whole 32-channel N tiles) and set tiles_y / grid_m for 16-row tiles
hipError_t launch_cell2x(const Cell2Plan& p_in, const ConvLSTMStepArgs& ea, void* h_sp, long long h_sp_bstride, hipStream_t s) {
    // split: 0 = columns, 1 = rows. Default: rows for both precisions (bf16x3: the column split spills fragment registers inside the loop at
    // 128 registers; plain bf16: the row split's full-line epilogue measured faster, profiles/r06_cell2x_ab.txt). VPX_OPT_EXPERIMENT bit 16 = columns.
    const int split = (g_experiment & 65536) ? 0 : 1;
    constexpr int LDS8 = CQGeom<4>::LDS;   // 80 KiB: two workgroups per CU
    static bool attr_set = false;
    if (!attr_set) {
        const void* fn[4] = {reinterpret_cast<const void*>(&cell2_kernel_x<false, 0>), reinterpret_cast<const void*>(&cell2_kernel_x<false, 1>),
                             reinterpret_cast<const void*>(&cell2_kernel_x<true, 0>), reinterpret_cast<const void*>(&cell2_kernel_x<true, 1>)};
        for (int i = 0; i < 4; ++i) {
            const hipError_t e = vpx_func_attr(fn[i], hipFuncAttributeMaxDynamicSharedMemorySize, LDS8);
            if (e != hipSuccess) return e;
        }
        attr_set = !g_dry_run;
    }
    const CellXEpi epi{ea, reinterpret_cast<char*>(h_sp), h_sp_bstride};
    Cell2Plan p = p_in;
    p._q = g_experiment;   // (read by the developer build's timing ablations only)
    const long long per_xcd = ((long long)p.grid_m * p.n_tiles + 7) / 8;
    const dim3 grid((unsigned)(per_xcd * 8));
    if (p.plain) {
        if (split == 1) VPX_LAUNCH((cell2_kernel_x<true, 1>), grid, dim3(512), LDS8, s, p, epi);
        else VPX_LAUNCH((cell2_kernel_x<true, 0>), grid, dim3(512), LDS8, s, p, epi);
    } else {
        if (split == 1) VPX_LAUNCH((cell2_kernel_x<false, 1>), grid, dim3(512), LDS8, s, p, epi);
        else VPX_LAUNCH((cell2_kernel_x<false, 0>), grid, dim3(512), LDS8, s, p, epi);
    }
    return vpx_hip_last_error();
}

}  // namespace vpx

#ifdef VPX_ABLATE
extern "C" int vpx_dbg_cell2x_trace(unsigned long long* out65536) {
    return (int)hipMemcpyFromSymbol(out65536, HIP_SYMBOL(vpx::cx_trace), sizeof(unsigned long long) * 8192 * 8);
}
#endif
