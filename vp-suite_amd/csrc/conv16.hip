// conv16.hip — c16: 3x3 'same' convolution of a split-format source with 16..64 channels into SIXTEEN output channels (round 6): the last
// glue layer of convlstm-shi's forecaster (ef_conv_lstm.py:36-65: `deconv3_leaky_1`, ConvTranspose2d(64, 16, 3, 1, 1) + LeakyReLU(0.2) on
// 1 280 frames of 64x64 per forward at B = 128). With 16 output columns an MFMA tile has nothing to amortise a weight stream or a 16-channel
// stage pipeline over: on the first-generation kernel the layer took 0.83 ms (on convq's 16-column form 1.22 ms). Here the WHOLE K of a
// pixel tile is resident:
//   * the layer's weights — at most 18 K = 32 steps x (hi, lo) x 16 columns = 36 KiB — are MFMA fragments in REGISTERS (144 per lane) for the
//     life of a persistent workgroup (one per CU, eight waves); they are the MFMA's ROW operand, so that a lane ends up with four consecutive
//     output channels of one pixel (one 16-byte store);
//   * a tile is 8 rows x 16 pixels; its 10x18 halo of ALL channels (180 positions x 256 B = 45 KiB) lands in one of three LDS buffers by
//     LDS-DMA (two copies in flight) while the third is multiplied: one counted wait + one barrier per TILE, nothing inside the K loop but
//     fragment reads (two per three MFMAs), MFMAs and the next copy's instructions, one at a time.
// Measured (tools/ab_glue.py, tools/ab_c16.py; 1 280 frames): 0.41 ms = 0.29 PFLOP/s algorithmic (x3: 0.71 PF of bf16 MFMA work). The parts,
// switched off one at a time in the developer build: MFMAs + fragment reads alone 222 us — 1.31 PF of bf16 MFMA work, the power ceiling
// DESIGN.md §3.8 measured on the fused cell; tile copies alone 200 us (1.34 GB + halo: HBM rate); stores 45 us; the bare tile loop 65 us.
// The whole is still nearer their sum than their maximum: what remains is overlap, not a shorter part.
// K = 32 steps pair two taps of a 16-channel stage exactly as cell2_kernel_q does (cq_tap_of / cq_stage_of, cell2_dev.h).
#include "cell2_dev.h"
#include "vpx_host.h"

namespace vpx {

constexpr int C16_NPOS = 180;                       // 10 x 18 halo positions of a tile
constexpr int C16_BUF = C16_NPOS * 256;             // tile buffer: [position][16 pieces of 16 B] — a pixel's channel run as it lies in memory
constexpr int C16_LDS = 3 * C16_BUF;                // 135 KiB: one workgroup per CU, two tile copies in flight under the third's MFMAs
// Where the 16 pieces of a pixel (16-channel stage s, channel half kh, part hi / lo; in memory at 64 s + 32 kh + 16 part) sit in the
// position's 256 bytes: slot = 8 kh + ((2 s + part) ^ (position & 7)). A ds_read_b128 is served in four lane groups, {0-3, 12-15, 20-27} etc.
// (MI355X_MICROARCH.md, LDS): k group 0's columns 0-3 and 12-15 together with k group 1's columns 4-11 — eight positions that differ
// mod 8 in the low half of the 256 bytes, eight in the high half: 16 different bank groups whatever the tap's offset. (An XOR with
// position & 15 over all 16 slots, the first attempt, put the two channel halves of a group two slots apart: two-way conflicts on half
// the taps, fragment reads at half rate — they, not the MFMAs, set the pace.) The copy pays nothing for the permutation: an LDS-DMA lane
// lands at base + 16 * lane whatever its SOURCE address is. Consecutive lanes still walk a pixel's own contiguous 256 bytes: a copy
// instruction touches 8 cache lines, not the 64 of position-major planes (the fused cell's staging; tried first here: 0.55 ms).
__device__ __forceinline__ int c16_addr(int pos, int stage, int khalf) { return pos * 256 + khalf * 128 + (((2 * stage) ^ (pos & 7)) << 4); }

// pack: [step q][part][k group][16 output channels][8 bf16]; element (out channel oc, in channel c, tap t) of the source at
// w[oc * s_oc + c * s_ic + t'] with t' = 8 - t for a transposed layer (correlation with the flipped kernel)
__global__ void c16_pack_kernel(const float* __restrict__ w, long long s_oc, long long s_ic, int S, int Q, int flip, char* __restrict__ dst) {
    const int total = Q * 1024;   // bf16 elements
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int i = e & 7, oc = (e >> 3) & 15, kg = (e >> 7) & 3, part = (e >> 9) & 1, q = e >> 10;
        const int p = q % 9, tsel = kg >> 1, khalf = kg & 1;
        const int stage = 2 * (q / 9) + cq_stage_of(p, tsel), t = cq_tap_of(p, tsel);
        float v = 0.0f;
        if (stage < S) v = w[(long long)oc * s_oc + (long long)(stage * 16 + khalf * 8 + i) * s_ic + (flip ? 8 - t : t)];
        unsigned hi, lo;
        c2_split(v, hi, lo);
        reinterpret_cast<unsigned short*>(dst)[e] = (unsigned short)(part ? lo : hi);
    }
}

struct C16Args {
    const char* x; long long x_bstride, x_tstride; int x_nT;   // split source [N][H][W][C]; image n at (n / x_nT) * x_bstride + (n % x_nT) * x_tstride
    const char* wpk; const float* bias;   // packed weights; bias [16] or null
    float* y;                             // fp32 NHWC [N][H][W][16], or null
    char* y_split;                        // the same in the split format (64 B per pixel), or null
    int N, H, W, C, tiles_x, tiles_y;
    long long ntiles;
    float slope;                          // LeakyReLU slope (0: none ... applied when act != 0)
    int act;
    int dbg;                              // developer build (VPX_ABLATE) only: timing ablations, VPX_OPT_EXPERIMENT bits 20-22
};
#ifdef VPX_ABLATE
#define C16_DBG(bit) (a.dbg & (bit))
#else
#define C16_DBG(bit) 0
#endif

template <int S>   // 16-channel stages: C = 16 S
__global__ __launch_bounds__(512, 2) void c16_kernel(const C16Args a) {
    constexpr int Q = (9 * S + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kg = lane >> 4;

    // ---- weights: this lane's row fragments of every step (row = output channel r16, k group kg), hi and lo ----
    bf16x8 wh[Q], wl[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const char* p = a.wpk + q * 2048 + (kg * 16 + r16) * 16;
        wh[q] = *reinterpret_cast<const bf16x8*>(p);
        wl[q] = *reinterpret_cast<const bf16x8*>(p + 1024);
    }
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + 4 * kg);
    // The loads above are COMPLETE, as far as the compiler knows, before the tile loop: the loop's LDS-DMA and its counted waits are inline
    // asm, and a load still pending in the compiler's books could get an s_waitcnt vmcnt(0) inside the loop — which would wait for the
    // copies in flight as well. Using the registers in an empty asm settles it.
#pragma unroll
    for (int q = 0; q < Q; ++q) asm volatile("" : "+v"(wh[q]), "+v"(wl[q]));
    asm volatile("" : "+v"(bias4));

    // ---- this thread's six pieces of a tile copy: buffer slot q = tid + 512 u -> (halo position, source piece); tile-independent parts ----
    int p_hy[6], p_hx[6], p_coff[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int q = tid + 512 * u, pos = q >> 4, kh = (q >> 3) & 1, sp = (q & 7) ^ (pos & 7);   // sp = 2 stage + part
        p_hy[u] = pos / 18;
        p_hx[u] = pos % 18;
        p_coff[u] = (sp >> 1) < S ? (sp >> 1) * 64 + kh * 32 + (sp & 1) * 16 : -1;   // (a stage the layer lacks: zeros — finite data for zero weights)
    }
    const unsigned rowpix = (unsigned)a.C * 4u;
    const unsigned tpi = (unsigned)(a.tiles_x * a.tiles_y), ntiles = (unsigned)a.ntiles;
    struct Where { const char* img; int n, y0, x0; };   // wave-uniform: source image, image index, the tile's first row / column
    auto locate = [&](unsigned tile) {
        const unsigned n = tile / tpi, tr = tile - n * tpi, ty = tr / (unsigned)a.tiles_x, tx = tr - ty * (unsigned)a.tiles_x;
        Where w;
        w.img = a.x + (size_t)(n / (unsigned)a.x_nT) * a.x_bstride + (size_t)(n % (unsigned)a.x_nT) * a.x_tstride;
        w.n = (int)n; w.y0 = (int)ty * 8; w.x0 = (int)tx * 16;
        return w;
    };
    auto issue_piece = [&](int u, const Where& w, int boff) {
        if (wave + 8 * u >= C16_NPOS * 16 / 64) return;   // (wave-uniform: past the 180 positions)
        const int gy = w.y0 - 1 + p_hy[u], gx = w.x0 - 1 + p_hx[u];
        const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && p_coff[u] >= 0;
        const char* src = ok ? w.img + (size_t)((unsigned)(gy * a.W + gx) * (unsigned long long)rowpix) + p_coff[u]
                             : reinterpret_cast<const char*>(c2_zero16);
        c2_dma16(src, smem + boff + wave * 1024 + u * 8192);
    };
    auto issue = [&](unsigned tile, int boff) {
        const Where w = locate(tile);
#pragma unroll
        for (int u = 0; u < 6; ++u) issue_piece(u, w, boff);
    };

    // fragment lane constants: pixel = (tile row `wave`, column r16); k group = tap half * 2 + channel half
    const int pos0 = wave * 18 + r16, tsel = kg >> 1, khalf = kg & 1;

    typedef __attribute__((address_space(3))) const char lds_char;
    typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8;
    lds_char* const L0 = (lds_char*)smem;
    const unsigned G = gridDim.x;
    unsigned tile = blockIdx.x;
    int b0 = 0, b1 = C16_BUF, b2 = 2 * C16_BUF;   // ring of three tile buffers: multiplied now | landing | being requested
    if (tile < ntiles) issue(tile, b0);
    if (tile + G < ntiles && !C16_DBG(2)) issue(tile + G, b1);
    for (; tile < ntiles; tile += G) {
        // this thread's pieces of `tile` have landed: everything but the newest batch (6 copies of waves 0-4, 5 of waves 5-7; loads
        // complete in order, so a pending store of the previous tile can only make this wait stricter)
        if (tile + G < ntiles && !C16_DBG(2)) { if (wave < 5) C2_WAIT_VM(6); else C2_WAIT_VM(5); }
        else C2_WAIT_VM(0);
        c2_barrier();      // ... everybody's have; and every wave has finished reading the buffer of the tile before
        // The copy of the tile after next goes out one instruction at a time BETWEEN the K steps: issued as a burst, the 45 KiB fill the CU's
        // miss queue and every wave sits in its next copy instruction for about the transfer time before it multiplies anything
        // (measured: copies alone 180 us + MFMAs alone 211 us = the whole kernel, nothing overlapped; tools/ab_c16.py)
        const bool more = tile + 2 * G < ntiles && !C16_DBG(2);
        Where w2{};
        if (more) w2 = locate(tile + 2 * G);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (!C16_DBG(1)) {
#pragma unroll
        for (int s0 = 0; s0 < S; s0 += 2) {
            const bool pair = s0 + 1 < S;
#pragma unroll
            for (int p = 0; p < 9; ++p) {
                if (p >= 5 && !pair) continue;   // (odd S: the last period is a lone stage; step 4's second tap half meets zero weights)
                const int q = (s0 >> 1) * 9 + p;
#pragma unroll
                for (int u = 0; u < 6; ++u)
                    if ((u * Q) / 6 == q && more) issue_piece(u, w2, b2);
                const int pos = pos0 + (tsel ? cq_slot(cq_tap_of(p, 1)) : cq_slot(cq_tap_of(p, 0)));
                const int stage = s0 + (tsel ? cq_stage_of(p, 1) : cq_stage_of(p, 0));
                const int f = b0 + c16_addr(pos, stage, khalf);   // (LDS-typed pointers: an integer detour through a generic pointer makes flat loads)
                const bf16x8 ph = *reinterpret_cast<const lds_bf16x8*>(L0 + f);
                const bf16x8 pl = *reinterpret_cast<const lds_bf16x8*>(L0 + (f ^ 16));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[q], pl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[q], ph, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[q], ph, acc, 0, 0, 0);
            }
        }
        } else if (more) {
#pragma unroll
            for (int u = 0; u < 6; ++u) issue_piece(u, w2, b2);
        }
        // ---- epilogue: lane = (pixel r16 of tile row `wave`, output channels 4 kg .. + 3) ----
        const Where w0 = locate(tile);
        const int n = w0.n, y = w0.y0 + wave, x = w0.x0 + r16;
        if (y < a.H && x < a.W && !C16_DBG(4)) {
            f32x4 v = acc + bias4;
            if (a.act) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : v[r] * a.slope;
            }
            const size_t pix = ((size_t)n * a.H + y) * a.W + x;
            if (a.y) *reinterpret_cast<f32x4*>(a.y + pix * 16 + 4 * kg) = v;
            if (a.y_split) {   // channel group kg >> 1 of the pixel: 16 B hi | 16 B lo; this lane's four channels are one half of each
                unsigned h[4], l[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) c2_split(v[r], h[r], l[r]);
                char* const o = a.y_split + pix * 64 + (kg >> 1) * 32 + (kg & 1) * 8;
                *reinterpret_cast<uint2*>(o) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
                *reinterpret_cast<uint2*>(o + 16) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
            }
        }
        const int t = b0; b0 = b1; b1 = b2; b2 = t;
    }
}

// the layers this kernel takes: bf16x3, 3x3, stride 1, pad 1 (plain or transposed), 16 output channels, 16 / 32 / 48 / 64 input channels
bool c16_applicable(const vpx_conv_desc* d) {
    if (g_experiment & (1 << 28)) return false;   // VPX_OPT_EXPERIMENT bit 28: the first-generation kernel (A/B runs, tests)
    return d->precision == VPX_PREC_BF16X3 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->Co == 16 &&
           (d->Ci & 15) == 0 && d->Ci >= 16 && d->Ci <= 64;
}
size_t c16_wpk_bytes(const vpx_conv_desc* d) { return (size_t)((9 * (d->Ci / 16) + 1) / 2) * 2048; }

int c16_forward(const vpx_conv_desc* d, const char* x_split, long long x_bstride, long long x_tstride, int x_nT, const float* w,
                const float* bias, float* y, char* y_split, char* wpk, bool packed, hipStream_t s) {
    const int S = d->Ci / 16, Q = (9 * S + 1) / 2;
    if (!packed) {
        if (!ws_write_ok(wpk, (size_t)Q * 2048, "weight pack (c16_pack_kernel)")) { set_error("%s", ws_violation()); return VPX_ERR_WORKSPACE; }
        // reference layouts: Conv2d [Co, Ci, 3, 3]; ConvTranspose2d [Ci, Co, 3, 3] (flipped taps)
        const long long s_oc = d->transposed ? 9 : (long long)d->Ci * 9, s_ic = d->transposed ? (long long)d->Co * 9 : 9;
        VPX_LAUNCH(c16_pack_kernel, dim3((Q * 1024 + 255) / 256), dim3(256), 0, s, w, s_oc, s_ic, S, Q, d->transposed ? 1 : 0, wpk);
        VPX_CHECK_HIP(vpx_hip_last_error());
    }
    C16Args a{};
    a.x = x_split; a.x_bstride = x_bstride; a.x_tstride = x_tstride; a.x_nT = x_nT > 1 ? x_nT : 1;
    a.wpk = wpk; a.bias = bias; a.y = y; a.y_split = y_split;
    a.N = d->N; a.H = d->H; a.W = d->W; a.C = d->Ci;
    a.tiles_x = (d->W + 15) / 16; a.tiles_y = (d->H + 7) / 8;
    a.ntiles = (long long)d->N * a.tiles_x * a.tiles_y;
    if (a.ntiles >= (1ll << 31)) { set_error("c16: too many tiles"); return VPX_ERR_UNSUPPORTED; }
    a.slope = d->leaky_slope; a.act = d->leaky_slope != 0.0f ? 1 : 0;
    a.dbg = (g_experiment >> 20) & 7;
    static bool attr_set = false;
    if (!attr_set) {
        const void* fn[4] = {reinterpret_cast<const void*>(&c16_kernel<1>), reinterpret_cast<const void*>(&c16_kernel<2>),
                             reinterpret_cast<const void*>(&c16_kernel<3>), reinterpret_cast<const void*>(&c16_kernel<4>)};
        for (int i = 0; i < 4; ++i) {
            const hipError_t e = vpx_func_attr(fn[i], hipFuncAttributeMaxDynamicSharedMemorySize, C16_LDS);
            if (e != hipSuccess) { set_error("c16: hipFuncSetAttribute failed"); return VPX_ERR_LAUNCH; }
        }
        attr_set = !g_dry_run;
    }
    const unsigned grid = (unsigned)(a.ntiles < 256 ? a.ntiles : 256);   // one persistent workgroup per CU
    if (S == 1) VPX_LAUNCH(c16_kernel<1>, dim3(grid), dim3(512), C16_LDS, s, a);
    else if (S == 2) VPX_LAUNCH(c16_kernel<2>, dim3(grid), dim3(512), C16_LDS, s, a);
    else if (S == 3) VPX_LAUNCH(c16_kernel<3>, dim3(grid), dim3(512), C16_LDS, s, a);
    else VPX_LAUNCH(c16_kernel<4>, dim3(grid), dim3(512), C16_LDS, s, a);
    VPX_CHECK_HIP(vpx_hip_last_error());
    return VPX_OK;
}

}  // namespace vpx
