// cell2.hip — second-generation fused ConvLSTM step for gfx950: split-bf16 ("bf16x3") arithmetic on PRE-SPLIT operands,
// one 8-wave workgroup per CU owning a 32x16-pixel tile x (4 gates x 32 channels), every operand byte moved
// HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4) with the next stage in flight across the chunk barriers.
// Computes exactly what conv_gemm_kernel<EpiConvLSTM, bf16x3> computes (same operand split, same accumulation order:
// outputs are identical up to fp32 summation order); restates conv_lstm_hzzone.py:59-68 / conv_lstm_ndrplz.py:31-41 like it.
//
// Why a second kernel (measured on the first one): with 256-pixel tiles every CU pulled 3 MB of packed
// weights per launch through its load path (43 us of a 243 us launch), each workgroup converted its activation halo
// fp32 -> (hi, lo) itself (n_tiles x halo overlap = 2.5-3.8x redundant VALU work) and the staging phases of the two
// co-resident workgroups hardly overlapped the other's MFMA phases (MFMA-only 160 us + everything-else 126 us -> 243 us).
// Here: 512 pixels share every weight chunk (half the traffic), operands arrive split (staging is a pure copy, no VALU,
// no registers), and the copy of stage s+1 / chunk c+2 runs under the MFMAs of stage s / chunk c.
//
// Split tensor format ("sp"): per pixel, per group of 8 channels: 8 hi bf16 (16 B) then 8 lo bf16 (16 B); a pixel row is
// C*4 bytes like the fp32 row it replaces. Producers: split_convert_kernel (block input x, initial state h0) and the
// cell epilogue itself (h_t for step t+1 — and for the weight-gradient kernel, which contracts the same operands).
//
// LDS images are "lane linear": a 16-byte piece lands at base + piece*16, and every fragment read of the MFMA loop
// is base(lane) + immediate with consecutive lanes on consecutive 16-byte slots -> conflict-free without padding:
//   activation stage (16 channels): [plane = part*2 + khalf][pos (34x18 halo, padded to 640)][16 B]   40 KiB, x2 buffers
//   weight chunk (one tap row = 3 k-steps of 16): [q = dx][part][khalf][n = gate*32 + j][16 B]       24 KiB, x3 ring
#include <stdlib.h>
#include <type_traits>

#include "cell2_dev.h"

namespace vpx {

// ---------------------------------------------------------------------------------------------------------------
// fp32 NHWC [npix][C] -> split format. One thread per (pixel, 8-channel group). HBM-bound, 8 B per element.
__global__ __launch_bounds__(256) void split_convert_kernel(const float* __restrict__ src, char* __restrict__ dst,
                                                            long long ngroups, int gpp /* groups per pixel = C/8 */) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= ngroups) return;
    const f32x4 a = *reinterpret_cast<const f32x4*>(src + idx * 8);
    const f32x4 b = *reinterpret_cast<const f32x4*>(src + idx * 8 + 4);
    unsigned h[8], l[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { c2_split(a[i], h[i], l[i]); c2_split(b[i], h[4 + i], l[4 + i]); }
    uint4 hv = {h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
    uint4 lv = {l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
    *reinterpret_cast<uint4*>(dst + idx * 32) = hv;
    *reinterpret_cast<uint4*>(dst + idx * 32 + 16) = lv;
    (void)gpp;
}

hipError_t launch_split_convert(const float* src, void* dst, long long npix, int C, hipStream_t s) {
    const long long ngroups = npix * (C / 8);
    if (!ws_write_ok(dst, (size_t)ngroups * 32, "operand conversion (split_convert_kernel)")) return hipErrorInvalidValue;
    const long long blocks = (ngroups + 255) / 256;
    VPX_LAUNCH(split_convert_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, reinterpret_cast<char*>(dst), ngroups, C / 8);
    return vpx_hip_last_error();
}

// ---------------------------------------------------------------------------------------------------------------
// weight repack: reference OIHW [4Ch, Cin+Ch, 3, 3] -> [n_tile][chunk = stage*3 + dy][q = dx][part][khalf][n][8 bf16]
// step p (0..8) of a two-stage period, lane half tsel (k groups 0,1 | 2,3): which stage of the period (0 even, 1 odd) and tap
// (cq_stage_of / cq_tap_of — the step schedule of a two-stage period — live in cell2_dev.h: cell2x.hip shares them)

// q form: [n_tile][step q][half = n >> 6][part][k group][n & 63][8 bf16] over the present stage sequence pk.stage_col[0 .. S-1]
// (a chunk is two 8 KiB halves, one per four column tiles: the half-tile kernel's weight ring turns over in halves)
__global__ void cell2_pack_q_kernel(const Cell2Pack pk, char* __restrict__ dst) {
    const long long total = (long long)pk.n_tiles * pk.chunks_total * (CQ_WCHUNK / 2);  // bf16 elements
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(e & 7);
        long long r = e >> 3;
        int n = (int)(r & 63); r >>= 6;
        const int kg = (int)(r & 3); r >>= 2;
        const int part = (int)(r & 1); r >>= 1;
        n += (int)(r & 1) * 64; r >>= 1;                         // half chunk: column tiles 0-3 | 4-7
        const int q = (int)(r % pk.chunks_total);
        const int n_tile = (int)(r / pk.chunks_total);
        const int p = q % 9, tsel = kg >> 1, khalf = kg & 1;
        const int stage = 2 * (q / 9) + cq_stage_of(p, tsel), tap = cq_tap_of(p, tsel);
        const int g = n >> 5, j = n & 31;
        const int ch = n_tile * 32 + j;
        float v = 0.0f;
        if (ch < pk.Ch && stage < pk.S) {
            const int row = pk.gate_pos[g] * pk.Ch + ch;
            const int col = pk.stage_col[stage] + khalf * 8 + i;   // column in [x | h]
            v = pk.w[((long long)row * pk.Ct + col) * 9 + tap];
        }
        unsigned hi, lo;
        c2_split(v, hi, lo);
        reinterpret_cast<unsigned short*>(dst)[e] = (unsigned short)(part ? lo : hi);
    }
}

__global__ void cell2_pack_kernel(const Cell2Pack pk, char* __restrict__ dst) {
    const long long total = (long long)pk.n_tiles * pk.chunks_total * (C2_WCHUNK / 2);  // bf16 elements
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(e & 7);
        long long r = e >> 3;
        const int n = (int)(r & 127); r >>= 7;
        const int khalf = (int)(r & 1); r >>= 1;
        const int part = (int)(r & 1); r >>= 1;
        const int q = (int)(r % 3); r /= 3;
        const int chunk = (int)(r % pk.chunks_total);
        const int n_tile = (int)(r / pk.chunks_total);
        const int stage = chunk / 3, dy = chunk - stage * 3;
        const int g = n >> 5, j = n & 31;
        const int ch = n_tile * 32 + j;
        float v = 0.0f;
        if (ch < pk.Ch) {
            const int row = pk.gate_pos[g] * pk.Ch + ch;
            const int col = pk.stage_col[stage] + khalf * 8 + i;   // column in [x | h]
            v = pk.w[((long long)row * pk.Ct + col) * 9 + dy * 3 + q];
        }
        unsigned hi, lo;
        c2_split(v, hi, lo);
        reinterpret_cast<unsigned short*>(dst)[e] = (unsigned short)(part ? lo : hi);
    }
}

hipError_t launch_cell2_pack(const Cell2Pack& pk, void* dst, hipStream_t s) {
    const long long total = (long long)pk.n_tiles * pk.chunks_total * ((pk.qform ? CQ_WCHUNK : C2_WCHUNK) / 2);
    if (!ws_write_ok(dst, (size_t)total * 2, "weight pack (cell2_pack_kernel)")) return hipErrorInvalidValue;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (pk.qform) VPX_LAUNCH(cell2_pack_q_kernel, dim3(blocks), dim3(256), 0, s, pk, reinterpret_cast<char*>(dst));
    else VPX_LAUNCH(cell2_pack_kernel, dim3(blocks), dim3(256), 0, s, pk, reinterpret_cast<char*>(dst));
    return vpx_hip_last_error();
}

size_t cell2_packed_bytes(int n_tiles, int chunks_total) { return (size_t)n_tiles * chunks_total * C2_WCHUNK; }
size_t cell2_packed_bytes_q(int n_tiles, int S) { return (size_t)n_tiles * cell2_qchunks(S) * CQ_WCHUNK; }

// ---------------------------------------------------------------------------------------------------------------
#ifdef VPX_ABLATE
// developer build only (make ablate): per-wave s_memtime stamps of ONE workgroup (block id = Cell2Plan::_p), read back with
// vpx_dbg_cell2_stamps(). Never compiled into the product library.
__device__ unsigned long long c2_stamps[8 * 64];
// ... and of EVERY workgroup of a half-tile q-form launch: start, loop end, end (s_memtime) and HW_ID | XCC_ID << 32 — which workgroups shared
// a CU and in what phase relation (tools/trace_cell2q.py); vpx_dbg_cell2_trace() reads it back.
__device__ unsigned long long c2_trace[8192 * 4];
#define C2_TRACE(k) do { if (wave == 0 && lane == 0 && L < 8192) c2_trace[L * 4 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define C2_STAMP(slot) do { if (stamp_on && lane == 0) c2_stamps[wave * 64 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define C2_EPI_STAMP(slot) do { if (stamp_on && (lane & 63) == 0) c2_stamps[(prow >> 2) * 64 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define C2_STAMP(slot) do { } while (0)
#define C2_EPI_STAMP(slot) do { } while (0)
#define C2_TRACE(k) do { } while (0)
#endif

// The ConvLSTM epilogue (conv_lstm_hzzone.py:62-68), plus the split copy of h_t for the next step / the weight gradient.
// Addressing: one 24-bit multiply per pixel (pixel index x Ch), every array is a wave-uniform base + that 32-bit element
// offset; FULL = the tile lies inside the image (no per-pixel bounds test). The first-generation epilogue spent ~45
// address instructions per pixel, 14 of them quarter-rate 32-bit multiplies — invisible next to a second resident
// workgroup, a quarter of the tile time with one workgroup per CU.
struct Cell2Epi {
    ConvLSTMStepArgs a;
    char* h_sp;               // split h_t [B][HW][Ch] (or null)
    long long h_sp_bstride;   // bytes between batch items

    template <bool FULL>
    __device__ __forceinline__ void run(const f32x16 (&acc)[4], int b, int y0, int x0, int n_tile, int prow, int j, int hh,
                                        int H, int W) const {
        const int ch = n_tile * 32 + j;
        if (ch >= a.Ch) return;
        const unsigned Ch = (unsigned)a.Ch;
        float bi = 0.f, bf = 0.f, bg = 0.f, bo = 0.f;
        if (a.bias) {
            bi = a.bias[a.gate_pos[0] * Ch + ch];
            bf = a.bias[a.gate_pos[1] * Ch + ch];
            bg = a.bias[a.gate_pos[2] * Ch + ch];
            bo = a.bias[a.gate_pos[3] * Ch + ch];
        }
        const size_t img = (size_t)b * H * W;                       // wave-uniform
        const float* const cin_b = a.c_in ? a.c_in + img * Ch : nullptr;
        float* const cout_b = a.c_out + img * Ch;
        float* const hout_b = a.h_out ? a.h_out + (size_t)b * a.h_bstride : nullptr;
        float* const g0 = a.gates ? a.gates + img * 4 * Ch : nullptr;
        char* const hsp_b = h_sp ? h_sp + (size_t)b * h_sp_bstride : nullptr;
        const int che = ch & ~1;
        const unsigned sp_off = (unsigned)((che >> 3) * 32 + (che & 7) * 2 + ((j & 1) ? 16 : 0));
        const int rowpix = (y0 + prow) * W + x0;                    // wave-uniform: pixel index of (tile row prow, column 0)
        constexpr int RB = 16;
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += RB) {
            bool ok[RB];
            unsigned eo[RB], po[RB];
            float cpv[RB], wi[RB], wf[RB], wo[RB];
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int r = r0 + u;
                const int k = (r & 3) + 8 * ((r >> 2) & 1) + 4 * hh;        // column inside the sub-tile row before rotation
                const int row = r >> 3;                                    // accumulator registers 8..15 hold the odd row
                const int px = row ? ((k + 14) & 15) : k;                  // c2_px
                const int pix = rowpix + row * W + px;
                ok[u] = FULL || (y0 + prow + row < H && x0 + px < W);
                po[u] = __umul24((unsigned)pix, Ch);
                eo[u] = po[u] + (unsigned)ch;
                cpv[u] = 0.f; wi[u] = 0.f; wf[u] = 0.f; wo[u] = 0.f;
                if (ok[u]) {
                    if (cin_b) cpv[u] = cin_b[eo[u]];
                    if (a.wci) { wi[u] = a.wci[eo[u]]; wf[u] = a.wcf[eo[u]]; }
                    if (a.wco) wo[u] = a.wco[eo[u]];
                }
            }
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                if (!ok[u]) continue;   // uniform over the lanes of a 32-lane half (depends on r and hh only)
                const int r = r0 + u;
                const float cp = cpv[u];
                const float ai = acc[0][r] + bi + wi[u] * cp, af = acc[1][r] + bf + wf[u] * cp;
                const float ag = acc[2][r] + bg;
                const float i_ = sigmoid_f(ai), f_ = sigmoid_f(af), g_ = tanh_f(ag);
                const float cn = lstm_c(f_, cp, i_, g_);
                const float ao = acc[3][r] + bo + wo[u] * cn;
                const float o_ = sigmoid_f(ao);
                const float hn = o_ * tanh_f(cn);
                cout_b[eo[u]] = cn;
                if (a.h_out) hout_b[eo[u]] = hn;
                if (g0) {
                    const unsigned go = 4u * po[u] + (unsigned)ch;
                    g0[go] = i_;
                    g0[go + Ch] = f_;
                    g0[go + 2 * Ch] = g_;
                    g0[go + 3 * Ch] = o_;
                }
                if (hsp_b) {
                    unsigned hi, lo;
                    c2_split(hn, hi, lo);
                    // neighbour lane (j ^ 1) inside the quad: quad_perm [1,0,3,2] = 0xB1
                    const unsigned nhi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, 0xB1, 0xF, 0xF, true);
                    const unsigned nlo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, 0xB1, 0xF, 0xF, true);
                    const unsigned word = (j & 1) ? (nlo | (lo << 16)) : (hi | (nhi << 16));
                    *reinterpret_cast<unsigned*>(hsp_b + (4u * po[u] + sp_off)) = word;
                }
            }
        }
    }

    // Vectorised epilogue for tiles that lie inside the image and cover 32 valid channels: the accumulators take a round trip
    // through this wave's private 16 KiB of LDS ([gate][pixel][32 ch], conflict-free both ways) so that a lane then owns FOUR
    // consecutive channels of a pixel — every global access is 16 bytes per lane (8 lanes = one pixel's 128-byte channel row)
    // and the address arithmetic runs once per four elements. Same arithmetic per element as run<>(). Three phases per
    // sub-tile so that finish() can order them: vec_load (cell state + peepholes, 16 vector loads per lane), vec_put
    // (accumulators -> LDS), vec_math (LDS -> gates -> c, h, split h).
    struct VecIn { unsigned eo[4]; f32x4 cp[4], wi[4], wf[4], wo[4]; };

    template <bool ROT = true>   // ROT: the 32x32x16 loop's pixel map (odd row rotated, c2_px); else pixel slot ip = row * 16 + column
    __device__ __forceinline__ void vec_load(VecIn& v, int b, int y0, int x0, int n_tile, int prow, int lane, int H, int W) const {
        const unsigned Ch = (unsigned)a.Ch;
        const int cg = lane & 7, p4 = lane >> 3;
        const unsigned ch = (unsigned)(n_tile * 32 + cg * 4);
        const size_t img = (size_t)b * H * W;
        const float* const cin_b = a.c_in ? a.c_in + img * Ch : nullptr;
        const int rowpix = (y0 + prow) * W + x0;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ip = k * 8 + p4;                               // pixel slot 0..31 of the sub-tile
            const int pix = rowpix + (ip >> 4) * W + (ROT ? c2_px(ip) : (ip & 15));
            v.eo[k] = __umul24((unsigned)pix, Ch) + ch;
            v.cp[k] = cin_b ? *reinterpret_cast<const f32x4*>(cin_b + v.eo[k]) : zero;
            v.wi[k] = a.wci ? *reinterpret_cast<const f32x4*>(a.wci + v.eo[k]) : zero;
            v.wf[k] = a.wci ? *reinterpret_cast<const f32x4*>(a.wcf + v.eo[k]) : zero;
            v.wo[k] = a.wco ? *reinterpret_cast<const f32x4*>(a.wco + v.eo[k]) : zero;
        }
    }

    __device__ __forceinline__ void vec_put(const f32x16 (&acc)[4], char* lds, int lane) const {
        const int j = lane & 31, hh = lane >> 5;
        float* ldsf = reinterpret_cast<float*>(lds);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * hh;     // MFMA row = pixel slot of the sub-tile
                ldsf[g * 1024 + i * 32 + j] = acc[g][r];
            }
        // (LDS operations of one wave execute in order: vec_math's reads see these writes without a barrier)
    }

    // (ab: timing-only ablations of the developer build — experiment bits 17 = no stores, 26 = no transcendentals; results are garbage)
    __device__ __forceinline__ void vec_math(const VecIn& v, const char* lds, int b, int n_tile, int lane, int H, int W, int ab = 0) const {
        const unsigned Ch = (unsigned)a.Ch;
        const float* ldsf = reinterpret_cast<const float*>(lds);
        const int cg = lane & 7, p4 = lane >> 3;
        const unsigned ch = (unsigned)(n_tile * 32 + cg * 4);
        f32x4 bi = {0.f, 0.f, 0.f, 0.f}, bf = bi, bg = bi, bo = bi;
        if (a.bias) {
            bi = *reinterpret_cast<const f32x4*>(a.bias + a.gate_pos[0] * Ch + ch);
            bf = *reinterpret_cast<const f32x4*>(a.bias + a.gate_pos[1] * Ch + ch);
            bg = *reinterpret_cast<const f32x4*>(a.bias + a.gate_pos[2] * Ch + ch);
            bo = *reinterpret_cast<const f32x4*>(a.bias + a.gate_pos[3] * Ch + ch);
        }
        const size_t img = (size_t)b * H * W;
        float* const cout_b = a.c_out + img * Ch;
        float* const hout_b = a.h_out ? a.h_out + (size_t)b * a.h_bstride : nullptr;
        float* const g0 = a.gates ? a.gates + img * 4 * Ch : nullptr;
        char* const hsp_b = h_sp ? h_sp + (size_t)b * h_sp_bstride : nullptr;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ip = k * 8 + p4;
            const f32x4 ai = *reinterpret_cast<const f32x4*>(ldsf + 0 * 1024 + ip * 32 + cg * 4);
            const f32x4 af = *reinterpret_cast<const f32x4*>(ldsf + 1 * 1024 + ip * 32 + cg * 4);
            const f32x4 ag = *reinterpret_cast<const f32x4*>(ldsf + 2 * 1024 + ip * 32 + cg * 4);
            const f32x4 ao = *reinterpret_cast<const f32x4*>(ldsf + 3 * 1024 + ip * 32 + cg * 4);
            f32x4 i4, f4, g4, o4, cn, hn;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float c0 = v.cp[k][e];
#ifdef VPX_ABLATE
                if (ab & (1 << 26)) {
                    i4[e] = ai[e] + bi[e] + v.wi[k][e] * c0; f4[e] = af[e] + bf[e] + v.wf[k][e] * c0; g4[e] = ag[e] + bg[e];
                    cn[e] = lstm_c(f4[e], c0, i4[e], g4[e]); o4[e] = ao[e] + bo[e] + v.wo[k][e] * cn[e]; hn[e] = o4[e] * cn[e];
                    continue;
                }
#endif
                i4[e] = sigmoid_f(ai[e] + bi[e] + v.wi[k][e] * c0);
                f4[e] = sigmoid_f(af[e] + bf[e] + v.wf[k][e] * c0);
                g4[e] = tanh_f(ag[e] + bg[e]);
                cn[e] = lstm_c(f4[e], c0, i4[e], g4[e]);
                o4[e] = sigmoid_f(ao[e] + bo[e] + v.wo[k][e] * cn[e]);
                hn[e] = o4[e] * tanh_f(cn[e]);
            }
            const unsigned eo = v.eo[k];
#ifdef VPX_ABLATE
            if (ab & (1 << 17)) { if (cn[0] + hn[1] + i4[2] + f4[3] + g4[0] + o4[1] == 1.2345e-30f) cout_b[eo] = 0.f; continue; }
#endif
            *reinterpret_cast<f32x4*>(cout_b + eo) = cn;
            if (a.h_out) *reinterpret_cast<f32x4*>(hout_b + eo) = hn;   // (null: the consumer reads the split copy below — VPX_FLAG_OUT_SPLIT)
            if (g0) {
                const unsigned go = 4u * (eo - ch) + ch;
                *reinterpret_cast<f32x4*>(g0 + go) = i4;
                *reinterpret_cast<f32x4*>(g0 + go + Ch) = f4;
                *reinterpret_cast<f32x4*>(g0 + go + 2 * Ch) = g4;
                *reinterpret_cast<f32x4*>(g0 + go + 3 * Ch) = o4;
            }
            if (hsp_b) {
                unsigned h[4], l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) c2_split(hn[e], h[e], l[e]);
                // the lane pair (cg even, cg odd) holds one 8-channel group = 32 bytes [8 hi | 8 lo]: the even lane stores the 16 hi
                // bytes, the odd lane the 16 lo bytes — it keeps its own half of that quantity and receives the partner's (quad_perm
                // [1,0,3,2]); one 16-byte store per lane and whole 32-byte sectors per instruction instead of two 8-byte stores
                const bool odd = (cg & 1) != 0;
                const unsigned h0 = h[0] | (h[1] << 16), h1 = h[2] | (h[3] << 16), l0 = l[0] | (l[1] << 16), l1 = l[2] | (l[3] << 16);
                const unsigned s0 = odd ? h0 : l0, s1 = odd ? h1 : l1;             // what the partner stores of mine
                const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s0, 0xB1, 0xF, 0xF, true);
                const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s1, 0xB1, 0xF, 0xF, true);
                char* dst = hsp_b + 4u * (eo - ch) + (ch >> 3) * 32 + (odd ? 16 : 0);   // the group's hi or lo half
                *reinterpret_cast<uint4*>(dst) = odd ? uint4{r0, r1, l0, l1} : uint4{h0, h1, r0, r1};
            }
        }
    }

    // 16x16x32 accumulators: acc[m][nt] = tile row m (16 pixels) x column tile nt (gate nt >> 1, channels (nt & 1) * 16 ..); a
    // lane holds column lane & 15 of pixels 4 * (lane >> 4) .. + 3. Rows 2*mp, 2*mp + 1 fill the same [gate][pixel 32][32 ch] image.
    __device__ __forceinline__ void vec_put16(const f32x4 (&acc)[4][8], int mp, char* lds, int lane) const {
        const int c16 = lane & 15, q4 = lane >> 4;
        float* ldsf = reinterpret_cast<float*>(lds);
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int nt = 0; nt < 8; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ldsf[(nt >> 1) * 1024 + (mm * 16 + 4 * q4 + r) * 32 + (nt & 1) * 16 + c16] = acc[2 * mp + mm][nt][r];
    }

    // (the q form is only selected for tiles inside the image and whole 32-channel tiles: cell2_q_applicable)
    __device__ __forceinline__ void finish16(const f32x4 (&acc)[4][8], char* smem, int wave, int lane, int b, int y0, int x0, int n_tile,
                                             int /*ngr*/, int H, int W, bool stamp_on = false, int ab = 0) const {
        const int prow = 4 * wave;
        char* const lds = smem + wave * 16384;
        VecIn v0, v1;
#ifdef VPX_ABLATE
        if (ab & (1 << 18)) {   // timing only: no epilogue (one never-taken store keeps the accumulators alive)
            float t = 0.f;
            for (int m = 0; m < 4; ++m) for (int nt = 0; nt < 8; ++nt) for (int r = 0; r < 4; ++r) t += acc[m][nt][r];
            if (t == 1.2345e-30f) a.c_out[0] = t;
            return;
        }
#endif
        C2_EPI_STAMP(43);
        vec_load<false>(v0, b, y0, x0, n_tile, prow, lane, H, W);
        C2_EPI_STAMP(44);
        c2_barrier();   // every wave has read its last fragments: the staging buffers become the epilogue's transposition space
        C2_EPI_STAMP(45);
        vec_put16(acc, 0, lds, lane);
        C2_EPI_STAMP(46);
        vec_load<false>(v1, b, y0, x0, n_tile, prow + 2, lane, H, W);
        C2_EPI_STAMP(47);
        vec_math(v0, lds, b, n_tile, lane, H, W, ab);
        C2_EPI_STAMP(48);
        vec_put16(acc, 1, lds, lane);
        C2_EPI_STAMP(49);
        vec_math(v1, lds, b, n_tile, lane, H, W, ab);
        C2_EPI_STAMP(50);
        (void)stamp_on;
    }

    __device__ __forceinline__ void finish(const f32x16 (&acc)[2][4], char* smem, int wave, int lane, int b, int y0, int x0, int n_tile,
                                           int /*ngr*/, int H, int W, bool stamp_on = false) const {
        const int j = lane & 31, hh = lane >> 5;
        const bool full = y0 + 32 <= H && x0 + 16 <= W;
        const bool vec = full && n_tile * 32 + 32 <= a.Ch && (a.Ch & 3) == 0;
        if (vec) {
            // Order matters (in-kernel stamps, 158 k-cycle tile: the epilogue took 17-21 k, of which 2.7 + 4.2 k were the two
            // sub-tiles' waits for their state / peephole loads): sub-tile 0's loads go out BEFORE the barrier and the LDS
            // round trip, sub-tile 1's before sub-tile 0's arithmetic — their latency runs under work that does not need them.
            const int prow = 4 * wave;   // (also the stamp row of C2_EPI_STAMP)
            char* const lds = smem + wave * 16384;
            VecIn v0, v1;
            C2_EPI_STAMP(43);
            vec_load(v0, b, y0, x0, n_tile, prow, lane, H, W);
            c2_barrier();   // every wave has read its last fragments: the staging buffers become the epilogue's transposition space
            C2_EPI_STAMP(44);
            vec_put(acc[0], lds, lane);
            vec_load(v1, b, y0, x0, n_tile, prow + 2, lane, H, W);
            C2_EPI_STAMP(45);
            vec_math(v0, lds, b, n_tile, lane, H, W);
            C2_EPI_STAMP(46);
            vec_put(acc[1], lds, lane);
            vec_math(v1, lds, b, n_tile, lane, H, W);
            C2_EPI_STAMP(47);
            return;
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            if (full) run<true>(acc[m], b, y0, x0, n_tile, 4 * wave + 2 * m, j, hh, H, W);
            else run<false>(acc[m], b, y0, x0, n_tile, 4 * wave + 2 * m, j, hh, H, W);
        }
        (void)stamp_on;
    }
};

// Plain epilogue of the same main loop ("conv2"): y = conv(src) (+ bias), output channel c of tile n_tile, group g, column j =
// (n_tile * gpt + g) * 32 + j; channels [0, split) go to out0, [split, Co) to out1 (either may be null = dropped). The
// accumulators take the same round trip through the wave's 16 KiB of LDS as Cell2Epi::run_vec, so a lane owns four
// consecutive channels of a pixel and stores 16 bytes. First user: the ConvLSTM data gradient (dG -> dx_t | dh_{t-1}).
struct Conv2Epi {
    const float* bias;
    int Co, split, gpt, accumulate;
    float* out0; long long bstride0; int ld0, _p0;
    float* out1; long long bstride1; int ld1, _p1;

    // stores one 32-pixel sub-tile (tile rows prow, prow + 1) from the wave's [group][pixel 32][32 ch] LDS image
    template <bool ROT>
    __device__ __forceinline__ void store_sub(const float* ldsf, int lane, int b, int y0, int x0, int n_tile, int ngr, int prow, int H, int W) const {
        const int cg = lane & 7, p4 = lane >> 3;
        const bool v4 = ((Co | split | ld0 | ld1) & 3) == 0;
        float* const o0 = out0 ? out0 + (size_t)b * bstride0 : nullptr;
        float* const o1 = out1 ? out1 + (size_t)b * bstride1 : nullptr;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g >= ngr) continue;
            const int c = (n_tile * gpt + g) * 32 + cg * 4;
            if (c >= Co) continue;
            const bool first = c < split;
            float* const ob = first ? o0 : o1;
            const unsigned ld = (unsigned)(first ? ld0 : ld1);
            const int cc = first ? c : c - split;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ip = k * 8 + p4;
                const int y = y0 + prow + (ip >> 4), x = x0 + (ROT ? c2_px(ip) : (ip & 15));
                f32x4 v = *reinterpret_cast<const f32x4*>(ldsf + g * 1024 + ip * 32 + cg * 4);
                if (y >= H || x >= W) continue;
                const size_t e = (size_t)__umul24((unsigned)(y * W + x), ld) + cc;
                if (v4) {
                    if (!ob) continue;
                    if (bias) v += *reinterpret_cast<const f32x4*>(bias + c);
                    if (accumulate) v += *reinterpret_cast<const f32x4*>(ob + e);
                    *reinterpret_cast<f32x4*>(ob + e) = v;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int cq = c + q;
                        if (cq >= Co) continue;
                        const bool f1 = cq < split;
                        float* const oq = f1 ? o0 : o1;
                        if (!oq) continue;
                        const size_t eq = (size_t)__umul24((unsigned)(y * W + x), (unsigned)(f1 ? ld0 : ld1)) + (f1 ? cq : cq - split);
                        float val = v[q] + (bias ? bias[cq] : 0.f);
                        if (accumulate) val += oq[eq];
                        oq[eq] = val;
                    }
                }
            }
        }
    }

    __device__ __forceinline__ void finish(const f32x16 (&acc)[2][4], char* smem, int wave, int lane, int b, int y0, int x0, int n_tile,
                                           int ngr, int H, int W, bool = false) const {
        c2_barrier();
        float* ldsf = reinterpret_cast<float*>(smem + wave * 16384);
        const int j = lane & 31, hh = lane >> 5;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (g < ngr)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ldsf[g * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + j] = acc[m][g][r];
            // (LDS operations of one wave execute in order: no barrier between these writes and the reads below)
            store_sub<true>(ldsf, lane, b, y0, x0, n_tile, ngr, 4 * wave + 2 * m, H, W);
        }
    }

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
            store_sub<false>(ldsf, lane, b, y0, x0, n_tile, ngr, 4 * wave + 2 * mp, H, W);
        }
    }
};


// Fragment registers of the MFMA loop: three activation sets (one per tap column dx, so the set of dx = 0 can be refilled
// for the next chunk while dx = 2 is still being multiplied) and two weight sets (even / odd gate group).
struct C2Frags {
    bf16x8 ah[3][2], al[3][2];
    bf16x8 bh[2], bl[2];
};

// Epi = Cell2Epi (the fused ConvLSTM step; every N tile holds four gate groups) or Conv2Epi (a plain convolution over the
// same main loop: the N tile's four 32-column groups are 128 consecutive output channels, of which the last tile may
// use fewer — ALLG = false skips the MFMAs of the unused groups under a wave-uniform branch).
template <class Epi, bool ALLG>
__global__ __launch_bounds__(512, 2) void cell2_kernel(const Cell2Plan P, const Epi epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, hh = lane >> 5;

    // XCD-aware tile decode (same rule as conv_gemm.hip's xcd_tile: N tile fastest, contiguous ranges per XCD)
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
    const int x0 = tx * 16, y0 = ty * 32;
    int ngr = 4;   // 32-column groups of this N tile that hold outputs
    if constexpr (!ALLG) { ngr = P.n_groups - n_tile * P.gpt; if (ngr > P.gpt) ngr = P.gpt; }

    char* const Abuf = smem;
    char* const Wbuf = smem + 2 * C2_ABUF;
#ifdef VPX_ABLATE
    const bool stamp_on = (int)L == P._p;
#endif
    C2_STAMP(0);

    // this thread's five pieces of an activation stage: piece = tid + 512 u -> (plane, halo position)
    int pixoff[5], choff[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const int piece = tid + 512 * u;
        const int plane = piece / C2_PLANE_POS, pos = piece - plane * C2_PLANE_POS;
        const int hy = pos / C2_HALO_W, hx = pos - hy * C2_HALO_W;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        const bool ok = pos < C2_NPOS && gy >= 0 && gy < P.H && gx >= 0 && gx < P.W;
        pixoff[u] = ok ? gy * P.W + gx : -1;
        choff[u] = (plane & 1) * 32 + (plane >> 1) * 16;  // plane = part*2 + khalf; pixel row: [group][hi 16 B | lo 16 B]
    }
    const int dma_off = (wave * 64) * 16;  // wave-uniform LDS offset of this wave's 64 pieces inside a 512-piece pass
    const char* const wtile = P.wpk + (size_t)n_tile * P.chunks_total * C2_WCHUNK + tid * 16;

    // one 16-byte piece per call: the copies of a sync point are spread over the following gate groups (one or two pieces
    // behind each group's MFMAs) instead of issued as a burst — a burst right after the barrier kept all eight waves off
    // the matrix pipe for the 0.3-0.8k cycles the 3-8 LDS-DMA issues take (measured with in-kernel stamps)
    // scalars of a stage (wave-uniform arithmetic on kernel arguments, no memory access)
    const int nx = P.nx, S = P.nx + P.nh;
    const char* const xb = P.seg[0].sp + (size_t)b * P.seg[0].bstride;
    const char* const hb = P.seg[1].sp + (size_t)b * P.seg[1].bstride;
    const unsigned xrow = (unsigned)P.seg[0].C * 4u, hrow = (unsigned)P.seg[1].C * 4u;
    auto stage_chunk0 = [&](int s) { return 3 * (s < nx ? s : P.hs_off + s - nx); };
    auto issue_A1 = [&](int s, int buf, int u) {
        const bool isx = s < nx;
        const char* base = isx ? xb + s * 64 : hb + (s - nx) * 64;   // 16 channels = 64 bytes of a split pixel row
        const unsigned prow = isx ? xrow : hrow;
        const char* src = pixoff[u] >= 0 ? base + (size_t)((unsigned)pixoff[u] * (unsigned long long)prow) + choff[u]
                                         : reinterpret_cast<const char*>(c2_zero16);
        c2_dma16(src, Abuf + buf * C2_ABUF + dma_off + u * 8192);
    };
    auto issue_W1 = [&](int chunk, int buf, int u) {
        c2_dma16(wtile + (size_t)chunk * C2_WCHUNK + u * 8192, Wbuf + buf * C2_WCHUNK + dma_off + u * 8192);
    };
    auto issue_A = [&](int s, int buf) {
#pragma unroll
        for (int u = 0; u < 5; ++u) issue_A1(s, buf, u);
    };
    auto issue_W = [&](int chunk, int buf) {
#pragma unroll
        for (int u = 0; u < 3; ++u) issue_W1(chunk, buf, u);
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][g][r] = 0.0f;

    // MFMA row j of a sub-tile = pixel (row j>>4, column c2_px(j)): the odd row's columns are rotated by 14 so that the 16
    // lanes a ds_read_b128 serves per cycle ({0-3,12-15,20-27} / {4-11,16-19,28-31}) fall on 16 different 16-byte bank
    // groups although the halo row pitch is 18 slots (unrotated: two 2-way conflicts per group)
    const int a_lane = hh * C2_PLANE + ((4 * wave + (j >> 4)) * C2_HALO_W + c2_px(j)) * 16;
    const int w_lane = hh * 2048 + j * 16;
    C2Frags F;
    auto load_A = [&](const char* A, int dy, int dx) {   // the four activation fragments of tap (dy, dx) -> set dx
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int off = ((2 * m + dy) * C2_HALO_W + dx) * 16;
            F.ah[dx][m] = *reinterpret_cast<const bf16x8*>(A + off);
            F.al[dx][m] = *reinterpret_cast<const bf16x8*>(A + 2 * C2_PLANE + off);
        }
    };
    auto load_B = [&](const char* Wb, int dx, int g) {   // the two weight fragments of (k-step dx, gate g) -> set g & 1
        F.bh[g & 1] = *reinterpret_cast<const bf16x8*>(Wb + dx * 8192 + g * 512);
        F.bl[g & 1] = *reinterpret_cast<const bf16x8*>(Wb + dx * 8192 + 4096 + g * 512);
    };

    // Pipeline (chunk c = tap row dy of stage s; weight ring of three chunks, two activation buffers):
    //   sync point P_c sits before the 9th of the 12 gate groups of chunk c: wait for this thread's pieces of chunk c+1 (and
    //   of stage s+1 when c is the stage's last chunk) -> barrier -> issue the copy of chunk c+2 (its ring slot was last read
    //   in chunk c-1, which every wave has left) and, in a stage's first chunk, of stage s+1's halo tile. The last group of
    //   chunk c then already reads the first fragments of chunk c+1: no LDS latency is exposed at a chunk boundary, and a
    //   copy has a whole chunk (weights) or two (activations) of MFMA time to land.
    if (S > 0) {
        issue_A(0, 0);
        issue_W(stage_chunk0(0), 0);
        issue_W(stage_chunk0(0) + 1, 1);
        C2_STAMP(1);
        C2_WAIT_VM(3);  // stage 0 and chunk 0 have landed (chunk 1 may still fly)
        c2_barrier();
        C2_STAMP(2);
        load_A(Abuf + a_lane, 0, 0);
        load_B(Wbuf + w_lane, 0, 0);
    }
    for (int s = 0; s < S; ++s) {
        const bool more = s + 1 < S;
        const int ck_cur = stage_chunk0(s), ck_next = stage_chunk0(s + 1);
        const char* A = Abuf + (s & 1) * C2_ABUF + a_lane;
        const char* An = Abuf + ((s + 1) & 1) * C2_ABUF + a_lane;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const char* Wb = Wbuf + dy * C2_WCHUNK + w_lane;
            const char* Wn = Wbuf + ((dy + 1) % 3) * C2_WCHUNK + w_lane;
            const bool has_next = dy < 2 || more;   // another chunk follows this one
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                const int dx = n >> 2, g = n & 3;
                if (n == 8) {
                    // ---- sync point P_c ----
                    if (s < 3) C2_STAMP(3 + (s * 3 + dy) * 3);
                    if (dy == 1 && more) C2_WAIT_VM(5); else C2_WAIT_VM(0);   // dy == 1: stage s+1's tile may still fly
                    if (s < 3) C2_STAMP(4 + (s * 3 + dy) * 3);
                    c2_barrier();
                    if (s < 3) C2_STAMP(5 + (s * 3 + dy) * 3);
                }
                // ---- fragments of the next gate group ----
                if (n < 11) {
                    const int ndx = (n + 1) >> 2, ng = (n + 1) & 3;
                    load_B(Wb, ndx, ng);
                    if (ng == 0) load_A(A, dy, ndx);
                } else if (has_next) {
                    load_B(Wn, 0, 0);
                    load_A(dy < 2 ? A : An, dy < 2 ? dy + 1 : 0, 0);
                }
                // ---- 6 MFMAs of gate group (dx, g) ----
                __builtin_amdgcn_s_setprio(1);
                if (ALLG || g < ngr)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    f32x16 c = acc[m][g];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.al[dx][m], F.bh[g & 1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.ah[dx][m], F.bl[g & 1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.ah[dx][m], F.bh[g & 1], c, 0, 0, 0);
                    acc[m][g] = c;
                }
                __builtin_amdgcn_s_setprio(0);
                // ---- this sync point's copies, spread behind the MFMAs of groups 8..11: weights first (3 pieces), then the
                //      next stage's halo tile (5 pieces) — the vmcnt waits at the sync points count on exactly this order ----
                if (n >= 8) {
                    const int k = n - 8;   // 0..3
                    if (dy == 0) {
                        if (k < 3) issue_W1(ck_cur + 2, 2, k);
                        if (more) {
                            if (k == 2) issue_A1(s + 1, (s + 1) & 1, 0);
                            if (k == 3) {
#pragma unroll
                                for (int u = 1; u < 5; ++u) issue_A1(s + 1, (s + 1) & 1, u);
                            }
                        }
                    } else if (more && k < 3) {
                        issue_W1(ck_next + dy - 1, dy - 1, k);
                    }
                }
            }
        }
    }
    C2_STAMP(40);
#ifdef VPX_ABLATE
    epi.finish(acc, smem, wave, lane, b, y0, x0, n_tile, ngr, P.H, P.W, stamp_on);
#else
    epi.finish(acc, smem, wave, lane, b, y0, x0, n_tile, ngr, P.H, P.W);
#endif
    C2_STAMP(42);
}

// ---------------------------------------------------------------------------------------------------------------
// The same main loop on v_mfma_f32_16x16x32_bf16 ("q form"). Same workgroup tile (32x16 pixels x 128 columns), same wave tile
// (4 tile rows x 128 columns = 32 accumulator tiles of 16x16: 128 registers), same activation stage image, same fragment bytes
// read from LDS per MFMA cycle — but the chip holds a higher clock on this MFMA shape (MI355X_MICROARCH.md, DVFS give-back (7):
// 1.12-1.14x the FLOP/s at equal cycles per FLOP), and the clock, not issue, bounds the 32x32x16 loop (1.81 GHz measured).
//   K = 32 step: k groups (lane >> 4) 0,1 = channel halves of tap tA, 2,3 = channel halves of tap tB of ONE 16-channel stage
//   (cell2_pack_q_kernel has the schedule); the lane's fragment address is base_kind(lane) + immediate, base kinds = the three
//   slot distances tB - tA that occur (1, 16, and the cross step's "other buffer").
//   M tile = one tile row (16 pixels): no column rotation needed, fragment reads are conflict-free as they stand.
//   Weight ring: three 16 KiB chunks (one step each); sync point S_q before the 6th of the 8 column tiles of step q: wait for
//   the own pieces of chunk q+1 -> barrier -> copy of chunk q+2 (its slot was last read in step q-1), and at steps 0 / 5 of a
//   period the copy of the period's odd stage / the next period's even stage (buffers last read in steps 8 / 4).
//   Fragments: one activation set (4 rows x hi/lo), refilled for step q+1 row by row behind the last column tile's MFMAs;
//   two weight sets (column tile nt + 1 is read before the MFMAs of nt).
struct CQFrags { bf16x8 ah[4], al[4], bh[2], bl[2]; };

// (cq_slot / cq_kind / cq_aoff: cell2_dev.h)

// NW = 8: the 32x16 tile, one workgroup per CU. NW = 4: the half tile (16x16 pixels), two workgroups per CU — the epilogue of
// one (transcendental-bound: ~18 k of a 158 k-cycle tile at NW = 8) and its prologue run under the other's MFMAs. Its weight
// ring holds TWO chunks and turns over in halves (a chunk = [half][...]: column tiles 0-3 | 4-7): sync point X_q sits before
// the weight read of column tile 4 of step q; after its barrier every wave has read tiles 0-3 of chunk q and all of chunk q-1, so
// half 1 of chunk q+1 and half 0 of chunk q+2 are requested there, each with a whole step to land (as in the ring of three).
// PLAIN (round 4; half tile only): VPX_PREC_BF16 — the hi parts of both operands only (BASELINE configs[1]'s literal dtype, ~2e-3,
// outside the 1e-4 bar): one MFMA per product instead of three, the lo planes of a stage and the lo half of every weight chunk are
// neither copied nor read (halves the LDS-DMA bytes and the fragment reads); same stage image, same pack, same epilogue.
template <class Epi, bool ALLG, int NW, bool PLAIN = false>
__global__ __launch_bounds__(64 * NW, 2) void cell2_kernel_q(const Cell2Plan P, const Epi epi) {
    using G = CQGeom<NW>;
    static_assert(!PLAIN || NW == 4, "plain-bf16 form: half tile only (its hi planes are exactly three pieces per thread)");
    constexpr int NPC = PLAIN ? G::NPIECE / 2 : G::NPIECE;   // stage-copy pieces per thread (planes 0, 1 = hi come first)
    constexpr int WPC = PLAIN ? 1 : G::WPIECE;               // copies per thread and half chunk (piece 0 = the hi part)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kg = lane >> 4;

    const unsigned L = blockIdx.x;
    const long long total = (long long)P.grid_m * P.n_tiles;
    const long long per_xcd = (total + 7) / 8;
    const long long sidx = (long long)(L & 7) * per_xcd + (L >> 3);
    if ((long long)(L >> 3) >= per_xcd || sidx >= total) return;
    int mt = (int)(sidx / P.n_tiles);
    int n_tile_ = (int)(sidx - (long long)mt * P.n_tiles);
    if ((P._q & 8) && per_xcd % P.n_tiles == 0) {   // diagnostic: the N tiles of a pixel tile D dispatch positions apart inside the XCD
        const long long i = L >> 3;
        const int D = (P._q >> 8) & 0xfff;
        if (D == 0) {                                // half a launch apart (no L2 sharing possible)
            const long long per_n = per_xcd / P.n_tiles;
            n_tile_ = (int)(i / per_n);
            mt = (int)(((long long)(L & 7) * per_xcd) / P.n_tiles + i % per_n);
        } else if (per_xcd % ((long long)D * P.n_tiles) == 0) {
            const long long blk = i / ((long long)D * P.n_tiles), r = i % ((long long)D * P.n_tiles);
            n_tile_ = (int)(r / D);
            mt = (int)(((long long)(L & 7) * per_xcd) / P.n_tiles + blk * D + r % D);
        }
    }
    const int n_tile = n_tile_;
    const int tx = mt % P.tiles_x;
    mt /= P.tiles_x;
    const int ty = mt % P.tiles_y;
    const int b = mt / P.tiles_y;
    const int x0 = tx * 16, y0 = ty * G::TH;
    int ngr = 4;
    if constexpr (!ALLG) { ngr = P.n_groups - n_tile * P.gpt; if (ngr > P.gpt) ngr = P.gpt; }

#ifdef VPX_ABLATE
    if (wave == 0 && lane == 0 && L < 8192) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        c2_trace[L * 4 + 3] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
    }
#endif
    C2_TRACE(0);
    char* const Abuf = smem;
    char* const Wbuf = smem + 2 * G::ABUF;

    int pixoff[G::NPIECE], choff[G::NPIECE];
#pragma unroll
    for (int u = 0; u < G::NPIECE; ++u) {
        const int piece = tid + G::NT * u;
        const int plane = piece / G::PLANE_POS, pos = piece - plane * G::PLANE_POS;
        const int hy = pos / C2_HALO_W, hx = pos - hy * C2_HALO_W;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        const bool ok = pos < G::NPOS && gy >= 0 && gy < P.H && gx >= 0 && gx < P.W;
        pixoff[u] = ok ? gy * P.W + gx : -1;
        choff[u] = (plane & 1) * 32 + (plane >> 1) * 16;
    }
    const int dma_off = (wave * 64) * 16;
    const char* const wtile = P.wpk + (size_t)n_tile * P.chunks_total * CQ_WCHUNK + tid * 16;

    const int nx = P.nx, S = P.nx + P.nh, Q = (9 * S + 1) / 2;
    const char* const xb = P.seg[0].sp + (size_t)b * P.seg[0].bstride;
    const char* const hb = P.seg[1].sp + (size_t)b * P.seg[1].bstride;
    const unsigned xrow = (unsigned)P.seg[0].C * 4u, hrow = (unsigned)P.seg[1].C * 4u;
    auto issue_A1 = [&](int s, int buf, int u) {
        const bool isx = s < nx;
        const char* base = isx ? xb + s * 64 : hb + (s - nx) * 64;
        const unsigned prow = isx ? xrow : hrow;
        const char* src = pixoff[u] >= 0 ? base + (size_t)((unsigned)pixoff[u] * (unsigned long long)prow) + choff[u]
                                         : reinterpret_cast<const char*>(c2_zero16);
        c2_dma16(src, Abuf + buf * G::ABUF + dma_off + u * (G::NT * 16));
    };
    auto issue_Wh = [&](int chunk, int slot, int half) {   // one 8 KiB half of a weight chunk
#pragma unroll
        for (int w = 0; w < WPC; ++w)
            c2_dma16(wtile + (size_t)chunk * CQ_WCHUNK + half * 8192 + w * (G::NT * 16),
                     Wbuf + slot * CQ_WCHUNK + half * 8192 + w * (G::NT * 16) + dma_off);
    };

#ifdef VPX_ABLATE
    const bool stamp_on = (int)L == P._p;
#endif
    C2_STAMP(0);
    f32x4 acc[4][8];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[m][nt][r] = 0.0f;

    // lane bases into the activation buffers (tap (0,0), tile row 4 * wave, buffer 0, hi plane): k group = tap half * 2 + channel half
    const int a_lane = (kg & 1) * G::PLANE + ((4 * wave) * C2_HALO_W + r16) * 16;
    const int base1 = a_lane + (kg >> 1) * 16;                        // tB one slot right of tA
    const int base16 = a_lane + (kg >> 1) * 256;                      // tB = (dy + 1, dx - 2): 16 slots further
    const int baseX = a_lane + (kg >> 1) * (G::ABUF - cq_slot(8) * 16);  // cross step: tap 0 of the odd stage (buffer 1)
    // weight fragments: lane base of ring slot 0 (wb0) and, NW = 4, of slot 1 (wb1); with two slots and nine steps per period
    // the slot of period step p alternates between periods: the two bases swap at each period end
    int wb0 = 2 * G::ABUF + kg * 1024 + r16 * 16, wb1 = wb0 + CQ_WCHUNK;
    CQFrags F;
    auto load_A1 = [&](int p, int m, int bx) {   // activation fragments (hi, lo) of tile row m for period step p
        const int base = cq_kind(p) == 0 ? base1 : (cq_kind(p) == 1 ? base16 : bx);
        const char* a = smem + base + cq_aoff(p, G::ABUF) + m * (C2_HALO_W * 16);
        F.ah[m] = *reinterpret_cast<const bf16x8*>(a);
        if constexpr (!PLAIN) F.al[m] = *reinterpret_cast<const bf16x8*>(a + 2 * G::PLANE);
    };
    auto load_B = [&](int p, int nt) {           // weight fragments (hi, lo) of column tile nt of period step p -> set nt & 1
        const int base = NW == 8 ? wb0 + (p % 3) * CQ_WCHUNK : ((p & 1) ? wb1 : wb0);
        const char* w = smem + base + (nt >> 2) * 8192 + (nt & 3) * 256;
        F.bh[nt & 1] = *reinterpret_cast<const bf16x8*>(w);
        if constexpr (!PLAIN) F.bl[nt & 1] = *reinterpret_cast<const bf16x8*>(w + 4096);
    };

    // NW = 8: waves 4-7 meet each sync point at THEIR tile 1 instead of tile 5, i.e. they run half a chunk behind waves 0-3: the two
    // waves of a SIMD then do not reach fragment reads, MFMA groups and the barrier in lockstep (MI355X_MICROARCH.md, two waves per
    // SIMD, item 9). Bit-identical outputs; +0.4..0.8 % on every block shape (tools/ab_exp.py; experiment bit 0 switches it off). A
    // static s_setprio 1 for waves 4-7 instead of the per-group flips measured -0.5..0.8 % and is not kept.
    const bool late_sync = NW == 8 && !(P._q & 1) && wave >= 4;
    if (S > 0) {
#pragma unroll
        for (int u = 0; u < NPC; ++u) issue_A1(0, 0, u);
        issue_Wh(0, 0, 0); issue_Wh(0, 0, 1);
        if (Q > 1) {   // NW = 8: all of chunk 1; NW = 4: its first half — two copies per thread either way (PLAIN: one)
            issue_Wh(1, 1, 0);
            if constexpr (NW == 8) issue_Wh(1, 1, 1);
            if constexpr (PLAIN) C2_WAIT_VM(1); else C2_WAIT_VM(2);
        } else C2_WAIT_VM(0);
        C2_STAMP(1);
        c2_barrier();
        C2_STAMP(2);
#pragma unroll
        for (int m = 0; m < 4; ++m) load_A1(0, m, a_lane);
        load_B(0, 0);
    }
    for (int s0 = 0; s0 < S; s0 += 2) {
        const bool odd = s0 + 1 < S;         // the period's odd stage exists
        const bool more = s0 + 2 < S;        // another period follows
        const int q0 = (s0 >> 1) * 9;
        const int par = (s0 >> 1) & 1;       // NW = 4: ring slot of period step 0
        const int bx = odd ? baseX : a_lane;  // without an odd stage the cross step's second half multiplies zero weights: read valid data
#pragma unroll
        for (int p = 0; p < 9; ++p) {
            if (p < 5 || odd) {
                const int q = q0 + p;
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) {
                    const bool stage_flies = (p == 1 && odd) || (p == 6 && more);   // the stage copy issued one step ago may still fly
                    if constexpr (NW == 8) {
                        if ((nt == 5 && !late_sync) || (nt == 1 && late_sync)) {
                            // ---- sync point S_q (waves 0-3 before their tile 5, waves 4-7 before their tile 1: see late_sync) ----
                            if (stage_flies) C2_WAIT_VM(5); else C2_WAIT_VM(0);
                            c2_barrier();
                        }
                    } else {
                        if (nt == 3) {
                            // ---- sync point X_q: the fragments of tiles 0-3 are in registers (lgkmcnt) before their half chunk is given away ----
                            if (stage_flies) { if constexpr (PLAIN) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); }
                            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                            c2_barrier();
                        }
                    }
                    // ---- weight fragments of the next column tile ----
                    //      (after the last step these reads, like the row refills below, fetch bytes nobody uses: cheaper than a branch)
                    if (nt < 7) load_B(p, nt + 1);
                    else load_B(p + 1, 0);
                    // ---- 12 MFMAs of column tile nt ----
                    __builtin_amdgcn_s_setprio(1);
                    if constexpr (ALLG) {
#pragma unroll
                        for (int m = 0; m < 4; ++m) {
                            f32x4 c = acc[m][nt];
                            if constexpr (!PLAIN) {
                                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.al[m], F.bh[nt & 1], c, 0, 0, 0);
                                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.ah[m], F.bl[nt & 1], c, 0, 0, 0);
                            }
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.ah[m], F.bh[nt & 1], c, 0, 0, 0);
                            acc[m][nt] = c;
                            // the last column tile frees row m's fragments: refill them for the next step
                            if (nt == 7) load_A1(p == 8 ? 0 : p + 1, m, bx);
                        }
                    } else {
                        // An N tile that holds fewer than four 32-column groups (a data gradient's Cin + Ch columns: 80, 160, 192 in
                        // convlstm-shi) skips the MFMAs of its empty column tiles — with ONE test per column tile (the first two always
                        // hold outputs). Inside the row loop the same test put a branch around each of the period's 288 MFMA groups:
                        // 497 branches and 689 waits in the loop's ISA against 33 and 171 in the all-groups form.
                        if (nt < 2 || nt < 2 * ngr) {
#pragma unroll
                            for (int m = 0; m < 4; ++m) {
                                f32x4 c = acc[m][nt];
                                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.al[m], F.bh[nt & 1], c, 0, 0, 0);
                                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.ah[m], F.bl[nt & 1], c, 0, 0, 0);
                                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.ah[m], F.bh[nt & 1], c, 0, 0, 0);
                                acc[m][nt] = c;
                            }
                        }
                        if (nt == 7) {
#pragma unroll
                            for (int m = 0; m < 4; ++m) load_A1(p == 8 ? 0 : p + 1, m, bx);
                        }
                    }
                    __builtin_amdgcn_s_setprio(0);
                    // ---- this sync point's copies, behind the MFMAs of the following column tiles: the weights first, then the stage ----
                    if constexpr (NW == 8) {
                        const int i0 = late_sync ? 1 : 5;   // first tile after this wave's sync point
                        if (nt == i0 && q + 2 < Q) { issue_Wh(q + 2, (p + 2) % 3, 0); issue_Wh(q + 2, (p + 2) % 3, 1); }
                        if (p == 0 && odd) {
                            if (nt == i0 + 1) { issue_A1(s0 + 1, 1, 0); issue_A1(s0 + 1, 1, 1); }
                            if (nt == i0 + 2) { issue_A1(s0 + 1, 1, 2); issue_A1(s0 + 1, 1, 3); issue_A1(s0 + 1, 1, 4); }
                        }
                        if (p == 5 && more) {
                            if (nt == i0 + 1) { issue_A1(s0 + 2, 0, 0); issue_A1(s0 + 2, 0, 1); }
                            if (nt == i0 + 2) { issue_A1(s0 + 2, 0, 2); issue_A1(s0 + 2, 0, 3); issue_A1(s0 + 2, 0, 4); }
                        }
                    } else {
                        if (nt == 3) {
                            const int sl = (p & 1) ^ par;   // ring slot of chunk q (and q + 2)
                            if (q + 1 < Q) issue_Wh(q + 1, sl ^ 1, 1);
                            if (q + 2 < Q) issue_Wh(q + 2, sl, 0);
                        }
                        if ((p == 0 && odd) || (p == 5 && more)) {
                            const int st = p == 0 ? s0 + 1 : s0 + 2, bf = p == 0 ? 1 : 0;
                            if (nt == 4) { issue_A1(st, bf, 0); issue_A1(st, bf, 1); issue_A1(st, bf, 2); }
                            if constexpr (!PLAIN) { if (nt == 5) { issue_A1(st, bf, 3); issue_A1(st, bf, 4); issue_A1(st, bf, 5); } }
                        }
                    }
                }
            }
        }
        if constexpr (NW == 4) { const int t = wb0; wb0 = wb1; wb1 = t; }
    }
    C2_STAMP(40);
    C2_TRACE(1);
#ifdef VPX_ABLATE
    if constexpr (std::is_same<Epi, Cell2Epi>::value) epi.finish16(acc, smem, wave, lane, b, y0, x0, n_tile, ngr, P.H, P.W, stamp_on, P._q);
    else
#endif
    epi.finish16(acc, smem, wave, lane, b, y0, x0, n_tile, ngr, P.H, P.W);
    C2_STAMP(41);
#ifdef VPX_ABLATE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // how long the epilogue's stores take to complete
#endif
    C2_STAMP(42);
    C2_TRACE(2);
}

// Half tile (cell2_kernel_q<.., 4>: 16x16-pixel tiles, two workgroups per CU) or the 32x16 tile? Measured (tools/ab_exp.py,
// B=128, five block shapes): the half tile is +1.5..7 % per forward step and +1.3..3 % per block forward + backward (one shape
// -0.5 %) — 14 % fewer cycles, of which the chip takes 10 % back as clock (1.96 -> 1.77 GHz: MFMA busy 60 -> 69 %, power-bound).
// VPX_OPT_EXPERIMENT bit 2 forces the full tile (A/B runs, tests).
static bool cell2_half_tile(const Cell2Plan& p, bool ragged_ok) {
    if (!ragged_ok && (p.H & 15) != 0) return false;   // the fused step's vector epilogue wants tiles inside the image
    if (!ragged_ok && (p.H & 31) != 0) return true;    // ... and 32-row tiles would not be
    return !(g_experiment & 4);
}

template <class Epi, bool ALLG>
static hipError_t launch_cell2_t(const Cell2Plan& plan, const Epi& epi, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = vpx_func_attr(reinterpret_cast<const void*>(&cell2_kernel<Epi, ALLG>), hipFuncAttributeMaxDynamicSharedMemorySize, C2_LDS);
        if (e != hipSuccess) return e;
        e = vpx_func_attr(reinterpret_cast<const void*>(&cell2_kernel_q<Epi, ALLG, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, CQGeom<8>::LDS);
        if (e != hipSuccess) return e;
        e = vpx_func_attr(reinterpret_cast<const void*>(&cell2_kernel_q<Epi, ALLG, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, CQGeom<4>::LDS);
        if (e != hipSuccess) return e;
        attr_set = !g_dry_run;
    }
    Cell2Plan p = plan;
    p.grid_m = plan.B * plan.tiles_x * plan.tiles_y;
    if (plan.qform) {
        p._q = g_experiment;
#ifdef VPX_ABLATE
        p._p = dev_switch("VPX_C2_STAMP_BLOCK", -1);
#endif
        if (plan.plain || cell2_half_tile(p, std::is_same<Epi, Conv2Epi>::value)) {
            p.tiles_y = (p.H + 15) / 16;
            p.grid_m = p.B * p.tiles_x * p.tiles_y;
            const long long per_xcd_h = ((long long)p.grid_m * p.n_tiles + 7) / 8;
            if constexpr (std::is_same<Epi, Cell2Epi>::value && ALLG) {
                if (plan.plain) {
                    static bool attr_plain = false;
                    if (!attr_plain) {
                        hipError_t e = vpx_func_attr(reinterpret_cast<const void*>(&cell2_kernel_q<Epi, ALLG, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, CQGeom<4>::LDS);
                        if (e != hipSuccess) return e;
                        attr_plain = !g_dry_run;
                    }
                    VPX_LAUNCH((cell2_kernel_q<Epi, ALLG, 4, true>), dim3((unsigned)(per_xcd_h * 8)), dim3(256), CQGeom<4>::LDS, s, p, epi);
                    return vpx_hip_last_error();
                }
            }
            if (plan.plain) return hipErrorInvalidValue;   // (only the fused cell step has the plain form)
#ifdef VPX_ABLATE
            if (g_experiment & (1 << 19)) {   // timing only: ONE workgroup per CU (an LDS request above half a CU's)
                static bool attr_big = false;
                if (!attr_big) { vpx_func_attr(reinterpret_cast<const void*>(&cell2_kernel_q<Epi, ALLG, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024); attr_big = true; }
                VPX_LAUNCH((cell2_kernel_q<Epi, ALLG, 4>), dim3((unsigned)(per_xcd_h * 8)), dim3(256), 100 * 1024, s, p, epi);
                return vpx_hip_last_error();
            }
#endif
            VPX_LAUNCH((cell2_kernel_q<Epi, ALLG, 4>), dim3((unsigned)(per_xcd_h * 8)), dim3(256), CQGeom<4>::LDS, s, p, epi);
            return vpx_hip_last_error();
        }
        const long long per_xcd_q = ((long long)p.grid_m * p.n_tiles + 7) / 8;
        VPX_LAUNCH((cell2_kernel_q<Epi, ALLG, 8>), dim3((unsigned)(per_xcd_q * 8)), dim3(512), CQGeom<8>::LDS, s, p, epi);
        return vpx_hip_last_error();
    }
#ifdef VPX_ABLATE
    p._p = dev_switch("VPX_C2_STAMP_BLOCK", -1);
#endif
    const long long per_xcd = ((long long)p.grid_m * p.n_tiles + 7) / 8;
    VPX_LAUNCH((cell2_kernel<Epi, ALLG>), dim3((unsigned)(per_xcd * 8)), dim3(512), C2_LDS, s, p, epi);
    return vpx_hip_last_error();
}

hipError_t launch_cell2(const Cell2Plan& plan, const ConvLSTMStepArgs& ea, void* h_sp, long long h_sp_bstride, hipStream_t s) {
    if (plan.qform && cell2x_selected()) {   // the eight-wave half tile (cell2x.hip): four waves per SIMD
        Cell2Plan p = plan;
        p.tiles_y = (p.H + 15) / 16;
        p.grid_m = p.B * p.tiles_x * p.tiles_y;
        return launch_cell2x(p, ea, h_sp, h_sp_bstride, s);
    }
    Cell2Epi epi{ea, reinterpret_cast<char*>(h_sp), h_sp_bstride};
    return launch_cell2_t<Cell2Epi, true>(plan, epi, s);
}

// ---- conv2: plain 3x3 'same' convolution of ONE split-format source (K = C channels in 16-channel stages) ----
// weight repack: element (out channel oc, in channel ic, tap) of the source tensor at w[ic * s_ic + (col0 + oc) * s_oc + tap],
// tap flipped for a data gradient -> the chunk layout of cell2_pack_kernel with n = g * 32 + j <-> oc = (n_tile * gpt + g) * 32 + j
__global__ void conv2_pack_kernel(const Conv2Pack pk, char* __restrict__ dst) {
    const long long total = (long long)pk.n_tiles * pk.chunks_total * (C2_WCHUNK / 2);  // bf16 elements
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(e & 7);
        long long r = e >> 3;
        const int n = (int)(r & 127); r >>= 7;
        const int khalf = (int)(r & 1); r >>= 1;
        const int part = (int)(r & 1); r >>= 1;
        const int q = (int)(r % 3); r /= 3;
        const int chunk = (int)(r % pk.chunks_total);
        const int n_tile = (int)(r / pk.chunks_total);
        const int stage = chunk / 3, dy = chunk - stage * 3;
        const int g = n >> 5, j = n & 31;
        const int oc = (n_tile * pk.gpt + g) * 32 + j;
        float v = 0.0f;
        if (g < pk.gpt && oc < pk.Co) {
            const int ic = stage * 16 + khalf * 8 + i;
            const int tap = pk.flip ? (2 - dy) * 3 + (2 - q) : dy * 3 + q;
            v = pk.w[(long long)ic * pk.s_ic + (long long)(pk.col0 + oc) * pk.s_oc + tap];
        }
        unsigned hi, lo;
        c2_split(v, hi, lo);
        reinterpret_cast<unsigned short*>(dst)[e] = (unsigned short)(part ? lo : hi);
    }
}

// q form of the same pack: [n_tile][step q][half][part][k group][n & 63][8 bf16], steps as in cell2_pack_q_kernel over the C / 16 stages
__global__ void conv2_pack_q_kernel(const Conv2Pack pk, int S, char* __restrict__ dst) {
    const long long total = (long long)pk.n_tiles * pk.chunks_total * (CQ_WCHUNK / 2);  // bf16 elements
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(e & 7);
        long long r = e >> 3;
        int n = (int)(r & 63); r >>= 6;
        const int kg = (int)(r & 3); r >>= 2;
        const int part = (int)(r & 1); r >>= 1;
        n += (int)(r & 1) * 64; r >>= 1;                         // half chunk: column tiles 0-3 | 4-7
        const int q = (int)(r % pk.chunks_total);
        const int n_tile = (int)(r / pk.chunks_total);
        const int p = q % 9, tsel = kg >> 1, khalf = kg & 1;
        const int stage = 2 * (q / 9) + cq_stage_of(p, tsel), t = cq_tap_of(p, tsel);
        const int g = n >> 5, j = n & 31;
        const int oc = (n_tile * pk.gpt + g) * 32 + j;
        float v = 0.0f;
        if (g < pk.gpt && oc < pk.Co && stage < S) {
            const int ic = stage * 16 + khalf * 8 + i;
            const int tap = pk.flip ? 8 - t : t;
            v = pk.w[(long long)ic * pk.s_ic + (long long)(pk.col0 + oc) * pk.s_oc + tap];
        }
        unsigned hi, lo;
        c2_split(v, hi, lo);
        reinterpret_cast<unsigned short*>(dst)[e] = (unsigned short)(part ? lo : hi);
    }
}

hipError_t launch_conv2_pack(const Conv2Pack& pk, void* dst, hipStream_t s) {
    const long long total = (long long)pk.n_tiles * pk.chunks_total * ((pk.qform ? CQ_WCHUNK : C2_WCHUNK) / 2);
    if (!ws_write_ok(dst, (size_t)total * 2, "weight pack (conv2_pack_kernel)")) return hipErrorInvalidValue;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (pk.qform) {
        const int S = (2 * pk.chunks_total) / 9;   // chunks_total = cell2_qchunks(S)
        VPX_LAUNCH(conv2_pack_q_kernel, dim3(blocks), dim3(256), 0, s, pk, S, reinterpret_cast<char*>(dst));
    } else {
        VPX_LAUNCH(conv2_pack_kernel, dim3(blocks), dim3(256), 0, s, pk, reinterpret_cast<char*>(dst));
    }
    return vpx_hip_last_error();
}

hipError_t launch_conv2(const Conv2Args& c, hipStream_t s) {
    Cell2Plan P{};
    P.B = c.B; P.H = c.H; P.W = c.W;
    P.tiles_x = (c.W + 15) / 16; P.tiles_y = (c.H + 31) / 32;
    P.n_groups = (c.Co + 31) / 32;
    P.n_tiles = conv2_tiles(c.Co); P.gpt = conv2_gpt(c.Co);
    P.nx = c.C / 16; P.nh = 0; P.hs_off = 0; P.chunks_total = c.qform ? cell2_qchunks(P.nx) : 3 * P.nx;
    P.qform = c.qform;
    P.seg[0] = Cell2Seg{c.src_sp, c.src_bstride, c.C, 0};
    P.seg[1] = Cell2Seg{c.src_sp, 0, c.C, 0};
    P.wpk = c.wpk;
    Conv2Epi epi{c.bias, c.Co, c.split, P.gpt, c.accumulate, c.out0, c.bstride0, c.ld0, 0, c.out1, c.bstride1, c.ld1, 0};
    return launch_cell2_t<Conv2Epi, false>(P, epi, s);
}

}  // namespace vpx

#ifdef VPX_ABLATE
extern "C" int vpx_dbg_cell2_trace(unsigned long long* out32768) {
    return (int)hipMemcpyFromSymbol(out32768, HIP_SYMBOL(vpx::c2_trace), sizeof(unsigned long long) * 8192 * 4);
}
extern "C" int vpx_dbg_cell2_stamps(unsigned long long* out512) {
    return (int)hipMemcpyFromSymbol(out512, HIP_SYMBOL(vpx::c2_stamps), sizeof(unsigned long long) * 512);
}
#endif
