// cell3.hip — fused ConvLSTM step for SMALL grids (small batch and / or 16x16 - 32x32 maps; conv_lstm_hzzone.py:59-68).
// The large-grid kernels tile N by 32 channels x 4 gates and M by 256-512 pixels: at B = 4 a 32x32 map gives 24-96 workgroups
// for 256 CUs, and the remedy so far was a K-split convolution (atomics, a memset) plus a pointwise gate kernel per step: two
// latency-bound launches, 38-46 us per cell step. This kernel cuts N finer instead:
//   * workgroup = (16x16-pixel tile, slice of 8 CHANNELS): its 32 MFMA columns are the four gates of those 8 channels, so the
//     LSTM update stays inside the workgroup; B = 4 on 32x32 maps x 96 channels -> 16 tiles x 12 slices = 192 workgroups;
//   * the slice's recurrent weights (9 taps x Ch x 32 columns, hi + lo bf16: 108 KiB at Ch = 96) are copied into LDS ONCE by
//     LDS-DMA and stay there for the whole K loop — no weight chunk ring, no chunk barriers;
//   * the input projection W_x * x_t is hoisted out of the recurrence for all T frames (one large launch of the
//     first-generation kernel, vpx_api.hip) and enters here through the epilogue, as on the K-split path before;
//   * h_{t-1} arrives in split-bf16 operand format (written by this kernel's own epilogue), its halo tile is staged per
//     16-channel stage by LDS-DMA into two buffers: copy of stage s+1 under the 27 MFMAs per wave of stage s (requesting ALL
//     stages into registers up front and feeding the buffers from there measured slower: 20.8 vs 18.7 us per launch);
//   * deterministic (no atomics): also the small-grid path when vpx_set_deterministic(1) had switched the K split off.
// Arithmetic = conv_gemm_kernel<EpiConvLSTM, bf16x3> / cell2_kernel (same operand split; fp32 summation order differs).
#include <stdlib.h>

#include "vpx_internal.h"

namespace vpx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int C3_HALO_W = 18;
constexpr int C3_NPOS = 18 * 18;               // halo positions of a 16x16 tile
constexpr int C3_PLANE_POS = 384;              // padded: 4 planes = 1536 pieces = 3 per thread
constexpr int C3_PLANE = C3_PLANE_POS * 16;
constexpr int C3_ABUF = 4 * C3_PLANE;          // 24576 B: planes [part * 2 + khalf][pos][8 bf16]
constexpr int C3_KSTEP = 2048;                 // weights of one k-step (tap x 16 channels): [part][khalf][32 columns][8 bf16]
constexpr int C3_MAX_CH = 96;                  // 54 k-steps = 108 KiB + two activation buffers = 156 KiB of the 160 KiB

__device__ const float c3_zero16[4] __attribute__((aligned(16))) = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void c3_dma16(const char* g, char* lds_wave_base) {   // see c2_dma16 (cell2.hip) for the why of the asm
    const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_wave_base;
    asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(g), "{m0}"(__builtin_amdgcn_readfirstlane(lds)) : "memory");
}
__device__ __forceinline__ int c3_px(int i) { return (i & 16) ? ((i + 14) & 15) : (i & 15); }   // column rotation of the odd row (bank spread, as c2_px)
__device__ __forceinline__ unsigned short c3_bf16_bits(float v) {
    __bf16 h = (__bf16)v;
    return __builtin_bit_cast(unsigned short, h);
}

// weight repack: reference OIHW [4Ch, Cin+Ch, 3, 3], recurrent columns only -> [slice][k-step = stage*9 + tap][part][khalf][n][8]
// with n = gate*8 + channel-in-slice
__global__ void cell3_pack_kernel(const Cell3Pack pk, char* __restrict__ dst) {
    const long long total = (long long)pk.n_slices * pk.nk * (C3_KSTEP / 2);
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(e & 7);
        long long r = e >> 3;
        const int n = (int)(r & 31); r >>= 5;
        const int khalf = (int)(r & 1); r >>= 1;
        const int part = (int)(r & 1); r >>= 1;
        const int kstep = (int)(r % pk.nk);
        const int slice = (int)(r / pk.nk);
        const int stage = kstep / 9, tap = kstep - stage * 9;
        const int g = n >> 3, ch = slice * 8 + (n & 7);
        const int row = pk.gate_pos[g] * pk.Ch + ch;
        const int col = pk.Cin + stage * 16 + khalf * 8 + i;
        const float v = pk.w[((long long)row * pk.Ct + col) * 9 + tap];
        const unsigned short h = c3_bf16_bits(v);
        const unsigned short l = c3_bf16_bits(v - __builtin_bit_cast(float, (unsigned)h << 16));
        reinterpret_cast<unsigned short*>(dst)[e] = part ? l : h;
    }
}

hipError_t launch_cell3_pack(const Cell3Pack& pk, void* dst, hipStream_t s) {
    const long long total = (long long)pk.n_slices * pk.nk * (C3_KSTEP / 2);
    if (!ws_write_ok(dst, (size_t)total * 2, "weight pack (cell3_pack_kernel)")) return hipErrorInvalidValue;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    VPX_LAUNCH(cell3_pack_kernel, dim3(blocks), dim3(256), 0, s, pk, reinterpret_cast<char*>(dst));
    return vpx_hip_last_error();
}

size_t cell3_packed_bytes(int Ch) { return (size_t)(Ch / 8) * (9 * Ch / 16) * C3_KSTEP; }

__global__ __launch_bounds__(512, 2) void cell3_kernel(const Cell3Args P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, hh = lane >> 5;

    // XCD-aware decode: the slices of one pixel tile (same halo tile) run back to back on one XCD
    const unsigned L = blockIdx.x;
    const long long total = (long long)P.B * P.tiles_x * P.tiles_y * P.n_slices;
    const long long per_xcd = (total + 7) / 8;
    const long long sidx = (long long)(L & 7) * per_xcd + (L >> 3);
    if ((long long)(L >> 3) >= per_xcd || sidx >= total) return;
    int mt = (int)(sidx / P.n_slices);
    const int slice = (int)(sidx - (long long)mt * P.n_slices);
    const int tx = mt % P.tiles_x;
    mt /= P.tiles_x;
    const int ty = mt % P.tiles_y;
    const int b = mt / P.tiles_y;
    const int x0 = tx * 16, y0 = ty * 16;
    const int Ch = P.ea.Ch;

    char* const Abuf = smem;
    char* const Wbuf = smem + 2 * C3_ABUF;
    const int S = P.h_sp ? Ch / 16 : 0;   // K stages; no recurrent operand (zero state at t = 0): the step is its epilogue only

    // ---- epilogue operands first: a lane will own four channels of one pixel (slot = MFMA row); the input projection, bias, cell
    //      state and peepholes do not depend on the contraction, so their loads go out now and land under it ----
    const int slot = lane >> 1, c4 = (lane & 1) * 4;
    const int py = y0 + 2 * wave + (slot >> 4), pxx = x0 + c3_px(slot);   // (whole tiles only: always inside the image)
    const unsigned uCh = (unsigned)Ch;
    const unsigned ch = (unsigned)(slice * 8 + c4);
    const unsigned pix = (unsigned)(py * P.W + pxx);
    const ConvLSTMStepArgs& a = P.ea;
    const size_t e = ((size_t)b * P.H * P.W + pix) * uCh + ch;
    const size_t pe = (size_t)pix * uCh + ch;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 z[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        z[g] = P.pre ? *reinterpret_cast<const f32x4*>(P.pre + (size_t)b * P.pre_bstride + (size_t)pix * 4 * uCh + a.gate_pos[g] * uCh + ch) : zero;
        if (a.bias) z[g] += *reinterpret_cast<const f32x4*>(a.bias + a.gate_pos[g] * uCh + ch);
    }
    const f32x4 cp = a.c_in ? *reinterpret_cast<const f32x4*>(a.c_in + e) : zero;
    const f32x4 pwi = a.wci ? *reinterpret_cast<const f32x4*>(a.wci + pe) : zero;
    const f32x4 pwf = a.wci ? *reinterpret_cast<const f32x4*>(a.wcf + pe) : zero;
    const f32x4 wo = a.wco ? *reinterpret_cast<const f32x4*>(a.wco + pe) : zero;

    f32x16 acc0, acc1;   // two accumulators (even / odd taps): consecutive MFMAs never depend on each other
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

    if (S > 0) {
        // this thread's three pieces of an activation stage
        int pixoff[3], choff[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int piece = tid + 512 * u;
            const int plane = piece / C3_PLANE_POS, pos = piece - plane * C3_PLANE_POS;
            const int hy = pos / C3_HALO_W, hx = pos - hy * C3_HALO_W;
            const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            const bool ok = pos < C3_NPOS && gy >= 0 && gy < P.H && gx >= 0 && gx < P.W;
            pixoff[u] = ok ? gy * P.W + gx : -1;
            choff[u] = (plane & 1) * 32 + (plane >> 1) * 16;   // plane = part*2 + khalf; split pixel row: [group of 8][hi 16 B | lo 16 B]
        }
        const char* const hb = P.h_sp + (size_t)b * P.h_bstride;
        const unsigned prow = (unsigned)Ch * 4u;
        auto issue_A = [&](int s, int buf) {
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const char* src = pixoff[u] >= 0 ? hb + (size_t)((unsigned)pixoff[u] * (unsigned long long)prow) + s * 64 + choff[u]
                                                 : reinterpret_cast<const char*>(c3_zero16);
                c3_dma16(src, Abuf + buf * C3_ABUF + (512 * u + wave * 64) * 16);
            }
        };
        issue_A(0, 0);
        // the slice's weights: nk * 128 pieces, lane-linear
        const int npieces = P.nk * (C3_KSTEP / 16);
        const char* const wsl = P.wpk + (size_t)slice * P.nk * C3_KSTEP;
        int nw = 0;   // pieces this wave issues (wave-uniform bounds: npieces is a multiple of 64), in k-step order
        for (int base = wave * 64; base < npieces; base += 512, ++nw)
            c3_dma16(wsl + (size_t)(base + lane) * 16, Wbuf + base * 16);
        // stage 0 reads k-steps 0-8 = pieces 0 .. 1151 = this thread's first three: wait for those and the activation tile only
        // (in-order vmcnt: the rest of the weights lands under stage 0's MFMAs; the stage-end wait below is vmcnt(0))
        switch (nw > 3 ? nw - 3 : 0) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
            case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");

        // fragment addressing: MFMA row i = pixel (tile row 2*wave + (i >> 4), column c3_px(i)), k half = hh
        const int a_lane = hh * C3_PLANE + ((2 * wave + (j >> 4)) * C3_HALO_W + c3_px(j)) * 16;
        const int w_lane = hh * 512 + j * 16;
        for (int s = 0; s < S; ++s) {
            if (s + 1 < S) issue_A(s + 1, (s + 1) & 1);
            const char* A = Abuf + (s & 1) * C3_ABUF + a_lane;
            const char* Wk = Wbuf + s * 9 * C3_KSTEP + w_lane;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int off = ((tap / 3) * C3_HALO_W + (tap % 3)) * 16;
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(A + off);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(A + 2 * C3_PLANE + off);
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(Wk + tap * C3_KSTEP);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(Wk + tap * C3_KSTEP + 1024);
                if (tap & 1) {
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc1, 0, 0, 0);
                } else {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc0, 0, 0, 0);
                }
            }
            // stage s+1 has landed and every wave is done with buffer s & 1 (the copy of stage s+2 goes there; after the last
            // stage the buffers become the epilogue's transposition space)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }

    // ---- epilogue: accumulators -> this wave's 4 KiB of the (now idle) first activation buffer -> a lane owns four channels of a
    //      pixel: pre-activation = recurrent sum + hoisted input projection + bias (+ peepholes), then the state update ----
    float* ldsf = reinterpret_cast<float*>(Abuf + wave * 4096);
#pragma unroll
    for (int r = 0; r < 16; ++r) ldsf[((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + j] = acc0[r] + acc1[r];
    // (LDS operations of one wave execute in order)
#pragma unroll
    for (int g = 0; g < 4; ++g) z[g] += *reinterpret_cast<const f32x4*>(ldsf + slot * 32 + g * 8 + c4);
    z[0] += pwi * cp;
    z[1] += pwf * cp;
    f32x4 i4, f4, g4, o4, cn, hn;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        i4[q] = sigmoid_f(z[0][q]);
        f4[q] = sigmoid_f(z[1][q]);
        g4[q] = tanh_f(z[2][q]);
        cn[q] = lstm_c(f4[q], cp[q], i4[q], g4[q]);
        o4[q] = sigmoid_f(z[3][q] + wo[q] * cn[q]);
        hn[q] = o4[q] * tanh_f(cn[q]);
    }
    *reinterpret_cast<f32x4*>(a.c_out + e) = cn;
    if (a.h_out) *reinterpret_cast<f32x4*>(a.h_out + (size_t)b * a.h_bstride + pe) = hn;   // (null: VPX_FLAG_OUT_SPLIT — the sequence goes out in operand format only)
    if (a.gates) {
        float* gs = a.gates + ((size_t)b * P.H * P.W + pix) * 4 * uCh + ch;
        *reinterpret_cast<f32x4*>(gs) = i4;
        *reinterpret_cast<f32x4*>(gs + uCh) = f4;
        *reinterpret_cast<f32x4*>(gs + 2 * uCh) = g4;
        *reinterpret_cast<f32x4*>(gs + 3 * uCh) = o4;
    }
    if (P.h_sp_out) {
        unsigned h[4], l[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned short hb16 = c3_bf16_bits(hn[q]);
            h[q] = hb16;
            l[q] = c3_bf16_bits(hn[q] - __builtin_bit_cast(float, (unsigned)hb16 << 16));
        }
        char* dst = P.h_sp_out + (size_t)b * P.h_sp_out_bstride + (size_t)pix * uCh * 4 + (ch >> 3) * 32 + (ch & 7) * 2;
        *reinterpret_cast<uint2*>(dst) = uint2{h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
        *reinterpret_cast<uint2*>(dst + 16) = uint2{l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
    }
}

// bf16x3, 3x3, recurrent channels in whole 16-channel stages that fit the LDS, whole 16x16 tiles. VPX_CELL3=0 disables.
int cell3_mode() {
    if (g_cell3_mode < 0) g_cell3_mode = dev_switch("VPX_CELL3", 1) ? 1 : 0;
    return g_cell3_mode;
}
bool cell3_applicable(const vpx_convlstm_desc* d) {
    if (!cell3_mode() || d->precision != VPX_PREC_BF16X3 || d->kh != 3 || d->kw != 3) return false;
    if ((d->Ch & 15) || d->Ch > C3_MAX_CH || (d->H & 15) || (d->W & 15)) return false;
    return true;
}

hipError_t launch_cell3(const Cell3Args& args, hipStream_t s) {
    Cell3Args P = args;
    P.tiles_x = P.W / 16; P.tiles_y = P.H / 16;
    P.n_slices = P.ea.Ch / 8;
    P.nk = 9 * P.ea.Ch / 16;
    const int lds = 2 * C3_ABUF + P.nk * C3_KSTEP;
    static int attr_lds = 0;
    if (lds > attr_lds) {
        hipError_t e = vpx_func_attr(reinterpret_cast<const void*>(&cell3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * C3_ABUF + (9 * C3_MAX_CH / 16) * C3_KSTEP);
        if (e != hipSuccess) return e;
        if (!g_dry_run) attr_lds = 2 * C3_ABUF + (9 * C3_MAX_CH / 16) * C3_KSTEP;   // (a dry run sets nothing: a real launch may follow it)
    }
    const long long total = (long long)P.B * P.tiles_x * P.tiles_y * P.n_slices;
    const long long per_xcd = (total + 7) / 8;
    VPX_LAUNCH(cell3_kernel, dim3((unsigned)(per_xcd * 8)), dim3(512), lds, s, P);
    return vpx_hip_last_error();
}

}  // namespace vpx
