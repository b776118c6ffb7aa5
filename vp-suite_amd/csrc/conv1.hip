// conv1.hip — c1: the ST-LSTM step's 1x1 convolutions (conv_last over mem = [c_new | m_new], predrnn.py:79-81; its adjoint in the
// backward pass; the decoupling tail's adapter, predrnn_v2.py:197-198) as a streaming kernel (round 4).
// These layers are 2 GFLOP over 50 MB: HBM-bound by a factor of 20 even at the bf16x3 matrix rate — but on the implicit-GEMM
// kernel (LDS-staged halo tiles, a weight stream per workgroup, one launch of 256 short workgroups) they took 42-44 us each,
// 4.9 % matrix-pipe use, five of them per cell step. Here nothing goes through LDS and nothing is synchronised:
//   * a wave keeps ITS share of the weights — Co / 4 columns x all K — as MFMA B fragments in registers for the whole kernel
//     (128 VGPRs: (Co, K) = (128, 256) | (256, 128); split into hi / lo bf16 once);
//   * the pixel fragment of a lane (pixel = lane & 15, k group = lane >> 4) is eight consecutive fp32 channels of one pixel: two
//     16-byte global loads, split in registers (v_cvt_pk_bf16_f32) — the input is read as plain fp32 NHWC, no operand format needed;
//     all K / 32 steps of a 16-pixel tile are in flight one tile ahead (16 loads per lane), the weights are the MFMA's row operand
//     so that a lane stores four consecutive output channels of its pixel with one 16-byte store;
//   * the four waves of a workgroup take the same pixels (the loads of three of them hit L1 / TA) and different columns.
// y[p][n] (+)= sum_c [x0 | x1][p][c] * w(n, c), fp32 in and out, products hi*hi + hi*lo + lo*hi (bf16x3), fp32 accumulation.
#include "cell2_dev.h"
#include "vpx_host.h"

namespace vpx {

// XSPLIT (round 6): the sources are already in the split operand format (the ST-LSTM gate stage writes c_new / m_new that way next to
// the fp32 tensors). The 32 bytes a lane loads per step — eight channels of a pixel — sit at the SAME address in both layouts (8 fp32 |
// 8 hi bf16 + 8 lo bf16), so the requests are unchanged and the two halves ARE the step's hi / lo fragments: the 16 fp32 -> (hi, lo)
// conversions per lane and step (~100 vector instructions against 6 MFMAs per column tile: the kernel was VALU-bound, 44-48 us for
// 50 MB) disappear.
template <int NTW, int KS, bool XSPLIT = false>   // column tiles (of 16) per wave, K = 32 steps
__global__ __launch_bounds__(256, 2) void c1_kernel(const C1Args a) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kg = lane >> 4;
    // B fragments: column n = (wave * NTW + t) * 16 + r16, k = ks * 32 + kg * 8 .. + 7
    bf16x8 bh[NTW][KS], bl[NTW][KS];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int n = (wave * NTW + t) * 16 + r16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            unsigned h[8], l[8];
            float wv[8];
            if (a.w_sc == 1 && n < a.Co) {   // channels contiguous: two 16-byte loads (eight scalar loads of 64 different lines each were TA-bound)
                const float* wp = a.w + (long long)n * a.w_sn + ks * 32 + kg * 8;
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp), w1 = *reinterpret_cast<const f32x4*>(wp + 4);
#pragma unroll
                for (int i = 0; i < 8; ++i) wv[i] = i < 4 ? w0[i] : w1[i - 4];
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int c = ks * 32 + kg * 8 + i;
                    wv[i] = n < a.Co ? a.w[(long long)n * a.w_sn + (long long)c * a.w_sc] : 0.0f;
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) c2_split(wv[i], h[i], l[i]);
            bh[t][ks] = __builtin_bit_cast(bf16x8, uint4{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)});
            bl[t][ks] = __builtin_bit_cast(bf16x8, uint4{l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)});
        }
    }
    const long long ntile = (a.npix + 15) / 16;
    const int ks0 = a.xc[0] / 32;   // steps served by the first source
    // A tile's 16 loads per lane are requested one tile AHEAD: raw[ks] is refilled for the next tile the moment this tile's step ks has
    // split it — the next tile's bytes travel while this one multiplies and stores (the first version requested a tile's loads, waited,
    // multiplied, stored, and only then turned to the next tile: four such round trips per workgroup, 34 us for 67 MB).
    f32x4 raw[KS][2];
    auto request = [&](long long tile, int ks) {
        const long long p = tile * 16 + r16;
        const bool first = ks < ks0;
        const float* src = (first ? a.x[0] + p * a.xld[0] + ks * 32 : a.x[1] + p * a.xld[1] + (ks - ks0) * 32) + kg * 8;
        if (tile < ntile && p < a.npix) { raw[ks][0] = *reinterpret_cast<const f32x4*>(src); raw[ks][1] = *reinterpret_cast<const f32x4*>(src + 4); }
        else { raw[ks][0] = f32x4{0.f, 0.f, 0.f, 0.f}; raw[ks][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    };
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) request(blockIdx.x, ks);
#pragma unroll 1
    for (long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        f32x4 acc[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 ah, al;
            if constexpr (XSPLIT) {
                ah = __builtin_bit_cast(bf16x8, raw[ks][0]);
                al = __builtin_bit_cast(bf16x8, raw[ks][1]);
            } else {
                unsigned h[8], l[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) c2_split(i < 4 ? raw[ks][0][i] : raw[ks][1][i - 4], h[i], l[i]);
                ah = __builtin_bit_cast(bf16x8, uint4{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)});
                al = __builtin_bit_cast(bf16x8, uint4{l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)});
            }
            request(tile + gridDim.x, ks);
            // the WEIGHTS are the MFMA's row operand: D[row = column 4 * kg + r of the tile][column = pixel r16] — a lane ends up with
            // four consecutive output channels of one pixel (one 16-byte store instead of four 4-byte ones)
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[t][ks], al, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[t][ks], ah, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[t][ks], ah, acc[t], 0, 0, 0);
            }
        }
        const long long q = tile * 16 + r16;
        if (q < a.npix) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                const int n = (wave * NTW + t) * 16 + 4 * kg;   // (Co, ysplit: multiples of 16 — c1_applicable)
                if (n >= a.Co) continue;
                const bool lo_half = n < a.ysplit;
                float* const yb = lo_half ? a.y[0] : a.y[1];
                const int ld = lo_half ? a.yld[0] : a.yld[1], nn = lo_half ? n : n - a.ysplit;
                f32x4* dst = reinterpret_cast<f32x4*>(yb + q * ld + nn);
                f32x4 v = acc[t];
                if (a.accumulate) { const f32x4 o = *dst; v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3]; }
                *dst = v;
            }
        }
    }
}

bool c1_applicable(const C1Args& a, int prec) {
    if (prec != VPX_PREC_BF16X3 || (g_experiment & 512)) return false;   // VPX_OPT_EXPERIMENT bit 9: the implicit-GEMM kernel (A/B runs, tests)
    const int K = a.xc[0] + a.xc[1];
    if ((a.xc[0] & 31) || (a.xc[1] & 31) || a.xc[0] < 32) return false;
    if (!((a.Co == 128 && K == 256) || (a.Co == 256 && K == 128) || (a.Co == 128 && K == 128))) return false;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!al16(a.x[0]) || (a.xld[0] & 3) || (a.xc[1] && (!al16(a.x[1]) || (a.xld[1] & 3)))) return false;
    if (a.w_sc == 1 && (!al16(a.w) || (a.w_sn & 3))) return false;
    // 16-byte stores of four consecutive output channels
    if ((a.ysplit & 3) || !al16(a.y[0]) || (a.yld[0] & 3) || (a.ysplit < a.Co && (!al16(a.y[1]) || (a.yld[1] & 3)))) return false;
    return a.npix > 0;
}

hipError_t launch_c1(const C1Args& a, hipStream_t s) {
    const int K = a.xc[0] + a.xc[1];
    const long long ntile = (a.npix + 15) / 16;
    const long long gmax = dev_switch("VPX_C1_GRID", 512);
    const unsigned grid = (unsigned)(ntile < gmax ? ntile : gmax);   // two workgroups per CU, one round: every workgroup reads the weights once
    if (a.x_split) {
        if (a.Co == 128 && K == 256) VPX_LAUNCH((c1_kernel<2, 8, true>), dim3(grid), dim3(256), 0, s, a);
        else if (a.Co == 256 && K == 128) VPX_LAUNCH((c1_kernel<4, 4, true>), dim3(grid), dim3(256), 0, s, a);
        else VPX_LAUNCH((c1_kernel<2, 4, true>), dim3(grid), dim3(256), 0, s, a);
        return vpx_hip_last_error();
    }
    if (a.Co == 128 && K == 256) VPX_LAUNCH((c1_kernel<2, 8>), dim3(grid), dim3(256), 0, s, a);
    else if (a.Co == 256 && K == 128) VPX_LAUNCH((c1_kernel<4, 4>), dim3(grid), dim3(256), 0, s, a);
    else VPX_LAUNCH((c1_kernel<2, 4>), dim3(grid), dim3(256), 0, s, a);
    return vpx_hip_last_error();
}

}  // namespace vpx
