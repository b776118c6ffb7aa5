// lstm_bwd.hip — BPTT kernels of the ConvLSTM block (the reference gets these from autograd over the unrolled python
// loop, conv_lstm_hzzone.py:52-70; here they are explicit):
//   convlstm_gate_bwd_kernel  HBM-bound: d(pre-activations) from (dh, dc), carries dc, reduces peephole grads over b
//   wgrad_kernel              dW partials: M = 64 gate rows, N = 64 input channels, all taps of a tap group at once,
//                             K = pixels of the (t, b, tile) work items of one K-slice; fp32 MFMA 32x32x2
//   wgrad_tg_kernel           the same contraction on split-bf16 operands (bf16x3 / bf16), 8 waves = rows x channel halves x
//                             tap groups, transposing LDS reads; two item buffers (up to 3x3) or one (5x5)
//   wgrad_bf16x3_kernel       older bf16 forms: 4-wave (ragged / unaligned operands, 1x1) and 8-wave 128-row (7x7 and larger)
//   wgrad_reduce_kernel       sums the K-slices and writes the reference's OIHW layout
//   colsum_kernel             bias gradient
#include "vpx_internal.h"

namespace vpx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned short gb_bf16_bits(float v) {   // round to nearest even, the split of conv_gemm.hip / cell2.hip
    __bf16 h = (__bf16)v;
    return __builtin_bit_cast(unsigned short, h);
}

__global__ __launch_bounds__(256) void convlstm_gate_bwd_kernel(const GateBwdArgs a) {
    __shared__ float db_vals[4][256];  // per-thread bias-gradient sums of this block (only when a.db_partial)
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int HW = a.HW, Ch = a.Ch;
    const bool active = idx < HW * Ch;
    const int pix = active ? idx / Ch : 0, ch = active ? idx - pix * Ch : 0;
    float sb0 = 0.f, sb1 = 0.f, sb2 = 0.f, sb3 = 0.f;
    if (active) {
    const size_t pc = (size_t)pix * Ch + ch;
    float wci = 0.f, wcf = 0.f, wco = 0.f;
    const bool peep = a.wci != nullptr;
    if (peep) { wci = a.wci[pc]; wcf = a.wcf[pc]; wco = a.wco[pc]; }
    float dpi = 0.f, dpf = 0.f, dpo = 0.f;
    // batch slice of this block row (b_slices > 1 only without peephole gradients: their sum over b has a single owner)
    const int bper = (a.B + (int)gridDim.y - 1) / (int)gridDim.y;
    const int b_lo = (int)blockIdx.y * bper, b_hi = b_lo + bper < a.B ? b_lo + bper : a.B;
    for (int b = b_lo; b < b_hi; ++b) {
        const size_t s = ((size_t)b * HW + pix) * Ch + ch;
        const float* gs = a.gates + ((size_t)b * HW + pix) * 4 * Ch + ch;
        const float i_ = gs[0], f_ = gs[Ch], g_ = gs[2 * Ch], o_ = gs[3 * Ch];
        const float cn = a.c_t[s];
        const float cp = a.c_prev ? a.c_prev[s] : 0.0f;
        float dht = a.dh_in ? a.dh_in[s] : 0.0f;
        if (a.dout) dht += a.dout[(size_t)b * a.dout_bstride + pc];
        const float tc = tanhf(cn);
        const float dao = dht * tc * o_ * (1.0f - o_);
        float dcn = (a.dc_in ? a.dc_in[s] : 0.0f) + dht * o_ * (1.0f - tc * tc);
        if (peep) { dcn += dao * wco; dpo += dao * cn; }
        const float dai = dcn * g_ * i_ * (1.0f - i_);
        const float daf = dcn * cp * f_ * (1.0f - f_);
        const float dag = dcn * i_ * (1.0f - g_ * g_);
        float dcp = dcn * f_;
        if (peep) { dcp += dai * wci + daf * wcf; dpi += dai * cp; dpf += daf * cp; }
        a.dc_out[s] = dcp;
        if (a.dG) {
            float* dg = a.dG + ((size_t)b * HW + pix) * 4 * Ch + ch;
            dg[a.gate_pos[0] * Ch] = dai;
            dg[a.gate_pos[1] * Ch] = daf;
            dg[a.gate_pos[2] * Ch] = dag;
            dg[a.gate_pos[3] * Ch] = dao;
        }
        if (a.dG_sp) {
            // operand format of the data- and weight-gradient kernels: [pixel][4Ch / 8][8 hi bf16 | 8 lo bf16]. Lanes 2k, 2k+1 hold
            // channels ch, ch+1 (Ch even): the even lane stores the hi pair, the odd lane the lo pair — 4-byte stores, and a
            // wave's stores of one gate cover a contiguous run of the pixel row
            char* const row = a.dG_sp + ((size_t)b * HW + pix) * 16 * Ch;
            const float dv[4] = {dai, daf, dag, dao};
            const int che = ch & ~1;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const unsigned short h = gb_bf16_bits(dv[g]);
                const unsigned hi = h, lo = gb_bf16_bits(dv[g] - __builtin_bit_cast(float, (unsigned)h << 16));
                const unsigned nhi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
                const unsigned nlo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, 0xB1, 0xF, 0xF, true);
                const unsigned n = (unsigned)(a.gate_pos[g] * Ch + che);
                const unsigned off = (n >> 3) * 32 + (n & 7) * 2 + ((ch & 1) ? 16 : 0);
                *reinterpret_cast<unsigned*>(row + off) = (ch & 1) ? (nlo | (lo << 16)) : (hi | (nhi << 16));
            }
        }
        sb0 += dai; sb1 += daf; sb2 += dag; sb3 += dao;
    }
    if (peep && a.dwci) {  // single owner per (slice, pix, ch): plain accumulate over the time steps
        const size_t po = (size_t)blockIdx.y * (size_t)a.peep_slice_stride + pc;
        a.dwci[po] += dpi;
        a.dwcf[po] += dpf;
        a.dwco[po] += dpo;
    }
    }  // active
    if (a.db_partial) {
        // bias gradient, bit-reproducible: thread k of the block owns channel (start + k) % Ch; output (gate, c) is the sum
        // over the block's positions of channel c in increasing k — a fixed order, no atomics; one plain row per block
        db_vals[0][threadIdx.x] = sb0; db_vals[1][threadIdx.x] = sb1;   // inactive threads hold zeros
        db_vals[2][threadIdx.x] = sb2; db_vals[3][threadIdx.x] = sb3;
        __syncthreads();
        const int start_ch = (int)(((long long)blockIdx.x * 256) % Ch);
        for (int n = threadIdx.x; n < 4 * Ch; n += 256) {
            const int gl = n / Ch, c = n - gl * Ch;
            float acc = 0.f;
            for (int k = (c - start_ch + Ch) % Ch; k < 256; k += Ch) acc += db_vals[gl][k];
            a.db_partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 * Ch + a.gate_pos[gl] * Ch + c] = acc;
        }
    }
}

hipError_t launch_gate_bwd(const GateBwdArgs& a, hipStream_t s) {
    const bool one_owner = a.dwci != nullptr && a.peep_slice_stride == 0;
    VPX_LAUNCH(convlstm_gate_bwd_kernel, dim3(gate_bwd_blocks(a.HW, a.Ch), gate_bwd_slices(a.HW, a.Ch, a.B, one_owner)), dim3(256), 0, s, a);
    return vpx_hip_last_error();
}

__global__ __launch_bounds__(256) void peep_reduce_kernel(const float* __restrict__ p0, const float* __restrict__ p1, const float* __restrict__ p2,
                                                          float* __restrict__ d0, float* __restrict__ d1, float* __restrict__ d2, int slices,
                                                          long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* p = blockIdx.y == 0 ? p0 : (blockIdx.y == 1 ? p1 : p2);
    float* d = blockIdx.y == 0 ? d0 : (blockIdx.y == 1 ? d1 : d2);
    float acc = 0.f;
    for (int s = 0; s < slices; ++s) acc += p[(size_t)s * n + i];   // fixed order: bit-reproducible
    d[i] = acc;
}

hipError_t launch_peep_reduce(const float* p0, const float* p1, const float* p2, float* d0, float* d1, float* d2, int slices, long long n,
                              hipStream_t s) {
    VPX_LAUNCH(peep_reduce_kernel, dim3((unsigned)((n + 255) / 256), 3), dim3(256), 0, s, p0, p1, p2, d0, d1, d2, slices, n);
    return vpx_hip_last_error();
}

// ---------------------------------------------------------------------------------------------------------------
// Column sums of a [rows][cols] matrix (bias gradients), bit-reproducible: level 1 sums row ranges into partial[block][cols]
// (within a block the threads of one column add their strided rows, then a fixed-order LDS combine), level 2 adds the
// block rows in order. Optional LeakyReLU': with `y` given, the summed (and, if `scaled` is given, stored) value is
// m * (y > 0 ? 1 : slope) — the glue's activation derivative from the sign of the forward output.
// scaled_sp (V = 4, cols % 8 == 0): the scaled matrix once more in the split-bf16 operand format — per row, per group of 8 columns
// [8 hi bf16 | 8 lo bf16]; a thread's four columns are one 8-byte half of each (the glue's data gradient then runs on convq)
template <int V>   // V = 4: columns handled as float4 groups (cols % 4 == 0 and 16-byte aligned operands), V = 1: scalar
__global__ __launch_bounds__(256) void colsum_l1_kernel(const float* __restrict__ m, const float* __restrict__ y, float slope,
                                                        float* __restrict__ scaled, float* __restrict__ partial, long long rows,
                                                        int cols, long long rows_per_block, char* __restrict__ scaled_sp) {
    typedef float vec __attribute__((ext_vector_type(V)));
    __shared__ vec comb[256];
    const long long r0 = blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    const int vcols = cols / V;
    for (int c0 = 0; c0 < vcols; c0 += 256) {
        const int cw = vcols - c0 < 256 ? vcols - c0 : 256;
        const int rpp = 256 / cw;                    // rows handled per pass
        const int t = threadIdx.x, col = c0 + t % cw, roff = t / cw;
        vec acc = 0.f;
        if (roff < rpp)
            for (long long r = r0 + roff; r < r1; r += rpp) {
                const size_t e = (size_t)r * vcols + col;
                vec v = reinterpret_cast<const vec*>(m)[e];
                if (y) {
                    const vec yv = reinterpret_cast<const vec*>(y)[e];
                    if constexpr (V == 1) v *= yv > 0.0f ? 1.0f : slope;
                    else
#pragma unroll
                        for (int i = 0; i < V; ++i) v[i] *= yv[i] > 0.0f ? 1.0f : slope;
                    if (scaled) reinterpret_cast<vec*>(scaled)[e] = v;
                }
                if constexpr (V == 4) {
                    if (scaled_sp) {
                        unsigned short h[4], l[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const __bf16 hb = (__bf16)v[i];
                            h[i] = __builtin_bit_cast(unsigned short, hb);
                            const __bf16 lb = (__bf16)(v[i] - __builtin_bit_cast(float, (unsigned)h[i] << 16));
                            l[i] = __builtin_bit_cast(unsigned short, lb);
                        }
                        char* grp = scaled_sp + ((size_t)r * vcols + col) / 2 * 32 + (col & 1) * 8;   // vcols is even: (r * vcols + col) / 2 = the row's 8-column group
                        *reinterpret_cast<uint2*>(grp) = uint2{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
                        *reinterpret_cast<uint2*>(grp + 16) = uint2{(unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16)};
                    }
                }
                acc += v;
            }
        comb[t] = acc;
        __syncthreads();
        if (t < cw) {
            vec sum = 0.f;
            for (int k = 0; k < rpp; ++k) sum += comb[k * cw + t];
            if (partial) reinterpret_cast<vec*>(partial)[(size_t)blockIdx.x * vcols + c0 + t] = sum;
        }
        __syncthreads();
    }
}
// level 2: 16 columns per block, the partial rows in 16 interleaved groups (thread g sums rows g, g+16, ...), then a
// fixed-order combine — bit-reproducible, and 16 x cols/16 threads instead of one serial walk per column (the serial form
// took 0.8 ms for the 4096 x 256 partials of a ConvLSTM bias gradient: 10 ms of a 136 ms training step)
__global__ __launch_bounds__(256) void colsum_l2_kernel(const float* __restrict__ partial, float* __restrict__ out, int nrows, int cols) {
    __shared__ float comb[16][17];
    const int cx = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cx;
    float sum = 0.f;
    if (c < cols) {   // four independent partial sums per thread (rows g + 16k, k mod 4), fixed combination order
        float p[4] = {0.f, 0.f, 0.f, 0.f};
        int r = g;
        for (; r + 48 < nrows; r += 64) {
#pragma unroll
            for (int k = 0; k < 4; ++k) p[k] += partial[(size_t)(r + 16 * k) * cols + c];
        }
        for (int k = 0; r < nrows; r += 16, ++k) p[k] += partial[(size_t)r * cols + c];
        sum = (p[0] + p[1]) + (p[2] + p[3]);
    }
    comb[g][cx] = sum;
    __syncthreads();
    if (g == 0 && c < cols) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) tot += comb[k][cx];
        out[c] = tot;
    }
}

hipError_t launch_colsum(const float* m, const float* y, float slope, float* scaled, float* out, float* partial_ws,
                         long long rows, int cols, hipStream_t s, char* scaled_sp) {
    int blocks = COLSUM_BLOCKS;
    if (rows / 32 < blocks) blocks = (int)(rows / 32 > 0 ? rows / 32 : 1);   // at least 32 rows per level-1 block
    const long long rpb = (rows + blocks - 1) / blocks;
    blocks = (int)((rows + rpb - 1) / rpb);
    const bool v4 = (cols & 3) == 0 && (((uintptr_t)m | (uintptr_t)y | (uintptr_t)scaled | (uintptr_t)partial_ws) & 15) == 0;
    if (out && !ws_write_ok(partial_ws, (size_t)blocks * cols * sizeof(float), "column-sum partials (colsum_l1_kernel)")) return hipErrorInvalidValue;
    if (scaled && !ws_write_ok(scaled, (size_t)rows * cols * sizeof(float), "scaled gradient (colsum_l1_kernel)")) return hipErrorInvalidValue;
    if (scaled_sp && (!v4 || (cols & 7) || ((uintptr_t)scaled_sp & 15))) return hipErrorInvalidValue;   // the split copy needs whole 8-column groups
    if (scaled_sp && !ws_write_ok(scaled_sp, (size_t)rows * cols * sizeof(float), "scaled gradient, split format (colsum_l1_kernel)")) return hipErrorInvalidValue;
    if (v4) VPX_LAUNCH(colsum_l1_kernel<4>, dim3(blocks), dim3(256), 0, s, m, y, slope, scaled, out ? partial_ws : nullptr, rows, cols, rpb, scaled_sp);
    else VPX_LAUNCH(colsum_l1_kernel<1>, dim3(blocks), dim3(256), 0, s, m, y, slope, scaled, out ? partial_ws : nullptr, rows, cols, rpb, (char*)nullptr);
    if (out) VPX_LAUNCH(colsum_l2_kernel, dim3((cols + 15) / 16), dim3(256), 0, s, partial_ws, out, blocks, cols);
    return vpx_hip_last_error();
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of the EF glue's few-channel layers (1 -> 16 and 16 -> 1 around the image; stride 1): Co*C*k*k <= 144 sums over
// every pixel of the batch — no matrix shape worth an MFMA tile (the MFMA kernels ran them at 15-25x their HBM time: 64-wide
// tiles for 1 or 16 useful rows / columns). Here: one thread per pixel (grid stride), all sums of its output-channel group in
// registers, fp32 FMA; wave reduction by shuffles, block partials, then colsum_l2_kernel — fixed order, bit-reproducible.
template <int COB, int C, int K>
__global__ __launch_bounds__(256) void wgrad_small_kernel(const float* __restrict__ dy, const float* __restrict__ x, int N, int H, int W,
                                                          int CO, int pad, float* __restrict__ partial) {
    constexpr int KK = K * K, ACC = COB * C * KK;
    float acc[ACC];
#pragma unroll
    for (int a = 0; a < ACC; ++a) acc[a] = 0.f;
    const long long npix = (long long)N * H * W;
    const int co0 = blockIdx.y * COB;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
        const int xx = (int)(p % W);
        const long long r = p / W;
        const int yy = (int)(r % H);
        const long long img = (r / H) * H;   // row index of the image's first row
        float g[COB];
#pragma unroll
        for (int i = 0; i < COB; ++i) g[i] = dy[p * CO + co0 + i];
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int iy = yy + ky - pad, ix = xx + kx - pad;
                const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
                const float* src = x + ((img + iy) * W + ix) * C;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float v = in ? src[c] : 0.f;
#pragma unroll
                    for (int i = 0; i < COB; ++i) acc[(i * C + c) * KK + ky * K + kx] += g[i] * v;
                }
            }
    }
    __shared__ float red[4][ACC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < ACC; ++a) {
        float v = acc[a];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) red[wave][a] = v;
    }
    __syncthreads();
    const int cols = CO * C * KK;
    for (int a = threadIdx.x; a < ACC; a += 256)
        partial[(size_t)blockIdx.x * cols + co0 * C * KK + a] = (red[0][a] + red[1][a]) + (red[2][a] + red[3][a]);
}

bool wgrad_small_applicable(int Co, int C, int kh, int kw, int stride, int pad) {
    static int env = -1;   // VPX_WGRAD_SMALL=0: MFMA kernels for these layers too (experiments)
    if (env < 0) env = dev_switch("VPX_WGRAD_SMALL", 1);
    if (!env || stride != 1 || kh != kw) return false;
    if (kh == 3 && pad == 1) return (C == 1 && Co % 16 == 0 && Co <= 64) || (C == 3 && Co % 4 == 0 && Co <= 64);
    if (kh == 1 && pad == 0) return C == 16 && (Co == 1 || Co == 3);
    return false;
}

// dw[Co][C][k][k] = sum over (n, y, x) dy[n,y,x,co] * x[n, y + ky - pad, x + kx - pad, c]; partial_ws: WGRAD_SMALL_BLOCKS * Co*C*k*k floats
hipError_t launch_wgrad_small(const float* dy, const float* x, int N, int H, int W, int Co, int C, int k, int pad, float* partial_ws,
                              float* dw, hipStream_t s) {
    const long long npix = (long long)N * H * W;
    int blocks = WGRAD_SMALL_BLOCKS;
    if ((npix + 255) / 256 < blocks) blocks = (int)((npix + 255) / 256);
    const int cols = Co * C * k * k;
    if (!ws_write_ok(partial_ws, (size_t)blocks * cols * sizeof(float), "weight-gradient partials (wgrad_small_kernel)")) return hipErrorInvalidValue;
    if (k == 3 && C == 1) VPX_LAUNCH((wgrad_small_kernel<16, 1, 3>), dim3(blocks, Co / 16), dim3(256), 0, s, dy, x, N, H, W, Co, pad, partial_ws);
    else if (k == 3 && C == 3) VPX_LAUNCH((wgrad_small_kernel<4, 3, 3>), dim3(blocks, Co / 4), dim3(256), 0, s, dy, x, N, H, W, Co, pad, partial_ws);
    else if (k == 1 && Co == 1) VPX_LAUNCH((wgrad_small_kernel<1, 16, 1>), dim3(blocks, 1), dim3(256), 0, s, dy, x, N, H, W, Co, pad, partial_ws);
    else if (k == 1 && Co == 3) VPX_LAUNCH((wgrad_small_kernel<3, 16, 1>), dim3(blocks, 1), dim3(256), 0, s, dy, x, N, H, W, Co, pad, partial_ws);
    else return hipErrorInvalidValue;
    VPX_LAUNCH(colsum_l2_kernel, dim3((cols + 15) / 16), dim3(256), 0, s, partial_ws, dw, blocks, cols);
    return vpx_hip_last_error();
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient. Work item = (t, b, spatial tile); slice `s` (see wg_block) owns items slice, slice + n_slices, ...
// LDS: dG tile [128 px][64 rows] (32 KiB) + activation halo tile [halo positions][64 ch].
// Each wave owns a 32 (rows) x 32 (channels) output block for up to WG_MAXT taps: acc[tap] += dG^T (px-contracted) A_tap.
// ---------------------------------------------------------------------------------------------------------------
// out-of-range vectors of the unconditional loads read this instead (no mask, no select afterwards)
__device__ const float wg_zero16[4] __attribute__((aligned(16))) = {0.f, 0.f, 0.f, 0.f};

// activation pixel (gy, gx) of the map the taps slide over -> element offset (in pixels) inside the source image, or false
__device__ __forceinline__ bool wg_apix(const WgradArgs& a, int gy, int gx, long long& pix) {
    if (a.a_sub) {
        if (gy < 0 || gy >= a.a_Hs || gx < 0 || gx >= a.a_Ws) return false;
        pix = (long long)(gy * a.a_sy + a.a_oy) * a.a_Wfull + (gx * a.a_sx + a.a_ox);
        return true;
    }
    if (gy < 0 || gy >= a.H || gx < 0 || gx >= a.W) return false;
    pix = (long long)gy * a.W + gx;
    return true;
}

// activation source of column half `h` for item (t, b): null when the half is unused or its operand absent (zero hidden
// state at t = 0 contributes nothing); C = channels per pixel of that source
__device__ __forceinline__ const float* wg_half_src(const WgradArgs& a, const WgradCHalf& h, int t, int b, int& C) {
    C = h.seg == 0 ? a.Cin : a.Ch;
    if (h.cn == 0) return nullptr;
    if (h.seg == 0) return a.x + (size_t)b * a.x_bstride + (size_t)t * a.x_tstride;
    if (t > 0) return a.hseq + (size_t)b * a.h_bstride + (size_t)(t - 1) * a.h_tstride;
    return a.h0 ? a.h0 + (size_t)b * a.HW * a.Ch : nullptr;
}

// Logical block (bx = row tile x column tile, slice) of hardware block blockIdx.x. Consecutive hardware blocks go to the
// eight XCDs round robin; the workgroups of one K slice read the same dG / activation tiles (every row tile needs the
// slice's activation tiles, every column tile its dG tiles), so a slice's blocks are given to ONE XCD, back to back, and
// meet in its L2. false: padding block of the rounded-up launch.
__device__ __forceinline__ bool wg_block(const WgradArgs& a, int& bx, int& slice) {
    const long long total = (long long)a.grid_x * a.grid_slices;
    const long long per_xcd = (total + 7) / 8;
    const unsigned L = blockIdx.x;
    const long long v = (long long)(L & 7) * per_xcd + (L >> 3);
    if ((long long)(L >> 3) >= per_xcd || v >= total) return false;
    slice = (int)(v / a.grid_x);
    bx = (int)(v - (long long)slice * a.grid_x);
    return true;
}

template <int MAXT>  // MAXT = exact number of taps this launch handles (branch-free MFMA block)
__global__ __launch_bounds__(NTHREADS, 2) void wgrad_kernel(const WgradArgs a, const int tap_base) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int wn = wave >> 1, wc = wave & 1;
    int bx, slice;
    if (!wg_block(a, bx, slice)) return;
    const int n_ct = a.n_ctiles;
    const int ct_id = bx % n_ct;
    const int nt_id = bx / n_ct;
    const WgradCTile ct = a.ct[ct_id];
    const int tap0 = tap_base + blockIdx.z * MAXT;
    const int halo_w = TILE_W + a.kw - 1, halo_h = TILE_H + a.kh - 1, npos = halo_w * halo_h;
    const int ph = a.use_org ? -a.org_y : a.kh / 2, pw = a.use_org ? -a.org_x : a.kw / 2;
    float* G_lds = reinterpret_cast<float*>(smem);                 // [128][64]
    float* A_lds = reinterpret_cast<float*>(smem + 128 * 64 * 4);  // [npos][64]

    f32x16 acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // per-tap LDS offsets (in floats) of the shifted activation pixel, relative to (py*halo_w + px)
    int tapoff[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        const int tp = tap0 + t;
        const int dy = tp / a.kw, dx = tp - dy * a.kw;
        tapoff[t] = (dy * halo_w + dx) * 64;
    }

    const int tiles = a.tiles_x * a.tiles_y;
    const long long n_items = (long long)a.T * a.B * tiles;
    const int n0 = nt_id * 64;  // first gate row of this workgroup
    const int myhalf = (tid >> 3) & 1;  // the vector staging loop keeps a thread on one 4-channel column (v & 15 == tid & 15)
    const WgradCHalf myh = ct.h[myhalf];
    for (long long w = slice; w < n_items; w += a.grid_slices) {
        const int tile = (int)(w % tiles);
        const long long tb = w / tiles;
        const int b = (int)(tb % a.B);
        const int t = (int)(tb / a.B);
        const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        const int y0 = ty * TILE_H, x0 = tx * TILE_W;
        // activation sources of this item's two column halves; this thread stages columns of half `myhalf`
        int C0, C1;
        const float* src0 = wg_half_src(a, ct.h[0], t, b, C0);
        const float* src1 = wg_half_src(a, ct.h[1], t, b, C1);
        if (!src0 && !src1) continue;  // (uniform branch)
        const float* src = myhalf ? src1 : src0;
        const int C = myhalf ? C1 : C0;
        const bool a_vec = (!src0 || ((C0 & 3) == 0 && (reinterpret_cast<uintptr_t>(src0) & 15) == 0)) &&
                           (!src1 || ((C1 & 3) == 0 && (reinterpret_cast<uintptr_t>(src1) & 15) == 0));
        const int ldG = a.ldG ? a.ldG : a.N4;
        const float* dg = a.dG + ((size_t)t * a.B + b) * a.HW * ldG;
        __syncthreads();
        // ---- stage dG tile: 128 pixels x 64 rows ----
        if (((a.N4 | ldG) & 3) == 0 && (reinterpret_cast<uintptr_t>(dg) & 15) == 0) {
            // (unconditional loads, out-of-range vectors from wg_zero16: see the bf16 forms — loads under a branch are drained one by one)
#pragma unroll 4
            for (int v = tid; v < 128 * 16; v += NTHREADS) {
                const int p = v >> 4, q4 = v & 15;
                const int gy = y0 + (p >> 4), gx = x0 + (p & 15);
                const int n = n0 + q4 * 4;
                const bool ok = gy < a.H && gx < a.W && n < a.N4;
                *reinterpret_cast<f32x4*>(G_lds + p * 64 + q4 * 4) =
                    *reinterpret_cast<const f32x4*>(ok ? dg + ((size_t)gy * a.W + gx) * ldG + n : wg_zero16);
            }
        } else {
            for (int e = tid; e < 128 * 64; e += NTHREADS) {
                const int p = e >> 6, q = e & 63;
                const int gy = y0 + (p >> 4), gx = x0 + (p & 15);
                const int n = n0 + q;
                float val = 0.f;
                if (gy < a.H && gx < a.W && n < a.N4) val = dg[((size_t)gy * a.W + gx) * ldG + n];
                G_lds[p * 64 + q] = val;
            }
        }
        // ---- stage activation halo tile: npos x (2 halves x 32 channels) ----
        if (a_vec) {
#pragma unroll 4
            for (int v = tid; v < npos * 16; v += NTHREADS) {
                const int pos = v >> 4, q4 = v & 15;
                const int hy = pos / halo_w, hx = pos - hy * halo_w;
                const int gy = y0 - ph + hy, gx = x0 - pw + hx;
                const int c = myh.c0 + (q4 & 7) * 4;
                long long pix = 0;
                const bool ok = wg_apix(a, gy, gx, pix) && src && c < C;
                *reinterpret_cast<f32x4*>(A_lds + pos * 64 + q4 * 4) =
                    *reinterpret_cast<const f32x4*>(ok ? src + pix * C + c : wg_zero16);
            }
        } else {
            for (int e = tid; e < npos * 64; e += NTHREADS) {
                const int pos = e >> 6, q = e & 63;
                const int hy = pos / halo_w, hx = pos - hy * halo_w;
                const int gy = y0 - ph + hy, gx = x0 - pw + hx;
                const int eh = q >> 5;  // (= (tid >> 5) & 1 here, not myhalf)
                const float* esrc = eh ? src1 : src0;
                const int eC = eh ? C1 : C0;
                const int c = ct.h[eh].c0 + (q & 31);
                float val = 0.f;
                long long pix;
                if (esrc && c < eC && wg_apix(a, gy, gx, pix)) val = esrc[pix * eC + c];
                A_lds[pos * 64 + q] = val;
            }
        }
        __syncthreads();
        // ---- contraction over the tile's 128 pixels, 2 per MFMA (lane half hh takes pixel 2*kk + hh) ----
        const float* gp = G_lds + wn * 32 + i;
        const float* ap = A_lds + wc * 32 + i;
#pragma unroll 4
        for (int kk = 0; kk < 64; ++kk) {
            const int p = 2 * kk + hh;
            const float av = gp[p * 64];
            const int abase = ((p >> 4) * halo_w + (p & 15)) * 64;
            float bv[MAXT];
#pragma unroll
            for (int t = 0; t < MAXT; ++t) bv[t] = ap[abase + tapoff[t]];
#pragma unroll
            for (int t = 0; t < MAXT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[t], acc[t], 0, 0, 0);
        }
    }
    // ---- slab[slice][tap][row][Ct] = acc (each element written exactly once per launch) ----
    const int col = ct.h[wc].cglobal + i;  // channel inside the concatenated [x | h] axis
    const bool col_ok = i < ct.h[wc].cn;
    const int n_out = a.n_out ? a.n_out : a.N4;
    float* slab = a.slabs + (size_t)slice * a.kh * a.kw * n_out * a.Ct;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        float* st = slab + (size_t)(tap0 + t) * n_out * a.Ct;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (n < a.N4 && col_ok) {
                const int row = a.blk ? (a.rowblk[n / a.blk] * a.blk + n % a.blk) : n;
                st[(size_t)row * a.Ct + col] = acc[t][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient, split-bf16 operands (bf16x3). Same decomposition as wgrad_kernel; both operands are contracted over
// PIXELS, i.e. they are needed k-major while HBM and LDS hold them channel-major. gfx950's transposing LDS read
// (ds_read_b64_tr_b16: a 16-lane group reads 4 rows x 16 columns and lane i receives column i of the 4 rows) delivers
// the MFMA fragments straight from the row-major tiles — no software transpose, no pre-shifted copies per tap.
// k-step = one tile row (16 pixels): lanes 0-31 take pixels 0-7, lanes 32-63 pixels 8-15 (two tr reads of 4 rows each).
// ---------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short wg_bf16_bits(float v) {
    __bf16 h = (__bf16)v;
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ void wg_split4(const f32x4 v, uint2& hi, uint2& lo) {
    unsigned short h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = wg_bf16_bits(v[e]);
        l[e] = wg_bf16_bits(v[e] - __builtin_bit_cast(float, (unsigned)h[e] << 16));
    }
    hi = uint2{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
    lo = uint2{(unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16)};
}
__device__ __forceinline__ bf16x8 wg_tr_frag(const char* base, const int pitch = 128) {
    // two transposing reads: rows (pixels) +0..3 and +4..7 of this lane half's 8-pixel group; `pitch` bytes per pixel row
    typedef bf16x4 __attribute__((address_space(3))) * lds_v4;
    const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(base));
    const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(base + 4 * pitch));
    return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

__device__ __forceinline__ int wg_aswz(const int off) { return off ^ ((off >> 2) & 0x40); }  // 128-byte rows, see GSWZ

// Items (t, b, tile) are software-pipelined through registers: while the MFMAs of item i run out of LDS, the global
// loads of item i+1 (GPRE dG vectors + APRE activation vectors per thread) are in flight; they are split to hi/lo bf16
// and stored to LDS after the barrier that ends item i. PIPE = 0 keeps the plain load-store-multiply order (fewer
// registers: two workgroups per CU).
// RB = 2: eight waves, 128 rows x 64 channels per workgroup — the activation halo tile (the larger of the two) is staged
// once for twice the rows, at two waves per SIMD (256 registers each) with the item pipeline on.
template <int MAXT, int APRE, int PIPE, int RB>
__global__ __launch_bounds__(NTHREADS * RB, (RB == 2 ? 2 : (PIPE ? 1 : 2))) void wgrad_bf16x3_kernel(const WgradArgs a, const int tap_base) {
    constexpr int NTH = NTHREADS * RB;
    constexpr int GROW = 64 * RB;   // dG rows per workgroup
    constexpr int GP = GROW * 2;    // bytes per pixel row of a dG plane
    const bool lo_terms = a.prec == VPX_PREC_BF16X3;  // plain bf16 uses the hi planes only (uniform branch)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, hh = lane >> 5;
    const int wn = wave >> 1, wc = wave & 1;
    int bx, slice;
    if (!wg_block(a, bx, slice)) return;
    const int n_ct = a.n_ctiles;
    const int ct_id = bx % n_ct;
    const int nt_id = bx / n_ct;
    const WgradCTile ct = a.ct[ct_id];
    const int tap0 = tap_base + blockIdx.z * MAXT;
    const int halo_w = TILE_W + a.kw - 1, halo_h = TILE_H + a.kh - 1, npos = halo_w * halo_h;
    const int ph = a.use_org ? -a.org_y : a.kh / 2, pw = a.use_org ? -a.org_x : a.kw / 2;
    // LDS planes: G_hi / G_lo [128 px][GROW rows] bf16, A_hi / A_lo [npos][64 channels] bf16 (128 B per position)
    char* G_hi = smem;
    char* G_lo = smem + 128 * GP;
    char* A_hi = smem + 2 * 128 * GP;
    char* A_lo = A_hi + npos * 128;
    constexpr int GVPR = GROW / 4;              // 4-float vectors per pixel row of the dG tile
    constexpr int GPRE = 128 * GVPR / NTH;      // dG vectors per thread and item (8)

    f32x16 acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // transposing-read address pattern of this lane inside its 16-lane group: row q, columns 4p..4p+3
    const int L = lane & 15, q = L >> 2, p = L & 3, half16 = (lane >> 4) & 1;
    // Bank swizzle: a half-wave's transposing read covers 4 consecutive rows x 64 B. At a row pitch of 128 B (256 B)
    // rows r and r+2 (all four rows) start in the same bank, so the 64-byte segment index of a row is XORed with
    // row bits: 128-B rows: segment ^= (row >> 1) & 1, 256-B rows: segment ^= row & 3 — the four rows of a read then
    // occupy four different 16-bank groups. For the dG planes the row bits come from the lane only (free); for the
    // activation planes they depend on the tap shift (wg_aswz per access).
    constexpr int GSWZ = RB == 2 ? 0xC0 : 0x40;
    int g_lane = ((8 * hh + q) * GROW + wn * 32 + 16 * half16 + 4 * p) * 2;  // bytes, + tile-row * 16 * GP
    g_lane ^= (g_lane >> 2) & GSWZ;
    const int a_lane = ((8 * hh + q) * 64 + wc * 32 + 16 * half16 + 4 * p) * 2;  // bytes, + position offset * 128
    int tapoff[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        const int tp = tap0 + t;
        const int dy = tp / a.kw, dx = tp - dy * a.kw;
        tapoff[t] = (dy * halo_w + dx) * 128;
    }

    const int tiles = a.tiles_x * a.tiles_y;
    const long long n_items = (long long)a.T * a.B * tiles;
    const int n0 = nt_id * GROW;
    const int ldG = a.ldG ? a.ldG : a.N4;
    const int q4 = tid & 15;              // this thread's 4-channel column of the activation tile
    const int qg = tid & (GVPR - 1);      // ... and its 4-row column of the dG tile
    const int myhalf = q4 >> 3;           // column half this thread stages
    const WgradCHalf myh = ct.h[myhalf];
    const int n_col = n0 + qg * 4, c_col = myh.c0 + (q4 & 7) * 4;

    constexpr int NV = PIPE ? APRE : 4;  // PIPE = 0 streams the tiles through 4 vectors at a time
    f32x4 gv[PIPE ? GPRE : 4], av[NV];
    struct ItemGeo { const float* src; const float* dg; int C, y0, x0; bool g_vec, a_vec; };
    // geometry of item w; returns false when the item contributes nothing (absent h at t = 0 in both column halves)
    auto item_geo = [&](long long w, ItemGeo& g) -> bool {
        const int tile = (int)(w % tiles);
        const long long tb = w / tiles;
        const int b = (int)(tb % a.B);
        const int t = (int)(tb / a.B);
        const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        g.y0 = ty * TILE_H; g.x0 = tx * TILE_W;
        int C0, C1;
        const float* src0 = wg_half_src(a, ct.h[0], t, b, C0);
        const float* src1 = wg_half_src(a, ct.h[1], t, b, C1);
        if (!src0 && !src1) return false;
        g.src = myhalf ? src1 : src0;  // null: this half stages zeros
        g.C = myhalf ? C1 : C0;
        g.dg = a.dG + ((size_t)t * a.B + b) * a.HW * ldG;
        g.g_vec = ((a.N4 | ldG) & 3) == 0 && (reinterpret_cast<uintptr_t>(g.dg) & 15) == 0;
        g.a_vec = (g.C & 3) == 0 && (reinterpret_cast<uintptr_t>(g.src) & 15) == 0;
        return true;
    };
    // vectors [u0, u0 + NU) of this thread: dG tile (128 pixels x 64 rows) and activation halo tile (npos x 64 channels)
    // (not in the register-pipelined 8-wave forms: measured slower there — PredRNN 5x5 training step 394 vs 310 ms; the
    // prefetched vectors are spilled as soon as they are all in flight at once)
    constexpr bool UNCOND = RB == 1 && !PIPE;
    // a.vec_all (host-checked: every operand 16-byte aligned, channel counts and strides multiples of 4): the loads are
    // issued unconditionally — out-of-range vectors read wg_zero16. Loads under divergent branches make the compiler
    // drain vmcnt between them (one memory latency EACH). (The returned mask only serves the ragged path's callers.)
    auto load_g = [&](const ItemGeo& g, int u0, auto nu, auto& dst) -> unsigned {
        unsigned mask = 0;
#pragma unroll
        for (int u = 0; u < nu; ++u) {
            const int pp = tid / GVPR + (u0 + u) * (NTH / GVPR);
            const int gy = g.y0 + (pp >> 4), gx = g.x0 + (pp & 15);
            if (UNCOND && a.vec_all) {
                const bool ok = pp < 128 && gy < a.H && gx < a.W && n_col < a.N4;
                dst[u] = *reinterpret_cast<const f32x4*>(ok ? g.dg + ((size_t)gy * a.W + gx) * ldG + n_col : wg_zero16);
                mask |= 1u << u;
                continue;
            }
            mask |= 1u << u;
            dst[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (pp < 128 && gy < a.H && gx < a.W) {
                const float* rowp = g.dg + ((size_t)gy * a.W + gx) * ldG;
                if (g.g_vec) { if (n_col < a.N4) dst[u] = *reinterpret_cast<const f32x4*>(rowp + n_col); }
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (n_col + e < a.N4) dst[u][e] = rowp[n_col + e];
                }
            }
        }
        return mask;
    };
    auto store_g = [&](int u0, auto nu, const auto& srcv, unsigned mask) {
#pragma unroll
        for (int u = 0; u < nu; ++u) {
            const int pp = tid / GVPR + (u0 + u) * (NTH / GVPR);
            if (pp < 128) {
                uint2 hi, lo;
                wg_split4((mask >> u) & 1 ? srcv[u] : f32x4{0.f, 0.f, 0.f, 0.f}, hi, lo);
                int off = pp * GP + qg * 8;
                off ^= (off >> 2) & GSWZ;
                *reinterpret_cast<uint2*>(G_hi + off) = hi;
                *reinterpret_cast<uint2*>(G_lo + off) = lo;
            }
        }
    };
    auto load_a = [&](const ItemGeo& g, int u0, auto nu, auto& dst) -> unsigned {
        unsigned mask = 0;
#pragma unroll
        for (int u = 0; u < nu; ++u) {
            const int pos = (tid >> 4) + (u0 + u) * (NTH / 16);
            const int hy = pos / halo_w, hx = pos - hy * halo_w;
            const int gy = g.y0 - ph + hy, gx = g.x0 - pw + hx;
            long long pix = 0;
            if (UNCOND && a.vec_all) {
                const bool ok = wg_apix(a, gy, gx, pix) && pos < npos && g.src != nullptr && c_col < g.C;
                dst[u] = *reinterpret_cast<const f32x4*>(ok ? g.src + pix * g.C + c_col : wg_zero16);
                mask |= 1u << u;
                continue;
            }
            mask |= 1u << u;
            dst[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (pos < npos && g.src && wg_apix(a, gy, gx, pix)) {
                const float* rowp = g.src + pix * g.C;
                if (g.a_vec) { if (c_col < g.C) dst[u] = *reinterpret_cast<const f32x4*>(rowp + c_col); }
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (c_col + e < g.C) dst[u][e] = rowp[c_col + e];
                }
            }
        }
        return mask;
    };
    auto store_a = [&](int u0, auto nu, const auto& srcv, unsigned mask) {
#pragma unroll
        for (int u = 0; u < nu; ++u) {
            const int pos = (tid >> 4) + (u0 + u) * (NTH / 16);
            if (pos < npos) {
                uint2 hi, lo;
                wg_split4((mask >> u) & 1 ? srcv[u] : f32x4{0.f, 0.f, 0.f, 0.f}, hi, lo);
                const int off = wg_aswz(pos * 128 + q4 * 8);
                *reinterpret_cast<uint2*>(A_hi + off) = hi;
                *reinterpret_cast<uint2*>(A_lo + off) = lo;
            }
        }
    };
    unsigned gmask = 0, amask = 0;  // validity of the vectors held in (gv, av): PIPE only
    auto load_item = [&](long long w) -> bool {  // whole item into (gv, av): PIPE only
        ItemGeo g;
        if (!item_geo(w, g)) return false;
        gmask = load_g(g, 0, std::integral_constant<int, PIPE ? GPRE : 4>{}, gv);
        amask = load_a(g, 0, std::integral_constant<int, NV>{}, av);
        return true;
    };
    auto store_item = [&]() {
        store_g(0, std::integral_constant<int, PIPE ? GPRE : 4>{}, gv, gmask);
        store_a(0, std::integral_constant<int, NV>{}, av, amask);
    };
    auto multiply = [&]() {  // 8 k-steps (tile rows) of 16 pixels
#pragma unroll 2
        for (int s = 0; s < TILE_H; ++s) {
            const bf16x8 gh = wg_tr_frag(G_hi + g_lane + s * 16 * GP, GP);
            const bf16x8 gl = wg_tr_frag(G_lo + g_lane + s * 16 * GP, GP);
            const int arow = a_lane + s * halo_w * 128;
#pragma unroll
            for (int t2 = 0; t2 < MAXT; ++t2) {
                const int aoff = wg_aswz(arow + tapoff[t2]);
                const bf16x8 ah = wg_tr_frag(A_hi + aoff);
                if (lo_terms) {
                    const bf16x8 al = wg_tr_frag(A_lo + aoff);
                    acc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gl, ah, acc[t2], 0, 0, 0);
                    acc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh, al, acc[t2], 0, 0, 0);
                }
                acc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh, ah, acc[t2], 0, 0, 0);
            }
        }
    };

    if constexpr (PIPE) {
        long long w = slice;
        bool have = w < n_items && load_item(w);
        while (w < n_items) {
            __syncthreads();  // every wave is done reading the previous item's planes
            if (have) store_item();
            __syncthreads();
            const bool had = have;
            w += a.grid_slices;
            have = w < n_items && load_item(w);  // next item's loads fly while this one is multiplied
            if (had) multiply();
        }
    } else {
        for (long long w = slice; w < n_items; w += a.grid_slices) {
            ItemGeo g;
            if (!item_geo(w, g)) continue;  // (uniform: depends on the item and the column tile only)
            __syncthreads();
            for (int u0 = 0; u0 < GPRE; u0 += 4) { const unsigned m = load_g(g, u0, std::integral_constant<int, 4>{}, gv); store_g(u0, std::integral_constant<int, 4>{}, gv, m); }
            for (int u0 = 0; u0 * (NTH / 16) < npos; u0 += 4) { const unsigned m = load_a(g, u0, std::integral_constant<int, 4>{}, av); store_a(u0, std::integral_constant<int, 4>{}, av, m); }
            __syncthreads();
            multiply();
        }
    }
    const int col = ct.h[wc].cglobal + i;
    const bool col_ok = i < ct.h[wc].cn;
    const int n_out = a.n_out ? a.n_out : a.N4;
    float* slab = a.slabs + (size_t)slice * a.kh * a.kw * n_out * a.Ct;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        float* st = slab + (size_t)(tap0 + t) * n_out * a.Ct;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (n < a.N4 && col_ok) {
                const int row = a.blk ? (a.rowblk[n / a.blk] * a.blk + n % a.blk) : n;
                st[(size_t)row * a.Ct + col] = acc[t][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient, split-bf16 operands, tap-group form: 8 waves = 2 (gate rows) x 2 (channel halves) x 2 (TAP GROUPS).
// The 64 x 64 x MAXT output tile of a workgroup is spread over 512 threads, so a wave carries ceil(MAXT/2) accumulator
// tiles (80 registers for 3x3) instead of nine: room for the hi-plane fragments of a whole k-step in flight under the
// MFMAs and for an item's global loads (GPRE + APRE vectors per thread), which wait in registers for a full item.
// LDS holds TWO item buffers. Per item and wave:   group 0:  multiply(i)  -> split+store(i+1) -> load(i+2)
//                                                   group 1:  split+store(i+1) -> load(i+2) -> multiply(i)
// (one barrier per item). Every SIMD hosts one wave of each group, so the VALU work of the hi/lo split runs under the
// other wave's MFMAs (waves w and w+4 of a workgroup share a SIMD). TH = tile rows per item: two buffers must fit the
// 160 KB, i.e. TH = 8 for kernels up to 3x3 (the only instantiation; TH = 4 for 5x5 lost to the 128-row form).
// ---------------------------------------------------------------------------------------------------------------
// workgroup barrier that orders LDS traffic only: outstanding global loads (the next item, held in registers) stay in flight
__device__ __forceinline__ void wg_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// DB = false: ONE item buffer (5x5 / 7x7 halo tiles: two do not fit) — every wave stores item i, multiplies it with the
// loads of item i+1 in flight, two barriers per item; the tap groups still halve the accumulator registers.
// ASP = true (round 2): the ACTIVATION operand arrives pre-split (x / h_t written in operand format by the cell2 forward):
// its halo tile is copied HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, six 16-byte pieces per thread and item, the XOR
// swizzle of the 128-byte rows applied through the choice of source piece) — no split VALU work for it (58 % of an item's
// conversion work), and the 24 registers that held its prefetched vectors go back to the compiler for fragment read-ahead.
// dG still comes through registers (fp32 from the gate-backward kernel). DB only.
template <int MAXT, int TH, int APRE, bool DB = true, bool ASP = false>
__global__ __launch_bounds__(512, 2) void wgrad_tg_kernel(const WgradArgs a, const int tap_base) {
    constexpr int NTH = 512;
    constexpr int TA = (MAXT + 1) / 2, TB = MAXT - TA;  // taps of group 0 / group 1
    constexpr int NPX = TH * TILE_W;                    // pixels per item
    constexpr int GPRE = NPX * 16 / NTH;                // dG vectors per thread and item
    const bool lo_terms = a.prec == VPX_PREC_BF16X3;    // plain bf16 uses the hi planes only (uniform branch)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // (readfirstlane: wave-uniform values the compiler cannot prove uniform — they then live in SGPRs, the item walk
    // and the group branches run on the scalar unit, and the column-tile record is read with scalar loads)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, hh = lane >> 5;
    const int tg = wave >> 2, wn = (wave >> 1) & 1, wc = wave & 1;
    int bx, slice;
    if (!wg_block(a, bx, slice)) return;
    bx = __builtin_amdgcn_readfirstlane(bx); slice = __builtin_amdgcn_readfirstlane(slice);
    const int n_ct = a.n_ctiles;
    const int ct_id = __builtin_amdgcn_readfirstlane(bx % n_ct);
    const WgradCHalf ch0 = a.ct[ct_id].h[0], ch1 = a.ct[ct_id].h[1];
    const int n0 = __builtin_amdgcn_readfirstlane((bx / n_ct) * 64);
    const int tap0 = tap_base + blockIdx.z * MAXT + (tg ? TA : 0);
    const int halo_w = TILE_W + a.kw - 1, halo_h = TH + a.kh - 1, npos = halo_w * halo_h;
    const int ph = a.use_org ? -a.org_y : a.kh / 2, pw = a.use_org ? -a.org_x : a.kw / 2;
    // one item buffer: G_hi | G_lo [NPX px][64 rows] bf16, A_hi | A_lo [npos][64 channels] bf16 (128 B per pixel / position)
    const int G_LO = NPX * 128, A_HI = 2 * NPX * 128, A_LO = A_HI + npos * 128, BUF = A_LO + npos * 128;

    f32x16 acc[TA];
#pragma unroll
    for (int t = 0; t < TA; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    const int L = lane & 15, q = L >> 2, p = L & 3, half16 = (lane >> 4) & 1;
    int g_lane = ((8 * hh + q) * 64 + wn * 32 + 16 * half16 + 4 * p) * 2;
    g_lane ^= (g_lane >> 2) & 0x40;  // bank swizzle of the 128-byte rows, see wgrad_bf16x3_kernel
    const int a_lane = ((8 * hh + q) * 64 + wc * 32 + 16 * half16 + 4 * p) * 2;
    int tapoff[TA];
#pragma unroll
    for (int t = 0; t < TA; ++t) {
        const int tp = tap0 + t;
        const int dy = tp / a.kw, dx = tp - dy * a.kw;
        tapoff[t] = (dy * halo_w + dx) * 128;
    }

    // ---- item walk: slice, slice + n_slices, ... as a mixed-radix counter (tx, ty, b, t): no division per item ----
    const int tiles_x = (a.W + TILE_W - 1) / TILE_W, tiles_y = (a.H + TH - 1) / TH, tiles = tiles_x * tiles_y;
    const int ns = a.grid_slices;
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
    const bool skip_t0 = !a.h0 && (ch0.cn == 0 || ch0.seg == 1) && (ch1.cn == 0 || ch1.seg == 1);
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

    // ---- staging: this thread's 4-wide column of both tiles and its pixel / halo rows prow + 32 u ----
    const int q4 = tid & 15, prow = tid >> 4;
    const int myhalf = q4 >> 3;
    const WgradCHalf myh = myhalf ? ch1 : ch0;
    const int ldG = a.ldG ? a.ldG : a.N4;
    const int n_col = n0 + q4 * 4, c_col = myh.c0 + (q4 & 7) * 4;
    int hyx[APRE];  // halo position of vector u: (row << 16) | column, -1: beyond the halo tile
#pragma unroll
    for (int u = 0; u < APRE; ++u) {
        const int pos = prow + 32 * u;
        const int hy = pos / halo_w;
        hyx[u] = pos < npos ? ((hy << 16) | (pos - hy * halo_w)) : -1;
    }
    // Every load is issued unconditionally (16-byte vectors; the host routes unaligned / ragged-channel operands to the
    // older kernels): out-of-range vectors read wg_zero16. Straight-line issue matters: loads under divergent branches
    // make the compiler drain vmcnt between them, one latency each.
    f32x4 gv[GPRE], av[APRE];
    auto load_item = [&](const Item& it) {
        const int y0 = it.ty * TH, x0 = it.tx * TILE_W;
        int C0, C1;
        const float* src0 = wg_half_src(a, ch0, it.t, it.b, C0);
        const float* src1 = wg_half_src(a, ch1, it.t, it.b, C1);
        const float* src = myhalf ? src1 : src0;  // null: this half stages zeros
        const int C = myhalf ? C1 : C0;
        const float* dg = a.dG + ((size_t)it.t * a.B + it.b) * a.HW * ldG;
        const bool a_ok = src != nullptr && c_col < C;
#pragma unroll
        for (int u = 0; u < GPRE; ++u) {
            const int pp = prow + 32 * u;
            const int gy = y0 + (pp >> 4), gx = x0 + (pp & 15);
            const bool ok = gy < a.H && gx < a.W && n_col < a.N4;
            gv[u] = *reinterpret_cast<const f32x4*>(ok ? dg + (gy * a.W + gx) * ldG + n_col : wg_zero16);
        }
        if constexpr (!ASP) {
#pragma unroll
            for (int u = 0; u < APRE; ++u) {
                const int gy = y0 - ph + (hyx[u] >> 16), gx = x0 - pw + (hyx[u] & 0xffff);
                long long pix = 0;
                const bool ok = wg_apix(a, gy, gx, pix) && hyx[u] >= 0 && a_ok;
                av[u] = *reinterpret_cast<const f32x4*>(ok ? src + (int)pix * C + c_col : wg_zero16);
            }
        }
    };
    // ---- ASP: this thread's six 16-byte pieces of the activation halo image (hi plane then lo plane, 8 pieces per position) ----
    constexpr int NPIECE = ASP ? 6 : 1;
    int pc_hyx[NPIECE], pc_off[NPIECE];   // halo position (row << 16 | col, -1 = none) / (half << 16) | byte offset inside the half's pixel row
    if constexpr (ASP) {
#pragma unroll
        for (int u = 0; u < NPIECE; ++u) {
            const int piece = tid + NTH * u;
            const int plane = piece >= npos * 8 ? 1 : 0;
            const int qq = piece - plane * npos * 8;
            const int pos = qq >> 3;
            const int logical = wg_aswz(pos * 128 + (qq & 7) * 16);   // the swizzle is an involution: physical slot -> logical offset
            const int sl = (logical & 127) >> 4;                       // logical 16-byte slot = 8 channels of the 64-channel row
            const int hy = pos / halo_w;
            const WgradCHalf hf = (sl >> 2) ? ch1 : ch0;
            const bool ok = piece < 2 * npos * 8 && (sl & 3) * 8 < hf.cn;
            pc_hyx[u] = ok ? ((hy << 16) | (pos - hy * halo_w)) : -1;
            pc_off[u] = ((sl >> 2) << 16) | (((hf.c0 + (sl & 3) * 8) >> 3) * 32 + plane * 16);
        }
    }
    auto dma_A = [&](const Item& it, char* buf) {
        if constexpr (ASP) {
            const int y0 = it.ty * TH, x0 = it.tx * TILE_W;
            // per half: base of this (t, b) image in its split tensor and bytes per pixel row; null = the half stages zeros
            const char* base[2];
            int prow_b[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const WgradCHalf hf = h ? ch1 : ch0;
                base[h] = nullptr; prow_b[h] = 0;
                if (hf.cn == 0) continue;
                if (hf.seg == 0) { base[h] = a.x_sp + (size_t)it.b * a.x_sp_bstride + (size_t)it.t * a.x_sp_tstride; prow_b[h] = a.Cin * 4; }
                else {
                    prow_b[h] = a.Ch * 4;
                    if (it.t > 0) base[h] = a.h_sp + (size_t)(it.t - 1) * a.h_sp_tstride + (size_t)it.b * a.h_sp_bstride;
                    else if (a.h0_sp) base[h] = a.h0_sp + (size_t)it.b * a.HW * a.Ch * 4;
                }
            }
            char* dst = buf + A_HI + (wave * 64) * 16;
#pragma unroll
            for (int u = 0; u < NPIECE; ++u) {
                const int h = pc_off[u] >> 16;
                const int gy = y0 - ph + (pc_hyx[u] >> 16), gx = x0 - pw + (pc_hyx[u] & 0xffff);
                const bool ok = pc_hyx[u] >= 0 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && base[h] != nullptr;
                const char* src = ok ? base[h] + (size_t)(gy * a.W + gx) * prow_b[h] + (pc_off[u] & 0xffff)
                                     : reinterpret_cast<const char*>(wg_zero16);
                if (tid + NTH * u < 2 * npos * 8) {
                    const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(dst + u * NTH * 16);
                    asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                                 :: "v"(src), "{m0}"(__builtin_amdgcn_readfirstlane(lds)) : "memory");
                }
            }
        }
    };
    auto store_item = [&](char* buf) {
#pragma unroll
        for (int u = 0; u < GPRE; ++u) {
            uint2 hi, lo;
            wg_split4(gv[u], hi, lo);
            int off = (prow + 32 * u) * 128 + q4 * 8;
            off ^= (off >> 2) & 0x40;
            *reinterpret_cast<uint2*>(buf + off) = hi;
            *reinterpret_cast<uint2*>(buf + G_LO + off) = lo;
        }
        if constexpr (!ASP) {
#pragma unroll
            for (int u = 0; u < APRE; ++u) {
                if (hyx[u] >= 0) {
                    uint2 hi, lo;
                    wg_split4(av[u], hi, lo);
                    const int off = wg_aswz((prow + 32 * u) * 128 + q4 * 8);
                    *reinterpret_cast<uint2*>(buf + A_HI + off) = hi;
                    *reinterpret_cast<uint2*>(buf + A_LO + off) = lo;
                }
            }
        }
    };
    // TH k-steps (tile rows) of 16 pixels. ONE copy of the loop serves both wave groups (two copies made the compiler
    // spill the prefetched vectors): the TB taps both groups have as one batch with the three terms issued term-major
    // (consecutive MFMAs accumulate into different tiles), then group 0's extra tap under a scalar branch. Measured
    // alternatives (64ch 64x64 block, B=128: this 4.32 ms): batches of <= 3 taps without unrolling 4.65; a branch-free block
    // of TA taps with a repeated tap in group 1's spare slot 5.24; the odd tap split over k between the groups 5.66; an
    // explicit sched_barrier-fenced read-ahead pipeline 5.31 — each loses to spills of the prefetched vectors. With pre-split
    // activations (ASP) two explicit fragment pipelines (two register sets; a rolling single set) were also tried: the compiler
    // hoists the swizzled addresses of every unrolled k-step, hits 256 VGPRs and spills (36-208 B/lane) — not kept.
    auto tap_batch = [&](const char* buf, const bf16x8& gh, const bf16x8& gl, int arow, auto t0_c, auto nb_c, auto lo_c) {
        constexpr int T0 = decltype(t0_c)::value, NB = decltype(nb_c)::value;
        constexpr bool LO = decltype(lo_c)::value;
        bf16x8 ah[NB], al[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int aoff = wg_aswz(arow + tapoff[T0 + j]);
            ah[j] = wg_tr_frag(buf + A_HI + aoff);
            if constexpr (LO) al[j] = wg_tr_frag(buf + A_LO + aoff);
        }
        if constexpr (ASP) {
            // all fragment reads of the batch ahead of its MFMAs (the freed prefetch registers pay for it): without this the
            // compiler re-uses one register quad for successive lo-plane fragments and waits for each read right before its MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, (LO ? 4 : 2) * NB, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, (LO ? 3 : 1) * NB, 0);
        }
        if constexpr (LO) {
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[T0 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gl, ah[j], acc[T0 + j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[T0 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh, al[j], acc[T0 + j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[T0 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh, ah[j], acc[T0 + j], 0, 0, 0);
    };
    auto multiply = [&](const char* buf, auto lo_c) {
        constexpr bool LO = decltype(lo_c)::value;
#pragma unroll 2
        for (int s = 0; s < TH; ++s) {
            const bf16x8 gh = wg_tr_frag(buf + g_lane + s * 16 * 128);
            bf16x8 gl = gh;
            if constexpr (LO) gl = wg_tr_frag(buf + G_LO + g_lane + s * 16 * 128);
            const int arow = a_lane + s * halo_w * 128;
            if constexpr (TB >= 1) tap_batch(buf, gh, gl, arow, std::integral_constant<int, 0>{}, std::integral_constant<int, TB>{}, lo_c);
            if constexpr (TA > TB) {
                if (tg == 0) tap_batch(buf, gh, gl, arow, std::integral_constant<int, TB>{}, std::integral_constant<int, 1>{}, lo_c);
            }
        }
    };
    auto multiply_g = [&](const char* buf) {  // (wave-uniform branch)
        if (lo_terms) multiply(buf, std::true_type{});
        else multiply(buf, std::false_type{});
    };

    Item nxt = cur;
    advance(nxt);
    if constexpr (!DB) {
        if (cur.t < a.T) load_item(cur);
        while (cur.t < a.T) {
            wg_lds_barrier();  // every wave is done reading the previous item
            store_item(smem);
            wg_lds_barrier();
            if (nxt.t < a.T) load_item(nxt);  // in flight during the multiply
            multiply_g(smem);
            cur = nxt;
            advance(nxt);
        }
    }
    if (DB && cur.t < a.T) { dma_A(cur, smem); load_item(cur); store_item(smem); }
    if (DB && nxt.t < a.T) load_item(nxt);
    if constexpr (ASP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the first item's DMA pieces (unknown to the compiler)
    __syncthreads();
    int bsel = 0;
    while (DB && cur.t < a.T) {
        char* bcur = smem + bsel * BUF;
        char* bnxt = smem + (bsel ^ 1) * BUF;
        Item nn = nxt;
        advance(nn);
        if constexpr (ASP) {
            // One straight-line order for every wave (the three-phase loop below makes the compiler keep TWO copies of the
            // accumulator tiles — 160 registers — and serialise every fragment read behind its MFMA): the other buffer was
            // multiplied in the previous iteration and every wave has passed that iteration's barrier, so its activation image
            // is overwritten right away by the copy of item i+1 (a whole item to land), dG(i+1) goes from registers to LDS,
            // the dG loads of item i+2 fly under the multiply of item i.
            if (nxt.t < a.T) { dma_A(nxt, bnxt); store_item(bnxt); }
            if (nn.t < a.T) load_item(nn);
            multiply_g(bcur);
            // the copy must have landed; the dG loads issued after it may stay in flight (vmcnt is in order)
            if (nn.t < a.T) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(GPRE) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
        // group 0: multiply, then stage; group 1: stage, then multiply — as ONE copy of each (a three-phase loop the
        // compiler must not unroll)
#ifdef VPX_ABLATE  // timing-only variants (results are wrong): 1 no multiply, 2 no split+store, 4 no global loads, 8 same order in both groups
        const bool do_mul = !(a.dbg & 1), do_st = !(a.dbg & 2), do_ld = !(a.dbg & 4);
        const int stage_phase = (tg == 1 && !(a.dbg & 8)) ? 0 : 2;
#else
        constexpr bool do_mul = true, do_st = true, do_ld = true;
        const int stage_phase = tg == 1 ? 0 : 2;
#endif
#pragma nounroll
        for (int phase = 0; phase < 3; ++phase) {
            if (phase == 1) { if (do_mul) multiply_g(bcur); }
            else if (phase == stage_phase) {
                if (nxt.t < a.T && do_st) store_item(bnxt);
                if (nn.t < a.T && do_ld) load_item(nn);
            }
        }
        }
        wg_lds_barrier();  // (not __syncthreads(): its fence would also wait for the global loads just issued)
        cur = nxt; nxt = nn; bsel ^= 1;
    }

    const WgradCHalf oh = wc ? ch1 : ch0;
    const int col = oh.cglobal + i;
    const bool col_ok = i < oh.cn;
    const int n_out = a.n_out ? a.n_out : a.N4;
    float* slab = a.slabs + (size_t)slice * a.kh * a.kw * n_out * a.Ct;
#pragma unroll
    for (int t = 0; t < TA; ++t) {
        if (tg == 1 && t >= TB) break;
        float* st = slab + (size_t)(tap0 + t) * n_out * a.Ct;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (n < a.N4 && col_ok) {
                const int row = a.blk ? (a.rowblk[n / a.blk] * a.blk + n % a.blk) : n;
                st[(size_t)row * a.Ct + col] = acc[t][r];
            }
        }
    }
}

template <int NTAPS>
static hipError_t launch_wgrad_group(const WgradArgs& a_in, int n_slices, int tap_base, int groups, size_t lds, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = vpx_func_attr(reinterpret_cast<const void*>(&wgrad_kernel<NTAPS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = !g_dry_run;
    }
    auto xcd_grid = [&](WgradArgs& w, int gx) {  // see wg_block
        w.grid_x = gx; w.grid_slices = n_slices;
        const long long total = (long long)gx * n_slices;
        return dim3((unsigned)(8 * ((total + 7) / 8)), 1, groups);
    };
    WgradArgs a = a_in;
#ifdef VPX_ABLATE
    a.dbg = dev_switch("VPX_WG_DBG", 0);
#endif
    const dim3 grid = xcd_grid(a, ((a.N4 + 63) / 64) * a.n_ctiles);
    if (a.prec == VPX_PREC_BF16X3 || a.prec == VPX_PREC_BF16) {
        // activation vectors per thread and item: halo positions * 16 / 256 (3x3: 12, 5x5: 15, 7x7: 20)
        const int npos = (TILE_H + a.kh - 1) * (TILE_W + a.kw - 1);
        auto go = [&](auto kern, dim3 g, int nth, size_t lds_bytes) -> hipError_t {
            hipError_t e = vpx_func_attr(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            VPX_LAUNCH(kern, g, dim3(nth), lds_bytes, s, a, tap_base);
            return vpx_hip_last_error();
        };
        {   // 16-byte vector loads throughout: aligned bases, channel counts and strides in multiples of 4 floats
            auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
            const int ldG = a.ldG ? a.ldG : a.N4;
            bool vec = ((a.N4 | ldG) & 3) == 0 && al16(a.dG);
            for (int c = 0; c < a.n_ctiles; ++c)
                for (int h = 0; h < 2; ++h) {
                    if (a.ct[c].h[h].cn == 0) continue;
                    if (a.ct[c].h[h].seg == 0) vec = vec && (a.Cin & 3) == 0 && al16(a.x) && ((a.x_bstride | a.x_tstride) & 3) == 0;
                    else vec = vec && (a.Ch & 3) == 0 && al16(a.hseq) && al16(a.h0) && ((a.h_bstride | a.h_tstride) & 3) == 0;
                }
            a.vec_all = vec ? 1 : 0;
        }
        // tap-group form (wgrad_tg_kernel)
        static int tg_env = -1;  // VPX_WGRAD_TG=0 falls back to the older forms below (experiments)
        if (tg_env < 0) tg_env = dev_switch("VPX_WGRAD_TG", 1);
        if constexpr (NTAPS >= 2) {
            const long long items = (long long)a.T * a.B * ((a.W + TILE_W - 1) / TILE_W) * ((a.H + TILE_H - 1) / TILE_H);
            if (tg_env && a.vec_all && items + n_slices < (1ll << 31)) {
                // kernels up to 3x3: two buffers of 8 x 16 pixels fit the LDS. Larger kernels would need 4-row items (halo
                // overhead 2.5x for 5x5): measured slower than the 128-row form below (PredRNN 5x5 step 332 vs 311 ms)
                if (a.kh <= 3 && a.kw <= 3) {
                    const size_t l2 = 2 * (size_t)(2 * 128 * 128 + 2 * npos * 128);
                    static int asp_env = -1;  // VPX_WGRAD_ASP=0: convert the activation operand in the kernel as before (experiments)
                    if (asp_env < 0) asp_env = dev_switch("VPX_WGRAD_ASP", 1);
                    if (a.a_split && asp_env && !a.a_sub && !a.use_org && NTAPS == 9 && npos * 16 <= 6 * 512)
                        return go(&wgrad_tg_kernel<NTAPS, 8, 6, true, true>, grid, 512, l2);
                    return go(&wgrad_tg_kernel<NTAPS, 8, 6>, grid, 512, l2);
                }
                static int tg1_env = -1;  // VPX_WGRAD_TG1=0: larger kernels on the 128-row form below
                if (tg1_env < 0) tg1_env = dev_switch("VPX_WGRAD_TG1", 1);
                if (tg1_env && npos <= 256)  // up to 5x5: one item buffer, 8 activation vectors per thread
                    return go(&wgrad_tg_kernel<NTAPS, 8, 8, false>, grid, 512, (size_t)(2 * 128 * 128 + 2 * npos * 128));
            }
        }
        static int rb_env = -1;  // VPX_WGRAD_RB=1 forces the 4-wave, 64-row form
        if (rb_env < 0) rb_env = dev_switch("VPX_WGRAD_RB", 0);
        // 8 waves / 128 rows / pipelined items when the 4-wave form's planes (> 80 KB: 5x5 and larger) allow one workgroup
        // per CU anyway; 3x3 keeps two independent 4-wave workgroups per CU (measured, training step: ConvLSTM 3x3
        // 46.2 ms vs 50.9 ms with the 8-wave form; PredRNN 5x5 134.7 ms vs 127.6 ms). VPX_WGRAD_RB=1/2 forces a form.
        if (a.N4 >= 128 && rb_env != 1 && (lds > 80 * 1024 || rb_env == 2)) {
            const int apre2 = (npos * 16 + 2 * NTHREADS - 1) / (2 * NTHREADS);
            const dim3 g2 = xcd_grid(a, ((a.N4 + 127) / 128) * a.n_ctiles);
            const size_t lds2 = 2 * 128 * 256 + (size_t)npos * 128 * 2;
            if (apre2 > 10 || lds2 > 160 * 1024) return hipErrorInvalidValue;  // kernels beyond 7x7 are not instantiated
            if (apre2 <= 4) return go(&wgrad_bf16x3_kernel<NTAPS, 4, 1, 2>, g2, 2 * NTHREADS, lds2);
            if (apre2 <= 6) return go(&wgrad_bf16x3_kernel<NTAPS, 6, 1, 2>, g2, 2 * NTHREADS, lds2);
            if (apre2 <= 8) return go(&wgrad_bf16x3_kernel<NTAPS, 8, 1, 2>, g2, 2 * NTHREADS, lds2);
            return go(&wgrad_bf16x3_kernel<NTAPS, 10, 1, 2>, g2, 2 * NTHREADS, lds2);
        }
        return go(&wgrad_bf16x3_kernel<NTAPS, 0, 0, 1>, grid, NTHREADS, lds);
    } else {
        VPX_LAUNCH(wgrad_kernel<NTAPS>, grid, dim3(NTHREADS), lds, s, a, tap_base);
    }
    return vpx_hip_last_error();
}

hipError_t launch_wgrad(const WgradArgs& a, int n_slices, hipStream_t s) {
    constexpr int MAXT = 9;  // taps per workgroup: 9 accumulator tiles per wave
    const int taps = a.kh * a.kw;
    if (!ws_write_ok(a.slabs, (size_t)n_slices * taps * (a.n_out ? a.n_out : a.N4) * a.Ct * sizeof(float), "weight-gradient slabs (wgrad_kernel)"))
        return hipErrorInvalidValue;
    const int full = taps / MAXT, rem = taps % MAXT;
    const int npos = (TILE_H + a.kh - 1) * (TILE_W + a.kw - 1);
    const size_t lds = 128 * 64 * 4 + (size_t)npos * 64 * 4;
    hipError_t e = hipSuccess;
    if (full) e = launch_wgrad_group<MAXT>(a, n_slices, 0, full, lds, s);
    if (e != hipSuccess) return e;
    switch (rem) {  // remainder group with its exact tap count (1x1 -> 1, 5x5 -> 7, 5x3 -> 6, 7x7 -> 4, ...)
        case 0: break;
        case 1: e = launch_wgrad_group<1>(a, n_slices, full * MAXT, 1, lds, s); break;
        case 2: e = launch_wgrad_group<2>(a, n_slices, full * MAXT, 1, lds, s); break;
        case 3: e = launch_wgrad_group<3>(a, n_slices, full * MAXT, 1, lds, s); break;
        case 4: e = launch_wgrad_group<4>(a, n_slices, full * MAXT, 1, lds, s); break;
        case 5: e = launch_wgrad_group<5>(a, n_slices, full * MAXT, 1, lds, s); break;
        case 6: e = launch_wgrad_group<6>(a, n_slices, full * MAXT, 1, lds, s); break;
        case 7: e = launch_wgrad_group<7>(a, n_slices, full * MAXT, 1, lds, s); break;
        case 8: e = launch_wgrad_group<8>(a, n_slices, full * MAXT, 1, lds, s); break;
    }
    return e;
}

// dW[n][c][tap] (OIHW, ld = Ct*taps) = sum_s slab[s][tap][n][c]
struct TapMap { int real_taps; int map[16]; };  // real_taps = 0: identity (launch taps == tensor taps)
__global__ void wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dW, int n_slices, int taps,
                                    int N4, int Ct, const TapMap tm, int tail_col0, int tail_slices) {
    const long long total = (long long)N4 * Ct * taps;
    const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (e >= total) return;
    // thread order follows the slab layout (coalesced reads); the OIHW write is strided but tiny
    const int c = (int)(e % Ct);
    long long r = e / Ct;
    const int n = (int)(r % N4);
    const int tap = (int)(r / N4);
    int real_taps = taps, dst_tap = tap;
    if (tm.real_taps) { real_taps = tm.real_taps; dst_tap = tm.map[tap]; if (dst_tap < 0) return; }
    float acc = 0.f;
    const size_t slab_sz = (size_t)taps * N4 * Ct;
    const int ns = c >= tail_col0 ? tail_slices : n_slices;   // wgrad2: the half-empty last column tile runs on fewer slices
    // eight independent partial sums (slab s -> sum s % 8), combined in a fixed order: the serial chain kept one load in flight
    // per thread (62 us on average for 30-150 MB of slabs; bit-reproducible either way)
    float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* src = slabs + e;
    int s = 0;
    for (; s + 8 <= ns; s += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) p[k] += src[(size_t)(s + k) * slab_sz];
    }
    for (int k = 0; s < ns; ++s, ++k) p[k] += src[(size_t)s * slab_sz];
    acc = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    dW[((size_t)n * Ct + c) * real_taps + dst_tap] = acc;
}

hipError_t launch_wgrad_reduce(const float* slabs, float* dW, int n_slices, int taps, int N4, int Ct, hipStream_t s) {
    const long long total = (long long)N4 * Ct * taps;
    VPX_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, slabs, dW, n_slices,
                       taps, N4, Ct, TapMap{}, Ct, n_slices);
    return vpx_hip_last_error();
}

hipError_t launch_wgrad_reduce_tail(const float* slabs, float* dW, int n_slices, int taps, int N4, int Ct, int tail_col0, int tail_slices,
                                    hipStream_t s) {
    const long long total = (long long)N4 * Ct * taps;
    VPX_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, slabs, dW, n_slices,
                       taps, N4, Ct, TapMap{}, tail_col0, tail_slices);
    return vpx_hip_last_error();
}

hipError_t launch_wgrad_reduce_map(const float* slabs, float* dW, int n_slices, int taps, int N4, int Ct, int real_taps,
                                   const int* tapmap, hipStream_t s) {
    if (taps > 16) return hipErrorInvalidValue;
    TapMap tm{};
    tm.real_taps = real_taps;
    for (int i = 0; i < taps; ++i) tm.map[i] = tapmap[i];
    const long long total = (long long)N4 * Ct * taps;
    VPX_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, slabs, dW, n_slices,
                       taps, N4, Ct, tm, Ct, n_slices);
    return vpx_hip_last_error();
}

// ---------------------------------------------------------------------------------------------------------------
// ST-LSTM backward, pointwise stages (predrnn.py:65-81 differentiated)
// ---------------------------------------------------------------------------------------------------------------
__global__ void st_bwd_out_kernel(const STBwdOutArgs a) {
    const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (e >= a.n) return;
    const long long pix = e / a.Ch;
    const int ch = (int)(e - pix * a.Ch);
    const float dh = a.dh_new ? a.dh_new[e] : 0.0f;
    const float o = a.o[e], tl = a.tl[e];
    a.dG7[pix * a.ldG + a.o_off + ch] = dh * tl * o * (1.0f - o);  // d(o_x + o_h + conv_o(mem))
    if (a.dlc_off >= 0) a.dG7[pix * a.ldG + a.dlc_off + ch] = dh * o * (1.0f - tl * tl);
    if (a.dlc) a.dlc[e] = dh * o * (1.0f - tl * tl);               // d conv_last(mem)
}

__global__ void st_bwd_gates_kernel(const STBwdGateArgs a) {
    const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const int Ch = a.Ch;
    if (e >= a.npix * Ch) return;
    const long long pix = e / Ch;
    const int ch = (int)(e - pix * Ch);
    float* dg = a.dG7 + pix * a.ldG + ch;
    {   // c group: c_new = f*c + i*g ; delta_c = i*g
        const float* gs = a.gates_c + pix * 3 * Ch + ch;
        const float i_ = gs[0], f_ = gs[Ch], g_ = gs[2 * Ch];
        const float dcn = (a.dcn_ext ? a.dcn_ext[e] : 0.0f) + a.dcn_conv[e];
        const float ddl = dcn + (a.ddc_ext ? a.ddc_ext[e] : 0.0f);
        dg[0] = ddl * g_ * i_ * (1.0f - i_);
        dg[Ch] = dcn * a.c[e] * f_ * (1.0f - f_);
        dg[2 * Ch] = ddl * i_ * (1.0f - g_ * g_);
        if (a.dc) a.dc[e] = dcn * f_;
    }
    {   // m group
        const float* gs = a.gates_m + pix * 3 * Ch + ch;
        const float i_ = gs[0], f_ = gs[Ch], g_ = gs[2 * Ch];
        const float dmn = (a.dmn_ext ? a.dmn_ext[e] : 0.0f) + a.dmn_conv[e];
        const float ddl = dmn + (a.ddm_ext ? a.ddm_ext[e] : 0.0f);
        dg[4 * Ch] = ddl * g_ * i_ * (1.0f - i_);
        dg[5 * Ch] = dmn * a.m[e] * f_ * (1.0f - f_);
        dg[6 * Ch] = ddl * i_ * (1.0f - g_ * g_);
        a.dm[e] = dmn * f_;
    }
}

// The same two stages, eight channels of a pixel per thread (Ch % 8 == 0): 16-byte accesses throughout and — SPLIT — dG7 written
// straight in the split operand format ([pixel][group of 8 channels][8 hi bf16 | 8 lo bf16]) that the one-launch weight gradient
// (stw, wgrad2.hip) stages by LDS-DMA and that the data-gradient convolutions read without conversion (ConvSeg.split): no fp32
// copy of dG7 exists then and the separate split_convert pass (117 MB read + written per cell step at B = 128) is gone.
typedef float f32x4_b __attribute__((ext_vector_type(4)));
struct F8 { f32x4_b a, b; };
__device__ __forceinline__ F8 ld8(const float* p) { return F8{*reinterpret_cast<const f32x4_b*>(p), *reinterpret_cast<const f32x4_b*>(p + 4)}; }
__device__ __forceinline__ F8 ld8z(const float* p, long long e) { return p ? ld8(p + e) : F8{f32x4_b{0.f, 0.f, 0.f, 0.f}, f32x4_b{0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ float f8get(const F8& v, int i) { return i < 4 ? v.a[i] : v.b[i - 4]; }
template <bool SPLIT>
__device__ __forceinline__ void st8(float* base, const float (&v)[8]) {   // base: the 8-channel group's 32 bytes
    if constexpr (SPLIT) {
        unsigned h[8], l[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const __bf16 hb = (__bf16)v[i];
            h[i] = __builtin_bit_cast(unsigned short, hb);
            const __bf16 lb = (__bf16)(v[i] - __builtin_bit_cast(float, h[i] << 16));
            l[i] = __builtin_bit_cast(unsigned short, lb);
        }
        uint4* d = reinterpret_cast<uint4*>(base);
        d[0] = uint4{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
        d[1] = uint4{l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
    } else {
        *reinterpret_cast<f32x4_b*>(base) = f32x4_b{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4_b*>(base + 4) = f32x4_b{v[4], v[5], v[6], v[7]};
    }
}

template <bool SPLIT>
__global__ __launch_bounds__(256) void st_bwd_out8_kernel(const STBwdOutArgs a) {
    const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const int G = a.Ch >> 3;
    if (t >= a.n / 8) return;
    const long long pix = t / G;
    const int ch = (int)(t - pix * G) * 8;
    const long long e = pix * a.Ch + ch;
    const F8 dh = ld8z(a.dh_new, e), o = ld8(a.o + e), tl = ld8(a.tl + e);
    float go[8], gl[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float d = f8get(dh, i), oo = f8get(o, i), tt = f8get(tl, i);
        go[i] = d * tt * oo * (1.0f - oo);
        gl[i] = d * oo * (1.0f - tt * tt);
    }
    st8<SPLIT>(a.dG7 + pix * a.ldG + a.o_off + ch, go);
    if (a.dlc_off >= 0) st8<SPLIT>(a.dG7 + pix * a.ldG + a.dlc_off + ch, gl);
    if (a.dlc) st8<false>(a.dlc + e, gl);
}

template <bool SPLIT>
__global__ __launch_bounds__(256) void st_bwd_gates8_kernel(const STBwdGateArgs a) {
    const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const int Ch = a.Ch, G = Ch >> 3;
    if (t >= a.npix * G) return;
    const long long pix = t / G;
    const int ch = (int)(t - pix * G) * 8;
    const long long e = pix * Ch + ch;
    float* dg = a.dG7 + pix * a.ldG + ch;
    {
        const float* gs = a.gates_c + pix * 3 * Ch + ch;
        const F8 i_ = ld8(gs), f_ = ld8(gs + Ch), g_ = ld8(gs + 2 * Ch), cv = ld8(a.c + e), dc1 = ld8z(a.dcn_ext, e), dc2 = ld8(a.dcn_conv + e),
                 dd = ld8z(a.ddc_ext, e);
        float r0[8], r1[8], r2[8], r3[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float ii = f8get(i_, k), ff = f8get(f_, k), gg = f8get(g_, k);
            const float dcn = f8get(dc1, k) + f8get(dc2, k);
            const float ddl = dcn + f8get(dd, k);
            r0[k] = ddl * gg * ii * (1.0f - ii);
            r1[k] = dcn * f8get(cv, k) * ff * (1.0f - ff);
            r2[k] = ddl * ii * (1.0f - gg * gg);
            r3[k] = dcn * ff;
        }
        st8<SPLIT>(dg, r0); st8<SPLIT>(dg + Ch, r1); st8<SPLIT>(dg + 2 * Ch, r2);
        if (a.dc) st8<false>(a.dc + e, r3);
    }
    {
        const float* gs = a.gates_m + pix * 3 * Ch + ch;
        const F8 i_ = ld8(gs), f_ = ld8(gs + Ch), g_ = ld8(gs + 2 * Ch), mv = ld8(a.m + e), dm1 = ld8z(a.dmn_ext, e), dm2 = ld8(a.dmn_conv + e),
                 dd = ld8z(a.ddm_ext, e);
        float r0[8], r1[8], r2[8], r3[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float ii = f8get(i_, k), ff = f8get(f_, k), gg = f8get(g_, k);
            const float dmn = f8get(dm1, k) + f8get(dm2, k);
            const float ddl = dmn + f8get(dd, k);
            r0[k] = ddl * gg * ii * (1.0f - ii);
            r1[k] = dmn * f8get(mv, k) * ff * (1.0f - ff);
            r2[k] = ddl * ii * (1.0f - gg * gg);
            r3[k] = dmn * ff;
        }
        st8<SPLIT>(dg + 4 * Ch, r0); st8<SPLIT>(dg + 5 * Ch, r1); st8<SPLIT>(dg + 6 * Ch, r2);
        st8<false>(a.dm + e, r3);
    }
}

static bool st_vec8_ok(int Ch, int ldG, const void* p0, const void* p1) {
    return !(Ch & 7) && !(ldG & 7) && !(reinterpret_cast<uintptr_t>(p0) & 15) && !(reinterpret_cast<uintptr_t>(p1) & 15);
}

hipError_t launch_st_bwd_out(const STBwdOutArgs& a, hipStream_t s) {
    if (a.split || st_vec8_ok(a.Ch, a.ldG, a.dG7, a.dlc_off >= 0 ? (const void*)a.dG7 : (const void*)a.dlc)) {
        const long long n8 = a.n / 8;
        if (a.split) VPX_LAUNCH(st_bwd_out8_kernel<true>, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, a);
        else VPX_LAUNCH(st_bwd_out8_kernel<false>, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, a);
        return vpx_hip_last_error();
    }
    VPX_LAUNCH(st_bwd_out_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, s, a);
    return vpx_hip_last_error();
}
hipError_t launch_st_bwd_gates(const STBwdGateArgs& a, hipStream_t s) {
    if (a.split || st_vec8_ok(a.Ch, a.ldG, a.dG7, a.dm)) {
        const long long n8 = a.npix * (a.Ch >> 3);
        if (a.split) VPX_LAUNCH(st_bwd_gates8_kernel<true>, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, a);
        else VPX_LAUNCH(st_bwd_gates8_kernel<false>, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, a);
        return vpx_hip_last_error();
    }
    const long long n = a.npix * a.Ch;
    VPX_LAUNCH(st_bwd_gates_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    return vpx_hip_last_error();
}

}  // namespace vpx
