// conv_gemm.hip — LDS-tiled implicit-GEMM "same" convolution for gfx950 with fused recurrent-cell epilogues.
//
// One kernel template serves every dense contraction of the hot path:
//   ConvLSTM step       conv([x_t | h_{t-1}]) -> 4Ch, epilogue = peephole gates + (c,h) update
//                       (replaces the cat/conv2d/chunk/sigmoid/tanh/mul/add sequence of conv_lstm_hzzone.py:59-68
//                        and conv_lstm_ndrplz.py:31-41)
//   plain conv          bias + store (data-gradient convs of BPTT, ST-LSTM conv_last, PredRNN frame head)
//
// fp32 path: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 64 FLOP/clk/SIMD = the chip's fp32 matrix peak).
// Wave tile: 32 pixels x (NG x 32 channels); A fragment = one ds_read_b128 (4 consecutive channels of the lane's pixel),
// B fragment = one ds_read_b128 per group (4 consecutive k of the lane's output channel); 4 MFMAs per group per read.
// MFMA k-pairing: lanes 0-31 supply k = kb+s, lanes 32-63 supply k = kb+4+s (s = 0..3) for BOTH operands, so a b128
// read feeds four 32x32x2 steps covering 8 consecutive k.
#include "vpx_internal.h"

namespace vpx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigmoid_f(float v) { return 1.0f / (1.0f + expf(-v)); }

// ---------------------------------------------------------------------------------------------------------------
// weight repack: reference OIHW -> [n_tile][chunk][n = g*32+j, g < NG][kk]  (kk = position inside the KC-deep chunk)
// ---------------------------------------------------------------------------------------------------------------
__global__ void pack_weights_kernel(const PackDesc pd, float* __restrict__ dst) {
    const int ntr = pd.NG * 32;  // rows per chunk
    const long long total = (long long)pd.n_tiles * pd.chunks_total * ntr * KC_F32;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int kk = (int)(e % KC_F32);
        long long r = e / KC_F32;
        const int n = (int)(r % ntr);
        r /= ntr;
        const int chunk = (int)(r % pd.chunks_total);
        const int n_tile = (int)(r / pd.chunks_total);
        // which stage does this chunk belong to?
        int s = 0;
        for (int i = 1; i < pd.nstage; ++i)
            if (chunk >= pd.stage[i].chunk0) s = i;
        const ConvStage st = pd.stage[s];
        const int kin = (chunk - st.chunk0) * KC_F32 + kk;  // k index inside the stage = tap*cn + (c - c0)
        float v = 0.0f;
        const int g = n >> 5, j = n & 31;
        if (kin < st.nq * 8 && g < pd.NG && pd.rowbase[st.seg][g] >= 0) {
            const int tap = kin / st.cn;
            const int c = st.c0 + kin % st.cn;
            const PackSeg sg = pd.seg[st.seg];
            const int chan = n_tile * pd.tile_stride + pd.goff[g] + j;
            if (c < sg.C && chan < pd.nch) {
                const int row = pd.rowbase[st.seg][g] + n_tile * pd.tile_stride + j;
                const int tp = pd.flip ? (pd.taps - 1 - tap) : tap;
                if (!pd.transposed)
                    v = sg.w[(long long)row * sg.ld_o + (long long)(sg.coff + c) * sg.ld_i + tp];
                else
                    v = sg.w[(long long)(sg.coff + c) * sg.ld_o + (long long)row * sg.ld_i + tp];
            }
        }
        dst[e] = v;
    }
}

hipError_t launch_pack_weights(const PackDesc& pd, float* dst, hipStream_t s) {
    const long long total = (long long)pd.n_tiles * pd.chunks_total * pd.NG * 32 * KC_F32;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, s, pd, dst);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// epilogues
// ---------------------------------------------------------------------------------------------------------------
struct TileCtx {
    int b, y0, x0, n_tile, wave, j, hh, H, W;
};

// accumulator register r of a 32x32 MFMA tile holds row (r&3) + 8*(r>>2) + 4*hh; wave-local pixel index = that row.
__device__ __forceinline__ bool tile_pixel(const TileCtx& t, int r, int& y, int& x) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * t.hh;
    y = t.y0 + 2 * t.wave + (i >> 4);
    x = t.x0 + (i & 15);
    return y < t.H && x < t.W;
}

struct EpiConvLSTM {
    static constexpr int NG = 4;
    ConvLSTMStepArgs a;
    __device__ __forceinline__ void operator()(const f32x16 (&acc)[4], const TileCtx& t) const {
        const int ch = t.n_tile * 32 + t.j;
        if (ch >= a.Ch) return;
        const int Ch = a.Ch;
        float bi = 0.f, bf = 0.f, bg = 0.f, bo = 0.f;
        if (a.bias) {
            bi = a.bias[a.gate_pos[0] * Ch + ch];
            bf = a.bias[a.gate_pos[1] * Ch + ch];
            bg = a.bias[a.gate_pos[2] * Ch + ch];
            bo = a.bias[a.gate_pos[3] * Ch + ch];
        }
        const size_t img = (size_t)t.b * t.H * t.W;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int y, x;
            if (!tile_pixel(t, r, y, x)) continue;
            const size_t pix = (size_t)y * t.W + x;
            const size_t sidx = (img + pix) * Ch + ch;
            const float cp = a.c_in ? a.c_in[sidx] : 0.0f;
            float ai = acc[0][r] + bi, af = acc[1][r] + bf, ag = acc[2][r] + bg, ao = acc[3][r] + bo;
            if (a.wci) {  // peepholes on the previous cell state (conv_lstm_hzzone.py:64-65)
                ai += a.wci[pix * Ch + ch] * cp;
                af += a.wcf[pix * Ch + ch] * cp;
            }
            const float i_ = sigmoid_f(ai), f_ = sigmoid_f(af), g_ = tanhf(ag);
            const float cn = f_ * cp + i_ * g_;
            if (a.wco) ao += a.wco[pix * Ch + ch] * cn;  // peephole on the NEW cell state (:67)
            const float o_ = sigmoid_f(ao);
            const float hn = o_ * tanhf(cn);
            a.c_out[sidx] = cn;
            a.h_out[(size_t)t.b * a.h_bstride + pix * Ch + ch] = hn;
            if (a.gates) {
                float* gs = a.gates + (img + pix) * 4 * Ch + ch;
                gs[0] = i_;
                gs[Ch] = f_;
                gs[2 * Ch] = g_;
                gs[3 * Ch] = o_;
            }
        }
    }
};

// ST-LSTM gate groups (predrnn.py:61-77). NGROUPS = 4: (i, f, g, o_pre) -> c ; NGROUPS = 3: (i', f', g') -> m.
template <int NGROUPS>
struct EpiSTGate {
    static constexpr int NG = NGROUPS;
    STGateArgs a;
    __device__ __forceinline__ void operator()(const f32x16 (&acc)[NGROUPS], const TileCtx& t) const {
        const int ch = t.n_tile * 32 + t.j;
        if (ch >= a.Ch) return;
        const int Ch = a.Ch;
        const size_t img = (size_t)t.b * t.H * t.W;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int y, x;
            if (!tile_pixel(t, r, y, x)) continue;
            const size_t pidx = img + (size_t)y * t.W + x;
            const size_t sidx = pidx * Ch + ch;
            const float i_ = sigmoid_f(acc[0][r]);
            const float f_ = sigmoid_f(acc[1][r] + a.forget_bias);
            const float g_ = tanhf(acc[2][r]);
            const float dlt = i_ * g_;
            a.s_new[sidx] = f_ * a.s_in[sidx] + dlt;
            a.delta[sidx] = dlt;
            if constexpr (NGROUPS == 4) a.o_pre[sidx] = acc[3][r];
            if (a.gates) {
                float* gs = a.gates + pidx * 3 * Ch + ch;
                gs[0] = i_;
                gs[Ch] = f_;
                gs[2 * Ch] = g_;
            }
        }
    }
};

struct EpiSTOut {
    static constexpr int NG = 1;
    STOutArgs a;
    __device__ __forceinline__ void operator()(const f32x16 (&acc)[1], const TileCtx& t) const {
        const int ch = t.n_tile * 32 + t.j;
        if (ch >= a.Ch) return;
        const size_t img = (size_t)t.b * t.H * t.W;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int y, x;
            if (!tile_pixel(t, r, y, x)) continue;
            const size_t sidx = (img + (size_t)y * t.W + x) * a.Ch + ch;
            const float o_ = sigmoid_f(a.o_pre[sidx] + acc[0][r]);  // predrnn.py:80
            const float tl = tanhf(a.lc[sidx]);                     // predrnn.py:81
            a.h_new[sidx] = o_ * tl;
            if (a.o_save) { a.o_save[sidx] = o_; a.tl_save[sidx] = tl; }
        }
    }
};

struct EpiPlain {
    static constexpr int NG = 4;
    PlainEpiArgs a;
    __device__ __forceinline__ void operator()(const f32x16 (&acc)[4], const TileCtx& t) const {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = t.n_tile * NT + g * 32 + t.j;
            if (co >= a.Co) continue;
            const float bv = a.bias ? a.bias[co] : 0.0f;
            float* dst;
            long long bs;
            int ld, cc;
            if (co < a.split) { dst = a.out0; bs = a.bstride0; ld = a.ld0; cc = co; }
            else { dst = a.out1; bs = a.bstride1; ld = a.ld1; cc = co - a.split; }
            if (!dst) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int y, x;
                if (!tile_pixel(t, r, y, x)) continue;
                float* p = dst + (size_t)t.b * bs + ((size_t)y * t.W + x) * ld + cc;
                const float v = acc[g][r] + bv;
                *p = a.accumulate ? (*p + v) : v;
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------
// main kernel, fp32 operands
// ---------------------------------------------------------------------------------------------------------------
constexpr int WROW_F32 = KC_F32 * 4 + 16;  // padded LDS row of one output channel's chunk slice (144 B: 9 x 16 B, odd)

template <class Epi>
__global__ __launch_bounds__(NTHREADS) void conv_gemm_f32_kernel(const ConvPlan P, const Epi epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NG = Epi::NG;
    constexpr int NTR = NG * 32;                 // weight rows (output channels x gate groups) per workgroup
    constexpr int WBUF_F32 = NTR * WROW_F32;     // one LDS weight buffer
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hh = lane >> 5;

    int mt = blockIdx.x;
    const int tx = mt % P.tiles_x;
    mt /= P.tiles_x;
    const int ty = mt % P.tiles_y;
    const int b = mt / P.tiles_y;
    const int n_tile = blockIdx.y;
    const int x0 = tx * TILE_W, y0 = ty * TILE_H;
    const int halo_w = TILE_W + P.kw - 1, halo_h = TILE_H + P.kh - 1;
    const int npos = halo_w * halo_h;
    const int ph = P.kh / 2, pw = P.kw / 2;

    char* A_lds = smem;
    char* W_lds = smem + P.a_bytes;

    f32x16 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.0f;

    // this lane's pixel inside the tile (MFMA row = lane & 31)
    const int py = 2 * wave + (j >> 4), px = j & 15;
    const float* wtile = P.wpk + (size_t)n_tile * P.chunks_total * (NTR * KC_F32);

    for (int s = 0; s < P.nstage; ++s) {
        const ConvStage st = P.stage[s];
        const ConvSeg sg = P.seg[st.seg];
        const int arow = st.cn * 4 + 16;  // bytes per halo position (odd multiple of 16 B -> conflict-free b128 reads)
        __syncthreads();                  // previous stage fully consumed
        // ---- stage the activation halo tile: positions x [c0, c0+cn) ----
        {
            const float* src = sg.ptr + (size_t)b * sg.bstride;
            const int ld = sg.ld ? sg.ld : sg.C;
            if (((sg.C | ld) & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
                const int v4n = st.cn >> 2;
                const int total = npos * v4n;
                for (int v = tid; v < total; v += NTHREADS) {
                    const int pos = v / v4n, c4 = v - pos * v4n;
                    const int hy = pos / halo_w, hx = pos - hy * halo_w;
                    const int gy = y0 - ph + hy, gx = x0 - pw + hx;
                    const int c = st.c0 + c4 * 4;
                    f32x4 val = {0.f, 0.f, 0.f, 0.f};
                    if (gy >= 0 && gy < P.H && gx >= 0 && gx < P.W && c < sg.C)
                        val = *reinterpret_cast<const f32x4*>(src + ((size_t)gy * P.W + gx) * ld + c);
                    *reinterpret_cast<f32x4*>(A_lds + pos * arow + c4 * 16) = val;
                }
            } else {
                const int total = npos * st.cn;
                for (int e = tid; e < total; e += NTHREADS) {
                    const int pos = e / st.cn, cc = e - pos * st.cn;
                    const int hy = pos / halo_w, hx = pos - hy * halo_w;
                    const int gy = y0 - ph + hy, gx = x0 - pw + hx;
                    const int c = st.c0 + cc;
                    float val = 0.f;
                    if (gy >= 0 && gy < P.H && gx >= 0 && gx < P.W && c < sg.C)
                        val = src[((size_t)gy * P.W + gx) * ld + c];
                    *reinterpret_cast<float*>(A_lds + pos * arow + cc * 4) = val;
                }
            }
        }
        // ---- weight chunk 0 of this stage ----
        const int nchunks = (st.nq * 8 + KC_F32 - 1) / KC_F32;
        const f32x4* wsrc = reinterpret_cast<const f32x4*>(wtile + (size_t)st.chunk0 * (NTR * KC_F32));
        f32x4 wr[NG];
#pragma unroll
        for (int it = 0; it < NG; ++it) wr[it] = wsrc[tid + it * NTHREADS];
#pragma unroll
        for (int it = 0; it < NG; ++it) {
            const int v = tid + it * NTHREADS;
            *reinterpret_cast<f32x4*>(W_lds + (v >> 3) * WROW_F32 + (v & 7) * 16) = wr[it];
        }
        __syncthreads();

        const int c8n = st.cn >> 3;
        int c8 = 0, tdx = 0, tdy = 0;
        const char* a_lane = A_lds + (py * halo_w + px) * arow + hh * 16;
        int tapoff = 0;
        for (int ck = 0; ck < nchunks; ++ck) {
            const bool more = ck + 1 < nchunks;
            if (more) {
                const f32x4* wn = wsrc + (size_t)(ck + 1) * (NTR * KC_F32 / 4);
#pragma unroll
                for (int it = 0; it < NG; ++it) wr[it] = wn[tid + it * NTHREADS];
            }
            const char* wb = W_lds + (ck & 1) * WBUF_F32 + j * WROW_F32 + hh * 16;
#pragma unroll
            for (int q = 0; q < KC_F32 / 8; ++q) {
                if (ck * (KC_F32 / 8) + q < st.nq) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4*>(a_lane + tapoff + c8 * 32);
                    f32x4 b4[NG];
#pragma unroll
                    for (int g = 0; g < NG; ++g)
                        b4[g] = *reinterpret_cast<const f32x4*>(wb + g * 32 * WROW_F32 + q * 32);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int g = 0; g < NG; ++g)
                            acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[k], b4[g][k], acc[g], 0, 0, 0);
                    if (++c8 == c8n) {
                        c8 = 0;
                        if (++tdx == P.kw) { tdx = 0; ++tdy; }
                        tapoff = (tdy * halo_w + tdx) * arow;
                    }
                }
            }
            if (more) {
                char* wdst = W_lds + ((ck + 1) & 1) * WBUF_F32;
#pragma unroll
                for (int it = 0; it < NG; ++it) {
                    const int v = tid + it * NTHREADS;
                    *reinterpret_cast<f32x4*>(wdst + (v >> 3) * WROW_F32 + (v & 7) * 16) = wr[it];
                }
            }
            __syncthreads();
        }
    }

    TileCtx t{b, y0, x0, n_tile, wave, j, hh, P.H, P.W};
    epi(acc, t);
}

template <class Epi>
static hipError_t launch_conv(const ConvPlan& plan, const Epi& epi, int n_tiles, hipStream_t s) {
    const size_t lds = (size_t)plan.a_bytes + 2 * (Epi::NG * 32 * WROW_F32);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_f32_kernel<Epi>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    dim3 grid(plan.B * plan.tiles_x * plan.tiles_y, n_tiles);
    hipLaunchKernelGGL(conv_gemm_f32_kernel<Epi>, grid, dim3(NTHREADS), lds, s, plan, epi);
    return hipGetLastError();
}

hipError_t launch_convlstm_step_f32(const ConvPlan& plan, const ConvLSTMStepArgs& ea, int n_tiles, hipStream_t s) {
    EpiConvLSTM e{ea};
    return launch_conv(plan, e, n_tiles, s);
}

hipError_t launch_conv_plain_f32(const ConvPlan& plan, const PlainEpiArgs& ea, int n_tiles, hipStream_t s) {
    EpiPlain e{ea};
    return launch_conv(plan, e, n_tiles, s);
}

hipError_t launch_st_cgroup_f32(const ConvPlan& plan, const STGateArgs& ea, int n_tiles, hipStream_t s) {
    EpiSTGate<4> e{ea};
    return launch_conv(plan, e, n_tiles, s);
}
hipError_t launch_st_mgroup_f32(const ConvPlan& plan, const STGateArgs& ea, int n_tiles, hipStream_t s) {
    EpiSTGate<3> e{ea};
    return launch_conv(plan, e, n_tiles, s);
}
hipError_t launch_st_out_f32(const ConvPlan& plan, const STOutArgs& ea, int n_tiles, hipStream_t s) {
    EpiSTOut e{ea};
    return launch_conv(plan, e, n_tiles, s);
}

// ---------------------------------------------------------------------------------------------------------------
// host-side plan helpers
// ---------------------------------------------------------------------------------------------------------------
int build_stages(ConvStage* st, int* chunks_total, const int* segC, int nseg, int taps, int cs, int kc) {
    int n = 0, chunk = 0;
    for (int sgi = 0; sgi < nseg; ++sgi) {
        const int Cp = (segC[sgi] + 7) / 8 * 8;
        for (int c0 = 0; c0 < Cp; c0 += cs) {
            if (n >= MAX_STAGE) return -1;
            const int cn = (Cp - c0 < cs) ? (Cp - c0) : cs;
            ConvStage s{};
            s.seg = sgi;
            s.c0 = c0;
            s.cn = cn;
            s.chunk0 = chunk;
            s.nq = taps * cn / 8;
            st[n++] = s;
            chunk += (s.nq * 8 + kc - 1) / kc;
        }
    }
    *chunks_total = chunk;
    return n;
}

int conv_a_bytes(const ConvStage* st, int nstage, int kh, int kw) {
    const int npos = (TILE_H + kh - 1) * (TILE_W + kw - 1);
    int m = 16;
    for (int i = 0; i < nstage; ++i) {
        const int bytes = npos * (st[i].cn * 4 + 16);
        if (bytes > m) m = bytes;
    }
    return (m + 15) / 16 * 16;
}

size_t packed_weight_bytes(int n_tiles, int chunks_total, int ng) {
    return (size_t)n_tiles * chunks_total * ng * 32 * KC_F32 * sizeof(float);
}

}  // namespace vpx
