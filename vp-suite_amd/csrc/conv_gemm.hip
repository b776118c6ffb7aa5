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
#include <stdlib.h>

#include "vpx_internal.h"

namespace vpx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_bits(float v) {
    __bf16 h = (__bf16)v;  // v_cvt_pk_bf16_f32: round to nearest even
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }
// v = hi + lo, both round-to-nearest-even bf16: |v - (hi + lo)| <= 2^-18 |v|. (A truncated hi saves ~2 VALU per
// element but doubles the error: measured 2.5e-5 vs 1.3e-5 on a 4-step cell — not worth it.)
__device__ __forceinline__ void split_bf16(float v, unsigned short& hi, unsigned short& lo) {
    hi = bf16_bits(v);
    lo = bf16_bits(v - bf16_to_f32(hi));
}

__host__ __device__ inline int mode_kstep(int prec) { return prec == VPX_PREC_F32 ? 8 : 16; }  // bf16x3 and bf16 share one layout   // channels per k-step
// k-depth of a weight chunk = qpc k-steps (2 by default; 3 for 3x3 kernels in the bf16 modes: 9 k-steps per 16-channel stage
// are then exactly 3 chunks instead of 4.5 -> 5 with a half-empty last one)
__host__ __device__ inline int mode_kc(int prec, int qpc = 2) { return mode_kstep(prec) * (qpc == 3 ? 3 : 2); }

// ---------------------------------------------------------------------------------------------------------------
// weight repack: reference OIHW -> [n_tile][chunk][n = g*32+j, g < NG][kk]  (kk = position inside the KC-deep chunk)
// ---------------------------------------------------------------------------------------------------------------
__global__ void pack_weights_kernel(const PackDesc pd, char* __restrict__ dst) {
    const int ntr = pd.NG * 32;                       // rows per chunk
    const int kc = mode_kc(pd.prec, pd.qpc);          // k-depth of a chunk
    const int row_bytes = kc * 4;                     // fp32: kc floats; bf16x3: kc hi-bf16 then kc lo-bf16
    const long long total = (long long)pd.n_tiles * pd.chunks_total * ntr * kc;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int kk = (int)(e % kc);
        long long r = e / kc;
        const int n = (int)(r % ntr);
        r /= ntr;
        const int chunk = (int)(r % pd.chunks_total);
        const int n_tile = (int)(r / pd.chunks_total);
        int s = 0;  // which stage does this chunk belong to?
        for (int i = 1; i < pd.nstage; ++i)
            if (chunk >= pd.stage[i].chunk0) s = i;
        const ConvStage st = pd.stage[s];
        const int kin = (chunk - st.chunk0) * kc + kk;  // k index inside the stage = tap*cn + (c - c0)
        float v = 0.0f;
        const int g = n >> 5, j = n & 31;
        if (kin < st.nq * mode_kstep(pd.prec) && g < pd.NG && pd.rowbase[st.seg][g] >= 0) {
            const int tap = kin / st.cn;
            const int c = st.c0 + kin % st.cn;
            const PackSeg sg = pd.seg[st.seg];
            const int chan = n_tile * pd.tile_stride + pd.goff[g] + j;
            if (c < sg.C && chan < pd.nch) {
                const int row = pd.rowbase[st.seg][g] + n_tile * pd.tile_stride + j;
                const int tp = pd.src_taps ? pd.tapmap[tap] : (pd.flip ? (pd.taps - 1 - tap) : tap);
                if (!pd.transposed)
                    v = sg.w[(long long)row * sg.ld_o + (long long)(sg.coff + c) * sg.ld_i + tp];
                else
                    v = sg.w[(long long)(sg.coff + c) * sg.ld_o + (long long)row * sg.ld_i + tp];
            }
        }
        char* rowp = dst + ((long long)(n_tile * pd.chunks_total + chunk) * ntr + n) * row_bytes;
        if (pd.prec == VPX_PREC_F32) {
            reinterpret_cast<float*>(rowp)[kk] = v;
        } else {
            unsigned short hi, lo;
            split_bf16(v, hi, lo);
            reinterpret_cast<unsigned short*>(rowp)[kk] = hi;
            reinterpret_cast<unsigned short*>(rowp)[kc + kk] = lo;
        }
    }
}

hipError_t launch_pack_weights(const PackDesc& pd, float* dst, hipStream_t s) {
    const long long total = (long long)pd.n_tiles * pd.chunks_total * pd.NG * 32 * mode_kc(pd.prec, pd.qpc);
    if (!ws_write_ok(dst, (size_t)total * 4, "weight pack (pack_weights_kernel)")) return hipErrorInvalidValue;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    VPX_LAUNCH(pack_weights_kernel, dim3(blocks), dim3(256), 0, s, pd, reinterpret_cast<char*>(dst));
    return vpx_hip_last_error();
}

// ---------------------------------------------------------------------------------------------------------------
// epilogues
// ---------------------------------------------------------------------------------------------------------------
struct TileCtx {
    int b, y0, x0, n_tile, prow, j, hh, H, W;  // prow = first row (inside the workgroup tile) of this 32-pixel MFMA tile
    int kz;                                    // K-split index of this workgroup (0 when the contraction is not split)
};

// accumulator register r of a 32x32 MFMA tile holds row (r&3) + 8*(r>>2) + 4*hh; wave-local pixel index = that row.
__device__ __forceinline__ bool tile_pixel(const TileCtx& t, int r, int& y, int& x) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * t.hh;
    y = t.y0 + t.prow + (i >> 4);
    x = t.x0 + (i & 15);
    return y < t.H && x < t.W;
}

struct EpiConvLSTM {
    static constexpr int NG = 4;
    static constexpr bool SPLITK = false;
    static constexpr bool Q3OK = true;
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
        // RB pixels per batch: their state / peephole loads are all issued before the first gate is evaluated, so a wave
        // exposes 16/RB memory latencies instead of 16 (stores may alias loads, the compiler will not hoist them itself)
        constexpr int RB = 4;
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += RB) {
            bool ok[RB];
            size_t pixv[RB];
            float cpv[RB], wi[RB], wf[RB], wo[RB];
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                int y, x;
                ok[u] = tile_pixel(t, r0 + u, y, x);
                pixv[u] = (size_t)y * t.W + x;
                cpv[u] = 0.f; wi[u] = 0.f; wf[u] = 0.f; wo[u] = 0.f;
                if (ok[u]) {
                    if (a.c_in) cpv[u] = a.c_in[(img + pixv[u]) * Ch + ch];
                    if (a.wci) { wi[u] = a.wci[pixv[u] * Ch + ch]; wf[u] = a.wcf[pixv[u] * Ch + ch]; }
                    if (a.wco) wo[u] = a.wco[pixv[u] * Ch + ch];
                }
            }
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                if (!ok[u]) continue;
                const int r = r0 + u;
                const size_t pix = pixv[u];
                const size_t sidx = (img + pix) * Ch + ch;
                const float cp = cpv[u];
                // peepholes on the previous cell state (conv_lstm_hzzone.py:64-65); absent peepholes contribute 0
                const float ai = acc[0][r] + bi + wi[u] * cp, af = acc[1][r] + bf + wf[u] * cp;
                const float ag = acc[2][r] + bg;
                const float i_ = sigmoid_f(ai), f_ = sigmoid_f(af), g_ = tanh_f(ag);
                const float cn = lstm_c(f_, cp, i_, g_);
                const float ao = acc[3][r] + bo + wo[u] * cn;  // peephole on the NEW cell state (:67)
                const float o_ = sigmoid_f(ao);
                const float hn = o_ * tanh_f(cn);
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
    }
};

// ST-LSTM gate groups (predrnn.py:61-77). NGROUPS = 4: (i, f, g, o_pre) -> c ; NGROUPS = 3: (i', f', g') -> m.
template <int NGROUPS>
struct EpiSTGate {
    static constexpr int NG = NGROUPS;
    static constexpr bool SPLITK = false;
    static constexpr bool Q3OK = false;
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
            const float g_ = tanh_f(acc[2][r]);
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
    static constexpr bool SPLITK = false;
    static constexpr bool Q3OK = false;
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
            const float tl = tanh_f(a.lc[sidx]);                     // predrnn.py:81
            a.h_new[sidx] = o_ * tl;
            if (a.o_save) { a.o_save[sidx] = o_; a.tl_save[sidx] = tl; }
        }
    }
};

template <int NGP>
struct EpiPlain {
    static constexpr int NG = NGP;
    static constexpr bool SPLITK = true;
    static constexpr bool Q3OK = true;
    PlainEpiArgs a;
    __device__ __forceinline__ void operator()(const f32x16 (&acc)[NGP], const TileCtx& t) const {
#pragma unroll
        for (int g = 0; g < NGP; ++g) {
            const int co = t.n_tile * (NGP * 32) + g * 32 + t.j;
            if (co >= a.Co) continue;
            const float bv = (a.bias && t.kz == 0) ? a.bias[co] : 0.0f;
            float* dst;
            long long bs;
            int ld, cc;
            if (co < a.split) { dst = a.out0; bs = a.bstride0; ld = a.ld0; cc = co; }
            else { dst = a.out1; bs = a.bstride1; ld = a.ld1; cc = co - a.split; }
            if (!dst && !a.sp_out) continue;
            // split copy: even lanes carry the hi pair (co, co+1), odd lanes the lo pair (co-1, co): one dword store per lane
            const int coe = co & ~1;
            const unsigned sp_off = (unsigned)((coe >> 3) * 32 + (coe & 7) * 2 + ((t.j & 1) ? 16 : 0));
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int y, x;
                if (!tile_pixel(t, r, y, x)) continue;
                const size_t pix = a.omap ? ((size_t)(y * a.oys + a.oyo) * a.Wmem + (x * a.oxs + a.oxo)) : ((size_t)y * t.W + x);
                float* p = dst ? dst + (size_t)t.b * bs + pix * ld + cc : nullptr;
                float v = acc[g][r] + bv;
                if (a.ksplit > 1) {  // K split over workgroups: partial sums meet in memory (destination pre-zeroed or +=)
                    unsafeAtomicAdd(p, v);
                    continue;
                }
                if (a.accumulate) v += *p;   // the activation applies to the completed sum (two convolutions into one output)
                if (a.leaky != 0.0f) v = v > 0.0f ? v : v * a.leaky;
                if (p) *p = v;
                if (a.sp_out) {
                    unsigned short h16, l16;
                    split_bf16(v, h16, l16);
                    const unsigned hi = h16, lo = l16;
                    const unsigned nhi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, 0xB1, 0xF, 0xF, true);   // lane j ^ 1
                    const unsigned nlo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, 0xB1, 0xF, 0xF, true);
                    const unsigned word = (t.j & 1) ? (nlo | (lo << 16)) : (hi | (nhi << 16));
                    *reinterpret_cast<unsigned*>(a.sp_out + (size_t)t.b * a.sp_bstride + pix * ((size_t)a.Co * 4) + sp_off) = word;
                }
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------
// main kernel. MODE 0: fp32 operands, v_mfma_f32_32x32x2_f32 (exact). MODE 1: split-bf16 ("bf16x3"): every operand is
// hi + lo (two bf16), the product is hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation —
// fp32-level accuracy (measured 4e-6 over a full 10->10 forward) at 3/16 of the fp32 MFMA cycles.
// Both modes use the same LDS footprints: an activation row is cn*4 B (+16 pad): fp32 values, or cn hi-bf16 followed by
// cn lo-bf16; a weight row is one chunk of one output channel: KC fp32, or KC hi followed by KC lo.
// ---------------------------------------------------------------------------------------------------------------

// 4 channels c .. c+3 (c % 4 == 0) of a pixel row: fp32 tensor -> the values; split-format tensor -> 8 bytes of hi halves + 8 of lo
__device__ __forceinline__ f32x4 stage_load(const float* pixel_row, int c, const bool split) {
    if (!split) return *reinterpret_cast<const f32x4*>(pixel_row + c);
    const char* p = reinterpret_cast<const char*>(pixel_row) + (c >> 3) * 32 + (c & 7) * 2;
    const uint2 h = *reinterpret_cast<const uint2*>(p), l = *reinterpret_cast<const uint2*>(p + 16);
    return __builtin_bit_cast(f32x4, uint4{h.x, h.y, l.x, l.y});
}

// one 4-channel vector of the activation halo tile -> LDS (fp32 as is; bf16 modes: hi half-row, then lo half-row)
// (split source: `val` already holds the 4 hi halves in its first 8 bytes and the 4 lo halves in the last 8)
template <int MODE>
__device__ __forceinline__ void stage_store(char* dst, int lo_off, const f32x4 val, const bool split = false) {
    if constexpr (MODE == 0) {
        *reinterpret_cast<f32x4*>(dst) = val;
    } else if (split) {
        const uint4 u = __builtin_bit_cast(uint4, val);
        *reinterpret_cast<uint2*>(dst) = uint2{u.x, u.y};
        *reinterpret_cast<uint2*>(dst + lo_off) = uint2{u.z, u.w};
    } else {
        unsigned short h0, h1, h2, h3, l0, l1, l2, l3;
        split_bf16(val[0], h0, l0); split_bf16(val[1], h1, l1);
        split_bf16(val[2], h2, l2); split_bf16(val[3], h3, l3);
        uint2 hv = {(unsigned)h0 | ((unsigned)h1 << 16), (unsigned)h2 | ((unsigned)h3 << 16)};
        uint2 lv = {(unsigned)l0 | ((unsigned)l1 << 16), (unsigned)l2 | ((unsigned)l3 << 16)};
        *reinterpret_cast<uint2*>(dst) = hv;
        *reinterpret_cast<uint2*>(dst + lo_off) = lv;
    }
}

// timing ablations (VPX_DBG bits) exist only in builds with -DVPX_ABLATE; the product kernel carries none of the tests
#ifdef VPX_ABLATE
#define DBGBIT(b) (P.dbg & (b))
#else
#define DBGBIT(b) false
#endif

template <int MODE, int QPCN = 2> struct ModeTraits {  // QPCN = k-steps per weight chunk (2 or 3)
    static constexpr int KSTEP = MODE == 0 ? 8 : 16;   // plain bf16 (MODE 2) shares the bf16x3 layout, only hi planes are read
    static constexpr int KC = KSTEP * QPCN;
    static constexpr int WROW_DATA = KC * 4;           // fp32: KC floats; bf16 modes: KC hi-bf16 then KC lo-bf16
};

// MW = 1: 4 waves (256 threads) per workgroup, 8x16 pixel tile. MW = 2: 8 waves (512 threads), 16x16 tile — twice the
// pixels share every weight chunk (the L2->CU load pipe, ~70 GB/s per CU, is what limits the bf16 modes), at <= 128
// registers so that two such workgroups (16 waves) stay resident per CU.
template <class Epi, int MODE, int MW, int MS = 1, int QPCN = 2>
__device__ __forceinline__ void conv_body(const ConvPlan& P, const Epi& epi, const int n_tile, const int m_tile) {
    constexpr int NTH = NTHREADS * MW;
    using MT = ModeTraits<MODE, QPCN>;
    constexpr int KC = MT::KC, KSTEP = MT::KSTEP;
    constexpr int WROW = MT::WROW_DATA + 16;   // padded LDS row of one output channel's chunk slice (odd multiple of 16 B)
    constexpr int QPC = KC / KSTEP;            // k-steps per chunk (2 in both modes)
    constexpr int NG = Epi::NG;
    constexpr int NTR = NG * 32;               // weight rows (gate groups x 32 channels) per workgroup
    constexpr int WBUF = NTR * WROW;           // one LDS weight buffer
    constexpr int WV4 = NTR * MT::WROW_DATA / 16;              // 16-byte vectors per weight chunk
    constexpr int WIT = (WV4 + NTH - 1) / NTH;                 // staging loads per thread per chunk
    constexpr int V4ROW = MT::WROW_DATA / 16;                  // 16-byte vectors per weight row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hh = lane >> 5;

    int mt = m_tile;
    const int tx = mt % P.tiles_x;
    mt /= P.tiles_x;
    const int ty = mt % P.tiles_y;
    const int b = mt / P.tiles_y;
    constexpr int TH = TILE_H * MW * MS;  // workgroup tile height: every wave owns MS sub-tiles of 2 rows x 16 pixels
    const int x0 = tx * TILE_W, y0 = ty * TH;
    const int sd = P.stride > 1 ? P.stride : 1;  // input step per output pixel
    const int halo_w = (TILE_W - 1) * sd + P.kw, halo_h = (TH - 1) * sd + P.kh;
    const int npos = halo_w * halo_h;
    const int ph = P.use_org ? -P.org_y : P.kh / 2, pw = P.use_org ? -P.org_x : P.kw / 2;  // halo origin = tile origin*sd - (ph, pw)
    const int Hin = P.Hin ? P.Hin : P.H, Win = P.Win ? P.Win : P.W;

    char* A_lds = smem;
    char* W_lds = smem + P.a_bytes;
    if (DBGBIT(256)) return;  // ablation: launch + dispatch floor

    f32x16 acc[MS][NG];
#pragma unroll
    for (int m = 0; m < MS; ++m)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][g][r] = 0.0f;

    // this lane's pixel inside the tile (MFMA row = lane & 31): wave w owns tile rows 2*MS*w .. 2*MS*w + 2*MS-1
    const int py = 2 * MS * wave + (j >> 4), px = j & 15;
    const char* wtile = reinterpret_cast<const char*>(P.wpk) + (size_t)n_tile * P.chunks_total * (NTR * MT::WROW_DATA);

    // ---- K loop ------------------------------------------------------------------------------------------------
    // Weight chunks form one stream over all stages: chunk g is multiplied out of LDS buffer g&1 while chunk g+1 travels
    // from L2 into registers (requested at the top of the iteration, stored to the other buffer after the MFMAs); a
    // stage boundary costs no weight bubble. One register set and no parity-dependent register choice: the two-set
    // variant made the compiler copy sets and wait for a load right after issuing it.
    // The loop keeps its scalar bookkeeping minimal (one cursor per stream, incremental tap offsets): scalar and vector
    // ALU work between two MFMA groups delays the wave's next MFMA issue.
    constexpr int CHUNK_BYTES = NTR * MT::WROW_DATA;
    // prefetch cursor: (stage, chunks left in it, this thread's source address). Stages of absent operands are not in
    // the plan, so the packed stream can have gaps at stage boundaries — the cursor jumps to the next stage's chunk0.
    auto stage_chunks = [&](int s) { return (P.stage[s].nq + QPC - 1) / QPC; };
    // K split (plain convolutions on small maps): workgroup z of ksplit owns the stages [s_begin, s_end)
    const int ksplit = P.ksplit > 1 ? P.ksplit : 1;
    const int kz = ksplit > 1 ? (int)blockIdx.z : 0;
    const int s_begin = kz * P.nstage / ksplit, s_end = (kz + 1) * P.nstage / ksplit;
    int ps = s_begin, prem = s_begin < s_end ? stage_chunks(s_begin) : 0;
    const char* wnext = wtile + (size_t)(s_begin < s_end ? P.stage[s_begin].chunk0 : 0) * CHUNK_BYTES + tid * 16;
    auto issue_load = [&](f32x4 (&wr)[WIT]) {  // returns silently past the end of the stream
        if (ps < s_end) {
#pragma unroll
            for (int it = 0; it < WIT; ++it)
                if (tid + it * NTH < WV4 && !DBGBIT(4)) wr[it] = *reinterpret_cast<const f32x4*>(wnext + it * NTH * 16);
            wnext += CHUNK_BYTES;
            if (--prem == 0 && ++ps < s_end) {
                prem = stage_chunks(ps);
                wnext = wtile + (size_t)P.stage[ps].chunk0 * CHUNK_BYTES + tid * 16;
            }
        }
    };
    // LDS destination of this thread's staging vectors (row padding applied once)
    int wdst_off[WIT];
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
        const int v = tid + it * NTH;
        wdst_off[it] = (v / V4ROW) * WROW + (v % V4ROW) * 16;
    }
    auto write_lds = [&](const f32x4 (&wr)[WIT], int buf) {
        char* wdst = W_lds + buf * WBUF;
#pragma unroll
        for (int it = 0; it < WIT; ++it)
            if (tid + it * NTH < WV4 && !DBGBIT(64)) *reinterpret_cast<f32x4*>(wdst + wdst_off[it]) = wr[it];
    };
    f32x4 wr[WIT];
#pragma unroll
    for (int it = 0; it < WIT; ++it) wr[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    issue_load(wr);  // chunk 0

    int gidx = 0;  // chunks consumed so far (parity selects LDS buffer / register set)
    const char* wb0 = W_lds + j * WROW + hh * 16;
    for (int s = s_begin; s < s_end; ++s) {
        const ConvStage st = P.stage[s];
        const ConvSeg sg = P.seg[st.seg];
        const int arow = st.cn * 4 + 16;  // bytes per halo position (odd multiple of 16 B -> conflict-free b128 reads)
        // (the barrier that ended the previous chunk guarantees the previous stage's tile is fully consumed)
        // ---- stage the activation halo tile: positions x [c0, c0+cn) ----
        {
            const float* src = sg.ptr + (size_t)b * sg.bstride;
            const int ld = sg.ld ? sg.ld : sg.C;
            if (DBGBIT(128)) {
                } else if (((sg.C | ld) & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
                const int v4n = st.cn >> 2;
                if ((v4n & (v4n - 1)) == 0) {
                    // cn/4 is a power of two (always for 16/32/64-channel stages): 256 threads cover 256/v4n halo
                    // positions per pass and (pos, hy, hx) advance by constants — no integer division in the loop
                    const int sh = 31 - __builtin_clz(v4n);
                    const int c4 = tid & (v4n - 1);
                    const int dpos = NTH >> sh;
                    const int dhy = dpos / halo_w, dhx = dpos - dhy * halo_w;
                    int pos = tid >> sh;
                    int hy = pos / halo_w, hx = pos - hy * halo_w;
                    const int c = st.c0 + c4 * 4;
                    const bool c_ok = c < sg.C && !DBGBIT(2);
                    const bool spl = MODE != 0 && sg.split != 0;
                    char* dstc = A_lds + (MODE == 0 ? c4 * 16 : c4 * 8);
                    if (DBGBIT(16)) {
                    for (; pos < npos; pos += dpos) {
                        const int gy = y0 * sd - ph + hy, gx = x0 * sd - pw + hx;
                        f32x4 val = {0.f, 0.f, 0.f, 0.f};
                        if (c_ok && gy >= 0 && gy < Hin && gx >= 0 && gx < Win)
                            val = stage_load(src + ((size_t)gy * Win + gx) * ld, c, spl);
                        stage_store<MODE>(dstc + pos * arow, st.cn * 2, val, spl);
                        hx += dhx; hy += dhy;
                        if (hx >= halo_w) { hx -= halo_w; ++hy; }
                    }
                    } else {
                    // AU loads in flight per thread before the first conversion: one exposed memory latency per AU
                    // positions instead of one per position (the MFMA fragment registers are dead here)
                    constexpr int AU = (MW >= 2 ? 3 : 4);
                    for (; pos < npos; pos += dpos * AU) {
                        f32x4 val[AU];
#pragma unroll
                        for (int u = 0; u < AU; ++u) {
                            const int gy = y0 * sd - ph + hy, gx = x0 * sd - pw + hx;
                            val[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                            if (pos + u * dpos < npos && c_ok && gy >= 0 && gy < Hin && gx >= 0 && gx < Win)
                                val[u] = stage_load(src + ((size_t)gy * Win + gx) * ld, c, spl);
                            hx += dhx; hy += dhy;
                            if (hx >= halo_w) { hx -= halo_w; ++hy; }
                        }
#pragma unroll
                        for (int u = 0; u < AU; ++u)
                            if (pos + u * dpos < npos) stage_store<MODE>(dstc + (pos + u * dpos) * arow, st.cn * 2, val[u], spl);
                    }
                    }
                } else {
                    const int total = npos * v4n;
                    for (int v = tid; v < total; v += NTH) {
                        const int pos = v / v4n, c4 = v - pos * v4n;
                        const int hy = pos / halo_w, hx = pos - hy * halo_w;
                        const int gy = y0 * sd - ph + hy, gx = x0 * sd - pw + hx;
                        const int c = st.c0 + c4 * 4;
                        f32x4 val = {0.f, 0.f, 0.f, 0.f};
                        const bool spl = MODE != 0 && sg.split != 0;
                        if (gy >= 0 && gy < Hin && gx >= 0 && gx < Win && c < sg.C && !DBGBIT(2))
                            val = stage_load(src + ((size_t)gy * Win + gx) * ld, c, spl);
                        if constexpr (MODE == 0) {
                            *reinterpret_cast<f32x4*>(A_lds + pos * arow + c4 * 16) = val;
                        } else if (spl) {
                            stage_store<MODE>(A_lds + pos * arow + c4 * 8, st.cn * 2, val, true);
                        } else {
                            unsigned short h0, h1, h2, h3, l0, l1, l2, l3;
                            split_bf16(val[0], h0, l0); split_bf16(val[1], h1, l1);
                            split_bf16(val[2], h2, l2); split_bf16(val[3], h3, l3);
                            uint2 hv = {(unsigned)h0 | ((unsigned)h1 << 16), (unsigned)h2 | ((unsigned)h3 << 16)};
                            uint2 lv = {(unsigned)l0 | ((unsigned)l1 << 16), (unsigned)l2 | ((unsigned)l3 << 16)};
                            *reinterpret_cast<uint2*>(A_lds + pos * arow + c4 * 8) = hv;
                            *reinterpret_cast<uint2*>(A_lds + pos * arow + st.cn * 2 + c4 * 8) = lv;
                        }
                    }
                }
            } else {
                const int total = npos * st.cn;
                for (int e = tid; e < total; e += NTH) {
                    const int pos = e / st.cn, cc = e - pos * st.cn;
                    const int hy = pos / halo_w, hx = pos - hy * halo_w;
                    const int gy = y0 * sd - ph + hy, gx = x0 * sd - pw + hx;
                    const int c = st.c0 + cc;
                    float val = 0.f;
                    if (gy >= 0 && gy < Hin && gx >= 0 && gx < Win && c < sg.C)
                        val = src[((size_t)gy * Win + gx) * ld + c];
                    if constexpr (MODE == 0) {
                        *reinterpret_cast<float*>(A_lds + pos * arow + cc * 4) = val;
                    } else {
                        unsigned short h, l;
                        split_bf16(val, h, l);
                        *reinterpret_cast<unsigned short*>(A_lds + pos * arow + cc * 2) = h;
                        *reinterpret_cast<unsigned short*>(A_lds + pos * arow + st.cn * 2 + cc * 2) = l;
                    }
                }
            }
        }


        const int ksn = st.cn / KSTEP;
        const int tap_dx = arow, tap_dy = (halo_w - P.kw) * arow;  // tap offset steps: next column / wrap to next row
        int ks = 0, tdx = 0, tapoff = 0;
        const char* a_lane = A_lds + (py * sd * halo_w + px * sd) * arow + hh * 16;
        const int sub_off = 2 * sd * halo_w * arow;  // next sub-tile of this wave: two tile rows further down
        if (s == s_begin) write_lds(wr, 0);  // very first chunk of the stream
        __syncthreads();
        for (int kq = 0; kq < st.nq; kq += QPC) {
            const int buf = gidx & 1;
            const char* wb = wb0 + buf * WBUF;
            const bool has_next = kq + QPC < st.nq || s + 1 < s_end;
            if (has_next) issue_load(wr);  // chunk g+1 travels while chunk g is multiplied
            __builtin_amdgcn_s_setprio(2);  // MFMA bursts outrank co-resident waves that are staging (+1-2 % measured)
#pragma unroll
            for (int q = 0; q < QPC; ++q) {
                if (q == 0 || kq + q < st.nq) {
                    if (DBGBIT(1)) {
                    } else if constexpr (MODE == 0) {
                        // fp32: one b128 = 4 consecutive channels; lanes 0-31 take k = 8*ks + s, lanes 32-63 k = 8*ks + 4 + s
                        f32x4 a4[MS];
#pragma unroll
                        for (int m = 0; m < MS; ++m)
                            a4[m] = *reinterpret_cast<const f32x4*>(a_lane + m * sub_off + tapoff + ks * 32);
                        f32x4 b4[NG];
#pragma unroll
                        for (int g = 0; g < NG; ++g)
                            b4[g] = *reinterpret_cast<const f32x4*>(wb + g * 32 * WROW + q * 32);
#pragma unroll
                        for (int k = 0; k < 4; ++k)
#pragma unroll
                            for (int g = 0; g < NG; ++g)
#pragma unroll
                                for (int m = 0; m < MS; ++m)
                                    acc[m][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[m][k], b4[g][k], acc[m][g], 0, 0, 0);
                    } else {
                        // bf16x3: one b128 = 8 consecutive channels; lanes 0-31 take k = 16*ks + 0..7, lanes 32-63 + 8..15
                        bf16x8 ah[MS], al[MS];
#pragma unroll
                        for (int m = 0; m < MS; ++m) {
                            const char* ap = a_lane + m * sub_off + tapoff + ks * 32;
                            ah[m] = *reinterpret_cast<const bf16x8*>(ap);
                            if constexpr (MODE == 1) al[m] = *reinterpret_cast<const bf16x8*>(ap + st.cn * 2);
                        }
                        // gates in pairs: keeps the live weight fragments at 2 x (hi, lo) = 16 registers
#pragma unroll
                        for (int g0 = 0; g0 < NG; g0 += 2) {
                            bf16x8 bh[2], bl[2];
#pragma unroll
                            for (int gg = 0; gg < 2; ++gg) {
                                if (g0 + gg < NG) {
                                    bh[gg] = *reinterpret_cast<const bf16x8*>(wb + (g0 + gg) * 32 * WROW + q * 32);
                                    if constexpr (MODE == 1)
                                        bl[gg] = *reinterpret_cast<const bf16x8*>(wb + (g0 + gg) * 32 * WROW + KC * 2 + q * 32);
                                }
                            }
#pragma unroll
                            for (int gg = 0; gg < 2; ++gg) {
                                if (g0 + gg < NG) {
#pragma unroll
                                    for (int m = 0; m < MS; ++m) {
                                        f32x16& c = acc[m][g0 + gg];
                                        if constexpr (MODE == 1) {  // bf16x3: the two cross terms; plain bf16 (MODE 2) keeps hi*hi only
                                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh[gg], c, 0, 0, 0);
                                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl[gg], c, 0, 0, 0);
                                        }
                                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh[gg], c, 0, 0, 0);
                                    }
                                }
                            }
                        }
                    }
                    if (++ks == ksn) {
                        ks = 0;
                        tapoff += tap_dx;
                        if (++tdx == P.kw) { tdx = 0; tapoff += tap_dy; }
                    }
                }
            }
            __builtin_amdgcn_s_setprio(0);
            if (has_next) write_lds(wr, buf ^ 1);  // every wave finished reading that buffer before the last barrier
            if (!DBGBIT(32)) __syncthreads();  // (bit 32: timing-only ablation of the per-chunk barrier; results are wrong)
            ++gidx;
        }
    }

    if (DBGBIT(8)) {  // ablation: keep the accumulators alive with one store instead of the epilogue
        if (acc[0][0][0] == 12345.678f) reinterpret_cast<float*>(const_cast<float*>(P.wpk))[0] = acc[MS - 1][0][1];
        return;
    }
#pragma unroll
    for (int m = 0; m < MS; ++m) {
        TileCtx t{b, y0, x0, n_tile, 2 * (MS * wave + m), j, hh, P.H, P.W, kz};
        epi(acc[m], t);
    }
}

// XCD-aware workgroup -> tile mapping. The hardware hands consecutive workgroup ids to the 8 XCDs round-robin, and every
// XCD has its own L2. The launch is a 1-D grid over (pixel tile, N tile) pairs (gridDim.y carries the N-tile count for
// the decode only when the launcher could not flatten); id L runs on XCD L % 8 as that XCD's (L / 8)-th workgroup.
// Each XCD gets a CONTIGUOUS range of the tile sequence, N tile fastest: the N tiles of one pixel tile (they read the same
// activation halo) and neighbouring pixel tiles (overlapping halos) then run back to back on ONE XCD and meet in its L2
// instead of each fetching from HBM.
__device__ __forceinline__ bool xcd_tile(const ConvPlan& P, unsigned L, unsigned n_tiles_y, int& m_tile, int& n_tile) {
    const int n_tiles = P.grid_n;
    if (n_tiles <= 0) {  // legacy 2-D launch: x = pixel tile, y = N tile
        m_tile = (int)L; n_tile = (int)blockIdx.y;
        return true;
    }
    const long long total = (long long)P.grid_m * n_tiles;
    const long long per_xcd = (total + 7) / 8;
    const long long s = (long long)(L & 7) * per_xcd + (L >> 3);
    if ((long long)(L >> 3) >= per_xcd || s >= total) return false;
    m_tile = (int)(s / n_tiles);
    n_tile = (int)(s - (long long)m_tile * n_tiles);
    (void)n_tiles_y;
    return true;
}

template <class Epi, int MODE, int MW, int MS = 1, int QPCN = 2>
__global__ __launch_bounds__(NTHREADS * MW, (MS == 2 ? 2 : (MW >= 2 ? 4 : 3))) void conv_gemm_kernel(const ConvPlan P, const Epi epi) {
    int m_tile, n_tile;
    if (!xcd_tile(P, blockIdx.x, gridDim.y, m_tile, n_tile)) return;
    conv_body<Epi, MODE, MW, MS, QPCN>(P, epi, n_tile, m_tile);
}

// Two independent contractions over the same pixel tiling in ONE launch (blockIdx.y < nA -> A, else B): the ST-LSTM's
// c-group and m-group each fill only one workgroup per CU on 16x16 maps; together they give every CU two.
template <class EpiA, class EpiB, int MODE, int MW>
__global__ __launch_bounds__(NTHREADS * MW, (MW == 2 ? 4 : 3)) void conv_gemm_dual_kernel(const ConvPlan PA, const EpiA epiA, const int nA,
                                                                                          const ConvPlan PB, const EpiB epiB) {
    int m_tile, n_tile;
    if (!xcd_tile(PA, blockIdx.x, gridDim.y, m_tile, n_tile)) return;
    if (n_tile < nA) conv_body<EpiA, MODE, MW>(PA, epiA, n_tile, m_tile);
    else conv_body<EpiB, MODE, MW>(PB, epiB, n_tile - nA, m_tile);
}

static bool xcd_map_enabled() {  // VPX_XCD_MAP=0 restores the plain 2-D grid (experiments)
    static int on = -1;
    if (on < 0) on = dev_switch("VPX_XCD_MAP", 1);
    return on != 0;
}

template <class Epi, int MODE, int MW, int MS = 1, int QPCN = 2>
static hipError_t launch_conv_m(const ConvPlan& plan, const Epi& epi, int n_tiles, hipStream_t s) {
    const size_t lds = (size_t)plan.a_bytes + 2 * (Epi::NG * 32 * (ModeTraits<MODE, QPCN>::WROW_DATA + 16));
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = vpx_func_attr(reinterpret_cast<const void*>(&conv_gemm_kernel<Epi, MODE, MW, MS, QPCN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = !g_dry_run;
    }
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    ConvPlan p2 = plan;
    p2.grid_m = plan.B * plan.tiles_x * plan.tiles_y;
    p2.grid_n = xcd_map_enabled() ? n_tiles : 0;
    const long long per_xcd = ((long long)p2.grid_m * n_tiles + 7) / 8;
    dim3 grid = p2.grid_n ? dim3((unsigned)(per_xcd * 8), 1, plan.ksplit > 1 ? plan.ksplit : 1)
                          : dim3(p2.grid_m, n_tiles, plan.ksplit > 1 ? plan.ksplit : 1);
    VPX_LAUNCH((conv_gemm_kernel<Epi, MODE, MW, MS, QPCN>), grid, dim3(NTHREADS * MW), lds, s, p2, epi);
    return vpx_hip_last_error();
}

template <class Epi>
static hipError_t launch_conv(const ConvPlan& plan_in, const Epi& epi, int n_tiles, hipStream_t s) {
    static int dbg = -1;
    if (dbg < 0) dbg = dev_switch("VPX_DBG", 0);
    ConvPlan plan = plan_in;
    plan.dbg = dbg;
    if (!Epi::SPLITK) plan.ksplit = 0;  // only the plain epilogue can combine partial sums
    const int mw = plan.mw >= 4 ? 4 : (plan.mw > 1 ? 2 : 1);
    if ((plan.H + TILE_H * mw - 1) / (TILE_H * mw) != plan.tiles_y) return hipErrorInvalidValue;  // host geometry mismatch
    if (plan.prec == VPX_PREC_F32) return plan.qpc == 3 ? hipErrorInvalidValue : launch_conv_m<Epi, 0, 1>(plan, epi, n_tiles, s);  // fp32 is MFMA-bound: MW=1 only
    if (plan.prec == VPX_PREC_BF16X3) {
        static int ms = -1;  // 16x16 tile as 4 waves x 2 sub-tiles (VPX_MS=2) instead of 8 waves x 1
        if (ms < 0) ms = dev_switch("VPX_MS", 1);
        if constexpr (Epi::Q3OK) {  // 3 k-steps per weight chunk (ConvLSTM cell and its data gradient on 3x3 kernels)
            if (plan.qpc == 3 && mw <= 2 && ms != 2)
                return mw == 2 ? launch_conv_m<Epi, 1, 2, 1, 3>(plan, epi, n_tiles, s) : launch_conv_m<Epi, 1, 1, 1, 3>(plan, epi, n_tiles, s);
        }
        if (plan.qpc == 3) return hipErrorInvalidValue;  // the weights were packed for a form that is not instantiated
        if (mw == 4) return launch_conv_m<Epi, 1, 4>(plan, epi, n_tiles, s);
        if (mw == 2 && ms == 2) return launch_conv_m<Epi, 1, 1, 2>(plan, epi, n_tiles, s);
        return mw == 2 ? launch_conv_m<Epi, 1, 2>(plan, epi, n_tiles, s) : launch_conv_m<Epi, 1, 1>(plan, epi, n_tiles, s);
    }
    if (plan.prec == VPX_PREC_BF16) {
        if constexpr (Epi::Q3OK) {
            if (plan.qpc == 3 && mw <= 2)
                return mw == 2 ? launch_conv_m<Epi, 2, 2, 1, 3>(plan, epi, n_tiles, s) : launch_conv_m<Epi, 2, 1, 1, 3>(plan, epi, n_tiles, s);
        }
        if (plan.qpc == 3) return hipErrorInvalidValue;
        return mw == 2 ? launch_conv_m<Epi, 2, 2>(plan, epi, n_tiles, s) : launch_conv_m<Epi, 2, 1>(plan, epi, n_tiles, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_convlstm_step_f32(const ConvPlan& plan, const ConvLSTMStepArgs& ea, int n_tiles, hipStream_t s) {
    EpiConvLSTM e{ea};
    return launch_conv(plan, e, n_tiles, s);
}

hipError_t launch_conv_plain_f32(const ConvPlan& plan_in, const PlainEpiArgs& ea_in, int n_tiles, hipStream_t s) {
    ConvPlan plan = plan_in;
    PlainEpiArgs ea = ea_in;
    plan.ksplit = ea.ksplit = (plan.ksplit > 1 && plan.ksplit <= plan.nstage) ? plan.ksplit : 0;
    if (plan.ksplit > 1 && ea.leaky != 0.0f) return hipErrorInvalidValue;  // partial sums cannot be activated
    switch (ea.ng) {
        case 1: return launch_conv(plan, EpiPlain<1>{ea}, n_tiles, s);
        case 2: return launch_conv(plan, EpiPlain<2>{ea}, n_tiles, s);
        case 3: return launch_conv(plan, EpiPlain<3>{ea}, n_tiles, s);
        case 4: return launch_conv(plan, EpiPlain<4>{ea}, n_tiles, s);
    }
    return hipErrorInvalidValue;
}

int plain_groups(int Co, long long m_tiles) {
    // Every N tile stages the activation halo tile again, so the narrowest padding is not the cheapest split: cost of a
    // pixel tile ~ n_tiles * (STAGE + ng) in units of one 32-channel group's MFMA work, STAGE = 2 (the data-gradient
    // conv of the 64+96-channel ConvLSTM, Co = 160 at 32x32, B=128: five 32-wide tiles 921 us, two 96-wide tiles 862 us
    // although they multiply 20 % padding). Ties -> more groups per tile.
    constexpr int STAGE = 2;
    int best = 4, best_cost = 1 << 30;
    for (int ng = 4; ng >= 1; --ng) {
        const int tiles = (Co + ng * 32 - 1) / (ng * 32);
        const int cost = tiles * (STAGE + ng);
        if (cost < best_cost) { best_cost = cost; best = ng; }
    }
    (void)m_tiles;  // occupancy on small maps comes from the K split (pick_ksplit), not from narrower N tiles
    return best;
}

int pick_ksplit(long long wgs, int nstage, bool bwd) {
    static int forced = -1;
    if (forced < 0) forced = dev_switch("VPX_KSPLIT", 0);
    if (g_deterministic) return 1;
    int k = 1;
    if (forced > 0) k = forced;
    // aim at one workgroup per CU: measured on 16x16 maps (64 pixel tiles), PredRNN forward 25.8 ms with 12 splits,
    // 24.7 ms with 4, 28.7 ms with 3; ConvLSTM (96,96,16x16) B=32 (192 workgroups) 155 TF fused, 164 TF with 2 splits
    else if (wgs > 0 && wgs < 256) k = (int)((256 + wgs - 1) / wgs);
    // data-gradient convs at exactly one workgroup per CU unsplit (PredRNN's 16x16 maps at B=128: 256 pixel tiles x 1 N
    // tile): two halves of K per tile give every CU a second workgroup to overlap with (training step 260.5 -> 252.9 ms;
    // 4 splits 269.4). Not in the forward pass: its split convs pay a clear + a separate output pass (inference 58.9 -> 60.1 ms)
    else if (bwd && wgs >= 256 && wgs < 384) k = 2;
    if (k > nstage) k = nstage;
    if (k > 16) k = 16;
    return k < 1 ? 1 : k;
}

void fill_plain_pack(PackDesc& pd, int Co, int first, int ng_in) {
    const int ng = ng_in > 0 ? ng_in : plain_groups(Co);
    pd.NG = ng;
    for (int s = 0; s < MAX_SEG; ++s)
        for (int g = 0; g < MAX_NG; ++g) pd.rowbase[s][g] = g < ng ? first + g * 32 : -1;
    for (int g = 0; g < MAX_NG; ++g) pd.goff[g] = g * 32;
    pd.tile_stride = ng * 32;
    pd.nch = Co;
    pd.n_tiles = plain_tiles_ng(Co, ng);
}

hipError_t launch_st_cgroup_f32(const ConvPlan& plan, const STGateArgs& ea, int n_tiles, hipStream_t s) {
    EpiSTGate<4> e{ea};
    return launch_conv(plan, e, n_tiles, s);
}
hipError_t launch_st_mgroup_f32(const ConvPlan& plan, const STGateArgs& ea, int n_tiles, hipStream_t s) {
    EpiSTGate<3> e{ea};
    return launch_conv(plan, e, n_tiles, s);
}
template <int MODE, int MW>
static hipError_t launch_st_dual_m(const ConvPlan& pc, const STGateArgs& ec, const ConvPlan& pm, const STGateArgs& em,
                                   int n_tiles, hipStream_t s) {
    using KA = EpiSTGate<4>;
    using KB = EpiSTGate<3>;
    const size_t lc = (size_t)pc.a_bytes + 2 * (4 * 32 * (ModeTraits<MODE>::WROW_DATA + 16));
    const size_t lm = (size_t)pm.a_bytes + 2 * (3 * 32 * (ModeTraits<MODE>::WROW_DATA + 16));
    // both bodies carve LDS as [activation tile | weight buffers] from offset 0 with their own a_bytes: size for the larger
    const size_t lds = lc > lm ? lc : lm;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = vpx_func_attr(reinterpret_cast<const void*>(&conv_gemm_dual_kernel<KA, KB, MODE, MW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = !g_dry_run;
    }
    if (lds > 160 * 1024 || pc.tiles_x != pm.tiles_x || pc.tiles_y != pm.tiles_y || pc.B != pm.B) return hipErrorInvalidValue;
    ConvPlan pc2 = pc;
    pc2.grid_m = pc.B * pc.tiles_x * pc.tiles_y;
    pc2.grid_n = 0;  // the two groups have different weights and sources: interleaving them per XCD measured 2.6 % slower
    (void)&xcd_map_enabled;
    const long long per_xcd = ((long long)pc2.grid_m * 2 * n_tiles + 7) / 8;
    dim3 grid = pc2.grid_n ? dim3((unsigned)(per_xcd * 8), 1) : dim3(pc2.grid_m, 2 * n_tiles);
    VPX_LAUNCH((conv_gemm_dual_kernel<KA, KB, MODE, MW>), grid, dim3(NTHREADS * MW), lds, s, pc2, KA{ec}, n_tiles, pm, KB{em});
    return vpx_hip_last_error();
}

hipError_t launch_st_gates_dual(const ConvPlan& pc, const STGateArgs& ec, const ConvPlan& pm, const STGateArgs& em,
                                int n_tiles, hipStream_t s) {
    if (pc.prec != pm.prec) return hipErrorInvalidValue;
    const bool mw2 = pc.mw == 2 && pm.mw == 2;
    if ((pc.H + TILE_H * (mw2 ? 2 : 1) - 1) / (TILE_H * (mw2 ? 2 : 1)) != pc.tiles_y) return hipErrorInvalidValue;
    if (pc.prec == VPX_PREC_F32) return launch_st_dual_m<0, 1>(pc, ec, pm, em, n_tiles, s);
    if (pc.prec == VPX_PREC_BF16X3) return mw2 ? launch_st_dual_m<1, 2>(pc, ec, pm, em, n_tiles, s) : launch_st_dual_m<1, 1>(pc, ec, pm, em, n_tiles, s);
    if (pc.prec == VPX_PREC_BF16) return mw2 ? launch_st_dual_m<2, 2>(pc, ec, pm, em, n_tiles, s) : launch_st_dual_m<2, 1>(pc, ec, pm, em, n_tiles, s);
    return hipErrorInvalidValue;
}

hipError_t launch_st_out_f32(const ConvPlan& plan, const STOutArgs& ea, int n_tiles, hipStream_t s) {
    EpiSTOut e{ea};
    return launch_conv(plan, e, n_tiles, s);
}

// ---------------------------------------------------------------------------------------------------------------
// host-side plan helpers
// ---------------------------------------------------------------------------------------------------------------
int build_stages(ConvStage* st, int* chunks_total, const int* segC, int nseg, int taps, int cs, int prec, int qpc_in) {
    const int kstep = mode_kstep(prec), qpc = (qpc_in == 3 && prec != VPX_PREC_F32) ? 3 : 2;
    if (cs % kstep) cs = (cs + kstep - 1) / kstep * kstep;
    int n = 0, chunk = 0;
    for (int sgi = 0; sgi < nseg; ++sgi) {
        const int Cp = (segC[sgi] + kstep - 1) / kstep * kstep;
        for (int c0 = 0; c0 < Cp; c0 += cs) {
            if (n >= MAX_STAGE) return -1;
            const int cn = (Cp - c0 < cs) ? (Cp - c0) : cs;
            ConvStage s{};
            s.seg = sgi;
            s.c0 = c0;
            s.cn = cn;
            s.chunk0 = chunk;
            s.nq = taps * cn / kstep;
            st[n++] = s;
            chunk += (s.nq + qpc - 1) / qpc;
        }
    }
    *chunks_total = chunk;
    return n;
}

// Channels per activation stage: the smallest stage that still fits the stage table buys the most workgroups per CU
// (LDS = halo tile + double-buffered weight chunk; 144 registers cap residency at 3 waves/SIMD). Measured on MI355X,
// fp32, B=32: 64 ch -> 95 TF (1 WG/CU), 32 -> 118 TF (2), 16 -> 120-128 TF (3).  VPX_CS overrides for experiments.
int pick_mw(int B, int H, int W, int n_tiles, int prec) {
    static int forced = -1;
    if (forced < 0) forced = dev_switch("VPX_MW", 0);
    if (prec == VPX_PREC_F32) return 1;  // fp32 is MFMA-bound: the 8-wave form is not instantiated
    if (forced == 1 || forced == 2 || (forced == 4 && prec == VPX_PREC_BF16X3)) return forced;
    // 8-wave workgroups halve the weight traffic per pixel; worth it only when the launch still fills the chip
    const long long wgs2 = (long long)B * ((H + 2 * TILE_H - 1) / (2 * TILE_H)) * ((W + TILE_W - 1) / TILE_W) * n_tiles;
    return wgs2 >= 512 ? 2 : 1;
}

int pick_stage_channels(const int* segC, int nseg, int kh, int kw, int ng, int prec, int mw, int stride, int qpc) {
    static int forced = -1;
    if (forced < 0) {
        forced = dev_switch("VPX_CS", 0);
        if (forced < 8 || (forced & 7) || forced > CS_MAX) forced = 0;
    }
    const int kstep = mode_kstep(prec);
    if (forced) return forced < kstep ? kstep : forced;
    const int npos = ((TILE_H * mw - 1) * stride + kh) * ((TILE_W - 1) * stride + kw);
    const int wbytes = 2 * ng * 32 * (mode_kc(prec, qpc) * 4 + 16);
    const int wg_cap = mw == 4 ? 1 : (mw == 2 ? 2 : 3);  // residency: 2 x 8 waves or 3 x 4 waves per CU (register budgets 128 / 168)
    int best = CS_MAX, best_wg = 0;
    for (int cs = kstep; cs <= CS_MAX; cs *= 2) {
        int nst = 0;
        for (int s = 0; s < nseg; ++s) nst += ((segC[s] + kstep - 1) / kstep * kstep + cs - 1) / cs;
        if (nst > MAX_STAGE) continue;
        int wg = (160 * 1024) / (npos * (cs * 4 + 16) + wbytes);
        if (wg > wg_cap) wg = wg_cap;
        // equal residency: the larger stage (fewer stage switches) — except for a single 32-column group, whose stages hold so few
        // MFMAs that the shorter copy of a small stage wins (64 -> 16 3x3 at 64x64, 1280 frames: 0.92 -> 0.83 ms with 16-channel stages)
        if (wg > best_wg || (wg == best_wg && (ng == 1 ? best_wg == 0 : cs > best))) { best = cs; best_wg = wg; }
    }
    return best;
}

// Does the implicit-GEMM launch with these K segments fit — a stage table of at most MAX_STAGE entries and activation tile + weight
// ring within the CU's 160 KiB of LDS? (Round 5, found by tools/fuzz_contract.py: a 7x7 data gradient over 4 * 288 gate channels has no
// 16- or 32-channel stage table short enough, and its 64-channel stages need 168 KB in the 8-wave form: the layout functions now fall
// back to the 4-wave form, or refuse the descriptor, instead of failing at the launch.)
bool conv_fits_lds(const int* segC, int nseg, int kh, int kw, int ng, int prec, int mw, int stride, int qpc) {
    ConvStage st[MAX_STAGE];
    int chunks = 0;
    const int n = build_stages(st, &chunks, segC, nseg, kh * kw, pick_stage_channels(segC, nseg, kh, kw, ng, prec, mw, stride, qpc), prec, qpc);
    if (n < 0) return false;
    const size_t lds = (size_t)conv_a_bytes(st, n, kh, kw, mw, stride) + 2 * (size_t)ng * 32 * (mode_kc(prec, qpc) * 4 + 16);
    return lds <= 160 * 1024;
}

int conv_a_bytes(const ConvStage* st, int nstage, int kh, int kw, int mw, int stride) {
    const int npos = ((TILE_H * mw - 1) * stride + kh) * ((TILE_W - 1) * stride + kw);
    int m = 16;
    for (int i = 0; i < nstage; ++i) {
        const int bytes = npos * (st[i].cn * 4 + 16);
        if (bytes > m) m = bytes;
    }
    return (m + 15) / 16 * 16;
}

size_t packed_weight_bytes(int n_tiles, int chunks_total, int ng, int prec, int qpc) {
    return (size_t)n_tiles * chunks_total * ng * 32 * mode_kc(prec, qpc) * sizeof(float);
}

}  // namespace vpx
