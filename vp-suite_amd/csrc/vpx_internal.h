// vpx_internal.h — shared declarations of libvpx_hip.so (gfx950 only; no CUDA / multi-backend paths).
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/vpx.h"

namespace vpx {

// ---------------------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution geometry.
//   M = output pixels: one workgroup owns a TILE_H x TILE_W patch of one image (128 pixels, 4 waves x 32 pixels)
//   N = NG "gate groups" x 32 channels: every wave holds, for its 32 pixels, the NG accumulators of the SAME 32
//       channels (i,f,g,o of one channel sit in one lane -> the LSTM epilogue is lane-local, no shuffles)
//   K = (source segment, channel stage, tap, channel): activations are staged ONCE per channel stage as a halo tile in
//       LDS (NHWC rows, channel-contiguous) and re-used by all kh*kw taps; weights stream through LDS in KC-deep chunks.
// ---------------------------------------------------------------------------------------------------------------
constexpr int TILE_H = 8;
constexpr int TILE_W = 16;
constexpr int NTHREADS = 256;
constexpr int NT = 128;      // GEMM-N per workgroup (4 groups x 32)
constexpr int MAX_NG = 4;
constexpr int CS_MAX = 64;   // largest activation stage (channels); pick_stage_channels() chooses per problem
constexpr int MAX_SEG = 3;
constexpr int MAX_STAGE = 32;

struct ConvSeg {          // one activation source, NHWC [B][H][W][C] with arbitrary batch stride (selects a time slice)
    const float* ptr;
    long long bstride;    // elements between consecutive batch items
    int C;                // channels taken from each pixel row
    int ld;               // elements between consecutive pixels (0 = C): lets a segment be a channel slice of a wider tensor
    int split, _p;        // 1: the tensor is in the split-bf16 operand format (same strides in 4-byte units; bf16 modes, C % 8 == 0): staged without conversion
};

struct ConvStage {        // one K-stage: channels [c0, c0+cn) of segment `seg`, all taps
    int seg, c0, cn;      // cn is a multiple of the mode's kstep (zero-padded beyond the segment's real C)
    int chunk0;           // first weight chunk of this stage inside the packed per-tile weight stream
    int nq;               // number of k-steps = taps * cn / kstep (kstep = 8 fp32, 16 bf16x3)
    int _p0, _p1, _p2;
};

struct ConvPlan {
    int B, H, W, kh, kw, tiles_x, tiles_y;
    int nseg, nstage;
    int chunks_total;     // weight chunks per N-tile
    int a_bytes;          // LDS bytes reserved for the activation stage
    int prec;             // VPX_PREC_F32 | VPX_PREC_BF16X3: operand mode of the contraction
    int mw;               // 32-pixel MFMA row tiles per wave (1: 8x16 workgroup tile, 2: 16x16); 0 is read as 1
    int grid_m, grid_n;   // set by the launcher: pixel tiles / N tiles of the flattened, XCD-aware 1-D grid (grid_n = 0: 2-D grid)
    int qpc;              // k-steps per weight chunk the weights were packed with (0/2: two; 3: three, bf16 modes only)
    int ksplit;           // > 1: the stage list is split over blockIdx.z (plain epilogue only; partial sums via atomics)
    int dbg;              // ablation bits for profiling (VPX_DBG): 1 skip MFMAs, 2 skip activation loads, 4 skip weight loads, 8 skip epilogue
    // generalised geometry (all 0 = the stride-1 "same" convolution every recurrent cell uses):
    int stride;           // input step per output pixel (0/1 or 2)
    int use_org;          // 1: the halo origin of output pixel (0,0) is (org_y, org_x) instead of (-kh/2, -kw/2)
    int org_y, org_x;
    int Hin, Win;         // input image size when it differs from the tile-space size H x W (0 = same)
    ConvSeg seg[MAX_SEG];
    ConvStage stage[MAX_STAGE];
    const float* wpk;     // packed weights [n_tiles][chunks_total][NG*32][KC]
};

// Source description for the weight repack kernel: where does packed element (n_tile, g, j, stage, tap, c) come from?
struct PackSeg {
    const float* w;       // OIHW tensor holding this segment's weights
    long long ld_o;       // elements per "O" index
    long long ld_i;       // elements per "I" index (= kh*kw for OIHW)
    int coff;             // offset of this segment's channel 0 along the contraction ("I", or "O" if transposed) axis
    int C;                // real channels of the segment
};
struct PackDesc {
    PackSeg seg[MAX_SEG];
    ConvStage stage[MAX_STAGE];
    int nstage, chunks_total, n_tiles, taps;
    int qpc;              // k-steps per chunk (0/2 or 3), see mode_kc()
    int prec;             // operand mode the packing is for (must match the ConvPlan that consumes it)
    int NG;               // groups per tile actually used (rows of unused groups are zero)
    int rowbase[MAX_SEG][MAX_NG];  // per segment: source output-row of (channel 0, tile 0) for group g; -1 = none
    int goff[MAX_NG];     // channel-index offset of group g used only for the validity test
    int tile_stride;      // channels advanced per n_tile
    int nch;              // number of valid output channels
    int transposed;       // 1: contraction runs over the weight's O axis, outputs over its I axis (data-gradient conv)
    int flip;             // 1: spatially flipped taps (data-gradient conv)
    int src_taps;         // taps of the SOURCE weight tensor when the packed kernel uses a subset / re-ordering of them (0 = same)
    int tapmap[16];       // with src_taps: packed tap t reads source tap tapmap[t]
};

void set_error(const char* fmt, ...);

// ---- dry run (vpx_set_option(VPX_OPT_DRY_RUN, 1)) -----------------------------------------------------------------
// Every entry point does ALL of its host-side work — argument checks, plans, workspace carving, the ws_write_ok() checks of
// the launchers — and skips the HIP runtime calls themselves. Needs no GPU: the CPU test-suite sweeps the models' layer
// shapes through every entry point this way (tests/test_workspace_contract.py), so a sizing rule that drifts from a launch
// fails in the container, not as memory corruption on the GPU box. Every HIP runtime call of the library goes through the
// wrappers below. A dry run leaves no state behind that a later real launch depends on: the launchers record a function attribute as set
// only when it really was (`attr_set = !g_dry_run`), so a process may switch the option off again and launch
// (tests/test_gpu_more.py::test_real_launch_after_dry_run).
extern int g_dry_run;
#define VPX_LAUNCH(...) do { if (!::vpx::g_dry_run) hipLaunchKernelGGL(__VA_ARGS__); } while (0)
static inline hipError_t vpx_hip_last_error() { return g_dry_run ? hipSuccess : hipGetLastError(); }
static inline hipError_t vpx_func_attr(const void* f, hipFuncAttribute a, int v) { return g_dry_run ? hipSuccess : hipFuncSetAttribute(f, a, v); }
static inline hipError_t vpx_memset_async(void* p, int v, size_t n, hipStream_t s) { return g_dry_run ? hipSuccess : hipMemsetAsync(p, v, n, s); }
static inline hipError_t vpx_memcpy_async(void* d, const void* s_, size_t n, hipMemcpyKind k, hipStream_t s) { return g_dry_run ? hipSuccess : hipMemcpyAsync(d, s_, n, k, s); }
static inline hipError_t vpx_memcpy2d_async(void* d, size_t dp, const void* s_, size_t sp, size_t w, size_t h, hipMemcpyKind k, hipStream_t s) {
    return g_dry_run ? hipSuccess : hipMemcpy2DAsync(d, dp, s_, sp, w, h, k, s);
}

// ---- workspace accounting ---------------------------------------------------------------------------------------
// Every extern "C" entry that is handed a workspace carves it with ONE Carver, which registers itself for the duration of the
// call (thread-local). The launchers that write derived data of a computed size (weight packs, K-slice slabs, operand
// conversions, partial sums) ask ws_write_ok() first: a write that starts inside a carved slot must end inside it, a write
// that starts elsewhere in the workspace must end before the workspace does. A size rule that drifted between a
// `*_workspace_bytes` query and the launch code is then an error return (VPX_ERR_WORKSPACE), never a write into memory the
// caller did not hand over (round 4's intermittent abort: the 5x5 convolutions of the TrajGRU flow generator packed
// 26 weight chunks per 32 channels into a workspace sized for 25).
static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
struct Carver {  // bump allocator over the caller's workspace
    struct Slot { const char* p; size_t n; };
    static constexpr int MAX_SLOTS = 64;
    char* base;
    size_t off, cap;
    Slot slot[MAX_SLOTS];
    int nslot;
    bool over;       // a take() ran past `cap`: the layout needs more than the caller gave
    Carver* prev;
    Carver(void* workspace, size_t bytes);   // starts at the first 256-byte boundary; registers itself (vpx_api.hip)
    ~Carver();
    Carver(const Carver&) = delete;
    Carver& operator=(const Carver&) = delete;
    float* take(size_t nfloat) {
        const size_t n = align256(nfloat * sizeof(float));
        float* p = reinterpret_cast<float*>(base + off);
        if (off > cap || n > cap - off) over = true;
        else if (nslot < MAX_SLOTS) slot[nslot++] = Slot{base + off, n};
        off += n;
        return p;
    }
    bool ok() const { return !over; }
    size_t used() const { return off; }
};
// true when [dst, dst + bytes) is a legal write for the running call (see above); false + ws_violation() text otherwise
bool ws_write_ok(const void* dst, size_t bytes, const char* what);
const char* ws_violation();   // "" when the running call had none
void ws_violation_clear();
// after carving: the layout must fit what the caller handed over
#define VPX_CHECK_CARVE(ws, who)                                                                                      \
    do {                                                                                                              \
        if (!(ws).ok()) {                                                                                             \
            set_error("%s: workspace layout needs %zu bytes, the caller's workspace holds %zu", who, (ws).used(), (ws).cap); \
            return VPX_ERR_WORKSPACE;                                                                                 \
        }                                                                                                             \
    } while (0)

int build_stages(ConvStage* st, int* chunks_total, const int* segC, int nseg, int taps, int cs, int prec, int qpc = 2);
int pick_stage_channels(const int* segC, int nseg, int kh, int kw, int ng, int prec, int mw = 1, int stride = 1, int qpc = 2);
int conv_a_bytes(const ConvStage* st, int nstage, int kh, int kw, int mw = 1, int stride = 1);
bool conv_fits_lds(const int* segC, int nseg, int kh, int kw, int ng, int prec, int mw = 1, int stride = 1, int qpc = 2);
// rows per wave: 2 (16x16 workgroup tile) when the operand mode profits (bf16x3 is LDS/issue-bound, not MFMA-bound) and
// the launch still has >= 2 workgroups per CU; else 1
int pick_mw(int B, int H, int W, int n_tiles, int prec);
// same rule from the number of 8x16 pixel tiles of the launch (the 8-wave form halves it)
inline int pick_mw_tiles(long long m_tiles, int n_tiles, int prec) { return (prec != 0 /* VPX_PREC_F32 */ && m_tiles / 2 * n_tiles >= 512) ? 2 : 1; }
inline void set_plan_tiles(ConvPlan& P, int mw) { P.mw = mw; P.tiles_x = (P.W + TILE_W - 1) / TILE_W; P.tiles_y = (P.H + TILE_H * mw - 1) / (TILE_H * mw); }
size_t packed_weight_bytes(int n_tiles, int chunks_total, int ng, int prec, int qpc = 2);

hipError_t launch_pack_weights(const PackDesc& pd, float* dst, hipStream_t s);

// ConvLSTM fused step (epilogue = gates + state update). Pointers NHWC.
struct ConvLSTMStepArgs {
    const float* bias;        // reference layout [4Ch] or null
    int gate_pos[4];          // position of logical gates (i,f,g,o) in the reference's 4Ch axis
    int Ch;
    const float* c_in;        // [B,H,W,Ch] or null (zeros)
    float* c_out;             // [B,H,W,Ch] (may alias c_in)
    const float* wci;         // [H,W,Ch] or null
    const float* wcf;
    const float* wco;
    float* h_out;             // h_t slab
    long long h_bstride;      // elements between batch items of h_out
    float* gates;             // [B,H,W,4Ch] post-activation (i,f,g,o) or null
};
hipError_t launch_convlstm_step_f32(const ConvPlan& plan, const ConvLSTMStepArgs& ea, int n_tiles, hipStream_t s);
// ---- second-generation fused cell (cell2.hip): bf16x3, 3x3, pre-split operands, all-DMA staging ----
struct Cell2Seg { const char* sp; long long bstride; int C; int _pad; };   // split tensor, BYTES between batch items, channels (multiple of 16)
// K stages (16 channels x 9 taps each) run x first, then h: nx / nh stages of the operands PRESENT in this launch (an
// absent operand — no input tensor, zero initial state — is skipped); stage k of h reads packed weight chunks
// 3 * (hs_off + k) .. +2, stage k of x chunks 3k .. 3k+2. No per-stage table: everything the loop needs is scalar
// arithmetic on these fields (a table walk cost an s_load + s_waitcnt lgkmcnt(0) per copy inside the MFMA loop).
struct Cell2Plan {
    int B, H, W, tiles_x, tiles_y, n_tiles, nx, nh, hs_off, chunks_total, grid_m, _p;
    int qform, _q;            // 1: the 16x16x32 main loop (cell2_kernel_q); wpk / chunks_total then describe K = 32 chunks (below)
    int plain;                // q form, half tile: VPX_PREC_BF16 (hi parts only, one MFMA per product)
    int n_groups, gpt;        // conv2 only: 32-column output groups in total / per N tile (the last tile may hold fewer)
    Cell2Seg seg[2];
    const char* wpk;          // [n_tiles][chunks_total][24576 B]
};
// 16x16x32 form ("q form"): the K = 32 steps pair two taps of a 16-channel stage over the PRESENT stage sequence s = 0 .. S-1
// (x stages, then h stages) in periods of two stages = 9 steps: even stage taps (0,1) (2,3) (4,5) (6,7), the cross step
// (tap 8 of the even stage | tap 0 of the odd stage), odd stage taps (1,2) (3,4) (5,6) (7,8); an odd S ends with (tap 8 | zero
// weights). One weight chunk = one step = [part][k group 0..3 = tap half * 2 + channel half][n][8 bf16] = 16 KiB; the pack
// therefore depends on which operands are present (an absent x or h changes the pairing).
static inline int cell2_qchunks(int S) { return (9 * S + 1) / 2; }
struct Cell2Pack {
    const float* w;           // reference OIHW [4Ch, Ct, 3, 3]
    int Ch, Ct, n_tiles, chunks_total;
    int qform, S;             // q form: S present stages, chunks_total = cell2_qchunks(S), stage_col[] lists the present stages
    int gate_pos[4];
    int stage_col[MAX_STAGE]; // column in [x | h] of the first channel of packed stage s (chunks 3s .. 3s+2)
};
hipError_t launch_split_convert(const float* src, void* dst, long long npix, int C, hipStream_t s);  // fp32 NHWC -> split format
hipError_t launch_cell2_pack(const Cell2Pack& pk, void* dst, hipStream_t s);
size_t cell2_packed_bytes(int n_tiles, int chunks_total);   // 32x32x16 form: chunks of 24 KiB (3 per stage)
size_t cell2_packed_bytes_q(int n_tiles, int S);             // q form: cell2_qchunks(S) chunks of 16 KiB
// Developer switches (VPX_* kernel-selection experiments of rounds 1-3, listed in DESIGN.md). The PRODUCT library never reads the
// process environment: dev_switch() yields the default there, so kernel selection depends on the descriptor and on
// vpx_set_option / vpx_set_deterministic only. The developer build (`make -C vp-suite_amd/csrc ablate`, -DVPX_DEV_SWITCHES; loaded
// through VPX_LIB by the scripts under tools/) reads the variable of that name.
#ifdef VPX_DEV_SWITCHES
static inline int dev_switch(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
static inline int dev_switch(const char*, int dflt) { return dflt; }
#endif
extern int g_experiment;   // vpx_api.hip: bits of kernel experiments in flight (vpx_set_option(VPX_OPT_EXPERIMENT)); 0 in the product
extern int g_mfma_shape;   // vpx_api.hip: -1 = not yet read from the environment (VPX_MFMA_SHAPE), else 0 / 1 (vpx_set_option)
int mfma_shape();
hipError_t launch_cell2(const Cell2Plan& plan, const ConvLSTMStepArgs& ea, void* h_sp, long long h_sp_bstride, hipStream_t s);
// the same step on the eight-wave half tile (cell2x.hip, round 6: 64-register wave tiles, four waves per SIMD); plan.qform launches only
// (maps in whole 16x16 tiles, whole 32-channel N tiles), tiles_y / grid_m set for 16-row tiles. VPX_OPT_EXPERIMENT bit 15 selects it
// (development state: the four-wave half tile stays the default until the A/B says otherwise)
hipError_t launch_cell2x(const Cell2Plan& plan, const ConvLSTMStepArgs& ea, void* h_sp, long long h_sp_bstride, hipStream_t s);
static inline bool cell2x_selected() { return (g_experiment & 32768) != 0; }
// c16 (conv16.hip): 3x3 stride-1 'same' (transposed or not) layers with 16 output channels on split input, weights resident in registers
bool c16_applicable(const ::vpx_conv_desc* d);
size_t c16_wpk_bytes(const ::vpx_conv_desc* d);
int c16_forward(const ::vpx_conv_desc* d, const char* x_split, long long x_bstride, long long x_tstride, int x_nT, const float* w,
                const float* bias, float* y, char* y_split, char* wpk, bool packed, hipStream_t s);
// conv2: the same kernel with a plain epilogue — 3x3 'same' convolution of one split-format source with C channels (C % 16 == 0)
// into Co fp32 NHWC output channels, [0, split) -> out0, [split, Co) -> out1. N tiling: 128-column tiles, balanced.
static inline int conv2_tiles(int Co) { return ((Co + 31) / 32 + 3) / 4; }
static inline int conv2_gpt(int Co) { const int g = (Co + 31) / 32, t = (g + 3) / 4; return (g + t - 1) / t; }
struct Conv2Pack {
    const float* w; long long s_oc, s_ic;   // element strides of the output / input channel (the tap index is contiguous)
    int Co, col0, n_tiles, gpt, chunks_total, flip;
    int qform, _p;                          // 1: K = 32 chunks (chunks_total = cell2_qchunks(C / 16))
};
struct Conv2Args {
    int B, H, W, C, Co, split, accumulate, qform;
    const char* src_sp; long long src_bstride;   // split source [B][HW][C], BYTES between batch items
    const char* wpk; const float* bias;
    float* out0; long long bstride0; int ld0, _p0;
    float* out1; long long bstride1; int ld1, _p1;
};
hipError_t launch_conv2_pack(const Conv2Pack& pk, void* dst, hipStream_t s);
hipError_t launch_conv2(const Conv2Args& c, hipStream_t s);

// ---- schedule-driven K = 32 convolution on split operands (convq.hip): EF stage glue and its adjoints ----
struct CQSeg {                 // one activation source: a (sub-)image of a split-format tensor
    const char* sp;            // tensor base
    long long bstride, tstride; int nT, _p0;   // image n = (n / nT) * bstride + (n % nT) * tstride  (BYTES; nT = 1: plain batch)
    int rowpitch, colpitch;    // bytes between consecutive tile-space rows / columns (a stride-2 sub-image: twice the tensor's)
    int org;                   // byte offset of the sub-image's pixel (0, 0) (sub-position (sy, sx) of a stride-2 source)
    int Hs, Ws;                // its extent: tile-space positions outside [0, Hs) x [0, Ws) read zeros
    int nstage, c0;            // 16-channel stages taken from it, first channel (multiple of 8)
    int _p1;
};
struct ConvQPlan {
    int B, H, W, tiles_x, tiles_y, n_tiles, grid_m;
    int n_groups, gpt;         // 32-column output groups in total / per N tile
    int S, SP, nsub, nchunk_total, pro_stage1;   // stages; stages per pass of the schedule; its entries; weight chunks; copy stage 1 in the prologue
    int oy, ox, nseg;
    int dbg, _pd;              // timing ablations (VPX_CQ_DBG, experiments): 1 no MFMAs, 2 no stage copies, 4 no weight copies, 8 no epilogue
    CQSeg seg[4];
    const char* wpk;           // [n_tile][chunk][16 KiB]
    unsigned long long sched[64];   // per step; only a chunk's FIRST entry is decoded in full (its events)
    unsigned offs[64];         // per step: (offA / 16) | on << 15 | (offB / 16) << 16 — all the inner loop reads
};
struct ConvQEpiArgs {          // bias, LeakyReLU, destinations (see ConvQEpi, convq.hip)
    const float* bias;
    float leaky;
    int Co, split, gpt, phases, accumulate;
    int oys, oxs, oyo, oxo, Hmem, Wmem;
    float* out0; long long bstride0; int ld0, _p0;
    float* out1; long long bstride1; int ld1, _p1;
    char* sp_out; long long sp_bstride;
};
// host-side description of the layer: terms = (source segment, tap, weight tap) per set of output column tiles
struct ConvQTerm { int seg, da, db, wtap; };   // wtap < 0: a zero-weight filler tap (keeps a short stage long enough to double-buffer)
struct ConvQGroupSet { int nt0, ntn, nterm; ConvQTerm term[32]; };   // 16-column tiles [nt0, nt0 + ntn) of the N tile
struct ConvQProblem {
    int N, H, W, halo;         // images, tile-space extent, 2 (taps in [-1,1]^2) or 4 ([-2,2]^2)
    int nseg; CQSeg seg[4]; int seg_wc0[4];   // + the weight tensor's input channel of each segment's first stage
    int ngs; ConvQGroupSet gs[4];
    int periodic;              // 1: one group set whose terms apply to every stage (term.seg ignored): the schedule covers 2 or 4 stages and repeats
    const float* w; long long s_oc, s_ic;     // weight element (oc, ic, tap) at w[ic * s_ic + (col0 + oc) * s_oc + tap]
    int Co, col0, phases;      // phases: group g of an N tile = output phase g of the SAME 32 channels (stride-2 transposed convolution)
};
size_t convq_wpk_bytes(const ConvQProblem& pr);
// packs the weights into wpk (unless they still are there from a previous call) and launches; VPX_OK or an error code (set_error)
int convq_run(const ConvQProblem& pr, const ConvQEpiArgs& ea, char* wpk, bool weights_packed, hipStream_t s);

// ---- small-grid fused cell (cell3.hip): 16x16-pixel tiles x 8-channel slices, the slice's recurrent weights resident in LDS,
//      the input projection hoisted (enters through `pre`) ----
struct Cell3Pack { const float* w; int Cin, Ch, Ct, n_slices, nk; int gate_pos[4]; };
struct Cell3Args {
    int B, H, W, tiles_x, tiles_y, n_slices, nk, _p;
    const char* h_sp; long long h_bstride;       // h_{t-1}, split format [B][HW][Ch], BYTES between batch items; null = zero state
    const char* wpk;                              // cell3_pack_kernel output
    const float* pre; long long pre_bstride;      // W_x * x_t of this step [b][HW][4Ch] (reference gate order), ELEMENTS between batch items; or null
    char* h_sp_out; long long h_sp_out_bstride;   // split h_t for the next step (or null), BYTES
    ConvLSTMStepArgs ea;
};
extern int g_cell3_mode;   // vpx_api.hip: -1 = not yet read from the environment (VPX_CELL3), else 0 / 1 (vpx_set_option)
int cell3_mode();
bool cell3_applicable(const vpx_convlstm_desc* d);
size_t cell3_packed_bytes(int Ch);
hipError_t launch_cell3_pack(const Cell3Pack& pk, void* dst, hipStream_t s);
hipError_t launch_cell3(const Cell3Args& args, hipStream_t s);

// pointwise half of the K-split step: pre-activations pre [B*HW, 4Ch] (reference gate order) -> gates, c, h (pointwise.hip)
hipError_t launch_convlstm_pointwise(const ConvLSTMStepArgs& ea, const float* pre, int B, long long HW, hipStream_t s,
                                     long long pre_bstride = 0);  // elements between batch items of pre (0 = dense)

// Plain epilogue: y = conv (+bias), channels [0,split) -> out0, [split, Co) -> out1 (either may be null = dropped).
struct PlainEpiArgs {
    const float* bias;        // [Co] or null
    int Co, split;
    float* out0; long long bstride0; int ld0;   // NHWC, ld = channels per pixel of the destination tensor
    float* out1; long long bstride1; int ld1;
    int accumulate;           // 1: += into destination
    int ksplit;               // copy of ConvPlan::ksplit (> 1: atomic accumulation, bias from split 0, no activation)
    int ng;                   // 32-channel groups per N tile (plain_groups(Co)); tile covers ng*32 output channels
    float leaky;              // LeakyReLU negative slope applied after the bias (0 = none; 1 would be identity)
    int omap;                 // 1: tile-space pixel (y,x) is stored at (y*oys + oyo, x*oxs + oxo) of a Wmem-wide image
    int oys, oyo, oxs, oxo, Wmem;
    char* sp_out;             // optional: the output AGAIN in split-bf16 operand format (cell2.hip; [pixel][Co/8][hi 8 | lo 8]) —
    long long sp_bstride;     //   BYTES between batch items; out0 may then be null (inference: the fp32 copy is never read)
};
hipError_t launch_conv_plain_f32(const ConvPlan& plan, const PlainEpiArgs& ea, int n_tiles, hipStream_t s);
// N tiling of a plain convolution with Co outputs: ng = 1..4 groups of 32 channels per workgroup, chosen to minimise
// channel padding (ties -> wider tiles: fewer re-reads of the activation tile).
int plain_groups(int Co, long long m_tiles = -1);
// K split of a plain convolution whose grid would have only `wgs` workgroups: number of stage ranges (1 = no split)
int pick_ksplit(long long wgs, int nstage, bool bwd = false);  // bwd: a data-gradient conv (see the rule in conv_gemm.hip)
extern int g_deterministic;  // vpx_set_deterministic(): 1 = never split K (no floating-point atomics)
inline int plain_tiles_ng(int Co, int ng) { return (Co + 32 * ng - 1) / (32 * ng); }
inline int plain_tiles(int Co) { return plain_tiles_ng(Co, plain_groups(Co)); }
// upper bound of the packed-weight rows over every tiling plain_groups() can return: Co rounded up to 128
inline int plain_rows_bound(int Co) { return (Co + 127) / 128 * 128; }
// fills NG / rowbase / goff / tile_stride / nch / n_tiles of a pack descriptor for a plain conv whose outputs start at
// row (or column, if transposed) `first` of every segment's weight tensor; ng <= 0: plain_groups(Co)
void fill_plain_pack(PackDesc& pd, int Co, int first, int ng = 0);

// ---- ST-LSTM (predrnn.py:57-83) epilogues; all tensors NHWC [B,HW,Ch] ----
struct STGateArgs {           // "c group": acc = (i, f, g, o_pre) from [x | h];  "m group": acc = (i', f', g') from [x | m]
    int Ch;
    float forget_bias;        // 1.0 (predrnn.py:23)
    const float* s_in;        // c_t (c group) or m_t (m group)
    float* s_new;             // c_new / m_new
    float* delta;             // delta_c / delta_m
    float* o_pre;             // c group only: o_x + o_h (pre-activation partial), else null
    float* gates;             // [B,HW,3Ch] post-activation (i,f,g) for the backward, or null
};
struct STOutArgs {            // h_new = sigmoid(o_pre + conv_o(mem)) * tanh(conv_last(mem))
    int Ch;
    const float* o_pre;
    const float* lc;          // conv_last(mem)
    float* h_new;
    float* o_save;            // sigmoid output, or null
    float* tl_save;           // tanh(conv_last) or null
};
hipError_t launch_st_cgroup_f32(const ConvPlan& plan, const STGateArgs& ea, int n_tiles, hipStream_t s);
hipError_t launch_st_mgroup_f32(const ConvPlan& plan, const STGateArgs& ea, int n_tiles, hipStream_t s);
hipError_t launch_st_out_f32(const ConvPlan& plan, const STOutArgs& ea, int n_tiles, hipStream_t s);
// c group and m group of one ST-LSTM step in a single launch (same pixel tiling, n_tiles channel tiles each)
hipError_t launch_st_gates_dual(const ConvPlan& pc, const STGateArgs& ec, const ConvPlan& pm, const STGateArgs& em,
                                int n_tiles, hipStream_t s);

// ---- BPTT pieces (lstm_bwd.hip) ----
struct GateBwdArgs {
    int B, HW, Ch;
    int gate_pos[4];          // where logical (i,f,g,o) live in the reference's 4Ch axis: dG is written in THAT order
    const float* gates;       // [B,HW,4Ch] saved post-activation (i,f,g,o) of this step
    const float* c_t;         // [B,HW,Ch] cell state after this step
    const float* c_prev;      // cell state before this step, or null (zeros)
    const float* dh_in;       // recurrent dh flowing in from step t+1 (or dhT), or null
    const float* dout;        // dL/d out[:, t] slab or null
    long long dout_bstride;
    const float* dc_in;       // dc flowing in (or dcT), or null
    float* dc_out;            // dc flowing out to step t-1
    const float* wci; const float* wcf; const float* wco;  // peepholes [HW,Ch] or null
    float* dwci; float* dwcf; float* dwco;                  // accumulated (+=) over steps, or null
    long long peep_slice_stride;                            // > 0: batch slice y accumulates into dwc*[y * stride + ..] (per-slice partials,
                                                            //      summed by launch_peep_reduce after the time loop); 0: one slice, in place
    float* dG;                // [B,HW,4Ch] d(pre-activation), reference gate order (or null when only dG_sp is wanted)
    char* dG_sp;              // the same tensor in split-bf16 operand format (cell2.hip), or null; needs an even Ch
    float* db_partial;        // [gridDim.x][4Ch] per-block column sums of dG (bias gradient partials), or null
};
inline int gate_bwd_blocks(int HW, int Ch) { return (HW * Ch + 255) / 256; }
// batch slices (grid.y): the kernel streams ~56 B per element with one element per thread and batch step, so it wants >= 8
// waves per SIMD in flight (2048 blocks); a (pixel, channel) thread then walks B / slices batch items. Peephole gradients (a sum
// over the batch with one owner per element) keep one slice unless the caller provides per-slice partial buffers (peep_slice_stride).
constexpr int GATE_BWD_MAX_SLICES = 8;
inline int gate_bwd_slices(int HW, int Ch, int B, bool peephole_grads) {
    if (peephole_grads) return 1;
    int s = (2048 + gate_bwd_blocks(HW, Ch) - 1) / gate_bwd_blocks(HW, Ch);
    if (s > GATE_BWD_MAX_SLICES) s = GATE_BWD_MAX_SLICES;
    if (s > B) s = B;
    return s < 1 ? 1 : s;
}
hipError_t launch_gate_bwd(const GateBwdArgs& a, hipStream_t s);
// dw[i] = sum over slices of part[slice * n + i], fixed order; the three peephole tensors in one launch (part / dw: 3 pointers each)
hipError_t launch_peep_reduce(const float* p0, const float* p1, const float* p2, float* d0, float* d1, float* d2, int slices, long long n,
                              hipStream_t s);
// out[c] = sum_r m[r][c] (* LeakyReLU'(y[r][c]) when y != null, the scaled matrix optionally stored), bit-reproducible;
// partial_ws: COLSUM_BLOCKS * cols floats of scratch
constexpr int COLSUM_BLOCKS = 4096;   // level-1 blocks: enough 256-thread blocks in flight to stream a GB-sized dy at HBM speed
hipError_t launch_colsum(const float* m, const float* y, float slope, float* scaled, float* out, float* partial_ws,
                         long long rows, int cols, hipStream_t s, char* scaled_sp = nullptr);   // scaled_sp: the scaled matrix again in the split format (cols % 8 == 0)

// A workgroup's 64 activation columns are two 32-channel halves (one per wave column), each its own slice
// [c0, c0+cn) of segment seg (0 = x, 1 = h); cglobal = column in [x|h]. Halves are paired in order ACROSS the
// segments, so 96 + 96 channels make three full tiles instead of four with two half-empty ones. cn = 0: unused half.
constexpr int WG_MAX_CTILES = 24;   // 64-channel column tiles of a weight gradient: up to 1536 activation channels (TrajGRU's ret: L*C = 1248)
struct WgradCHalf { int seg, c0, cn, cglobal; };
struct WgradCTile { WgradCHalf h[2]; };
// tiles for C0 channels of segment 0 (0 = no such operand) and C1 of segment 1 whose first column is cg1; -1: > cap tiles
static inline int wgrad_make_ctiles(WgradCTile* ct, int cap, int C0, int C1, int cg1) {
    int nh = 0;
    for (int seg = 0; seg < 2; ++seg) {
        const int C = seg ? C1 : C0, cg = seg ? cg1 : 0;
        for (int c0 = 0; c0 < C; c0 += 32, ++nh) {
            if (nh >= 2 * cap) return -1;
            ct[nh >> 1].h[nh & 1] = WgradCHalf{seg, c0, (C - c0 < 32) ? C - c0 : 32, cg + c0};
        }
    }
    if (nh & 1) ct[nh >> 1].h[1] = WgradCHalf{0, 0, 0, 0};
    return (nh + 1) >> 1;
}
struct WgradArgs {
    int T, B, H, W, HW, kh, kw, tiles_x, tiles_y;
    int N4, Cin, Ch, Ct;      // contraction-side rows (e.g. 4Ch gate rows), segment channel counts, Ct = Cin + Ch
    int ldG;                  // elements between consecutive pixels of dG (0 = N4)
    int blk;                  // row-block size for the output-row map (0 = identity map)
    int rowblk[8];            // output row of dG channel n = rowblk[n / blk] * blk + n % blk
    int n_out;                // rows of the weight-gradient tensor (0 = N4)
    int prec;                 // VPX_PREC_F32 (exact) | VPX_PREC_BF16X3 (split-bf16 operands via transposing LDS reads)
    const float* dG;          // [T][B,HW,ldG]
    const float* x; long long x_bstride, x_tstride;        // x[b,t] slabs (null if no input)
    const float* hseq; long long h_bstride, h_tstride;     // forward outputs: h_{t-1} = hseq[b, t-1]
    const float* h0;          // [B,HW,Ch] or null
    int n_ctiles;
    WgradCTile ct[WG_MAX_CTILES];
    float* slabs;             // [n_slices][taps][N4][Ct]
    // generalisation for strided / transposed convolutions (conv_api.hip); all 0 = the stride-1 "same" case above
    int a_sub;                // 1: the activation operand is the sub-image (gy*a_sy + a_oy, gx*a_sx + a_ox) of a larger image
    int a_sy, a_sx, a_oy, a_ox;
    int a_Hs, a_Ws;           // valid rows / columns of the sub-image
    int a_Wfull;              // pixels per row of the full image (the batch stride is x_bstride)
    int use_org, org_y, org_x;  // tap (0,0) reads activation pixel (y + org_y, x + org_x) instead of (y - kh/2, x - kw/2)
    int dbg;                    // timing ablations, -DVPX_ABLATE builds only (VPX_WG_DBG: 1 = no multiply, 2 = stage the first item only)
    // activation operand pre-split (ConvLSTM block after a cell2 forward): x, h_{t-1} and h_0 again in the split-bf16 operand
    // format (cell2.hip); the tap-group kernel then stages the halo tile by LDS-DMA instead of load + split + store
    int a_split;
    const char* x_sp; long long x_sp_bstride, x_sp_tstride;    // BYTES
    const char* h_sp; long long h_sp_bstride, h_sp_tstride;    // h_t of step t at h_sp + t * tstride (time-major slots), BYTES
    const char* h0_sp;                                         // [B][HW][Ch] or null
    const char* g_sp;           // dG again in split format [T][B][HW][N4] (gate-backward kernel), or null: with a_split, selects wgrad2.hip
    int vec_all;                // set by launch_wgrad: every operand allows 16-byte vector loads (bf16 forms: unconditional load issue)
    int grid_x, grid_slices;    // set by launch_wgrad: logical grid (row tile x column tile, K slice) behind the XCD-aware 1-D launch
    int w2_nh, w2_ns_half;      // set by launch_wgrad2: row tiles of a half-empty last column tile (0 = none) and their slice count
};
hipError_t launch_wgrad(const WgradArgs& a, int n_slices, hipStream_t s);
hipError_t launch_wgrad_reduce(const float* slabs, float* dW, int n_slices, int taps, int N4, int Ct, hipStream_t s);
// conv_small.hip: the same layers' forward (and the 1x1 adjoint) as streaming kernels; kind from conv_small_kind (0 = not one)
int conv_small_kind(const vpx_conv_desc* d);
hipError_t launch_conv_small(const vpx_conv_desc* d, int kind, const float* x, const float* w, const float* bias, float* y, char* y_sp,
                             hipStream_t s);
// few-channel stride-1 layers of the EF glue (1|3 -> Co 3x3 'same', 16 -> 1|3 1x1): direct fp32 sums, no MFMA (lstm_bwd.hip)
constexpr int WGRAD_SMALL_BLOCKS = 2048;
bool wgrad_small_applicable(int Co, int C, int kh, int kw, int stride, int pad);
hipError_t launch_wgrad_small(const float* dy, const float* x, int N, int H, int W, int Co, int C, int k, int pad, float* partial_ws,
                              float* dw, hipStream_t s);
// wgrad2.hip: both operands pre-split (3x3, bf16x3); writes slabs [used_slices][9][N4][Ct] like launch_wgrad
bool wgrad2_applicable(const WgradArgs& a);
int wgrad2_target_wgs();   // workgroups a wgrad2 launch aims at (512 = two rounds of one-per-CU workgroups; VPX_WGRAD2_WGS overrides)
// K slices of a wgrad2 launch over `rows128` row tiles x n_ctiles column tiles (the last one half-empty or not), before the caps
// by the caller's slab space and the item count. One rule for the launch (launch_wgrad2) and the slab sizing (convlstm_layout).
static inline int wgrad2_slices(int target, int rows128, int n_ctiles, bool half_tail) {
    const int nh = half_tail ? rows128 : 0, nf = rows128 * n_ctiles - nh;
    if (nf <= 0) return target / (rows128 > 0 ? rows128 : 1);
    return half_tail ? (8 * target) / (8 * nf + 5 * nh) : target / nf;
}
// columns >= *tail_col0 (the half-empty last column tile, if any) were written to the first *tail_slices slabs only
hipError_t launch_wgrad2(const WgradArgs& a, int max_slices, int* used_slices, int* tail_col0, int* tail_slices, hipStream_t s);
hipError_t launch_wgrad_reduce_tail(const float* slabs, float* dW, int n_slices, int taps, int N4, int Ct, int tail_col0, int tail_slices,
                                    hipStream_t s);
// wgrad2.hip, glue form: one stride residue of a stage-glue layer's weight gradient with BOTH operands in the split format (g_sp: the
// contraction-side rows, x_sp: the activation image, a_sub / use_org addressing); slabs [*used_slices][kh * kw][N4][Ct]
bool wgrad2g_applicable(const WgradArgs& a);
hipError_t launch_wgrad2g(const WgradArgs& a, int max_slices, int* used_slices, hipStream_t s);
// st_pointwise.hip: the pointwise stages behind K-split c5 launches (small grids)
struct STGatesKSArgs {
    long long npix; int Ch, ks; long long pstride; float fbias;
    const float* part;                       // ks buffers [B,HW,7Ch] of pre-activation partial sums: (i,f,g,o | i',f',g')
    const float* c; const float* m;
    float *c_new, *m_new, *delta_c, *delta_m, *o_pre, *gates_c, *gates_m;   // gates_*: [B,HW,3Ch] or null
    char *cn_sp, *mn_sp;                     // c_new / m_new once more in the split format, or null
};
struct STSplitShadows { const char* in[5]; char* out[3]; char* dg8; int set; };   // vpx_stlstm_shadows of the running call (stlstm_api.hip)
STSplitShadows st_shadows_of(const vpx_stlstm_shadows* p);
struct STOutKSArgs {
    char* h_sp; int Ch;                      // h_new once more in the split format, or null
    long long n; int ks; long long pstride;  // n = B*HW*Ch
    const float* part;                       // ks buffers [B,HW,Ch]: partial sums of conv_o(mem)
    const float* o_pre; const float* lc;
    float *h_new, *o_save, *tl_save;
};
hipError_t launch_st_gates_ks(const STGatesKSArgs& a, hipStream_t s);
hipError_t launch_st_out_ks(const STOutKSArgs& a, hipStream_t s);
hipError_t launch_sum_partials(float* out, const float* part, long long pstride, int ks, long long n, int accumulate, hipStream_t s);
// up to five such sums in one launch; job i: out[i][e] (+)= sum_k part[i][k * n[i] + e]  (n[i] % 4 == 0)
hipError_t launch_sum_partials_multi(int njobs, float* const* out, const float* const* part, const long long* n, const int* ks, const int* acc, hipStream_t s);
// conv1.hip, c1: 1x1 convolution as a streaming kernel (weights resident in registers), fp32 NHWC in and out, bf16x3 arithmetic
struct C1Args {
    const float* x[2]; int xld[2], xc[2];   // up to two sources concatenated along K: pixel pitch (floats), channels (multiples of 32; xc[1] = 0: one source)
    long long npix;
    const float* w; long long w_sn, w_sc;   // weight element (column n, channel c) at w[n * w_sn + c * w_sc]
    float* y[2]; int yld[2], ysplit;        // columns [0, ysplit) go to y[0], the rest to y[1] (pixel pitches yld)
    int Co, accumulate;
    int x_split, _pad;                      // 1: the sources are in the split-bf16 operand format (same pixel pitch in bytes: xld * 4) — read as MFMA fragments, no conversion
};
bool c1_applicable(const C1Args& a, int prec);   // (Co, K) in {(128, 256), (256, 128), (128, 128)}, aligned operands, bf16x3
hipError_t launch_c1(const C1Args& a, hipStream_t s);
// convq.hip, c5: 5x5 'same' convolutions on 16x16-pixel tiles over ONE split-format source, a table of jobs per launch
constexpr int C5_MAX_JOBS = 12;
struct C5Src { const char* p; long long bstride; int prow, _pad; };   // split-format tensor [B][H][W][C]: bytes per image / per pixel
struct C5Job {
    int nrange, r_src[3], r_c0[3], r_n[3];   // the job's K: channel ranges [r_c0, r_c0 + r_n) of source r_src, multiples of 8, in stage order
    int S8, Q, n_tiles, nt_active;           // derived (c5_prepare_job): 8-channel stages, K = 32 steps, N tiles, column tiles in use
    int epi;                                 // 0: fp32 destination (optional +=); 1: ST-LSTM gate group; 2: ST-LSTM output gate
    int Co, ld, accumulate;                  // epi 0: output channels, destination pixel pitch (floats), += instead of =
    const char* wpk;                         // packed weights [n_tile][Q][chunk]
    float* out; long long out_bstride;       // epi 0
    int Ch, ng; float fbias;                 // epi 1 / 2: channels of the state tensors [B,HW,Ch]; epi 1: gate groups (4: i,f,g,o_pre; 3: i',f',g'), forget bias
    const float* e_in0; const float* e_in1;  // epi 1: s_in (c or m); epi 2: o_pre, conv_last(mem)
    float* e_out[4];                         // epi 1: s_new, delta, o_pre (ng = 4), saved gates [B,HW,3Ch] or null; epi 2: h_new, o_save, tl_save (or null)
    char* e_sp;                              // epi 1 / 2 / 3: s_new / h_new once more in the split format, or null
    // epi 3 (3x3 jobs): the ConvLSTM step (conv_lstm_hzzone.py:59-68 / conv_lstm_ndrplz.py:31-41) on a gate-interleaved N tile of NT * 4 channels:
    // e_in0 = c_in (or null: zeros), e_in1..3 = peepholes Wci, Wcf, Wco [H,W,Ch] (or null together), e_out[0] = h_out (or null), e_out[1] = c_out
    const float* e_in2; const float* e_in3; const float* bias;   // bias: reference layout [4Ch] or null
    int gate_pos[4];                         // position of the logical gates (i,f,g,o) in the reference's 4Ch axis
    long long h_bstride, sp_bstride;         // elements between batch items of h_out; BYTES between batch items of e_sp
    // derived (launch_c5): the ranges resolved against C5Plan::src — first byte of range r in sample 0 (or null), bytes per sample / per pixel.
    // (The kernel reads these, not src[r_src[r]]: a look-up of the source table inside its K loop was a scalar load with the step's reads in flight.)
    const char* r_p[3]; long long r_bs[3]; int r_prow[3], _rpad;
};
struct C5Plan {
    int B, H, W, tiles_x, tiles_y, m_tiles;
    C5Src src[4];
    int ks;                                  // kernel size of every job: 0 | 5 = 5x5 (ST-LSTM step), 3 = 3x3 (ConvLSTM step on small grids)
    unsigned long long* stamps; int stamp_block;   // developer timing stamps (null in the product; vpx_dbg_c5_stamps)
    int ablate, order;                       // ablate: developer build only (VPX_C5_ABLATE). order: 0 = the N tiles of a pixel tile adjacent in an XCD's dispatch order, 1 = the pixel tiles of an N tile adjacent
    int njobs; C5Job job[C5_MAX_JOBS];
};
static_assert(sizeof(C5Plan) <= 4096, "C5Plan travels as a kernel argument (4 KiB)");
struct C5PackRange { const float* w; long long s_oc, s_c; int c0; int gate0[4]; };   // weights of one K range (see c5_pack_kernel)
size_t c5_wpk_bytes(int K, int Co, int NT, int gates = 0, int ks = 5);
int c5_prepare_job(C5Job& j, int NT, const C5PackRange* rg, int gates, int flip, bool packed, hipStream_t s, int gate_major = 0, int ks = 5);
// c5_prepare_job only QUEUES a job's weight pack: launch_c5 packs everything queued in one launch before it starts; a new library call
// (Carver construction) drops what an earlier, failed call may have left queued
hipError_t c5_flush_packs(hipStream_t s);
void c5_drop_pending_packs();
void c5_chunk_job(const C5Job& full, const C5PackRange* prf, int k, int ks, C5Job& j, C5PackRange* pr);   // K-split: chunk k of ks
size_t c5_chunk_wpk_bytes(int K, int k, int ks, int cols, int NT);
hipError_t launch_c5(const C5Plan& P, int NT, hipStream_t s);   // NT = 8: 128-column N tiles, 4: 64-column
// wgrad2.hip, stw: the four 5x5 weight gradients of one ST-LSTM cell step in one launch (operands in split format)
constexpr int STW_MAX_PAIRS = 48;
struct STWSrc { const char* sp; int C; };          // split tensor [B][HW][C]
struct STWHalf { int src, c0, cn, cglobal; };      // channels [c0, c0 + cn) of source src = columns cglobal.. of the tensor (cn = 0: empty)
struct STWPair { int n0, tensor; STWHalf h[2]; };  // rows n0 .. n0 + 127 of dG7 x 64 columns
struct STWArgs {
    int B, H, W, HW, Ch, N7;     // N7: channels per pixel of the row operand (8Ch)
    const char* g_sp;            // dG8 [B][HW][8Ch], split format: gate blocks (i,f,g | o | i',f',g') + d conv_last
    STWSrc src[5];               // x, h, m, c_new, m_new
    int npairs, npairs5, n_slices;   // pairs [0, npairs5): k x k tensors (three tap passes); [npairs5, npairs): the 1 x 1 tensor (centre tap)
    STWPair pair[STW_MAX_PAIRS];
    float* slabs;                // [n_slices][npairs][25][128][64]
    size_t slab_stride;          // floats per slice
    unsigned long long* stamps; int stamp_block, _spad;   // developer timing stamps (null in the product; vpx_dbg_stw_stamps)
};
struct STWOut { float* dW[5]; int Ct[5]; int ntaps[5]; signed char blockmap[5][8]; };   // Wx, Wh, Wm, Wo, Wlast
int stw_build(STWArgs& a, STWOut& o, int B, int H, int W, int Cin, int Ch);   // fills the pair table; returns npairs (-1: too many)
int stw_slices(int npairs, long long items);
hipError_t launch_stw(const STWArgs& a, const STWOut& o, hipStream_t s);       // kernel + slice reduction into dW[0..3] (overwritten)
// same, but launch tap t lands at tap index tapmap[t] of a dW with real_taps taps per (row, channel); tapmap[t] < 0: dropped
hipError_t launch_wgrad_reduce_map(const float* slabs, float* dW, int n_slices, int taps, int N4, int Ct, int real_taps,
                                   const int* tapmap, hipStream_t s);

// ---- ST-LSTM backward pointwise stages ----
struct STBwdOutArgs {         // stage A: through h_new = o * tanh(lc)
    long long n;              // B*HW*Ch
    int Ch, ldG, o_off;       // d(o pre-activation) is written into dG7[pix*ldG + o_off + ch]
    int dlc_off;              // >= 0: d conv_last goes to dG7[pix*ldG + dlc_off + ch] in dG7's own format instead of the fp32 tensor dlc
    const float* dh_new; const float* o; const float* tl;
    float* dG7; float* dlc;
    int split;                // dG7 is written in the split operand format (Ch % 8 == 0, ldG % 8 == 0); dlc stays fp32
};
struct STBwdGateArgs {        // stage B: through the two gate groups
    long long npix;           // B*HW
    int Ch, ldG;
    const float* gates_c; const float* gates_m;    // [B,HW,3Ch] (i,f,g)
    const float* c; const float* m;                // cell inputs
    const float* dcn_ext; const float* dmn_ext;    // incoming grads of c_new / m_new (may be null)
    const float* ddc_ext; const float* ddm_ext;    // incoming grads of delta_c / delta_m (may be null)
    const float* dcn_conv; const float* dmn_conv;  // grads of c_new / m_new through conv_o / conv_last
    float* dG7;               // blocks (i,f,g | o | i',f',g'), o block already filled by stage A
    float* dc;                // out: dL/dc (may be null)
    float* dm;                // out: direct part of dL/dm = dm_new_total * f' (conv part is accumulated later)
    int split;                // dG7 blocks written in the split operand format
};
// ---- LayerNorm ST-LSTM variant (layernorm.hip) ----
hipError_t launch_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* xhat,
                                float* stats, double* partial, int B, long long n, hipStream_t s);
hipError_t launch_layernorm_bwd(const float* dy, int ldy, int Cb, const int* blk, const float* xhat, const float* stats,
                                const float* gamma, int B, int HW, int C, double* partial, float* sums, float* du,
                                float* dgamma, float* dbeta, hipStream_t s);
struct STLNGateArgs {
    long long npix; int Ch;
    const float* xc; const float* hc; const float* mc;   // normalised conv outputs [B,HW,7Ch] / [..,4Ch] / [..,3Ch]
    const float* c; const float* m;
    float* c_new; float* m_new; float* delta_c; float* delta_m; float* o_pre;
    float* mem;                 // [B,HW,2Ch] = cat(c_new, m_new)
    float* gates_c; float* gates_m;   // [B,HW,3Ch] each or null
};
hipError_t launch_st_ln_gates(const STLNGateArgs& a, hipStream_t s);
hipError_t launch_st_ln_out(const float* o_pre, const float* oc, const float* lc, float* h_new, float* o_save,
                            float* tl_save, long long n, hipStream_t s);
hipError_t launch_st_bwd_out(const STBwdOutArgs& a, hipStream_t s);
hipError_t launch_st_bwd_gates(const STBwdGateArgs& a, hipStream_t s);

// ---- decoupling-loss tail (predrnn_v2.py:197-198, 209-211) pointwise/reduction stages ----
// stats[b, co] = (s_cc, s_mm, s_cm, |cos|) over the H*W axis of Yc = A*delta_c, Ym = A*delta_m (NHWC [B,HW,Ch])
hipError_t launch_decouple_stats(const float* yc, const float* ym, float* stats, int B, int HW, int Ch, hipStream_t s);
hipError_t launch_decouple_mean(const float* stats, float* value, int n, hipStream_t s);
hipError_t launch_decouple_bwd_pointwise(const float* yc, const float* ym, const float* stats, const float* dvalue,
                                         float* dyc, float* dym, int B, int HW, int Ch, hipStream_t s);
hipError_t launch_axpy(float* y, const float* x, long long n, hipStream_t s);  // y += x

hipError_t launch_nchw_to_nhwc(const float* src, float* dst, int N, int C, int H, int W, hipStream_t s);
hipError_t launch_nhwc_to_nchw(const float* src, float* dst, int N, int C, int H, int W, hipStream_t s);

#ifdef __HIPCC__
// Gate nonlinearities on the hardware exp2/rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each). |abs error| < 3e-7 for both,
// far inside the 1e-4 parity budget; the libm expf/tanhf they replace cost ~20 VALU each and made the bf16x3 kernel
// VALU-bound (9.4 VALU per MFMA measured).  VPX_ACCURATE_MATH=1 at build time restores libm.
#ifdef VPX_ACCURATE_MATH
__device__ __forceinline__ float sigmoid_f(float v) { return 1.0f / (1.0f + expf(-v)); }
__device__ __forceinline__ float tanh_f(float v) { return tanhf(v); }
#else
__device__ __forceinline__ float sigmoid_f(float v) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * v));
}
__device__ __forceinline__ float tanh_f(float v) {
    // tanh(v) = 1 - 2 / (1 + e^{2v}); saturates cleanly for large |v| (exp2 -> inf / 0)
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * v));
}
#endif
// c_t = f * c_{t-1} + i * g (conv_lstm_hzzone.py:66) with ITS contraction spelled out: one rounded product i * g, then ONE fused multiply-add.
// Left to the compiler, `f * c + i * g` becomes fma(f, c, i * g) in one kernel and fma(i, g, f * c) in another (its SLP / contraction choice
// depends on the surrounding code): kernel forms that sum the same products in the same order then still differed in the last bit of c_t
// (round 6: cell2_kernel_x against cell2_kernel_q, 13 % of the elements of one shape by one ulp). Every fused ConvLSTM epilogue calls this.
__device__ __forceinline__ float lstm_c(float f, float c_prev, float i, float g) { return __builtin_fmaf(f, c_prev, i * g); }

#endif

}  // namespace vpx
