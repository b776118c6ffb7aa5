/*
 * vpx.h — C ABI of libvpx_hip.so: the MI355X (gfx950) implementation of vp-suite's spatiotemporal recurrent hot path.
 *
 * The reference (AIS-Bonn/vp-suite) has NO native/FFI layer: its hot path is a sequence of ATen calls issued from
 * Python. Each entry point below therefore replaces one Python-level callable of the reference; the
 * binding a maintainer adds is a ctypes stub. All pointers are DEVICE pointers (hipMalloc'd / torch
 * CUDA tensors), fp32 unless stated. The library never allocates, never synchronises and enqueues everything on the
 * `stream` argument (a hipStream_t passed as void*; NULL = default stream). The caller owns every buffer including
 * workspace and reserve. Return value: 0 = OK, negative = error (message via vpx_last_error(), thread-local).
 *
 * Entry point                      replaces (reference file:line)
 * -------------------------------  -----------------------------------------------------------------------------
 * vpx_convlstm_seq_fwd / _bwd      model_blocks.ConvLSTM.forward            vp_suite/model_blocks/conv_lstm_hzzone.py:38-70
 *                                  ConvLSTMCell.forward (T=1, IFOG)         vp_suite/model_blocks/conv_lstm_ndrplz.py:28-43
 *                                  ConvLSTM(ndrplz) per-layer time loop     vp_suite/model_blocks/conv_lstm_ndrplz.py:112-121
 * vpx_stlstm_step_fwd / _bwd       SpatioTemporalLSTMCell.forward           vp_suite/model_blocks/predrnn.py:57-83
 * vpx_acstlstm_step_fwd / _bwd     ActionConditionalSpatioTemporalLSTMCell.forward  vp_suite/model_blocks/predrnn.py:139-169
 * vpx_trajgru_seq_fwd / _bwd       TrajGRU.forward (time loop, warps, gates)        vp_suite/model_blocks/traj_gru.py:164-214
 * vpx_decouple_fwd / _bwd          adapter + normalize + |cos| + mean       vp_suite/models/predrnn_v2.py:197-198,209-211
 * vpx_conv2d_nhwc_fwd / _bwd       F.conv2d 1x1 / kxk "same", stride 1      vp_suite/models/predrnn_v2.py:223 (conv_last)
 * vpx_conv2d_ex_fwd / _bwd         Conv2d / ConvTranspose2d + LeakyReLU     vp_suite/models/precipitation_nowcasting/ef_blocks.py:15-49
 * vpx_mse_loss                     MSE measure + loss provider              vp_suite/base/base_measure.py:55-57, measure/loss_provider.py:48-51
 * vpx_adam_step                    torch.optim.Adam(model.parameters(), lr)  vp_suite/vpsuite.py:353, base/base_model.py:174-176
 * vpx_nchw_to_nhwc / nhwc_to_nchw  (layout adaptors at the boundary; the reference is NCHW throughout)
 *
 * Layouts. VPX_LAYOUT_NHWC ("channels last", the library's native layout):
 *      x [B,T,H,W,Cin]   h/c states [B,H,W,Ch]   out [B,T,H,W,Ch]   peepholes [H,W,Ch]
 *    VPX_LAYOUT_NCHW (the reference's layout; the library transposes through the workspace):
 *      x [B,T,Cin,H,W]   h/c states [B,Ch,H,W]   out [B,T,Ch,H,W]   peepholes [1,Ch,H,W]
 *    Convolution weights / biases and their gradients are ALWAYS in the reference's parameter layout
 *    (OIHW, e.g. _conv.weight [4Ch, Cin+Ch, kh, kw]) so reference checkpoints are used unchanged.
 */
#ifndef VPX_H_
#define VPX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VPX_VERSION 100

enum { VPX_OK = 0, VPX_ERR_ARG = -1, VPX_ERR_WORKSPACE = -2, VPX_ERR_LAUNCH = -3, VPX_ERR_UNSUPPORTED = -4 };

enum { VPX_GATE_IFGO = 0 /* chunk order (i,f,g,o): conv_lstm_hzzone.py:62 */,
       VPX_GATE_IFOG = 1 /* split order (i,f,o,g): conv_lstm_ndrplz.py:34 */ };
enum { VPX_LAYOUT_NHWC = 0, VPX_LAYOUT_NCHW = 1 };
enum { VPX_PREC_F32 = 0    /* exact fp32: v_mfma_f32_32x32x2_f32, fp32 operands + fp32 accumulate */,
       VPX_PREC_BF16X3 = 1 /* split bf16 (hi/lo) operands, 3 bf16 MFMAs per product, fp32 accumulate (~fp32 accuracy) */,
       VPX_PREC_BF16 = 2   /* bf16 operands, fp32 accumulate, fp32 state and I/O */ };
enum { VPX_FLAG_OUT_SPLIT = 8 /* ConvLSTM forward (inference): `out` receives the output sequence in the split-bf16 operand format
                                 ([B][T][H*W][Ch] split-encoded, the byte size of the fp32 tensor it replaces) and NO fp32 copy is written
                                 — for a consumer that reads the operand format (vpx_conv2d_ex_fwd_from_split); hT, if not NULL, is
                                 still fp32. Only where vpx_convlstm_writes_split_output() says so, and not with SAVE_FOR_BWD */,
       VPX_FLAG_X_SPLIT = 4 /* ConvLSTM forward: `x` holds the input sequence in the split-bf16 operand format (below) instead of
                              fp32 — only where vpx_convlstm_takes_split_input() says so; saves the conversion pass */,
       VPX_FLAG_SAVE_FOR_BWD = 1 /* forward fills `reserve` (gate activations + cell states per step) */,
       VPX_FLAG_WEIGHTS_PACKED = 2 /* ConvLSTM forward: `workspace` still holds the weight packs of a previous forward call with the SAME
                                      descriptor (flags aside), weight values and set of present operands (x / h0 NULL or not) — the caller
                                      keeps one workspace per block; saves the repack launches of a small-batch inference step.
                                      ST-LSTM: `workspace` still holds the repacked weights (LayerNorm variant, forward: and the
                                      transposed LayerNorm parameters) of a previous call with the SAME values and desc (caller keeps one workspace per cell per forward; the
                                      backward has its own workspace and additionally needs the same set of requested
                                      data gradients dx / dh / dm as the call that packed) */ };

typedef struct vpx_convlstm_desc {
    int32_t B, T, Cin, Ch, H, W, kh, kw; /* padding is kh/2, kw/2 ("same"), stride 1 — the only form the cells use */
    int32_t gate_order;                   /* VPX_GATE_* */
    int32_t layout;                       /* VPX_LAYOUT_* */
    int32_t precision;                    /* VPX_PREC_* */
    int32_t flags;                        /* VPX_FLAG_* */
} vpx_convlstm_desc;

typedef struct vpx_stlstm_desc {
    int32_t B, Cin, Ch, H, W, k; /* filter_size k (odd), stride 1, padding k/2: predrnn.py:22 */
    int32_t layer_norm;          /* 0/1: LayerNorm([C,H,W]) after conv_x/h/m/o (predrnn.py:24-40) */
    int32_t layout, precision, flags;
} vpx_stlstm_desc;

int vpx_version(void);
const char* vpx_last_error(void);
/* Run-to-run bit reproducibility (the counterpart of torch.use_deterministic_algorithms, which the reference inherits
 * from PyTorch). Default 0: convolutions on small feature maps split their contraction over workgroups and combine
 * partial sums with floating-point atomics (results equal up to fp32 summation order, ~1e-7 relative). 1: no atomics
 * anywhere (slower on 16x16 / 32x32 maps; TrajGRU's warp backward switches to integer atomics, whose sum is order-independent:
 * vpx_trajgru_warp_bwd_det). Process-wide; returns the previous setting. */
int vpx_set_deterministic(int on);
/* Kernel-selection switches for A/B measurements and parity tests inside one process (results never depend on them beyond
 * fp32 summation order). Returns the previous value, or a negative error code for an unknown option.
 *   VPX_OPT_CELL2        0 = the first-generation fused cell kernel everywhere; 1 (default) = the second-generation kernel
 *                        (pre-split operands, LDS-DMA staging) where it applies and fills the chip; 2 = wherever it applies */
#define VPX_OPT_CELL2 1
/*   VPX_OPT_CELL3        small grids (small batch and / or 16x16 - 32x32 maps): 1 (default) = the sliced fused step (8-channel
 *                        slices, weights resident in LDS, no atomics) where it applies; 0 = K-split convolution + gate kernel */
#define VPX_OPT_CELL3 2
/*   VPX_OPT_MFMA_SHAPE   bf16 MFMA shape of the second-generation kernels' main loop (fused cell step, its data gradient): 0 =
 *                        v_mfma_f32_32x32x16_bf16, 1 = v_mfma_f32_16x16x32_bf16 (K = 32 steps pair two taps of a 16-channel
 *                        stage; same wave tile, same LDS traffic, the chip holds a higher clock on it). Same products, same
 *                        operand split: results differ in fp32 summation order only */
#define VPX_OPT_MFMA_SHAPE 3
/*   VPX_OPT_EXPERIMENT   bits of kernel variants under measurement (A/B inside one process; 0 = the product's behaviour):
 *                        1 no sync-point stagger in the 32x16-tile cell; 4 the 32x16 tile instead of the half tile (fused cell,
 *                        data gradient); 8 (+ distance << 8) diagnostic placement of a pixel tile's N tiles in the dispatch order;
 *                        16 the 32x16 tile for the stage-glue kernel (convq); 32 the hoisted input projection of the small-grid
 *                        paths on the first-generation kernel; 64 the ST-LSTM step's k x k weight gradients on the first-generation
 *                        launches instead of the one-launch split-operand kernel (stw, wgrad2.hip); 128 its k x k data gradients on the
 *                        first-generation kernel instead of the 16x16-tile job-table kernel (c5, convq.hip); 256 the same for its forward
 *                        launches (gate groups, conv_o + output gate); 512 the 1x1 layers (conv_last, its adjoint, the decoupling
 *                        tail's adapter) on the implicit-GEMM kernel instead of the streaming one (c1, conv1.hip); 1024 the c5 launches
 *                        unsplit on grids below their pixel-tile bar (48 tiles forward, 96 backward; tests); 2048 the first-generation launches instead of the
 *                        K-split c5 jobs on those grids; 4096 the half tile of the fused ConvLSTM step
 *                        instead of its narrow-tile form (c3: c5_kernel<4, 3>) on grids of at most 256 half-tile workgroups;
 *                        8192 c3 with 32-column tiles (c5_kernel<2, 3>) instead of 64-column ones; 16384 the stage glue's data
 *                        gradients on the first-generation kernel where the schedule-driven one (convq) would take them; 32768 the fused
 *                        ConvLSTM step on the eight-wave half tile (cell2_kernel_x: 64-register wave tiles, four waves per SIMD) instead of the
 *                        four-wave one, 65536 its column split instead of the row split; 1 << 27 the ST-LSTM step's conv_last (1x1) on the fp32 c_new / m_new (converted in the
 *                        kernel) instead of on the split copies its gate stage leaves; 1 << 28 3x3 layers with 16 output channels on the
 *                        first-generation kernel instead of the resident-weights one (csrc/conv16.hip); 1 << 29 the stage glue's weight
 *                        gradients on the tap-group kernel (fp32 operands) instead of wgrad2_kernel's glue form on split copies */
#define VPX_OPT_EXPERIMENT 4
/*   VPX_OPT_DRY_RUN      1 = every entry point does all of its host-side work (argument checks, kernel selection, workspace carving
 *                        and the bounds checks of everything it would write into the workspace) but issues no HIP call: needs no GPU
 *                        and touches none of the pointers (they only have to be non-NULL where the call requires a tensor). A sizing
 *                        rule of a `*_workspace_bytes` query that disagrees with the launch code returns VPX_ERR_WORKSPACE. For the
 *                        CPU test-suite (tests/test_workspace_contract.py). Leaves no state behind: switch it off again and launch. */
#define VPX_OPT_DRY_RUN 5
int vpx_set_option(int option, int value);
/* A counter that advances with every vpx_set_option / vpx_set_deterministic call: callers that cache anything kernel-selection
 * dependent (a workspace with weight packs, VPX_FLAG_WEIGHTS_PACKED) key it on this value. */
int vpx_option_epoch(void);

/* ---- ConvLSTM over a sequence ------------------------------------------------------------------------------ */
size_t vpx_convlstm_workspace_bytes(const vpx_convlstm_desc* d); /* scratch, contents undefined between calls */
/* Split-bf16 operand format ("split"): a [N,H,W,C] fp32 NHWC tensor re-encoded, same byte count, as per pixel, per group of 8
 * channels: 8 hi bf16 (round-to-nearest of the value) then 8 lo bf16 (round-to-nearest of value - hi) — what the bf16x3
 * kernels multiply. Producers: vpx_conv2d_ex_fwd_split (the stage glue feeding a recurrent block). 1 = this descriptor's
 * forward consumes x in that form when VPX_FLAG_X_SPLIT is set (inference: the second-generation cell kernel, and the small-grid
 * kernel where its hoisted input projection runs on the schedule-driven convolution), 0 = fp32 only. */
int vpx_convlstm_takes_split_input(const vpx_convlstm_desc* d);
int vpx_convlstm_writes_split_output(const vpx_convlstm_desc* d);   /* 1 = VPX_FLAG_OUT_SPLIT is available for this descriptor */
size_t vpx_convlstm_reserve_bytes(const vpx_convlstm_desc* d);   /* saved-for-backward, 0 without SAVE_FOR_BWD */

/* x may be NULL (all-zero input: conv_lstm_hzzone.py:54-56), h0/c0 may be NULL (zero state: :40-45), bias may be NULL,
 * Wci/Wcf/Wco may be NULL together (no peephole = the ndrplz cell). hT/cT may be NULL. */
int vpx_convlstm_seq_fwd(const vpx_convlstm_desc* d, const float* x, const float* h0, const float* c0,
                         const float* W, const float* bias, const float* Wci, const float* Wcf, const float* Wco,
                         float* out, float* hT, float* cT, void* reserve, size_t reserve_bytes, void* workspace,
                         size_t workspace_bytes, void* stream);

/* BPTT. `out`/`reserve` are the forward's. dout/dhT/dcT may be NULL (zero). Every gradient output may be NULL
 * (skipped). dW/db/dWc* are OVERWRITTEN (not accumulated). */
int vpx_convlstm_seq_bwd(const vpx_convlstm_desc* d, const float* x, const float* h0, const float* c0,
                         const float* W, const float* Wci, const float* Wcf, const float* Wco, const float* out,
                         const void* reserve, size_t reserve_bytes, const float* dout, const float* dhT,
                         const float* dcT, float* dx, float* dh0, float* dc0, float* dW, float* db, float* dWci,
                         float* dWcf, float* dWco, void* workspace, size_t workspace_bytes, void* stream);

/* ---- ST-LSTM cell step (PredRNN-V2) -------------------------------------------------------------------------- */
size_t vpx_stlstm_workspace_bytes(const vpx_stlstm_desc* d);
size_t vpx_stlstm_reserve_bytes(const vpx_stlstm_desc* d);
/* Split-format shadows. The bf16x3 5x5 kernels read every activation in the split operand format (above); a tensor that a
 * previous step produced — h_new is the next step's h and the next layer's x, m_new the next layer's m — need not be converted
 * again when the caller hands its split copy back. The shadows are an ARGUMENT of the step call they belong to
 * (vpx_stlstm_step_fwd_ex / _bwd_ex; round 4 passed them through a thread-local setter that a failed or skipped call could leave
 * armed for an unrelated one — removed):
 *   vpx_stlstm_uses_split(d)   1 when calls with this descriptor consume / produce shadows, else 0 (they are ignored)
 *   in[5]  = {x, h, m, c_new, m_new}: NULL or the tensor once more in the split format, B*H*W*C*4 bytes (forward reads the first
 *            three, backward all five; a NULL entry is converted by the library as before)
 *   out[3] = {h_new, c_new, m_new}: NULL or caller buffers of B*H*W*Ch*4 bytes the forward fills with these outputs in the split
 *            format (ignored by the backward)
 *   dg8_out (backward only; NULL = off): DEFERRED WEIGHT GRADIENTS. The step writes its d(pre-activations) dG8 [B][H*W][8Ch]
 *            (gate blocks i,f,g | o | i',f',g' + d conv_last; split format, B*H*W*8Ch*4 bytes) there, computes the data gradients as
 *            usual and leaves dWx .. dWlast alone (they may be NULL). The caller keeps the dG8 of the T steps of one cell and the five
 *            sources x, h, m, c_new, m_new (split format) in dense slabs [T][B][H*W][C] and calls vpx_stlstm_wgrad_batch ONCE with a
 *            descriptor whose B is T*B: the same kernel over all the steps' images — one launch, one slab reduction and no per-step
 *            accumulation of the results. Available where vpx_stlstm_defers_wgrad(d) says 1 (5x5, bf16x3, channels in 8s, no LayerNorm).
 *   NHWC layout only: on reference-layout (NCHW) descriptors in[] / out[] are ignored and a non-NULL dg8_out is refused with
 *   VPX_ERR_UNSUPPORTED (never silently dropped). Without dg8_out a NULL dW pointer means "this weight is frozen": that gradient is
 *   not computed.
 *   `shadows` may be NULL: vpx_stlstm_step_fwd / _bwd are exactly that. */
typedef struct vpx_stlstm_shadows { const void* in[5]; void* out[3]; void* dg8_out; } vpx_stlstm_shadows;
int vpx_stlstm_uses_split(const vpx_stlstm_desc* d);
int vpx_stlstm_defers_wgrad(const vpx_stlstm_desc* d);
size_t vpx_stlstm_wgrad_batch_workspace_bytes(const vpx_stlstm_desc* d);   /* d->B = all images of the batch (T*B) */
/* dg8_split [N][H*W][8Ch], src5_split = {x [N][H*W][Cin], h, m, c_new, m_new [N][H*W][Ch]} with N = d->B, all in the split format;
 * weight gradients in reference layout (as vpx_stlstm_step_bwd), OVERWRITTEN. */
int vpx_stlstm_wgrad_batch(const vpx_stlstm_desc* d, const void* dg8_split, const void* const* src5_split, float* dWx, float* dWh,
                           float* dWm, float* dWo, float* dWlast, void* workspace, size_t workspace_bytes, void* stream);

/* Weights in reference layout: Wx [7Ch,Cin,k,k] Wh [4Ch,Ch,k,k] Wm [3Ch,Ch,k,k] Wo [Ch,2Ch,k,k] Wlast [Ch,2Ch,1,1].
 * ln: NULL or 8 pointers {x_gamma,x_beta,h_gamma,h_beta,m_gamma,m_beta,o_gamma,o_beta}, each in reference [C,H,W]. */
int vpx_stlstm_step_fwd(const vpx_stlstm_desc* d, const float* x, const float* h, const float* c, const float* m,
                        const float* Wx, const float* Wh, const float* Wm, const float* Wo, const float* Wlast,
                        const float* const* ln, float* h_new, float* c_new, float* m_new, float* delta_c,
                        float* delta_m, void* reserve, size_t reserve_bytes, void* workspace, size_t workspace_bytes,
                        void* stream);

/* c_new / m_new are the forward's outputs (the operands of conv_o / conv_last). Incoming gradients may be NULL (zero);
 * every gradient output may be NULL (skipped); weight gradients are OVERWRITTEN. ln / dln: the 8 LayerNorm parameter
 * tensors and their gradients (reference layout [C,H,W]); NULL unless desc.layer_norm. */
int vpx_stlstm_step_bwd(const vpx_stlstm_desc* d, const float* x, const float* h, const float* c, const float* m,
                        const float* c_new, const float* m_new,
                        const float* Wx, const float* Wh, const float* Wm, const float* Wo, const float* Wlast,
                        const float* const* ln, const void* reserve, size_t reserve_bytes, const float* dh_new,
                        const float* dc_new, const float* dm_new, const float* ddelta_c, const float* ddelta_m,
                        float* dx, float* dh, float* dc, float* dm, float* dWx, float* dWh, float* dWm, float* dWo,
                        float* dWlast, float* const* dln, void* workspace, size_t workspace_bytes, void* stream);
/* the same calls with split-format shadows (see above) */
int vpx_stlstm_step_fwd_ex(const vpx_stlstm_desc* d, const float* x, const float* h, const float* c, const float* m,
                           const float* Wx, const float* Wh, const float* Wm, const float* Wo, const float* Wlast,
                           const float* const* ln, float* h_new, float* c_new, float* m_new, float* delta_c,
                           float* delta_m, void* reserve, size_t reserve_bytes, void* workspace, size_t workspace_bytes,
                           void* stream, const vpx_stlstm_shadows* shadows);
int vpx_stlstm_step_bwd_ex(const vpx_stlstm_desc* d, const float* x, const float* h, const float* c, const float* m,
                           const float* c_new, const float* m_new,
                           const float* Wx, const float* Wh, const float* Wm, const float* Wo, const float* Wlast,
                           const float* const* ln, const void* reserve, size_t reserve_bytes, const float* dh_new,
                           const float* dc_new, const float* dm_new, const float* ddelta_c, const float* ddelta_m,
                           float* dx, float* dh, float* dc, float* dm, float* dWx, float* dWh, float* dWm, float* dWo,
                           float* dWlast, float* const* dln, void* workspace, size_t workspace_bytes, void* stream,
                           const vpx_stlstm_shadows* shadows);

/* ---- decoupling-loss term: mean_{b,ch} |cos(normalize(A*dc), normalize(A*dm))| over H*W ---------------------- */
/* delta_c/delta_m [B,H,W,Ch] (NHWC) ; adapter [Ch,Ch] ; value / dvalue: 1 float on device.
 * precision (vpx_precision): arithmetic of the tail's three 1x1 contractions (adapter, its adjoint, its weight gradient) —
 * the caller passes the model's own operand mode, so a VPX_PREC_F32 model gets exact fp32 here too (round 3 ran this tail in
 * bf16x3 for every caller). The normalisation / cosine / mean are fp32 with double partial sums in every mode. */
size_t vpx_decouple_workspace_bytes(int B, int Ch, int H, int W);
int vpx_decouple_fwd(const float* delta_c, const float* delta_m, const float* adapter, float* value, int B, int Ch,
                     int H, int W, int precision, void* workspace, size_t workspace_bytes, void* stream);
int vpx_decouple_bwd(const float* delta_c, const float* delta_m, const float* adapter, const float* dvalue,
                     float* d_delta_c, float* d_delta_m, float* d_adapter, int B, int Ch, int H, int W, int precision,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ---- training tail (the caller side of the path: base_model.py:168-176, vpsuite.py:353) ------------------------- */
/* loss = scale * mean_{b,t} sum_{c,h,w} (pred - target)^2   (base_measure.py:55-57, image_wise.py:25, loss_provider.py:48-51)
 * pred/target: n_elements = B*T*C*H*W floats in any (identical) layout, n_frames = B*T; loss: 1 float on device;
 * dpred (nullable): d loss / d pred, written in the same pass. Deterministic (fixed reduction order, double partials). */
size_t vpx_mse_loss_workspace_bytes(void);
int vpx_mse_loss(const float* pred, const float* target, long long n_elements, long long n_frames, float scale,
                 float* loss, float* dpred, void* workspace, size_t workspace_bytes, void* stream);
/* One torch.optim.Adam step (amsgrad off) over flat, 16-byte aligned buckets of n floats; step >= 1 is the iteration
 * count after this update; grad is multiplied by grad_scale first (1/world_size after a summing all-reduce).
 * Hyper-parameters are doubles: the derived scalars (1 - beta, lr / (1 - beta1^t), ...) are formed in double as PyTorch
 * forms them in Python floats, and rounded to fp32 once. */
int vpx_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, double lr,
                  double beta1, double beta2, double eps, double weight_decay, int step, double grad_scale, void* stream);

/* ---- plain stride-1 "same" convolution, NHWC, optional bias; y [N,H,W,Co] = conv(x [N,H,W,Ci], w [Co,Ci,kh,kw]) --- */
size_t vpx_conv2d_workspace_bytes(int Ci, int Co, int kh, int kw);
int vpx_conv2d_nhwc_fwd(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Ci,
                        int Co, int kh, int kw, int precision, void* workspace, size_t workspace_bytes, void* stream);

/* backward of the same convolution: dx [N,H,W,Ci], dw [Co,Ci,kh,kw], db [Co]; each may be NULL (skipped), all OVERWRITTEN */
size_t vpx_conv2d_bwd_workspace_bytes(int N, int H, int W, int Ci, int Co, int kh, int kw);
int vpx_conv2d_nhwc_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int N, int H,
                        int W, int Ci, int Co, int kh, int kw, int precision, void* workspace, size_t workspace_bytes,
                        void* stream);

/* ---- general 2-D convolution / transposed convolution with fused bias + LeakyReLU (the EF "stage glue":
 *      vp_suite/models/precipitation_nowcasting/ef_blocks.py:15-49, layer table ef_conv_lstm.py:36-65) -------------- */
typedef struct vpx_conv_desc {
    int32_t N, H, W, Ci, Co;       /* input x [N,H,W,Ci] (NHWC) */
    int32_t kh, kw, stride, pad;   /* stride 1 or 2; any padding */
    int32_t transposed;            /* 0: nn.Conv2d weight [Co,Ci,kh,kw]; 1: nn.ConvTranspose2d weight [Ci,Co,kh,kw] */
    float leaky_slope;             /* LeakyReLU negative slope fused after the bias; 0 = no activation */
    int32_t precision;             /* VPX_PREC_* */
    int32_t out_pad_h, out_pad_w;  /* transposed only: nn.ConvTranspose2d output_padding (0 .. stride-1) */
} vpx_conv_desc;
int vpx_conv2d_ex_out_shape(const vpx_conv_desc* d, int* Ho, int* Wo);
size_t vpx_conv2d_ex_workspace_bytes(const vpx_conv_desc* d);
/* y [N,Ho,Wo,Co]. A stride-2 transposed convolution runs as 4 output-phase launches of the same kernel. */
int vpx_conv2d_ex_fwd(const vpx_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                      void* workspace, size_t workspace_bytes, void* stream);
/* The same layer with the output (also) in the split-bf16 operand format (see vpx_convlstm_takes_split_input): y_split
 * [N,Ho,Wo,Co] split-encoded, Co % 8 == 0; y may be NULL (inference: nobody reads the fp32 copy). */
int vpx_conv2d_ex_fwd_split(const vpx_conv_desc* d, const float* x, const float* w, const float* bias, float* y, void* y_split,
                            void* workspace, size_t workspace_bytes, void* stream);
/* The same layer on SPLIT-format input (x_split: [N,H,W,Ci] split-encoded; image n at (n / x_nT) * x_bstride + (n % x_nT) *
 * x_tstride bytes, x_bstride = 0: dense, x_nT <= 1: plain batch), on the schedule-driven K = 32 kernel (csrc/convq.hip): bf16x3,
 * Ci % 16 == 0, stride 1 or 2, taps within one pixel of the (sub-)image grid (3x3 pad 1, 4x4 stride 2 pad 1, plain or transposed)
 * — vpx_conv2d_ex_takes_split says whether a descriptor qualifies (0 no; 1 yes; 2 yes, and on the schedule-driven K = 32 kernel,
 *   which is worth a vpx_split_convert of an fp32 input for stride-2 transposed layers — or, for 3x3 stride-1 pad-1 layers with 16
 *   output channels and 16..64 input channels, on the resident-weights kernel of csrc/conv16.hip). y (fp32) and y_split may each be NULL, not both.
 * weights_packed: the workspace still holds this layer's packed weights (same values, same descriptor) — the pack launch is skipped
 *   (both kernels; on the first-generation kernel for the single-launch forms: a stride-2 transposed layer's four phase launches
 *   share the space and pack every time). */
int vpx_conv2d_ex_takes_split(const vpx_conv_desc* d);
/* fp32 channels-last pixels [n_pixels][C] -> the split-bf16 operand format (per pixel and 8 channels: 8 hi bf16, 8 lo bf16; the
 * same n_pixels * C * 4 bytes), C % 8 == 0: what the recurrent blocks and vpx_conv2d_ex_fwd_split write themselves. */
int vpx_split_convert(const float* x, void* x_split, long long n_pixels, int C, void* stream);
size_t vpx_conv2d_ex_split_workspace_bytes(const vpx_conv_desc* d);
int vpx_conv2d_ex_fwd_from_split(const vpx_conv_desc* d, const void* x_split, long long x_bstride, long long x_tstride, int x_nT,
                                 const float* w, const float* bias, float* y, void* y_split, int weights_packed, void* workspace,
                                 size_t workspace_bytes, void* stream);
/* Backward of the same layer (the reference gets it from autograd over nn.Conv2d / nn.ConvTranspose2d (+ LeakyReLU),
 * ef_blocks.py:15-49): dy [N,Ho,Wo,Co] is the gradient w.r.t. the layer OUTPUT (after bias and activation); y is that
 * output as vpx_conv2d_ex_fwd produced it — needed (and only read) when d->leaky_slope != 0: the activation derivative is
 * taken from its sign. dx [N,H,W,Ci], dw (layout of w) and db [Co] are written; each may be NULL. db and the LeakyReLU'
 * scaling are one pass over dy, summed in a fixed order (bit-reproducible).
 * dx is the adjoint layer run forward (transposed <-> plain, through vpx_conv2d_ex_fwd's kernels); dw contracts dy with
 * the stride-decimated sub-images of x (or x with those of dy) on the MFMA weight-gradient kernel. Needs kh, kw >= stride. */
size_t vpx_conv2d_ex_bwd_workspace_bytes(const vpx_conv_desc* d);
int vpx_conv2d_ex_bwd(const vpx_conv_desc* d, const float* x, const float* w, const float* y, const float* dy, float* dx,
                      float* dw, float* db, void* workspace, size_t workspace_bytes, void* stream);
/* Round 6: with bf16x3 operands and channel counts in groups of 8 (vpx_conv2d_ex_bwd_uses_split(d) != 0) the weight gradient runs
 * per stride residue on split copies of x and dy (csrc/wgrad2.hip, glue form). _bwd converts x itself; a caller that ran the forward
 * through vpx_conv2d_ex_fwd_from_split hands that x_split [N,H,W,Ci] back to _bwd_ex and saves the pass (x stays required: residues of a
 * single tap and the other operand modes read it). x_split == NULL: exactly vpx_conv2d_ex_bwd. */
int vpx_conv2d_ex_bwd_uses_split(const vpx_conv_desc* d);
int vpx_conv2d_ex_bwd_ex(const vpx_conv_desc* d, const float* x, const void* x_split, const float* w, const float* y, const float* dy,
                         float* dx, float* dw, float* db, void* workspace, size_t workspace_bytes, void* stream);

/* y = act(conv(x, w) + bias [+ y]): the stride-1 "same" convolution of vpx_conv2d_nhwc_fwd with an optional accumulate
 * into the destination (two convolutions summed into one output) and LeakyReLU (slope >= 0, 0 = none) applied to the sum.
 * Workspace: vpx_conv2d_workspace_bytes(Ci, Co, kh, kw). */
int vpx_conv2d_nhwc_fwd_ex(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Ci, int Co,
                           int kh, int kw, int precision, int accumulate, float leaky_slope, void* workspace,
                           size_t workspace_bytes, void* stream);
/* ---- TrajGRU over a sequence (vp_suite/model_blocks/traj_gru.py:164-214), time-major NHWC ---------------------------------- *
 * One call runs the block's whole time loop: the input projection i2h of all frames (one launch), then per step the flow generator
 * (two 5x5 convolutions summed + LeakyReLU, the 5x5 flow convolution, :134-146), the L bilinear warps of h_{t-1} along -flow with the
 * reference's normalisation (divide by W-1 / H-1, grid_sample align_corners=False, zero padding, :148-162), the 1x1 `ret`
 * convolution and the GRU gates r = s(i0+h0), u = s(i1+h1), m = leaky(i2 + r*h2), h_t = u*h_{t-1} + (1-u)*m (:190-203). The backward
 * is the explicit BPTT schedule of the same launches (the warped operand is recomputed per step, not stored).
 *   x  [T][B][H*W][Cin] or NULL (no input: zero projection)      h0 [B][H*W][C] or NULL (zero state); not both NULL
 *   params / dparams: 10 pointers in the order (i2h, i2f_conv1, h2f_conv1, flows_conv, ret) x (weight OIHW, bias); with x == NULL the
 *        first four gradients are not written (may be NULL). Parameter gradients are OVERWRITTEN.
 *   hs [T][B][H*W][C]: h_1 .. h_T (forward output, backward input)       dout [T][B][H*W][C] or NULL, dhT [B][H*W][C] or NULL
 *   reserve: vpx_trajgru_reserve_bytes (0 without VPX_FLAG_SAVE_FOR_BWD); workspace: vpx_trajgru_workspace_bytes
 * The warp backward scatters with float atomics (like torch's grid_sample backward: sum order depends on timing); under
 * vpx_set_deterministic(1) it scatters 2^40-scaled 64-bit integers instead (associative, hence bit-reproducible; resolution 9e-13,
 * range +-8.4e6 per element). Restrictions: C % 4 == 0, i2h a square odd stride-1 'same' convolution, slope > 0, zoneout = 0. */
typedef struct vpx_trajgru_desc {
    int32_t B, T, Cin, C, H, W;
    int32_t L;             /* flow fields / warps per step */
    int32_t k_i2h;         /* kernel size of the input projection */
    int32_t precision;     /* vpx_precision: arithmetic of the five convolutions */
    int32_t flags;         /* VPX_FLAG_SAVE_FOR_BWD */
    float slope;           /* LeakyReLU negative slope (> 0) */
} vpx_trajgru_desc;
size_t vpx_trajgru_workspace_bytes(const vpx_trajgru_desc* d);
size_t vpx_trajgru_reserve_bytes(const vpx_trajgru_desc* d);
int vpx_trajgru_seq_fwd(const vpx_trajgru_desc* d, const float* x, const float* h0, const float* const* params, float* hs, void* reserve,
                        size_t reserve_bytes, void* workspace, size_t workspace_bytes, void* stream);
int vpx_trajgru_seq_bwd(const vpx_trajgru_desc* d, const float* x, const float* h0, const float* const* params, const float* hs,
                        const void* reserve, size_t reserve_bytes, const float* dout, const float* dhT, float* dx, float* dh0,
                        float* const* dparams, void* workspace, size_t workspace_bytes, void* stream);

/* ---- action-conditional ST-LSTM cell, one step (vp_suite/model_blocks/predrnn.py:86-169), NHWC ------------------------------ *
 * replaces ActionConditionalSpatioTemporalLSTMCell.forward (:139-169) and its autograd: x_concat = conv_x(x), h_concat = conv_h(h),
 * a_concat = conv_a(a), m_concat = conv_m(m) — Conv2d WITH bias, each followed by LayerNorm([C,H,W]) when layer_norm (:102-136) —
 * h_concat * a_concat (:144), both gate groups and the state updates (:146-164), conv_o / conv_last on mem = (c_new | m_new), the
 * output gate (:165-167). Tensors are [B][H*W][C] (x: Cin channels; h, c, m, a and all outputs: Ch).
 *   params / dparams: 12 pointers, (conv_x, conv_h, conv_a, conv_m, conv_o, conv_last) x (weight OIHW, bias); conv_x [7Ch,Cin,k,k],
 *        conv_h / conv_a [4Ch,Ch,k,k], conv_m [3Ch,Ch,k,k], conv_o [Ch,2Ch,k,k], conv_last [Ch,2Ch,1,1]. A NULL dparams entry is skipped
 *        (frozen parameter); gradients are OVERWRITTEN.
 *   ln / dln: 10 pointers, (x, h, a, m, o) x (weight, bias) in the reference's [C,H,W] layout; NULL unless layer_norm (eps = 1e-5).
 *   backward: dh_new is required; dc_new / dm_new / ddc / ddm (gradients of c_new, m_new, delta_c, delta_m) may be NULL = zero;
 *        dx / dh / dc / dm / da may be NULL (not wanted).
 *   reserve: vpx_acstlstm_reserve_bytes (0 without VPX_FLAG_SAVE_FOR_BWD), written by the forward, read by the backward.
 * First-generation convolution kernels in every precision; k odd, stride 1. */
typedef struct vpx_acstlstm_desc {
    int32_t B, Cin, Ch, H, W;
    int32_t k;             /* kernel size of conv_x / conv_h / conv_a / conv_m / conv_o ('same', stride 1) */
    int32_t layer_norm;    /* LayerNorm after those five convolutions */
    int32_t precision;     /* VPX_PREC_* of the six convolutions */
    int32_t flags;         /* VPX_FLAG_SAVE_FOR_BWD */
    float forget_bias;     /* added to both forget gates' pre-activations (the reference: 1.0) */
} vpx_acstlstm_desc;
size_t vpx_acstlstm_workspace_bytes(const vpx_acstlstm_desc* d);
size_t vpx_acstlstm_reserve_bytes(const vpx_acstlstm_desc* d);
int vpx_acstlstm_step_fwd(const vpx_acstlstm_desc* d, const float* x, const float* h, const float* c, const float* m, const float* a,
                          const float* const* params, const float* const* ln, float* h_new, float* c_new, float* m_new, float* delta_c,
                          float* delta_m, void* reserve, size_t reserve_bytes, void* workspace, size_t workspace_bytes, void* stream);
int vpx_acstlstm_step_bwd(const vpx_acstlstm_desc* d, const float* x, const float* h, const float* c, const float* m, const float* a,
                          const float* const* params, const float* const* ln, const void* reserve, size_t reserve_bytes, const float* dh_new,
                          const float* dc_new, const float* dm_new, const float* ddc, const float* ddm, float* dx, float* dh, float* dc, float* dm,
                          float* da, float* const* dparams, float* const* dln, void* workspace, size_t workspace_bytes, void* stream);

/* ---- LayerNorm([C,H,W]) per sample on NHWC tensors (predrnn.py:27-40, 105-135; eps = 1e-5) ---------------------------------- *
 * x, y, xhat: [B][n = H*W*C]; gamma, beta (and dgamma, dbeta): [H*W][C], i.e. the reference's [C,H,W] parameters channels-last;
 * stats [B][2] = (mean, 1/std) and xhat (optional in the forward: NULL skips the store) feed the backward. Sums run in double over
 * fixed chunks (bit-reproducible); dgamma / dbeta are OVERWRITTEN. */
size_t vpx_layernorm_workspace_bytes(int B);
int vpx_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* xhat, float* stats, int B, long long n,
                      void* workspace, size_t workspace_bytes, void* stream);
int vpx_layernorm_bwd(const float* dy, const float* xhat, const float* stats, const float* gamma, float* dx, float* dgamma, float* dbeta,
                      int B, int HW, int C, void* workspace, size_t workspace_bytes, void* stream);

/* ---- layout adaptors: src [N,C,H,W] <-> dst [N,H,W,C] -------------------------------------------------------- */
int vpx_nchw_to_nhwc(const float* src, float* dst, int N, int C, int H, int W, void* stream);
int vpx_nhwc_to_nchw(const float* src, float* dst, int N, int C, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VPX_H_ */
