"""PredRNN-V2 ("predrnn-pp") — drop-in for vp_suite/models/predrnn_v2.py:11-365, incl. the action-conditional form.

Hyper-parameter names/defaults, state_dict keys (`cell_list.{l}.conv_{x,h,m,o}.0.weight`,
`cell_list.{l}.conv_last.weight`, `conv_last.weight`, `adapter.weight`), the forward contract
`forward(x[b,T_total,c,h,w], pred_frames, train=...) -> (pred[b,p,c,h,w], {"ST-LSTM decouple loss": scalar})`,
scheduled-sampling schedules and the custom train_iter (forward + reversed forward, averaged) follow the reference.
Every ST-LSTM cell step (4 fused launches), the decoupling-loss tail (K4) and the 1x1 frame head (K5) run in
libvpx_hip; patchify is a pure permutation and the input blend one elementwise expression.
Action-conditional form (:62-121, 143-149, 178-221): `action_conditional=True` forces `conv_actions_on_input` and
`reverse_scheduled_sampling` like the reference; frames and the spatially broadcast actions pass two stride-2 5x5
convolutions each (`conv_input1/2`, `action_conv_input1/2`), the ST-LSTM cells multiply conv_h(h) by conv_a(action)
(`ActionConditionalSpatioTemporalLSTMCell`) and two stride-2 transposed convolutions (`deconv_output1/2`, optionally with
the encoder residuals) map the top hidden state back to a patch frame — all of them on the library's convolution
kernels (vpx_conv2d_ex_fwd/_bwd, output padding resolved from the requested size like `output_size=` does)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..base import VPModel, _progress
from ..model_blocks import ActionConditionalSpatioTemporalLSTMCell as ACSTCell
from ..model_blocks import SpatioTemporalLSTMCell as STCell


class _CellBanks:
    """The operand slabs of one training forward (split operand format, time-major, dense) and one ops.STWeightBank per cell.
    Slot s of a state slab = that state after step s - 1 (slot 0: the zero initial state), so every operand of step t of every cell is
    slot t (+ a constant) of ONE slab: h of cell i: H[i] slot t; x of cell i > 0: H[i-1] slot t + 1; m of cell 0: M[L-1] slot t (the
    zig-zag memory of the previous step's top cell), of cell i > 0: M[i-1] slot t + 1; the step's own c_new / m_new: slot t + 1 of C[i] /
    M[i]; the frames entering cell 0: X slot t (converted by the step)."""

    def __init__(self, model, geo, T):
        dev = model.adapter.weight.device
        self.L = L = len(geo)
        new = lambda slots, n: torch.empty(slots, n, dtype=torch.float32, device=dev)   # noqa: E731
        b, cin0, ch, h, w, k = geo[0]
        n = b * h * w * ch
        self.H, self.C, self.M = [new(T + 1, n) for _ in range(L)], [new(T + 1, n) for _ in range(L)], [new(T + 1, n) for _ in range(L)]
        for slab in self.H + self.M:
            slab[0].zero_()
        self.X = new(T, b * h * w * cin0)
        self.banks, self.weights = [], []
        self.precision = model.cell_precision   # ONE source for the banks and for every step they serve
        for i, g in enumerate(geo):
            cell = model.cell_list[i]
            bank = ops.STWeightBank((cell.conv_x[0].weight, cell.conv_h[0].weight, cell.conv_m[0].weight, cell.conv_o[0].weight, cell.conv_last.weight),
                                    *g, T, ops.PRECISIONS[model.cell_precision])
            bank.set_sources(self.X[0] if i == 0 else self.H[i - 1][1], self.H[i][0], self.M[L - 1][0] if i == 0 else self.M[i - 1][1],
                             self.C[i][1], self.M[i][1])
            self.banks.append(bank)
            self.weights.append(bank.weights())

    def step(self, cell, i, t, x, h, c, m, delta_out=None):
        x_sp = self.X[t] if i == 0 else self.H[i - 1][t + 1]
        m_sp = self.M[self.L - 1][t] if i == 0 else self.M[i - 1][t + 1]
        slots = (self.banks[i], t, (x_sp, self.H[i][t], m_sp), (self.H[i][t + 1], self.C[i][t + 1], self.M[i][t + 1]), i == 0)
        return ops.stlstm_step(x, h, c, m, *self.weights[i], precision=self.precision, wsholder=cell._ws, slots=slots, delta_out=delta_out)


class PredRNN_V2(VPModel):
    NAME = "PredRNN++"
    PAPER_REFERENCE = "https://arxiv.org/abs/2103.09504"
    CODE_REFERENCE = "https://github.com/thuml/predrnn-pytorch"
    MATCHES_REFERENCE: str = "Yes"
    CAN_HANDLE_ACTIONS = True
    NEEDS_COMPLETE_INPUT = True

    patch_size = 4
    num_layers = 3
    num_hidden = [128, 128, 128, 128]
    filter_size = 5
    stride = 1
    inflated_action_dim = 3
    layer_norm: bool = False
    conv_actions_on_input: bool = True
    residual_on_action_conv: bool = True

    reverse_input: bool = True
    decoupling_loss_scale = 100.0
    scheduled_sampling: bool = True
    sampling_stop_iter: int = 50000
    sampling_changing_rate = 2e-5
    reverse_scheduled_sampling: bool = False
    r_sampling_step_1: int = 25000
    r_sampling_step_2: int = 50000
    r_exp_alpha: int = 5000
    training_iteration: int = None
    sampling_eta: float = None
    cell_precision = "f32"  #: arithmetic of the fused ST-LSTM kernels
    #: training_loss runs the sequence and its time-reversal (predrnn_v2.py:326-352) as ONE batch of 2B samples: the two passes share the
    #: weights and nothing else, so this is the same arithmetic per sample with every launch on twice the grid and half the launches
    #: (what the 2-sample shards of BASELINE configs[4] need most). False: two forward passes one after the other, as the reference does.
    #: REQUIRES every term of `loss_provider` to be a mean over the batch of per-sample values (MSE and the decoupling term are; a
    #: batch-coupled or sum-normalised measure is not: set this to False for those). Costs the activation memory of a 2B batch; falls back
    #: to two passes by itself when `actions` is not a [B, T, a] tensor or when a 2B pass would not fit the free device memory.
    fuse_reversed_pass: bool = True
    #: training: the five weight gradients of a cell are computed ONCE per forward pass over all of its steps (ops.STWeightBank) instead of
    #: once per step with an autograd accumulation per step and tensor. Same sums in another order. Where the library cannot (LayerNorm,
    #: action-conditional cells, operand modes other than bf16x3, filter sizes other than 5) the steps compute them as before.
    defer_weight_gradients: bool = True
    #: ... while the operand slabs of a pass (dG8 of every step and cell: T*B*H*W*8Ch*4 bytes per cell, plus the state slabs) stay below
    #: this many bytes; a larger pass computes the weight gradients step by step as before (nothing is kept beyond a step's own context)
    #: (and below half of the device memory that is free when the forward starts)
    BANK_BYTES_LIMIT = 64 << 30
    #: the decoupling term of all layer-steps of a pass in ONE library call each way (ops.decouple_term_batched): the steps write their
    #: delta_c / delta_m into one slab. False: one adapter convolution + statistics + mean per layer-step, as the reference does.
    #: Used while the slab stays below `DECOUPLE_SLAB_LIMIT` bytes: beyond that the per-step tails win back what they lose in launches
    #: by finding their operands in the Infinity Cache (measured at 2B = 256, 16x16 maps: 3.8 GB slab, 316 vs 313 ms per training step;
    #: at B = 32: 88.2 vs 90.5 ms; the 128x128x3 shard at 2B = 4: 89.8 vs 94.7 ms; its inference at B = 4: 28.0 vs 31.0 ms).
    batch_decoupling_tail: bool = True
    DECOUPLE_SLAB_LIMIT = 2 << 30

    def __init__(self, device, **model_kwargs):
        super().__init__(device, **model_kwargs)
        self.patch_c = self.patch_size * self.patch_size * self.img_c
        self.patch_a = self.action_size
        self.patch_h = self.rnn_h = self.img_h // self.patch_size
        self.patch_w = self.rnn_w = self.img_w // self.patch_size
        if self.action_conditional:  # the action-conditional graph only exists in this form (predrnn_v2.py:64-70)
            self.conv_actions_on_input = True
            self.reverse_scheduled_sampling = True
        else:
            self.conv_actions_on_input = False
            self.residual_on_action_conv = False
        nh, k = self.num_hidden, self.filter_size
        if self.conv_actions_on_input:
            if self.rnn_h % 4 or self.rnn_w % 4:
                raise ValueError(f"action-conditional {self.NAME}: the patch grid {self.rnn_h}x{self.rnn_w} must be divisible by 4 "
                                 f"(two stride-2 convolutions, predrnn_v2.py:73-75)")
            self.rnn_h //= 4
            self.rnn_w //= 4

            def down(c_in, c_out):
                return nn.Conv2d(c_in, c_out, k, stride=2, padding=k // 2, bias=False)

            def up(c_in, c_out):
                return nn.ConvTranspose2d(c_in, c_out, k, stride=2, padding=k // 2, bias=False)
            self.conv_input1, self.conv_input2 = down(self.patch_c, nh[0] // 2), down(nh[0] // 2, nh[0])
            self.action_conv_input1, self.action_conv_input2 = down(self.patch_a, nh[0] // 2), down(nh[0] // 2, nh[0])
            top = nh[self.num_layers - 1]
            self.deconv_output1, self.deconv_output2 = up(top, top // 2), up(top // 2, self.patch_c)

        cells = []
        for i in range(self.num_layers):
            if i > 0:
                c_in = nh[i - 1]
            elif self.action_conditional:
                c_in = nh[0] if self.conv_actions_on_input else self.patch_c + self.patch_a
            else:
                c_in = self.patch_c
            cell = (ACSTCell if self.action_conditional else STCell)(c_in, nh[i], self.rnn_h, self.rnn_w, k, self.stride,
                                                                    self.layer_norm)
            cell.precision = self.cell_precision
            cells.append(cell)
        self.cell_list = nn.ModuleList(cells)
        if not self.conv_actions_on_input:   # (absent when the deconvolutions produce the frame, predrnn_v2.py:108-114)
            self.conv_last = nn.Conv2d(nh[self.num_layers - 1], self.patch_c, kernel_size=1, stride=1, padding=0, bias=False)
        adap = nh[self.num_layers - 1] if self.action_conditional else nh[0]
        self.adapter = nn.Conv2d(adap, adap, kernel_size=1, stride=1, padding=0, bias=False)
        self.training_iteration = 1
        self.sampling_eta = 1.0
        self.NON_CONFIG_VARS.extend(["training_iteration, sampling_eta"])

    def pred_1(self, x, **kwargs):
        return self(x, pred_frames=1, **kwargs)[0].squeeze(dim=1)

    def _decouple_term(self, delta_c, delta_m):
        # adapter 1x1 conv + normalize + |cosine| + mean, fused in libvpx_hip (vpx_decouple_fwd/_bwd)
        return ops.decouple_term(delta_c, delta_m, self.adapter.weight, self.cell_precision)

    def forward(self, x, pred_frames: int = 1, **kwargs):
        b, total_frames = x.shape[:2]
        context_frames = total_frames - pred_frames
        if context_frames < 1:
            raise ValueError("Model {self.NAME} needs input sequences that also include the target frames!")
        train = kwargs.get("train", False)
        dev = x.device
        x_patch = self._reshape_patch(x)
        a_patch = None
        if self.action_conditional:
            actions = kwargs.get("actions", None)
            # predrnn_v2.py:141-147 raises for the missing-actions placeholder ([b, T] zeros, never equal to a [b, T, a] tensor) and
            # for a wrong last dimension; an all-zero [b, T, a] batch (a robot at rest) is valid there and here. No device sync.
            if actions is None or actions.dim() != 3 or actions.shape[-1] != self.action_size:
                raise ValueError("Given actions are None or of the wrong size!")
            a_patch = actions.to(dev)[..., None, None].expand(-1, -1, -1, self.patch_h, self.patch_w)
        prec = self.cell_precision
        nh, top = self.num_hidden, self.num_layers - 1
        # split-format shadows of the states live from one step of THIS loop to the next and no longer (ops.new_shadow_epoch)
        ops.new_shadow_epoch()
        banks = self._weight_banks(b, total_frames - 1) if (train and torch.is_grad_enabled()) else None
        n_ls = (total_frames - 1) * self.num_layers
        slab = None
        if self.batch_decoupling_tail and not self.action_conditional and len(set(nh[:self.num_layers])) == 1 and \
                8 * n_ls * b * nh[0] * self.rnn_h * self.rnn_w <= self.DECOUPLE_SLAB_LIMIT:
            slab = ops.new_channels_last((2, n_ls * b, nh[0], self.rnn_h, self.rnn_w), dev)   # [delta_c | delta_m][layer-step][sample]
        deltas = []

        def zeros(i):
            return torch.zeros(b, nh[i], self.rnn_h, self.rnn_w, device=dev)
        h_t = [zeros(i) for i in range(self.num_layers)]
        c_t = [zeros(i) for i in range(self.num_layers)]
        memory = zeros(0)
        # Test-time sampling masks are constants (predrnn_v2.py:300-309: zeros, and ones over the context frames in reverse mode): the
        # blend mask * x + (1 - mask) * x_gen (:171-176) then IS one of its two operands, bit for bit on finite inputs — taken directly instead of
        # through four elementwise launches per predicted frame (2 % of a small-batch forward).
        mask_true = kwargs.get("_mask_true")   # (training_loss: the masks of a fused forward + reversed pair, drawn in the reference's order)
        if mask_true is None and train:
            mask_true = self._scheduled_sampling(b, context_frames, pred_frames, train)
        first_blend = 1 if self.reverse_scheduled_sampling else context_frames
        k = self.filter_size
        x_gen, next_frames, decouple = None, [], []
        for t in range(total_frames - 1):
            if t < first_blend:
                net = x_patch[:, t]
            elif train:
                mk = mask_true[:, t - first_blend]
                net = mk * x_patch[:, t] + (1 - mk) * x_gen
            else:
                net = x_patch[:, t] if (self.reverse_scheduled_sampling and t < context_frames) else x_gen
            action = a_patch[:, t] if a_patch is not None else None
            if self.conv_actions_on_input:   # two stride-2 convolutions on the frame and on the action map (:178-188)
                shape1 = net.shape[-2:]
                net = in1 = ops.conv2d_ex(net, self.conv_input1.weight, None, 2, k // 2, False, 0.0, prec)
                shape2 = net.shape[-2:]
                net = in2 = ops.conv2d_ex(net, self.conv_input2.weight, None, 2, k // 2, False, 0.0, prec)
                action = ops.conv2d_ex(action.contiguous(), self.action_conv_input1.weight, None, 2, k // 2, False, 0.0, prec)
                action = ops.conv2d_ex(action, self.action_conv_input2.weight, None, 2, k // 2, False, 0.0, prec)
            for i in range(self.num_layers):
                inp = net if i == 0 else h_t[i - 1]
                if self.action_conditional:
                    h_t[i], c_t[i], memory, d_c, d_m = self.cell_list[i](inp, h_t[i], c_t[i], memory, action)
                else:
                    ls = t * self.num_layers + i
                    dout = None if slab is None else (slab[0, ls * b:(ls + 1) * b], slab[1, ls * b:(ls + 1) * b])
                    if banks is not None:
                        h_t[i], c_t[i], memory, d_c, d_m = banks.step(self.cell_list[i], i, t, inp, h_t[i], c_t[i], memory, dout)
                    else:
                        h_t[i], c_t[i], memory, d_c, d_m = self.cell_list[i](inp, h_t[i], c_t[i], memory, delta_out=dout, use_shadows=True,
                                                                             precision=prec)
                if slab is not None:
                    deltas += [d_c, d_m]
                else:
                    decouple.append(self._decouple_term(d_c, d_m))
            if self.conv_actions_on_input:   # two stride-2 transposed convolutions back to the patch grid (:212-218)
                res2, res1 = (in2, in1) if self.residual_on_action_conv else (0, 0)
                x_gen = ops.conv_transpose2d_to_size(h_t[top] + res2, self.deconv_output1.weight, 2, k // 2, shape2, prec)
                x_gen = ops.conv_transpose2d_to_size(x_gen + res1, self.deconv_output2.weight, 2, k // 2, shape1, prec)
            else:
                x_gen = ops.conv2d_same(h_t[top], self.conv_last.weight, None, prec)
            next_frames.append(x_gen)
        pred = self._reshape_patch_back(torch.stack(next_frames[-pred_frames:], dim=1))
        if slab is not None:   # mean over the layer-steps of the per-step means = the mean over the slab (equal batch per step)
            loss = ops.decouple_term_batched(slab, self.adapter.weight, self.cell_precision, n_ls, b, deltas)
        else:
            loss = torch.mean(torch.stack(decouple, dim=0))
        return pred, {"ST-LSTM decouple loss": self.decoupling_loss_scale * loss}

    def _weight_banks(self, b, T):
        """ops.STWeightBank per cell for one training forward of T steps (defer_weight_gradients), or None where the library cannot."""
        if not self.defer_weight_gradients or self.action_conditional or self.layer_norm or len(set(self.num_hidden[:self.num_layers])) != 1:
            return None
        # a bank collects the weight gradients of a cell whose every step saves for the backward: with a frozen cell (or a frozen model) the
        # first steps have nothing that requires a gradient and save nothing — the plain per-step path handles that (as the reference does)
        for cell in self.cell_list:
            ws = (cell.conv_x[0].weight, cell.conv_h[0].weight, cell.conv_m[0].weight, cell.conv_o[0].weight, cell.conv_last.weight)
            if not all(w.requires_grad for w in ws):
                return None
        cin = [self.patch_c] + list(self.num_hidden[:self.num_layers - 1])
        geo = [(b, cin[i], self.num_hidden[i], self.rnn_h, self.rnn_w, self.filter_size) for i in range(self.num_layers)]
        if not all(ops.STWeightBank.available(*g, self.cell_precision) for g in geo):
            return None
        px = b * self.rnn_h * self.rnn_w
        slab_bytes = sum(4 * px * (T * 8 * g[2] + 3 * (T + 1) * g[2]) for g in geo) + 4 * px * T * geo[0][1]
        limit = self.BANK_BYTES_LIMIT
        dev = self.adapter.weight.device
        if dev.type == "cuda":
            limit = min(limit, torch.cuda.mem_get_info(dev)[0] // 2 + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev))
        if slab_bytes > limit:
            return None
        return _CellBanks(self, geo, T)

    # ---- patch (un)folding: channel order (p_h, p_w, c)  (predrnn_v2.py:232-250) ----
    def _reshape_patch(self, x):
        b, t, c, h, w = x.shape
        if (self.img_c, self.img_h, self.img_w) != (c, h, w):
            raise ValueError(f"shape mismatch: expected {(self.img_c, self.img_h, self.img_w)}, got {(c, h, w)}")
        p = self.patch_size
        x = x.reshape(b, t, c, self.patch_h, p, self.patch_w, p).permute(0, 1, 4, 6, 2, 3, 5)
        return x.reshape(b, t, -1, self.patch_h, self.patch_w)

    def _reshape_patch_back(self, x_patch):
        b, t, cpp = x_patch.shape[:3]
        p = self.patch_size
        c = cpp // (p * p)
        x = x_patch.reshape(b, t, p, p, c, self.patch_h, self.patch_w).permute(0, 1, 4, 5, 2, 6, 3)
        return x.reshape(b, t, c, self.patch_h * p, self.patch_w * p)

    # ---- sampling schedules (predrnn_v2.py:252-317) ----
    def _flag_tensor(self, batch_size, n, device):
        return torch.zeros(batch_size, n, self.patch_c, self.patch_h, self.patch_w, device=device)

    def _reserve_schedule_sampling(self, batch_size, context_frames, pred_frames):
        itr = self.training_iteration
        if itr < self.r_sampling_step_1:
            r_eta, eta = 0.5, 0.5
        elif itr < self.r_sampling_step_2:
            r_eta = 1.0 - 0.5 * math.exp(-float(itr - self.r_sampling_step_1) / self.r_exp_alpha)
            eta = 0.5 - (0.5 / (self.r_sampling_step_2 - self.r_sampling_step_1)) * (itr - self.r_sampling_step_1)
        else:
            r_eta, eta = 1.0, 0.0
        dev = self._rng_device()
        r_flip = torch.rand(batch_size, context_frames - 1, device=dev)
        flip = torch.rand(batch_size, pred_frames - 1, device=dev)
        r_flag = self._flag_tensor(batch_size, context_frames - 1, dev)
        r_flag[r_flip < r_eta] = 1
        flag = self._flag_tensor(batch_size, pred_frames - 1, dev)
        flag[flip < eta] = 1
        return torch.cat([r_flag, flag], dim=1)

    def _std_schedule_sampling(self, batch_size, context_frames, pred_frames):
        dev = self._rng_device()
        if not self.scheduled_sampling:
            # the reference returns a (0.0, zeros) tuple here (predrnn_v2.py:285-287), which its own forward cannot
            # index; the evident intent — no ground-truth frames mixed in — is the all-zero mask
            return self._flag_tensor(batch_size, pred_frames - 1, dev)
        if self.training_iteration < self.sampling_stop_iter:
            self.sampling_eta -= self.sampling_changing_rate
        else:
            self.sampling_eta = 0.0
        flip = torch.rand(batch_size, pred_frames - 1, device=dev)
        flag = self._flag_tensor(batch_size, pred_frames - 1, dev)
        flag[flip < self.sampling_eta] = 1
        return flag

    def _test_schedule_sampling(self, batch_size, context_frames, pred_frames):
        n = context_frames + pred_frames - 2 if self.reverse_scheduled_sampling else pred_frames - 1
        flag = self._flag_tensor(batch_size, n, self._rng_device())
        if self.reverse_scheduled_sampling:
            flag[:, :context_frames - 1] = 1
        return flag

    def _scheduled_sampling(self, batch_size, context_frames, pred_frames, train):
        if not train:
            return self._test_schedule_sampling(batch_size, context_frames, pred_frames)
        if self.reverse_scheduled_sampling:
            return self._reserve_schedule_sampling(batch_size, context_frames, pred_frames)
        return self._std_schedule_sampling(batch_size, context_frames, pred_frames)

    def _rng_device(self):
        return self.adapter.weight.device

    def training_loss(self, inp, targets, pred_frames, loss_provider, reversed_pair=None, **fwd_kwargs):
        """Loss of ONE training iteration (predrnn_v2.py:326-352): forward with the training-time sampling mask; with
        `reverse_input` the same on the time-reversed sequence, the two losses averaged; bumps `training_iteration`.
        `reversed_pair` = (input, target) of the reversed sequence as `unpack_data(reverse=True)` makes them; by default
        the flip of the complete input sequence (identical whenever the data holds exactly context+pred frames)."""
        fwd_kwargs.pop("train", None)
        if self.reverse_input and reversed_pair is None:
            inp_r = torch.flip(inp, dims=[1])
            reversed_pair = (inp_r, inp_r[:, inp.shape[1] - pred_frames:])
        acts = fwd_kwargs.get("actions")
        acts_ok = acts is None or (torch.is_tensor(acts) and (acts.dim() == 3 or acts.numel() == 0))   # (a layout this cannot duplicate: two passes)
        if self.reverse_input and self.fuse_reversed_pass and acts_ok and reversed_pair[0].shape == inp.shape and \
                reversed_pair[1].shape == targets.shape:
            # One batch of 2B: rows [0, B) the sequence, rows [B, 2B) its reversal. Exact: the samples of a batch only meet in the two
            # batch MEANS of the loss — MSE (mean over b, t of the per-frame sum, base_measure.py:57) and the decoupling term (mean over
            # b, channel, predrnn_v2.py:197-211, then over steps x layers) — and a mean over two halves of equal size is the average of
            # the halves' means, i.e. (total + total_rev) / 2 term by term. The sampling masks are drawn in the reference's order (the
            # forward pass's, then the reversed pass's: same RNG stream, same per-call decrement of sampling_eta); the actions go to
            # both halves unreversed, as the reference passes them (:340-341).
            b, ctx = inp.shape[0], inp.shape[1] - pred_frames
            m1 = self._scheduled_sampling(b, ctx, pred_frames, True)
            m2 = self._scheduled_sampling(b, ctx, pred_frames, True)
            kw = dict(fwd_kwargs)
            if kw.get("actions") is not None and torch.is_tensor(kw["actions"]) and kw["actions"].dim() == 3:
                kw["actions"] = torch.cat([kw["actions"], kw["actions"]], dim=0)
            inp_r, targets_r = reversed_pair
            preds, ml = self(torch.cat([inp, inp_r], dim=0), pred_frames=pred_frames, train=True, _mask_true=torch.cat([m1, m2], dim=0), **kw)
            total = self._total_loss(preds, torch.cat([targets, targets_r], dim=0), ml, loss_provider)
            self.training_iteration += 1
            return total
        preds, ml = self(inp, pred_frames=pred_frames, train=True, **fwd_kwargs)
        total = self._total_loss(preds, targets, ml, loss_provider)
        if self.reverse_input:
            inp_r, targets_r = reversed_pair
            preds_r, ml_r = self(inp_r, pred_frames=pred_frames, train=True, **fwd_kwargs)
            total = (total + self._total_loss(preds_r, targets_r, ml_r, loss_provider)) / 2
        self.training_iteration += 1
        return total

    def train_iter(self, config, loader, optimizer, loss_provider, epoch):
        """forward on the sequence and on its time-reversal, losses averaged, one optimizer step; counts training
        iterations for the sampling schedule (predrnn_v2.py:319-365)."""
        loop = _progress(loader)
        for data in loop:
            inp, targets, actions = self.unpack_data(data, config)
            rev = self.unpack_data(data, config, reverse=True)[:2] if self.reverse_input else None
            total = self.training_loss(inp, targets, config["pred_frames"], loss_provider, actions=actions, reversed_pair=rev)
            optimizer.zero_grad()
            total.backward()
            optimizer.step()
            if hasattr(loop, "set_postfix"):
                loop.set_postfix(loss=total.item())
