/*
 * vpx_oracle.c — CPU restatement (plain C) of the reference's ConvLSTM / ST-LSTM recurrence.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under oracle/ is imported, linked or executed by the product path
 * (vp-suite_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only as the
 * checker. Parity status: PINNED against golden vectors produced by the real reference imported in the build
 * container (tools/gen_golden.py -> tests/golden/ npz files, checked by tests/test_oracle.py).
 *
 * Layout everywhere: the reference's own NCHW, fp32 storage. Convolution sums are accumulated in double and rounded
 * once, so the oracle is the "exact fp32" answer and deviates from the reference (oneDNN fp32) only by the
 * reference's own rounding noise (~1e-7).
 *
 * Functions and the reference lines they restate:
 *   orc_conv2d                 torch.nn.functional.conv2d as called at conv_lstm_hzzone.py:60, conv_lstm_ndrplz.py:33,
 *                              predrnn.py:58-60,80-81 (cross-correlation, zero padding, stride)
 *   orc_convlstm_seq_fwd/bwd   vp_suite/model_blocks/conv_lstm_hzzone.py:38-70 (gate order i,f,g,o + peepholes)
 *                              vp_suite/model_blocks/conv_lstm_ndrplz.py:28-43 (gate order i,f,o,g, no peephole)
 *   orc_stlstm_step_fwd        vp_suite/model_blocks/predrnn.py:57-83 (+ LayerNorm variant :24-40)
 *   orc_decouple_fwd           vp_suite/models/predrnn_v2.py:197-198,209-211 (adapter 1x1, normalize, |cos|, mean)
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_GATE_IFGO 0 /* hzzone: chunk order (i, f, g, o)   conv_lstm_hzzone.py:62 */
#define ORC_GATE_IFOG 1 /* ndrplz: split order (i, f, o, g)   conv_lstm_ndrplz.py:34 */

static inline float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

/* y[B,Co,Ho,Wo] (+)= conv2d(x[B,Ci,H,W], w[Co,Ci,kh,kw]) + bias; cross-correlation like torch. */
void orc_conv2d(const float* x, const float* w, const float* bias, float* y, int B, int Ci, int H, int W, int Co,
                int kh, int kw, int stride, int ph, int pw, int accumulate) {
    const int Ho = (H + 2 * ph - kh) / stride + 1;
    const int Wo = (W + 2 * pw - kw) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Co; ++co) {
            for (int oy = 0; oy < Ho; ++oy)
                for (int ox = 0; ox < Wo; ++ox) {
                    double acc = bias ? (double)bias[co] : 0.0;
                    for (int ci = 0; ci < Ci; ++ci) {
                        const float* xp = x + ((size_t)b * Ci + ci) * H * W;
                        const float* wp = w + ((size_t)co * Ci + ci) * kh * kw;
                        for (int ky = 0; ky < kh; ++ky) {
                            const int iy = oy * stride - ph + ky;
                            if (iy < 0 || iy >= H) continue;
                            for (int kx = 0; kx < kw; ++kx) {
                                const int ix = ox * stride - pw + kx;
                                if (ix < 0 || ix >= W) continue;
                                acc += (double)xp[(size_t)iy * W + ix] * (double)wp[ky * kw + kx];
                            }
                        }
                    }
                    float* yp = y + (((size_t)b * Co + co) * Ho + oy) * Wo + ox;
                    *yp = accumulate ? (float)((double)*yp + acc) : (float)acc;
                }
        }
}

/* dx[B,Ci,H,W] = conv2d_backward_input(dy[B,Co,H,W], w) for stride 1 "same" convs (the only kind in the cells). */
static void conv2d_bwd_input_same(const float* dy, const float* w, float* dx, int B, int Ci, int H, int W, int Co,
                                  int kh, int kw) {
    const int ph = kh / 2, pw = kw / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int ci = 0; ci < Ci; ++ci)
            for (int iy = 0; iy < H; ++iy)
                for (int ix = 0; ix < W; ++ix) {
                    double acc = 0.0;
                    for (int co = 0; co < Co; ++co) {
                        const float* dyp = dy + ((size_t)b * Co + co) * H * W;
                        const float* wp = w + ((size_t)co * Ci + ci) * kh * kw;
                        for (int ky = 0; ky < kh; ++ky) {
                            const int oy = iy + ph - ky;
                            if (oy < 0 || oy >= H) continue;
                            for (int kx = 0; kx < kw; ++kx) {
                                const int ox = ix + pw - kx;
                                if (ox < 0 || ox >= W) continue;
                                acc += (double)dyp[(size_t)oy * W + ox] * (double)wp[ky * kw + kx];
                            }
                        }
                    }
                    dx[(((size_t)b * Ci + ci) * H + iy) * W + ix] = (float)acc;
                }
}

/* dw[Co,Ci,kh,kw] += sum_{b,y,x} dy[b,co,y,x] * x[b,ci,y+ky-ph,x+kx-pw]  (stride 1, same) */
static void conv2d_bwd_weight_same(const float* dy, const float* x, double* dw, int B, int Ci, int H, int W, int Co,
                                   int kh, int kw) {
    const int ph = kh / 2, pw = kw / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int co = 0; co < Co; ++co)
        for (int ci = 0; ci < Ci; ++ci)
            for (int ky = 0; ky < kh; ++ky)
                for (int kx = 0; kx < kw; ++kx) {
                    double acc = 0.0;
                    for (int b = 0; b < B; ++b) {
                        const float* dyp = dy + ((size_t)b * Co + co) * H * W;
                        const float* xp = x + ((size_t)b * Ci + ci) * H * W;
                        for (int oy = 0; oy < H; ++oy) {
                            const int iy = oy - ph + ky;
                            if (iy < 0 || iy >= H) continue;
                            for (int ox = 0; ox < W; ++ox) {
                                const int ix = ox - pw + kx;
                                if (ix < 0 || ix >= W) continue;
                                acc += (double)dyp[(size_t)oy * W + ox] * (double)xp[(size_t)iy * W + ix];
                            }
                        }
                    }
                    dw[(((size_t)co * Ci + ci) * kh + ky) * kw + kx] += acc;
                }
}

/* gather channel block g (of 4) of a [B,4Ch,H,W] tensor index */
#define GIDX(b, g, ch, p) ((((size_t)(b) * 4 * Ch + (size_t)(g) * Ch + (ch)) * HW) + (p))

/*
 * ConvLSTM over a sequence. x: [B,T,Cin,H,W] or NULL (zeros: conv_lstm_hzzone.py:54-56); h0/c0: [B,Ch,H,W] or NULL (zeros: :40-45)
 * Wt: [4Ch, Cin+Ch, kh, kw]; bias [4Ch] or NULL; Wci/Wcf/Wco [Ch,H,W] or NULL (no peephole = ndrplz).
 * out: [B,T,Ch,H,W]; hT,cT: [B,Ch,H,W].
 * reserve (optional, for bwd): gates [T][B,4Ch,H,W] post-activation in storage order (i,f,g,o) + cs [T][B,Ch,H,W].
 */
int orc_convlstm_seq_fwd(const float* x, const float* h0, const float* c0, const float* Wt, const float* bias,
                         const float* Wci, const float* Wcf, const float* Wco, float* out, float* hT, float* cT,
                         float* gates_save, float* c_save, int B, int T, int Cin, int Ch, int H, int W, int kh, int kw,
                         int gate_order) {
    const int HW = H * W, Ct = Cin + Ch;
    const size_t nst = (size_t)B * Ch * HW;
    float* cat = (float*)calloc((size_t)B * Ct * HW, sizeof(float));
    float* pre = (float*)malloc((size_t)B * 4 * Ch * HW * sizeof(float));
    float* h = (float*)calloc(nst, sizeof(float));
    float* c = (float*)calloc(nst, sizeof(float));
    if (!cat || !pre || !h || !c) return -1;
    if (h0) memcpy(h, h0, nst * sizeof(float));
    if (c0) memcpy(c, c0, nst * sizeof(float));
    /* position of logical gates (i,f,g,o) inside the 4Ch conv output */
    const int gi = 0, gf = 1, gg = gate_order == ORC_GATE_IFGO ? 2 : 3, go = gate_order == ORC_GATE_IFGO ? 3 : 2;
    for (int t = 0; t < T; ++t) {
        for (int b = 0; b < B; ++b) { /* cat([x_t, h], dim=1)   conv_lstm_hzzone.py:59 */
            if (x)
                memcpy(cat + (size_t)b * Ct * HW, x + ((size_t)b * T + t) * Cin * HW, (size_t)Cin * HW * sizeof(float));
            memcpy(cat + ((size_t)b * Ct + Cin) * HW, h + (size_t)b * Ch * HW, (size_t)Ch * HW * sizeof(float));
        }
        orc_conv2d(cat, Wt, bias, pre, B, Ct, H, W, 4 * Ch, kh, kw, 1, kh / 2, kw / 2, 0);
#pragma omp parallel for collapse(2) schedule(static)
        for (int b = 0; b < B; ++b)
            for (int ch = 0; ch < Ch; ++ch)
                for (int p = 0; p < HW; ++p) {
                    const size_t s = ((size_t)b * Ch + ch) * HW + p;
                    const size_t pp = (size_t)ch * HW + p;
                    const float cp = c[s];
                    float ai = pre[GIDX(b, gi, ch, p)], af = pre[GIDX(b, gf, ch, p)];
                    float ag = pre[GIDX(b, gg, ch, p)], ao = pre[GIDX(b, go, ch, p)];
                    if (Wci) ai += Wci[pp] * cp; /* :64 */
                    if (Wcf) af += Wcf[pp] * cp; /* :65 */
                    const float i_ = sigmoidf_(ai), f_ = sigmoidf_(af), g_ = tanhf(ag);
                    const float cn = f_ * cp + i_ * g_; /* :66 */
                    if (Wco) ao += Wco[pp] * cn;        /* :67 peephole on the NEW cell state */
                    const float o_ = sigmoidf_(ao);
                    const float hn = o_ * tanhf(cn); /* :68 */
                    c[s] = cn;
                    h[s] = hn;
                    out[(((size_t)b * T + t) * Ch + ch) * HW + p] = hn;
                    if (gates_save) {
                        float* gs = gates_save + (size_t)t * B * 4 * Ch * HW;
                        gs[GIDX(b, 0, ch, p)] = i_;
                        gs[GIDX(b, 1, ch, p)] = f_;
                        gs[GIDX(b, 2, ch, p)] = g_;
                        gs[GIDX(b, 3, ch, p)] = o_;
                    }
                    if (c_save) c_save[(size_t)t * nst + s] = cn;
                }
    }
    if (hT) memcpy(hT, h, nst * sizeof(float));
    if (cT) memcpy(cT, c, nst * sizeof(float));
    free(cat); free(pre); free(h); free(c);
    return 0;
}

/*
 * BPTT through orc_convlstm_seq_fwd. Needs gates_save/c_save from the forward.
 * Inputs: dout [B,T,Ch,H,W], dhT, dcT [B,Ch,H,W] (any may be NULL = zeros).
 * Outputs (any may be NULL): dx [B,T,Cin,H,W], dh0, dc0, dW [4Ch,Ct,kh,kw], db [4Ch], dWci/dWcf/dWco [Ch,H,W].
 */
int orc_convlstm_seq_bwd(const float* x, const float* h0, const float* c0, const float* Wt, const float* Wci,
                         const float* Wcf, const float* Wco, const float* out, const float* gates_save,
                         const float* c_save, const float* dout, const float* dhT, const float* dcT, float* dx,
                         float* dh0, float* dc0, float* dW, float* db, float* dWci, float* dWcf, float* dWco, int B,
                         int T, int Cin, int Ch, int H, int W, int kh, int kw, int gate_order) {
    const int HW = H * W, Ct = Cin + Ch;
    const size_t nst = (size_t)B * Ch * HW, ng = (size_t)B * 4 * Ch * HW;
    const size_t nw = (size_t)4 * Ch * Ct * kh * kw;
    float* dh = (float*)calloc(nst, sizeof(float));
    float* dc = (float*)calloc(nst, sizeof(float));
    float* dpre = (float*)malloc(ng * sizeof(float));
    float* dcat = (float*)malloc((size_t)B * Ct * HW * sizeof(float));
    float* cat = (float*)calloc((size_t)B * Ct * HW, sizeof(float));
    double* dWacc = (double*)calloc(nw, sizeof(double));
    double* dbacc = (double*)calloc((size_t)4 * Ch, sizeof(double));
    double* dpi = (double*)calloc((size_t)Ch * HW, sizeof(double));
    double* dpf = (double*)calloc((size_t)Ch * HW, sizeof(double));
    double* dpo = (double*)calloc((size_t)Ch * HW, sizeof(double));
    if (!dh || !dc || !dpre || !dcat || !cat || !dWacc || !dbacc || !dpi || !dpf || !dpo) return -1;
    if (dhT) memcpy(dh, dhT, nst * sizeof(float));
    if (dcT) memcpy(dc, dcT, nst * sizeof(float));
    const int gi = 0, gf = 1, gg = gate_order == ORC_GATE_IFGO ? 2 : 3, go = gate_order == ORC_GATE_IFGO ? 3 : 2;
    for (int t = T - 1; t >= 0; --t) {
        const float* gs = gates_save + (size_t)t * ng;
        const float* cn_ = c_save + (size_t)t * nst;
        for (int b = 0; b < B; ++b) /* serial over b: peephole grads reduce over the batch */
#pragma omp parallel for schedule(static)
            for (int ch = 0; ch < Ch; ++ch)
                for (int p = 0; p < HW; ++p) {
                    const size_t s = ((size_t)b * Ch + ch) * HW + p;
                    const size_t pp = (size_t)ch * HW + p;
                    const float i_ = gs[GIDX(b, 0, ch, p)], f_ = gs[GIDX(b, 1, ch, p)];
                    const float g_ = gs[GIDX(b, 2, ch, p)], o_ = gs[GIDX(b, 3, ch, p)];
                    const float cn = cn_[s];
                    const float cp = t > 0 ? c_save[(size_t)(t - 1) * nst + s] : (c0 ? c0[s] : 0.0f);
                    const float dht = dh[s] + (dout ? dout[(((size_t)b * T + t) * Ch + ch) * HW + p] : 0.0f);
                    const float tc = tanhf(cn);
                    const float dao = dht * tc * o_ * (1.0f - o_);
                    float dcn = dc[s] + dht * o_ * (1.0f - tc * tc);
                    if (Wco) { dcn += dao * Wco[pp]; dpo[pp] += (double)dao * cn; }
                    const float dai = dcn * g_ * i_ * (1.0f - i_);
                    const float daf = dcn * cp * f_ * (1.0f - f_);
                    const float dag = dcn * i_ * (1.0f - g_ * g_);
                    float dcp = dcn * f_;
                    if (Wci) { dcp += dai * Wci[pp]; dpi[pp] += (double)dai * cp; }
                    if (Wcf) { dcp += daf * Wcf[pp]; dpf[pp] += (double)daf * cp; }
                    dc[s] = dcp;
                    dpre[GIDX(b, gi, ch, p)] = dai;
                    dpre[GIDX(b, gf, ch, p)] = daf;
                    dpre[GIDX(b, gg, ch, p)] = dag;
                    dpre[GIDX(b, go, ch, p)] = dao;
                }
        /* rebuild cat = [x_t, h_{t-1}] */
        for (int b = 0; b < B; ++b) {
            if (x)
                memcpy(cat + (size_t)b * Ct * HW, x + ((size_t)b * T + t) * Cin * HW, (size_t)Cin * HW * sizeof(float));
            float* hd = cat + ((size_t)b * Ct + Cin) * HW;
            if (t > 0)
                memcpy(hd, out + ((size_t)b * T + (t - 1)) * Ch * HW, (size_t)Ch * HW * sizeof(float));
            else if (h0)
                memcpy(hd, h0 + (size_t)b * Ch * HW, (size_t)Ch * HW * sizeof(float));
            else
                memset(hd, 0, (size_t)Ch * HW * sizeof(float));
        }
        conv2d_bwd_weight_same(dpre, cat, dWacc, B, Ct, H, W, 4 * Ch, kh, kw);
        for (int b = 0; b < B; ++b)
            for (int n = 0; n < 4 * Ch; ++n) {
                double a = 0.0;
                const float* q = dpre + ((size_t)b * 4 * Ch + n) * HW;
                for (int p = 0; p < HW; ++p) a += q[p];
                dbacc[n] += a;
            }
        conv2d_bwd_input_same(dpre, Wt, dcat, B, Ct, H, W, 4 * Ch, kh, kw);
        for (int b = 0; b < B; ++b) {
            if (dx && x)
                memcpy(dx + ((size_t)b * T + t) * Cin * HW, dcat + (size_t)b * Ct * HW, (size_t)Cin * HW * sizeof(float));
            memcpy(dh + (size_t)b * Ch * HW, dcat + ((size_t)b * Ct + Cin) * HW, (size_t)Ch * HW * sizeof(float));
        }
    }
    if (dh0) memcpy(dh0, dh, nst * sizeof(float));
    if (dc0) memcpy(dc0, dc, nst * sizeof(float));
    if (dW) for (size_t i = 0; i < nw; ++i) dW[i] = (float)dWacc[i];
    if (db) for (int i = 0; i < 4 * Ch; ++i) db[i] = (float)dbacc[i];
    for (size_t i = 0; i < (size_t)Ch * HW; ++i) {
        if (dWci) dWci[i] = (float)dpi[i];
        if (dWcf) dWcf[i] = (float)dpf[i];
        if (dWco) dWco[i] = (float)dpo[i];
    }
    free(dh); free(dc); free(dpre); free(dcat); free(cat); free(dWacc); free(dbacc); free(dpi); free(dpf); free(dpo);
    return 0;
}

/* LayerNorm over [C,H,W] per sample, eps 1e-5, affine gamma/beta [C,H,W]   (predrnn.py:27,31,35,39) */
static void layer_norm_chw(float* y, int B, size_t n, const float* gamma, const float* beta) {
    for (int b = 0; b < B; ++b) {
        float* p = y + (size_t)b * n;
        double mean = 0.0, var = 0.0;
        for (size_t i = 0; i < n; ++i) mean += p[i];
        mean /= (double)n;
        for (size_t i = 0; i < n; ++i) { const double d = p[i] - mean; var += d * d; }
        var /= (double)n;
        const double rstd = 1.0 / sqrt(var + 1e-5);
        for (size_t i = 0; i < n; ++i) p[i] = (float)(((double)p[i] - mean) * rstd * gamma[i] + beta[i]);
    }
}

/*
 * One ST-LSTM step (predrnn.py:57-83). x [B,Cin,H,W]; h,c,m [B,Ch,H,W]; Wx [7Ch,Cin,k,k]; Wh [4Ch,Ch,k,k]; Wm [3Ch,Ch,k,k];
 * Wo [Ch,2Ch,k,k]; Wlast [Ch,2Ch,1,1]; ln_*: NULL or LayerNorm gamma/beta pairs for conv_x/h/m/o outputs.
 * Outputs h_new,c_new,m_new,delta_c,delta_m [B,Ch,H,W].
 */
int orc_stlstm_step_fwd(const float* x, const float* h, const float* c, const float* m, const float* Wx,
                        const float* Wh, const float* Wm, const float* Wo, const float* Wlast, const float* ln_x_g,
                        const float* ln_x_b, const float* ln_h_g, const float* ln_h_b, const float* ln_m_g,
                        const float* ln_m_b, const float* ln_o_g, const float* ln_o_b, float* h_new, float* c_new,
                        float* m_new, float* delta_c, float* delta_m, int B, int Cin, int Ch, int H, int W, int k) {
    const int HW = H * W, pd = k / 2;
    const float forget_bias = 1.0f; /* predrnn.py:23 */
    float* xc = (float*)malloc((size_t)B * 7 * Ch * HW * sizeof(float));
    float* hc = (float*)malloc((size_t)B * 4 * Ch * HW * sizeof(float));
    float* mc = (float*)malloc((size_t)B * 3 * Ch * HW * sizeof(float));
    float* mem = (float*)malloc((size_t)B * 2 * Ch * HW * sizeof(float));
    float* oc = (float*)malloc((size_t)B * Ch * HW * sizeof(float));
    float* lc = (float*)malloc((size_t)B * Ch * HW * sizeof(float));
    if (!xc || !hc || !mc || !mem || !oc || !lc) return -1;
    orc_conv2d(x, Wx, NULL, xc, B, Cin, H, W, 7 * Ch, k, k, 1, pd, pd, 0); /* :58 */
    orc_conv2d(h, Wh, NULL, hc, B, Ch, H, W, 4 * Ch, k, k, 1, pd, pd, 0);  /* :59 */
    orc_conv2d(m, Wm, NULL, mc, B, Ch, H, W, 3 * Ch, k, k, 1, pd, pd, 0);  /* :60 */
    if (ln_x_g) {
        layer_norm_chw(xc, B, (size_t)7 * Ch * HW, ln_x_g, ln_x_b);
        layer_norm_chw(hc, B, (size_t)4 * Ch * HW, ln_h_g, ln_h_b);
        layer_norm_chw(mc, B, (size_t)3 * Ch * HW, ln_m_g, ln_m_b);
    }
#define XI(b, g, ch, p) ((((size_t)(b) * 7 * Ch + (size_t)(g) * Ch + (ch)) * HW) + (p))
#define HI(b, g, ch, p) ((((size_t)(b) * 4 * Ch + (size_t)(g) * Ch + (ch)) * HW) + (p))
#define MI(b, g, ch, p) ((((size_t)(b) * 3 * Ch + (size_t)(g) * Ch + (ch)) * HW) + (p))
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int ch = 0; ch < Ch; ++ch)
            for (int p = 0; p < HW; ++p) {
                const size_t s = ((size_t)b * Ch + ch) * HW + p;
                /* x split (i,f,g,i',f',g',o) :61 ; h split (i,f,g,o) :62 ; m split (i,f,g) :63 */
                const float it = sigmoidf_(xc[XI(b, 0, ch, p)] + hc[HI(b, 0, ch, p)]);               /* :65 */
                const float ft = sigmoidf_(xc[XI(b, 1, ch, p)] + hc[HI(b, 1, ch, p)] + forget_bias); /* :66 */
                const float gt = tanhf(xc[XI(b, 2, ch, p)] + hc[HI(b, 2, ch, p)]);                   /* :67 */
                const float dcv = it * gt;                                                           /* :69 */
                const float cn = ft * c[s] + dcv;                                                    /* :70 */
                const float ip = sigmoidf_(xc[XI(b, 3, ch, p)] + mc[MI(b, 0, ch, p)]);               /* :72 */
                const float fp = sigmoidf_(xc[XI(b, 4, ch, p)] + mc[MI(b, 1, ch, p)] + forget_bias); /* :73 */
                const float gp = tanhf(xc[XI(b, 5, ch, p)] + mc[MI(b, 2, ch, p)]);                   /* :74 */
                const float dmv = ip * gp;                                                           /* :76 */
                const float mn = fp * m[s] + dmv;                                                    /* :77 */
                c_new[s] = cn; m_new[s] = mn; delta_c[s] = dcv; delta_m[s] = dmv;
                mem[(((size_t)b * 2 * Ch + ch) * HW) + p] = cn;        /* cat((c_new, m_new), 1) :79 */
                mem[(((size_t)b * 2 * Ch + Ch + ch) * HW) + p] = mn;
            }
    orc_conv2d(mem, Wo, NULL, oc, B, 2 * Ch, H, W, Ch, k, k, 1, pd, pd, 0); /* conv_o(mem) :80 */
    if (ln_o_g) layer_norm_chw(oc, B, (size_t)Ch * HW, ln_o_g, ln_o_b);
    orc_conv2d(mem, Wlast, NULL, lc, B, 2 * Ch, H, W, Ch, 1, 1, 1, 0, 0, 0); /* conv_last(mem) :81 */
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int ch = 0; ch < Ch; ++ch)
            for (int p = 0; p < HW; ++p) {
                const size_t s = ((size_t)b * Ch + ch) * HW + p;
                const float ot = sigmoidf_(xc[XI(b, 6, ch, p)] + hc[HI(b, 3, ch, p)] + oc[s]); /* :80 */
                h_new[s] = ot * tanhf(lc[s]);                                                  /* :81 */
            }
    free(xc); free(hc); free(mc); free(mem); free(oc); free(lc);
    return 0;
}

/*
 * Decoupling-loss term for one (layer, step): mean_{b,ch} | cos( normalize(A*dc), normalize(A*dm) ) | over the H*W axis.
 * adapter [Ch,Ch] (1x1 conv, no bias). predrnn_v2.py:197-198 (adapter + F.normalize(dim=2), eps 1e-12),
 * :210-211 (cosine_similarity(dim=2), eps 1e-8, abs, mean).
 */
double orc_decouple_fwd(const float* delta_c, const float* delta_m, const float* adapter, int B, int Ch, int HW) {
    double total = 0.0;
    double* a = (double*)malloc((size_t)HW * sizeof(double));
    double* bb = (double*)malloc((size_t)HW * sizeof(double));
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Ch; ++co) {
            double na = 0.0, nb = 0.0;
            for (int p = 0; p < HW; ++p) {
                double sa = 0.0, sb = 0.0;
                for (int ci = 0; ci < Ch; ++ci) {
                    const double w = adapter[(size_t)co * Ch + ci];
                    sa += w * delta_c[((size_t)b * Ch + ci) * HW + p];
                    sb += w * delta_m[((size_t)b * Ch + ci) * HW + p];
                }
                a[p] = (float)sa; bb[p] = (float)sb;
                na += a[p] * a[p]; nb += bb[p] * bb[p];
            }
            na = fmax(sqrt(na), 1e-12); nb = fmax(sqrt(nb), 1e-12);
            double dot = 0.0, n2a = 0.0, n2b = 0.0;
            for (int p = 0; p < HW; ++p) {
                const double u = a[p] / na, v = bb[p] / nb;
                dot += u * v; n2a += u * u; n2b += v * v;
            }
            /* torch.cosine_similarity: x.y / max(|x|*|y|, eps)-style clamping; |u|=|v|=1 here unless degenerate */
            const double den = fmax(sqrt(n2a) * sqrt(n2b), 1e-8);
            total += fabs(dot / den);
        }
    free(a); free(bb);
    return total / ((double)B * Ch);
}

int orc_version(void) { return 1; }
