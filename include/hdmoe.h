/* libhdmoe_hip.so -- C ABI of the MI355X (gfx950) kernels behind the HDMOEM denoising hot path.
 *
 * The reference (cs2mosa/Heterogeneous-MOE-for-Diffusion-models) is pure PyTorch: it has no FFI / operator
 * registry, its "interface" for this path is the ATen call sequence inside models/model_internals.py,
 * models/model_components.py and models/model_config{1,2}.py.  Each entry point below replaces one such
 * sequence (cited as reference file:line); the Python nn.Module mirror in
 * heterogeneous-moe-for-diffusion-models_amd/models/ binds them with ctypes.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is DEVICE memory unless the comment says "host array".
 *   - activations are NHWC / [rows][C] contiguous; `dtype` (HDMOE_F32 | HDMOE_BF16) is their element type.
 *     Statistics, per-sample scalars, (B,F) embeddings, parameters and parameter gradients are fp32.
 *   - `stream` is a hipStream_t (void*); kernels are enqueued on it, nothing syncs, allocates or frees,
 *     so every call is hipGraph-capturable.  Scratch / accumulators are caller-provided.
 *   - return 0 on success, HDMOE_EINVAL / HDMOE_EDTYPE / HDMOE_ELAUNCH (<0) otherwise; never throws.
 *   - "grouped": rows of one launch belong to up to HDMOE_MAX_GROUPS experts; seg[g]..seg[g+1] (device int32)
 *     is the row range of expert g; per-group kernel sizes come as host arrays of length ngroups.
 */
#ifndef HDMOE_H
#define HDMOE_H
#ifdef __cplusplus
extern "C" {
#endif

#ifndef HDMOE_OK
#define HDMOE_OK 0
#define HDMOE_EINVAL (-1)
#define HDMOE_EDTYPE (-2)
#define HDMOE_ELAUNCH (-3)
#define HDMOE_F32 0
#define HDMOE_BF16 1
#define HDMOE_F16 3  /* hdmoe_cast only: fp16 tensors at the module boundary (`.half()` callers) are converted at ingest / egress */
#define HDMOE_F32S 2 /* fp32 activations computed as split bf16 (hi + lo, three MFMAs per product): weight images = bf16 [hi | lo] planes */
#define HDMOE_MAX_GROUPS 8
#endif

typedef void* hdmoe_stream_t; /* hipStream_t */
#ifdef __HIPCC__
#define HS hipStream_t
#else
#define HS hdmoe_stream_t
#endif

int hdmoe_version(void);

/* ---- K3: MP_Conv = weight prep + implicit-GEMM conv  (model_internals.py:253-275) ------------------------ */
/* w_raw/gain_ptr/kh/kw: host arrays [ngroups].  wf: [g][tap][O][Ipad]; wd (optional): [g][tap'][I][Opad]
 * (tap' flipped when flip=1: the dgrad operand).  normalize=0: plain nn.Conv2d weight (Vit_expert.patch).
 * mutate=1: training-mode forced weight normalisation, written back into w_raw (:254-256). */
int hdmoe_wprep_fwd(float* const* w_raw, const float* const* gain_ptr, float gain_val, const int* kh, const int* kw,
                    int ngroups, int O, int I, int Ipad, int Opad, void* wf, long wf_stride, void* wd, long wd_stride,
                    int normalize, int mutate, int flip, int dtype, HS stream);
/* G: per-group [tap][O][I] fp32 gradient w.r.t. the effective weight (from hdmoe_conv_wgrad);
 * dw: per-group (O,I,kh,kw) fp32; dgain: per-group scalar accumulators or NULL. */
int hdmoe_wprep_bwd(const float* const* w_raw, const float* const* gain_ptr, float gain_val, const float* const* G,
                    float* const* dw, float* const* dgain, const int* kh, const int* kw, int ngroups, int O, int I,
                    int normalize, HS stream);
/* y = alpha*conv(x, w) + beta*res.  x [N][H][W][Cphys]; logical Cin = Cphys (+1 implicit all-ones channel when
 * ones=1: Unet_expert's torch.cat([x, ones]), model_components.py:416); y [N][Ho][Wo][Cstore], Cstore <= Cout.
 * Also runs dgrad (w = wd, pads flipped) and every F.linear (H = W = 1 or N = 1). */
int hdmoe_conv_fwd(const void* x, const void* w, void* y, const void* res, float alpha, float beta, const int* seg,
                   int ngroups, long wstride, int N, int H, int W, int Ho, int Wo, int Cin, int Cphys, int Ipad, int Cout,
                   int Cstore, int stride, int ones, const int* kh, const int* kw, const int* pt, const int* pl, int dtype,
                   HS stream);
/* G[g] ([tap][Cout][Cin] fp32, pre-zeroed) += dy^T * shifted(x); G: host array of device pointers. */
int hdmoe_conv_wgrad(const void* x, const void* dy, float* const* G, const int* seg, int ngroups, int N, int H, int W,
                     int Ho, int Wo, int Cin, int Cphys, int Cout, int stride, int ones, const int* kh, const int* kw,
                     const int* pt, const int* pl, int dtype, HS stream);

/* Atomic-free weight gradient for the k x k (k = 3, 5) bf16 layers (csrc/wgrad6.hip): same contract as hdmoe_conv_wgrad plus a
 * caller-provided workspace of hdmoe_conv_wgrad6_ws_kib(...) KiB (0 = shape outside the kernel's domain).  hdmoe_conv_wgrad6
 * returns 1 without launching when it does not apply; the caller then uses hdmoe_conv_wgrad. */
int hdmoe_conv_wgrad6_ws_kib(int ngroups, int N, int H, int W, int Cin, int Cout, const int* kh, const int* kw, int dtype);
int hdmoe_conv_wgrad6(const void* x, const void* dy, float* const* G, const int* seg, int ngroups, int N, int H, int W, int Cin,
                      int Cout, const int* kh, const int* kw, const int* pt, const int* pl, void* ws, long ws_bytes, int dtype,
                      int defer, HS stream);
/* defer != 0: the partial slabs stay in `ws` (2 x ws_kib KiB: one region per kernel-size class) and are summed into G later by ONE
 * batched launch per 16 (layer, class) items.  Per deferred call i: G + 8 i = its per-expert slabs, seg[i], ws[i], and
 * dims + 16 i = {ngroups, N, H, W, Cin, Cout, dtype, 0, kh[0..7]}. */
/* Input gradient and (deferred) weight gradient of one grouped k x k bf16 layer with 3x3 and 5x5 experts in ONE launch (csrc/bwd6.hip):
 * dx = alpha * dgrad(dy, wd) like hdmoe_conv_fwd on the flipped image, partial dW slabs into ws like hdmoe_conv_wgrad6(defer = 1).
 * Returns 1 without launching when the layer is outside the domain. */
int hdmoe_conv_bwd6(const void* x, const void* dy, const void* wd, void* dx, float* const* G, const int* seg, int ngroups,
                    long wd_stride, int N, int H, int W, int Cin, int Cout, const int* kh, const int* kw, const int* pt,
                    const int* pl, float alpha, void* ws, long ws_bytes, int dtype, HS stream);
int hdmoe_conv_bwd6s(const void* x, const void* dy, const void* wd, void* dx, float* const* G, const int* seg, int ngroups,
                     long wd_stride, long wd_plane, int N, int H, int W, int Cin, int Cout, const int* kh, const int* kw,
                     const int* pt, const int* pl, float alpha, void* ws, long ws_bytes, const float* in_scale, const float* in_shift,
                     int in_relu, int hi_only, HS stream);   /* fp32 tensors, split bf16 (3x3); in_scale / in_shift (or NULL): x is relu(x * scale[n][c] + shift[n][c]);
                                                 hi_only = 1: bf16 operands (the hi halves) with fp32 accumulation -- one MFMA per product instead of three */
/* GroupNorm(1, C) + ReLU of the router trunks fused into the neighbouring convs (model_components.py:100-112): the conv writes per-sample
 * partial statistics of its output, hdmoe_gn1_finalize turns them into mean / rstd and a per-(sample, channel) scale / shift, the NEXT
 * conv (and its weight gradient, hdmoe_conv_bwd6s) apply relu(x * scale + shift) while staging x, hdmoe_gn1_relu_mean is the last
 * GroupNorm + ReLU + AdaptiveAvgPool2d(1). */
int hdmoe_conv_fwd_split_gn(const void* x, const void* w, void* y, const float* in_scale, const float* in_shift, int in_relu,
                            float* stats_ws, long wstride, long wplane, int N, int H, int W, int Cin, int Cout, float alpha, HS stream);
/* k x k bf16 expert conv with the FiLM of Unet_block (mp_silu(y * emb) + dropout, model_components.py:242-246) as a second output of its
 * epilogue; 1 = outside the fused kernel's domain, nothing launched. */
int hdmoe_conv_fwd_film(const void* x, const void* w, void* y, void* h, const float* e, unsigned long long seed,
                        const unsigned long long* seed_dev, float p, float alpha, const int* seg, int ngroups, long wstride, int N, int H, int W,
                        int Cin, int Cout, const int* kh, const int* kw, const int* pt, const int* pl, int dtype, HS stream);
int hdmoe_conv_split_stats_slots(int H, int W, int Cout);   /* partial-statistics slots per sample of hdmoe_conv_fwd_split_gn (0: outside its domain) */
int hdmoe_gn1_finalize(float* scale, float* shift, float* mean, float* rstd, const float* ws, const float* gamma, const float* beta,
                       int N, int slots, int C, long count, float eps, HS stream);
int hdmoe_gn1_relu_mean(float* out, const float* y, const float* scale, const float* shift, int N, long S, int C, HS stream);
int hdmoe_gn1_finalize_relu_mean(float* out, float* scale, float* shift, float* mean, float* rstd, const float* y, const float* ws,
                                 const float* gamma, const float* beta, int N, int slots, long S, int C, float eps, HS stream);   /* the two above in one launch */
int hdmoe_conv_wgrad6_reduce_batch(float* const* G, const int* const* seg, float* const* ws, const int* dims, int n, HS stream);

/* development hook of the conv6 kernels: `buf` = device array of 8 x 64 uint64 receiving workgroup 0's in-kernel clock stamps
 * (tag << 56 | s_memtime) of every later launch; NULL switches it off (tools/conv6_check.py --stamps). */
int hdmoe_conv6_debug_stamps(void* buf);

/* Multi-tensor weight bank: one prep launch per forward and one gradient-finish launch per backward for ALL MP_Conv weights
 * of a model.  descs: device array of descriptors (layout = hdmoe_wbank_desc_bytes() bytes each, see csrc/wbank.hip);
 * rows: device int32 pairs (descriptor index, output-channel row), one workgroup per pair. */
int hdmoe_wbank_desc_bytes(void);
int hdmoe_wbank_prep(const void* descs, const int* rows, int nrows, int mutate, HS stream);
int hdmoe_wbank_bwd(const void* descs, const int* rows, int nrows, HS stream);

/* ---- K4/K7: pointwise, broadcast, relayout  (model_internals.py:33-127, model_components.py:232-253) ----- */
int hdmoe_axpby(void* out, const void* x, const void* y, float a, float b, long n, int dtype, HS stream);      /* a*x + b*y (y may be NULL) : mp_sum */
int hdmoe_sum_n(void* out, const void* const* srcs, const float* src_scale, int n, long nelem, int dtype, HS stream);   /* sum_k src_scale[k] * srcs[k], n <= 16 tensors (16-byte aligned), src_scale = host array or NULL (all 1): fan-out backward */
int hdmoe_affine(void* out, const void* x, float a, float c, long n, int dtype, HS stream);                     /* a*x + c */
int hdmoe_mul(void* out, const void* x, const void* y, long n, int dtype, HS stream);
/* One Heun stage of the EDM sampler with the sigma schedule on the device (reference Utils/EDM_sampler.py:90-107): t = float64 [N + 1] device
 * array, idx = device int32 stage counter.  sched_pick: *sigma = t[*idx + off];  heun_euler: x_next = x_hat + (t[i+1] - t[i]) (x_hat - den) / t[i];
 * heun_correct: out = x_hat + h (0.5 (x_hat - den) / t[i] + 0.5 (x_next - den2) / t[i+1]);  idx_advance: *idx += 1.  fp32 latents of n elements. */
int hdmoe_sched_pick(float* sigma, const double* t, const int* idx, int off, HS stream);
int hdmoe_idx_advance(int* idx, HS stream);
int hdmoe_heun_euler(float* xn, const float* xh, const float* den, const double* t, const int* idx, long n, HS stream);
int hdmoe_heun_correct(float* out, const float* xh, const float* den, const float* xn, const float* den2, const double* t, const int* idx, long n, HS stream);
/* Measurement aid: `blocks` x 256 threads, 8 x `iters` dependent v_exp_f32 per thread (out: blocks * 256 floats).  bench.py times it to state the
 * transcendental issue rate the attention kernels (reference models/model_internals.py:374-404: one exp per score) are bounded by. */
int hdmoe_exp_rate(float* out, int blocks, int iters, HS stream);
int hdmoe_cast(void* out, const void* x, long n, int dt_in, int dt_out, HS stream);
/* Separable even-length FIR resampling, channel-last (resample(x, f, mode) for f other than [1, 1]; reference models/model_internals.py:95-127):
 * up == 0: stride-2 depthwise correlation with outer(k, k) and padding `pad` (F.conv2d there); up == 1: its transpose (F.conv_transpose2d).
 * taps: L <= 8 host floats.  Each direction is the other's backward. */
int hdmoe_fir_resample(void* y, const void* x, const float* taps, int L, int pad, float scale, int up, int N, int H, int W, int Ho, int Wo, int C,
                       int dtype, HS stream);
int hdmoe_mp_silu_fwd(void* out, const void* x, long n, int dtype, HS stream);
int hdmoe_mp_silu_bwd(void* dx, const void* dy, const void* x, long n, int dtype, HS stream);
/* fused decoder-block entry (model_components.py:232-253): the block input feeds mp_silu AND the skip / residual path */
int hdmoe_mp_silu_bwd_add(void* dx, const void* dy, const void* x, const void* gx, float sx, long n, int dtype, HS stream);   /* dx = sx * gx + dy * mp_silu'(x) */
int hdmoe_cat2_silu_fwd(void* out, void* out_h, const void* a, const void* b, float wa, float wb, int Ca, int Cb, long rows,
                        int dtype, HS stream);                                                                     /* out = mp_cat, out_h = mp_silu(out) */
int hdmoe_cat2_silu_bwd(void* da, void* db, const void* gcat, const void* gh, const void* xcat, float wa, float wb, int Ca,
                        int Cb, long rows, int dtype, HS stream);
int hdmoe_sigmoid_fwd(void* out, const void* x, float a, long n, int dtype, HS stream);                         /* sigmoid(a*x) */
int hdmoe_sigmoid_bwd(void* dx, const void* dy, const void* y, float a, long n, int dtype, HS stream);
int hdmoe_film_silu_fwd(void* out, const void* u, const float* e, int N, long HW, int C, int dtype, HS stream);  /* mp_silu(u * e[n][c]) */
int hdmoe_film_silu_bwd(void* du, float* de, const void* da, const void* u, const float* e, int N, long HW, int C,
                        int dtype, HS stream);                                                                   /* de accumulates */
int hdmoe_film_silu_drop_fwd(void* out, const void* u, const float* e, int N, long HW, int C, unsigned long long seed,
                             const unsigned long long* seed_dev, float p, int dtype, HS stream);                  /* + F.dropout fused (:245-246) */
int hdmoe_film_silu_drop_bwd(void* du, float* de, const void* da, const void* u, const float* e, int N, long HW, int C,
                             unsigned long long seed, const unsigned long long* seed_dev, float p, int dtype, HS stream);
int hdmoe_scale_rows_fwd(void* out, const void* x, const float* s, long rows, long L, int dtype, HS stream);    /* out[r][:] = s[r]*x[r][:] */
int hdmoe_scale_rows_bwd(void* dx, float* ds, const void* dy, const void* x, const float* s, long rows, long L,
                         int dtype, HS stream);                                                                  /* dx and/or ds (accumulates) */
int hdmoe_cat2_fwd(void* out, const void* a, const void* b, float wa, float wb, int Ca, int Cb, long rows, int dtype, HS stream); /* mp_cat */
int hdmoe_cat2_bwd(void* da, void* db, const void* dout, float wa, float wb, int Ca, int Cb, long rows, int dtype, HS stream);
int hdmoe_pool2(void* out, const void* x, int N, int Ho, int Wo, int C, float scale, int dtype, HS stream);      /* resample 'down' (scale .25) / bwd of 'up' (1) */
int hdmoe_upsample2(void* out, const void* x, int N, int Ho, int Wo, int C, float scale, int dtype, HS stream);  /* resample 'up' (1) / bwd of 'down' (.25) */
int hdmoe_seq_reduce(float* out, const void* x, int N, long S, int C, float scale, int dtype, HS stream);        /* out[n][c] += scale*sum_s x[n][s][c] */
int hdmoe_seq_bcast_add(void* out, const void* x, const float* t, int N, long S, int C, float scale, int dtype, HS stream);
int hdmoe_bias_add(void* out, const void* x, const float* bias, long rows, long L, int dtype, HS stream);
int hdmoe_colsum(float* out, const void* dy, long rows, long L, int dtype, HS stream);                          /* accumulates */
int hdmoe_lerp_param_fwd(void* out, const void* a, const void* b, const float* alpha, long n, int dtype, HS stream);   /* model_config2.py:291 */
int hdmoe_lerp_param_bwd(void* da, void* db, float* dalpha, const void* g, const void* a, const void* b,
                         const float* alpha, long n, int dtype, HS stream);
int hdmoe_gate_mix_fwd(void* out, float* gate, const void* logits, const void* U, const void* A, long rows, int C,
                       int dtype, HS stream);                                                                    /* model_config2.py:297-301 */
int hdmoe_gate_mix_bwd(void* dU, void* dA, void* dlogits, const void* dout, const float* dgate, const float* gate,
                       const void* U, const void* A, long rows, int C, int dtype, HS stream);
int hdmoe_softmax_rows_fwd(float* out, const float* x, long rows, int C, float scale, HS stream);               /* model_components.py:64 */
int hdmoe_softmax_rows_bwd(float* dx, const float* dy, const float* y, long rows, int C, float scale, HS stream);
int hdmoe_nchw_to_nhwc(void* out, const float* x, const float* s, int N, int C, long HW, int dtype, HS stream);  /* + per-sample scale (c_in) */
int hdmoe_nhwc_to_nchw(float* out, const void* F, const float* sf, const float* x, const float* sx, int N, int C,
                       long HW, int dtype, HS stream);                                                           /* sf*F + sx*x : D_x, model_config2.py:449 */
int hdmoe_patch_relayout(void* out, const void* in, int N, int H, int W, int C, int p, int hp, int wp, int order,
                         int to_img, int dtype, HS stream);                                                      /* PixelShuffle / patchify */
int hdmoe_fourier(float* out, const float* x, const float* freqs, const float* phases, int B, int F, HS stream); /* model_internals.py:171-174 */
int hdmoe_edm_coeffs(float* coef, const float* sigma, int nsig, float sigma_data, int B, HS stream);             /* model_config2.py:431-438 */
int hdmoe_sigmoid_scaling(float* sv, float* su, float* pair, const float* c_noise, float tp, float soft, int B, HS stream); /* :244-249 */
int hdmoe_adaln_fwd(float* out, const float* x, const float* cond, long B, int F, HS stream);                   /* model_components.py:148-151 */
int hdmoe_adaln_bwd(float* dx, float* dcond, const float* g, const float* x, const float* cond, long B, int F, HS stream);
int hdmoe_take_col_pos_fwd(float* out, const float* w, long B, int E, int e, HS stream);                         /* w[:,e] where > 0 (model_config1.py:26,35) */
int hdmoe_take_col_pos_bwd(float* dw, const float* g, const float* w, long B, int E, int e, HS stream);          /* dw pre-zeroed */
/* Philox key = seed + (*seed_dev) * golden (seed_dev: optional device step counter, so captured graphs replay with fresh draws) */
int hdmoe_dropout(void* out, const void* x, unsigned long long seed, const unsigned long long* seed_dev, float p, long n, int dtype,
                  HS stream);                                                                                    /* F.dropout, model_components.py:245-246 */
int hdmoe_randn(float* out, unsigned long long seed, const unsigned long long* seed_dev, float scale, long n, HS stream); /* randn*zeta, :155-156 */
int hdmoe_seed_advance(unsigned long long* seed_dev, HS stream);

/* ---- K6: norms  (model_internals.py:8-30; nn.GroupNorm / nn.LayerNorm in model_components.py) ------------ */
int hdmoe_pixelnorm_fwd(void* xn, void* h, const void* x, long rows, int C, int dtype, HS stream);               /* h = mp_silu(xn), optional */
int hdmoe_pixelnorm_bwd(void* dx, const void* dxn, const void* dh, const void* x, long rows, int C, float sx, int dtype, HS stream);   /* sx scales dxn */
int hdmoe_groupnorm_fwd(void* y, float* mean, float* rstd, const void* x, const float* gamma, const float* beta, int N,
                        long S, int C, int G, int act, float eps, int dtype, HS stream);                         /* act: 0 none, 1 relu, 2 mp_silu */
int hdmoe_groupnorm_bwd(void* dx, float* dgamma, float* dbeta, float* ws, const void* dy, const void* x,
                        const float* gamma, const float* beta, const float* mean, const float* rstd, int N, long S, int C,
                        int G, int act, int dtype, HS stream);                                                   /* ws: 2*N*G floats */
/* small batches: each sample's rows are split over `parts` (<= 64) workgroups.  fwd ws: 2*N*parts*G floats (partial mean / M2,
 * merged in a fixed order); bwd ws: 2*N*G floats that the CALLER ZEROES (row-range blocks add into them). */
int hdmoe_groupnorm_fwd_split(void* y, float* mean, float* rstd, float* ws, int parts, const void* x, const float* gamma,
                              const float* beta, int N, long S, int C, int G, int act, float eps, int dtype, HS stream);
int hdmoe_groupnorm_bwd_bcast(void* dx, float* dgamma, float* dbeta, float* ws, const float* g, float scale, const void* x,
                              const float* gamma, const float* beta, const float* mean, const float* rstd, int N, long S, int C, int G,
                              int act, int dtype, HS stream);   /* incoming gradient = g[n][c] * scale at every position (mean over S follows the norm) */
int hdmoe_groupnorm_bwd_split(void* dx, float* dgamma, float* dbeta, float* ws, int parts, const void* dy, const void* x,
                              const float* gamma, const float* beta, const float* mean, const float* rstd, int N, long S, int C,
                              int G, int act, int dtype, HS stream);
/* Router-trunk backward with bf16 gradient tensors (GroupNorm(1, C) + ReLU of Router.hard_route, reference models/model_components.py:100-112;
 * x = the conv output, fp32).  gn1t_bwd: dz bf16 [N][S][C] or (dz == null) g[n][c] * gscale at every position -> dx bf16; ws: 2 N floats;
 * dgamma / dbeta accumulate.  gn1t_act: out bf16 = relu(y * scale[n][c] + shift[n][c]) (scale == null: bf16(y)) -- the conv input as the bf16
 * weight-gradient program reads it. */
int hdmoe_gn1t_bwd(void* dx, float* dgamma, float* dbeta, float* ws, const void* dz, const float* g, float gscale, const float* x, const float* gamma,
                   const float* beta, const float* mean, const float* rstd, int N, long S, int C, HS stream);
int hdmoe_gn1t_act(void* out, const float* y, const float* scale, const float* shift, int N, long S, int C, HS stream);
int hdmoe_layernorm_fwd(void* y, float* mean, float* rstd, const void* x, const float* gamma, const float* beta, long rows,
                        int C, float eps, int dtype, HS stream);
int hdmoe_layernorm_bwd(void* dx, float* dgamma, float* dbeta, const void* dy, const void* x, const float* gamma,
                        const float* mean, const float* rstd, long rows, int C, int dtype, HS stream);

/* ---- K5: attention core  (model_internals.py:374-404) -------------------------------------------------------- */
int hdmoe_attn_fwd(void* out, float* lse, const void* q, const void* k, const void* v, const float* bias, int B, int Sq,
                   int Skv, int H, int D, int Sb, int dtype, HS stream);
int hdmoe_attn_bwd(void* dq, void* dk, void* dv, float* dbias, float* delta, const void* dout, const void* out,
                   const void* q, const void* k, const void* v, const float* lse, const float* bias, int B, int Sq,
                   int Skv, int H, int D, int Sb, int dtype, HS stream);

/* rel_pos_bias resize for S > S0 (model_internals.py:388-397: F.interpolate(..., mode='bicubic', align_corners=False)).
 * fwd: out [H][S][S] <- table [H][S0][S0];  bwd: dtable (caller zeroes) += transpose of the same linear map applied to dout. */
int hdmoe_bicubic_fwd(float* out, const float* table, int H, int S0, int S, HS stream);
int hdmoe_bicubic_bwd(float* dtable, const float* dout, int H, int S0, int S, HS stream);

/* ---- ViT expert BANK: ragged tokens (csrc/ragged.hip, csrc/attention.hip) ----------------------------------------
 * The reference evaluates its ViT experts one by one on the samples routed to each (model_config1.py:25-37,
 * model_components.py:435-706).  The bank keeps all routed rows in one padded tensor [R][Sp][C]: rows [seg[g], seg[g+1])
 * (device int32, from hdmoe_dispatch_plan) belong to expert g and hold lens[g] (host int array) real tokens followed by
 * zero padding.  srcs / dsts / gamma / ... are host arrays of per-expert device pointers. */
int hdmoe_rag_pack(void* dst, const void* const* srcs, const float* const* pos, const int* seg, const int* lens, int ngroups,
                   int R, int Sp, int C, int dtype, HS stream);
int hdmoe_rag_pack_bwd(void* const* dsrcs, float* const* dpos, const void* ddst, const int* seg, const int* lens, int ngroups,
                       int R, int Sp, int C, int dtype, HS stream);
int hdmoe_rag_unpack(void* const* dsts, const void* src, const int* seg, const int* lens, int ngroups, int R, int Sp, int C,
                     int dtype, HS stream);
int hdmoe_rag_unpack_bwd(void* dsrc, const void* const* ddsts, const int* seg, const int* lens, int ngroups, int R, int Sp,
                         int C, int dtype, HS stream);
int hdmoe_rag_select(void* y, const void* const* outs, const int* seg, int ngroups, int R, long row_bytes, HS stream);
int hdmoe_rag_select_bwd(void* const* douts, const void* dy, const int* seg, int ngroups, int R, long row_bytes, HS stream);
/* nn.GroupNorm(G, C) over each row's real tokens (+ act: 0 none, 1 relu, 2 mp_silu); mean / rstd [R][G] */
int hdmoe_gn_rag_fwd(void* y, float* mean, float* rstd, const void* x, const float* const* gamma, const float* const* beta,
                     const int* seg, const int* lens, int ngroups, int R, int Sp, int C, int G, int act, float eps, int dtype,
                     HS stream);
int hdmoe_gn_rag_bwd(void* dx, float* const* dgamma, float* const* dbeta, const void* dy, const void* x,
                     const float* const* gamma, const float* const* beta, const float* mean, const float* rstd, const int* seg,
                     const int* lens, int ngroups, int R, int Sp, int C, int G, int act, int dtype, HS stream);
/* nn.LayerNorm(C) per token with the row's expert's affine; mean / rstd [R * Sp] */
int hdmoe_ln_rag_fwd(void* y, float* mean, float* rstd, const void* x, const float* const* gamma, const float* const* beta,
                     const int* seg, int ngroups, int R, int Sp, int C, float eps, int dtype, HS stream);
int hdmoe_ln_rag_bwd(void* dx, float* const* dgamma, float* const* dbeta, const void* dy, const void* x,
                     const float* const* gamma, const float* mean, const float* rstd, const int* seg, int ngroups, int R, int Sp,
                     int C, int dtype, HS stream);
/* self-attention over each row's real tokens with the expert's rel_pos_bias table bias[g] [H][sb[g]][sb[g]]
 * (model_internals.py:374-404); lse / delta [R][H][Sp]; dbias[g] accumulate (+=), or dbias == NULL */
int hdmoe_attn_rag_fwd(void* out, float* lse, const void* q, const void* k, const void* v, const float* const* bias,
                       const int* seg, const int* lens, const int* sb, int ngroups, int R, int Sp, int H, int D, int dtype,
                       HS stream);
int hdmoe_attn_rag_bwd(void* dq, void* dk, void* dv, float* const* dbias, float* delta, const void* dout, const void* out,
                       const void* q, const void* k, const void* v, const float* lse, const float* const* bias, const int* seg,
                       const int* lens, const int* sb, int ngroups, int R, int Sp, int H, int D, int dtype, HS stream);

/* ---- many small linear layers over ONE input in one launch (csrc/mlinear.hip): Unet_block.emb_layer of every block
 * (model_components.py:232-236) and MP_Attention.q_time / k_time / v_time of every ViT block (model_internals.py:360-372).
 * w[l] = prepared fp32 weight image of layer l, [ngroups][O[l]][Ipad]; lens = host array O[0..L); y / dy: host arrays of L device
 * pointers to [R][O[l]] fp32; G: host array of L * 8 device pointers to the [O[l]][I] fp32 gradient slabs (NULL = skip), += */
int hdmoe_mlinear_fwd(float* const* y, const float* x, const float* const* w, const int* seg, const int* lens, int L, int R, int I,
                      int Ipad, int ngroups, float c, HS stream);
int hdmoe_mlinear_dgrad(float* dx, const float* const* dy, const float* const* w, const int* seg, const int* lens, int L, int R,
                        int I, int Ipad, int ngroups, HS stream);
int hdmoe_mlinear_wgrad(float* const* G, const float* const* dy, const float* x, const int* seg, const int* lens, int L, int R, int I,
                        int ngroups, HS stream);

/* ---- K1/K2: router head + dispatch  (model_components.py:155-168, model_config1.py:11-39) ---------------------- */
int hdmoe_router_head_fwd(float* sparse, float* probs, float* xout, int* idx, const float* logits, const float* noise,
                          const float* mask, long B, int E, int k, HS stream);
int hdmoe_router_head_bwd(float* dlogits, const float* dsparse, const float* dprobs, const float* dxout,
                          const float* sparse, const float* probs, const int* idx, const float* mask, long B, int E, int k,
                          HS stream);
int hdmoe_dispatch_plan(int* perm, int* row_expert, float* row_w, int* inv, int* seg, const float* sparse, int B, int E,
                        int kcap, HS stream);
/* counts[e] += seg[e+1] - seg[e] as float: rows routed to expert e since the last optimizer step (read by hdmoe_mt_adamw's `use` flags;
 * cleared by the caller behind the update) */
int hdmoe_seg_counts(float* counts, const int* seg, int E, HS stream);
/* counts[e] += rows of `sparse` (B, E) with a weight > 0 in column e: the same bookkeeping on the path that evaluates every expert on
 * the whole batch (no dispatch plan); reference models/model_config1.py:26-29. */
int hdmoe_route_counts(float* counts, const float* sparse, int B, int E, HS stream);
int hdmoe_gather_rows(void* dst, const void* src, const int* perm, long R, long L, int dtype, HS stream);
int hdmoe_combine_rows_fwd(void* out, const void* ys, const int* inv, const float* row_w, long B, int kcap, long L,
                           int dtype, HS stream);
int hdmoe_combine_rows_bwd(void* dys, float* dsparse, const void* dout, const void* ys, const int* perm,
                           const int* row_expert, const float* row_w, long R, int E, long L, int dtype, HS stream);

/* ---- N2 (next to the path): EDM_LOSS fused  (Utils/utils.py:127-172) ---------------------------------------------- */
/* out[5] = loss, denoising, balance, z_loss, pure_loss; aux: 2E+3 floats kept for the backward; sse: B floats scratch */
int hdmoe_edm_loss_fwd(float* out, float* aux, float* sse, const float* denoised, const float* target, const float* log_var,
                       const float* pU, const float* pV, const float* rU, const float* rV, int B, long L, int E, float unet_bal,
                       float vit_bal, float z_bal, HS stream);
int hdmoe_edm_loss_bwd(float* dD, float* dlv, float* dpU, float* dpV, float* drU, float* drV, const float* gin, const float* aux,
                       const float* sse, const float* denoised, const float* target, const float* log_var, const float* rU,
                       const float* rV, int B, long L, int E, float unet_bal, float vit_bal, float z_bal, HS stream);

/* ---- the 33-channel first conv of Unet_expert (torch.cat([x, ones]), reference models/model_components.py:416) on the conv6 / wgrad6
 * kernels (csrc/ones6.hip): the ones channel becomes a per-expert border-aware bias map, its weight gradient a sum of dy over pixel
 * rectangles.  wf [g][tap][O][Ipad] / wd [g][tap][C + 1][Opad]: the weight images of the (C + 1)-channel layer; gbias, S: fp32
 * workspaces [ngroups][H][W][O]; G33: weight-gradient slabs [tap][O][C + 1] (+=); G32: zeroed workspace slabs [tap][O][C]; ws: as
 * hdmoe_conv_wgrad6.  Both return 1 without launching anything outside the domain (bf16, conv6 shapes, k in {3, 5} for the backward). */
int hdmoe_conv6_ones_fwd(const void* x, const void* wf, void* y, float* gbias, float alpha, const int* seg, int ngroups, long wstride,
                         int N, int H, int W, int C, int O, int Ipad, const int* kh, int dtype, HS stream);
int hdmoe_conv6_ones_bwd(const void* x, const void* dy, const void* wd, void* dx, float* const* G33, float* S, float* const* G32, const int* seg,
                         int ngroups, long wdstride, int N, int H, int W, int C, int O, int Opad, const int* kh, float alpha, void* ws, long ws_bytes,
                         int dtype, HS stream);

/* ---- K3+K4 fused: Unet_block main branch as one persistent launch (csrc/blk6.hip; reference models/model_components.py:240-253) ----
 * forward:  u = conv(x, w1); h = dropout_p(mp_silu(u * e[n][c])); y = alpha * conv(h, w2) + beta * res   (u, h, y written; the
 *           activation tile stays in LDS between the two convs).  x [N][H][W][Cin], u / h / y / res [N][H][W][C] bf16, e fp32 [N][C];
 *           w1 [g][tap][C][Cin], w2 [g][tap][C][C] forward weight images; kh: per-expert square kernel size, "same" padding.
 * backward: dh = alpha_mid * dgrad(dy, wd2) (stays in LDS); du = dropout / mp_silu / FiLM backward (written); de [N][C] += ...;
 *           dx = alpha * dgrad(du, wd1).  wd2 [g][tap][C][C], wd1 [g][tap][Cin][C] flipped dgrad images; u from the forward.
 * Both return 1 without launching outside the kernel's domain (bf16, W in {16, 32}, H % (256 / W) == 0, k in {3, 5, 7},
 * Cin % 32 == 0, C in {32, 64}). */
int hdmoe_unet_block_fwd(const void* x, const void* w1, const void* w2, void* u, void* h, void* y, const void* res, const float* e,
                         unsigned long long seed, const unsigned long long* seed_dev, float p, float alpha, float beta, const int* seg,
                         int ngroups, long w1stride, long w2stride, int N, int H, int W, int Cin, int C, const int* kh, int dtype,
                         HS stream);
int hdmoe_blk6_debug_stamps(void* buf);   /* development: 8 x 64 u64 device buffer for workgroup 0's in-kernel time stamps, or NULL */
int hdmoe_unet_block_bwd(const void* dy, const void* wd2, const void* wd1, const void* u, void* du, void* dx, float* de, const float* e,
                         unsigned long long seed, const unsigned long long* seed_dev, float p, float alpha, float alpha_mid, const int* seg,
                         int ngroups, long wd2stride, long wd1stride, int N, int H, int W, int Cin, int C, const int* kh, int dtype,
                         HS stream);

/* ---- N3: fused multi-tensor clip_grad_norm_ + AdamW  (Utils/training.py:55-65,195-197) ----------------------------------- */
/* descs: device array of {p, g, m, v, step, use addresses, numel, group} (hdmoe_opt_desc_bytes() bytes each); chunks: device int32 pairs
 * (descriptor index, 4096-element chunk index).  The clip coefficient min(1, max_norm/(sqrt(sumsq)+1e-6)) is read on the device.
 * step: the tensor's own AdamW step count (device float, advanced by hdmoe_mt_adamw); use: device float, > 0 when the tensor received a
 * gradient this step, or 0 = always -- a tensor of an expert that got no sample is skipped like a grad-None tensor in torch.optim.AdamW
 * (reference models/model_config1.py:26-29 leaves such an expert out of the graph; Utils/training.py:195-197). */
int hdmoe_opt_desc_bytes(void);
int hdmoe_mt_sumsq(float* sumsq, const void* descs, const int* chunks, int nchunks, float* ws, HS stream);   /* ws: nchunks floats; deterministic */
int hdmoe_mt_clip_scale(const void* descs, const int* chunks, int nchunks, const float* sumsq, float max_norm, HS stream);
int hdmoe_mt_adamw(const void* descs, const int* chunks, int nchunks, int ntensors, const float* sumsq, float max_norm, const float* group_lr,
                   const float* group_wd, int ngroups, float beta1, float beta2, float eps, HS stream);

#undef HS
#ifdef __cplusplus
}
#endif
#endif /* HDMOE_H */
