// K4 / K7: HBM-bound pointwise, broadcast and relayout kernels of the HDMOE hot path (fwd + bwd).
// All activation tensors are NHWC / [rows][C]; "vector path" tensors ((B,F) embeddings, per-sample
// scalars) are always fp32.  Grid-stride loops, <= 2048 workgroups of 256 threads (guide G11).
#include "common.h"
#include "hdmoe.h"

namespace {

constexpr int TPB = 256;
static inline unsigned grid_for(long n) {
  long b = (n + TPB - 1) / TPB;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (unsigned)b;
}
#define GRID_STRIDE(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

// ---------------------------------------------------------------- generic pointwise
template <typename T>
__global__ void axpby_kernel(T* out, const T* x, const T* y, float a, float b, long n) {
  GRID_STRIDE(i, n) {
    float v = a * to_f(x[i]);
    if (y) v += b * to_f(y[i]);
    out[i] = from_f<T>(v);
  }
}
template <typename T>
__global__ void affine_kernel(T* out, const T* x, float a, float c, long n) {
  GRID_STRIDE(i, n) out[i] = from_f<T>(a * to_f(x[i]) + c);
}
template <typename T>
__global__ void mul_kernel(T* out, const T* x, const T* y, long n) {
  GRID_STRIDE(i, n) out[i] = from_f<T>(to_f(x[i]) * to_f(y[i]));
}
template <typename TI, typename TO>
__global__ void cast_kernel(TO* out, const TI* x, long n) {
  GRID_STRIDE(i, n) out[i] = from_f<TO>(to_f(x[i]));
}
// fp16 exists at the module boundary only (`.half()` callers, reference tests/test_model/test_Unet_expert.py:106-115): converted to fp32
// at ingest and back at egress, the arithmetic in between is fp32 / bf16
__global__ void cast_h2f_kernel(float* out, const _Float16* x, long n) { GRID_STRIDE(i, n) out[i] = (float)x[i]; }
__global__ void cast_f2h_kernel(_Float16* out, const float* x, long n) { GRID_STRIDE(i, n) out[i] = (_Float16)x[i]; }
// Separable FIR resampling with an even-length filter f (resample(x, f, mode), reference model_internals.py:95-127), channel-last:
//   down: y[oy][ox] = scale * sum_{i,j} k[i] k[j] x[2 oy - pad + i][2 ox - pad + j]           (F.conv2d, stride 2, padding pad, depthwise)
//   up:   y[oy][ox] = scale * sum over (iy, i) with 2 iy - pad + i == oy (and likewise in x) of k[i] k[j] x[iy][ix]   (F.conv_transpose2d)
// each is the other's adjoint, so the same two kernels serve the backward passes (with the other one's scale).
// ---- EDM sampler, one Heun solver stage on the device (reference Utils/EDM_sampler.py:90-107): the sigma schedule t[0..N] (float64, as the
// host computes it) and the stage index live in device memory, so a captured hipGraph of a stage replays for every stage without host arithmetic
__global__ void sched_pick_kernel(float* sigma, const double* t, const int* idx, int off) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *sigma = (float)t[*idx + off];
}
__global__ void idx_advance_kernel(int* idx) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *idx += 1;
}
// x_next = x_hat + (t_next - t_hat) * (x_hat - denoised) / t_hat
__global__ void heun_euler_kernel(float* xn, const float* xh, const float* den, const double* t, const int* idx, long n) {
  const int i = *idx;
  const double th = t[i], h = t[i + 1] - th;
  const float a = (float)(1.0 + h / th), b = (float)(-h / th);
  GRID_STRIDE(e, n) xn[e] = a * xh[e] + b * den[e];
}
// x_out = x_hat + h * (0.5 * (x_hat - denoised) / t_hat + 0.5 * (x_next - denoised') / t_next)
__global__ void heun_correct_kernel(float* out, const float* xh, const float* den, const float* xn, const float* den2, const double* t, const int* idx, long n) {
  const int i = *idx;
  const double th = t[i], tn = t[i + 1], h = tn - th;
  const float a1 = (float)(1.0 + 0.5 * h / th), b1 = (float)(-0.5 * h / th), a2 = (float)(0.5 * h / tn), b2 = (float)(-0.5 * h / tn);
  GRID_STRIDE(e, n) {
    const float u = a1 * xh[e] + b1 * den[e];
    const float v = a2 * xn[e] + b2 * den2[e];
    out[e] = u + v;
  }
}
struct FirTaps { float k[8]; };
template <typename T>
__global__ void fir_down_kernel(T* y, const T* x, FirTaps f, int L, int pad, float scale, int H, int W, int Ho, int Wo, int C, long total) {
  GRID_STRIDE(e, total) {
    const int c = (int)(e % C); long t = e / C;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho); const long n = t / Ho;
    float acc = 0.f;
    for (int i = 0; i < L; ++i) {
      const int iy = 2 * oy - pad + i;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int j = 0; j < L; ++j) {
        const int ix = 2 * ox - pad + j;
        if ((unsigned)ix >= (unsigned)W) continue;
        acc += f.k[i] * f.k[j] * to_f(x[((n * H + iy) * W + ix) * C + c]);
      }
    }
    y[e] = from_f<T>(scale * acc);
  }
}
template <typename T>
__global__ void fir_up_kernel(T* y, const T* x, FirTaps f, int L, int pad, float scale, int H, int W, int Ho, int Wo, int C, long total) {
  GRID_STRIDE(e, total) {                                    // (H, W): input size; (Ho, Wo): output size
    const int c = (int)(e % C); long t = e / C;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho); const long n = t / Ho;
    float acc = 0.f;
    for (int i = 0; i < L; ++i) {
      const int ty = oy + pad - i;
      if (ty < 0 || (ty & 1)) continue;
      const int iy = ty >> 1;
      if (iy >= H) continue;
      for (int j = 0; j < L; ++j) {
        const int tx = ox + pad - j;
        if (tx < 0 || (tx & 1)) continue;
        const int ix = tx >> 1;
        if (ix >= W) continue;
        acc += f.k[i] * f.k[j] * to_f(x[((n * H + iy) * W + ix) * C + c]);
      }
    }
    y[e] = from_f<T>(scale * acc);
  }
}
template <typename T>
__global__ void mp_silu_fwd_kernel(T* out, const T* x, long n) {
  GRID_STRIDE(i, n) out[i] = from_f<T>(mp_silu_f(to_f(x[i])));
}
template <typename T>
__global__ void mp_silu_bwd_kernel(T* dx, const T* dy, const T* x, long n) {
  GRID_STRIDE(i, n) dx[i] = from_f<T>(to_f(dy[i]) * mp_silu_grad_f(to_f(x[i])));
}
template <typename T>
__global__ void sigmoid_fwd_kernel(T* out, const T* x, float a, long n) {   // sigmoid(a*x)
  GRID_STRIDE(i, n) out[i] = from_f<T>(1.f / (1.f + __expf(-a * to_f(x[i]))));
}
template <typename T>
__global__ void sigmoid_bwd_kernel(T* dx, const T* dy, const T* y, float a, long n) {
  GRID_STRIDE(i, n) { const float s = to_f(y[i]); dx[i] = from_f<T>(to_f(dy[i]) * a * s * (1.f - s)); }
}

// (the counter RNG -- philox / u01 / mix_seed -- lives in common.h: the conv epilogue with fused FiLM + dropout draws the same bits)
__global__ void seed_advance_kernel(unsigned long long* seed_dev) { *seed_dev += 1ull; }

// dropout keep-mask of element i (4 elements share one Philox call): the same (seed, index) gives the same bit in fwd and bwd
DEVI bool keep_bit(long i, uint32_t seed_lo, uint32_t seed_hi, float p) {
  uint32_t r[4];
  const long q = i >> 2;
  philox((uint32_t)q, (uint32_t)(q >> 32), seed_lo, seed_hi, r);
  return u01(r[i & 3]) >= p;
}

// ---------------------------------------------------------------- FiLM + mp_silu   (Unet_block, model_components.py:242-243)
template <typename T>
__global__ void film_silu_fwd_kernel(T* out, const T* u, const float* e, long HW, int C, long n) {
  GRID_STRIDE(i, n) {
    const long row = i / C; const int c = (int)(i - row * C);
    const long s = row / HW;
    out[i] = from_f<T>(mp_silu_f(to_f(u[i]) * e[s * C + c]));
  }
}
// one block per (pixel chunk, sample): du = da*silu'(u*e)*e ; de[n,c] += sum_pix da*silu'(u*e)*u
template <typename T>
__global__ void film_silu_bwd_kernel(T* du, float* de, const T* da, const T* u, const float* e, long HW, int C, int chunk) {
  extern __shared__ float sm[];
  const int s = blockIdx.y;
  for (int c = threadIdx.x; c < C; c += blockDim.x) sm[c] = 0.f;
  __syncthreads();
  const long p0 = (long)blockIdx.x * chunk;
  const long p1 = (p0 + chunk < HW) ? p0 + chunk : HW;
  const long base = (long)s * HW * C;
  for (long i = p0 * C + threadIdx.x; i < p1 * C; i += blockDim.x) {
    const int c = (int)(i % C);
    const float ev = e[(long)s * C + c], uv = to_f(u[base + i]);
    const float g = to_f(da[base + i]) * mp_silu_grad_f(uv * ev);
    du[base + i] = from_f<T>(g * ev);
    atomicAdd(&sm[c], g * uv);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) atomicAdd(&de[(long)s * C + c], sm[c]);
}

// ---------------------------------------------------------------- per-sample scalar scale
template <typename T>
__global__ void scale_rows_fwd_kernel(T* out, const T* x, const float* s, long L, long n) {
  GRID_STRIDE(i, n) out[i] = from_f<T>(to_f(x[i]) * s[i / L]);
}
template <typename T>
__global__ void scale_rows_bwd_kernel(T* dx, float* ds, const T* dy, const T* x, const float* s, long L, int chunk) {
  __shared__ float sm[16];
  const int r = blockIdx.y;
  const long p0 = (long)blockIdx.x * chunk;
  const long p1 = (p0 + chunk < L) ? p0 + chunk : L;
  const float sv = s[r];
  float acc = 0.f;
  for (long i = p0 + threadIdx.x; i < p1; i += blockDim.x) {
    const float g = to_f(dy[(long)r * L + i]);
    if (dx) dx[(long)r * L + i] = from_f<T>(g * sv);
    if (ds) acc += g * to_f(x[(long)r * L + i]);
  }
  if (ds) {
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) atomicAdd(&ds[r], acc);
  }
}

template <typename T>
__global__ void scale_rows_bwd_vec_kernel(T* dx, float* ds, const T* dy, const T* x, const float* s, long Lv, int chunk) {
  constexpr int W = VT<T>::W;
  __shared__ float sm[16];
  const int r = blockIdx.y;
  const long p0 = (long)blockIdx.x * chunk;
  const long p1 = (p0 + chunk < Lv) ? p0 + chunk : Lv;
  const float sv = s[r];
  float acc = 0.f;
  for (long i = p0 + threadIdx.x; i < p1; i += blockDim.x) {
    float g[W], xv[W];
    vload<T>(g, dy + ((long)r * Lv + i) * W);
    if (ds) {
      vload<T>(xv, x + ((long)r * Lv + i) * W);
#pragma unroll
      for (int j = 0; j < W; ++j) acc += g[j] * xv[j];
    }
    if (dx) {
#pragma unroll
      for (int j = 0; j < W; ++j) g[j] *= sv;
      vstore<T>(dx + ((long)r * Lv + i) * W, g);
    }
  }
  if (ds) {
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) atomicAdd(&ds[r], acc);
  }
}

// ---------------------------------------------------------------- mp_cat  (model_internals.py:69-92)
template <typename T>
__global__ void cat2_fwd_kernel(T* out, const T* a, const T* b, float wa, float wb, int Ca, int Cb, long rows) {
  const int C = Ca + Cb;
  GRID_STRIDE(i, rows * C) {
    const long r = i / C; const int c = (int)(i - r * C);
    out[i] = from_f<T>(c < Ca ? wa * to_f(a[r * Ca + c]) : wb * to_f(b[r * Cb + c - Ca]));
  }
}
template <typename T>
__global__ void cat2_bwd_kernel(T* da, T* db, const T* dout, float wa, float wb, int Ca, int Cb, long rows) {
  const int C = Ca + Cb;
  GRID_STRIDE(i, rows * C) {
    const long r = i / C; const int c = (int)(i - r * C);
    const float g = to_f(dout[i]);
    if (c < Ca) da[r * Ca + c] = from_f<T>(wa * g);
    else db[r * Cb + c - Ca] = from_f<T>(wb * g);
  }
}

// ---------------------------------------------------------------- resample (model_internals.py:95-127)
template <typename T>   // out[n][h][w][c] = scale * sum_{2x2} x[n][2h+i][2w+j][c]
__global__ void pool2_kernel(T* out, const T* x, int Ho, int Wo, int C, float scale, long n) {
  GRID_STRIDE(i, n) {
    long t = i; const int c = (int)(t % C); t /= C;
    const int w = (int)(t % Wo); t /= Wo;
    const int h = (int)(t % Ho); const long s = t / Ho;
    const T* p = x + (((s * 2 * Ho + 2 * h) * 2 * Wo) + 2 * w) * C + c;
    const long rs = (long)2 * Wo * C;
    out[i] = from_f<T>(scale * (to_f(p[0]) + to_f(p[C]) + to_f(p[rs]) + to_f(p[rs + C])));
  }
}
template <typename T>   // out[n][h][w][c] = scale * x[n][h/2][w/2][c]
__global__ void upsample2_kernel(T* out, const T* x, int Ho, int Wo, int C, float scale, long n) {
  GRID_STRIDE(i, n) {
    long t = i; const int c = (int)(t % C); t /= C;
    const int w = (int)(t % Wo); t /= Wo;
    const int h = (int)(t % Ho); const long s = t / Ho;
    out[i] = from_f<T>(scale * to_f(x[((s * (Ho / 2) + h / 2) * (Wo / 2) + w / 2) * C + c]));
  }
}

// 16-byte vector forms (C a multiple of the vector width): one thread per output vector, 32-bit index math per row
template <typename T>
__global__ void pool2_vec_kernel(T* out, const T* x, int Ho, int Wo, int cv, float scale, long nv) {
  constexpr int W = VT<T>::W;
  GRID_STRIDE(i, nv) {
    const long row = i / ((long)Wo * cv);                      // (sample, output row)
    const int rem = (int)(i - row * ((long)Wo * cv));
    const int w = rem / cv, c = rem - w * cv;
    const T* p = x + ((row * 2) * (2 * Wo) + 2 * w) * (long)cv * W + (long)c * W;
    const long rs = (long)2 * Wo * cv * W;
    float a[W], b[W], d[W], e[W];
    vload<T>(a, p); vload<T>(b, p + (long)cv * W); vload<T>(d, p + rs); vload<T>(e, p + rs + (long)cv * W);
#pragma unroll
    for (int j = 0; j < W; ++j) a[j] = scale * (a[j] + b[j] + d[j] + e[j]);
    vstore<T>(out + i * W, a);
  }
}
template <typename T>
__global__ void upsample2_vec_kernel(T* out, const T* x, int Ho, int Wo, int cv, float scale, long nv) {
  constexpr int W = VT<T>::W;
  GRID_STRIDE(i, nv) {
    const long row = i / ((long)Wo * cv);                      // (sample, output row): sample = row / Ho, h = row % Ho
    const int rem = (int)(i - row * ((long)Wo * cv));
    const int w = rem / cv, c = rem - w * cv;
    const long s = row / Ho; const int h = (int)(row - s * Ho);
    float a[W];
    vload<T>(a, x + (((s * (Ho / 2) + h / 2) * (Wo / 2) + w / 2) * (long)cv + c) * W);
#pragma unroll
    for (int j = 0; j < W; ++j) a[j] *= scale;
    vstore<T>(out + i * W, a);
  }
}

// ---------------------------------------------------------------- sequence reduce / broadcast  ([N][S][C] <-> fp32 [N][C])
template <typename T>
__global__ void seq_reduce_kernel(float* out, const T* x, long S, int C, float scale, int chunk) {
  extern __shared__ float sm[];
  const int s = blockIdx.y;
  for (int c = threadIdx.x; c < C; c += blockDim.x) sm[c] = 0.f;
  __syncthreads();
  const long p0 = (long)blockIdx.x * chunk;
  const long p1 = (p0 + chunk < S) ? p0 + chunk : S;
  const long base = (long)s * S * C;
  for (long i = p0 * C + threadIdx.x; i < p1 * C; i += blockDim.x) atomicAdd(&sm[i % C], to_f(x[base + i]));
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) atomicAdd(&out[(long)s * C + c], scale * sm[c]);
}
template <typename T>   // out = (x ? x : 0) + scale * t[n][c]
__global__ void seq_bcast_add_kernel(T* out, const T* x, const float* t, long S, int C, float scale, long n) {
  GRID_STRIDE(i, n) {
    const long row = i / C; const int c = (int)(i - row * C);
    const float v = scale * t[(row / S) * C + c];
    out[i] = from_f<T>(x ? to_f(x[i]) + v : v);
  }
}
// out[r][i] = x[r][i] + bias[i]   /   colsum: dbias[i] += sum_r dy[r][i]
template <typename T>
__global__ void bias_add_kernel(T* out, const T* x, const float* bias, long L, long n) {
  GRID_STRIDE(i, n) out[i] = from_f<T>(to_f(x[i]) + bias[i % L]);
}
template <typename T>
__global__ void colsum_kernel(float* out, const T* dy, long rows, long L, int rchunk) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L) return;
  const long r0 = (long)blockIdx.y * rchunk;
  const long r1 = (r0 + rchunk < rows) ? r0 + rchunk : rows;
  float acc = 0.f;
  for (long r = r0; r < r1; ++r) acc += to_f(dy[r * L + i]);
  atomicAdd(&out[i], acc);
}

// ---------------------------------------------------------------- lerp with a learnable scalar  (model_config2.py:291)
template <typename T>
__global__ void lerp_param_fwd_kernel(T* out, const T* a, const T* b, const float* alpha, long n) {
  const float al = *alpha;
  GRID_STRIDE(i, n) { const float av = to_f(a[i]); out[i] = from_f<T>(av + al * (to_f(b[i]) - av)); }
}
template <typename T>
__global__ void lerp_param_bwd_kernel(T* da, T* db, float* dalpha, const T* g, const T* a, const T* b, const float* alpha, long n) {
  __shared__ float sm[16];
  const float al = *alpha;
  float acc = 0.f;
  GRID_STRIDE(i, n) {
    const float gv = to_f(g[i]);
    da[i] = from_f<T>((1.f - al) * gv);
    db[i] = from_f<T>(al * gv);
    acc += gv * (to_f(b[i]) - to_f(a[i]));
  }
  acc = block_sum(acc, sm);
  if (threadIdx.x == 0) atomicAdd(dalpha, acc);
}

template <typename T>
__global__ void lerp_param_bwd_vec_kernel(T* da, T* db, float* dalpha, const T* g, const T* a, const T* b, const float* alpha, long nv) {
  constexpr int W = VT<T>::W;
  __shared__ float sm[16];
  const float al = *alpha;
  float acc = 0.f;
  GRID_STRIDE(v, nv) {
    float gv[W], av[W], bv[W], o1[W], o2[W];
    vload<T>(gv, g + v * W); vload<T>(av, a + v * W); vload<T>(bv, b + v * W);
#pragma unroll
    for (int j = 0; j < W; ++j) { o1[j] = (1.f - al) * gv[j]; o2[j] = al * gv[j]; acc += gv[j] * (bv[j] - av[j]); }
    vstore<T>(da + v * W, o1); vstore<T>(db + v * W, o2);
  }
  acc = block_sum(acc, sm);
  if (threadIdx.x == 0) atomicAdd(dalpha, acc);
}

// ---------------------------------------------------------------- output gate  (model_config2.py:297-301)
// g = softmax(logits[.,2]); mixed = g0*U + g1*A; out = ((1-t)*U + t*mixed)/sqrt((1-t)^2+t^2), t = 0.5
template <typename T>
__global__ void gate_mix_fwd_kernel(T* out, float* gate, const T* logits, const T* U, const T* A, int C, long rows) {
  const float k = 0.70710678118654752f;   // 0.5 / sqrt(0.5)
  GRID_STRIDE(i, rows * C) {
    const long r = i / C; const int c = (int)(i - r * C);
    const float l0 = to_f(logits[2 * r]), l1 = to_f(logits[2 * r + 1]);
    const float m = fmaxf(l0, l1);
    const float e0 = __expf(l0 - m), e1 = __expf(l1 - m);
    const float g0 = e0 / (e0 + e1), g1 = e1 / (e0 + e1);
    if (c == 0) { gate[2 * r] = g0; gate[2 * r + 1] = g1; }
    const float u = to_f(U[i]), a = to_f(A[i]);
    out[i] = from_f<T>(k * (u + g0 * u + g1 * a));
  }
}
// one thread per pixel row
template <typename T>
__global__ void gate_mix_bwd_kernel(T* dU, T* dA, T* dlogits, const T* dout, const float* dgate, const float* gate,
                                    const T* U, const T* A, int C, long rows) {
  const float k = 0.70710678118654752f;
  GRID_STRIDE(r, rows) {
    const float g0 = gate[2 * r], g1 = gate[2 * r + 1];
    float d0 = dgate ? dgate[2 * r] : 0.f, d1 = dgate ? dgate[2 * r + 1] : 0.f;
    for (int c = 0; c < C; ++c) {
      const long i = r * C + c;
      const float go = k * to_f(dout[i]);
      const float u = to_f(U[i]), a = to_f(A[i]);
      dU[i] = from_f<T>(go * (1.f + g0));
      dA[i] = from_f<T>(go * g1);
      d0 += go * u; d1 += go * a;
    }
    const float dot = d0 * g0 + d1 * g1;
    dlogits[2 * r] = from_f<T>(g0 * (d0 - dot));
    dlogits[2 * r + 1] = from_f<T>(g1 * (d1 - dot));
  }
}

// 16-byte vector forms: LPR = C / W consecutive lanes own one pixel row (LPR a power of two <= 64, so rows never straddle a wave)
template <typename T>
__global__ void gate_mix_fwd_vec_kernel(T* out, float* gate, const T* logits, const T* U, const T* A, int lpr, long nvec) {
  constexpr int W = VT<T>::W;
  const float k = 0.70710678118654752f;
  GRID_STRIDE(v, nvec) {
    const long r = v / lpr;
    const float l0 = to_f(logits[2 * r]), l1 = to_f(logits[2 * r + 1]);
    const float m = fmaxf(l0, l1);
    const float e0 = __expf(l0 - m), e1 = __expf(l1 - m);
    const float g0 = e0 / (e0 + e1), g1 = e1 / (e0 + e1);
    if (v - r * lpr == 0) { gate[2 * r] = g0; gate[2 * r + 1] = g1; }
    float u[W], a[W];
    vload<T>(u, U + v * W); vload<T>(a, A + v * W);
#pragma unroll
    for (int j = 0; j < W; ++j) u[j] = k * (u[j] + g0 * u[j] + g1 * a[j]);
    vstore<T>(out + v * W, u);
  }
}
template <typename T>
__global__ void gate_mix_bwd_vec_kernel(T* dU, T* dA, T* dlogits, const T* dout, const float* dgate, const float* gate,
                                        const T* U, const T* A, int lpr, long nvec, long nvec_pad) {
  constexpr int W = VT<T>::W;
  const float k = 0.70710678118654752f;
  // nvec_pad is a multiple of 64: every lane of a wave takes part in the shuffles (tail lanes carry zeros)
  GRID_STRIDE(v, nvec_pad) {
    const long r = v / lpr;
    const bool ok = v < nvec;
    float g0 = 0.f, g1 = 0.f, d0 = 0.f, d1 = 0.f;
    if (ok) {
      g0 = gate[2 * r]; g1 = gate[2 * r + 1];
      float go[W], u[W], a[W];
      vload<T>(go, dout + v * W); vload<T>(u, U + v * W); vload<T>(a, A + v * W);
#pragma unroll
      for (int j = 0; j < W; ++j) { go[j] *= k; d0 += go[j] * u[j]; d1 += go[j] * a[j]; u[j] = go[j] * (1.f + g0); a[j] = go[j] * g1; }
      vstore<T>(dU + v * W, u); vstore<T>(dA + v * W, a);
    }
    for (int o = lpr >> 1; o > 0; o >>= 1) { d0 += __shfl_xor(d0, o, 64); d1 += __shfl_xor(d1, o, 64); }
    if (ok && v - r * lpr == 0) {
      if (dgate) { d0 += dgate[2 * r]; d1 += dgate[2 * r + 1]; }
      const float dot = d0 * g0 + d1 * g1;
      dlogits[2 * r] = from_f<T>(g0 * (d0 - dot));
      dlogits[2 * r + 1] = from_f<T>(g1 * (d1 - dot));
    }
  }
}

// ---------------------------------------------------------------- row softmax for tiny C (Scaling_router, model_components.py:64)
__global__ void softmax_rows_fwd_kernel(float* out, const float* x, int C, float scale, long rows) {
  GRID_STRIDE(r, rows) {
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, x[r * C + c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += __expf(x[r * C + c] - m);
    for (int c = 0; c < C; ++c) out[r * C + c] = scale * __expf(x[r * C + c] - m) / s;
  }
}
__global__ void softmax_rows_bwd_kernel(float* dx, const float* dy, const float* y, int C, float scale, long rows) {
  GRID_STRIDE(r, rows) {    // y = scale * p
    float dot = 0.f;
    for (int c = 0; c < C; ++c) dot += dy[r * C + c] * y[r * C + c];
    for (int c = 0; c < C; ++c) dx[r * C + c] = y[r * C + c] * (dy[r * C + c] - dot / scale);
  }
}

// ---------------------------------------------------------------- layout: NCHW fp32 <-> NHWC T with per-sample scale
template <typename T>   // out_nhwc[n][p][c] = s[n] * x_nchw[n][c][p]
__global__ void nchw_to_nhwc_kernel(T* out, const float* x, const float* s, int C, long HW, long n) {
  GRID_STRIDE(i, n) {
    long t = i; const int c = (int)(t % C); t /= C;
    const long p = t % HW; const long b = t / HW;
    const float v = x[(b * C + c) * HW + p];
    out[i] = from_f<T>(s ? v * s[b] : v);
  }
}
template <typename T>   // out_nchw = sf[n]*F_nhwc + (x ? sx[n]*x_nchw : 0)
__global__ void nhwc_to_nchw_kernel(float* out, const T* F, const float* sf, const float* x, const float* sx, int C, long HW, long n) {
  GRID_STRIDE(i, n) {
    long t = i; const long p = t % HW; t /= HW;
    const int c = (int)(t % C); const long b = t / C;
    float v = to_f(F[(b * HW + p) * C + c]);
    if (sf) v *= sf[b];
    if (x) v += (sx ? sx[b] : 1.f) * x[i];
    out[i] = v;
  }
}

// ---------------------------------------------------------------- patch <-> image relayout (Vit_expert, model_components.py:698-704)
// tok[b][(hp,wp)][f] <-> img[b][hp*p+i][wp*p+j][c];  order 0: f = (i*p+j)*C + c ;  order 1 (PixelShuffle): f = c*p*p + i*p + j
template <typename T, bool TO_IMG>
__global__ void patch_relayout_kernel(T* out, const T* in, int H, int W, int C, int p, int hp, int wp, int order, long n) {
  // iterates over image elements of the *padded-free* image (H x W); tokens cover hp*p x wp*p
  GRID_STRIDE(i, n) {
    long t = i; const int c = (int)(t % C); t /= C;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H); const long b = t / H;
    const int ph = y / p, ii = y % p, pw = x / p, jj = x % p;
    const int f = order ? c * p * p + ii * p + jj : (ii * p + jj) * C + c;
    const long tk = ((b * hp + ph) * wp + pw) * ((long)C * p * p) + f;
    if (TO_IMG) out[i] = in[tk];
    else out[tk] = in[i];
  }
}

// 16-byte version (C % (16 / sizeof(T)) == 0): one thread per 8- (bf16) / 4-channel (fp32) vector of an image pixel; the index
// arithmetic (five divisions by run-time values) is paid once per vector instead of once per element.  order 0: the token side is
// a 16-byte vector too; order 1 (PixelShuffle): the vector's channels sit p*p elements apart in the token.
template <typename T, bool TO_IMG>
__global__ void patch_relayout_vec_kernel(T* out, const T* in, int H, int W, int C, int p, int hp, int wp, int order, long nv) {
  constexpr int VW = VT<T>::W;
  const int CV = C / VW;
  GRID_STRIDE(i, nv) {
    long t = i; const int c0 = (int)(t % CV) * VW; t /= CV;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H); const long b = t / H;
    const int ph = y / p, ii = y - ph * p, pw = x / p, jj = x - pw * p;
    const long tb = ((b * hp + ph) * wp + pw) * ((long)C * p * p);
    T* img = (TO_IMG ? out : const_cast<T*>(in)) + i * VW;
    if (order == 0) {
      T* tk = (TO_IMG ? const_cast<T*>(in) : out) + tb + (long)(ii * p + jj) * C + c0;
      if (TO_IMG) *reinterpret_cast<uint4*>(img) = *reinterpret_cast<const uint4*>(tk);
      else *reinterpret_cast<uint4*>(tk) = *reinterpret_cast<const uint4*>(img);
    } else {
      const int pp = p * p;
      T* tk = (TO_IMG ? const_cast<T*>(in) : out) + tb + (long)c0 * pp + ii * p + jj;
      alignas(16) T v[VW];
      if (TO_IMG) {
#pragma unroll
        for (int k = 0; k < VW; ++k) v[k] = tk[(long)k * pp];
        *reinterpret_cast<uint4*>(img) = *reinterpret_cast<const uint4*>(v);
      } else {
        *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(img);
#pragma unroll
        for (int k = 0; k < VW; ++k) tk[(long)k * pp] = v[k];
      }
    }
  }
}

// image -> tokens in PixelShuffle order (f = c*p*p + i*p + j) with the 16-byte vector on the TOKEN side: a thread owns VW consecutive j
// of one (token, c, i) and gathers them from VW neighbouring pixels (the image-side-vector form above scatters 2-byte stores p*p
// elements apart: 75 us for 67 MB; this form writes whole vectors).  Needs p % VW == 0.
template <typename T>
__global__ void patch_to_tokens_o1_vec_kernel(T* tok, const T* img, int H, int W, int C, int p, int hp, int wp, long nv) {
  constexpr int VW = VT<T>::W;
  const int jv = p / VW;                                     // vectors per (c, i) row of a token
  GRID_STRIDE(v, nv) {
    long t = v;
    const int j0 = (int)(t % jv) * VW; t /= jv;
    const int ii = (int)(t % p); t /= p;
    const int c = (int)(t % C); t /= C;
    const int pw = (int)(t % wp); t /= wp;
    const int ph = (int)(t % hp); const long b = t / hp;
    const int y = ph * p + ii, x0 = pw * p + j0;
    alignas(16) T vals[VW];
#pragma unroll
    for (int k = 0; k < VW; ++k) vals[k] = (y < H && x0 + k < W) ? img[((b * H + y) * W + x0 + k) * C + c] : from_f<T>(0.f);
    *reinterpret_cast<uint4*>(tok + v * VW) = *reinterpret_cast<const uint4*>(vals);
  }
}

// ---------------------------------------------------------------- small vector-path kernels
__global__ void fourier_kernel(float* out, const float* x, const float* freqs, const float* phases, int F, long n) {
  GRID_STRIDE(i, n) {                                                  // model_internals.py:171-174
    const long b = i / F; const int f = (int)(i - b * F);
    out[i] = cosf(x[b] * freqs[f] + phases[f]) * 1.41421356237309515f;
  }
}
// EDM preconditioning coefficients (model_config2.py:431-438): coef[0..3][B] = c_skip, c_out, c_in, c_noise
__global__ void edm_coeffs_kernel(float* coef, const float* sigma, int nsig, float sd, int B) {
  GRID_STRIDE(b, B) {
    const float s = sigma[nsig == 1 ? 0 : b];
    const float q = s * s + sd * sd;
    coef[b] = sd * sd / q;
    coef[B + b] = s * sd / sqrtf(q);
    coef[2 * B + b] = 1.f / sqrtf(q);
    coef[3 * B + b] = logf(s) * 0.25f;
  }
}
// closed-form path scaling (model_config2.py:244-249): out[0][B]=s_vit, out[1][B]=s_unet, pair[B][2]
__global__ void sigmoid_scaling_kernel(float* sv, float* su, float* pair, const float* c_noise, float tp, float soft, int B) {
  GRID_STRIDE(b, B) {
    const float w = 1.f / (1.f + expf(-((c_noise[b] * 4.f - tp) / soft)));
    const float v = (w + 1e-2f) * 2.f, u = ((1.f - w) + 1e-2f) * 2.f;
    sv[b] = v; su[b] = u; pair[2 * b] = v; pair[2 * b + 1] = u;
  }
}



// ---------------------------------------------------------------- 16-byte vectorised variants of the hot pointwise ops
template <typename T>
__global__ void axpby_vec_kernel(T* out, const T* x, const T* y, float a, float b, long nv) {
  constexpr int W = VT<T>::W;
  GRID_STRIDE(v, nv) {
    float f[W], g[W];
    vload<T>(f, x + v * W);
    if (y) { vload<T>(g, y + v * W);
#pragma unroll
      for (int j = 0; j < W; ++j) f[j] = a * f[j] + b * g[j];
    } else {
#pragma unroll
      for (int j = 0; j < W; ++j) f[j] = a * f[j];
    }
    vstore<T>(out + v * W, f);
  }
}
template <typename T>
__global__ void mp_silu_fwd_vec_kernel(T* out, const T* x, long nv) {
  constexpr int W = VT<T>::W;
  GRID_STRIDE(v, nv) {
    float f[W];
    vload<T>(f, x + v * W);
#pragma unroll
    for (int j = 0; j < W; ++j) f[j] = mp_silu_f(f[j]);
    vstore<T>(out + v * W, f);
  }
}
template <typename T>
__global__ void mp_silu_bwd_vec_kernel(T* dx, const T* dy, const T* x, long nv) {
  constexpr int W = VT<T>::W;
  GRID_STRIDE(v, nv) {
    float f[W], g[W];
    vload<T>(f, x + v * W); vload<T>(g, dy + v * W);
#pragma unroll
    for (int j = 0; j < W; ++j) f[j] = g[j] * mp_silu_grad_f(f[j]);
    vstore<T>(dx + v * W, f);
  }
}
// dx = gx + dy * silu'(x): the gradient of a tensor that feeds mp_silu AND a second consumer (residual / skip), in one pass
template <typename T>
__global__ void mp_silu_bwd_add_vec_kernel(T* dx, const T* dy, const T* x, const T* gx, float sx, long nv) {
  constexpr int W = VT<T>::W;
  GRID_STRIDE(v, nv) {
    float f[W], g[W], r[W];
    vload<T>(f, x + v * W); vload<T>(g, dy + v * W); vload<T>(r, gx + v * W);
#pragma unroll
    for (int j = 0; j < W; ++j) f[j] = sx * r[j] + g[j] * mp_silu_grad_f(f[j]);
    vstore<T>(dx + v * W, f);
  }
}
template <typename T>
__global__ void film_silu_fwd_vec_kernel(T* out, const T* u, const float* e, long HW, int C, long nv, uint32_t seed_lo, uint32_t seed_hi,
                                         const unsigned long long* seed_dev, float p) {
  constexpr int W = VT<T>::W;
  const int cv = C / W;
  if (p > 0.f) mix_seed(seed_lo, seed_hi, seed_dev);
  const float inv = p > 0.f ? 1.f / (1.f - p) : 1.f;
  GRID_STRIDE(v, nv) {
    const long row = v / cv; const int c0 = (int)(v - row * cv) * W;
    const float* ep = e + (row / HW) * C + c0;
    float f[W];
    vload<T>(f, u + v * W);
    uint32_t r4[W];
    if (p > 0.f) {                                           // F.dropout fused (model_components.py:245-246): W = 4 or 8 elements
#pragma unroll
      for (int q = 0; q < W / 4; ++q) { const long qi = (v * W) / 4 + q; philox((uint32_t)qi, (uint32_t)(qi >> 32), seed_lo, seed_hi, r4 + 4 * q); }
    }
#pragma unroll
    for (int j = 0; j < W; ++j) {
      float a = mp_silu_f(f[j] * ep[j]);
      if (p > 0.f) a = u01(r4[j]) >= p ? a * inv : 0.f;
      f[j] = a;
    }
    vstore<T>(out + v * W, f);
  }
}
// block = (pixel chunk, sample); a thread's vectors all carry the same channel chunk (256 % (C/W) == 0): register partials
template <typename T>
__global__ __launch_bounds__(256) void film_silu_bwd_vec_kernel(T* du, float* de, const T* da, const T* u, const float* e, long HW, int C, int chunk,
                                                               uint32_t seed_lo, uint32_t seed_hi, const unsigned long long* seed_dev, float p) {
  constexpr int W = VT<T>::W;
  extern __shared__ float sm[];
  if (p > 0.f) mix_seed(seed_lo, seed_hi, seed_dev);
  const float inv = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const int s = blockIdx.y, cv = C / W;
  for (int c = threadIdx.x; c < C; c += blockDim.x) sm[c] = 0.f;
  __syncthreads();
  const long p0 = (long)blockIdx.x * chunk;
  const long p1 = (p0 + chunk < HW) ? p0 + chunk : HW;
  const long base = (long)s * HW * C;
  const int c0 = (threadIdx.x % cv) * W;
  float ev[W], acc[W];
#pragma unroll
  for (int j = 0; j < W; ++j) { ev[j] = e[(long)s * C + c0 + j]; acc[j] = 0.f; }
  for (long v = p0 * cv + threadIdx.x; v < p1 * cv; v += 256) {
    float uv[W], g[W];
    vload<T>(uv, u + base + v * W); vload<T>(g, da + base + v * W);
    if (p > 0.f) {
      uint32_t r4[W];
#pragma unroll
      for (int q = 0; q < W / 4; ++q) { const long qi = (base + v * W) / 4 + q; philox((uint32_t)qi, (uint32_t)(qi >> 32), seed_lo, seed_hi, r4 + 4 * q); }
#pragma unroll
      for (int j = 0; j < W; ++j) g[j] = u01(r4[j]) >= p ? g[j] * inv : 0.f;
    }
#pragma unroll
    for (int j = 0; j < W; ++j) {
      g[j] *= mp_silu_grad_f(uv[j] * ev[j]);
      acc[j] += g[j] * uv[j];
      g[j] *= ev[j];
    }
    vstore<T>(du + base + v * W, g);
  }
#pragma unroll
  for (int j = 0; j < W; ++j) atomicAdd(&sm[c0 + j], acc[j]);
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) atomicAdd(&de[(long)s * C + c], sm[c]);
}
template <typename T>
__global__ void scale_rows_fwd_vec_kernel(T* out, const T* x, const float* s, long Lv, long nv) {
  constexpr int W = VT<T>::W;
  GRID_STRIDE(v, nv) {
    const float sv = s[v / Lv];
    float f[W];
    vload<T>(f, x + v * W);
#pragma unroll
    for (int j = 0; j < W; ++j) f[j] *= sv;
    vstore<T>(out + v * W, f);
  }
}
template <typename T>
__global__ void cat2_fwd_vec_kernel(T* out, const T* a, const T* b, float wa, float wb, int Ca, int Cb, long rows) {
  constexpr int W = VT<T>::W;
  const int cva = Ca / W, cv = (Ca + Cb) / W;
  GRID_STRIDE(v, rows * cv) {
    const long r = v / cv; const int c = (int)(v - r * cv);
    float f[W];
    const bool first = c < cva;
    vload<T>(f, first ? a + (r * cva + c) * W : b + (r * (cv - cva) + c - cva) * W);
    const float wgt = first ? wa : wb;
#pragma unroll
    for (int j = 0; j < W; ++j) f[j] *= wgt;
    vstore<T>(out + v * W, f);
  }
}
template <typename T>
__global__ void cat2_bwd_vec_kernel(T* da, T* db, const T* dout, float wa, float wb, int Ca, int Cb, long rows) {
  constexpr int W = VT<T>::W;
  const int cva = Ca / W, cv = (Ca + Cb) / W;
  GRID_STRIDE(v, rows * cv) {
    const long r = v / cv; const int c = (int)(v - r * cv);
    float f[W];
    vload<T>(f, dout + v * W);
    const bool first = c < cva;
    const float wgt = first ? wa : wb;
#pragma unroll
    for (int j = 0; j < W; ++j) f[j] *= wgt;
    vstore<T>(first ? da + (r * cva + c) * W : db + (r * (cv - cva) + c - cva) * W, f);
  }
}
// mp_cat + mp_silu of the result in one pass (decoder blocks: the concatenation feeds mp_silu and the skip / residual path), and the
// matching backward: d(cat) = g_cat + g_h * silu'(cat), split and weighted
template <typename T>
__global__ void cat2_silu_fwd_vec_kernel(T* out, T* out_h, const T* a, const T* b, float wa, float wb, int Ca, int Cb, long rows) {
  constexpr int W = VT<T>::W;
  const int cva = Ca / W, cv = (Ca + Cb) / W;
  GRID_STRIDE(v, rows * cv) {
    const long r = v / cv; const int c = (int)(v - r * cv);
    float f[W];
    const bool first = c < cva;
    vload<T>(f, first ? a + (r * cva + c) * W : b + (r * (cv - cva) + c - cva) * W);
    const float wgt = first ? wa : wb;
#pragma unroll
    for (int j = 0; j < W; ++j) f[j] *= wgt;
    vstore<T>(out + v * W, f);
    T tmp[W];
#pragma unroll
    for (int j = 0; j < W; ++j) tmp[j] = from_f<T>(f[j]);         // mp_silu sees the stored (rounded) value, like the two-kernel form
#pragma unroll
    for (int j = 0; j < W; ++j) f[j] = mp_silu_f(to_f(tmp[j]));
    vstore<T>(out_h + v * W, f);
  }
}
template <typename T>
__global__ void cat2_silu_bwd_vec_kernel(T* da, T* db, const T* gcat, const T* gh, const T* xcat, float wa, float wb, int Ca, int Cb, long rows) {
  constexpr int W = VT<T>::W;
  const int cva = Ca / W, cv = (Ca + Cb) / W;
  GRID_STRIDE(v, rows * cv) {
    const long r = v / cv; const int c = (int)(v - r * cv);
    float f[W], g[W], x[W];
    vload<T>(g, gh + v * W); vload<T>(x, xcat + v * W);
    if (gcat) vload<T>(f, gcat + v * W);
    else {
#pragma unroll
      for (int j = 0; j < W; ++j) f[j] = 0.f;
    }
    const bool first = c < cva;
    const float wgt = first ? wa : wb;
#pragma unroll
    for (int j = 0; j < W; ++j) f[j] = (f[j] + g[j] * mp_silu_grad_f(x[j])) * wgt;
    vstore<T>(first ? da + (r * cva + c) * W : db + (r * (cv - cva) + c - cva) * W, f);
  }
}
// seq_reduce with a fixed channel chunk per thread (256 % (C/W) == 0)
template <typename T>
__global__ __launch_bounds__(256) void seq_reduce_vec_kernel(float* out, const T* x, long S, int C, float scale, int chunk) {
  constexpr int W = VT<T>::W;
  extern __shared__ float sm[];
  const int s = blockIdx.y, cv = C / W;
  for (int c = threadIdx.x; c < C; c += blockDim.x) sm[c] = 0.f;
  __syncthreads();
  const long p0 = (long)blockIdx.x * chunk;
  const long p1 = (p0 + chunk < S) ? p0 + chunk : S;
  const long base = (long)s * S * C;
  const int c0 = (threadIdx.x % cv) * W;
  float acc[W];
#pragma unroll
  for (int j = 0; j < W; ++j) acc[j] = 0.f;
  for (long v = p0 * cv + threadIdx.x; v < p1 * cv; v += 256) {
    float f[W];
    vload<T>(f, x + base + v * W);
#pragma unroll
    for (int j = 0; j < W; ++j) acc[j] += f[j];
  }
#pragma unroll
  for (int j = 0; j < W; ++j) atomicAdd(&sm[c0 + j], acc[j]);
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) atomicAdd(&out[(long)s * C + c], scale * sm[c]);
}
// Deterministic forms (no atomics: the router's average pool feeds the top-k, and run-to-run float-add order would make the
// logits -- and through bf16 rounding every downstream tensor -- differ between identical calls).
// (a) one block per sample; thread (r, c) = (tid / cv, tid % cv) walks rows r, r + R, ... of its 16-byte channel chunk, then a
//     fixed-order tree over the R = 256 / cv row lanes.
template <typename T>
__global__ __launch_bounds__(256) void seq_reduce_det_vec_kernel(float* out, const T* x, long S, int C, float scale) {
  constexpr int W = VT<T>::W;
  extern __shared__ float sm[];                                // [R][C]
  const int n = blockIdx.x, cv = C / W, R = 256 / cv;
  const int r = threadIdx.x / cv, c0 = (threadIdx.x % cv) * W;
  const T* xs = x + (long)n * S * C;
  float acc[W];
#pragma unroll
  for (int j = 0; j < W; ++j) acc[j] = 0.f;
  for (long s0 = r; s0 < S; s0 += R) {
    float f[W];
    vload<T>(f, xs + s0 * C + c0);
#pragma unroll
    for (int j = 0; j < W; ++j) acc[j] += f[j];
  }
#pragma unroll
  for (int j = 0; j < W; ++j) sm[r * C + c0 + j] = acc[j];
  __syncthreads();
  for (int st = R >> 1; st >= 1; st >>= 1) {
    if (r < st) {
#pragma unroll
      for (int j = 0; j < W; ++j) sm[r * C + c0 + j] += sm[(r + st) * C + c0 + j];
    }
    __syncthreads();
  }
  if (r == 0) {
#pragma unroll
    for (int j = 0; j < W; ++j) out[(long)n * C + c0 + j] += scale * sm[c0 + j];
  }
}
// (b) any C: one thread per (sample, channel), four interleaved partial sums combined in a fixed order
template <typename T>
__global__ void seq_reduce_det_kernel(float* out, const T* x, long S, int C, float scale) {
  const int n = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const T* xs = x + (long)n * S * C + c;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  long s0 = 0;
  for (; s0 + 4 <= S; s0 += 4) {
    a0 += to_f(xs[s0 * C]); a1 += to_f(xs[(s0 + 1) * C]); a2 += to_f(xs[(s0 + 2) * C]); a3 += to_f(xs[(s0 + 3) * C]);
  }
  for (; s0 < S; ++s0) a0 += to_f(xs[s0 * C]);
  out[(long)n * C + c] += scale * ((a0 + a1) + (a2 + a3));
}
template <typename T> static inline bool chunk_fixed_ok(int C) {
  const int W = VT<T>::W;
  return C % W == 0 && C / W <= 256 && 256 % (C / W) == 0;
}

// ---------------------------------------------------------------- adaLN (Router, model_components.py:148-151): x*(1+gamma)+beta, cond = [gamma | beta]
__global__ void adaln_fwd_kernel(float* out, const float* x, const float* cond, int F, long n) {
  GRID_STRIDE(i, n) {
    const long b = i / F; const int f = (int)(i - b * F);
    out[i] = x[i] * (1.f + cond[b * 2 * F + f]) + cond[b * 2 * F + F + f];
  }
}
__global__ void adaln_bwd_kernel(float* dx, float* dcond, const float* g, const float* x, const float* cond, int F, long n) {
  GRID_STRIDE(i, n) {
    const long b = i / F; const int f = (int)(i - b * F);
    dx[i] = g[i] * (1.f + cond[b * 2 * F + f]);
    dcond[b * 2 * F + f] = g[i] * x[i];
    dcond[b * 2 * F + F + f] = g[i];
  }
}
// positive part of one column of the sparse gate matrix: out[b] = w[b][e] > 0 ? w[b][e] : 0   (NaN -> 0, as `mask = w > 0`)
__global__ void take_col_pos_fwd_kernel(float* out, const float* w, int E, int e, long B) {
  GRID_STRIDE(b, B) { const float v = w[b * E + e]; out[b] = v > 0.f ? v : 0.f; }
}
__global__ void take_col_pos_bwd_kernel(float* dw, const float* g, const float* w, int E, int e, long B) {
  GRID_STRIDE(b, B) { if (w[b * E + e] > 0.f) dw[b * E + e] = g[b]; }
}

// F.dropout(p): keep with prob 1-p, scale 1/(1-p); the mask is regenerated from (seed, element index) in bwd
template <typename T>
__global__ void dropout_kernel(T* out, const T* x, uint32_t seed_lo, uint32_t seed_hi, const unsigned long long* seed_dev, float p, long n) {
  mix_seed(seed_lo, seed_hi, seed_dev);
  const float inv = 1.f / (1.f - p);
  GRID_STRIDE(q, (n + 3) / 4) {
    uint32_t r[4];
    philox((uint32_t)q, (uint32_t)(q >> 32), seed_lo, seed_hi, r);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long i = q * 4 + j;
      if (i < n) out[i] = from_f<T>(u01(r[j]) >= p ? to_f(x[i]) * inv : 0.f);
    }
  }
}
__global__ void randn_kernel(float* out, uint32_t seed_lo, uint32_t seed_hi, const unsigned long long* seed_dev, float scale, long n) {
  mix_seed(seed_lo, seed_hi, seed_dev);
  GRID_STRIDE(q, (n + 3) / 4) {
    uint32_t r[4];
    philox((uint32_t)q, (uint32_t)(q >> 32), seed_lo, seed_hi, r);
    const float a0 = sqrtf(-2.f * logf(u01(r[0]))), a1 = sqrtf(-2.f * logf(u01(r[2])));
    const float t0 = 6.28318530717958648f * u01(r[1]), t1 = 6.28318530717958648f * u01(r[3]);
    const float v[4] = {a0 * cosf(t0), a0 * sinf(t0), a1 * cosf(t1), a1 * sinf(t1)};
#pragma unroll
    for (int j = 0; j < 4; ++j) if (q * 4 + j < n) out[q * 4 + j] = scale * v[j];
  }
}

// out = sum of up to 16 same-shaped tensors: the backward of a fan-out (a tensor consumed by n layers) in ONE pass -- autograd's own
// accumulation is n - 1 separate add launches over the same data
struct SumSrcs { const void* p[16]; float sc[16]; };
template <typename T>
__global__ void sum_n_kernel(T* out, SumSrcs s, int n, long nv, long nelem) {
  constexpr int W = VT<T>::W;
  GRID_STRIDE(v, nv) {
    if ((v + 1) * W <= nelem) {
      float f[W], g[W];
      vload<T>(f, (const T*)s.p[0] + v * W);
#pragma unroll
      for (int j = 0; j < W; ++j) f[j] *= s.sc[0];
      for (int k = 1; k < n; ++k) {
        vload<T>(g, (const T*)s.p[k] + v * W);
#pragma unroll
        for (int j = 0; j < W; ++j) f[j] += s.sc[k] * g[j];
      }
      vstore<T>(out + v * W, f);
    } else {
      for (long e = v * W; e < nelem; ++e) {
        float f = 0.f;
        for (int k = 0; k < n; ++k) f += s.sc[k] * to_f(((const T*)s.p[k])[e]);
        out[e] = from_f<T>(f);
      }
    }
  }
}

}  // namespace

#define DT_SWITCH(dtype, CALL)                    \
  if ((dtype) == HDMOE_F32) { using T = float; CALL; }   \
  else if ((dtype) == HDMOE_BF16) { using T = bf16; CALL; } \
  else return HDMOE_EDTYPE;                       \
  return hdmoe_launch_status();

#define L1D(kernel, n, ...) hipLaunchKernelGGL(kernel, dim3(grid_for(n)), dim3(TPB), 0, stream, __VA_ARGS__)

// Measurement aid (bench.py "attention", tools/exp_rate.py): issue rate of v_exp_f32, the instruction that bounds the attention kernels.
// Every thread runs 8 independent chains a = exp2(-a) (the negation is a source modifier: one v_exp_f32 per link), `iters` links each.
__global__ __launch_bounds__(256) void exp_rate_kernel(float* out, int iters) {
  float a[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = 0.25f + 0.001f * (float)((threadIdx.x + k) & 63);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = __builtin_amdgcn_exp2f(-a[k]);
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += a[k];
  out[(long)blockIdx.x * 256 + threadIdx.x] = s;
}

extern "C" {

int hdmoe_axpby(void* out, const void* x, const void* y, float a, float b, long n, int dtype, hipStream_t stream) {
  DT_SWITCH(dtype, if (n % VT<T>::W == 0 && al16(out) && al16(x) && al16(y)) L1D(axpby_vec_kernel<T>, n / VT<T>::W, (T*)out, (const T*)x, (const T*)y, a, b, n / VT<T>::W);
                   else L1D(axpby_kernel<T>, n, (T*)out, (const T*)x, (const T*)y, a, b, n))
}
/* EDM sampler stage pieces (fp32 latents; t: device float64 schedule of N + 1 values, idx: device stage index) */
int hdmoe_sched_pick(float* sigma, const double* t, const int* idx, int off, hipStream_t stream) {
  if (!sigma || !t || !idx) return HDMOE_EINVAL;
  hipLaunchKernelGGL(sched_pick_kernel, dim3(1), dim3(64), 0, stream, sigma, t, idx, off);
  return hdmoe_launch_status();
}
int hdmoe_idx_advance(int* idx, hipStream_t stream) {
  if (!idx) return HDMOE_EINVAL;
  hipLaunchKernelGGL(idx_advance_kernel, dim3(1), dim3(64), 0, stream, idx);
  return hdmoe_launch_status();
}
int hdmoe_heun_euler(float* xn, const float* xh, const float* den, const double* t, const int* idx, long n, hipStream_t stream) {
  if (!xn || !xh || !den || !t || !idx) return HDMOE_EINVAL;
  L1D(heun_euler_kernel, n, xn, xh, den, t, idx, n);
  return hdmoe_launch_status();
}
int hdmoe_heun_correct(float* out, const float* xh, const float* den, const float* xn, const float* den2, const double* t, const int* idx, long n,
                       hipStream_t stream) {
  if (!out || !xh || !den || !xn || !den2 || !t || !idx) return HDMOE_EINVAL;
  L1D(heun_correct_kernel, n, out, xh, den, xn, den2, t, idx, n);
  return hdmoe_launch_status();
}
int hdmoe_exp_rate(float* out, int blocks, int iters, hipStream_t stream) {
  if (!out || blocks < 1 || iters < 1) return HDMOE_EINVAL;
  hipLaunchKernelGGL(exp_rate_kernel, dim3(blocks), dim3(256), 0, stream, out, iters);
  return hdmoe_launch_status();
}
int hdmoe_sum_n(void* out, const void* const* srcs, const float* src_scale, int n, long nelem, int dtype, hipStream_t stream) {
  if (!out || !srcs || n < 1 || n > 16) return HDMOE_EINVAL;
  SumSrcs s;
  bool al = al16(out);
  for (int k = 0; k < 16; ++k) { s.p[k] = srcs[k < n ? k : 0]; s.sc[k] = (src_scale && k < n) ? src_scale[k] : 1.f; al = al && al16(s.p[k]); }
  if (!al) return HDMOE_EINVAL;
  DT_SWITCH(dtype, L1D(sum_n_kernel<T>, (nelem + VT<T>::W - 1) / VT<T>::W, (T*)out, s, n, (nelem + VT<T>::W - 1) / VT<T>::W, nelem))
}
int hdmoe_affine(void* out, const void* x, float a, float c, long n, int dtype, hipStream_t stream) {
  DT_SWITCH(dtype, L1D(affine_kernel<T>, n, (T*)out, (const T*)x, a, c, n))
}
int hdmoe_mul(void* out, const void* x, const void* y, long n, int dtype, hipStream_t stream) {
  DT_SWITCH(dtype, L1D(mul_kernel<T>, n, (T*)out, (const T*)x, (const T*)y, n))
}
int hdmoe_cast(void* out, const void* x, long n, int dt_in, int dt_out, hipStream_t stream) {
  if (dt_in == HDMOE_F32 && dt_out == HDMOE_BF16) L1D((cast_kernel<float, bf16>), n, (bf16*)out, (const float*)x, n);
  else if (dt_in == HDMOE_BF16 && dt_out == HDMOE_F32) L1D((cast_kernel<bf16, float>), n, (float*)out, (const bf16*)x, n);
  else if (dt_in == HDMOE_F32 && dt_out == HDMOE_F32) L1D((cast_kernel<float, float>), n, (float*)out, (const float*)x, n);
  else if (dt_in == HDMOE_BF16 && dt_out == HDMOE_BF16) L1D((cast_kernel<bf16, bf16>), n, (bf16*)out, (const bf16*)x, n);
  else if (dt_in == HDMOE_F16 && dt_out == HDMOE_F32) L1D(cast_h2f_kernel, n, (float*)out, (const _Float16*)x, n);
  else if (dt_in == HDMOE_F32 && dt_out == HDMOE_F16) L1D(cast_f2h_kernel, n, (_Float16*)out, (const float*)x, n);
  else return HDMOE_EDTYPE;
  return hdmoe_launch_status();
}
/* up == 0: y (N, Ho, Wo, C) = scale * stride-2 depthwise correlation of x (N, H, W, C) with outer(k, k), padding `pad`;
 * up == 1: the transposed operation (x is the small tensor).  taps: L <= 8 host floats (already normalised by the caller). */
int hdmoe_fir_resample(void* y, const void* x, const float* taps, int L, int pad, float scale, int up, int N, int H, int W, int Ho, int Wo, int C,
                       int dtype, hipStream_t stream) {
  if (!y || !x || !taps || L < 1 || L > 8 || pad < 0 || N < 0) return HDMOE_EINVAL;
  FirTaps f;
  for (int i = 0; i < 8; ++i) f.k[i] = i < L ? taps[i] : 0.f;
  const long total = (long)N * Ho * Wo * C;
  if (total == 0) return HDMOE_OK;
  DT_SWITCH(dtype, if (up) L1D(fir_up_kernel<T>, total, (T*)y, (const T*)x, f, L, pad, scale, H, W, Ho, Wo, C, total);
                   else L1D(fir_down_kernel<T>, total, (T*)y, (const T*)x, f, L, pad, scale, H, W, Ho, Wo, C, total))
}
int hdmoe_mp_silu_fwd(void* out, const void* x, long n, int dtype, hipStream_t stream) {
  DT_SWITCH(dtype, if (n % VT<T>::W == 0 && al16(out) && al16(x)) L1D(mp_silu_fwd_vec_kernel<T>, n / VT<T>::W, (T*)out, (const T*)x, n / VT<T>::W);
                   else L1D(mp_silu_fwd_kernel<T>, n, (T*)out, (const T*)x, n))
}
int hdmoe_mp_silu_bwd(void* dx, const void* dy, const void* x, long n, int dtype, hipStream_t stream) {
  DT_SWITCH(dtype, if (n % VT<T>::W == 0 && al16(dx) && al16(dy) && al16(x)) L1D(mp_silu_bwd_vec_kernel<T>, n / VT<T>::W, (T*)dx, (const T*)dy, (const T*)x, n / VT<T>::W);
                   else L1D(mp_silu_bwd_kernel<T>, n, (T*)dx, (const T*)dy, (const T*)x, n))
}
/* dx = sx * gx + dy * mp_silu'(x); 16-byte aligned, n % (16 / esz) == 0 */
int hdmoe_mp_silu_bwd_add(void* dx, const void* dy, const void* x, const void* gx, float sx, long n, int dtype, hipStream_t stream) {
  if (!dx || !dy || !x || !gx || !(al16(dx) && al16(dy) && al16(x) && al16(gx))) return HDMOE_EINVAL;
  DT_SWITCH(dtype, if (n % VT<T>::W == 0) L1D(mp_silu_bwd_add_vec_kernel<T>, n / VT<T>::W, (T*)dx, (const T*)dy, (const T*)x, (const T*)gx, sx, n / VT<T>::W);
                   else return HDMOE_EINVAL)
}
/* out = mp_cat(a, b) (weights wa, wb), out_h = mp_silu(out); Ca, Cb multiples of 16 / esz */
int hdmoe_cat2_silu_fwd(void* out, void* out_h, const void* a, const void* b, float wa, float wb, int Ca, int Cb, long rows, int dtype,
                        hipStream_t stream) {
  if (!(al16(out) && al16(out_h) && al16(a) && al16(b)) || !out || !out_h) return HDMOE_EINVAL;
  DT_SWITCH(dtype, if (Ca % VT<T>::W == 0 && Cb % VT<T>::W == 0)
                     L1D(cat2_silu_fwd_vec_kernel<T>, rows * (Ca + Cb) / VT<T>::W, (T*)out, (T*)out_h, (const T*)a, (const T*)b, wa, wb, Ca, Cb, rows);
                   else return HDMOE_EINVAL)
}
/* (da, db) = split((gcat + gh * mp_silu'(xcat)) * (wa | wb)); gcat may be NULL */
int hdmoe_cat2_silu_bwd(void* da, void* db, const void* gcat, const void* gh, const void* xcat, float wa, float wb, int Ca, int Cb,
                        long rows, int dtype, hipStream_t stream) {
  if (!(al16(da) && al16(db) && al16(gcat) && al16(gh) && al16(xcat)) || !gh || !xcat) return HDMOE_EINVAL;
  DT_SWITCH(dtype, if (Ca % VT<T>::W == 0 && Cb % VT<T>::W == 0)
                     L1D(cat2_silu_bwd_vec_kernel<T>, rows * (Ca + Cb) / VT<T>::W, (T*)da, (T*)db, (const T*)gcat, (const T*)gh, (const T*)xcat, wa, wb, Ca, Cb, rows);
                   else return HDMOE_EINVAL)
}
int hdmoe_sigmoid_fwd(void* out, const void* x, float a, long n, int dtype, hipStream_t stream) {
  DT_SWITCH(dtype, L1D(sigmoid_fwd_kernel<T>, n, (T*)out, (const T*)x, a, n))
}
int hdmoe_sigmoid_bwd(void* dx, const void* dy, const void* y, float a, long n, int dtype, hipStream_t stream) {
  DT_SWITCH(dtype, L1D(sigmoid_bwd_kernel<T>, n, (T*)dx, (const T*)dy, (const T*)y, a, n))
}
int hdmoe_film_silu_fwd(void* out, const void* u, const float* e, int N, long HW, int C, int dtype, hipStream_t stream) {
  const long n = (long)N * HW * C;
  DT_SWITCH(dtype, if (C % VT<T>::W == 0 && al16(out) && al16(u)) L1D(film_silu_fwd_vec_kernel<T>, n / VT<T>::W, (T*)out, (const T*)u, e, HW, C, n / VT<T>::W, 0u, 0u, nullptr, 0.f);
                   else L1D(film_silu_fwd_kernel<T>, n, (T*)out, (const T*)u, e, HW, C, n))
}
// FiLM + mp_silu + F.dropout in one pass (vectorised layouts only: C % (16/sizeof(T)) == 0)
int hdmoe_film_silu_drop_fwd(void* out, const void* u, const float* e, int N, long HW, int C, unsigned long long seed,
                             const unsigned long long* seed_dev, float p, int dtype, hipStream_t stream) {
  const long n = (long)N * HW * C;
  if (p < 0.f || p >= 1.f) return HDMOE_EINVAL;
  DT_SWITCH(dtype, if (C % VT<T>::W == 0 && al16(out) && al16(u)) L1D(film_silu_fwd_vec_kernel<T>, n / VT<T>::W, (T*)out, (const T*)u, e, HW, C, n / VT<T>::W,
                                                                     (uint32_t)seed, (uint32_t)(seed >> 32), seed_dev, p);
                   else return HDMOE_EINVAL)
}
int hdmoe_film_silu_drop_bwd(void* du, float* de, const void* da, const void* u, const float* e, int N, long HW, int C,
                             unsigned long long seed, const unsigned long long* seed_dev, float p, int dtype, hipStream_t stream) {
  if (N > 65535 || p < 0.f || p >= 1.f) return HDMOE_EINVAL;
  const int chunk = 256;
  dim3 grid(cdiv(HW, chunk), N);
  DT_SWITCH(dtype, if (chunk_fixed_ok<T>(C) && al16(du) && al16(da) && al16(u))
                     hipLaunchKernelGGL(film_silu_bwd_vec_kernel<T>, grid, dim3(TPB), C * sizeof(float), stream, (T*)du, de, (const T*)da, (const T*)u, e, HW, C, chunk,
                                        (uint32_t)seed, (uint32_t)(seed >> 32), seed_dev, p);
                   else return HDMOE_EINVAL)
}
int hdmoe_film_silu_bwd(void* du, float* de, const void* da, const void* u, const float* e, int N, long HW, int C,
                        int dtype, hipStream_t stream) {
  if (N > 65535) return HDMOE_EINVAL;
  const int chunk = 256;
  dim3 grid(cdiv(HW, chunk), N);
  DT_SWITCH(dtype, if (chunk_fixed_ok<T>(C) && al16(du) && al16(da) && al16(u))
                     hipLaunchKernelGGL(film_silu_bwd_vec_kernel<T>, grid, dim3(TPB), C * sizeof(float), stream, (T*)du, de, (const T*)da, (const T*)u, e, HW, C, chunk, 0u, 0u, nullptr, 0.f);
                   else hipLaunchKernelGGL(film_silu_bwd_kernel<T>, grid, dim3(TPB), C * sizeof(float), stream, (T*)du, de,
                                      (const T*)da, (const T*)u, e, HW, C, chunk))
}
int hdmoe_scale_rows_fwd(void* out, const void* x, const float* s, long rows, long L, int dtype, hipStream_t stream) {
  const long n = rows * L;
  DT_SWITCH(dtype, if (L % VT<T>::W == 0 && al16(out) && al16(x)) L1D(scale_rows_fwd_vec_kernel<T>, n / VT<T>::W, (T*)out, (const T*)x, s, L / VT<T>::W, n / VT<T>::W);
                   else L1D(scale_rows_fwd_kernel<T>, n, (T*)out, (const T*)x, s, L, n))
}
int hdmoe_scale_rows_bwd(void* dx, float* ds, const void* dy, const void* x, const float* s, long rows, long L,
                         int dtype, hipStream_t stream) {
  if (rows > 65535) return HDMOE_EINVAL;
  const int chunk = 8192;
  dim3 grid(cdiv(L, chunk), (unsigned)rows);
  DT_SWITCH(dtype, if (L % VT<T>::W == 0 && al16(dx) && al16(dy) && al16(x)) {
                     const long Lv = L / VT<T>::W;
                     const int vchunk = 2048;                    // 16-byte vectors per workgroup: 8 per thread
                     hipLaunchKernelGGL(scale_rows_bwd_vec_kernel<T>, dim3(cdiv(Lv, vchunk), (unsigned)rows), dim3(TPB), 0, stream, (T*)dx, ds,
                                        (const T*)dy, (const T*)x, s, Lv, vchunk);
                   } else hipLaunchKernelGGL(scale_rows_bwd_kernel<T>, grid, dim3(TPB), 0, stream, (T*)dx, ds, (const T*)dy,
                                      (const T*)x, s, L, chunk))
}
int hdmoe_cat2_fwd(void* out, const void* a, const void* b, float wa, float wb, int Ca, int Cb, long rows, int dtype,
                   hipStream_t stream) {
  DT_SWITCH(dtype, if (Ca % VT<T>::W == 0 && Cb % VT<T>::W == 0 && al16(out) && al16(a) && al16(b))
                     L1D(cat2_fwd_vec_kernel<T>, rows * (Ca + Cb) / VT<T>::W, (T*)out, (const T*)a, (const T*)b, wa, wb, Ca, Cb, rows);
                   else L1D(cat2_fwd_kernel<T>, rows * (Ca + Cb), (T*)out, (const T*)a, (const T*)b, wa, wb, Ca, Cb, rows))
}
int hdmoe_cat2_bwd(void* da, void* db, const void* dout, float wa, float wb, int Ca, int Cb, long rows, int dtype,
                   hipStream_t stream) {
  DT_SWITCH(dtype, if (Ca % VT<T>::W == 0 && Cb % VT<T>::W == 0 && al16(da) && al16(db) && al16(dout))
                     L1D(cat2_bwd_vec_kernel<T>, rows * (Ca + Cb) / VT<T>::W, (T*)da, (T*)db, (const T*)dout, wa, wb, Ca, Cb, rows);
                   else L1D(cat2_bwd_kernel<T>, rows * (Ca + Cb), (T*)da, (T*)db, (const T*)dout, wa, wb, Ca, Cb, rows))
}
int hdmoe_pool2(void* out, const void* x, int N, int Ho, int Wo, int C, float scale, int dtype, hipStream_t stream) {
  const long n = (long)N * Ho * Wo * C;
  DT_SWITCH(dtype, if (C % VT<T>::W == 0 && al16(out) && al16(x)) L1D(pool2_vec_kernel<T>, n / VT<T>::W, (T*)out, (const T*)x, Ho, Wo, C / VT<T>::W, scale, n / VT<T>::W);
                   else L1D(pool2_kernel<T>, n, (T*)out, (const T*)x, Ho, Wo, C, scale, n))
}
int hdmoe_upsample2(void* out, const void* x, int N, int Ho, int Wo, int C, float scale, int dtype, hipStream_t stream) {
  if ((Ho | Wo) & 1) return HDMOE_EINVAL;
  const long n = (long)N * Ho * Wo * C;
  DT_SWITCH(dtype, if (C % VT<T>::W == 0 && al16(out) && al16(x)) L1D(upsample2_vec_kernel<T>, n / VT<T>::W, (T*)out, (const T*)x, Ho, Wo, C / VT<T>::W, scale, n / VT<T>::W);
                   else L1D(upsample2_kernel<T>, n, (T*)out, (const T*)x, Ho, Wo, C, scale, n))
}
int hdmoe_seq_reduce(float* out, const void* x, int N, long S, int C, float scale, int dtype, hipStream_t stream) {
  if (N > 65535 || C > 8192) return HDMOE_EINVAL;
  // accumulates into `out` (callers pass zeros or a running sum); deterministic: no atomics, fixed summation order
  DT_SWITCH(dtype, if (chunk_fixed_ok<T>(C) && al16(x)) hipLaunchKernelGGL(seq_reduce_det_vec_kernel<T>, dim3(N), dim3(256),
                                                                           (size_t)(256 / (C / VT<T>::W)) * C * sizeof(float), stream,
                                                                           out, (const T*)x, S, C, scale);
                   else hipLaunchKernelGGL(seq_reduce_det_kernel<T>, dim3(cdiv(C, TPB), N), dim3(TPB), 0, stream, out, (const T*)x, S, C, scale))
}
int hdmoe_seq_bcast_add(void* out, const void* x, const float* t, int N, long S, int C, float scale, int dtype,
                        hipStream_t stream) {
  const long n = (long)N * S * C;
  DT_SWITCH(dtype, L1D(seq_bcast_add_kernel<T>, n, (T*)out, (const T*)x, t, S, C, scale, n))
}
int hdmoe_bias_add(void* out, const void* x, const float* bias, long rows, long L, int dtype, hipStream_t stream) {
  const long n = rows * L;
  DT_SWITCH(dtype, L1D(bias_add_kernel<T>, n, (T*)out, (const T*)x, bias, L, n))
}
int hdmoe_colsum(float* out, const void* dy, long rows, long L, int dtype, hipStream_t stream) {
  const int rchunk = 64;
  if (cdiv(rows, rchunk) > 65535) return HDMOE_EINVAL;
  dim3 grid(cdiv(L, TPB), cdiv(rows, rchunk));
  DT_SWITCH(dtype, hipLaunchKernelGGL(colsum_kernel<T>, grid, dim3(TPB), 0, stream, out, (const T*)dy, rows, L, rchunk))
}
int hdmoe_lerp_param_fwd(void* out, const void* a, const void* b, const float* alpha, long n, int dtype, hipStream_t stream) {
  DT_SWITCH(dtype, L1D(lerp_param_fwd_kernel<T>, n, (T*)out, (const T*)a, (const T*)b, alpha, n))
}
int hdmoe_lerp_param_bwd(void* da, void* db, float* dalpha, const void* g, const void* a, const void* b,
                         const float* alpha, long n, int dtype, hipStream_t stream) {
  DT_SWITCH(dtype, if (n % VT<T>::W == 0 && al16(da) && al16(db) && al16(g) && al16(a) && al16(b)) {
                     long blocks = (n / VT<T>::W + TPB - 1) / TPB; if (blocks > 1024) blocks = 1024;   // (one atomic per block)
                     hipLaunchKernelGGL(lerp_param_bwd_vec_kernel<T>, dim3((unsigned)blocks), dim3(TPB), 0, stream, (T*)da, (T*)db, dalpha, (const T*)g,
                                        (const T*)a, (const T*)b, alpha, n / VT<T>::W);
                   } else L1D(lerp_param_bwd_kernel<T>, n, (T*)da, (T*)db, dalpha, (const T*)g, (const T*)a, (const T*)b, alpha, n))
}
int hdmoe_gate_mix_fwd(void* out, float* gate, const void* logits, const void* U, const void* A, long rows, int C,
                       int dtype, hipStream_t stream) {
  DT_SWITCH(dtype, {
    const int lpr = C / VT<T>::W;
    if (C % VT<T>::W == 0 && lpr >= 1 && lpr <= 64 && (lpr & (lpr - 1)) == 0 && al16(out) && al16(U) && al16(A))
      L1D(gate_mix_fwd_vec_kernel<T>, rows * lpr, (T*)out, gate, (const T*)logits, (const T*)U, (const T*)A, lpr, rows * lpr);
    else L1D(gate_mix_fwd_kernel<T>, rows * C, (T*)out, gate, (const T*)logits, (const T*)U, (const T*)A, C, rows);
  })
}
int hdmoe_gate_mix_bwd(void* dU, void* dA, void* dlogits, const void* dout, const float* dgate, const float* gate,
                       const void* U, const void* A, long rows, int C, int dtype, hipStream_t stream) {
  DT_SWITCH(dtype, {
    const int lpr = C / VT<T>::W;
    if (C % VT<T>::W == 0 && lpr >= 1 && lpr <= 64 && (lpr & (lpr - 1)) == 0 && al16(dU) && al16(dA) && al16(dout) && al16(U) && al16(A))
      L1D(gate_mix_bwd_vec_kernel<T>, (rows * lpr + 63) / 64 * 64, (T*)dU, (T*)dA, (T*)dlogits, (const T*)dout, dgate, gate, (const T*)U,
          (const T*)A, lpr, rows * lpr, (rows * lpr + 63) / 64 * 64);
    else L1D(gate_mix_bwd_kernel<T>, rows, (T*)dU, (T*)dA, (T*)dlogits, (const T*)dout, dgate, gate, (const T*)U, (const T*)A, C, rows);
  })
}
int hdmoe_softmax_rows_fwd(float* out, const float* x, long rows, int C, float scale, hipStream_t stream) {
  L1D(softmax_rows_fwd_kernel, rows, out, x, C, scale, rows);
  return hdmoe_launch_status();
}
int hdmoe_softmax_rows_bwd(float* dx, const float* dy, const float* y, long rows, int C, float scale, hipStream_t stream) {
  L1D(softmax_rows_bwd_kernel, rows, dx, dy, y, C, scale, rows);
  return hdmoe_launch_status();
}
int hdmoe_nchw_to_nhwc(void* out, const float* x, const float* s, int N, int C, long HW, int dtype, hipStream_t stream) {
  const long n = (long)N * C * HW;
  DT_SWITCH(dtype, L1D(nchw_to_nhwc_kernel<T>, n, (T*)out, x, s, C, HW, n))
}
int hdmoe_nhwc_to_nchw(float* out, const void* F, const float* sf, const float* x, const float* sx, int N, int C, long HW,
                       int dtype, hipStream_t stream) {
  const long n = (long)N * C * HW;
  DT_SWITCH(dtype, L1D(nhwc_to_nchw_kernel<T>, n, out, (const T*)F, sf, x, sx, C, HW, n))
}
int hdmoe_patch_relayout(void* out, const void* in, int N, int H, int W, int C, int p, int hp, int wp, int order,
                         int to_img, int dtype, hipStream_t stream) {
  if (hp * p < H || wp * p < W) return HDMOE_EINVAL;
  const long n = (long)N * H * W * C;
  if (!to_img && order == 1 && al16(out) && (dtype == HDMOE_BF16 || dtype == HDMOE_F32) && p % (dtype == HDMOE_BF16 ? 8 : 4) == 0) {
    const long nv = (long)N * hp * wp * C * p * p / (dtype == HDMOE_BF16 ? 8 : 4);      // covers the padded tokens too (zeros)
    DT_SWITCH(dtype, L1D(patch_to_tokens_o1_vec_kernel<T>, nv, (T*)out, (const T*)in, H, W, C, p, hp, wp, nv))
  }
  if (al16(out) && al16(in) && C % (dtype == HDMOE_BF16 ? 8 : 4) == 0 && (dtype == HDMOE_BF16 || dtype == HDMOE_F32)) {
    const long nv = n / (dtype == HDMOE_BF16 ? 8 : 4);
    if (to_img) { DT_SWITCH(dtype, L1D((patch_relayout_vec_kernel<T, true>), nv, (T*)out, (const T*)in, H, W, C, p, hp, wp, order, nv)) }
    else { DT_SWITCH(dtype, L1D((patch_relayout_vec_kernel<T, false>), nv, (T*)out, (const T*)in, H, W, C, p, hp, wp, order, nv)) }
  }
  if (to_img) { DT_SWITCH(dtype, L1D((patch_relayout_kernel<T, true>), n, (T*)out, (const T*)in, H, W, C, p, hp, wp, order, n)) }
  else { DT_SWITCH(dtype, L1D((patch_relayout_kernel<T, false>), n, (T*)out, (const T*)in, H, W, C, p, hp, wp, order, n)) }
}
int hdmoe_fourier(float* out, const float* x, const float* freqs, const float* phases, int B, int F, hipStream_t stream) {
  const long n = (long)B * F;
  L1D(fourier_kernel, n, out, x, freqs, phases, F, n);
  return hdmoe_launch_status();
}
int hdmoe_edm_coeffs(float* coef, const float* sigma, int nsig, float sigma_data, int B, hipStream_t stream) {
  if (nsig != 1 && nsig != B) return HDMOE_EINVAL;
  L1D(edm_coeffs_kernel, B, coef, sigma, nsig, sigma_data, B);
  return hdmoe_launch_status();
}
int hdmoe_sigmoid_scaling(float* sv, float* su, float* pair, const float* c_noise, float tp, float soft, int B,
                          hipStream_t stream) {
  L1D(sigmoid_scaling_kernel, B, sv, su, pair, c_noise, tp, soft, B);
  return hdmoe_launch_status();
}
int hdmoe_adaln_fwd(float* out, const float* x, const float* cond, long B, int F, hipStream_t stream) {
  L1D(adaln_fwd_kernel, B * F, out, x, cond, F, B * F);
  return hdmoe_launch_status();
}
int hdmoe_adaln_bwd(float* dx, float* dcond, const float* g, const float* x, const float* cond, long B, int F, hipStream_t stream) {
  L1D(adaln_bwd_kernel, B * F, dx, dcond, g, x, cond, F, B * F);
  return hdmoe_launch_status();
}
int hdmoe_take_col_pos_fwd(float* out, const float* w, long B, int E, int e, hipStream_t stream) {
  if (e < 0 || e >= E) return HDMOE_EINVAL;
  L1D(take_col_pos_fwd_kernel, B, out, w, E, e, B);
  return hdmoe_launch_status();
}
int hdmoe_take_col_pos_bwd(float* dw, const float* g, const float* w, long B, int E, int e, hipStream_t stream) {
  if (e < 0 || e >= E) return HDMOE_EINVAL;
  L1D(take_col_pos_bwd_kernel, B, dw, g, w, E, e, B);
  return hdmoe_launch_status();
}
int hdmoe_dropout(void* out, const void* x, unsigned long long seed, const unsigned long long* seed_dev, float p, long n, int dtype,
                  hipStream_t stream) {
  if (p < 0.f || p >= 1.f) return HDMOE_EINVAL;
  DT_SWITCH(dtype, L1D(dropout_kernel<T>, (n + 3) / 4, (T*)out, (const T*)x, (uint32_t)seed, (uint32_t)(seed >> 32), seed_dev, p, n))
}
int hdmoe_randn(float* out, unsigned long long seed, const unsigned long long* seed_dev, float scale, long n, hipStream_t stream) {
  L1D(randn_kernel, (n + 3) / 4, out, (uint32_t)seed, (uint32_t)(seed >> 32), seed_dev, scale, n);
  return hdmoe_launch_status();
}
int hdmoe_seed_advance(unsigned long long* seed_dev, hipStream_t stream) {
  hipLaunchKernelGGL(seed_advance_kernel, dim3(1), dim3(1), 0, stream, seed_dev);
  return hdmoe_launch_status();
}

}  // extern "C"
