// K6: normalisation kernels (fwd + bwd), NHWC / [rows][C]; statistics always in fp32.
//   pixel-norm  : normalize(x, dim=[1])            reference models/model_internals.py:8-30, model_components.py:238
//   GroupNorm   : nn.GroupNorm(G, C) (+ReLU / +mp_silu fused)   model_components.py:102-109, :491, :530
//   LayerNorm   : nn.LayerNorm(C)                  model_components.py:495-496, :645
#include <stdlib.h>
#include "common.h"
#include "hdmoe.h"

namespace {

constexpr int TPB = 256;
#define GRID_STRIDE(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)
static inline unsigned grid_for(long n) {
  long b = (n + TPB - 1) / TPB;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// ---------------------------------------------------------------- pixel norm (one thread per pixel row)
template <typename T>
__global__ void pixelnorm_fwd_kernel(T* xn, T* h, const T* x, int C, long rows) {
  const float rc = rsqrtf((float)C);
  GRID_STRIDE(r, rows) {
    const T* p = x + r * C;
    float ss = 0.f;
    for (int c = 0; c < C; ++c) { const float v = to_f(p[c]); ss += v * v; }
    const float inv = 1.f / (1e-4f + sqrtf(ss) * rc);
    for (int c = 0; c < C; ++c) {
      const float v = to_f(p[c]) * inv;
      xn[r * C + c] = from_f<T>(v);
      if (h) h[r * C + c] = from_f<T>(mp_silu_f(v));
    }
  }
}
// g = dxn + dh * mp_silu'(xn);  dx = g/d - x * (sum g*x) * rc / (d^2 * n)
template <typename T>
__global__ void pixelnorm_bwd_kernel(T* dx, const T* dxn, const T* dh, const T* x, int C, long rows, float sx) {
  const float rc = rsqrtf((float)C);
  GRID_STRIDE(r, rows) {
    const T* p = x + r * C;
    float ss = 0.f;
    for (int c = 0; c < C; ++c) { const float v = to_f(p[c]); ss += v * v; }
    const float nrm = sqrtf(ss);
    const float d = 1e-4f + nrm * rc, inv = 1.f / d;
    float dot = 0.f;
    for (int c = 0; c < C; ++c) {
      const float v = to_f(p[c]);
      float g = dxn ? sx * to_f(dxn[r * C + c]) : 0.f;
      if (dh) g += to_f(dh[r * C + c]) * mp_silu_grad_f(v * inv);
      dot += g * v;
    }
    const float k2 = nrm > 0.f ? dot * rc * inv * inv / nrm : 0.f;
    for (int c = 0; c < C; ++c) {
      const float v = to_f(p[c]);
      float g = dxn ? sx * to_f(dxn[r * C + c]) : 0.f;
      if (dh) g += to_f(dh[r * C + c]) * mp_silu_grad_f(v * inv);
      dx[r * C + c] = from_f<T>(g * inv - k2 * v);
    }
  }
}

// ---------------------------------------------------------------- GroupNorm
DEVI float act_f(float z, int act) { return act == 1 ? fmaxf(z, 0.f) : (act == 2 ? mp_silu_f(z) : z); }
DEVI float act_grad_f(float z, int act) { return act == 1 ? (z > 0.f ? 1.f : 0.f) : (act == 2 ? mp_silu_grad_f(z) : 1.f); }

// per-group block reduction: thread t owns group t % G (requires blockDim % G == 0); result for group g in red[g]
DEVI void group_reduce(float v, int G, float* part, float* red) {
  __syncthreads();
  part[threadIdx.x] = v;
  __syncthreads();
  if ((int)threadIdx.x < G) {
    float s = 0.f;
    for (int t = threadIdx.x; t < (int)blockDim.x; t += G) s += part[t];
    red[threadIdx.x] = s;
  }
  __syncthreads();
}

// one block per sample; items = (position s, group g) pairs, item -> Cg contiguous channels; two-pass variance
template <typename T>
__global__ __launch_bounds__(512) void groupnorm_stats_kernel(float* mean, float* rstd, const T* x, long S, int C, int G, float eps) {
  __shared__ float part[512];
  __shared__ float red[64];
  const int n = blockIdx.x, Cg = C / G;
  const T* xs = x + (long)n * S * C;
  const long items = S * G;
  float acc = 0.f;
  for (long it = threadIdx.x; it < items; it += blockDim.x) {
    const T* p = xs + (it / G) * C + (it % G) * Cg;
    for (int c = 0; c < Cg; ++c) acc += to_f(p[c]);
  }
  group_reduce(acc, G, part, red);
  const float m = red[threadIdx.x % G] / (float)(S * Cg);
  acc = 0.f;
  for (long it = threadIdx.x; it < items; it += blockDim.x) {
    const T* p = xs + (it / G) * C + (it % G) * Cg;
    for (int c = 0; c < Cg; ++c) { const float d = to_f(p[c]) - m; acc += d * d; }
  }
  group_reduce(acc, G, part, red);
  if ((int)threadIdx.x < G) {
    mean[(long)n * G + threadIdx.x] = m;
    rstd[(long)n * G + threadIdx.x] = rsqrtf(red[threadIdx.x] / (float)(S * Cg) + eps);
  }
}
template <typename T>
__global__ void groupnorm_apply_kernel(T* y, const T* x, const float* gamma, const float* beta, const float* mean,
                                       const float* rstd, long S, int C, int G, int act, long n) {
  const int Cg = C / G;
  GRID_STRIDE(i, n) {
    const long row = i / C; const int c = (int)(i - row * C);
    const long sg = (row / S) * G + c / Cg;
    const float z = (to_f(x[i]) - mean[sg]) * rstd[sg] * gamma[c] + beta[c];
    y[i] = from_f<T>(act_f(z, act));
  }
}
// bwd pass 1: per (sample, group) s1 = sum dz*gamma, s2 = sum dz*gamma*xhat ; per-channel dgamma/dbeta (atomics)
template <typename T>
__global__ __launch_bounds__(512) void groupnorm_bwd_stats_kernel(float* s1, float* s2, float* dgamma, float* dbeta, const T* dy,
                                                                 const T* x, const float* gamma, const float* beta,
                                                                 const float* mean, const float* rstd, long S, int C, int G, int act) {
  extern __shared__ float sm[];          // [2*C] per-channel partials
  __shared__ float part[512];
  __shared__ float red[64];
  const int n = blockIdx.x, Cg = C / G;
  for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) sm[c] = 0.f;
  __syncthreads();
  const long base = (long)n * S * C;
  const long items = S * G;
  const int g = threadIdx.x % G;
  const float m = mean[(long)n * G + g], rs = rstd[(long)n * G + g];
  float a1 = 0.f, a2 = 0.f;
  for (long it = threadIdx.x; it < items; it += blockDim.x) {
    const long off = base + (it / G) * C + g * Cg;
    for (int c = 0; c < Cg; ++c) {
      const int ch = g * Cg + c;
      const float xh = (to_f(x[off + c]) - m) * rs;
      const float dz = to_f(dy[off + c]) * act_grad_f(xh * gamma[ch] + beta[ch], act);
      a1 += dz * gamma[ch];
      a2 += dz * gamma[ch] * xh;
      atomicAdd(&sm[ch], dz * xh);
      atomicAdd(&sm[C + ch], dz);
    }
  }
  group_reduce(a1, G, part, red);
  if ((int)threadIdx.x < G) s1[(long)n * G + threadIdx.x] = red[threadIdx.x];
  group_reduce(a2, G, part, red);
  if ((int)threadIdx.x < G) s2[(long)n * G + threadIdx.x] = red[threadIdx.x];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    atomicAdd(&dgamma[c], sm[c]);
    atomicAdd(&dbeta[c], sm[C + c]);
  }
}
template <typename T>
__global__ void groupnorm_bwd_apply_kernel(T* dx, const T* dy, const T* x, const float* gamma, const float* beta, const float* mean,
                                           const float* rstd, const float* s1, const float* s2, long S, int C, int G, int act, long n) {
  const int Cg = C / G;
  const float invm = 1.f / (float)(S * Cg);
  GRID_STRIDE(i, n) {
    const long row = i / C; const int c = (int)(i - row * C);
    const long sg = (row / S) * G + c / Cg;
    const float rs = rstd[sg];
    const float xh = (to_f(x[i]) - mean[sg]) * rs;
    const float dz = to_f(dy[i]) * act_grad_f(xh * gamma[c] + beta[c], act);
    dx[i] = from_f<T>(rs * (dz * gamma[c] - invm * (s1[sg] + xh * s2[sg])));
  }
}

// ---------------------------------------------------------------- LayerNorm (one thread per row, C <= ~1024)
template <typename T>
__global__ void layernorm_fwd_kernel(T* y, float* mean, float* rstd, const T* x, const float* gamma, const float* beta, int C, float eps, long rows) {
  GRID_STRIDE(r, rows) {
    const T* p = x + r * C;
    float m = 0.f;
    for (int c = 0; c < C; ++c) m += to_f(p[c]);
    m /= (float)C;
    float v = 0.f;
    for (int c = 0; c < C; ++c) { const float d = to_f(p[c]) - m; v += d * d; }
    const float rs = rsqrtf(v / (float)C + eps);
    mean[r] = m; rstd[r] = rs;
    for (int c = 0; c < C; ++c) y[r * C + c] = from_f<T>((to_f(p[c]) - m) * rs * gamma[c] + beta[c]);
  }
}
template <typename T>
__global__ void layernorm_bwd_kernel(T* dx, float* dgamma, float* dbeta, const T* dy, const T* x, const float* gamma,
                                     const float* mean, const float* rstd, int C, long rows) {
  extern __shared__ float sm[];     // [2*C]
  for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) sm[c] = 0.f;
  __syncthreads();
  GRID_STRIDE(r, rows) {
    const float m = mean[r], rs = rstd[r];
    float a1 = 0.f, a2 = 0.f;
    for (int c = 0; c < C; ++c) {
      const float g = to_f(dy[r * C + c]), xh = (to_f(x[r * C + c]) - m) * rs;
      a1 += g * gamma[c];
      a2 += g * gamma[c] * xh;
      atomicAdd(&sm[c], g * xh);
      atomicAdd(&sm[C + c], g);
    }
    const float ic = 1.f / (float)C;
    for (int c = 0; c < C; ++c) {
      const float g = to_f(dy[r * C + c]), xh = (to_f(x[r * C + c]) - m) * rs;
      dx[r * C + c] = from_f<T>(rs * (g * gamma[c] - ic * (a1 + xh * a2)));
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    atomicAdd(&dgamma[c], sm[c]);
    atomicAdd(&dbeta[c], sm[C + c]);
  }
}


// =====================================================================================================================
// Vectorised variants (16-byte accesses, guide G13).  A row of C channels is spread over LPR = pow2 >= C/VW lanes, one
// 16-byte vector per lane; row reductions are xor-shuffles inside the LPR-lane group (no LDS).
// =====================================================================================================================
template <int LPR> DEVI float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T, int LPR>
__global__ __launch_bounds__(256) void pixelnorm_fwd_vec_kernel(T* xn, T* h, const T* x, int C, long rows) {
  constexpr int W = VT<T>::W;
  const int sub = threadIdx.x % LPR;
  const bool act = sub * W < C;
  const float rc = rsqrtf((float)C);
  for (long r = ((long)blockIdx.x * blockDim.x + threadIdx.x) / LPR; r < rows; r += (long)gridDim.x * blockDim.x / LPR) {
    float v[W];
    float ss = 0.f;
    if (act) {
      vload<T>(v, x + r * C + sub * W);
#pragma unroll
      for (int j = 0; j < W; ++j) ss += v[j] * v[j];
    }
    ss = group_sum<LPR>(ss);
    const float inv = 1.f / (1e-4f + sqrtf(ss) * rc);
    if (act) {
      float o[W];
#pragma unroll
      for (int j = 0; j < W; ++j) o[j] = v[j] * inv;
      vstore<T>(xn + r * C + sub * W, o);
      if (h) {
#pragma unroll
        for (int j = 0; j < W; ++j) o[j] = mp_silu_f(o[j]);
        vstore<T>(h + r * C + sub * W, o);
      }
    }
  }
}
template <typename T, int LPR>
__global__ __launch_bounds__(256) void pixelnorm_bwd_vec_kernel(T* dx, const T* dxn, const T* dh, const T* x, int C, long rows, float sx) {
  constexpr int W = VT<T>::W;
  const int sub = threadIdx.x % LPR;
  const bool act = sub * W < C;
  const float rc = rsqrtf((float)C);
  for (long r = ((long)blockIdx.x * blockDim.x + threadIdx.x) / LPR; r < rows; r += (long)gridDim.x * blockDim.x / LPR) {
    float v[W], g[W], t[W];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < W; ++j) { v[j] = 0.f; g[j] = 0.f; }
    if (act) {
      vload<T>(v, x + r * C + sub * W);
      if (dxn) {
        vload<T>(g, dxn + r * C + sub * W);
#pragma unroll
        for (int j = 0; j < W; ++j) g[j] *= sx;
      }
#pragma unroll
      for (int j = 0; j < W; ++j) ss += v[j] * v[j];
    }
    ss = group_sum<LPR>(ss);
    const float nrm = sqrtf(ss);
    const float inv = 1.f / (1e-4f + nrm * rc);
    float dot = 0.f;
    if (act) {
      if (dh) {
        vload<T>(t, dh + r * C + sub * W);
#pragma unroll
        for (int j = 0; j < W; ++j) g[j] += t[j] * mp_silu_grad_f(v[j] * inv);
      }
#pragma unroll
      for (int j = 0; j < W; ++j) dot += g[j] * v[j];
    }
    dot = group_sum<LPR>(dot);
    const float k2 = nrm > 0.f ? dot * rc * inv * inv / nrm : 0.f;
    if (act) {
#pragma unroll
      for (int j = 0; j < W; ++j) t[j] = g[j] * inv - k2 * v[j];
      vstore<T>(dx + r * C + sub * W, t);
    }
  }
}

template <typename T, int LPR>
__global__ __launch_bounds__(256) void layernorm_fwd_vec_kernel(T* y, float* mean, float* rstd, const T* x, const float* gamma,
                                                               const float* beta, int C, float eps, long rows) {
  constexpr int W = VT<T>::W;
  const int sub = threadIdx.x % LPR;
  const bool act = sub * W < C;
  float gm[W], bt[W];
#pragma unroll
  for (int j = 0; j < W; ++j) { gm[j] = act ? gamma[sub * W + j] : 0.f; bt[j] = act ? beta[sub * W + j] : 0.f; }
  for (long r = ((long)blockIdx.x * blockDim.x + threadIdx.x) / LPR; r < rows; r += (long)gridDim.x * blockDim.x / LPR) {
    float v[W];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < W; ++j) v[j] = 0.f;
    if (act) vload<T>(v, x + r * C + sub * W);
#pragma unroll
    for (int j = 0; j < W; ++j) s += v[j];
    const float m = group_sum<LPR>(s) / (float)C;
    float q = 0.f;
    if (act) {
#pragma unroll
      for (int j = 0; j < W; ++j) { const float d = v[j] - m; q += d * d; }
    }
    const float rs = rsqrtf(group_sum<LPR>(q) / (float)C + eps);
    if (sub == 0) { mean[r] = m; rstd[r] = rs; }
    if (act) {
#pragma unroll
      for (int j = 0; j < W; ++j) v[j] = (v[j] - m) * rs * gm[j] + bt[j];
      vstore<T>(y + r * C + sub * W, v);
    }
  }
}
template <typename T, int LPR>
__global__ __launch_bounds__(256) void layernorm_bwd_vec_kernel(T* dx, float* dgamma, float* dbeta, const T* dy, const T* x,
                                                               const float* gamma, const float* mean, const float* rstd, int C, long rows) {
  constexpr int W = VT<T>::W;
  extern __shared__ float sm[];     // [2*C]
  for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) sm[c] = 0.f;
  __syncthreads();
  const int sub = threadIdx.x % LPR;
  const bool act = sub * W < C;
  float gm[W], dg[W], db[W];
#pragma unroll
  for (int j = 0; j < W; ++j) { gm[j] = act ? gamma[sub * W + j] : 0.f; dg[j] = 0.f; db[j] = 0.f; }
  const float ic = 1.f / (float)C;
  for (long r = ((long)blockIdx.x * blockDim.x + threadIdx.x) / LPR; r < rows; r += (long)gridDim.x * blockDim.x / LPR) {
    float v[W], g[W];
#pragma unroll
    for (int j = 0; j < W; ++j) { v[j] = 0.f; g[j] = 0.f; }
    const float m = mean[r], rs = rstd[r];
    float a1 = 0.f, a2 = 0.f;
    if (act) {
      vload<T>(v, x + r * C + sub * W);
      vload<T>(g, dy + r * C + sub * W);
#pragma unroll
      for (int j = 0; j < W; ++j) {
        v[j] = (v[j] - m) * rs;                      // xhat
        a1 += g[j] * gm[j]; a2 += g[j] * gm[j] * v[j];
        dg[j] += g[j] * v[j]; db[j] += g[j];
      }
    }
    a1 = group_sum<LPR>(a1); a2 = group_sum<LPR>(a2);
    if (act) {
#pragma unroll
      for (int j = 0; j < W; ++j) g[j] = rs * (g[j] * gm[j] - ic * (a1 + v[j] * a2));
      vstore<T>(dx + r * C + sub * W, g);
    }
  }
  if (act) {
#pragma unroll
    for (int j = 0; j < W; ++j) { atomicAdd(&sm[sub * W + j], dg[j]); atomicAdd(&sm[C + sub * W + j], db[j]); }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) { atomicAdd(&dgamma[c], sm[c]); atomicAdd(&dbeta[c], sm[C + c]); }
}

// block reduction per group for the vectorised GroupNorm kernels: thread t carries channel chunk (t % cv); first a strided
// tree over the 512/cv threads of each chunk, then the cv chunk sums are folded into their groups.  Result in red[g].
DEVI void vec_group_reduce(float v, int cv, int W, int Cg, int G, float* part, float* red) {
  __syncthreads();
  part[threadIdx.x] = v;
  __syncthreads();
  for (int stride = (int)blockDim.x >> 1; stride >= cv; stride >>= 1) {
    if ((int)threadIdx.x < stride) part[threadIdx.x] += part[threadIdx.x + stride];
    __syncthreads();
  }
  if ((int)threadIdx.x < G) {
    float s = 0.f;
    const int per = Cg / W;                                  // chunks per group
    for (int c = threadIdx.x * per; c < (threadIdx.x + 1) * per; ++c) s += part[c];
    red[threadIdx.x] = s;
  }
  __syncthreads();
}

// ---- GroupNorm, vectorised: one 512-thread block per sample; a thread's vectors all belong to one group and one channel
// chunk (requires 512 % (C/W) == 0 and (C/G) % W == 0)
template <typename T>
__global__ __launch_bounds__(512) void groupnorm_stats_vec_kernel(float* mean, float* rstd, const T* x, long S, int C, int G, float eps) {
  constexpr int W = VT<T>::W;
  __shared__ float part[512];
  __shared__ float red[64];
  const int n = blockIdx.x, Cg = C / G, cv = C / W;
  const T* xs = x + (long)n * S * C;
  const long nv = S * cv;
  const int mych = (threadIdx.x % cv) * W;
  const int g = mych / Cg;
  // group_reduce expects thread t to own group t % G; remap through a per-thread slot instead: reduce over all threads of group g
  float acc = 0.f;
  for (long v = threadIdx.x; v < nv; v += blockDim.x) {
    float f[W];
    vload<T>(f, xs + v * W);
#pragma unroll
    for (int j = 0; j < W; ++j) acc += f[j];
  }
  vec_group_reduce(acc, cv, W, Cg, G, part, red);
  const float m = red[g] / (float)(S * Cg);
  acc = 0.f;
  for (long v = threadIdx.x; v < nv; v += blockDim.x) {
    float f[W];
    vload<T>(f, xs + v * W);
#pragma unroll
    for (int j = 0; j < W; ++j) { const float d = f[j] - m; acc += d * d; }
  }
  const float mkeep = m;
  vec_group_reduce(acc, cv, W, Cg, G, part, red);
  if ((int)threadIdx.x % cv == 0 || true) {
    if ((int)threadIdx.x < cv && (mych % Cg) == 0) {        // first channel chunk of each group writes its statistics
      mean[(long)n * G + g] = mkeep;
      rstd[(long)n * G + g] = rsqrtf(red[g] / (float)(S * Cg) + eps);
    }
  }
}
// Small batches: one 512-thread block per sample leaves most of the chip idle (32 samples = 32 blocks).  Split each sample's
// rows over `parts` blocks: every block reduces its row range to per-group (mean, M2) with the same two-pass scheme, and a
// second kernel merges the partials in a fixed order (Chan et al.), so the result stays deterministic.
template <typename T>
__global__ __launch_bounds__(512) void groupnorm_part_stats_vec_kernel(float* ws, const T* x, long S, int C, int G, int parts) {
  constexpr int W = VT<T>::W;
  __shared__ float part[512];
  __shared__ float red[64];
  const int n = blockIdx.x, pi = blockIdx.y, Cg = C / G, cv = C / W;
  const long rpp = (S + parts - 1) / parts;
  const long r0 = min((long)pi * rpp, S), r1 = min(r0 + rpp, S);
  const T* xs = x + ((long)n * S + r0) * C;
  const long nv = (r1 - r0) * cv;
  const int mych = (threadIdx.x % cv) * W;
  const int g = mych / Cg;
  const float cnt = (float)((r1 - r0) * Cg);
  float acc = 0.f;
  for (long v = threadIdx.x; v < nv; v += blockDim.x) {
    float f[W];
    vload<T>(f, xs + v * W);
#pragma unroll
    for (int j = 0; j < W; ++j) acc += f[j];
  }
  vec_group_reduce(acc, cv, W, Cg, G, part, red);
  const float m = cnt > 0.f ? red[g] / cnt : 0.f;
  acc = 0.f;
  for (long v = threadIdx.x; v < nv; v += blockDim.x) {
    float f[W];
    vload<T>(f, xs + v * W);
#pragma unroll
    for (int j = 0; j < W; ++j) { const float d = f[j] - m; acc += d * d; }
  }
  const float mkeep = m;
  vec_group_reduce(acc, cv, W, Cg, G, part, red);
  if ((int)threadIdx.x < cv && (mych % Cg) == 0) {
    float* o = ws + (((long)n * parts + pi) * G + g) * 2;
    o[0] = mkeep; o[1] = red[g];
  }
}
__global__ void groupnorm_merge_kernel(float* mean, float* rstd, const float* ws, long S, int C, int G, int parts, float eps, long NG) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // (sample, group)
  if (i >= NG) return;
  const long n = i / G; const int g = (int)(i - n * G);
  const int Cg = C / G;
  const long rpp = (S + parts - 1) / parts;
  float cnt = 0.f, m = 0.f, M2 = 0.f;
  for (int pi = 0; pi < parts; ++pi) {
    const long r0 = min((long)pi * rpp, S), r1 = min(r0 + rpp, S);
    const float c2 = (float)((r1 - r0) * Cg);
    if (c2 <= 0.f) continue;
    const float* o = ws + ((n * parts + pi) * G + g) * 2;
    const float d = o[0] - m, tot = cnt + c2;
    m += d * (c2 / tot);
    M2 += o[1] + d * d * (cnt * c2 / tot);
    cnt = tot;
  }
  mean[i] = m;
  rstd[i] = rsqrtf(M2 / cnt + eps);
}

template <typename T>
__global__ void groupnorm_apply_vec_kernel(T* y, const T* x, const float* gamma, const float* beta, const float* mean,
                                           const float* rstd, long S, int C, int G, int act, long nvec) {
  constexpr int W = VT<T>::W;
  const int Cg = C / G, cv = C / W;
  GRID_STRIDE(v, nvec) {
    const long row = v / cv; const int c0 = (int)(v - row * cv) * W;
    const long sg = (row / S) * G + c0 / Cg;
    const float m = mean[sg], rs = rstd[sg];
    float f[W];
    vload<T>(f, x + v * W);
#pragma unroll
    for (int j = 0; j < W; ++j) f[j] = act_f((f[j] - m) * rs * gamma[c0 + j] + beta[c0 + j], act);
    vstore<T>(y + v * W, f);
  }
}
// W-element vector access for a tensor whose element type differs from the one that sets the vector width (router-trunk backward in bf16
// mode: fp32 activations, 4 per thread, next to bf16 gradients, 4 per thread = 8 bytes)
template <typename TD, int W> struct VecW;
template <> struct VecW<float, 4> {
  static DEVI void load(float* f, const float* p) { vload<float>(f, p); }
  static DEVI void store(float* p, const float* f) { vstore<float>(p, f); }
};
template <> struct VecW<bf16, 8> {
  static DEVI void load(float* f, const bf16* p) { vload<bf16>(f, p); }
  static DEVI void store(bf16* p, const float* f) { vstore<bf16>(p, f); }
};
template <> struct VecW<bf16, 4> {
  static DEVI void load(float* f, const bf16* p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = (float)v[j];
  }
  static DEVI void store(bf16* p, const float* f) {
    bf16x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (bf16)f[j];
    *reinterpret_cast<bf16x4*>(p) = v;
  }
};
template <> struct VecW<float, 8> {                          // two 16-byte accesses
  static DEVI void load(float* f, const float* p) { vload<float>(f, p); vload<float>(f + 4, p + 4); }
  static DEVI void store(float* p, const float* f) { vstore<float>(p, f); vstore<float>(p + 4, f + 4); }
};
template <typename T, typename TD = T, int WV = VT<T>::W>
__global__ __launch_bounds__(512) void groupnorm_bwd_stats_vec_kernel(float* s1, float* s2, float* dgamma, float* dbeta, const TD* dy,
                                                                     const T* x, const float* gamma, const float* beta, const float* mean,
                                                                     const float* rstd, long S, int C, int G, int act, int parts,
                                                                     const float* dyb, float dybs) {
  // dyb (or null): the incoming gradient is dyb[n][c] * dybs at every position (the backward of a mean over the positions) -- read from
  // the (N, C) tensor instead of a materialised broadcast
  constexpr int W = WV;
  __shared__ float part[512], part2[512];
  const int n = blockIdx.x, Cg = C / G, cv = C / W;
  // parts > 1 (small batches): blockIdx.y owns a row range and adds its sums into the pre-zeroed s1 / s2
  const long rpp = (S + parts - 1) / parts;
  const long r0 = min((long)blockIdx.y * rpp, S), r1 = min(r0 + rpp, S);
  const long base = ((long)n * S + r0) * C;
  const long nv = (r1 - r0) * cv;
  const int mych = (threadIdx.x % cv) * W;
  const int g = mych / Cg;
  const float m = mean[(long)n * G + g], rs = rstd[(long)n * G + g];
  float gm[W], bt[W], dg[W], db[W];
#pragma unroll
  for (int j = 0; j < W; ++j) { gm[j] = gamma[mych + j]; bt[j] = beta[mych + j]; dg[j] = 0.f; db[j] = 0.f; }
  float a1 = 0.f, a2 = 0.f;
  float gb[W];
#pragma unroll
  for (int j = 0; j < W; ++j) gb[j] = dyb ? dyb[(long)n * C + mych + j] * dybs : 0.f;
  for (long v = threadIdx.x; v < nv; v += blockDim.x) {
    float f[W], d[W];
    VecW<T, W>::load(f, x + base + v * W);
    if (dyb) {
#pragma unroll
      for (int j = 0; j < W; ++j) d[j] = gb[j];
    } else {
      VecW<TD, W>::load(d, dy + base + v * W);
    }
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const float xh = (f[j] - m) * rs;
      const float dz = d[j] * act_grad_f(xh * gm[j] + bt[j], act);
      a1 += dz * gm[j]; a2 += dz * gm[j] * xh;
      dg[j] += dz * xh; db[j] += dz;
    }
  }
  __shared__ float red[64];
  vec_group_reduce(a1, cv, W, Cg, G, part, red);
  if ((int)threadIdx.x < G) { if (parts > 1) atomicAdd(&s1[(long)n * G + threadIdx.x], red[threadIdx.x]); else s1[(long)n * G + threadIdx.x] = red[threadIdx.x]; }
  vec_group_reduce(a2, cv, W, Cg, G, part, red);
  if ((int)threadIdx.x < G) { if (parts > 1) atomicAdd(&s2[(long)n * G + threadIdx.x], red[threadIdx.x]); else s2[(long)n * G + threadIdx.x] = red[threadIdx.x]; }
  // per-channel partials: strided tree over the threads that share a channel chunk, then one global atomic per channel
#pragma unroll
  for (int j = 0; j < W; ++j) {
    __syncthreads();
    part[threadIdx.x] = dg[j]; part2[threadIdx.x] = db[j];
    __syncthreads();
    for (int stride = (int)blockDim.x >> 1; stride >= cv; stride >>= 1) {
      if ((int)threadIdx.x < stride) { part[threadIdx.x] += part[threadIdx.x + stride]; part2[threadIdx.x] += part2[threadIdx.x + stride]; }
      __syncthreads();
    }
    if ((int)threadIdx.x < cv) { atomicAdd(&dgamma[threadIdx.x * W + j], part[threadIdx.x]); atomicAdd(&dbeta[threadIdx.x * W + j], part2[threadIdx.x]); }
  }
}
template <typename T, typename TD = T, int WV = VT<T>::W>
__global__ void groupnorm_bwd_apply_vec_kernel(TD* dx, const TD* dy, const T* x, const float* gamma, const float* beta, const float* mean,
                                               const float* rstd, const float* s1, const float* s2, long S, int C, int G, int act, long nvec,
                                               const float* dyb, float dybs) {
  constexpr int W = WV;
  const int Cg = C / G, cv = C / W;
  const float invm = 1.f / (float)(S * Cg);
  GRID_STRIDE(v, nvec) {
    const long row = v / cv; const int c0 = (int)(v - row * cv) * W;
    const long sg = (row / S) * G + c0 / Cg;
    const float m = mean[sg], rs = rstd[sg], u = s1[sg], w = s2[sg];
    float f[W], d[W];
    VecW<T, W>::load(f, x + v * W);
    if (dyb) {
#pragma unroll
      for (int j = 0; j < W; ++j) d[j] = dyb[(row / S) * C + c0 + j] * dybs;
    } else {
      VecW<TD, W>::load(d, dy + v * W);
    }
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const float xh = (f[j] - m) * rs;
      const float dz = d[j] * act_grad_f(xh * gamma[c0 + j] + beta[c0 + j], act);
      d[j] = rs * (dz * gamma[c0 + j] - invm * (u + xh * w));
    }
    VecW<TD, W>::store(dx + v * W, d);
  }
}

// a = relu(y * scale[n][c] + shift[n][c]) as bf16 (scale == null: a plain fp32 -> bf16 conversion): the input of a router-trunk conv as the
// bf16 weight-gradient program reads it (the forward never materialises it: conv6s applies the same transform while it stages its tiles)
__global__ void gn1t_act_kernel(bf16* out, const float* y, const float* scale, const float* shift, long S, int C, long nvec) {
  const int cv = C / 8;
  GRID_STRIDE(v, nvec) {
    const long row = v / cv; const int c0 = (int)(v - row * cv) * 8;
    float f[8];
    vload<float>(f, y + v * 8); vload<float>(f + 4, y + v * 8 + 4);
    if (scale) {
      const long sc = (row / S) * C + c0;
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] * scale[sc + j] + shift[sc + j], 0.f);
    }
    vstore<bf16>(out + v * 8, f);
  }
}

template <typename T> static inline int lpr_for(int C) {
  const int nv = C / VT<T>::W;
  int l = 1;
  while (l < nv) l <<= 1;
  return l;
}
#define LPR_SWITCH(lpr, CALL)                         \
  switch (lpr) {                                      \
    case 1: { constexpr int L = 1; CALL; } break;     \
    case 2: { constexpr int L = 2; CALL; } break;     \
    case 4: { constexpr int L = 4; CALL; } break;     \
    case 8: { constexpr int L = 8; CALL; } break;     \
    case 16: { constexpr int L = 16; CALL; } break;   \
    case 32: { constexpr int L = 32; CALL; } break;   \
    default: { constexpr int L = 64; CALL; } break;   \
  }
template <typename T> static inline bool row_vec_ok(int C, const void* a, const void* b, const void* c, const void* d) {
  const int W = VT<T>::W;
  auto al = [](const void* p) { return p == nullptr || (uintptr_t)p % 16 == 0; };
  return C % W == 0 && C / W <= 64 && al(a) && al(b) && al(c) && al(d);
}
static inline unsigned rows_grid(long rows, int lpr) {
  long b = (rows * lpr + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}
template <typename T> static inline bool gn_vec_ok(int C, int G, const void* a, const void* b, const void* c) {
  const int W = VT<T>::W;
  auto al = [](const void* p) { return p == nullptr || (uintptr_t)p % 16 == 0; };
  if (C % W || (C / G) % W) return false;
  const int cv = C / W;
  return cv <= 512 && 512 % cv == 0 && al(a) && al(b) && al(c);
}

}  // namespace

#define DT_SWITCH(dtype, CALL)                    \
  if ((dtype) == HDMOE_F32) { using T = float; CALL; }   \
  else if ((dtype) == HDMOE_BF16) { using T = bf16; CALL; } \
  else return HDMOE_EDTYPE;

// ---------------------------------------------------------------- GroupNorm(1, C) fused into the neighbouring convs (router trunks)
// Router.hard_route is conv -> GroupNorm(1, C) -> ReLU three times, then AdaptiveAvgPool2d(1) (reference model_components.py:100-112).
// The split-bf16 conv kernel leaves per-sample partial (sum, sum of squares) slots of its OUTPUT (conv6s_body.h); this kernel turns
// them into mean / rstd and the per-(sample, channel) affine  scale = gamma * rstd, shift = beta - mean * rstd * gamma  that the NEXT
// conv (and its weight gradient) apply to the tensor while staging it: the normalised activation is never written.
namespace {
__global__ void gn1_finalize_kernel(float* scale, float* shift, float* mean, float* rstd, const float* ws, const float* gamma, const float* beta,
                                    int slots, int C, float inv_count, float eps) {
  __shared__ float sm[2];
  const int n = blockIdx.x;
  // wave 0: lane l takes slots l, l + 64, .. (independent loads), then a fixed-shape shuffle tree -- the order depends on `slots` only,
  // never on the batch; fp64 for the few dozen partials (the cancellation in E[y^2] - mean^2)
  double s1 = 0.0, s2 = 0.0;
  if (threadIdx.x < 64) {
    for (int k = threadIdx.x; k < slots; k += 64) {
      const float2 v = *reinterpret_cast<const float2*>(ws + ((long)n * slots + k) * 2);
      s1 += v.x; s2 += v.y;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { s1 += __shfl_down(s1, d, 64); s2 += __shfl_down(s2, d, 64); }
  }
  if (threadIdx.x == 0) {
    const double m = s1 * inv_count;
    double var = s2 * inv_count - m * m;
    if (var < 0.0) var = 0.0;
    sm[0] = (float)m; sm[1] = (float)(1.0 / sqrt(var + (double)eps));
    mean[n] = sm[0]; rstd[n] = sm[1];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float g = gamma[c] * sm[1];
    scale[(long)n * C + c] = g;
    shift[(long)n * C + c] = beta[c] - sm[0] * g;
  }
}
// out[n][c] = mean over the S positions of relu(y[n][s][c] * scale[n][c] + shift[n][c]): the last GroupNorm + ReLU + average pool of the
// trunk in one read of y (fixed summation order)
constexpr int GN1_POOL_THREADS = 1024;                          // one block per sample: 16 waves keep enough 16-byte loads in flight
// ws != null: the statistics of y come as the producing conv's partial slots (the gn1_finalize step folded in: a launch less in the
// router's serial chain -- where a tiny kernel queues behind the other branch's persistent conv); scale / shift / mean / rstd are
// then OUTPUTS (kept for the backward).
__global__ __launch_bounds__(GN1_POOL_THREADS) void gn1_relu_mean_kernel(float* out, const float* y, float* scale, float* shift, long S, int C,
                                                                       const float* ws, const float* gamma, const float* beta, float* mean,
                                                                       float* rstd, int slots, float inv_count, float eps) {
  extern __shared__ float part[];                              // [threads / C4][C]
  __shared__ float smr[2];
  const int n = blockIdx.x, C4 = C / 4, rows = GN1_POOL_THREADS / C4;
  const int cq = threadIdx.x % C4, rw = threadIdx.x / C4;
  typedef __attribute__((ext_vector_type(4))) float f4;
  if (ws) {                                                    // same arithmetic and order as gn1_finalize_kernel
    double s1 = 0.0, s2 = 0.0;
    if (threadIdx.x < 64) {
      for (int k = threadIdx.x; k < slots; k += 64) {
        const float2 v = *reinterpret_cast<const float2*>(ws + ((long)n * slots + k) * 2);
        s1 += v.x; s2 += v.y;
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) { s1 += __shfl_down(s1, d, 64); s2 += __shfl_down(s2, d, 64); }
    }
    if (threadIdx.x == 0) {
      const double m = s1 * inv_count;
      double var = s2 * inv_count - m * m;
      if (var < 0.0) var = 0.0;
      smr[0] = (float)m; smr[1] = (float)(1.0 / sqrt(var + (double)eps));
      mean[n] = smr[0]; rstd[n] = smr[1];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += GN1_POOL_THREADS) {
      const float g = gamma[c] * smr[1];
      scale[(long)n * C + c] = g;
      shift[(long)n * C + c] = beta[c] - smr[0] * g;
    }
  }
  f4 acc = (f4)(0.f);
  if (rw < rows) {
    f4 sc, sh;
    if (ws) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float g = gamma[4 * cq + e] * smr[1]; sc[e] = g; sh[e] = beta[4 * cq + e] - smr[0] * g; }
    } else {
      sc = *reinterpret_cast<const f4*>(scale + (long)n * C + 4 * cq); sh = *reinterpret_cast<const f4*>(shift + (long)n * C + 4 * cq);
    }
    const float* yp = y + (long)n * S * C + 4 * cq;
    long s2 = rw;
    for (; s2 + 3 * rows < S; s2 += 4 * rows) {                 // four independent 16-byte loads in flight per thread (one block per sample:
      f4 v[4];                                                  //  the kernel lives on memory-level parallelism, not on occupancy)
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f4*>(yp + (s2 + (long)u * rows) * C);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += fmaxf(v[u][e] * sc[e] + sh[e], 0.f);
    }
    for (; s2 < S; s2 += rows) {
      const f4 v = *reinterpret_cast<const f4*>(yp + s2 * C);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += fmaxf(v[e] * sc[e] + sh[e], 0.f);
    }
    *reinterpret_cast<f4*>(part + (long)rw * C + 4 * cq) = acc;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += GN1_POOL_THREADS) {
    float t = 0.f;
    for (int k = 0; k < rows; ++k) t += part[(long)k * C + c];
    out[(long)n * C + c] = t / (float)S;
  }
}
}  // namespace

extern "C" {

int hdmoe_pixelnorm_fwd(void* xn, void* h, const void* x, long rows, int C, int dtype, hipStream_t stream) {
  DT_SWITCH(dtype, if (row_vec_ok<T>(C, xn, h, x, nullptr)) {
    const int lpr = lpr_for<T>(C);
    LPR_SWITCH(lpr, hipLaunchKernelGGL((pixelnorm_fwd_vec_kernel<T, L>), dim3(rows_grid(rows, lpr)), dim3(256), 0, stream, (T*)xn, (T*)h, (const T*)x, C, rows))
    return hdmoe_launch_status();
  })
  DT_SWITCH(dtype, hipLaunchKernelGGL(pixelnorm_fwd_kernel<T>, dim3(grid_for(rows)), dim3(TPB), 0, stream, (T*)xn, (T*)h, (const T*)x, C, rows))
  return hdmoe_launch_status();
}
int hdmoe_pixelnorm_bwd(void* dx, const void* dxn, const void* dh, const void* x, long rows, int C, float sx, int dtype, hipStream_t stream) {
  if (!dxn && !dh) return HDMOE_EINVAL;
  DT_SWITCH(dtype, if (row_vec_ok<T>(C, dx, dxn, dh, x)) {
    const int lpr = lpr_for<T>(C);
    LPR_SWITCH(lpr, hipLaunchKernelGGL((pixelnorm_bwd_vec_kernel<T, L>), dim3(rows_grid(rows, lpr)), dim3(256), 0, stream, (T*)dx, (const T*)dxn, (const T*)dh, (const T*)x, C, rows, sx))
    return hdmoe_launch_status();
  })
  DT_SWITCH(dtype, hipLaunchKernelGGL(pixelnorm_bwd_kernel<T>, dim3(grid_for(rows)), dim3(TPB), 0, stream, (T*)dx, (const T*)dxn, (const T*)dh, (const T*)x, C, rows, sx))
  return hdmoe_launch_status();
}
static int gn_check(int N, int C, int G) {
  if (N < 1 || G < 1 || G > 64 || (G & (G - 1)) || C % G || C > 4096) return HDMOE_EINVAL;   // G must divide 512 (thread<->group map)
  return HDMOE_OK;
}
static int groupnorm_fwd_impl(void* y, float* mean, float* rstd, float* ws, int parts, const void* x, const float* gamma, const float* beta,
                              int N, long S, int C, int G, int act, float eps, int dtype, hipStream_t stream) {
  if (gn_check(N, C, G) || parts < 1 || parts > 64) return HDMOE_EINVAL;
  const long n = (long)N * S * C;
  DT_SWITCH(dtype, if (gn_vec_ok<T>(C, G, y, x, nullptr)) {
    const long nvec = n / VT<T>::W;
    if (ws && parts > 1) {
      hipLaunchKernelGGL(groupnorm_part_stats_vec_kernel<T>, dim3(N, parts), dim3(512), 0, stream, ws, (const T*)x, S, C, G, parts);
      hipLaunchKernelGGL(groupnorm_merge_kernel, dim3(cdiv((long)N * G, 64)), dim3(64), 0, stream, mean, rstd, ws, S, C, G, parts, eps, (long)N * G);
    } else {
      hipLaunchKernelGGL(groupnorm_stats_vec_kernel<T>, dim3(N), dim3(512), 0, stream, mean, rstd, (const T*)x, S, C, G, eps);
    }
    hipLaunchKernelGGL(groupnorm_apply_vec_kernel<T>, dim3(grid_for(nvec)), dim3(TPB), 0, stream, (T*)y, (const T*)x, gamma, beta, mean, rstd, S, C, G, act, nvec);
    return hdmoe_launch_status();
  })
  DT_SWITCH(dtype, {
    hipLaunchKernelGGL(groupnorm_stats_kernel<T>, dim3(N), dim3(512), 0, stream, mean, rstd, (const T*)x, S, C, G, eps);
    hipLaunchKernelGGL(groupnorm_apply_kernel<T>, dim3(grid_for(n)), dim3(TPB), 0, stream, (T*)y, (const T*)x, gamma, beta, mean, rstd, S, C, G, act, n);
  })
  return hdmoe_launch_status();
}
int hdmoe_groupnorm_fwd(void* y, float* mean, float* rstd, const void* x, const float* gamma, const float* beta, int N,
                        long S, int C, int G, int act, float eps, int dtype, hipStream_t stream) {
  return groupnorm_fwd_impl(y, mean, rstd, nullptr, 1, x, gamma, beta, N, S, C, G, act, eps, dtype, stream);
}
int hdmoe_groupnorm_fwd_split(void* y, float* mean, float* rstd, float* ws, int parts, const void* x, const float* gamma,
                              const float* beta, int N, long S, int C, int G, int act, float eps, int dtype, hipStream_t stream) {
  return groupnorm_fwd_impl(y, mean, rstd, ws, parts, x, gamma, beta, N, S, C, G, act, eps, dtype, stream);
}
// ws: 2*N*G floats of scratch; dgamma/dbeta accumulate (caller zeroes)
static int groupnorm_bwd_impl(void* dx, float* dgamma, float* dbeta, float* ws, int parts, const void* dy, const void* x, const float* gamma,
                              const float* beta, const float* mean, const float* rstd, int N, long S, int C, int G, int act,
                              int dtype, hipStream_t stream, const float* dyb = nullptr, float dybs = 1.f) {
  if (gn_check(N, C, G) || parts < 1 || parts > 64) return HDMOE_EINVAL;
  if (dyb && dy == nullptr) dy = x;                           // (only for the vector-path alignment check; never read)
  const long n = (long)N * S * C;
  float* s1 = ws; float* s2 = ws + (long)N * G;
  DT_SWITCH(dtype, if (gn_vec_ok<T>(C, G, dx, dy, x)) {
    const long nvec = n / VT<T>::W;
    hipLaunchKernelGGL(groupnorm_bwd_stats_vec_kernel<T>, dim3(N, parts), dim3(512), 0, stream, s1, s2, dgamma, dbeta, (const T*)dy, (const T*)x,
                       gamma, beta, mean, rstd, S, C, G, act, parts, dyb, dybs);
    // 512, not 2048 blocks: this pass belongs to the router trunks' backward, which has slack, and a chip-filling grid of it holds up the
    // expert branch's kernels on the critical stream (same-box A/B 2048 -> 512: 15.52 -> 15.40 ms/step; the reverse experiment, four
    // row-range blocks per sample in the statistics pass, cost +0.4 ms)
    static const long gcap = getenv("HDMOE_GNB_GRID") ? atol(getenv("HDMOE_GNB_GRID")) : 512;
    unsigned gb = grid_for(nvec); if (gb > gcap) gb = (unsigned)gcap;
    hipLaunchKernelGGL(groupnorm_bwd_apply_vec_kernel<T>, dim3(gb), dim3(TPB), 0, stream, (T*)dx, (const T*)dy, (const T*)x,
                       gamma, beta, mean, rstd, s1, s2, S, C, G, act, nvec, dyb, dybs);
    return hdmoe_launch_status();
  })
  if (dyb) return HDMOE_EINVAL;                               // the broadcast form exists for the vector kernels only
  DT_SWITCH(dtype, {
    hipLaunchKernelGGL(groupnorm_bwd_stats_kernel<T>, dim3(N), dim3(512), 2 * C * sizeof(float), stream, s1, s2, dgamma, dbeta,
                       (const T*)dy, (const T*)x, gamma, beta, mean, rstd, S, C, G, act);
    hipLaunchKernelGGL(groupnorm_bwd_apply_kernel<T>, dim3(grid_for(n)), dim3(TPB), 0, stream, (T*)dx, (const T*)dy, (const T*)x,
                       gamma, beta, mean, rstd, s1, s2, S, C, G, act, n);
  })
  return hdmoe_launch_status();
}
int hdmoe_groupnorm_bwd(void* dx, float* dgamma, float* dbeta, float* ws, const void* dy, const void* x, const float* gamma,
                        const float* beta, const float* mean, const float* rstd, int N, long S, int C, int G, int act,
                        int dtype, hipStream_t stream) {
  return groupnorm_bwd_impl(dx, dgamma, dbeta, ws, 1, dy, x, gamma, beta, mean, rstd, N, S, C, G, act, dtype, stream);
}
/* The same with the incoming gradient given as g[n][c] * scale at every position (backward of a mean over the S positions that follows
 * the norm): nothing of size N*S*C is read but x.  fp32 / bf16 vector shapes only (C % vector width == 0). */
int hdmoe_groupnorm_bwd_bcast(void* dx, float* dgamma, float* dbeta, float* ws, const float* g, float scale, const void* x,
                              const float* gamma, const float* beta, const float* mean, const float* rstd, int N, long S, int C, int G,
                              int act, int dtype, hipStream_t stream) {
  if (!g) return HDMOE_EINVAL;
  return groupnorm_bwd_impl(dx, dgamma, dbeta, ws, 1, nullptr, x, gamma, beta, mean, rstd, N, S, C, G, act, dtype, stream, g, scale);
}
/* parts > 1: ws (2*N*G floats) must be zeroed by the caller; the row-range blocks add into it */
int hdmoe_groupnorm_bwd_split(void* dx, float* dgamma, float* dbeta, float* ws, int parts, const void* dy, const void* x,
                              const float* gamma, const float* beta, const float* mean, const float* rstd, int N, long S, int C,
                              int G, int act, int dtype, hipStream_t stream) {
  return groupnorm_bwd_impl(dx, dgamma, dbeta, ws, parts, dy, x, gamma, beta, mean, rstd, N, S, C, G, act, dtype, stream);
}
/* Router-trunk backward in bf16 mode (GroupNorm(1, C) + ReLU; x = the conv output y_l, fp32): the incoming gradient dz is bf16 [N][S][C] (or, dz ==
 * null, g[n][c] * gscale at every position), the result dx is written as bf16 -- the operands of the bf16 dgrad / weight-gradient programs.
 * ws: 2 * N floats; dgamma / dbeta accumulate.  C % 4 == 0 and 512 % (C / 4) == 0. */
int hdmoe_gn1t_bwd(void* dx, float* dgamma, float* dbeta, float* ws, const void* dz, const float* g, float gscale, const float* x, const float* gamma,
                   const float* beta, const float* mean, const float* rstd, int N, long S, int C, hipStream_t stream) {
  if (!dx || !dgamma || !dbeta || !ws || !x || (!dz && !g) || gn_check(N, C, 1)) return HDMOE_EINVAL;
  // 8 channels per thread (16-byte accesses to the bf16 tensors, two per fp32 vector) or 4 (HDMOE_GN1T_W=4: 8-byte bf16 accesses)
  static const int wsel = getenv("HDMOE_GN1T_W") ? atoi(getenv("HDMOE_GN1T_W")) : 4;   // (measured: 171 vs 214 us for stats + apply at C = 128, 13.58 vs 13.93 ms/step)
  const int Wv = (wsel == 4 || C % 8 || 512 % (C / 8)) ? 4 : 8;
  if (C % Wv || C / Wv > 512 || 512 % (C / Wv) || ((uintptr_t)x & 15) || ((uintptr_t)dx & 15) || ((uintptr_t)dz & 15)) return HDMOE_EINVAL;
  const long nvec = (long)N * S * C / Wv;
  float* s1 = ws; float* s2 = ws + N;
  static const long gcap = getenv("HDMOE_GNB_GRID") ? atol(getenv("HDMOE_GNB_GRID")) : 512;
  unsigned gb = grid_for(nvec); if (gb > gcap) gb = (unsigned)gcap;
  if (Wv == 8) {
    hipLaunchKernelGGL((groupnorm_bwd_stats_vec_kernel<float, bf16, 8>), dim3(N, 1), dim3(512), 0, stream, s1, s2, dgamma, dbeta, (const bf16*)dz, x, gamma, beta,
                       mean, rstd, S, C, 1, 1, 1, dz ? nullptr : g, gscale);
    hipLaunchKernelGGL((groupnorm_bwd_apply_vec_kernel<float, bf16, 8>), dim3(gb), dim3(TPB), 0, stream, (bf16*)dx, (const bf16*)dz, x, gamma, beta, mean, rstd,
                       s1, s2, S, C, 1, 1, nvec, dz ? nullptr : g, gscale);
  } else {
    hipLaunchKernelGGL((groupnorm_bwd_stats_vec_kernel<float, bf16, 4>), dim3(N, 1), dim3(512), 0, stream, s1, s2, dgamma, dbeta, (const bf16*)dz, x, gamma, beta,
                       mean, rstd, S, C, 1, 1, 1, dz ? nullptr : g, gscale);
    hipLaunchKernelGGL((groupnorm_bwd_apply_vec_kernel<float, bf16, 4>), dim3(gb), dim3(TPB), 0, stream, (bf16*)dx, (const bf16*)dz, x, gamma, beta, mean, rstd,
                       s1, s2, S, C, 1, 1, nvec, dz ? nullptr : g, gscale);
  }
  return hdmoe_launch_status();
}
/* out (bf16 [N][S][C]) = relu(y * scale[n][c] + shift[n][c])  (scale == null: out = bf16(y)); C % 8 == 0 */
int hdmoe_gn1t_act(void* out, const float* y, const float* scale, const float* shift, int N, long S, int C, hipStream_t stream) {
  if (!out || !y || (scale == nullptr) != (shift == nullptr) || N < 1 || S < 1 || C % 8 || !al16(out) || !al16(y)) return HDMOE_EINVAL;
  const long nvec = (long)N * S * C / 8;
  unsigned gb = grid_for(nvec); if (gb > 1024) gb = 1024;
  hipLaunchKernelGGL(gn1t_act_kernel, dim3(gb), dim3(TPB), 0, stream, (bf16*)out, y, scale, shift, S, C, nvec);
  return hdmoe_launch_status();
}
int hdmoe_layernorm_fwd(void* y, float* mean, float* rstd, const void* x, const float* gamma, const float* beta, long rows,
                        int C, float eps, int dtype, hipStream_t stream) {
  DT_SWITCH(dtype, if (row_vec_ok<T>(C, y, x, nullptr, nullptr)) {
    const int lpr = lpr_for<T>(C);
    LPR_SWITCH(lpr, hipLaunchKernelGGL((layernorm_fwd_vec_kernel<T, L>), dim3(rows_grid(rows, lpr)), dim3(256), 0, stream, (T*)y, mean, rstd, (const T*)x, gamma, beta, C, eps, rows))
    return hdmoe_launch_status();
  })
  DT_SWITCH(dtype, hipLaunchKernelGGL(layernorm_fwd_kernel<T>, dim3(grid_for(rows)), dim3(TPB), 0, stream, (T*)y, mean, rstd, (const T*)x, gamma, beta, C, eps, rows))
  return hdmoe_launch_status();
}
int hdmoe_layernorm_bwd(void* dx, float* dgamma, float* dbeta, const void* dy, const void* x, const float* gamma,
                        const float* mean, const float* rstd, long rows, int C, int dtype, hipStream_t stream) {
  if (C > 4096) return HDMOE_EINVAL;
  DT_SWITCH(dtype, if (row_vec_ok<T>(C, dx, dy, x, nullptr)) {
    const int lpr = lpr_for<T>(C);
    unsigned gv = rows_grid(rows, lpr); if (gv > 512) gv = 512;
    LPR_SWITCH(lpr, hipLaunchKernelGGL((layernorm_bwd_vec_kernel<T, L>), dim3(gv), dim3(256), 2 * C * sizeof(float), stream, (T*)dx, dgamma, dbeta,
                                       (const T*)dy, (const T*)x, gamma, mean, rstd, C, rows))
    return hdmoe_launch_status();
  })
  unsigned g = grid_for(rows);
  if (g > 256) g = 256;
  DT_SWITCH(dtype, hipLaunchKernelGGL(layernorm_bwd_kernel<T>, dim3(g), dim3(TPB), 2 * C * sizeof(float), stream, (T*)dx, dgamma, dbeta,
                                      (const T*)dy, (const T*)x, gamma, mean, rstd, C, rows))
  return hdmoe_launch_status();
}


/* partial slots [N][slots][2] of a conv output (hdmoe_conv_fwd_split_gn) -> mean / rstd [N] and scale / shift [N][C] of GroupNorm(1, C) */
int hdmoe_gn1_finalize(float* scale, float* shift, float* mean, float* rstd, const float* ws, const float* gamma, const float* beta, int N,
                       int slots, int C, long count, float eps, hipStream_t stream) {
  if (!scale || !shift || !mean || !rstd || !ws || !gamma || !beta || N < 1 || slots < 1 || C < 1 || count < 1) return HDMOE_EINVAL;
  hipLaunchKernelGGL(gn1_finalize_kernel, dim3(N), dim3(128), 0, stream, scale, shift, mean, rstd, ws, gamma, beta, slots, C, 1.f / (float)count, eps);
  return hdmoe_launch_status();
}
/* out [N][C] = mean_s relu(y [N][S][C] * scale [N][C] + shift [N][C])   (fp32; C % 4 == 0, C <= 1024) */
int hdmoe_gn1_relu_mean(float* out, const float* y, const float* scale, const float* shift, int N, long S, int C, hipStream_t stream) {
  if (!out || !y || !scale || !shift || N < 1 || S < 1 || C < 4 || C % 4 || C > 1024) return HDMOE_EINVAL;
  const int rows = GN1_POOL_THREADS / (C / 4);
  hipLaunchKernelGGL(gn1_relu_mean_kernel, dim3(N), dim3(GN1_POOL_THREADS), (size_t)rows * C * sizeof(float), stream, out, y, (float*)scale, (float*)shift,
                     S, C, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr, 0, 0.f, 0.f);
  return hdmoe_launch_status();
}
/* hdmoe_gn1_finalize + hdmoe_gn1_relu_mean in one launch: statistics from the conv's partial slots ws [N][slots][2]; scale / shift [N][C] and
 * mean / rstd [N] are written (for the backward), out [N][C] = mean_s relu(GroupNorm(1, C)(y)). */
int hdmoe_gn1_finalize_relu_mean(float* out, float* scale, float* shift, float* mean, float* rstd, const float* y, const float* ws,
                                 const float* gamma, const float* beta, int N, int slots, long S, int C, float eps, hipStream_t stream) {
  if (!out || !y || !scale || !shift || !mean || !rstd || !ws || !gamma || !beta || N < 1 || slots < 1 || S < 1 || C < 4 || C % 4 || C > 1024)
    return HDMOE_EINVAL;
  const int rows = GN1_POOL_THREADS / (C / 4);
  hipLaunchKernelGGL(gn1_relu_mean_kernel, dim3(N), dim3(GN1_POOL_THREADS), (size_t)rows * C * sizeof(float), stream, out, y, scale, shift, S, C, ws,
                     gamma, beta, mean, rstd, slots, 1.f / ((float)S * (float)C), eps);
  return hdmoe_launch_status();
}

}  // extern "C"
