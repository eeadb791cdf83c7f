// K5: multi-head attention core of MP_Attention (reference models/model_internals.py:374-404), flash
// style: the S_q x S_kv score map is never materialised (the reference allocates B x H x S x S).
//   q [B][Sq][E], k/v [B][Skv][E], head h = channels [h*D, (h+1)*D), scores/sqrt(D) (+ rel_pos_bias[h][i][j]),
//   softmax over kv, out [B][Sq][E]; lse [B][H][Sq] is kept for the backward.
// Head dim is tiny here (D = E/H = 4): QK^T has K = 4, so this is exp/VALU work, not MFMA work.  One thread owns
// one query row (fwd, dq) or one key row (dk/dv); the other side streams through LDS in fp32 tiles and is
// read by every lane at the same address (LDS broadcast, conflict-free).
#include "common.h"
#include "hdmoe.h"

namespace {

constexpr int TQ = 256;     // rows owned by a block (one per thread)
constexpr int TK = 128;     // streamed rows per LDS tile
constexpr int CH = 16;      // online-softmax chunk

template <typename T, int D, bool BIAS>
__global__ __launch_bounds__(TQ) void attn_fwd_kernel(T* out, float* lse, const T* q, const T* k, const T* v, const float* bias,
                                                     int Sq, int Skv, int H, int Sb, float scale) {
  __shared__ float sk[TK * D], sv[TK * D];
  const int b = blockIdx.z, h = blockIdx.y, i = blockIdx.x * TQ + threadIdx.x;
  const int E = H * D;
  const bool act = i < Sq;
  float qv[D], o[D];
#pragma unroll
  for (int d = 0; d < D; ++d) { qv[d] = act ? to_f(q[((long)b * Sq + i) * E + h * D + d]) * scale : 0.f; o[d] = 0.f; }
  float m = -INFINITY, l = 0.f;
  const float* brow = (BIAS && act) ? bias + ((long)h * Sb + i) * Sb : nullptr;
  for (int j0 = 0; j0 < Skv; j0 += TK) {
    const int nj = min(TK, Skv - j0);
    __syncthreads();
    for (int e = threadIdx.x; e < nj * D; e += TQ) {
      const long src = ((long)b * Skv + j0 + e / D) * E + h * D + e % D;
      sk[e] = to_f(k[src]); sv[e] = to_f(v[src]);
    }
    __syncthreads();
    if (!act) continue;
    for (int c0 = 0; c0 < nj; c0 += CH) {
      const int nc = min(CH, nj - c0);
      float s[CH];
      float cm = -INFINITY;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        float a = -INFINITY;
        if (c < nc) {
          a = 0.f;
#pragma unroll
          for (int d = 0; d < D; ++d) a += qv[d] * sk[(c0 + c) * D + d];
          if (BIAS) a += brow[j0 + c0 + c];
        }
        s[c] = a; cm = fmaxf(cm, a);
      }
      const float mn = fmaxf(m, cm);
      const float corr = __expf(m - mn);        // m = -inf on the first chunk -> 0
      l *= corr;
#pragma unroll
      for (int d = 0; d < D; ++d) o[d] *= corr;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (c < nc) {
          const float p = __expf(s[c] - mn);
          l += p;
#pragma unroll
          for (int d = 0; d < D; ++d) o[d] += p * sv[(c0 + c) * D + d];
        }
      }
      m = mn;
    }
  }
  if (act) {
    const float il = 1.f / l;
#pragma unroll
    for (int d = 0; d < D; ++d) out[((long)b * Sq + i) * E + h * D + d] = from_f<T>(o[d] * il);
    lse[((long)b * H + h) * Sq + i] = m + __logf(l);
  }
}

// dq (+ delta, + dbias): thread = query row
template <typename T, int D, bool BIAS>
__global__ __launch_bounds__(TQ) void attn_bwd_dq_kernel(T* dq, float* delta, float* dbias, const T* dout, const T* out, const T* q,
                                                        const T* k, const T* v, const float* lse, const float* bias, int Sq, int Skv,
                                                        int H, int Sb, float scale) {
  __shared__ float sk[TK * D], sv[TK * D];
  const int b = blockIdx.z, h = blockIdx.y, i = blockIdx.x * TQ + threadIdx.x;
  const int E = H * D;
  const bool act = i < Sq;
  float qv[D], dov[D], acc[D];
  float dl = 0.f, ls = 0.f;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const long idx = ((long)b * Sq + i) * E + h * D + d;
    qv[d] = act ? to_f(q[idx]) * scale : 0.f;
    dov[d] = act ? to_f(dout[idx]) : 0.f;
    if (act) dl += dov[d] * to_f(out[idx]);
    acc[d] = 0.f;
  }
  if (act) { ls = lse[((long)b * H + h) * Sq + i]; delta[((long)b * H + h) * Sq + i] = dl; }
  const long boff = ((long)h * Sb + i) * Sb;
  for (int j0 = 0; j0 < Skv; j0 += TK) {
    const int nj = min(TK, Skv - j0);
    __syncthreads();
    for (int e = threadIdx.x; e < nj * D; e += TQ) {
      const long src = ((long)b * Skv + j0 + e / D) * E + h * D + e % D;
      sk[e] = to_f(k[src]); sv[e] = to_f(v[src]);
    }
    __syncthreads();
    if (!act) continue;
#pragma unroll 8
    for (int j = 0; j < nj; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) { s += qv[d] * sk[j * D + d]; dp += dov[d] * sv[j * D + d]; }
      if (BIAS) s += bias[boff + j0 + j];
      const float ds = __expf(s - ls) * (dp - dl);
#pragma unroll
      for (int d = 0; d < D; ++d) acc[d] += ds * sk[j * D + d];
    }
  }
  if (act) {
#pragma unroll
    for (int d = 0; d < D; ++d) dq[((long)b * Sq + i) * E + h * D + d] = from_f<T>(acc[d] * scale);
  }
}

// dk, dv: thread = key row; queries stream through LDS
template <typename T, int D, bool BIAS>
__global__ __launch_bounds__(TQ) void attn_bwd_dkv_kernel(T* dk, T* dv, const T* dout, const T* q, const T* k, const T* v,
                                                         const float* lse, const float* delta, const float* bias, int Sq, int Skv,
                                                         int H, int Sb, float scale) {
  __shared__ float sq[TK * D], sdo[TK * D], sl[TK], sd[TK];
  const int b = blockIdx.z, h = blockIdx.y, j = blockIdx.x * TQ + threadIdx.x;
  const int E = H * D;
  const bool act = j < Skv;
  float kv[D], vv[D], ak[D], av[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const long idx = ((long)b * Skv + j) * E + h * D + d;
    kv[d] = act ? to_f(k[idx]) : 0.f;
    vv[d] = act ? to_f(v[idx]) : 0.f;
    ak[d] = 0.f; av[d] = 0.f;
  }
  for (int i0 = 0; i0 < Sq; i0 += TK) {
    const int ni = min(TK, Sq - i0);
    __syncthreads();
    for (int e = threadIdx.x; e < ni * D; e += TQ) {
      const long src = ((long)b * Sq + i0 + e / D) * E + h * D + e % D;
      sq[e] = to_f(q[src]) * scale; sdo[e] = to_f(dout[src]);
    }
    for (int e = threadIdx.x; e < ni; e += TQ) {
      sl[e] = lse[((long)b * H + h) * Sq + i0 + e];
      sd[e] = delta[((long)b * H + h) * Sq + i0 + e];
    }
    __syncthreads();
    if (!act) continue;
#pragma unroll 8
    for (int i = 0; i < ni; ++i) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) { s += sq[i * D + d] * kv[d]; dp += sdo[i * D + d] * vv[d]; }
      if (BIAS) s += bias[((long)h * Sb + i0 + i) * Sb + j];
      const float p = __expf(s - sl[i]);
      const float ds = p * (dp - sd[i]);
#pragma unroll
      for (int d = 0; d < D; ++d) { av[d] += p * sdo[i * D + d]; ak[d] += ds * sq[i * D + d]; }
    }
  }
  if (act) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const long idx = ((long)b * Skv + j) * E + h * D + d;
      dk[idx] = from_f<T>(ak[d]);           // sq already carries `scale`
      dv[idx] = from_f<T>(av[d]);
    }
  }
}

// d(rel_pos_bias)[h][i][j] = sum_b ds[b][h][i][j].  One thread per (h, i, j) walks the batch (no atomics: with B samples
// adding into the same H*S*S words the atomic version was 256-way contended and cost ~100 us per call on S = 64).
template <typename T, int D>
__global__ void attn_dbias_kernel(float* dbias, const T* dout, const T* q, const T* k, const T* v, const float* lse,
                                  const float* delta, const float* bias, int B, int Sq, int Skv, int H, int Sb, float scale) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)H * Sq * Skv) return;
  const int j = (int)(idx % Skv);
  const int i = (int)((idx / Skv) % Sq);
  const int h = (int)(idx / ((long)Skv * Sq));
  const int E = H * D;
  const float bv = bias[((long)h * Sb + i) * Sb + j];
  float acc = 0.f;
  for (int b = 0; b < B; ++b) {
    const T* qp = q + ((long)b * Sq + i) * E + h * D;
    const T* dop = dout + ((long)b * Sq + i) * E + h * D;
    const T* kp = k + ((long)b * Skv + j) * E + h * D;
    const T* vp = v + ((long)b * Skv + j) * E + h * D;
    float s = 0.f, dp = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) { s += to_f(qp[d]) * to_f(kp[d]); dp += to_f(dop[d]) * to_f(vp[d]); }
    const long li = ((long)b * H + h) * Sq + i;
    acc += __expf(s * scale + bv - lse[li]) * (dp - delta[li]);
  }
  dbias[((long)h * Sb + i) * Sb + j] = acc;
}

template <typename T, int D>
int attn_fwd_launch(void* out, float* lse, const void* q, const void* k, const void* v, const float* bias, int B, int Sq, int Skv,
                    int H, int Sb, hipStream_t st) {
  dim3 grid(cdiv(Sq, TQ), H, B);
  if (bias) hipLaunchKernelGGL((attn_fwd_kernel<T, D, true>), grid, dim3(TQ), 0, st, (T*)out, lse, (const T*)q, (const T*)k, (const T*)v, bias, Sq,
                     Skv, H, Sb, 1.f / sqrtf((float)D));
  else hipLaunchKernelGGL((attn_fwd_kernel<T, D, false>), grid, dim3(TQ), 0, st, (T*)out, lse, (const T*)q, (const T*)k, (const T*)v, bias, Sq,
                     Skv, H, Sb, 1.f / sqrtf((float)D));
  return hdmoe_launch_status();
}
template <typename T, int D>
int attn_bwd_launch(void* dq, void* dk, void* dv, float* dbias, float* delta, const void* dout, const void* out, const void* q,
                    const void* k, const void* v, const float* lse, const float* bias, int B, int Sq, int Skv, int H, int Sb,
                    hipStream_t st) {
  const float scale = 1.f / sqrtf((float)D);
#define ATTN_BWD(BB)                                                                                                                  \
  hipLaunchKernelGGL((attn_bwd_dq_kernel<T, D, BB>), dim3(cdiv(Sq, TQ), H, B), dim3(TQ), 0, st, (T*)dq, delta, dbias, (const T*)dout,  \
                     (const T*)out, (const T*)q, (const T*)k, (const T*)v, lse, bias, Sq, Skv, H, Sb, scale);                          \
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, D, BB>), dim3(cdiv(Skv, TQ), H, B), dim3(TQ), 0, st, (T*)dk, (T*)dv, (const T*)dout,      \
                     (const T*)q, (const T*)k, (const T*)v, lse, delta, bias, Sq, Skv, H, Sb, scale);
  if (bias) { ATTN_BWD(true) } else { ATTN_BWD(false) }
  if (bias && dbias) {
    const long nthreads = (long)H * Sq * Skv;
    hipLaunchKernelGGL((attn_dbias_kernel<T, D>), dim3(cdiv(nthreads, 256)), dim3(256), 0, st, dbias, (const T*)dout, (const T*)q, (const T*)k,
                       (const T*)v, lse, delta, bias, B, Sq, Skv, H, Sb, scale);
  }
  return hdmoe_launch_status();
}

}  // namespace

#define D_SWITCH(D, CALL)                       \
  switch (D) {                                  \
    case 1: { constexpr int DD = 1; CALL; }     \
    case 2: { constexpr int DD = 2; CALL; }     \
    case 4: { constexpr int DD = 4; CALL; }     \
    case 8: { constexpr int DD = 8; CALL; }     \
    case 16: { constexpr int DD = 16; CALL; }   \
    case 32: { constexpr int DD = 32; CALL; }   \
    default: return HDMOE_EINVAL;               \
  }

extern "C" {

int hdmoe_attn_fwd(void* out, float* lse, const void* q, const void* k, const void* v, const float* bias, int B, int Sq, int Skv,
                   int H, int D, int Sb, int dtype, hipStream_t stream) {
  if (B < 1 || B > 65535 || H < 1 || H > 65535 || Sq < 1 || Skv < 1 || (bias && (Sb < Sq || Sb < Skv))) return HDMOE_EINVAL;
  if (dtype == HDMOE_F32) { D_SWITCH(D, return (attn_fwd_launch<float, DD>(out, lse, q, k, v, bias, B, Sq, Skv, H, Sb, stream))) }
  if (dtype == HDMOE_BF16) { D_SWITCH(D, return (attn_fwd_launch<bf16, DD>(out, lse, q, k, v, bias, B, Sq, Skv, H, Sb, stream))) }
  return HDMOE_EDTYPE;
}

// delta: [B][H][Sq] fp32 scratch; dbias: [H][Sb][Sb] fp32 accumulator (caller zeroes) or null
int hdmoe_attn_bwd(void* dq, void* dk, void* dv, float* dbias, float* delta, const void* dout, const void* out, const void* q,
                   const void* k, const void* v, const float* lse, const float* bias, int B, int Sq, int Skv, int H, int D, int Sb,
                   int dtype, hipStream_t stream) {
  if (B < 1 || B > 65535 || H < 1 || H > 65535 || Sq < 1 || Skv < 1 || (bias && (Sb < Sq || Sb < Skv)) || (dbias && !bias)) return HDMOE_EINVAL;
  if (dtype == HDMOE_F32) { D_SWITCH(D, return (attn_bwd_launch<float, DD>(dq, dk, dv, dbias, delta, dout, out, q, k, v, lse, bias, B, Sq, Skv, H, Sb, stream))) }
  if (dtype == HDMOE_BF16) { D_SWITCH(D, return (attn_bwd_launch<bf16, DD>(dq, dk, dv, dbias, delta, dout, out, q, k, v, lse, bias, B, Sq, Skv, H, Sb, stream))) }
  return HDMOE_EDTYPE;
}

}  // extern "C"
