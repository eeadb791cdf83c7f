// Ragged-token kernels of the ViT expert BANK (Vit_expert / Vit_block, reference models/model_components.py:435-706).
//
// The reference runs its ViT experts one after the other on the samples routed to each (models/model_config1.py:25-37).  The experts
// share every shape except the patch size, i.e. the number of tokens per sample (S_e = ceil(R/p_e)^2), so the bank keeps the routed
// rows in ONE padded tensor tok[R][Sp][C] (Sp = max S_e; row r belongs to expert g(r) with rows [seg[g], seg[g+1]) from the device-side
// dispatch plan, and holds len[g] real tokens followed by padding) and runs each layer of all experts as one launch:
//   * token-wise layers (linears through the grouped conv kernels, LayerNorm, activations) simply run over the padding too: padded
//     tokens never influence real ones there, and their gradients are exactly zero (nothing downstream reads them), so they add
//     nothing to any weight gradient;
//   * the two layers that mix tokens -- GroupNorm statistics and attention -- take the per-expert token count (this file: GroupNorm;
//     attention.hip: hdmoe_attn_rag_*), write zeros to the padding and send zero gradient into it;
//   * pack / unpack / select move between the padded tensor and per-expert compact tensors (patch embedding in, un-patching out).
// One launch per layer for ALL experts instead of one per expert: the ~1100 few-microsecond launches per step of the per-expert
// path were the step's host-enqueue floor (hipGraph replay costs ~4.8 us of host time per kernel node on this stack).
#include "common.h"
#include "hdmoe.h"

namespace {

constexpr int TPB = 256;
struct Rag {
  const int* seg; int ng;
  int len[HDMOE_MAX_GROUPS];
};
DEVI int rag_group(const Rag& r, int row) {
  int g = -1;
  for (int i = 0; i < r.ng; ++i)
    if (row >= r.seg[i] && row < r.seg[i + 1]) g = i;
  return g;
}
struct Ptrs { const void* p[HDMOE_MAX_GROUPS]; };
struct MPtrs { void* p[HDMOE_MAX_GROUPS]; };
struct FPtrs { const float* p[HDMOE_MAX_GROUPS]; };
struct MFPtrs { float* p[HDMOE_MAX_GROUPS]; };

// ---------------------------------------------------------------- pack: compact per-expert tokens (+ pos_emb) -> padded rows
template <typename T>
__global__ void rag_pack_kernel(T* dst, Ptrs srcs, FPtrs pos, Rag rg, int R, int Sp, int C) {
  const long n = (long)R * Sp * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); const long t = i / C;
    const int s = (int)(t % Sp), r = (int)(t / Sp);
    const int g = rag_group(rg, r);
    float v = 0.f;
    if (g >= 0 && s < rg.len[g]) {
      v = to_f(((const T*)srcs.p[g])[((long)r * rg.len[g] + s) * C + c]);
      if (pos.p[g]) v += pos.p[g][(long)s * C + c];
    }
    dst[i] = from_f<T>(v);
  }
}
// backward of pack: every expert's compact gradient (zero outside its rows), pos_emb gradients accumulated
template <typename T>
__global__ void rag_pack_bwd_kernel(MPtrs dsrcs, MFPtrs dpos, const T* ddst, Rag rg, int R, int Sp, int C) {
  const long n = (long)R * Sp * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); const long t = i / C;
    const int s = (int)(t % Sp), r = (int)(t / Sp);
    const int g = rag_group(rg, r);
    const T v = ddst[i];
    for (int e = 0; e < rg.ng; ++e)
      if (s < rg.len[e]) ((T*)dsrcs.p[e])[((long)r * rg.len[e] + s) * C + c] = e == g ? v : from_f<T>(0.f);
    if (g >= 0 && s < rg.len[g] && dpos.p[g]) atomicAdd(&dpos.p[g][(long)s * C + c], to_f(v));
  }
}
// unpack: padded rows -> every expert's compact tensor (all rows; rows of other experts carry other tokens -- finite, unused)
template <typename T>
__global__ void rag_unpack_kernel(MPtrs dsts, const T* src, Rag rg, int R, int Sp, int C) {
  const long n = (long)R * Sp * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); const long t = i / C;
    const int s = (int)(t % Sp), r = (int)(t / Sp);
    const T v = src[i];
    for (int e = 0; e < rg.ng; ++e)
      if (s < rg.len[e]) ((T*)dsts.p[e])[((long)r * rg.len[e] + s) * C + c] = v;
  }
}
template <typename T>
__global__ void rag_unpack_bwd_kernel(T* dsrc, Ptrs ddsts, Rag rg, int R, int Sp, int C) {
  const long n = (long)R * Sp * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); const long t = i / C;
    const int s = (int)(t % Sp), r = (int)(t / Sp);
    float v = 0.f;
    for (int e = 0; e < rg.ng; ++e)
      if (s < rg.len[e]) v += to_f(((const T*)ddsts.p[e])[((long)r * rg.len[e] + s) * C + c]);
    dsrc[i] = from_f<T>(v);
  }
}
// select: y[r][:] = outs[g(r)][r][:]   (16-byte vectors; L16 per row);  backward: douts[e][r] = e == g(r) ? dy[r] : 0
__global__ void rag_select_kernel(uint4* y, Ptrs outs, Rag rg, int R, long L16) {
  const long n = (long)R * L16;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int g = rag_group(rg, (int)(i / L16));
    y[i] = g >= 0 ? ((const uint4*)outs.p[g])[i] : make_uint4(0, 0, 0, 0);
  }
}
__global__ void rag_select_bwd_kernel(MPtrs douts, const uint4* dy, Rag rg, int R, long L16) {
  const long n = (long)R * L16;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int g = rag_group(rg, (int)(i / L16));
    const uint4 v = dy[i];
    for (int e = 0; e < rg.ng; ++e) ((uint4*)douts.p[e])[i] = e == g ? v : make_uint4(0, 0, 0, 0);
  }
}

// ---------------------------------------------------------------- GroupNorm over a row's REAL tokens (+ activation), one block per row
// nn.GroupNorm on (B, C, S): statistics over (C / G channels) x S tokens per sample (reference Vit_block.GN, :529-531).
DEVI float act_f(float v, int act) { return act == 1 ? fmaxf(v, 0.f) : (act == 2 ? mp_silu_f(v) : v); }
DEVI float act_grad_f(float v, int act) { return act == 1 ? (v > 0.f ? 1.f : 0.f) : (act == 2 ? mp_silu_grad_f(v) : 1.f); }

// deterministic block sum (fixed shuffle tree, then the waves' partials in wave order): every thread gets the total
DEVI float block_sum(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();                                          // `red` may still be read from the previous call
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < TPB / 64; ++w) t += red[w];
  return t;
}

template <typename T>
__global__ __launch_bounds__(TPB) void gn_rag_fwd_kernel(T* y, float* mean, float* rstd, const T* x, FPtrs gamma, FPtrs beta, Rag rg,
                                                        int Sp, int C, int G, int act, float eps) {
  __shared__ float s1[32], s2[32];
  const int r = blockIdx.x, tid = threadIdx.x;
  const int g = rag_group(rg, r);
  const int len = g >= 0 ? rg.len[g] : 0;
  const int Cg = C / G, n = len * C;
  const T* xr = x + (long)r * Sp * C;
  T* yr = y + (long)r * Sp * C;
  __shared__ float red[TPB / 64];
  // statistics with a fixed summation order (a sample's output bits must not depend on its batch): two-pass mean / variance
  const float inv_m = len > 0 ? 1.f / (float)(len * Cg) : 0.f;
  for (int gr = 0; gr < G; ++gr) {
    float p = 0.f;
    for (int e = tid; e < len * Cg; e += TPB) p += to_f(xr[(e / Cg) * C + gr * Cg + e % Cg]);
    const float tot = block_sum(p, red);
    if (tid == 0) s1[gr] = tot;
  }
  __syncthreads();
  for (int gr = 0; gr < G; ++gr) {
    const float mu = s1[gr] * inv_m;
    float p = 0.f;
    for (int e = tid; e < len * Cg; e += TPB) { const float d = to_f(xr[(e / Cg) * C + gr * Cg + e % Cg]) - mu; p += d * d; }
    const float tot = block_sum(p, red);
    if (tid == 0) s2[gr] = tot;
  }
  __syncthreads();
  if (tid < G) { mean[(long)r * G + tid] = s1[tid] * inv_m; rstd[(long)r * G + tid] = rsqrtf(s2[tid] * inv_m + eps); }
  for (int e = tid; e < Sp * C; e += TPB) {
    float o = 0.f;
    if (e < n) {
      const int c = e % C, gr = c / Cg;
      const float xh = (to_f(xr[e]) - s1[gr] * inv_m) * rsqrtf(s2[gr] * inv_m + eps);
      o = act_f(xh * gamma.p[g][c] + beta.p[g][c], act);
    }
    yr[e] = from_f<T>(o);
  }
}
template <typename T>
__global__ __launch_bounds__(TPB) void gn_rag_bwd_kernel(T* dx, MFPtrs dgamma, MFPtrs dbeta, const T* dy, const T* x, FPtrs gamma, FPtrs beta,
                                                        const float* mean, const float* rstd, Rag rg, int Sp, int C, int G, int act) {
  __shared__ float a1[32], a2[32];
  extern __shared__ float sm[];                              // [2 * C] per-channel sums of this row
  const int r = blockIdx.x, tid = threadIdx.x;
  const int g = rag_group(rg, r);
  const int len = g >= 0 ? rg.len[g] : 0;
  const int Cg = C / G, n = len * C;
  const T* xr = x + (long)r * Sp * C;
  const T* dyr = dy + (long)r * Sp * C;
  T* dxr = dx + (long)r * Sp * C;
  __shared__ float red[TPB / 64];
  for (int c = tid; c < 2 * C; c += TPB) sm[c] = 0.f;
  __syncthreads();
  // the two per-group sums that enter dx: fixed summation order (dx of a sample must not depend on its batch)
  for (int gr = 0; gr < G; ++gr) {
    const float mu = mean[(long)r * G + gr], rs = rstd[(long)r * G + gr];
    float p1 = 0.f, p2 = 0.f;
    for (int q = tid; q < len * Cg; q += TPB) {
      const int c = gr * Cg + q % Cg, e = (q / Cg) * C + c;
      const float xh = (to_f(xr[e]) - mu) * rs;
      const float dz = to_f(dyr[e]) * act_grad_f(xh * gamma.p[g][c] + beta.p[g][c], act);
      p1 += dz * gamma.p[g][c];
      p2 += dz * gamma.p[g][c] * xh;
      atomicAdd(&sm[c], dz * xh);                            // per-channel parameter sums (added to global memory with atomics anyway)
      atomicAdd(&sm[C + c], dz);
    }
    const float t1 = block_sum(p1, red), t2 = block_sum(p2, red);
    if (tid == 0) { a1[gr] = t1; a2[gr] = t2; }
  }
  __syncthreads();
  const float inv_m = len > 0 ? 1.f / (float)(len * Cg) : 0.f;
  for (int e = tid; e < Sp * C; e += TPB) {
    float o = 0.f;
    if (e < n) {
      const int c = e % C, gr = c / Cg;
      const float rs = rstd[(long)r * G + gr];
      const float xh = (to_f(xr[e]) - mean[(long)r * G + gr]) * rs;
      const float dz = to_f(dyr[e]) * act_grad_f(xh * gamma.p[g][c] + beta.p[g][c], act);
      o = rs * (dz * gamma.p[g][c] - inv_m * (a1[gr] + xh * a2[gr]));
    }
    dxr[e] = from_f<T>(o);                                  // zero gradient into the padding
  }
  if (g >= 0)
    for (int c = tid; c < C; c += TPB) { atomicAdd(&dgamma.p[g][c], sm[c]); atomicAdd(&dbeta.p[g][c], sm[C + c]); }
}

// ---------------------------------------------------------------- LayerNorm per token with per-expert affine
// One workgroup per row; 32 consecutive lanes own one token (lane l holds channels l, l + 32, ...: coalesced 64-byte rows at C = 32),
// the per-token sums are 5-step butterflies inside the 32-lane half-wave, the parameter gradients accumulate per lane over the row's
// tokens and leave the block as one atomic per channel.
template <int CPL> DEVI float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <typename T, int CPL>
__global__ __launch_bounds__(TPB) void ln_rag_fwd_kernel(T* y, float* mean, float* rstd, const T* x, FPtrs gamma, FPtrs beta, Rag rg,
                                                        int Sp, int C, float eps) {
  const int r = blockIdx.x, l = threadIdx.x & 31, slot = threadIdx.x >> 5;
  const int g = rag_group(rg, r);
  float ga[CPL], be[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int c = l + 32 * k;
    ga[k] = (g >= 0 && c < C) ? gamma.p[g][c] : 0.f;
    be[k] = (g >= 0 && c < C) ? beta.p[g][c] : 0.f;
  }
  const float ic = 1.f / (float)C;
  for (int s = slot; s < Sp; s += TPB / 32) {
    const long t = (long)r * Sp + s;
    float xv[CPL], m = 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k) { const int c = l + 32 * k; xv[k] = c < C ? to_f(x[t * C + c]) : 0.f; m += xv[k]; }
    m = half_sum<CPL>(m) * ic;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k) { const float d = (l + 32 * k < C) ? xv[k] - m : 0.f; v += d * d; }
    const float rs = rsqrtf(half_sum<CPL>(v) * ic + eps);
    if (l == 0) { mean[t] = m; rstd[t] = rs; }
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = l + 32 * k;
      if (c < C) y[t * C + c] = from_f<T>(g >= 0 ? (xv[k] - m) * rs * ga[k] + be[k] : 0.f);
    }
  }
}
template <typename T, int CPL>
__global__ __launch_bounds__(TPB) void ln_rag_bwd_kernel(T* dx, MFPtrs dgamma, MFPtrs dbeta, const T* dy, const T* x, FPtrs gamma,
                                                        const float* mean, const float* rstd, Rag rg, int Sp, int C) {
  __shared__ float red[2][TPB / 32][32 * CPL];
  const int r = blockIdx.x, l = threadIdx.x & 31, slot = threadIdx.x >> 5;
  const int g = rag_group(rg, r);
  float ga[CPL], dg[CPL], db[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int c = l + 32 * k;
    ga[k] = (g >= 0 && c < C) ? gamma.p[g][c] : 0.f;
    dg[k] = 0.f; db[k] = 0.f;
  }
  const float ic = 1.f / (float)C;
  for (int s = slot; s < Sp; s += TPB / 32) {
    const long t = (long)r * Sp + s;
    const float m = mean[t], rs = rstd[t];
    float gv[CPL], xh[CPL], a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = l + 32 * k;
      gv[k] = (g >= 0 && c < C) ? to_f(dy[t * C + c]) : 0.f;
      xh[k] = c < C ? (to_f(x[t * C + c]) - m) * rs : 0.f;
      a1 += gv[k] * ga[k];
      a2 += gv[k] * ga[k] * xh[k];
      dg[k] += gv[k] * xh[k];
      db[k] += gv[k];
    }
    a1 = half_sum<CPL>(a1); a2 = half_sum<CPL>(a2);
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = l + 32 * k;
      if (c < C) dx[t * C + c] = from_f<T>(rs * (gv[k] * ga[k] - ic * (a1 + xh[k] * a2)));     // g < 0: gv = ga = 0 -> 0
    }
  }
#pragma unroll
  for (int k = 0; k < CPL; ++k) { red[0][slot][l + 32 * k] = dg[k]; red[1][slot][l + 32 * k] = db[k]; }
  __syncthreads();
  if (g >= 0)
    for (int e = threadIdx.x; e < 2 * C; e += TPB) {
      const int which = e >= C, c = e - which * C;
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < TPB / 32; ++w) tot += red[which][w][c];
      atomicAdd(which ? &dbeta.p[g][c] : &dgamma.p[g][c], tot);
    }
}

static inline unsigned grid_for(long n) { long b = (n + TPB - 1) / TPB; if (b > 4096) b = 4096; if (b < 1) b = 1; return (unsigned)b; }
static inline bool mk_rag(Rag& rg, const int* seg, const int* lens, int ng) {
  if (!seg || !lens || ng < 1 || ng > HDMOE_MAX_GROUPS) return false;
  rg.seg = seg; rg.ng = ng;
  for (int g = 0; g < HDMOE_MAX_GROUPS; ++g) rg.len[g] = g < ng ? lens[g] : 0;
  return true;
}
template <typename P, typename Q> static inline void fill(P& dst, Q* const* src, int ng) {
  for (int g = 0; g < HDMOE_MAX_GROUPS; ++g) dst.p[g] = (src && g < ng) ? src[g] : nullptr;
}

}  // namespace

#define RAG_DT(dtype, STMT)                                   \
  if (dtype == HDMOE_F32) { typedef float T; STMT; }          \
  else if (dtype == HDMOE_BF16) { typedef bf16 T; STMT; }     \
  else return HDMOE_EDTYPE;

extern "C" {

/* lens: host array [ngroups] of tokens per expert; seg: device row offsets [ngroups + 1]; srcs / dsts: host arrays of device pointers */
int hdmoe_rag_pack(void* dst, const void* const* srcs, const float* const* pos, const int* seg, const int* lens, int ngroups, int R,
                   int Sp, int C, int dtype, hipStream_t stream) {
  Rag rg; if (!mk_rag(rg, seg, lens, ngroups) || !dst || !srcs) return HDMOE_EINVAL;
  Ptrs s; FPtrs p; fill(s, srcs, ngroups); fill(p, pos, ngroups);
  RAG_DT(dtype, hipLaunchKernelGGL(rag_pack_kernel<T>, dim3(grid_for((long)R * Sp * C)), dim3(TPB), 0, stream, (T*)dst, s, p, rg, R, Sp, C))
  return hdmoe_launch_status();
}
int hdmoe_rag_pack_bwd(void* const* dsrcs, float* const* dpos, const void* ddst, const int* seg, const int* lens, int ngroups, int R,
                       int Sp, int C, int dtype, hipStream_t stream) {
  Rag rg; if (!mk_rag(rg, seg, lens, ngroups) || !ddst || !dsrcs) return HDMOE_EINVAL;
  MPtrs s; MFPtrs p; fill(s, dsrcs, ngroups); fill(p, dpos, ngroups);
  RAG_DT(dtype, hipLaunchKernelGGL(rag_pack_bwd_kernel<T>, dim3(grid_for((long)R * Sp * C)), dim3(TPB), 0, stream, s, p, (const T*)ddst, rg, R, Sp, C))
  return hdmoe_launch_status();
}
int hdmoe_rag_unpack(void* const* dsts, const void* src, const int* seg, const int* lens, int ngroups, int R, int Sp, int C, int dtype,
                     hipStream_t stream) {
  Rag rg; if (!mk_rag(rg, seg, lens, ngroups) || !src || !dsts) return HDMOE_EINVAL;
  MPtrs d; fill(d, dsts, ngroups);
  RAG_DT(dtype, hipLaunchKernelGGL(rag_unpack_kernel<T>, dim3(grid_for((long)R * Sp * C)), dim3(TPB), 0, stream, d, (const T*)src, rg, R, Sp, C))
  return hdmoe_launch_status();
}
int hdmoe_rag_unpack_bwd(void* dsrc, const void* const* ddsts, const int* seg, const int* lens, int ngroups, int R, int Sp, int C,
                         int dtype, hipStream_t stream) {
  Rag rg; if (!mk_rag(rg, seg, lens, ngroups) || !dsrc || !ddsts) return HDMOE_EINVAL;
  Ptrs d; fill(d, ddsts, ngroups);
  RAG_DT(dtype, hipLaunchKernelGGL(rag_unpack_bwd_kernel<T>, dim3(grid_for((long)R * Sp * C)), dim3(TPB), 0, stream, (T*)dsrc, d, rg, R, Sp, C))
  return hdmoe_launch_status();
}
/* y[r] = outs[g(r)][r]; row_bytes % 16 == 0 */
int hdmoe_rag_select(void* y, const void* const* outs, const int* seg, int ngroups, int R, long row_bytes, hipStream_t stream) {
  const int zeros[HDMOE_MAX_GROUPS] = {0};
  Rag rg; if (!mk_rag(rg, seg, zeros, ngroups) || !y || !outs || row_bytes % 16) return HDMOE_EINVAL;
  Ptrs o; fill(o, outs, ngroups);
  hipLaunchKernelGGL(rag_select_kernel, dim3(grid_for((long)R * (row_bytes / 16))), dim3(TPB), 0, stream, (uint4*)y, o, rg, R, row_bytes / 16);
  return hdmoe_launch_status();
}
int hdmoe_rag_select_bwd(void* const* douts, const void* dy, const int* seg, int ngroups, int R, long row_bytes, hipStream_t stream) {
  const int zeros[HDMOE_MAX_GROUPS] = {0};
  Rag rg; if (!mk_rag(rg, seg, zeros, ngroups) || !dy || !douts || row_bytes % 16) return HDMOE_EINVAL;
  MPtrs o; fill(o, douts, ngroups);
  hipLaunchKernelGGL(rag_select_bwd_kernel, dim3(grid_for((long)R * (row_bytes / 16))), dim3(TPB), 0, stream, o, (const uint4*)dy, rg, R, row_bytes / 16);
  return hdmoe_launch_status();
}
/* GroupNorm(G, C) over each row's real tokens + activation (0 none, 1 relu, 2 mp_silu); mean / rstd [R][G]; gamma / beta per expert */
int hdmoe_gn_rag_fwd(void* y, float* mean, float* rstd, const void* x, const float* const* gamma, const float* const* beta,
                     const int* seg, const int* lens, int ngroups, int R, int Sp, int C, int G, int act, float eps, int dtype,
                     hipStream_t stream) {
  Rag rg; if (!mk_rag(rg, seg, lens, ngroups) || G < 1 || G > 32 || C % G) return HDMOE_EINVAL;
  FPtrs ga, be; fill(ga, gamma, ngroups); fill(be, beta, ngroups);
  RAG_DT(dtype, hipLaunchKernelGGL(gn_rag_fwd_kernel<T>, dim3(R), dim3(TPB), 0, stream, (T*)y, mean, rstd, (const T*)x, ga, be, rg, Sp, C, G, act, eps))
  return hdmoe_launch_status();
}
int hdmoe_gn_rag_bwd(void* dx, float* const* dgamma, float* const* dbeta, const void* dy, const void* x, const float* const* gamma,
                     const float* const* beta, const float* mean, const float* rstd, const int* seg, const int* lens, int ngroups,
                     int R, int Sp, int C, int G, int act, int dtype, hipStream_t stream) {
  Rag rg; if (!mk_rag(rg, seg, lens, ngroups) || G < 1 || G > 32 || C % G) return HDMOE_EINVAL;
  FPtrs ga, be; MFPtrs dg, db; fill(ga, gamma, ngroups); fill(be, beta, ngroups); fill(dg, dgamma, ngroups); fill(db, dbeta, ngroups);
  RAG_DT(dtype, hipLaunchKernelGGL(gn_rag_bwd_kernel<T>, dim3(R), dim3(TPB), 2 * C * sizeof(float), stream, (T*)dx, dg, db, (const T*)dy, (const T*)x, ga, be,
                                   mean, rstd, rg, Sp, C, G, act))
  return hdmoe_launch_status();
}
/* LayerNorm(C) per token of [R][Sp][C] with the row's expert's affine; mean / rstd [R * Sp] */
int hdmoe_ln_rag_fwd(void* y, float* mean, float* rstd, const void* x, const float* const* gamma, const float* const* beta,
                     const int* seg, int ngroups, int R, int Sp, int C, float eps, int dtype, hipStream_t stream) {
  const int zeros[HDMOE_MAX_GROUPS] = {0};
  Rag rg; if (!mk_rag(rg, seg, zeros, ngroups)) return HDMOE_EINVAL;
  FPtrs ga, be; fill(ga, gamma, ngroups); fill(be, beta, ngroups);
#define LN_CPL(KERNEL, ...)                                                                          \
  if (C <= 32) hipLaunchKernelGGL((KERNEL<T, 1>), dim3(R), dim3(TPB), 0, stream, __VA_ARGS__);       \
  else if (C <= 64) hipLaunchKernelGGL((KERNEL<T, 2>), dim3(R), dim3(TPB), 0, stream, __VA_ARGS__);  \
  else if (C <= 128) hipLaunchKernelGGL((KERNEL<T, 4>), dim3(R), dim3(TPB), 0, stream, __VA_ARGS__); \
  else hipLaunchKernelGGL((KERNEL<T, 8>), dim3(R), dim3(TPB), 0, stream, __VA_ARGS__);
  if (C < 1 || C > 256) return HDMOE_EINVAL;
  RAG_DT(dtype, LN_CPL(ln_rag_fwd_kernel, (T*)y, mean, rstd, (const T*)x, ga, be, rg, Sp, C, eps))
  return hdmoe_launch_status();
}
int hdmoe_ln_rag_bwd(void* dx, float* const* dgamma, float* const* dbeta, const void* dy, const void* x, const float* const* gamma,
                     const float* mean, const float* rstd, const int* seg, int ngroups, int R, int Sp, int C, int dtype,
                     hipStream_t stream) {
  const int zeros[HDMOE_MAX_GROUPS] = {0};
  Rag rg; if (!mk_rag(rg, seg, zeros, ngroups)) return HDMOE_EINVAL;
  FPtrs ga; MFPtrs dg, db; fill(ga, gamma, ngroups); fill(dg, dgamma, ngroups); fill(db, dbeta, ngroups);
  if (C < 1 || C > 256) return HDMOE_EINVAL;
  RAG_DT(dtype, LN_CPL(ln_rag_bwd_kernel, (T*)dx, dg, db, (const T*)dy, (const T*)x, ga, mean, rstd, rg, Sp, C))
  return hdmoe_launch_status();
}

}  // extern "C"
