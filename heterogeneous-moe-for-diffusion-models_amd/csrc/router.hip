// K1 / K2: noisy-top-k router head and the sync-free sample dispatch.
//   head     : Router.forward tail, reference models/model_components.py:155-168
//   dispatch : router_to_unet_experts, reference models/model_config1.py:11-39 -- the reference gathers x[mask]
//              per expert behind a host sync (mask.any()); here a single-workgroup plan kernel builds the
//              expert-contiguous permutation on the device (LDS histogram + wavefront ballot scan), so the
//              whole step stays capturable in a hipGraph.
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

// ---------------------------------------------------------------- router head (one thread per row; E <= 64)
__global__ void router_head_fwd_kernel(float* sparse, float* probs, float* xout, int* idx, const float* logits, const float* noise,
                                       const float* mask, int E, int k, long B) {
  GRID_STRIDE(b, B) {
    const float* lg = logits + b * E;
    float* xo = xout + b * E;
    float mx = -INFINITY;
    for (int e = 0; e < E; ++e) {
      float v = lg[e];
      if (noise) v += noise[b * E + e];
      if (mask && mask[b * E + e] == 0.f) v = -INFINITY;
      xo[e] = v;
      mx = fmaxf(mx, v);
    }
    float sum = 0.f;
    for (int e = 0; e < E; ++e) sum += expf(xo[e] - mx);
    for (int e = 0; e < E; ++e) { probs[b * E + e] = expf(xo[e] - mx) / sum; sparse[b * E + e] = 0.f; }
    // top-k by repeated argmax; ties -> lowest index (torch.topk's tie order is implementation-defined)
    unsigned long long taken = 0ull;
    float top = 0.f;
    for (int j = 0; j < k; ++j) {
      int bi = -1; float bv = 0.f;
      for (int e = 0; e < E; ++e) {
        if ((taken >> e) & 1ull) continue;
        if (bi < 0 || xo[e] > bv) { bi = e; bv = xo[e]; }
      }
      taken |= 1ull << bi;
      idx[b * k + j] = bi;
      if (j == 0) top = bv;
    }
    float ws = 0.f;
    for (int j = 0; j < k; ++j) ws += expf(xo[idx[b * k + j]] - top);
    for (int j = 0; j < k; ++j) sparse[b * E + idx[b * k + j]] = expf(xo[idx[b * k + j]] - top) / ws;
  }
}

__global__ void router_head_bwd_kernel(float* dlogits, const float* dsparse, const float* dprobs, const float* dxout, const float* sparse,
                                       const float* probs, const int* idx, const float* mask, int E, int k, long B) {
  GRID_STRIDE(b, B) {
    float* dl = dlogits + b * E;
    float dot = 0.f;
    if (dprobs) for (int e = 0; e < E; ++e) dot += probs[b * E + e] * dprobs[b * E + e];
    for (int e = 0; e < E; ++e) {
      float g = dxout ? dxout[b * E + e] : 0.f;
      if (dprobs) g += probs[b * E + e] * (dprobs[b * E + e] - dot);
      dl[e] = g;
    }
    if (dsparse) {
      float dw = 0.f;
      for (int j = 0; j < k; ++j) { const int e = idx[b * k + j]; dw += sparse[b * E + e] * dsparse[b * E + e]; }
      for (int j = 0; j < k; ++j) { const int e = idx[b * k + j]; dl[e] += sparse[b * E + e] * (dsparse[b * E + e] - dw); }
    }
    if (mask) for (int e = 0; e < E; ++e) if (mask[b * E + e] == 0.f) dl[e] = 0.f;
  }
}

// ---------------------------------------------------------------- dispatch plan (single workgroup)
// routed pair (b, e) <=> sparse[b][e] > 0 (NaN > 0 is false: an all-masked sample is routed nowhere, as in the reference).
// Output order: expert-major, sample-minor (stable) == the order x[mask] produces per expert.
__global__ __launch_bounds__(1024) void dispatch_plan_kernel(int* perm, int* row_expert, float* row_w, int* inv, int* seg,
                                                            const float* sparse, int B, int E, int kcap) {
  __shared__ int cnt[64 + 1];
  __shared__ int wtot[16];
  __shared__ int base;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;
  const int R = B * kcap;
  for (int r = tid; r < R; r += blockDim.x) { perm[r] = -1; row_expert[r] = -1; row_w[r] = 0.f; inv[r] = -1; }
  if (tid <= E) cnt[tid] = 0;
  __syncthreads();
  for (int b = tid; b < B; b += blockDim.x)
    for (int e = 0; e < E; ++e)
      if (sparse[(long)b * E + e] > 0.f) atomicAdd(&cnt[e], 1);
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int e = 0; e <= E; ++e) { const int c = e < E ? cnt[e] : 0; cnt[e] = run; seg[e] = run < R ? run : R; run += c; }
  }
  __syncthreads();
  for (int e = 0; e < E; ++e) {
    if (tid == 0) base = cnt[e];
    __syncthreads();
    for (int b0 = 0; b0 < B; b0 += blockDim.x) {
      const int b = b0 + tid;
      const float w = b < B ? sparse[(long)b * E + e] : 0.f;
      const bool f = w > 0.f;
      const unsigned long long bal = __ballot(f);
      const int pre = __popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) wtot[wv] = __popcll(bal);
      __syncthreads();
      int off = 0, tot = 0;
      for (int i = 0; i < nw; ++i) { if (i < wv) off += wtot[i]; tot += wtot[i]; }
      if (f) {
        const int pos = base + off + pre;
        int slot = 0;
        for (int e2 = 0; e2 < e; ++e2) slot += sparse[(long)b * E + e2] > 0.f ? 1 : 0;
        if (pos < R) {
          perm[pos] = b; row_expert[pos] = e; row_w[pos] = w;
          if (slot < kcap) inv[b * kcap + slot] = pos;
        }
      }
      __syncthreads();
      if (tid == 0) base += tot;
      __syncthreads();
    }
  }
}

// dst[r][:] = src[perm[r]][:]  (zeros for unused rows); 16-byte units
__global__ void gather_rows_kernel(uint4* dst, const uint4* src, const int* perm, long Lv, long n) {
  GRID_STRIDE(i, n) {
    const long r = i / Lv, c = i - r * Lv;
    const int s = perm[r];
    dst[i] = s >= 0 ? src[(long)s * Lv + c] : make_uint4(0, 0, 0, 0);
  }
}
template <typename T>
__global__ void gather_rows_scalar_kernel(T* dst, const T* src, const int* perm, long L, long n) {
  GRID_STRIDE(i, n) {
    const long r = i / L, c = i - r * L;
    const int s = perm[r];
    dst[i] = s >= 0 ? src[(long)s * L + c] : from_f<T>(0.f);
  }
}
// out[b][:] = sum_j w[r_j] * ys[r_j][:], r_j = inv[b][j]  (weights optional)
template <typename T>
__global__ void combine_rows_fwd_kernel(T* out, const T* ys, const int* inv, const float* row_w, int kcap, long L, long n) {
  GRID_STRIDE(i, n) {
    const long b = i / L, c = i - b * L;
    float acc = 0.f;
    for (int j = 0; j < kcap; ++j) {
      const int r = inv[b * kcap + j];
      if (r >= 0) {
        const float w = row_w ? row_w[r] : 1.f;
        if (w > 0.f) acc += w * to_f(ys[(long)r * L + c]);
      }
    }
    out[i] = from_f<T>(acc);
  }
}
// dys[r][:] = w[r] * dout[perm[r]][:] ;  dsparse[perm[r]][row_expert[r]] += <dout[perm[r]], ys[r]>
template <typename T>
__global__ void combine_rows_bwd_kernel(T* dys, float* dsparse, const T* dout, const T* ys, const int* perm, const int* row_expert,
                                        const float* row_w, int E, long L, int chunk) {
  __shared__ float sm[16];
  const long nck = (L + chunk - 1) / chunk;
  const long r = blockIdx.x / nck;                              // 1-D grid: (row, chunk) pairs, no 65535-row limit
  const int b = perm[r];
  const long p0 = (long)(blockIdx.x - r * nck) * chunk;
  const long p1 = (p0 + chunk < L) ? p0 + chunk : L;
  const float w = b >= 0 ? (row_w ? row_w[r] : 1.f) : 0.f;
  float acc = 0.f;
  for (long i = p0 + threadIdx.x; i < p1; i += blockDim.x) {
    const float g = b >= 0 ? to_f(dout[(long)b * L + i]) : 0.f;
    dys[(long)r * L + i] = from_f<T>(w * g);
    if (dsparse && b >= 0) acc += g * to_f(ys[(long)r * L + i]);
  }
  if (dsparse) {
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0 && b >= 0) atomicAdd(&dsparse[(long)b * E + row_expert[r]], acc);
  }
}

// counts ACCUMULATE: several forwards may share one optimizer step (gradient accumulation); FusedAdamW.step clears them behind the update
__global__ void seg_counts_kernel(float* counts, const int* seg, int E) {
  if ((int)threadIdx.x < E) counts[threadIdx.x] += (float)(seg[threadIdx.x + 1] - seg[threadIdx.x]);
}
// the same from the sparse gate weights (B, E) when no dispatch plan exists (experts evaluated on the whole batch): rows with weight > 0
__global__ void route_counts_kernel(float* counts, const float* sparse, int B, int E) {
  const int e = blockIdx.x;
  float c = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) c += sparse[(long)b * E + e] > 0.f ? 1.f : 0.f;
  __shared__ float sm[16];
  c = block_sum(c, sm);
  if (threadIdx.x == 0) counts[e] += c;
}

}  // namespace

extern "C" {

int hdmoe_router_head_fwd(float* sparse, float* probs, float* xout, int* idx, const float* logits, const float* noise,
                          const float* mask, long B, int E, int k, hipStream_t stream) {
  if (E < 1 || E > 64 || k < 1 || k > E) return HDMOE_EINVAL;
  hipLaunchKernelGGL(router_head_fwd_kernel, dim3(grid_for(B)), dim3(TPB), 0, stream, sparse, probs, xout, idx, logits, noise, mask, E, k, B);
  return hdmoe_launch_status();
}
int hdmoe_router_head_bwd(float* dlogits, const float* dsparse, const float* dprobs, const float* dxout, const float* sparse,
                          const float* probs, const int* idx, const float* mask, long B, int E, int k, hipStream_t stream) {
  if (E < 1 || E > 64 || k < 1 || k > E) return HDMOE_EINVAL;
  hipLaunchKernelGGL(router_head_bwd_kernel, dim3(grid_for(B)), dim3(TPB), 0, stream, dlogits, dsparse, dprobs, dxout, sparse, probs, idx, mask, E, k, B);
  return hdmoe_launch_status();
}
int hdmoe_dispatch_plan(int* perm, int* row_expert, float* row_w, int* inv, int* seg, const float* sparse, int B, int E, int kcap,
                        hipStream_t stream) {
  if (B < 1 || E < 1 || E > 64 || kcap < 1 || kcap > E) return HDMOE_EINVAL;
  hipLaunchKernelGGL(dispatch_plan_kernel, dim3(1), dim3(1024), 0, stream, perm, row_expert, row_w, inv, seg, sparse, B, E, kcap);
  return hdmoe_launch_status();
}
// counts[e] = rows routed to expert e in this step (float): the optimizer skips the tensors of an expert without rows, like the reference,
// whose `if not mask.any(): continue` leaves such an expert's gradients None (models/model_config1.py:26-29)
int hdmoe_seg_counts(float* counts, const int* seg, int E, hipStream_t stream) {
  if (!counts || !seg || E < 1 || E > 64) return HDMOE_EINVAL;
  hipLaunchKernelGGL(seg_counts_kernel, dim3(1), dim3(64), 0, stream, counts, seg, E);
  return hdmoe_launch_status();
}
int hdmoe_route_counts(float* counts, const float* sparse, int B, int E, hipStream_t stream) {
  if (!counts || !sparse || B < 1 || E < 1 || E > 64) return HDMOE_EINVAL;
  hipLaunchKernelGGL(route_counts_kernel, dim3(E), dim3(256), 0, stream, counts, sparse, B, E);
  return hdmoe_launch_status();
}
int hdmoe_gather_rows(void* dst, const void* src, const int* perm, long R, long L, int dtype, hipStream_t stream) {
  const long esz = dtype == HDMOE_F32 ? 4 : 2;
  if (dtype != HDMOE_F32 && dtype != HDMOE_BF16) return HDMOE_EDTYPE;
  if ((L * esz) % 16 == 0 && (uintptr_t)dst % 16 == 0 && (uintptr_t)src % 16 == 0) {
    const long Lv = L * esz / 16, n = R * Lv;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(n)), dim3(TPB), 0, stream, (uint4*)dst, (const uint4*)src, perm, Lv, n);
  } else if (dtype == HDMOE_F32) {
    hipLaunchKernelGGL(gather_rows_scalar_kernel<float>, dim3(grid_for(R * L)), dim3(TPB), 0, stream, (float*)dst, (const float*)src, perm, L, R * L);
  } else {
    hipLaunchKernelGGL(gather_rows_scalar_kernel<bf16>, dim3(grid_for(R * L)), dim3(TPB), 0, stream, (bf16*)dst, (const bf16*)src, perm, L, R * L);
  }
  return hdmoe_launch_status();
}
int hdmoe_combine_rows_fwd(void* out, const void* ys, const int* inv, const float* row_w, long B, int kcap, long L, int dtype,
                           hipStream_t stream) {
  const long n = B * L;
  if (dtype == HDMOE_F32) hipLaunchKernelGGL(combine_rows_fwd_kernel<float>, dim3(grid_for(n)), dim3(TPB), 0, stream, (float*)out, (const float*)ys, inv, row_w, kcap, L, n);
  else if (dtype == HDMOE_BF16) hipLaunchKernelGGL(combine_rows_fwd_kernel<bf16>, dim3(grid_for(n)), dim3(TPB), 0, stream, (bf16*)out, (const bf16*)ys, inv, row_w, kcap, L, n);
  else return HDMOE_EDTYPE;
  return hdmoe_launch_status();
}
int hdmoe_combine_rows_bwd(void* dys, float* dsparse, const void* dout, const void* ys, const int* perm, const int* row_expert,
                           const float* row_w, long R, int E, long L, int dtype, hipStream_t stream) {
  const int chunk = 8192;
  if (R < 0 || L < 1 || R * (long)cdiv(L, chunk) >= (1l << 31)) return HDMOE_EINVAL;
  if (R == 0) return HDMOE_OK;
  dim3 grid((unsigned)(R * (long)cdiv(L, chunk)));
  if (dtype == HDMOE_F32) hipLaunchKernelGGL(combine_rows_bwd_kernel<float>, grid, dim3(TPB), 0, stream, (float*)dys, dsparse, (const float*)dout, (const float*)ys, perm, row_expert, row_w, E, L, chunk);
  else if (dtype == HDMOE_BF16) hipLaunchKernelGGL(combine_rows_bwd_kernel<bf16>, grid, dim3(TPB), 0, stream, (bf16*)dys, dsparse, (const bf16*)dout, (const bf16*)ys, perm, row_expert, row_w, E, L, chunk);
  else return HDMOE_EDTYPE;
  return hdmoe_launch_status();
}
int hdmoe_version(void) { return 300; }

}  // extern "C"
