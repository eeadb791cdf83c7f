// Row N2 (next to the path): EDM_LOSS fused on the device (reference Utils/utils.py:127-172), fwd + bwd.
//   pure   = clamp(mean((D - x0)^2 / exp(lv) + lv), max 50),  lv = clamp(log_var, -10, 10)   (lv = 0 when absent)
//   bal    = clamp(lu * E * sum_e mean_b(pU)^2 + lv_ * E * sum_e mean_b(pV)^2, max 50)
//   z      = clamp(zb * mean_b clamp(logsumexp(clamp(rawU,-50,50))^2, max 100) + same for V, max 50)
//   loss   = clamp(pure + z + bal, max 50)
// out[0..4] = loss, denoising, balance, z_loss, pure_loss.  No host sync (.item()) anywhere.
#include "common.h"
#include "hdmoe.h"

namespace {

__global__ void sse_rows_kernel(float* sse, const float* d, const float* t, long L, int chunk) {
  __shared__ float sm[16];
  const int b = blockIdx.y;
  const long p0 = (long)blockIdx.x * chunk;
  const long p1 = (p0 + chunk < L) ? p0 + chunk : L;
  float acc = 0.f;
  for (long i = p0 + threadIdx.x; i < p1; i += blockDim.x) { const float e = d[(long)b * L + i] - t[(long)b * L + i]; acc += e * e; }
  acc = block_sum(acc, sm);
  if (threadIdx.x == 0) atomicAdd(&sse[b], acc);
}

// (a kernel, not hipMemsetAsync: a memset NODE of a captured hipGraph was observed to stop clearing its target once the process made
//  another allocation after the capture -- the per-sample sums then accumulated across replays; see DESIGN.md "hipGraph hazards")
__global__ void zero_f32_kernel(float* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

DEVI float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// single block: all (B) / (B,E) reductions.  aux[0..2E) = column means of pU | pV (kept for the backward)
__global__ __launch_bounds__(256) void edm_loss_reduce_kernel(float* out, float* aux, const float* sse, const float* log_var,
                                                             const float* pU, const float* pV, const float* rU, const float* rV,
                                                             int B, long L, int E, float lu, float lvit, float zb) {
  __shared__ float sm[16];
  __shared__ float colU[64], colV[64];
  const int tid = threadIdx.x;
  float pure = 0.f, den = 0.f, zu = 0.f, zv = 0.f;
  for (int b = tid; b < B; b += blockDim.x) {
    const float lv = log_var ? clampf(log_var[b], -10.f, 10.f) : 0.f;
    const float ms = sse[b] / (float)L;
    pure += ms * __expf(-lv) + lv;
    den += ms;
    for (int which = 0; which < 2; ++which) {
      const float* r = (which ? rV : rU) + (long)b * E;
      float m = -INFINITY;
      for (int e = 0; e < E; ++e) m = fmaxf(m, clampf(r[e], -50.f, 50.f));
      float s = 0.f;
      for (int e = 0; e < E; ++e) s += __expf(clampf(r[e], -50.f, 50.f) - m);
      const float lse = m + __logf(s);
      const float z = fminf(lse * lse, 100.f);
      if (which) zv += z; else zu += z;
    }
  }
  pure = block_sum(pure, sm) / (float)B;
  den = block_sum(den, sm) / (float)B;
  zu = block_sum(zu, sm) / (float)B;
  zv = block_sum(zv, sm) / (float)B;
  // column means of the gate probabilities: every thread takes rows tid, tid + 256, ... of one column at a time (a lone thread per
  // column walked B dependent loads: 36 us on the serial tail of the step); fixed order, no atomics
  for (int e = 0; e < E; ++e) {
    float a = 0.f, c = 0.f;
    for (int b = tid; b < B; b += blockDim.x) { a += pU[(long)b * E + e]; c += pV[(long)b * E + e]; }
    a = block_sum(a, sm); c = block_sum(c, sm);
    if (tid == 0) { colU[e] = a / (float)B; colV[e] = c / (float)B; aux[e] = colU[e]; aux[E + e] = colV[e]; }
  }
  __syncthreads();
  if (tid == 0) {
    float bu = 0.f, bv = 0.f;
    for (int e = 0; e < E; ++e) { bu += colU[e] * colU[e]; bv += colV[e] * colV[e]; }
    const float bal_raw = lu * E * bu + lvit * E * bv;
    const float z_raw = zb * zu + zb * zv;
    const float pure_c = fminf(pure, 50.f), bal = fminf(bal_raw, 50.f), z = fminf(z_raw, 50.f);
    const float tot = pure_c + z + bal;
    out[0] = fminf(tot, 50.f); out[1] = den; out[2] = bal; out[3] = z; out[4] = pure_c;
    // pass-through flags for the backward (clamp(max) has zero gradient beyond the bound; NaN compares false -> 0)
    aux[2 * E + 0] = (tot <= 50.f && pure <= 50.f) ? 1.f : 0.f;
    aux[2 * E + 1] = (tot <= 50.f && bal_raw <= 50.f) ? 1.f : 0.f;
    aux[2 * E + 2] = (tot <= 50.f && z_raw <= 50.f) ? 1.f : 0.f;
  }
}

// gradient of out[0] (scaled by *gin) w.r.t. denoised
__global__ void edm_loss_bwd_dense_kernel(float* dD, const float* gin, const float* aux, const float* d, const float* t,
                                          const float* log_var, int B, long L, int E, long n) {
  const float g = gin[0] * aux[2 * E + 0];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long b = i / L;
    const float lv = log_var ? clampf(log_var[b], -10.f, 10.f) : 0.f;
    dD[i] = g * 2.f * (d[i] - t[i]) * __expf(-lv) / ((float)B * (float)L);
  }
}
// ... w.r.t. log_var, gate probs and raw logits (one thread per sample)
__global__ void edm_loss_bwd_small_kernel(float* dlv, float* dpU, float* dpV, float* drU, float* drV, const float* gin, const float* aux,
                                          const float* sse, const float* log_var, const float* rU, const float* rV, int B, long L, int E,
                                          float lu, float lvit, float zb) {
  const float g = gin[0];
  const float gp = g * aux[2 * E + 0], gb = g * aux[2 * E + 1], gz = g * aux[2 * E + 2];
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
    if (dlv) {
      const float raw = log_var[b];
      const float lv = clampf(raw, -10.f, 10.f);
      const bool in = raw >= -10.f && raw <= 10.f;
      dlv[b] = in ? gp * (1.f - sse[b] / (float)L * __expf(-lv)) / (float)B : 0.f;
    }
    for (int which = 0; which < 2; ++which) {
      const float* r = (which ? rV : rU) + (long)b * E;
      float* dr = (which ? drV : drU) + (long)b * E;
      float* dp = (which ? dpV : dpU) + (long)b * E;
      const float lam = which ? lvit : lu;
      float m = -INFINITY;
      for (int e = 0; e < E; ++e) m = fmaxf(m, clampf(r[e], -50.f, 50.f));
      float s = 0.f;
      for (int e = 0; e < E; ++e) s += __expf(clampf(r[e], -50.f, 50.f) - m);
      const float lse = m + __logf(s);
      const float k = (lse * lse <= 100.f) ? gz * zb * 2.f * lse / (float)B : 0.f;
      for (int e = 0; e < E; ++e) {
        const bool in = r[e] >= -50.f && r[e] <= 50.f;
        dr[e] = in ? k * __expf(clampf(r[e], -50.f, 50.f) - m) / s : 0.f;
        dp[e] = gb * lam * E * 2.f * aux[which * E + e] / (float)B;
      }
    }
  }
}

}  // namespace

extern "C" {

// out: 5 floats; aux: 2E+3 floats; sse: B floats (zeroed here).  denoised/target: fp32 [B][L] (any consistent layout).
int hdmoe_edm_loss_fwd(float* out, float* aux, float* sse, const float* denoised, const float* target, const float* log_var,
                       const float* pU, const float* pV, const float* rU, const float* rV, int B, long L, int E, float unet_bal,
                       float vit_bal, float z_bal, hipStream_t stream) {
  if (B < 1 || B > 65535 || E < 1 || E > 64 || L < 1) return HDMOE_EINVAL;
  hipLaunchKernelGGL(zero_f32_kernel, dim3(cdiv(B, 256)), dim3(256), 0, stream, sse, B);
  const int chunk = 4096;
  hipLaunchKernelGGL(sse_rows_kernel, dim3(cdiv(L, chunk), B), dim3(256), 0, stream, sse, denoised, target, L, chunk);
  hipLaunchKernelGGL(edm_loss_reduce_kernel, dim3(1), dim3(256), 0, stream, out, aux, sse, log_var, pU, pV, rU, rV, B, L, E, unet_bal,
                     vit_bal, z_bal);
  return hdmoe_launch_status();
}
int hdmoe_edm_loss_bwd(float* dD, float* dlv, float* dpU, float* dpV, float* drU, float* drV, const float* gin, const float* aux,
                       const float* sse, const float* denoised, const float* target, const float* log_var, const float* rU,
                       const float* rV, int B, long L, int E, float unet_bal, float vit_bal, float z_bal, hipStream_t stream) {
  if (B < 1 || E < 1 || E > 64 || L < 1) return HDMOE_EINVAL;
  const long n = (long)B * L;
  long blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(edm_loss_bwd_dense_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dD, gin, aux, denoised, target, log_var, B, L, E, n);
  hipLaunchKernelGGL(edm_loss_bwd_small_kernel, dim3(cdiv(B, 256)), dim3(256), 0, stream, dlv, dpU, dpV, drU, drV, gin, aux, sse, log_var,
                     rU, rV, B, L, E, unet_bal, vit_bal, z_bal);
  return hdmoe_launch_status();
}

}  // extern "C"
