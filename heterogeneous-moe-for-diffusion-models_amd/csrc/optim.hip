// Row N3 (the step right after the path): fused multi-tensor gradient-norm clip + AdamW (reference Utils/training.py:55-65,195-197:
// torch.optim.AdamW with 4 LR groups, torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)).
// One launch walks every parameter tensor through a (tensor, chunk) table; the clip coefficient is read from device memory, so
// norm -> clip -> update needs no host sync.
#include "common.h"
#include "hdmoe.h"

namespace {

struct __attribute__((aligned(8))) OptDesc {   // mirrored by hdmoe_hip/optim.py
  unsigned long long p, g, m, v;              // device addresses (m, v may be 0 for norm/scale-only tables)
  unsigned long long step;                    // float: this tensor's own AdamW step count (torch.optim.AdamW keeps one per tensor), or 0
  unsigned long long use;                     // float: > 0 when the tensor received a gradient this step (rows routed to its expert), or 0 = always
  long numel;
  int group, pad0;
};
constexpr int OPT_CHUNK = 4096;

// Two deterministic passes (no float atomics): the clip coefficient multiplies every update, so a run-to-run difference in the last bit of
// the norm would let data-parallel replicas drift apart bit by bit.
__global__ __launch_bounds__(256) void mt_sumsq_kernel(float* partial, const OptDesc* descs, const int2* chunks) {
  __shared__ float sm[16];
  const int2 c = chunks[blockIdx.x];
  const OptDesc d = descs[c.x];
  const float* g = (const float*)d.g;
  const long i0 = (long)c.y * OPT_CHUNK;
  const long i1 = i0 + OPT_CHUNK < d.numel ? i0 + OPT_CHUNK : d.numel;
  float acc = 0.f;
  for (long i = i0 + threadIdx.x; i < i1; i += 256) { const float v = g[i]; acc += v * v; }
  acc = block_sum(acc, sm);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
__global__ __launch_bounds__(1024) void mt_sumsq_final_kernel(float* out, const float* partial, int n) {
  __shared__ float sm[16];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) acc += partial[i];       // fixed order per thread, fixed tree below
  acc = block_sum(acc, sm);
  if (threadIdx.x == 0) out[0] = acc;
}
// coef = min(1, max_norm / (sqrt(sumsq) + 1e-6))  (torch.nn.utils.clip_grad_norm_)
DEVI float clip_coef(const float* sumsq, float max_norm) {
  if (!sumsq) return 1.f;
  return fminf(1.f, max_norm / (sqrtf(*sumsq) + 1e-6f));
}
__global__ __launch_bounds__(256) void mt_scale_kernel(const OptDesc* descs, const int2* chunks, const float* sumsq, float max_norm) {
  const float coef = clip_coef(sumsq, max_norm);
  const int2 c = chunks[blockIdx.x];
  const OptDesc d = descs[c.x];
  float* g = (float*)d.g;
  const long i0 = (long)c.y * OPT_CHUNK;
  const long i1 = i0 + OPT_CHUNK < d.numel ? i0 + OPT_CHUNK : d.numel;
  for (long i = i0 + threadIdx.x; i < i1; i += 256) g[i] *= coef;
}
struct AdamArgs { float lr[8], wd[8]; float beta1, beta2, eps, max_norm; };
// A tensor whose expert received no sample this step is skipped altogether -- the reference leaves its .grad None and torch.optim.AdamW
// then applies neither weight decay nor moment decay nor a step-count increment (reference model_config1.py:26-29 + Utils/training.py:195-197).
__global__ __launch_bounds__(256) void mt_step_kernel(const OptDesc* descs, int ntensors) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= ntensors) return;
  const OptDesc d = descs[t];
  if (!d.step) return;
  if (d.use && !(*(const float*)d.use > 0.f)) return;
  *(float*)d.step += 1.f;
}
__global__ __launch_bounds__(256) void mt_adamw_kernel(const OptDesc* descs, const int2* chunks, const float* sumsq, AdamArgs a) {
  const int2 c = chunks[blockIdx.x];
  const OptDesc d = descs[c.x];
  if (d.use && !(*(const float*)d.use > 0.f)) return;
  const float coef = clip_coef(sumsq, a.max_norm);
  float* p = (float*)d.p; const float* g = (const float*)d.g; float* m = (float*)d.m; float* v = (float*)d.v;
  const float lr = a.lr[d.group], wd = a.wd[d.group];
  const float step = *(const float*)d.step;                  // already advanced by mt_step_kernel
  const float bc1 = 1.f - powf(a.beta1, step), bc2_sqrt = sqrtf(1.f - powf(a.beta2, step));
  const float step_size = lr / bc1;
  const long i0 = (long)c.y * OPT_CHUNK;
  const long i1 = i0 + OPT_CHUNK < d.numel ? i0 + OPT_CHUNK : d.numel;
  for (long i = i0 + threadIdx.x; i < i1; i += 256) {
    const float gi = g[i] * coef;
    float pi = p[i] * (1.f - lr * wd);                       // decoupled weight decay
    const float mi = a.beta1 * m[i] + (1.f - a.beta1) * gi;
    const float vi = a.beta2 * v[i] + (1.f - a.beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    pi -= step_size * mi / (sqrtf(vi) / bc2_sqrt + a.eps);
    p[i] = pi;
  }
}

}  // namespace

extern "C" {

int hdmoe_opt_desc_bytes(void) { return (int)sizeof(OptDesc); }
// sumsq (1 float) = sum over all tensors of g^2; ws: nchunks floats of scratch
int hdmoe_mt_sumsq(float* sumsq, const void* descs, const int* chunks, int nchunks, float* ws, hipStream_t stream) {
  if (!sumsq || (nchunks > 0 && !ws)) return HDMOE_EINVAL;
  if (nchunks > 0) hipLaunchKernelGGL(mt_sumsq_kernel, dim3(nchunks), dim3(256), 0, stream, ws, (const OptDesc*)descs, (const int2*)chunks);
  hipLaunchKernelGGL(mt_sumsq_final_kernel, dim3(1), dim3(1024), 0, stream, sumsq, ws, nchunks > 0 ? nchunks : 0);
  return hdmoe_launch_status();
}
int hdmoe_mt_clip_scale(const void* descs, const int* chunks, int nchunks, const float* sumsq, float max_norm, hipStream_t stream) {
  if (nchunks > 0) hipLaunchKernelGGL(mt_scale_kernel, dim3(nchunks), dim3(256), 0, stream, (const OptDesc*)descs, (const int2*)chunks, sumsq, max_norm);
  return hdmoe_launch_status();
}
// lr / wd: host arrays of ngroups (<= 8) floats; sumsq may be NULL (no clipping).  Every descriptor carries its tensor's step counter (device
// float, advanced here) and optionally a "used this step" flag; bias corrections are computed on the device from the per-tensor count.
int hdmoe_mt_adamw(const void* descs, const int* chunks, int nchunks, int ntensors, const float* sumsq, float max_norm, const float* group_lr,
                   const float* group_wd, int ngroups, float beta1, float beta2, float eps, hipStream_t stream) {
  if (ngroups < 1 || ngroups > 8 || ntensors < 0) return HDMOE_EINVAL;
  AdamArgs a;
  for (int i = 0; i < 8; ++i) { a.lr[i] = group_lr[i < ngroups ? i : 0]; a.wd[i] = group_wd[i < ngroups ? i : 0]; }
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.max_norm = max_norm;
  if (ntensors > 0) hipLaunchKernelGGL(mt_step_kernel, dim3(cdiv(ntensors, 256)), dim3(256), 0, stream, (const OptDesc*)descs, ntensors);
  if (nchunks > 0) hipLaunchKernelGGL(mt_adamw_kernel, dim3(nchunks), dim3(256), 0, stream, (const OptDesc*)descs, (const int2*)chunks, sumsq, a);
  return hdmoe_launch_status();
}

}  // extern "C"
