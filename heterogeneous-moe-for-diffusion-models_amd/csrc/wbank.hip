// Multi-tensor weight bank: every MP_Conv weight of a model is prepared by ONE launch per forward (normalise -> gain/sqrt(fan_in)
// -> cast -> forward image [tap][O][Ipad] + flipped dgrad image [tap'][I][Opad]) and every weight gradient is finished by ONE
// launch per backward (normalisation backward of the [tap][O][I] wgrad slab, accumulated straight into the parameter's .grad).
// Same arithmetic as wprep_fwd / wprep_bwd in conv.hip (reference models/model_internals.py:253-259), ~500 launches per step fewer.
#include "common.h"
#include "hdmoe.h"

namespace {

struct __attribute__((aligned(8))) WBDesc {   // mirrored byte-for-byte by hdmoe_hip/bank.py (numpy structured dtype)
  unsigned long long w_raw, wf, wd, G, dw;   // device addresses
  int O, I, kh, kw, Ipad, Opad, dtype, normalize, mutate_ok, pad0;
  float gain, out_scale;
  long long wf_plane, wd_plane;              // dtype == HDMOE_F32S: elements between the hi and the lo bf16 image
};

template <typename T> DEVI void wb_store(void* base, long idx, float v) { ((T*)base)[idx] = from_f<T>(v); }

constexpr int WB_PREP_ROW = 3200;            // rows up to this many elements are normalised in registers + LDS (one global read of the row)

__global__ __launch_bounds__(128) void wbank_prep_kernel(const WBDesc* descs, const int2* rows, int mutate) {
  __shared__ float sm[16];
  __shared__ float rowl[WB_PREP_ROW];
  const int2 ro = rows[blockIdx.x];
  const WBDesc d = descs[ro.x];
  const int o = ro.y, tid = threadIdx.x;
  const int taps = d.kh * d.kw, fan = d.I * taps;
  float* w = (float*)d.w_raw + (long)o * fan;
  const bool cached = fan <= WB_PREP_ROW;      // (uniform over the block)
  float scale = 1.f;
  if (cached) {
    // one pass over HBM: the row sits in LDS from here on (before: norm pass, in-place rewrite, second norm pass and a strided image
    // pass each paid a global round trip -- 127 us for the model's ~35 k rows, serial at the head of every training step)
    float ss = 0.f;
    for (int e = tid; e < fan; e += blockDim.x) { const float v = w[e]; rowl[e] = v; ss += v * v; }
    if (d.normalize) {
      const float c = rsqrtf((float)fan);
      ss = block_sum(ss, sm);
      float inv = 1.f / (1e-4f + sqrtf(ss) * c);
      if (mutate && d.mutate_ok) {
        float s2 = 0.f;
        for (int e = tid; e < fan; e += blockDim.x) { const float v = rowl[e] * inv; rowl[e] = v; w[e] = v; s2 += v * v; }
        s2 = block_sum(s2, sm);
        inv = 1.f / (1e-4f + sqrtf(s2) * c);
      }
      scale = inv * d.gain * c;
    }
    __syncthreads();
  } else if (d.normalize) {
    const float c = rsqrtf((float)fan);
    float ss = 0.f;
    for (int e = tid; e < fan; e += blockDim.x) { const float v = w[e]; ss += v * v; }
    ss = block_sum(ss, sm);
    float inv = 1.f / (1e-4f + sqrtf(ss) * c);
    if (mutate && d.mutate_ok) {
      for (int e = tid; e < fan; e += blockDim.x) w[e] = w[e] * inv;
      __syncthreads();
      float s2 = 0.f;
      for (int e = tid; e < fan; e += blockDim.x) { const float v = w[e]; s2 += v * v; }
      s2 = block_sum(s2, sm);
      inv = 1.f / (1e-4f + sqrtf(s2) * c);
    }
    scale = inv * d.gain * c;
  }
  const bool f32 = d.dtype == HDMOE_F32;
  // image pass in (tap, input channel) order: the forward image [tap][O][Ipad] is written in runs of I consecutive elements; only the
  // flipped image stays scattered
  for (int e2 = tid; e2 < fan; e2 += blockDim.x) {
    const int t = e2 / d.I, i = e2 - t * d.I;
    const float v = (cached ? rowl[i * taps + t] : w[i * taps + t]) * scale;
    const long fi = ((long)t * d.O + o) * d.Ipad + i;
    const long di = ((long)(taps - 1 - t) * d.I + i) * d.Opad + o;
    if (f32) { wb_store<float>((void*)d.wf, fi, v); if (d.wd) wb_store<float>((void*)d.wd, di, v); }
    else { wb_store<bf16>((void*)d.wf, fi, v); if (d.wd) wb_store<bf16>((void*)d.wd, di, v); }
    if (d.dtype == HDMOE_F32S) {                             // split-bf16: the lo plane holds bf16(v - hi)
      const float lo = v - (float)(bf16)v;
      wb_store<bf16>((void*)d.wf, d.wf_plane + fi, lo);
      if (d.wd) wb_store<bf16>((void*)d.wd, d.wd_plane + di, lo);
    }
  }
}

constexpr int WB_LDS_FLOATS = 2560;          // rows up to this many elements (+ padding) stage their wgrad slab row through LDS (10 KB: 16 blocks per CU stay resident)

__global__ __launch_bounds__(128) void wbank_bwd_kernel(const WBDesc* descs, const int2* rows) {
  __shared__ float sm[16];
  extern __shared__ float gl[];               // [tap][I + 1]: the row's slice of the [tap][O][I] slab, read coalesced, used in weight order
  const int2 ro = rows[blockIdx.x];
  const WBDesc d = descs[ro.x];
  const int o = ro.y, tid = threadIdx.x;
  const int taps = d.kh * d.kw, fan = d.I * taps;
  const float* w = (const float*)d.w_raw + (long)o * fan;
  const float* G = (const float*)d.G;
  float* dw = (float*)d.dw + (long)o * fan;
  if (taps > 1 && taps * (d.I + 1) <= WB_LDS_FLOATS) {
    const int Ip = d.I + 1;                   // (+1: consecutive taps of one input channel fall into different banks)
    for (int e2 = tid; e2 < fan; e2 += blockDim.x) {
      const int t = e2 / d.I, i = e2 - t * d.I;
      gl[t * Ip + i] = G[((long)t * d.O + o) * d.I + i];
    }
    __syncthreads();
    if (!d.normalize) {
      for (int e = tid; e < fan; e += blockDim.x) { const int i = e / taps, t = e - i * taps; dw[e] += d.out_scale * gl[t * Ip + i]; }
      return;
    }
    const float c = rsqrtf((float)fan);
    float ss = 0.f, gw = 0.f;
    for (int e = tid; e < fan; e += blockDim.x) {
      const int i = e / taps, t = e - i * taps;
      const float v = w[e];
      ss += v * v;
      gw += v * gl[t * Ip + i];
    }
    ss = block_sum(ss, sm);
    gw = block_sum(gw, sm);
    const float n = sqrtf(ss);
    const float dd = 1e-4f + n * c;
    const float s = d.gain * c;
    const float k1 = d.out_scale * s / dd;
    const float k2 = n > 0.f ? d.out_scale * s * c * gw / (dd * dd * n) : 0.f;
    for (int e = tid; e < fan; e += blockDim.x) {
      const int i = e / taps, t = e - i * taps;
      dw[e] += k1 * gl[t * Ip + i] - k2 * w[e];
    }
    return;
  }
  if (!d.normalize) {
    for (int e = tid; e < fan; e += blockDim.x) {
      const int i = e / taps, t = e - i * taps;
      dw[e] += d.out_scale * G[((long)t * d.O + o) * d.I + i];
    }
    return;
  }
  const float c = rsqrtf((float)fan);
  float ss = 0.f, gw = 0.f;
  for (int e = tid; e < fan; e += blockDim.x) {
    const int i = e / taps, t = e - i * taps;
    const float v = w[e];
    ss += v * v;
    gw += v * G[((long)t * d.O + o) * d.I + i];
  }
  ss = block_sum(ss, sm);
  gw = block_sum(gw, sm);
  const float n = sqrtf(ss);
  const float dd = 1e-4f + n * c;
  const float s = d.gain * c;
  const float k1 = d.out_scale * s / dd;
  const float k2 = n > 0.f ? d.out_scale * s * c * gw / (dd * dd * n) : 0.f;
  for (int e = tid; e < fan; e += blockDim.x) {
    const int i = e / taps, t = e - i * taps;
    dw[e] += k1 * G[((long)t * d.O + o) * d.I + i] - k2 * w[e];
  }
}

}  // namespace

extern "C" {

int hdmoe_wbank_desc_bytes(void) { return (int)sizeof(WBDesc); }

// descs: device array of WBDesc; rows: device int32 pairs (descriptor index, output row); nrows blocks
int hdmoe_wbank_prep(const void* descs, const int* rows, int nrows, int mutate, hipStream_t stream) {
  if (nrows < 1) return HDMOE_OK;
  hipLaunchKernelGGL(wbank_prep_kernel, dim3(nrows), dim3(128), 0, stream, (const WBDesc*)descs, (const int2*)rows, mutate);
  return hdmoe_launch_status();
}
int hdmoe_wbank_bwd(const void* descs, const int* rows, int nrows, hipStream_t stream) {
  if (nrows < 1) return HDMOE_OK;
  hipLaunchKernelGGL(wbank_bwd_kernel, dim3(nrows), dim3(128), WB_LDS_FLOATS * sizeof(float), stream, (const WBDesc*)descs, (const int2*)rows);
  return hdmoe_launch_status();
}

}  // extern "C"
