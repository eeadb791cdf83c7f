// Pointwise forward / dgrad with a LONG contraction and few outputs: y[m][o] = alpha * sum_k x[m][k] * w[o][k] + beta * res[m][o],
// K = Cin >= 512 (HDMOE_KGEMM_MINK), Cout <= 64, one expert.  These are the ViT experts' patch embedding (K = C * p^2 up to 8192 input features per
// token, reference models/model_components.py:670-679) and the input gradient of unpatch_proj (:700-706).  The general 1x1 path
// (conv.hip: conv_fwd5) tiles positions and walks K inside one workgroup: with 2048 tokens that is 8 workgroups on a 256-CU chip
// (285 us for a 33 MB read).  Here both operands are K-contiguous in memory, so MFMA fragments are plain 16-byte global loads (no
// LDS): a workgroup owns 32 rows, its eight waves split K in 64-element chunks (a lane reads 64 contiguous bytes of its row per
// chunk = four k-steps; the k <-> MFMA-slot assignment is arbitrary as long as both operands use the same one), and the partial
// sums meet in LDS in a fixed order (deterministic forward).
#include <stdlib.h>
#include "common.h"
#include "conv_args.h"
#include "hdmoe.h"

namespace {

struct KArgs { const bf16* x; const bf16* w; bf16* y; const bf16* res; long M; int K, O; float alpha, beta; };

template <int NT>
__global__ __launch_bounds__(512) void kgemm_kernel(KArgs a) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [8 waves][NT][16 regs][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const long m0 = (long)blockIdx.x * 32;
  const long mr = m0 + r < a.M ? m0 + r : a.M - 1;             // rows past the end read the last row, are never stored
  const bf16* xrow = a.x + mr * a.K + 32 * h;
  const bf16* wrow = a.w + (long)r * a.K + 32 * h;
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x16)(0.f);
  const int nchunks = a.K >> 6;
  uint4 fx[2][4], fw[2][NT][4];
  auto load = [&](int buf, int c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) fx[buf][j] = *reinterpret_cast<const uint4*>(xrow + 64 * c + 8 * j);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) fw[buf][t][j] = *reinterpret_cast<const uint4*>(wrow + (long)32 * t * a.K + 64 * c + 8 * j);
  };
  auto mma = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[buf][t][j]), __builtin_bit_cast(bf16x8, fx[buf][j]), acc[t], 0, 0, 0);
  };
  int c = wave;
  if (c < nchunks) load(0, c);
  for (; c < nchunks; c += 16) {
    if (c + 8 < nchunks) load(1, c + 8);
    mma(0);
    if (c + 8 < nchunks) {
      if (c + 16 < nchunks) load(0, c + 16);
      mma(1);
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) red[((wave * NT + t) * 16 + reg) * 64 + lane] = acc[t][reg];
  __syncthreads();
  for (int e = tid; e < NT * 1024; e += 512) {
    const int l = e & 63, reg = (e >> 6) & 15, t = e >> 10;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) v += red[((w * NT + t) * 16 + reg) * 64 + l];
    const long m = m0 + (l & 31);
    const int o = 32 * t + acc_row(reg, l);
    if (m < a.M) {
      v *= a.alpha;
      if (a.res) v += a.beta * (float)a.res[m * a.O + o];
      a.y[m * a.O + o] = (bf16)v;
    }
  }
}

}  // namespace

// Returns HDMOE_OK after launching, a negative status on a launch error, or 1 when the layer is outside this file's domain.
int kgemm_try_launch(const ConvArgs& a, int dtype, hipStream_t stream) {
  static const bool off = getenv("HDMOE_KGEMM") && atoi(getenv("HDMOE_KGEMM")) == 0;
  if (off || dtype != HDMOE_BF16 || a.ngroups != 1 || a.seg || a.stride != 1 || a.ones || a.kh[0] != 1 || a.kw[0] != 1 || a.pt[0] || a.pl[0]) return 1;
  static const int mink = getenv("HDMOE_KGEMM_MINK") ? atoi(getenv("HDMOE_KGEMM_MINK")) : 512;   // (768: the text projections of the fusion cross-attention, 37 + 31 -> ~2 x 12 us on the serial stage; same-box step -0.1 ms)
  if (a.Cin != a.Cphys || a.Ipad != a.Cin || a.Cin % 64 || a.Cin < mink || a.Cout != a.Cstore || a.Cout % 32 || a.Cout > 64) return 1;
  if (a.Ho != a.H || a.Wo != a.W || (((uintptr_t)a.x | (uintptr_t)a.w) & 15)) return 1;
  KArgs k;
  k.x = (const bf16*)a.x; k.w = (const bf16*)a.w; k.y = (bf16*)a.y; k.res = (const bf16*)a.res;
  k.M = (long)a.N * a.H * a.W; k.K = a.Cin; k.O = a.Cout; k.alpha = a.alpha; k.beta = a.beta;
  const long blocks = (k.M + 31) / 32;
  if (blocks > 0x7fffffffl) return 1;
  const int NT = a.Cout / 32;
  const size_t lds = (size_t)8 * NT * 4096;
  if (NT == 1) hipLaunchKernelGGL(kgemm_kernel<1>, dim3((unsigned)blocks), dim3(512), lds, stream, k);
  else hipLaunchKernelGGL(kgemm_kernel<2>, dim3((unsigned)blocks), dim3(512), lds, stream, k);
  return hdmoe_launch_status();
}
