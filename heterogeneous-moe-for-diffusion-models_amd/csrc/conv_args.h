// Argument block shared by the conv forward / dgrad kernels (conv.hip: v5 and the odd-shape kernels; conv6.hip: the
// persistent LDS-DMA pipelined kernel).
#pragma once
#include "common.h"

struct ConvArgs {
  const void* x;      // [N][H][W][Cphys]
  const void* w;      // [g][tap][Cout][Ipad]
  void* y;            // [N][Ho][Wo][Cstore]
  const void* res;    // optional [N][Ho][Wo][Cstore]:  y = alpha*acc + beta*res
  const int* seg;     // [ngroups+1] row offsets or null
  long wstride;
  int N, H, W, Ho, Wo, Cin, Cphys, Ipad, Cout, Cstore, stride, ones, ngroups;
  int n0;             // first row of this launch (row-per-blockIdx.y kernels: 65535 rows per launch)
  int kh[HDMOE_MAX_GROUPS], kw[HDMOE_MAX_GROUPS], pt[HDMOE_MAX_GROUPS], pl[HDMOE_MAX_GROUPS];
  float alpha, beta;
};

// Fused pro-/epilogue of the conv6 kernels (all optional):
//   in_scale/in_shift [N][Cin] fp32 + in_relu: the staged input is relu(x * scale[n][c] + shift[n][c]) -- GroupNorm(1,C) + ReLU of the
//   producing layer folded into this layer's staging pass (Router.hard_route, reference model_components.py:100-112);
//   stats [N][2] fp32 (caller zeroes): per-sample sum and sum of squares of the fp32 outputs, accumulated for the NEXT GroupNorm.
//   film_e != null (bf16 conv6 only): second output film_h = dropout_p(mp_silu(y * film_e[n][c])) (FiLM of Unet_block, conv6_common.h).
struct ConvFuse {
  const float* in_scale;
  const float* in_shift;
  float* stats;
  int in_relu;
  const float* film_e = nullptr; void* film_h = nullptr; const unsigned long long* film_seed_dev = nullptr;
  unsigned long long film_seed = 0; float film_p = 0.f;
};

// Returns HDMOE_OK after launching, a negative status on a launch error, or 1 when the shape is outside conv6's domain
// (the caller then takes the general kernels).
int conv6_try_launch(const ConvArgs& a, const ConvFuse* fuse, int dtype, hipStream_t stream);

void* hdmoe_debug_stamp_buffer();     // development: the buffer registered with hdmoe_conv6_debug_stamps (conv6.hip), or null
// Whole-image streaming kernel for 32 x 32 maps (conv7.hip).  Same return convention.
int conv7_try_launch(const ConvArgs& a, int dtype, hipStream_t stream);

// Split-bf16 variant for fp32 tensors (conv6s.hip): w = bf16 [hi | lo][g][tap][Cout][Cin], `wplane_elems` elements per plane.
int conv6_split_try_launch(const ConvArgs& a, long wplane_elems, const ConvFuse* fuse, hipStream_t stream);

// Pointwise (linear / 1x1, stride 1) weight gradient (lwgrad.hip): G[g] [Cout][Cin] fp32 slabs (+=).  Same return convention.
int lwg_try_launch(const void* x, const void* dy, float* const* G, const int* seg, int ngroups, int N, long HW, int Cin, int Cout,
                   int dtype, hipStream_t stream);

// Pointwise forward / dgrad with Cin >= 512 and Cout <= 64 (kgemm.hip).  Same return convention.
int kgemm_try_launch(const ConvArgs& a, int dtype, hipStream_t stream);

// k x k fp32 weight gradient for tiny input channel counts (taps * Cin <= 64: the stem), one expert (lwgrad.hip).  Same return convention.
int swg_try_launch(const void* x, const void* dy, float* G, int N, int H, int W, int Cin, int Cout, int k, int pt, int pl, int dtype,
                   hipStream_t stream);

// k x k (k = 1 or 3) bf16 weight gradient for tiny output channel counts (Cout <= 4: output head, gate), one expert (lwgrad.hip).  Same return convention.
int towg_try_launch(const void* x, const void* dy, float* G, int N, int H, int W, int Cin, int Cout, int k, int pt, int pl, int dtype,
                    hipStream_t stream);

// Grouped fp32 linear on one-position rows with a long input, 256 <= Cin <= 1024 (the experts' text projection; mlinear.hip).  Same return convention.
int glin_try_launch(const ConvArgs& a, int dtype, hipStream_t stream);
