// Many small linear layers that read the SAME input in one launch: the per-block conditioning projections.
//   Unet_block.emb_layer  (reference models/model_components.py:232-236: emb = 1 + emb_layer(embedding) * gain) -- 14 blocks, and
//   MP_Attention.q_time / k_time / v_time (reference models/model_internals.py:360-372)            -- 3 per ViT block,
// each a (rows x 64) x (64 x 32..128) product per expert: 2-8 MFLOP that cost a launch (+ a dependent-launch gap on the critical
// path) for the forward, the input gradient and the weight gradient of EVERY layer.  Here all L layers of a bank go out together:
//   y_l[r][o]  = c + sum_i x[r][i] * W_l,g(r)[o][i]          (W = the weight bank's prepared fp32 image: normalised, gain folded in)
//   dx[r][i]   = sum_l sum_o dy_l[r][o] * W_l,g(r)[o][i]     (also the sum over the layers that autograd would do with L - 1 adds)
//   dW_l,g[o][i] += sum_{r in g} dy_l[r][o] * x[r][i]        (one owner thread per element: no atomics, deterministic)
// fp32 FMA code on purpose: the whole family is ~100 MFLOP per step and lives on the fp32 vector path.
#include <stdlib.h>
#include "common.h"
#include "conv_args.h"
#include "hdmoe.h"

namespace {

constexpr int ML_MAXL = 16;                // the argument block has to stay under the 4 KB kernarg limit
struct MLArgs {
  const float* x; const int* seg;
  const float* w[ML_MAXL];     // layer image: [ngroups][O][Ipad]
  float* y[ML_MAXL];           // forward: outputs; wgrad: unused
  const float* dy[ML_MAXL];
  float* G[ML_MAXL][HDMOE_MAX_GROUPS];
  int O[ML_MAXL];
  int L, R, I, Ipad, ngroups;
  float c;
};
DEVI int ml_group(const MLArgs& a, int r) {
  if (!a.seg) return 0;
  int g = -1;
  for (int k = 0; k < a.ngroups; ++k)
    if (r >= a.seg[k] && r < a.seg[k + 1]) g = k;
  return g;
}

// grid (row, layer); thread = output channel
__global__ __launch_bounds__(128) void mlin_fwd_kernel(MLArgs a) {
  extern __shared__ float sx[];
  const int r = blockIdx.x, l = blockIdx.y, O = a.O[l];
  for (int i = threadIdx.x; i < a.I; i += blockDim.x) sx[i] = a.x[(long)r * a.I + i];
  __syncthreads();
  const int g = ml_group(a, r);
  for (int o = threadIdx.x; o < O; o += blockDim.x) {
    float acc = 0.f;
    if (g >= 0) {
      const float* wr = a.w[l] + ((long)g * O + o) * a.Ipad;
      for (int i = 0; i < a.I; ++i) acc += sx[i] * wr[i];
      acc += a.c;
    }
    a.y[l][(long)r * O + o] = acc;
  }
}
// grid (row); 256 threads = 4 slices of the (layer, output) list x 64 input channels each (I <= 256: up to 4 channels per thread);
// the slices meet in LDS.  Weight loads are issued 8 at a time (one at a time the loop ran at one L2 latency per term).
__global__ __launch_bounds__(256) void mlin_dgrad_kernel(MLArgs a, float* dx) {
  extern __shared__ float sd[];                              // [sum of O over the layers] dy of this row, then [4][I] partial sums
  const int r = blockIdx.x;
  const int g = ml_group(a, r);
  const int slice = threadIdx.x >> 6, il = threadIdx.x & 63;
  int tot = 0;
  for (int l = 0; l < a.L; ++l) {
    for (int o = threadIdx.x; o < a.O[l]; o += 256) sd[tot + o] = a.dy[l][(long)r * a.O[l] + o];
    tot += a.O[l];
  }
  __syncthreads();
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (g >= 0) {
    int base = 0;
    for (int l = 0; l < a.L; ++l) {
      const int O = a.O[l];
      const float* wl = a.w[l] + (long)g * O * a.Ipad;
      for (int o0 = slice * 8; o0 < O; o0 += 32) {             // this slice's outputs of the layer, 8 at a time
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = il + 64 * k;
          if (i < a.I) {
            float wv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) wv[j] = wl[(long)(o0 + j < O ? o0 + j : O - 1) * a.Ipad + i];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[k] += (o0 + j < O ? sd[base + o0 + j] : 0.f) * wv[j];
          }
        }
      }
      base += O;
    }
  }
  float* part = sd + tot;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int i = il + 64 * k; if (i < a.I) part[slice * a.I + i] = acc[k]; }
  __syncthreads();
  for (int i = threadIdx.x; i < a.I; i += 256) dx[(long)r * a.I + i] = part[i] + part[a.I + i] + part[2 * a.I + i] + part[3 * a.I + i];
}
// grid (o-tile of 4 rows, layer, group); thread = input channel i (x4 output rows per block); walks the group's rows
__global__ __launch_bounds__(256) void mlin_wgrad_kernel(MLArgs a) {
  const int l = blockIdx.y, g = blockIdx.z, O = a.O[l];
  const int o0 = blockIdx.x * 4;
  if (o0 >= O || !a.G[l][g]) return;
  const int r0 = a.seg ? a.seg[g] : 0, r1 = a.seg ? a.seg[g + 1] : a.R;
  int ok[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) ok[k] = o0 + k < O ? o0 + k : O - 1;          // clamped: loads stay unconditional
  for (int i = threadIdx.x; i < a.I; i += blockDim.x) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int rb = r0; rb < r1; rb += 8) {                        // 8 rows per trip, all loads first
      float xv[8], dv[8][4];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = rb + j < r1 ? rb + j : r1 - 1;
        xv[j] = a.x[(long)r * a.I + i];
#pragma unroll
        for (int k = 0; k < 4; ++k) dv[j][k] = a.dy[l][(long)r * O + ok[k]];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (rb + j < r1) {
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[k] += dv[j][k] * xv[j];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) if (o0 + k < O) a.G[l][g][(long)(o0 + k) * a.I + i] += acc[k];
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// ONE grouped fp32 linear layer with a LONG input (the experts' text projection, 768 -> emb_size, reference model_components.py:261
// map_text: rows = routed samples, one position each).  hdmoe_conv_fwd sent it to the one-row-per-block kernel of conv.hip (92 us on
// the U-Net branch's chain, 62 us on the ViT branch's).  Here a wave owns four consecutive rows and a quarter of the outputs: the rows
// sit in registers (I / 64 floats per lane and row), a weight row is ONE coalesced read per 256 inputs shared by the four rows, the four
// dot products meet by cross-lane adds.
struct GLArgs { const float* x; const float* w; float* y; const float* res; const int* seg; long wstride; int R, I, Ipad, O, ngroups; float alpha, beta; };

typedef __attribute__((ext_vector_type(4))) float glf4;
// (ONE fixed fma order for both code paths below: a row's result must not depend on whether its quad straddles a segment boundary --
//  the batch-independence test compares a sample's output bit for bit across batch compositions)
DEVI float gl_dot4(float acc, const glf4 x, const glf4 w) {
  return __builtin_fmaf(x[3], w[3], __builtin_fmaf(x[2], w[2], __builtin_fmaf(x[1], w[1], __builtin_fmaf(x[0], w[0], acc))));
}

template <int NJ>                                             // NJ = ceil(I / 256) float4 pieces per lane and row
__global__ __launch_bounds__(256) void glin_f32_kernel(GLArgs a) {
  typedef glf4 f4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = blockIdx.x * 4;
  int g[4];
  f4 xv[4][NJ];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + k;
    g[k] = -1;
    if (r < a.R) {
      if (!a.seg) g[k] = 0;
      else for (int i = 0; i < a.ngroups; ++i) if (r >= a.seg[i] && r < a.seg[i + 1]) g[k] = i;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int i = 4 * lane + 256 * j;
      xv[k][j] = (g[k] >= 0 && i < a.I) ? *reinterpret_cast<const f4*>(a.x + (long)r * a.I + i) : (f4)(0.f);
    }
  }
  const bool same = g[0] == g[1] && g[1] == g[2] && g[2] == g[3];   // (rows are sorted by group: true except at a segment boundary)
  const int oq = (a.O + 3) / 4, o0 = wave * oq, o1 = min(a.O, o0 + oq);
  for (int o = o0; o < o1; ++o) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (same) {
      if (g[0] >= 0) {
        const float* wr = a.w + (long)g[0] * a.wstride + (long)o * a.Ipad;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int i = 4 * lane + 256 * j;
          const f4 wv = i < a.I ? *reinterpret_cast<const f4*>(wr + i) : (f4)(0.f);
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[k] = gl_dot4(acc[k], xv[k][j], wv);
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (g[k] < 0) continue;
        const float* wr = a.w + (long)g[k] * a.wstride + (long)o * a.Ipad;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int i = 4 * lane + 256 * j;
          const f4 wv = i < a.I ? *reinterpret_cast<const f4*>(wr + i) : (f4)(0.f);
          acc[k] = gl_dot4(acc[k], xv[k][j], wv);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = wave_sum(acc[k]);
    if (lane < 4 && g[lane] >= 0) {                           // (rows outside every group stay untouched, as in conv.hip)
      const long at = (long)(r0 + lane) * a.O + o;
      float v = a.alpha * (lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3]);
      if (a.res) v += a.beta * a.res[at];
      a.y[at] = v;
    }
  }
}

static bool ml_fill(MLArgs& a, const void* x, const int* seg, const float* const* w, const int* O, int L, int R, int I, int Ipad, int ngroups) {
  if (!x || !w || !O || L < 1 || L > ML_MAXL || R < 0 || I < 1 || I > 512 || Ipad < I || ngroups < 1 || ngroups > HDMOE_MAX_GROUPS) return false;
  a.x = (const float*)x; a.seg = seg; a.L = L; a.R = R; a.I = I; a.Ipad = Ipad; a.ngroups = ngroups; a.c = 0.f;
  for (int l = 0; l < ML_MAXL; ++l) {
    a.w[l] = l < L ? w[l] : nullptr; a.O[l] = l < L ? O[l] : 0; a.y[l] = nullptr; a.dy[l] = nullptr;
    for (int g = 0; g < HDMOE_MAX_GROUPS; ++g) a.G[l][g] = nullptr;
    if (l < L && (!w[l] || O[l] < 1)) return false;
  }
  return true;
}

}  // namespace

// Grouped fp32 linear with 256 <= I <= 1024 on one-position rows (H = W = 1): see glin_f32_kernel.  Same return convention as the other
// *_try_launch helpers (1 = outside the domain).
int glin_try_launch(const ConvArgs& c, int dtype, hipStream_t stream) {
  static const bool off = getenv("HDMOE_GLIN") && atoi(getenv("HDMOE_GLIN")) == 0;
  if (off || dtype != HDMOE_F32 || c.stride != 1 || c.ones || c.H != 1 || c.W != 1 || c.Ho != 1 || c.Wo != 1 || c.Cin != c.Cphys) return 1;
  if (c.Cin < 256 || c.Cin > 1024 || c.Cin % 4 || c.Ipad % 4 || c.Cout != c.Cstore || c.N < 1) return 1;
  for (int g = 0; g < c.ngroups; ++g) if (c.kh[g] != 1 || c.kw[g] != 1 || c.pt[g] || c.pl[g]) return 1;
  if (((uintptr_t)c.x | (uintptr_t)c.w) & 15) return 1;
  GLArgs a;
  a.x = (const float*)c.x; a.w = (const float*)c.w; a.y = (float*)c.y; a.res = (const float*)c.res; a.seg = c.seg; a.wstride = c.wstride;
  a.R = c.N; a.I = c.Cin; a.Ipad = c.Ipad; a.O = c.Cout; a.ngroups = c.ngroups; a.alpha = c.alpha; a.beta = c.beta;
  const dim3 grid((unsigned)((c.N + 3) / 4));
  const int nj = (c.Cin + 255) / 256;
  if (nj == 1) hipLaunchKernelGGL(glin_f32_kernel<1>, grid, dim3(256), 0, stream, a);
  else if (nj == 2) hipLaunchKernelGGL(glin_f32_kernel<2>, grid, dim3(256), 0, stream, a);
  else if (nj == 3) hipLaunchKernelGGL(glin_f32_kernel<3>, grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(glin_f32_kernel<4>, grid, dim3(256), 0, stream, a);
  return hdmoe_launch_status();
}

extern "C" {

/* y[l] [R][O[l]] = c + x [R][I] . w[l][g(r)]^T   for l < L layers sharing the input; w[l] = prepared fp32 image [ngroups][O[l]][Ipad];
 * lens = host array O[0..L) ; seg: device row offsets or NULL (one group) */
int hdmoe_mlinear_fwd(float* const* y, const float* x, const float* const* w, const int* seg, const int* lens, int L, int R, int I,
                      int Ipad, int ngroups, float c, hipStream_t stream) {
  MLArgs a;
  if (!y || !ml_fill(a, x, seg, w, lens, L, R, I, Ipad, ngroups)) return HDMOE_EINVAL;
  for (int l = 0; l < L; ++l) { if (!y[l]) return HDMOE_EINVAL; a.y[l] = y[l]; }
  a.c = c;
  if (R == 0) return HDMOE_OK;
  hipLaunchKernelGGL(mlin_fwd_kernel, dim3(R, L), dim3(128), I * sizeof(float), stream, a);
  return hdmoe_launch_status();
}
/* dx [R][I] = sum_l dy[l] [R][O[l]] . w[l][g(r)] */
int hdmoe_mlinear_dgrad(float* dx, const float* const* dy, const float* const* w, const int* seg, const int* lens, int L, int R, int I,
                        int Ipad, int ngroups, hipStream_t stream) {
  MLArgs a;
  if (!dx || !dy || !ml_fill(a, dx, seg, w, lens, L, R, I, Ipad, ngroups)) return HDMOE_EINVAL;
  int maxO = 0;
  for (int l = 0; l < L; ++l) { if (!dy[l]) return HDMOE_EINVAL; a.dy[l] = dy[l]; if (lens[l] > maxO) maxO = lens[l]; }
  if (R == 0) return HDMOE_OK;
  int sumO = 0;
  for (int l = 0; l < L; ++l) sumO += lens[l];
  if (I > 256) return HDMOE_EINVAL;
  hipLaunchKernelGGL(mlin_dgrad_kernel, dim3(R), dim3(256), (size_t)(sumO + 4 * I) * sizeof(float), stream, a, dx);
  return hdmoe_launch_status();
}
/* G[l * 8 + g] [O[l]][I] += sum over the rows of group g of dy[l]^T . x    (G: host array of L * 8 device pointers, NULL = skip) */
int hdmoe_mlinear_wgrad(float* const* G, const float* const* dy, const float* x, const int* seg, const int* lens, int L, int R, int I,
                        int ngroups, hipStream_t stream) {
  MLArgs a;
  const float* dummy[ML_MAXL];
  for (int l = 0; l < ML_MAXL; ++l) dummy[l] = x;
  if (!G || !dy || !ml_fill(a, x, seg, dummy, lens, L, R, I, I, ngroups)) return HDMOE_EINVAL;
  int maxO = 0;
  for (int l = 0; l < L; ++l) {
    if (!dy[l]) return HDMOE_EINVAL;
    a.dy[l] = dy[l]; if (lens[l] > maxO) maxO = lens[l];
    for (int g = 0; g < ngroups; ++g) a.G[l][g] = G[l * HDMOE_MAX_GROUPS + g];
  }
  if (R == 0) return HDMOE_OK;
  hipLaunchKernelGGL(mlin_wgrad_kernel, dim3((maxO + 3) / 4, L, ngroups), dim3(I < 256 ? ((I + 63) / 64 * 64) : 256), 0, stream, a);
  return hdmoe_launch_status();
}

}  // extern "C"
