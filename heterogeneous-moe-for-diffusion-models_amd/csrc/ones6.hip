// The first conv of every Unet_expert reads torch.cat([x, ones_like(x[:, :1])]) (reference models/model_components.py:416): 33 input
// channels, outside the I % 32 == 0 domain of the conv6 / wgrad6 kernels, so it ran on the general kernels (conv_fwd3 84 us,
// conv_wgrad2 245 us, a v5 dgrad) at 5-8 % MFMA on the U-Net branch's critical chain.  The ones channel is a constant, so its part of
// the layer is a per-expert, border-aware BIAS MAP
//     bias[g][y][x][o] = sum over the taps (ky, kx) whose input pixel (y + ky - p, x + kx - p) lies inside the image of w[g][o][32][ky][kx]
// (zero padding applies to the concatenated tensor, so the ones vanish outside the image too), and its weight gradient is a sum of dy
// over the same pixel rectangles.  The 32 real channels then run through conv6 / wgrad6:
//   forward:  bias map from the prepared weight image (one small launch), conv6 over x with the image's row pitch left at Ipad (the
//             33rd column is simply never fetched) and the map added inside the alpha scale of the epilogue;
//   backward: dx = conv6 over dy with the flipped image (its 33rd row per tap skipped through the tap stride), dW[:, :32] by wgrad6
//             into compact slabs, dW[:, 32] from per-expert pixel sums of dy; one scatter launch adds both into the bank's [tap][O][33] slab.
#include <stdlib.h>
#include "common.h"
#include "conv_args.h"
#include "conv6_common.h"
#include "hdmoe.h"
#include "conv6_body.h"

namespace {

template <int MT, int NT>
__global__ __launch_bounds__(64 * C6_NW) void conv6_ones_kernel(C6Args a) {
  conv6_body<MT, NT>(a, blockIdx.x, gridDim.x);
}

struct OnesGeo { int ks[HDMOE_MAX_GROUPS]; };

// bias[g][p][o] = sum of the valid taps' ones-channel weights (bf16 image, fp32 sum); thread = (g, pixel, o)
__global__ __launch_bounds__(256) void ones_bias_kernel(float* bias, const bf16* wf, long wstride, int H, int W, int C, int O, int Ipad, OnesGeo geo, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int o = (int)(i % O);
  long t = i / O;
  const int xx = (int)(t % W); t /= W;
  const int yy = (int)(t % H);
  const int g = (int)(t / H);
  const int k = geo.ks[g], p = (k - 1) >> 1;
  const bf16* w = wf + (long)g * wstride + (long)o * Ipad + C;
  float s = 0.f;
  for (int ky = 0; ky < k; ++ky) {
    if ((unsigned)(yy + ky - p) >= (unsigned)H) continue;
    for (int kx = 0; kx < k; ++kx)
      if ((unsigned)(xx + kx - p) < (unsigned)W) s += (float)w[(long)(ky * k + kx) * O * Ipad];
  }
  bias[i] = s;
}

// S[g][p][o] = sum over the rows of expert g of dy[n][p][o]; block = (8 pixels, g), thread = (pixel, channel): fixed order, no atomics
__global__ __launch_bounds__(256) void ones_pixel_sum_kernel(float* S, const bf16* dy, const int* seg, int N, int HW, int O) {
  const int g = blockIdx.y;
  const int ppb = 256 / O;                                      // pixels per block (O = 32 -> 8)
  const int pl = threadIdx.x / O, o = threadIdx.x - pl * O;
  const int p = blockIdx.x * ppb + pl;
  if (pl >= ppb || p >= HW) return;
  const int n0 = seg ? seg[g] : 0, n1 = seg ? seg[g + 1] : N;
  const bf16* d = dy + (long)p * O + o;
  const long rs = (long)HW * O;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int n = n0;
  for (; n + 4 <= n1; n += 4) {
    a0 += (float)d[(long)n * rs]; a1 += (float)d[(long)(n + 1) * rs]; a2 += (float)d[(long)(n + 2) * rs]; a3 += (float)d[(long)(n + 3) * rs];
  }
  for (; n < n1; ++n) a0 += (float)d[(long)n * rs];
  S[((long)g * HW + p) * O + o] = (a0 + a1) + (a2 + a3);
}

// G33[g][t][o][0 .. C) += G32[g][t][o][:],  G33[g][t][o][C] += sum over the tap's valid pixel rectangle of S[g][p][o]; block = (tap, g)
struct OnesPtrs { float* g33[HDMOE_MAX_GROUPS]; const float* g32[HDMOE_MAX_GROUPS]; };
__global__ __launch_bounds__(256) void ones_wgrad_scatter_kernel(OnesPtrs ptrs, const float* S, int H, int W, int C, int O, OnesGeo geo) {
  __shared__ float red[256];
  const int g = blockIdx.y, k = geo.ks[g], taps = k * k, t = blockIdx.x;
  if (t >= taps) return;
  const int p = (k - 1) >> 1, ky = t / k, kx = t - ky * k;
  float* G33 = ptrs.g33[g];
  const float* G32 = ptrs.g32[g];
  for (int e = threadIdx.x; e < O * C; e += 256) {
    const int o = e / C, i = e - o * C;
    G33[((long)t * O + o) * (C + 1) + i] += G32[((long)t * O + o) * C + i];
  }
  // the ones column: 256 / O pixel lanes x O channels
  const int ppb = 256 / O, pl = threadIdx.x / O, o = threadIdx.x - pl * O;
  const int y0 = max(0, p - ky), y1 = min(H, H + p - ky), x0 = max(0, p - kx), x1 = min(W, W + p - kx);
  const int rw = x1 - x0, npx = (y1 - y0) * rw;
  float s = 0.f;
  if (pl < ppb)
    for (int q = pl; q < npx; q += ppb) {
      const int yy = y0 + q / rw, xx = x0 + q % rw;
      s += S[(((long)g * H + yy) * W + xx) * O + o];
    }
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < O) {
    float tot = 0.f;
    for (int l = 0; l < ppb; ++l) tot += red[l * O + threadIdx.x];
    G33[((long)t * O + threadIdx.x) * (C + 1) + C] += tot;
  }
}

int launch_conv6_pitched(const ConvArgs& c, int rowpitch, int tapstride, long wimage_elems, const float* gbias, hipStream_t stream) {
  C6Plan plan;
  if (conv6_plan(c, HDMOE_BF16, plan)) return 1;
  C6Args& a = plan.a;
  a.w_rowpitch = rowpitch; a.w_tapstride = tapstride; a.gbias = gbias;
  const long wbytes = ((long)(c.ngroups - 1) * c.wstride + wimage_elems) * 2;
  if (wbytes >= (1l << 31)) return 1;
  a.wbytes = (int)wbytes;
  static unsigned long long attr_set = 0;
  if (hdmoe_first_on_device(attr_set)) {
#define C6O_ATTR(M, Nt) (void)hipFuncSetAttribute((const void*)conv6_ones_kernel<M, Nt>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
    C6O_ATTR(1, 1); C6O_ATTR(1, 2); C6O_ATTR(2, 1); C6O_ATTR(2, 2);
  }
#define C6O_LAUNCH(M, Nt) hipLaunchKernelGGL((conv6_ones_kernel<M, Nt>), dim3(plan.G), dim3(64 * C6_NW), plan.lds, stream, a)
  if (plan.MT == 2) { if (plan.NT == 2) C6O_LAUNCH(2, 2); else C6O_LAUNCH(2, 1); }
  else { if (plan.NT == 2) C6O_LAUNCH(1, 2); else C6O_LAUNCH(1, 1); }
  return hdmoe_launch_status();
}

bool ones_domain(int ngroups, int H, int W, int C, int O, int Ipad, const int* kh, OnesGeo& geo) {
  if (ngroups < 1 || ngroups > HDMOE_MAX_GROUPS || C % 32 || O % 32 || O > 256 || 256 % O || Ipad < C + 1 || Ipad % 8 || H < 8 || !(W == 16 || W % 32 == 0)) return false;
  for (int g = 0; g < HDMOE_MAX_GROUPS; ++g) {
    const int k = kh[g < ngroups ? g : 0];
    if (k != 3 && k != 5 && k != 7) return false;
    geo.ks[g] = k;
  }
  return true;
}

}  // namespace

extern "C" {

/* y = alpha * conv(cat([x, ones]), w) for the experts of a layer: x [N][H][W][C] bf16 (C % 32 == 0), wf = forward weight image
 * [g][tap][O][Ipad] of the (C + 1)-channel layer (hdmoe_wbank_prep / hdmoe_wprep_fwd; the ones channel is column C), gbias: workspace
 * [ngroups][H][W][O] fp32 (written here).  Returns 1 without launching outside the domain (bf16, conv6's shapes). */
int hdmoe_conv6_ones_fwd(const void* x, const void* wf, void* y, float* gbias, float alpha, const int* seg, int ngroups, long wstride,
                         int N, int H, int W, int C, int O, int Ipad, const int* kh, int dtype, hipStream_t stream) {
  static const bool off = getenv("HDMOE_ONES6") && atoi(getenv("HDMOE_ONES6")) == 0;
  if (!x || !wf || !y || !gbias || N < 0) return HDMOE_EINVAL;
  OnesGeo geo;
  if (off || dtype != HDMOE_BF16 || !ones_domain(ngroups, H, W, C, O, Ipad, kh, geo) || (((uintptr_t)gbias) & 15)) return 1;
  if (N == 0) return HDMOE_OK;
  ConvArgs c;
  c.x = x; c.w = wf; c.y = y; c.res = nullptr; c.seg = seg; c.wstride = wstride;
  c.N = N; c.H = H; c.W = W; c.Ho = H; c.Wo = W; c.Cin = C; c.Cphys = C; c.Ipad = C; c.Cout = O; c.Cstore = O;
  c.stride = 1; c.ones = 0; c.ngroups = ngroups; c.n0 = 0; c.alpha = alpha; c.beta = 0.f;
  int maxk = 0;
  for (int g = 0; g < HDMOE_MAX_GROUPS; ++g) { c.kh[g] = c.kw[g] = geo.ks[g]; c.pt[g] = c.pl[g] = (geo.ks[g] - 1) / 2; if (g < ngroups && geo.ks[g] > maxk) maxk = geo.ks[g]; }
  C6Plan probe;
  if (conv6_plan(c, HDMOE_BF16, probe)) return 1;
  const long n = (long)ngroups * H * W * O;
  hipLaunchKernelGGL(ones_bias_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, gbias, (const bf16*)wf, wstride, H, W, C, O, Ipad, geo, n);
  return launch_conv6_pitched(c, Ipad, O * Ipad, (long)maxk * maxk * O * Ipad, gbias, stream);
}

/* Backward of the same layer.  dx [N][H][W][C] = alpha * dgrad (NULL: not needed); wd = flipped dgrad image [g][tap][C + 1][Opad];
 * G33: the layer's weight-gradient slabs [tap][O][C + 1] fp32 (+=); S: workspace [ngroups][H][W][O] fp32; G32: zeroed workspace slabs
 * [tap][O][C] fp32 per expert; ws / ws_bytes: hdmoe_conv_wgrad6_ws_kib(ngroups, N, H, W, C, O, kh, kh, dtype) KiB. */
int hdmoe_conv6_ones_bwd(const void* x, const void* dy, const void* wd, void* dx, float* const* G33, float* S, float* const* G32, const int* seg,
                         int ngroups, long wdstride, int N, int H, int W, int C, int O, int Opad, const int* kh, float alpha, void* ws, long ws_bytes,
                         int dtype, hipStream_t stream) {
  static const bool off = getenv("HDMOE_ONES6") && atoi(getenv("HDMOE_ONES6")) == 0;
  if (!x || !dy || !G33 || !S || !G32 || N < 0) return HDMOE_EINVAL;
  OnesGeo geo;
  if (off || dtype != HDMOE_BF16 || !ones_domain(ngroups, H, W, C, O, C + 8, kh, geo) || Opad % 8 || Opad < O) return 1;
  int k2[HDMOE_MAX_GROUPS], pd[HDMOE_MAX_GROUPS];
  for (int g = 0; g < HDMOE_MAX_GROUPS; ++g) { k2[g] = geo.ks[g]; pd[g] = (geo.ks[g] - 1) / 2; }
  const long kib = hdmoe_conv_wgrad6_ws_kib(ngroups, N, H, W, C, O, k2, k2, dtype);
  if (kib == 0 || !ws || ws_bytes < 1024 * kib || (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)ws) & 15)) return 1;
  for (int g = 0; g < ngroups; ++g) if (!G32[g] || !G33[g] || ((uintptr_t)G32[g] & 15)) return 1;
  if (N == 0) return HDMOE_OK;
  if (dx) {
    if (!wd) return HDMOE_EINVAL;
    ConvArgs c;                                             // dgrad as a forward conv over dy: "Cout" = C input channels, "Cin" = O
    c.x = dy; c.w = wd; c.y = dx; c.res = nullptr; c.seg = seg; c.wstride = wdstride;
    c.N = N; c.H = H; c.W = W; c.Ho = H; c.Wo = W; c.Cin = O; c.Cphys = O; c.Ipad = O; c.Cout = C; c.Cstore = C;
    c.stride = 1; c.ones = 0; c.ngroups = ngroups; c.n0 = 0; c.alpha = alpha; c.beta = 0.f;
    int maxk = 0;
    for (int g = 0; g < HDMOE_MAX_GROUPS; ++g) { c.kh[g] = c.kw[g] = geo.ks[g]; c.pt[g] = c.pl[g] = pd[g]; if (g < ngroups && geo.ks[g] > maxk) maxk = geo.ks[g]; }
    const int rc = launch_conv6_pitched(c, Opad, (C + 1) * Opad, (long)maxk * maxk * (C + 1) * Opad, nullptr, stream);
    if (rc) return rc;
  }
  const int rc = hdmoe_conv_wgrad6(x, dy, G32, seg, ngroups, N, H, W, C, O, k2, k2, pd, pd, ws, ws_bytes, dtype, 0, stream);
  if (rc) return rc < 0 ? rc : HDMOE_EINVAL;                // (the dgrad is already out: a refusal here would leave the layer half done)
  const int HW = H * W, ppb = 256 / O;
  hipLaunchKernelGGL(ones_pixel_sum_kernel, dim3(cdiv(HW, ppb), ngroups), dim3(256), 0, stream, S, (const bf16*)dy, seg, N, HW, O);
  OnesPtrs ptrs;
  int maxtaps = 0;
  for (int g = 0; g < HDMOE_MAX_GROUPS; ++g) {
    ptrs.g33[g] = g < ngroups ? G33[g] : nullptr; ptrs.g32[g] = g < ngroups ? G32[g] : nullptr;
    if (g < ngroups && geo.ks[g] * geo.ks[g] > maxtaps) maxtaps = geo.ks[g] * geo.ks[g];
  }
  hipLaunchKernelGGL(ones_wgrad_scatter_kernel, dim3(maxtaps, ngroups), dim3(256), 0, stream, ptrs, S, H, W, C, O, geo);
  return hdmoe_launch_status();
}

}  // extern "C"
