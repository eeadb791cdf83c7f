// Input gradient AND weight gradient of a k x k bf16 expert layer in ONE launch ("horizontal" fusion): the first G6 workgroups run the
// conv6 program on dy with the flipped weights (dgrad), the others the wgrad6 programs (3x3 class, then 5x5 class) on (x, dy).
// The two are independent, both read dy, and on this model's layer sizes each alone is dominated by its fixed costs (launch, pipeline
// prologue, tail of a persistent grid): back to back they cost ~30 + ~40 us per layer on the backward chain of the U-Net branch,
// side by side in one grid about the longer of the two.  A fork onto a second stream inside the graph would also overlap them, but a
// hipGraph with internal branches no longer runs concurrently with the other branch's graph (measured: the ViT backward graph then
// waits for the whole U-Net backward graph).
#include <stdlib.h>
#include "common.h"
#include "conv_args.h"
#include "conv6_common.h"
#include "hdmoe.h"
#include "conv6_body.h"
#include "conv7_body.h"
#include "wgrad6_body.h"
#include "wgrad7_body.h"
#include "wgrad8_body.h"
#include "conv6s_body.h"

namespace {

template <int MT, int NT, int TWS, int OT>
__global__ __launch_bounds__(512) void bwd6_kernel(C6Args c, W6Args a3, W6Args a5, int G6, int ibs, int obs) {
  const int b = blockIdx.x;
  if (b < G6) { conv6_body<MT, NT>(c, b, G6); return; }
  int r = b - G6;
  const int bx = r % ibs; r /= ibs;
  const int by = r % obs;
  const int z = r / obs;
  if (z < a3.chunks) wgrad6_body<3, TWS, OT, false>(a3, bx, by, z);
  else wgrad6_body<5, TWS, OT, false>(a5, bx, by, z - a3.chunks);
}

template <int MT, int NT, int TWS, int OT>
void launch_bwd6(const C6Plan& cp, const W6DualPlan& wp, hipStream_t stream) {
  static unsigned long long attr = 0;
  if (hdmoe_first_on_device(attr)) { (void)hipFuncSetAttribute((const void*)bwd6_kernel<MT, NT, TWS, OT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
  const size_t lds = cp.lds > wp.lds ? cp.lds : wp.lds;
  const unsigned grid = cp.G + (unsigned)(wp.ibs * wp.obs * (wp.c[0].chunks + wp.c[1].chunks));
  hipLaunchKernelGGL((bwd6_kernel<MT, NT, TWS, OT>), dim3(grid), dim3(512), lds, stream, cp.a, wp.c[0], wp.c[1], (int)cp.G, wp.ibs, wp.obs);
}

// The same with the whole-image streaming kernel (conv7_body.h) as the dgrad program: 32 x 32 maps, enough images to fill the chip.
template <int CO, int KMASK, int TWS, int OT>
__global__ __launch_bounds__(512) void bwd7_kernel(C7Args c, W6Args a3, W6Args a5, int G7, int ibs, int obs) {
  const int b = blockIdx.x;
  if (b < G7) { conv7_body<CO, KMASK, TWS == 4>(c, b, G7); return; }
  int r = b - G7;
  if (TWS == 5 && OT == 0) {                                 // 32 x 32 maps: the streaming weight-gradient programs (output chunks of 32)
    const int icw = a3.icw > 0 ? a3.icw : 1;
    const int nbx3 = a3.Cin / (32 * icw), nby3 = a3.Cout / (32 * a3.ocw), n3 = nbx3 * nby3 * a3.chunks;
    if (r < n3) {                                            // 3x3 class: wgrad8, icw x ocw channel chunks per workgroup
      const int bx = r % nbx3; r /= nbx3;
      const int by = r % nby3, z = r / nby3, pairs = a3.icw * a3.ocw;
      if (pairs == 0) wgrad7_body<3>(a3, bx, by, z);         // (HDMOE_WGRAD8=0)
      else if (pairs == 4) wgrad8_body3<4, 8>(a3, bx, by, z);
      else if (pairs == 2) wgrad8_body3<2, 8>(a3, bx, by, z);
      else wgrad8_body3<2, 16>(a3, bx, by, z);
      return;
    }
    r -= n3;
    const int bx = r % ibs; r /= ibs;
    wgrad7_body<5>(a5, bx, r % obs, r / obs);
  } else if (OT > 0) {
    const int bx = r % ibs; r /= ibs;
    const int by = r % obs;
    const int z = r / obs;
    if (z < a3.chunks) wgrad6_body<3, TWS, OT == 0 ? 1 : OT, false>(a3, bx, by, z);
    else wgrad6_body<5, TWS, OT == 0 ? 1 : OT, false>(a5, bx, by, z - a3.chunks);
  }
}

template <int CO, int KMASK, int TWS, int OT>
void launch_bwd7(const C7Plan& cp, const W6DualPlan& wp, hipStream_t stream) {
  static unsigned long long attr = 0;
  if (hdmoe_first_on_device(attr)) { (void)hipFuncSetAttribute((const void*)bwd7_kernel<CO, KMASK, TWS, OT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
  const size_t lds = cp.lds > wp.lds ? cp.lds : wp.lds;
  const int obs = OT == 0 ? wp.c[0].Cout / 32 : wp.obs;      // OT == 0: wgrad7 / wgrad8 (output chunks of 32)
  unsigned nw = (unsigned)(wp.ibs * obs * (wp.c[0].chunks + wp.c[1].chunks));
  if (OT == 0) nw = (unsigned)((wp.c[0].Cin / (32 * (wp.c[0].icw > 0 ? wp.c[0].icw : 1))) * (wp.c[0].Cout / (32 * wp.c[0].ocw)) * wp.c[0].chunks + wp.ibs * obs * wp.c[1].chunks);
  hipLaunchKernelGGL((bwd7_kernel<CO, KMASK, TWS, OT>), dim3(cp.G + nw), dim3(512), lds, stream, cp.a, wp.c[0], wp.c[1], (int)cp.G, wp.ibs, obs);
}

// The same for a router-trunk layer (fp32 tensors on the bf16 pipe: conv6_split program + wgrad6<SPLIT> program).
template <int NT, int TWS, int OT>
__global__ __launch_bounds__(512) void bwd6s_kernel(C6SArgs c, W6Args a3, int G6, int ibs, int obs) {
  const int b = blockIdx.x;
  if (b < G6) { conv6s_body<NT>(c, b, G6); return; }
  int r = b - G6;
  const int bx = r % ibs; r /= ibs;
  wgrad6_body<3, TWS, OT, true>(a3, bx, r % obs, r / obs);
}
template <int NT, int TWS, int OT>
void launch_bwd6s(const C6SPlan& cp, const W6DualPlan& wp, hipStream_t stream) {
  static unsigned long long attr = 0;
  if (hdmoe_first_on_device(attr)) { (void)hipFuncSetAttribute((const void*)bwd6s_kernel<NT, TWS, OT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
  const size_t lds = cp.lds > wp.lds ? cp.lds : wp.lds;
  const unsigned grid = cp.G + (unsigned)(wp.ibs * wp.obs * wp.c[0].chunks);
  hipLaunchKernelGGL((bwd6s_kernel<NT, TWS, OT>), dim3(grid), dim3(512), lds, stream, cp.sa, wp.c[0], (int)cp.G, wp.ibs, wp.obs);
}

}  // namespace

extern "C" {

/* hdmoe_conv_bwd6 for fp32 tensors computed as split bf16 (3x3 layers of the router trunks): wd = [hi | lo] bf16 dgrad image with
 * `wd_plane` elements per plane. */
int hdmoe_conv_bwd6s(const void* x, const void* dy, const void* wd, void* dx, float* const* G, const int* seg, int ngroups, long wd_stride,
                     long wd_plane, int N, int H, int W, int Cin, int Cout, const int* kh, const int* kw, const int* pt, const int* pl,
                     float alpha, void* ws, long ws_bytes, const float* in_scale, const float* in_shift, int in_relu, int hi_only, hipStream_t stream) {
  static const bool off = getenv("HDMOE_BWD6") && atoi(getenv("HDMOE_BWD6")) == 0;
  if (off || !dx || !wd || ngroups < 1 || ngroups > HDMOE_MAX_GROUPS || Cout % 16) return 1;
  W6DualPlan wp;
  if (wgrad6_plan_split(x, dy, G, seg, ngroups, N, H, W, Cin, Cout, kh, kw, pt, pl, ws, ws_bytes, wp)) return 1;
  if ((in_scale == nullptr) != (in_shift == nullptr) || (in_scale && (ngroups != 1 || seg))) return HDMOE_EINVAL;
  wp.c[0].in_scale = in_scale; wp.c[0].in_shift = in_shift; wp.c[0].in_relu = in_relu;   // the weight gradient sees relu(x * scale + shift)
  wp.c[0].hi_only = hi_only ? 1 : 0;
  ConvArgs c;
  c.x = dy; c.w = wd; c.y = dx; c.res = nullptr; c.seg = seg; c.wstride = wd_stride;
  c.N = N; c.H = H; c.W = W; c.Ho = H; c.Wo = W; c.Cin = Cout; c.Cphys = Cout; c.Ipad = Cout; c.Cout = Cin; c.Cstore = Cin;
  c.stride = 1; c.ones = 0; c.ngroups = ngroups; c.n0 = 0; c.alpha = alpha; c.beta = 0.f;
  for (int g = 0; g < HDMOE_MAX_GROUPS; ++g) { c.kh[g] = 3; c.kw[g] = 3; c.pt[g] = 1; c.pl[g] = 1; }
  C6SPlan cp;
  if (conv6s_plan(c, wd_plane, nullptr, cp)) return 1;
  cp.sa.nprod = hi_only ? 1 : 3;
#define BWD6S_GO(Nt)                                                                             \
  do {                                                                                           \
    if (wp.TWS == 5) { if (wp.OT == 2) launch_bwd6s<Nt, 5, 2>(cp, wp, stream); else launch_bwd6s<Nt, 5, 1>(cp, wp, stream); } \
    else { if (wp.OT == 2) launch_bwd6s<Nt, 4, 2>(cp, wp, stream); else launch_bwd6s<Nt, 4, 1>(cp, wp, stream); }            \
  } while (0)
  if (cp.NT == 2) BWD6S_GO(2); else BWD6S_GO(1);
  return hdmoe_launch_status();
}

/* dx = alpha * dgrad(dy, wd)  and  partial slabs of dW into ws (deferred reduction, as hdmoe_conv_wgrad6(..., defer = 1)) for one grouped
 * k x k bf16 layer with 3x3 and 5x5 experts (stride 1, "same" padding pt = (k - 1) / 2).  wd: flipped dgrad weight image
 * [g][tap][Cin][Cout] (hdmoe_wprep_fwd / the weight bank).  Returns 1 without launching when the layer is outside the domain. */
int hdmoe_conv_bwd6(const void* x, const void* dy, const void* wd, void* dx, float* const* G, const int* seg, int ngroups, long wd_stride,
                    int N, int H, int W, int Cin, int Cout, const int* kh, const int* kw, const int* pt, const int* pl, float alpha,
                    void* ws, long ws_bytes, int dtype, hipStream_t stream) {
  static const bool off = getenv("HDMOE_BWD6") && atoi(getenv("HDMOE_BWD6")) == 0;
  if (off || dtype != HDMOE_BF16 || !dx || !wd || ngroups < 1 || ngroups > HDMOE_MAX_GROUPS || Cout % 16) return 1;
  W6DualPlan wp;
  if (wgrad6_plan_dual(x, dy, G, seg, ngroups, N, H, W, Cin, Cout, kh, kw, pt, pl, ws, ws_bytes, dtype, wp)) return 1;
  ConvArgs c;                                              // the dgrad as a forward conv over dy
  c.x = dy; c.w = wd; c.y = dx; c.res = nullptr; c.seg = seg; c.wstride = wd_stride;
  c.N = N; c.H = H; c.W = W; c.Ho = H; c.Wo = W; c.Cin = Cout; c.Cphys = Cout; c.Ipad = Cout; c.Cout = Cin; c.Cstore = Cin;
  c.stride = 1; c.ones = 0; c.ngroups = ngroups; c.n0 = 0; c.alpha = alpha; c.beta = 0.f;
  for (int g = 0; g < HDMOE_MAX_GROUPS; ++g) {
    const int s = g < ngroups ? g : 0;
    c.kh[g] = kh[s]; c.kw[g] = kw[s]; c.pt[g] = kh[s] - 1 - pt[s]; c.pl[g] = kw[s] - 1 - pl[s];
  }
  {
    C7Plan cp7;                                            // 32 x 32 maps: the streaming kernel as the dgrad program (wgrad6 handles 3x3 / 5x5 only)
    static const bool w7 = !(getenv("HDMOE_WGRAD7") && atoi(getenv("HDMOE_WGRAD7")) == 0);
    if (!conv7_plan(c, dtype, cp7) && cp7.kmask == 3 && (cp7.w16 != 0) == (wp.TWS == 4)) {
#define BWD7_GO(Co)                                                                              \
  do {                                                                                           \
    if (wp.TWS == 5) { if (w7) launch_bwd7<Co, 3, 5, 0>(cp7, wp, stream); else if (wp.OT == 2) launch_bwd7<Co, 3, 5, 2>(cp7, wp, stream); else launch_bwd7<Co, 3, 5, 1>(cp7, wp, stream); } \
    else { if (wp.OT == 2) launch_bwd7<Co, 3, 4, 2>(cp7, wp, stream); else launch_bwd7<Co, 3, 4, 1>(cp7, wp, stream); }            \
  } while (0)
      if (cp7.CO == 2) BWD7_GO(2); else BWD7_GO(1);
      return hdmoe_launch_status();
    }
  }
  C6Plan cp;
  if (conv6_plan(c, dtype, cp)) return 1;
#define BWD6_GO(M, Nt)                                                                           \
  do {                                                                                           \
    if (wp.TWS == 5) { if (wp.OT == 2) launch_bwd6<M, Nt, 5, 2>(cp, wp, stream); else launch_bwd6<M, Nt, 5, 1>(cp, wp, stream); } \
    else { if (wp.OT == 2) launch_bwd6<M, Nt, 4, 2>(cp, wp, stream); else launch_bwd6<M, Nt, 4, 1>(cp, wp, stream); }            \
  } while (0)
  if (cp.MT == 2) { if (cp.NT == 2) BWD6_GO(2, 2); else BWD6_GO(2, 1); }
  else { if (cp.NT == 2) BWD6_GO(1, 2); else BWD6_GO(1, 1); }
  return hdmoe_launch_status();
}

}  // extern "C"
