// conv6 for fp32 tensors on the bf16 matrix pipe ("split-bf16"): the fp32 router trunks (Router.hard_route, reference
// models/model_components.py:100-112) feed torch.topk, whose indices have to match the fp32 reference, so they cannot run in plain
// bf16 -- and fp32-input MFMA runs at 1/16 of the bf16 rate (MI355X_MICROARCH.md, Matrix cores).  Here every fp32 operand is split
// into two bf16 numbers, v = hi + lo (hi = bf16(v), lo = bf16(v - hi): 16 significant bits), and a product becomes three MFMAs
//     w * x  ~=  w_hi * x_hi  +  w_lo * x_hi  +  w_hi * x_lo          (the dropped w_lo * x_lo term is 2^-16 of the product)
// accumulated in fp32: 3/16 of the fp32-MFMA time at ~1e-5 relative accuracy, inside every tolerance of the fp32 parity tests
// (router top-k indices included: the logit margins are >= 5e-2).
//
// Same persistent structure as conv6.hip (one workgroup per CU, static unit list, weights by LDS-DMA, one barrier per stage) with
//   * weights pre-split by the weight prep into two bf16 images [hi | lo][g][tap][Cout][Cin];
//   * the fp32 halo tile of a 32-channel chunk loaded to REGISTERS (buffer loads, zero padding by the hardware bounds check), split
//     there -- after the optional fused GroupNorm + ReLU of the producing layer, relu(x * scale[n][c] + shift[n][c]) -- and written
//     to LDS as a hi and a lo bf16 image in conv6's swizzled [pixel][32 ch] layout; the two images serve three stage groups
//     (hi x W_hi, hi x W_lo, lo x W_hi), so the halo is single-buffered: the next chunk's registers are loaded beside the chunk's
//     last stage and converted between two barriers;
//   * fp32 epilogue (16-byte stores straight from the accumulator quads), optionally accumulating per-sample sum / sum of squares
//     for the GroupNorm that follows.
// Domain: 3x3 kernels, stride 1, Cin % 32 == 0, Cout % 32 == 0, W == 16 or W % 32 == 0 (the trunk and its dgrad); else conv.hip.
#include <stdlib.h>
#include "conv_args.h"
#include "conv6_common.h"
#include "hdmoe.h"
#include "conv6s_body.h"

namespace {

template <int NT, bool P3>
__global__ __launch_bounds__(64 * C6_NW) void conv6_split_kernel(C6SArgs sa) {
  conv6s_body<NT, P3>(sa, blockIdx.x, gridDim.x);
}

}  // namespace

// x, y (and res) fp32; w = bf16 [hi | lo][g][tap][Cout][Cin] with `wplane_elems` elements between the two planes
int conv6s_plan(const ConvArgs& c, long wplane_elems, const ConvFuse* fuse, C6SPlan& plan, bool p3) {
  static const bool off = getenv("HDMOE_CONV6") && atoi(getenv("HDMOE_CONV6")) == 0;
  if (off) return 1;
  if (c.stride != 1 || c.ones || c.Cphys != c.Cin || c.Ipad != c.Cin || c.Cin % 32 || c.Cout % 32 || c.Cstore != c.Cout) return 1;
  if (c.Ho != c.H || c.Wo != c.W || !(c.W == 16 || c.W % 32 == 0) || c.H < 8) return 1;
  for (int g = 0; g < c.ngroups; ++g) if (c.kh[g] != 3 || c.kw[g] != 3) return 1;
  if (((uintptr_t)c.x | (uintptr_t)c.w | (uintptr_t)c.y | (uintptr_t)c.res) & 15) return 1;
  const long xbytes = (long)c.N * c.H * c.W * c.Cin * 4;
  const long wbytes = (wplane_elems + (long)(c.ngroups - 1) * c.wstride + 9l * c.Cout * c.Cin) * 2;
  if (xbytes >= (1l << 31) || wbytes >= (1l << 31) || (long)c.N * c.H * c.W * c.Cout >= (1l << 31)) return 1;
  C6SArgs& sa = plan.sa;
  C6Args& a = sa.c;
  a.x = c.x; a.w = c.w; a.y = c.y; a.res = c.res; a.seg = c.seg; a.wstride = c.wstride;
  a.N = c.N; a.H = c.H; a.W = c.W; a.Cin = c.Cin; a.Cout = c.Cout; a.ngroups = c.ngroups; a.alpha = c.alpha; a.beta = c.beta;
  static const int dbg = getenv("HDMOE_C6S_DBG") ? atoi(getenv("HDMOE_C6S_DBG")) : 0;   // development ablations (conv6s_body.h): 1 no MFMA, 2 no in-loop DMA, 4 no stores, 8 no halo conversion
  a.xbytes = (int)xbytes; a.wbytes = (int)wbytes; a.dbg = dbg; a.stamps = nullptr;
  for (int g = 0; g < HDMOE_MAX_GROUPS; ++g) { a.ks[g] = c.kh[g]; a.pt[g] = c.pt[g]; a.pl[g] = c.pl[g]; a.order[g] = g; }
  a.TW = c.W >= 32 ? 32 : 16; a.tws = a.TW == 32 ? 5 : 4; a.TH = 256 / a.TW;
  a.tiles_x = c.W / a.TW;
  a.tpi = a.tiles_x * (int)cdiv(c.H, a.TH);
  const int NT = c.Cout % 64 == 0 ? 2 : 1, NB = 32 * NT;
  a.nblk = c.Cout / NB;
  const int ppt = ((a.TH + 2) * (a.TW + 2) + 15) / 16;
  if (ppt > 22) return 1;
  a.hb_bytes = 2 * ppt * 1024;                           // one plane (hi or lo) of the two-tile halo image
  const int LDS_CAP = 160 * 1024;
  int T = 0;
  const int tab_bytes = (fuse && fuse->in_scale) ? 16 * c.Cin : 0;   // scale / shift table of the fused input transform (conv6s_body.h)
  if (tab_bytes && c.Cin > 512) return 1;
  // p3 (the stand-alone forward kernel): a weight stage holds the hi and the lo image of its taps (conv6s_body.h P3)
  const int planes = p3 ? 2 : 1;
  for (int t = 9; t >= 3; --t)
    if (t * (NB / 16) <= 40 && 2 * a.hb_bytes + 2 * planes * t * NB * 64 + tab_bytes <= LDS_CAP && (t == 9 || t == 5 || t == 3)) { T = t; break; }
  if (!T) return 1;
  a.T = T; a.wb_bytes = planes * T * NB * 64;
  auto recip = [](int d) { return (unsigned)((1ull << 32) / (unsigned)d + 1); };
  a.m_nblk = a.nblk == 1 ? 0xFFFFFFFFu : recip(a.nblk); a.m_T = recip(T); a.m_tpi = a.tpi == 1 ? 0xFFFFFFFFu : recip(a.tpi);
  a.m_tx = a.tiles_x == 1 ? 0xFFFFFFFFu : recip(a.tiles_x);
  sa.wplane = (int)(wplane_elems * 2);
  sa.in_scale = fuse ? fuse->in_scale : nullptr; sa.in_shift = fuse ? fuse->in_shift : nullptr; sa.in_relu = fuse ? fuse->in_relu : 0;
  sa.stats = fuse ? fuse->stats : nullptr;
  sa.nprod = 3;
  {
    static const bool drain = getenv("HDMOE_C6S_COUNTED") && atoi(getenv("HDMOE_C6S_COUNTED")) == 0;   // 0: plain stores, full drain (A/B)
    const long yb = (long)c.N * c.H * c.W * c.Cout * 4;
    a.ybytes = (yb < (1l << 32) && !drain) ? (unsigned)yb : 0u;
  }
  const size_t lds = 2 * (size_t)a.hb_bytes + 2 * (size_t)a.wb_bytes + tab_bytes;
  const long tiles = (long)c.N * a.tpi;
  long ub = ((tiles + 1) / 2 + c.ngroups) * a.nblk;
  // 224, not 256, persistent workgroups: one of these takes a CU's whole register file, so on a full grid NOTHING of the other branches
  // can start until it ends; leaving 32 CUs free lets their small kernels through (16.17 -> 16.00 ms/step, three same-box pairs;
  // 192 / 208: 16.04 / 16.08, 128: 16.27)
  static const long gcap = getenv("HDMOE_C6S_G") ? atol(getenv("HDMOE_C6S_G")) : 224;
  plan.G = (unsigned)(ub < gcap ? ub : gcap);
  plan.NT = NT; plan.lds = lds;
  return 0;
}

int conv6_split_try_launch(const ConvArgs& c, long wplane_elems, const ConvFuse* fuse, hipStream_t stream) {
  static const bool p3 = !(getenv("HDMOE_C6S_P3") && atoi(getenv("HDMOE_C6S_P3")) == 0);   // 0: the three products as separate passes (A/B)
  C6SPlan plan;
  if (conv6s_plan(c, wplane_elems, fuse, plan, p3)) return 1;
  static unsigned long long attr_set = 0;
  if (hdmoe_first_on_device(attr_set)) {
    (void)hipFuncSetAttribute((const void*)conv6_split_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv6_split_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv6_split_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv6_split_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  if (p3) {
    if (plan.NT == 2) hipLaunchKernelGGL((conv6_split_kernel<2, true>), dim3(plan.G), dim3(64 * C6_NW), plan.lds, stream, plan.sa);
    else hipLaunchKernelGGL((conv6_split_kernel<1, true>), dim3(plan.G), dim3(64 * C6_NW), plan.lds, stream, plan.sa);
  } else {
    if (plan.NT == 2) hipLaunchKernelGGL((conv6_split_kernel<2, false>), dim3(plan.G), dim3(64 * C6_NW), plan.lds, stream, plan.sa);
    else hipLaunchKernelGGL((conv6_split_kernel<1, false>), dim3(plan.G), dim3(64 * C6_NW), plan.lds, stream, plan.sa);
  }
  return hdmoe_launch_status();
}
