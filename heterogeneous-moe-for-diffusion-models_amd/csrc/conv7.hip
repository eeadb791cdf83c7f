// K3 (third generation): whole-image streaming convolution for the k x k expert layers on 32 x 32 feature maps -- forward and dgrad of
// MP_Conv (reference models/model_internals.py:253-275), all experts of a layer in one launch (models/model_config1.py:25-37).
// Design notes: conv7_body.h.  Domain: bf16, stride 1, H = W = 32 or H = W = 16, square k in {3, 5, 7} with "same" padding (k - 1) / 2,
// Cin % 32 == 0 (<= 256), Cout % 32 == 0 (<= 256), at least HDMOE_C7_MINN images (a unit is a whole image: fewer images than CUs leave CUs idle,
// conv6's 256-pixel units fill the chip better then).  Everything else stays on conv6 / conv.hip.
#include <stdlib.h>
#include "conv_args.h"
#include "hdmoe.h"
#include "conv7_body.h"

namespace {

template <int CO, int KMASK, bool W16>
__global__ __launch_bounds__(512) void conv7_kernel(C7Args a) { conv7_body<CO, KMASK, W16>(a, blockIdx.x, gridDim.x); }

}  // namespace

// 0 = planned, 1 = outside the domain
int conv7_plan(const ConvArgs& c, int dtype, C7Plan& plan) {
  static const bool off = getenv("HDMOE_CONV7") && atoi(getenv("HDMOE_CONV7")) == 0;
  static const int minn = getenv("HDMOE_C7_MINN") ? atoi(getenv("HDMOE_C7_MINN")) : 192;
  if (off) return 1;
  if (dtype != HDMOE_BF16 || c.stride != 1 || c.ones || c.Cphys != c.Cin || c.Ipad != c.Cin || c.Cin % 32 || c.Cin > 256 || c.Cstore != c.Cout) return 1;
  if (c.Cout % 32 || c.Cout > 256) return 1;                 // (more than 64 output channels: blocks of 64 / 32 walked over the same image)
  const bool w16 = c.H == 16;
  if (!((c.H == 32 && c.W == 32) || (c.H == 16 && c.W == 16)) || c.Ho != c.H || c.Wo != c.W || c.N < minn) return 1;
  int kmask = 0;
  long maxtaps = 0;
  for (int g = 0; g < c.ngroups; ++g) {
    const int k = c.kh[g];
    if (c.kw[g] != k || (k != 3 && k != 5 && k != 7) || c.pt[g] != (k - 1) / 2 || c.pl[g] != (k - 1) / 2) return 1;
    kmask |= k == 3 ? 1 : (k == 5 ? 2 : 4);
    if ((long)k * k > maxtaps) maxtaps = (long)k * k;
  }
  if (((uintptr_t)c.x | (uintptr_t)c.w | (uintptr_t)c.y | (uintptr_t)c.res) & 15) return 1;
  const long xbytes = (long)c.N * c.H * c.W * c.Cin * 2;
  const long wbytes = ((long)(c.ngroups - 1) * c.wstride + maxtaps * c.Cout * c.Cin) * 2;
  if (xbytes >= (1l << 31) || wbytes >= (1l << 31) || (long)c.N * c.H * c.W * c.Cout >= (1l << 31)) return 1;
  C7Args& a = plan.a;
  a.x = c.x; a.w = c.w; a.y = c.y; a.res = c.res; a.seg = c.seg; a.wstride = c.wstride;
  a.N = c.N; a.Cin = c.Cin; a.Cout = c.Cout; a.ngroups = c.ngroups; a.alpha = c.alpha; a.beta = c.beta;
  a.xbytes = (int)xbytes; a.wbytes = (int)wbytes;
  static const int dbg = getenv("HDMOE_C7_DBG") ? atoi(getenv("HDMOE_C7_DBG")) : 0;
  a.dbg = dbg;
  a.stamps = (unsigned long long*)hdmoe_debug_stamp_buffer();
  for (int g = 0; g < HDMOE_MAX_GROUPS; ++g) { a.ks[g] = c.kh[g]; a.order[g] = g; }
  for (int i = 1; i < c.ngroups; ++i)                       // groups by descending kernel size (heaviest images first)
    for (int k = i; k > 0 && a.ks[a.order[k]] > a.ks[a.order[k - 1]]; --k) { const int t = a.order[k]; a.order[k] = a.order[k - 1]; a.order[k - 1] = t; }
  static const long gcap_env = getenv("HDMOE_C7_G") ? atol(getenv("HDMOE_C7_G")) : 0;
  const long gcap = gcap_env > 0 ? gcap_env : 256;
  const long units = w16 ? (c.N + 1) / 2 + c.ngroups : c.N;    // (16 x 16: pairs of images of one expert; an upper bound for any routing)
  plan.G = (unsigned)(units < gcap ? units : gcap);
  plan.CO = c.Cout % 64 == 0 ? 2 : 1;
  plan.w16 = w16 ? 1 : 0;
  plan.lds = w16 ? C7Lds<true>::BYTES : C7Lds<false>::BYTES;
  plan.kmask = (kmask & 4) ? 7 : 3;                          // instantiated kernel-size sets: {3, 5} and {3, 5, 7}
  return 0;
}

template <int CO, int KMASK, bool W16>
static void conv7_launch_t(const C7Plan& p, hipStream_t stream) {
  static unsigned long long attr = 0;
  if (hdmoe_first_on_device(attr)) { (void)hipFuncSetAttribute((const void*)conv7_kernel<CO, KMASK, W16>, hipFuncAttributeMaxDynamicSharedMemorySize, C7Lds<W16>::BYTES); }
  hipLaunchKernelGGL((conv7_kernel<CO, KMASK, W16>), dim3(p.G), dim3(512), C7Lds<W16>::BYTES, stream, p.a);
}

void conv7_launch(const C7Plan& p, hipStream_t stream) {
#define C7_GO(Co, Km) do { if (p.w16) conv7_launch_t<Co, Km, true>(p, stream); else conv7_launch_t<Co, Km, false>(p, stream); } while (0)
  if (p.CO == 2) { if (p.kmask == 7) C7_GO(2, 7); else C7_GO(2, 3); }
  else { if (p.kmask == 7) C7_GO(1, 7); else C7_GO(1, 3); }
}

int conv7_try_launch(const ConvArgs& c, int dtype, hipStream_t stream) {
  C7Plan plan;
  if (conv7_plan(c, dtype, plan)) return 1;
  conv7_launch(plan, stream);
  return hdmoe_launch_status();
}
