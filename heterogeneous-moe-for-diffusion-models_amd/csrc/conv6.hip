// K3 (second generation): persistent, LDS-DMA pipelined implicit-GEMM convolution for the k x k (k = 3, 5, 7) layers of the
// U-Net experts and the router trunks -- forward and dgrad of MP_Conv (reference models/model_internals.py:253-275), grouped by
// expert with heterogeneous kernel sizes in one launch (reference models/model_config1.py:25-37 runs them one expert at a time).
//
// Why a second kernel: conv_fwd5 (conv.hip) gives every 256-pixel tile its own workgroup, which stages operands through registers
// behind two barriers per stage; a workgroup spent ~14 % of its life issuing MFMAs.  Here
//   * ONE workgroup per CU (8 waves, 2 per SIMD) walks a static list of work units (256 or 512 output pixels x 32/64 channels),
//     dealt round-robin in descending cost order (5x5 / 7x7 experts first), so the operand pipeline never drains between tiles;
//   * operands go global -> LDS by DMA (global_load_lds_dwordx4, no staging registers): the input halo tile of one 32-channel
//     chunk (double-buffered, the next chunk's / next unit's tile is fetched in portions beside the current chunk's stages) and
//     the weights of one stage = T taps x 32 channels (double-buffered ring); one barrier per stage;
//   * LDS images are lane-linear 1-KB DMA pieces (16 rows x 64 B); bank conflicts are avoided by an XOR swizzle applied on the
//     SOURCE address of the DMA and on the ds_read_b128 address (16-byte slot ^= (row >> 2) & 3, cdna_hip_programming.md T2 / rule 21);
//   * MFMA 32x32x16 bf16 with A = weights, B = pixels: an accumulator lane owns one pixel and 4 consecutive channels per register
//     quad; the epilogue pairs quads across the two half-waves with v_permlane32_swap (T21) and stores 16 bytes per lane.
// Domain (everything else stays on conv.hip): stride 1, square k in {3,5,7}, Cin % 32 == 0, Cout % 32 == 0, W == 16 or W % 32 == 0.
#include <stdlib.h>
#include "conv_args.h"
#include "conv6_common.h"
#include "hdmoe.h"
#include "conv6_body.h"

namespace {

template <int MT, int NT, bool FILM = false>
__global__ __launch_bounds__(64 * C6_NW) void conv6_bf16_kernel(C6Args a) {
  conv6_body<MT, NT, FILM>(a, blockIdx.x, gridDim.x);
}

}  // namespace

static void* g_c6_stamps = nullptr;
// development hook (tools/conv6_check.py --stamps): 8 x 64 u64 device buffer receiving workgroup 0's in-kernel time stamps
extern "C" int hdmoe_conv6_debug_stamps(void* buf) { g_c6_stamps = buf; return HDMOE_OK; }
void* hdmoe_debug_stamp_buffer() { return g_c6_stamps; }      // (shared with conv7.hip)

// Launch geometry of conv6 for one layer (shared with the fused backward launch of bwd6.hip).  0 = planned, 1 = outside the domain.
int conv6_plan(const ConvArgs& c, int dtype, C6Plan& plan) {
  static const bool off = getenv("HDMOE_CONV6") && atoi(getenv("HDMOE_CONV6")) == 0;
  if (off) return 1;
  if (dtype != HDMOE_BF16 || c.stride != 1 || c.ones || c.Cphys != c.Cin || c.Ipad != c.Cin || c.Cin % 32 || c.Cout % 32 || c.Cstore != c.Cout) return 1;
  if (c.Ho != c.H || c.Wo != c.W || !(c.W == 16 || c.W % 32 == 0) || c.H < 8) return 1;
  int maxk = 0, mink = 99;
  for (int g = 0; g < c.ngroups; ++g) {
    if (c.kh[g] != c.kw[g] || (c.kh[g] != 3 && c.kh[g] != 5 && c.kh[g] != 7)) return 1;
    if (c.kh[g] > maxk) maxk = c.kh[g];
    if (c.kh[g] < mink) mink = c.kh[g];
  }
  if (((uintptr_t)c.x | (uintptr_t)c.w | (uintptr_t)c.y | (uintptr_t)c.res) & 15) return 1;
  long maxtaps = 0;
  for (int g = 0; g < c.ngroups; ++g) if ((long)c.kh[g] * c.kw[g] > maxtaps) maxtaps = (long)c.kh[g] * c.kw[g];
  const long xbytes = (long)c.N * c.H * c.W * c.Cin * 2;
  const long wbytes = ((long)(c.ngroups - 1) * c.wstride + maxtaps * c.Cout * c.Cin) * 2;
  if (xbytes >= (1l << 31) || wbytes >= (1l << 31) || (long)c.N * c.H * c.W * c.Cout >= (1l << 31)) return 1;
  C6Args& a = plan.a;
  a.x = c.x; a.w = c.w; a.y = c.y; a.res = c.res; a.seg = c.seg; a.wstride = c.wstride;
  a.N = c.N; a.H = c.H; a.W = c.W; a.Cin = c.Cin; a.Cout = c.Cout; a.ngroups = c.ngroups; a.alpha = c.alpha; a.beta = c.beta;
  a.xbytes = (int)xbytes; a.wbytes = (int)wbytes;
  static const int dbg = getenv("HDMOE_C6_DBG") ? atoi(getenv("HDMOE_C6_DBG")) : 0;
  a.dbg = dbg;
  a.stamps = (unsigned long long*)g_c6_stamps;
  a.w_rowpitch = c.Cin; a.w_tapstride = c.Cout * c.Cin; a.gbias = nullptr;
  a.film_e = nullptr; a.film_h = nullptr; a.film_seed_dev = nullptr; a.film_seed_lo = 0; a.film_seed_hi = 0; a.film_p = 0.f;
  for (int g = 0; g < HDMOE_MAX_GROUPS; ++g) { a.ks[g] = c.kh[g]; a.pt[g] = c.pt[g]; a.pl[g] = c.pl[g]; a.order[g] = g; }
  for (int i = 1; i < c.ngroups; ++i)                       // groups by descending kernel size (longest units first)
    for (int k = i; k > 0 && a.ks[a.order[k]] > a.ks[a.order[k - 1]]; --k) { const int t = a.order[k]; a.order[k] = a.order[k - 1]; a.order[k - 1] = t; }
  a.TW = c.W >= 32 ? 32 : 16; a.tws = a.TW == 32 ? 5 : 4; a.TH = 256 / a.TW;
  a.tiles_x = c.W / a.TW;
  a.tpi = a.tiles_x * (int)cdiv(c.H, a.TH);
  const int NT = c.Cout % 64 == 0 ? 2 : 1, NB = 32 * NT;
  a.nblk = c.Cout / NB;
  const int ppt = ((a.TH + maxk - 1) * (a.TW + maxk - 1) + 15) / 16;
  if (ppt > 34) return 1;
  const long tiles = (long)c.N * a.tpi;
  static const int force_mt = getenv("HDMOE_C6_MT") ? atoi(getenv("HDMOE_C6_MT")) : 0;
  static const int force_t = getenv("HDMOE_C6_T") ? atoi(getenv("HDMOE_C6_T")) : 0;
  static const int wg2_env = getenv("HDMOE_C6_WG2") ? atoi(getenv("HDMOE_C6_WG2")) : 0;
  const bool wg2 = wg2_env && NT == 1;                       // two co-resident workgroups per CU (<1,1> needs 105 VGPRs: 4 waves / SIMD fit)
  const int LDS_CAP = wg2 ? 80 * 1024 : 160 * 1024;
  int MT = (tiles * a.nblk >= 4 * 256) ? 2 : 1;             // 512-pixel units only when every CU still gets >= 2 of them
  if (wg2) MT = 1;
  if (force_mt) MT = force_mt;
  int T = 0;
  for (; MT >= 1; --MT) {
    a.hb_bytes = C6_NW * ((MT * ppt + C6_NW - 1) / C6_NW) * 1024;   // every wave owns the same number of 1-KB pieces
    // largest stage (fewest barriers) that fits next to the two halo buffers; at most 40 DMA pieces per stage
    int best = 0, best_stages = 1 << 30;
    for (int t = 9; t >= 2; --t) {
      if (t * (NB / 16) > 40 || 2 * a.hb_bytes + 2 * t * NB * 64 > LDS_CAP) continue;
      int stages = 0;
      for (int g = 0; g < c.ngroups; ++g) stages += (c.kh[g] * c.kh[g] + t - 1) / t;
      if (stages <= best_stages) { best_stages = stages; best = t; }       // ties: the smaller stage (more even split of the taps)
    }
    if (best) { T = best; break; }
  }
  if (!T) return 1;
  if (force_t) T = force_t;
  a.T = T; a.wb_bytes = T * NB * 64;
  auto recip = [](int d) { return (unsigned)((1ull << 32) / (unsigned)d + 1); };
  a.m_nblk = a.nblk == 1 ? 0xFFFFFFFFu : recip(a.nblk); a.m_T = recip(T); a.m_tpi = a.tpi == 1 ? 0xFFFFFFFFu : recip(a.tpi);
  a.m_tx = a.tiles_x == 1 ? 0xFFFFFFFFu : recip(a.tiles_x);
  const size_t lds = 2 * (size_t)a.hb_bytes + 2 * (size_t)a.wb_bytes;
  if (lds > (size_t)LDS_CAP || T * (NB / 16) > 40) return 1;
  long ub = (tiles + MT - 1) / MT + c.ngroups;
  ub *= a.nblk;
  static const long gcap_env = getenv("HDMOE_C6_G") ? atol(getenv("HDMOE_C6_G")) : 0;       // (A/B aid: fewer persistent workgroups leave CUs to the other branches)
  const long gcap = wg2 ? 512 : (gcap_env > 0 ? gcap_env : 256);
  plan.G = (unsigned)(ub < gcap ? ub : gcap);
  plan.MT = MT; plan.NT = NT; plan.lds = lds;
  return 0;
}

int conv6_try_launch(const ConvArgs& c, const ConvFuse* fuse, int dtype, hipStream_t stream) {
  if (fuse && (fuse->in_scale || fuse->in_shift || fuse->stats)) return 1;
  C6Plan plan;
  if (conv6_plan(c, dtype, plan)) return 1;
  if (fuse && fuse->film_e) {                                // FiLM epilogue: second output tensor
    if (!fuse->film_h || (((uintptr_t)fuse->film_h) & 15) || c.res) return 1;
    plan.a.film_e = fuse->film_e; plan.a.film_h = fuse->film_h; plan.a.film_seed_dev = fuse->film_seed_dev;
    plan.a.film_seed_lo = (unsigned)fuse->film_seed; plan.a.film_seed_hi = (unsigned)(fuse->film_seed >> 32); plan.a.film_p = fuse->film_p;
  }
  const C6Args& a = plan.a;
  const unsigned G = plan.G;
  const size_t lds = plan.lds;
  const int MT = plan.MT, NT = plan.NT;
  static unsigned long long attr_set = 0;
  if (hdmoe_first_on_device(attr_set)) {
#define C6_ATTR(M, Nt) (void)hipFuncSetAttribute((const void*)conv6_bf16_kernel<M, Nt>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
    C6_ATTR(1, 1); C6_ATTR(1, 2); C6_ATTR(2, 1); C6_ATTR(2, 2);
#define C6F_ATTR(M, Nt) (void)hipFuncSetAttribute((const void*)conv6_bf16_kernel<M, Nt, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
    C6F_ATTR(1, 1); C6F_ATTR(1, 2); C6F_ATTR(2, 1); C6F_ATTR(2, 2);
  }
  if (a.film_e) {
#define C6F_LAUNCH(M, Nt) hipLaunchKernelGGL((conv6_bf16_kernel<M, Nt, true>), dim3(G), dim3(64 * C6_NW), lds, stream, a)
    if (MT == 2) { if (NT == 2) C6F_LAUNCH(2, 2); else C6F_LAUNCH(2, 1); }
    else { if (NT == 2) C6F_LAUNCH(1, 2); else C6F_LAUNCH(1, 1); }
    return hdmoe_launch_status();
  }
#define C6_LAUNCH(M, Nt) hipLaunchKernelGGL((conv6_bf16_kernel<M, Nt>), dim3(G), dim3(64 * C6_NW), lds, stream, a)
  if (MT == 2) { if (NT == 2) C6_LAUNCH(2, 2); else C6_LAUNCH(2, 1); }
  else { if (NT == 2) C6_LAUNCH(1, 2); else C6_LAUNCH(1, 1); }
  return hdmoe_launch_status();
}
