// K3 + K4 fused: the main branch of a Unet_block as ONE persistent launch for all experts of a layer (reference
// models/model_components.py:240-253):
//     conv_res1 (k x k) -> * (1 + emb_layer(e) * gain) -> mp_silu -> F.dropout -> conv_res2 (k x k) -> mp_sum with the residual
// and, with the flipped weight images, its input-gradient chain
//     dgrad(conv_res2) -> dropout / mp_silu / FiLM backward -> dgrad(conv_res1).
//
// Why: at this model's widths (32 / 64 channels on 32x32 / 16x16 latents) a k x k layer is 10-20 GFLOP and 67 MB -- below the bf16 ridge and
// a handful of work units per CU, so the three launches of a block (conv6, film_silu, conv6) are dominated by their fixed costs and by the
// HBM round trips of the intermediate tensor: written by the first conv, read and re-written by the FiLM pass, read by the second conv.
// Here a work unit is one 256-pixel output tile (8 x 32 or 16 x 16) of one routed row: the workgroup computes the first conv on the tile
// grown by (k - 1) / 2 rows on either side (halo recompute, columns are whole image rows), applies the middle op to the accumulators,
// leaves the activation in LDS as the second conv's input image and runs the second conv from there.  Per block and pixel the forward then
// moves x once in and u (the pre-activation, for the backward), h (the activation, for conv_res2's weight gradient) and y once out -- the
// three intermediate READS and one launch boundary per conv disappear, and a unit carries 2.25-2.5x the MFMA work of a conv6 unit behind
// one pipeline prologue.  LDS per workgroup (160 KB): two x-chunk buffers, the intermediate image (1-2 planes of 32 channels), two weight
// stages; the DMA pieces, swizzle and tap loop are conv6's (conv6_body.h).
//
// Domain: bf16, W in {16, 32}, H a multiple of 256 / W, square odd k in {3, 5, 7} with "same" padding, Ca % 32 == 0, Cm in {32, 64},
// Cb % 32 == 0.  Everything else returns 1 (the caller runs the layers one by one).
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "conv_args.h"
#include "conv6_common.h"
#include "hdmoe.h"
#include "blk6_body.h"

namespace {

template <int NW, int NTM, int NTB, int MODE>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void blk6_kernel(B6Args a) {
  blk6_body<NW, NTM, NTB, MODE>(a, blockIdx.x, gridDim.x);
}

struct B6Plan { B6Args a; int NW, NTM, NTB; unsigned G; size_t lds; };
void* g_b6_stamps = nullptr;
constexpr int B6_EB = 256;                    // bytes of LDS behind the weight buffers: the unit's FiLM vector (<= 64 floats)

int blk6_plan(const void* x, const void* wa, const void* wb, void* y, const void* res, const int* seg, int ngroups, long wa_stride,
              long wb_stride, int N, int H, int W, int Ca, int Cm, int Cb, const int* ks, B6Plan& plan) {
  static const bool off = getenv("HDMOE_BLK6") && atoi(getenv("HDMOE_BLK6")) == 0;
  if (off || !(W == 16 || W == 32) || Ca % 32 || Cb % 32 || !(Cm == 32 || Cm == 64) || ngroups < 1 || ngroups > HDMOE_MAX_GROUPS) return 1;
  int maxk = 0;
  for (int g = 0; g < ngroups; ++g) {
    if (ks[g] != 3 && ks[g] != 5 && ks[g] != 7) return 1;
    if (ks[g] > maxk) maxk = ks[g];
  }
  if (((uintptr_t)x | (uintptr_t)wa | (uintptr_t)wb | (uintptr_t)y | (uintptr_t)res) & 15) return 1;
  const long taps = (long)maxk * maxk;
  const long xbytes = (long)N * H * W * Ca * 2;
  const long wabytes = ((long)(ngroups - 1) * wa_stride + taps * Cm * Ca) * 2, wbbytes = ((long)(ngroups - 1) * wb_stride + taps * Cb * Cm) * 2;
  if (xbytes >= (1l << 31) || wabytes >= (1l << 31) || wbbytes >= (1l << 31) || (long)N * H * W * (Cb > Cm ? Cb : Cm) >= (1l << 31)) return 1;
  B6Args& a = plan.a;
  a.x = x; a.wa = wa; a.wb = wb; a.y = y; a.res = res; a.seg = seg; a.wa_stride = wa_stride; a.wb_stride = wb_stride;
  a.N = N; a.H = H; a.W = W; a.Ca = Ca; a.Cm = Cm; a.Cb = Cb; a.ngroups = ngroups;
  a.xbytes = (int)xbytes; a.wabytes = (int)wabytes; a.wbbytes = (int)wbbytes;
  for (int g = 0; g < HDMOE_MAX_GROUPS; ++g) { a.ks[g] = g < ngroups ? ks[g] : ks[0]; a.order[g] = g; }
  for (int i = 1; i < ngroups; ++i)
    for (int k = i; k > 0 && a.ks[a.order[k]] > a.ks[a.order[k - 1]]; --k) { const int t = a.order[k]; a.order[k] = a.order[k - 1]; a.order[k - 1] = t; }
  const int NTM = Cm / 32, NTB = Cb % 64 == 0 ? 2 : 1;
  const int nbmax = 32 * (NTM > NTB ? NTM : NTB);
  static const int force_t = getenv("HDMOE_B6_T") ? atoi(getenv("HDMOE_B6_T")) : 0;
  static const int force_nw = getenv("HDMOE_B6_NW") ? atoi(getenv("HDMOE_B6_NW")) : 0;
  // Geometry candidates, best first: 4-wave workgroups, two per CU (<= 80 KB of LDS each, one x buffer, conv A over <= 12 blocks), on
  // the 256-pixel tile or -- 64-channel layers, whose intermediate image is twice as large -- on a 128-pixel tile; else one 8-wave
  // workgroup per CU with two x buffers.
  struct Cand { int nw, th; };
  const Cand cands[3] = {{4, 256 / W}, {8, 256 / W}, {4, 128 / W}};     // (128-pixel tiles measured slower than the 8-wave form: last resort)
  bool found = false;
  for (const Cand& cd : cands) {
    if ((force_nw == 4 || force_nw == 8) && cd.nw != force_nw) continue;
    const int th = cd.th;
    if (th < 4 || H % th || (th * W) % 32) continue;
    if (cd.nw == 4 && (th * W == 128) && NTM != 2) continue;
    const int ppt = ((th + 2 * (maxk - 1)) * (W + maxk - 1) + 15) / 16;
    const int nblkA = (th + maxk - 1) * W / 32;
    if (ppt > 48 || ((th + maxk - 1) * W) % 32) continue;
    if (cd.nw == 4 && (nblkA > (NTM == 2 ? 8 : 12) || ppt > 36)) continue;
    const int xb = cd.nw * ((ppt + cd.nw - 1) / cd.nw) * 1024;
    const int hbp = (((th + maxk - 1) * (W + maxk - 1) * 64) + 1023) / 1024 * 1024;
    const int cap = cd.nw == 4 ? 80 * 1024 : 160 * 1024;
    const int fixed = (cd.nw == 4 ? 1 : 2) * xb + NTM * hbp + B6_EB;
    int best = 0, best_stages = 1 << 30;
    for (int t = 9; t >= 2; --t) {
      if (t * (nbmax / 16) > 40 || fixed + 2 * t * nbmax * 64 > cap) continue;
      int stages = 0;
      for (int g = 0; g < ngroups; ++g) stages += (ks[g] * ks[g] + t - 1) / t;
      if (stages <= best_stages) { best_stages = stages; best = t; }
    }
    if (force_t >= 2 && force_t <= 9 && force_t * (nbmax / 16) <= 40 && fixed + 2 * force_t * nbmax * 64 <= cap) best = force_t;
    if (!best) continue;
    a.TH = th; a.tpi = H / th; a.xb_bytes = xb; a.hb_plane = hbp; a.T = best; a.wb_bytes = best * nbmax * 64; a.nxp = xb / 1024 / cd.nw;
    plan.NW = cd.nw;
    plan.lds = (size_t)fixed + 2 * (size_t)a.wb_bytes;
    found = true;
    break;
  }
  if (!found) return 1;
  auto recip = [](int d) { return d == 1 ? 0xFFFFFFFFu : (unsigned)((1ull << 32) / (unsigned)d + 1); };
  a.m_tpi = recip(a.tpi); a.m_T = recip(a.T);
  const long units = (long)N * a.tpi;
  static const int gcap_env = getenv("HDMOE_B6_G") ? atoi(getenv("HDMOE_B6_G")) : 0;
  const long gcap = gcap_env > 0 ? gcap_env : (plan.NW == 4 ? 512 : 256);
  plan.G = (unsigned)(units < gcap ? units : gcap);
  plan.NTM = NTM; plan.NTB = NTB;
  a.stamps = (unsigned long long*)g_b6_stamps;
  static const int dbg = getenv("HDMOE_B6_DBG") ? atoi(getenv("HDMOE_B6_DBG")) : 0;
  a.dbg = dbg;
  static const int desync = getenv("HDMOE_B6_DESYNC") ? atoi(getenv("HDMOE_B6_DESYNC")) : 0;
  a.desync = desync;
  return 0;
}

template <int MODE>
int blk6_launch(const B6Plan& plan, hipStream_t stream) {
  static unsigned long long attr_set = 0;
  if (hdmoe_first_on_device(attr_set)) {
#define B6_ATTR(W, M, B, L) (void)hipFuncSetAttribute((const void*)blk6_kernel<W, M, B, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, L * 1024)
    B6_ATTR(8, 1, 1, 160); B6_ATTR(8, 1, 2, 160); B6_ATTR(8, 2, 1, 160); B6_ATTR(8, 2, 2, 160);
    B6_ATTR(4, 1, 1, 80); B6_ATTR(4, 1, 2, 80); B6_ATTR(4, 2, 1, 80); B6_ATTR(4, 2, 2, 80);
  }
#define B6_LAUNCH(W, M, B) hipLaunchKernelGGL((blk6_kernel<W, M, B, MODE>), dim3(plan.G), dim3(64 * W), plan.lds, stream, plan.a)
#define B6_GO(W)                                                                                  \
  do {                                                                                            \
    if (plan.NTM == 2) { if (plan.NTB == 2) B6_LAUNCH(W, 2, 2); else B6_LAUNCH(W, 2, 1); }       \
    else { if (plan.NTB == 2) B6_LAUNCH(W, 1, 2); else B6_LAUNCH(W, 1, 1); }                     \
  } while (0)
  if (plan.NW == 4) B6_GO(4); else B6_GO(8);
  return hdmoe_launch_status();
}

}  // namespace

extern "C" {

// development hook (tools/blk6_bench.py --stamps): 8 x 64 u64 device buffer receiving workgroup 0's in-kernel time stamps
int hdmoe_blk6_debug_stamps(void* buf) { g_b6_stamps = buf; return HDMOE_OK; }

/* Forward of Unet_block's main branch for all experts of a layer (reference models/model_components.py:240-253):
 *   u = conv(x, w1)                         [N][H][W][C]   (written: the backward needs the pre-activation; NULL: not stored)
 *   h = dropout_p(mp_silu(u * e[n][c]))     [N][H][W][C]   (written: conv_res2's weight gradient reads it; NULL: not stored)
 *   y = alpha * conv(h, w2) + beta * res    [N][H][W][C]
 * x [N][H][W][Cin] bf16 is the block's (already mp_silu'd) input; w1 [g][tap][C][Cin], w2 [g][tap][C][C]: forward weight images of
 * hdmoe_wbank_prep / hdmoe_wprep_fwd (w?stride elements per expert); kh: per-expert square kernel size ("same" padding); e fp32 [N][C];
 * seed / seed_dev / p: the dropout of hdmoe_film_silu_drop_fwd (same Philox stream: bit-identical to the three separate launches).
 * Returns 1 without launching when the layer is outside the kernel's domain. */
int hdmoe_unet_block_fwd(const void* x, const void* w1, const void* w2, void* u, void* h, void* y, const void* res, const float* e,
                         unsigned long long seed, const unsigned long long* seed_dev, float p, float alpha, float beta, const int* seg,
                         int ngroups, long w1stride, long w2stride, int N, int H, int W, int Cin, int C, const int* kh, int dtype,
                         hipStream_t stream) {
  if (!x || !w1 || !w2 || !y || !e || N < 0 || p < 0.f || p >= 1.f) return HDMOE_EINVAL;       // (u, h may be NULL: inference, nothing saved for a backward)
  if (dtype != HDMOE_BF16) return 1;
  if (((uintptr_t)u | (uintptr_t)h) & 15) return 1;
  if (N == 0) return HDMOE_OK;
  B6Plan plan;
  if (blk6_plan(x, w1, w2, y, res, seg, ngroups, w1stride, w2stride, N, H, W, Cin, C, C, kh, plan)) return 1;
  B6Args& a = plan.a;
  a.alpha = alpha; a.beta = beta; a.alpha_mid = 1.f; a.mode = 0; a.e = e; a.u = u; a.hmid = h; a.de = nullptr;
  a.seed_dev = seed_dev; a.seed_lo = (unsigned)seed; a.seed_hi = (unsigned)(seed >> 32); a.p = p;
  return blk6_launch<0>(plan, stream);
}

/* Input-gradient chain of the same branch:
 *   dh = dgrad(dy, wd2)  (never written),   du = dropout / mp_silu / FiLM backward of dh   [N][H][W][C] (written: conv_res1's weight gradient),
 *   de [N][C] += sum_pixels d/de,           dx = alpha * dgrad(du, wd1)                    [N][H][W][Cin]
 * wd2 [g][tap][C][C], wd1 [g][tap][Cin][C]: flipped dgrad weight images; u: the pre-activation saved by the forward.  The two weight
 * gradients (x with du, h with dy) stay with hdmoe_conv_wgrad6. */
int hdmoe_unet_block_bwd(const void* dy, const void* wd2, const void* wd1, const void* u, void* du, void* dx, float* de, const float* e,
                         unsigned long long seed, const unsigned long long* seed_dev, float p, float alpha, float alpha_mid, const int* seg,
                         int ngroups, long wd2stride, long wd1stride, int N, int H, int W, int Cin, int C, const int* kh, int dtype,
                         hipStream_t stream) {
  if (!dy || !wd2 || !wd1 || !u || !du || !dx || !e || N < 0 || p < 0.f || p >= 1.f) return HDMOE_EINVAL;
  if (dtype != HDMOE_BF16) return 1;
  if (((uintptr_t)u | (uintptr_t)du) & 15) return 1;
  if (N == 0) return HDMOE_OK;
  B6Plan plan;
  if (blk6_plan(dy, wd2, wd1, dx, nullptr, seg, ngroups, wd2stride, wd1stride, N, H, W, C, C, Cin, kh, plan)) return 1;
  B6Args& a = plan.a;
  a.alpha = alpha; a.beta = 0.f; a.alpha_mid = alpha_mid; a.mode = 1; a.e = e; a.u = const_cast<void*>(u); a.hmid = du; a.de = de;
  a.seed_dev = seed_dev; a.seed_lo = (unsigned)seed; a.seed_hi = (unsigned)(seed >> 32); a.p = p;
  return blk6_launch<1>(plan, stream);
}

}  // extern "C"
