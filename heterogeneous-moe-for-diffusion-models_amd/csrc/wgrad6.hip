// K3 weight gradient, second generation: persistent, LDS-DMA double-buffered, atomic-free.
//
// dW[g][tap][o][i] = sum over the pixels p of expert g's rows of dy[p][o] * x[p + tap][i]  (autograd of MP_Conv, reference
// models/model_internals.py:253-275 via F.conv2d): a GEMM whose contraction runs over PIXELS while the tiles are stored
// [pixel][channel], so both MFMA operands are fetched with the transposing LDS read ds_read_b64_tr_b16 (two per fragment).
//
// What changed against conv_wgrad2 (conv.hip), which spent ~40 % of its time in the fp32 atomic flush and staged every 128-pixel
// tile through registers behind two barriers:
//   * a workgroup (8 waves) owns one (expert, 32*OT output channels, 32 input channels) block of dW for a contiguous range of
//     256-pixel tiles and keeps ALL its taps' accumulators in registers: wave w holds taps w, w+8, .. (both channel halves) plus one
//     tile of the remaining tap, so every wave runs the same static program (3x3: 3 MFMAs per k-step, 5x5: 7);
//   * tiles (dy [256 px][32*OT] and the x halo [(TH+k-1) x (TW+k-1) px][32]) arrive by LDS-DMA into two buffers, the next tile's
//     pieces are issued one per k-step inside the current tile's loop; ONE barrier per tile;
//   * kernel size and tile width are template parameters: every LDS address of the unrolled 16-k-step loop is a per-lane constant
//     plus an instruction immediate -- no address arithmetic in the loop;
//   * results leave as plain 128-byte-run stores into a per-(expert, pixel-partition) partial slab; wgrad6_reduce_kernel sums the
//     partitions into the bank's [tap][O][I] gradient slab in a fixed order (deterministic; float atomics ran at ~1.3 TB/s).
// Domain: bf16, stride 1, square k in {3, 5}, Cin % 32 == 0, Cout % 32 == 0, W == 16 or W % 32 == 0; everything else: conv.hip.
#include <stdlib.h>
#include "common.h"
#include "conv_args.h"
#include "hdmoe.h"
#include "wgrad6_body.h"

namespace {

template <int KS, int TWS, int OT, bool SPLIT>
__global__ __launch_bounds__(512) void wgrad6_kernel(W6Args a) {
  wgrad6_body<KS, TWS, OT, SPLIT>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}
// Both kernel-size classes of a layer (3x3 experts and 5x5 experts) in ONE launch: partition slots [0, a3.chunks) run the 3x3 program,
// the rest the 5x5 program.  As two launches the 3x3 class -- a third of the 5x5 class's FLOPs but latency-bound, hence just as long
// -- ran in front of the 5x5 class; side by side it disappears under it.
template <int TWS, int OT>
__global__ __launch_bounds__(512) void wgrad6_dual_kernel(W6Args a3, W6Args a5) {
  if ((int)blockIdx.z < a3.chunks) wgrad6_body<3, TWS, OT, false>(a3, blockIdx.x, blockIdx.y, blockIdx.z);
  else wgrad6_body<5, TWS, OT, false>(a5, blockIdx.x, blockIdx.y, blockIdx.z - a3.chunks);
}

// G[g][e] += sum over the non-empty partitions c of ws[gi][c][e]   (fixed order: deterministic)
struct W6Ptrs { float* G[HDMOE_MAX_GROUPS]; };
// One 16-byte element per 8 adjacent lanes: lane part p sums the partitions c = p, p + 8, .. (8 independent streams per element keep
// enough loads in flight -- one thread per element read the ~50 MB of partials at ~1 TB/s), then a fixed-order butterfly.
__global__ __launch_bounds__(256) void wgrad6_reduce_kernel(W6Args a, W6Ptrs gp, long n4) {
  const int gi = blockIdx.y;
  int slot0 = 0, nch = 0;
  for (int k = 0; k <= gi; ++k) {
    const int g = a.groups[k];
    const int units = ((a.seg ? a.seg[g + 1] : a.N) - (a.seg ? a.seg[g] : 0)) * a.tpi;
    slot0 += nch;
    nch = (units + a.upw - 1) / a.upw;
  }
  if (nch == 0) return;
  float4* Gg = reinterpret_cast<float4*>(gp.G[gi]);
  const float4* W = reinterpret_cast<const float4*>(a.ws + (long)slot0 * a.ws_item);
  const long stride4 = a.ws_item / 4;
  const int part = threadIdx.x & 7;
  for (long e0 = (long)blockIdx.x * 32; e0 < n4; e0 += (long)gridDim.x * 32) {
    const long e = e0 + (threadIdx.x >> 3);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < n4)
      for (int c = part; c < nch; c += 8) { const float4 v = W[c * stride4 + e]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      s.x += __shfl_xor(s.x, o, 8); s.y += __shfl_xor(s.y, o, 8); s.z += __shfl_xor(s.z, o, 8); s.w += __shfl_xor(s.w, o, 8);
    }
    if (part == 0 && e < n4) {
      float4 o = Gg[e];
      o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
      Gg[e] = o;
    }
  }
}

// The same reduction for up to 16 (layer, kernel-size class) items in one launch: the weight bank defers every layer's reduction to the
// end of the backward pass (each layer keeps its own workspace region until then), which turns ~64 few-microsecond launches per step
// into four.  blockIdx.z = item, blockIdx.y = expert of the item's class.
struct W6RItem {
  const float* ws; const int* seg; float* G[HDMOE_MAX_GROUPS];
  long ws_item;
  int groups[HDMOE_MAX_GROUPS]; int ngr, N, tpi, upw;
};
struct W6RBatch { W6RItem it[16]; };
__global__ __launch_bounds__(256) void wgrad6_reduce_multi_kernel(W6RBatch b) {
  const W6RItem& a = b.it[blockIdx.z];
  const int gi = blockIdx.y;
  if (gi >= a.ngr) return;
  int slot0 = 0, nch = 0;
  for (int k = 0; k <= gi; ++k) {
    const int g = a.groups[k];
    const int units = ((a.seg ? a.seg[g + 1] : a.N) - (a.seg ? a.seg[g] : 0)) * a.tpi;
    slot0 += nch;
    nch = (units + a.upw - 1) / a.upw;
  }
  if (nch == 0) return;
  float4* Gg = reinterpret_cast<float4*>(a.G[gi]);
  const float4* W = reinterpret_cast<const float4*>(a.ws + (long)slot0 * a.ws_item);
  const long stride4 = a.ws_item / 4, n4 = stride4;
  const int part = threadIdx.x & 7;
  for (long e0 = (long)blockIdx.x * 32; e0 < n4; e0 += (long)gridDim.x * 32) {
    const long e = e0 + (threadIdx.x >> 3);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < n4)
      for (int c = part; c < nch; c += 8) { const float4 v = W[c * stride4 + e]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      s.x += __shfl_xor(s.x, o, 8); s.y += __shfl_xor(s.y, o, 8); s.z += __shfl_xor(s.z, o, 8); s.w += __shfl_xor(s.w, o, 8);
    }
    if (part == 0 && e < n4) {
      float4 o = Gg[e];
      o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
      Gg[e] = o;
    }
  }
}

// Pixel partitioning of one kernel-size class: upw tiles per workgroup, `slots` partition slots (an upper bound that holds for any
// routing: sum over the class's experts of ceil(units_g / upw) <= units_l / upw + ngr).
// Workgroups that share one partition slot.  wgrad6 / wgrad7: one per 32-channel input chunk x one per 32 OT output channels.  3x3 classes of
// bf16 layers on 32 x 32 maps run the wgrad8 program (wgrad8_body.h) in the fused backward launch: 2 x 2 chunks per workgroup where the layer
// has them, so a slot has a quarter of the workgroups and the class gets four times the slots for the same number of workgroups.
int w6_wgs_per_slot(int H, int W, int Cin, int Cout, int ks, int dtype, int* icw, int* ocw) {
  static const bool w8 = !(getenv("HDMOE_WGRAD8") && atoi(getenv("HDMOE_WGRAD8")) == 0);
  const int OT = Cout % 64 == 0 ? 2 : 1;
  int ic = 1, oc = 1, n = (Cin / 32) * (Cout / (32 * OT));
  if (w8 && dtype == HDMOE_BF16 && ks == 3 && H == 32 && W == 32) {
    ic = Cin % 64 == 0 ? 2 : 1; oc = Cout % 64 == 0 ? 2 : 1;
    n = (Cin / 32 / ic) * (Cout / 32 / oc);
  }
  if (icw) *icw = w8 ? ic : 0;                               // 0: the wgrad7 program (one chunk pair per workgroup) for this class
  if (ocw) *ocw = oc;
  return n;
}

void w6_partition(long units_l, int ngr, int ngroups, int wgs, int& upw, int& slots, bool split, int tpi, const int* kh, int ks) {
  const int ibs = wgs, obs = 1;
  // Workgroups per kernel-size class.  bf16 expert layers: 128 -- every partition writes (and the reduction re-reads) a whole
  // [tap][O][I] fp32 slab, at 256 a 64->64 layer moved 71 MB of partials for 2 x 17 MB of operands, and in the fused backward launch
  // the 256 conv workgroups fill the chip anyway (same-box A/B 256 -> 128: 16.58 -> 16.15 ms/step).  The fp32 router layers
  // measure the same (16.15-16.2 with 128, 16.25-16.4 with 256 or 512).
  static const long target_bf = getenv("HDMOE_W6_PARTS") ? atol(getenv("HDMOE_W6_PARTS")) : 128;
  static const long target_sp = getenv("HDMOE_W6_PARTS_SPLIT") ? atol(getenv("HDMOE_W6_PARTS_SPLIT")) : 128;
  long parts = (split ? target_sp : target_bf) / ((long)ibs * obs);
  // Two kernel-size classes in one launch (3x3 and 5x5 experts): the workgroups are shared out by WORK (taps x rows), not evenly -- with 128 + 128
  // the 3x3 class finished in a third of the 5x5 class's time and its CUs idled (round 4; HDMOE_W6_BALANCE=0: the even split)
  static const bool balance = !(getenv("HDMOE_W6_BALANCE") && atoi(getenv("HDMOE_W6_BALANCE")) == 0);
  // One class alone (the router-trunk layers in bf16-operand mode).  HDMOE_W6_PARTS_SINGLE=256 makes the launch itself 20 % faster on the trunk
  // shapes (B = 256: 128 -> 128 233 -> 184 us, 64 -> 128 120 -> 92, 32 -> 64 44 -> 38) but the replayed step 0.05 ms SLOWER on the same box: the
  // router's backward runs beside the U-Net bank's backward, which is the critical path, and more workgroups take CUs from it.  Default: as before.
  static const long target_one = getenv("HDMOE_W6_PARTS_SINGLE") ? atol(getenv("HDMOE_W6_PARTS_SINGLE")) : target_bf;
  if (!split && ngr == ngroups) parts = target_one / ((long)ibs * obs);
  if (balance && !split && kh && ngr < ngroups) {
    long wsum = 0;
    for (int g = 0; g < ngroups; ++g) wsum += (long)kh[g] * kh[g];
    parts = 2 * target_bf * ((long)ngr * ks * ks) / wsum / ((long)ibs * obs);
  }
  if (parts < 1) parts = 1;
  const long class_units = (units_l * ngr + ngroups - 1) / ngroups;
  long u = (class_units + parts - 1) / parts; if (u < 1) u = 1;
  // whole images per partition (tpi tiles each): the streaming weight-gradient program (wgrad7_body.h) walks a slot image by image; wgrad6 and
  // the reductions recompute the same partition from the same inputs
  if (tpi < 1) tpi = 1;
  u = (u + tpi - 1) / tpi * tpi;
  while (units_l / u + ngr > 1024) u += tpi;
  upw = (int)u; slots = (int)(units_l / u + ngr);
}

template <int KS, int TWS, int OT, bool SPLIT>
void launch_w6(const W6Args& a, int ibs, int obs, hipStream_t stream) {
  constexpr int TW = 1 << TWS, TH = 256 >> TWS, HP16 = ((TW + KS - 1) * (TH + KS - 1) + 15) / 16;
  const size_t lds = 2 * (size_t)(HP16 * 1024 + 16 * OT * 1024);     // two DMA buffers, or (SPLIT) a hi and a lo plane
  static unsigned long long attr = 0;
  if (hdmoe_first_on_device(attr)) { (void)hipFuncSetAttribute((const void*)wgrad6_kernel<KS, TWS, OT, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
  hipLaunchKernelGGL((wgrad6_kernel<KS, TWS, OT, SPLIT>), dim3(ibs, obs, a.chunks), dim3(512), lds, stream, a);
}

}  // namespace

extern "C" {

// Workspace KiB hdmoe_conv_wgrad6 needs for a launch of this shape (0: outside the kernel's domain, use hdmoe_conv_wgrad).
int hdmoe_conv_wgrad6_ws_kib(int ngroups, int N, int H, int W, int Cin, int Cout, const int* kh, const int* kw, int dtype) {
  if ((dtype != HDMOE_BF16 && dtype != HDMOE_F32S) || Cin % 32 || Cout % 32 || !(W == 16 || W % 32 == 0) || H < 8 || ngroups < 1 || ngroups > HDMOE_MAX_GROUPS) return 0;
  long maxtaps = 0;
  for (int g = 0; g < ngroups; ++g) {
    if (kh[g] != kw[g] || (kh[g] != 3 && kh[g] != 5) || (dtype == HDMOE_F32S && kh[g] != 3)) return 0;
    if ((long)kh[g] * kw[g] > maxtaps) maxtaps = (long)kh[g] * kw[g];
  }
  const int TW = W >= 32 ? 32 : 16, TH = 256 / TW, OT = Cout % 64 == 0 ? 2 : 1;
  const long units_l = (long)N * (W / TW) * cdiv(H, TH);
  long bytes = 0;
  bool done[HDMOE_MAX_GROUPS] = {false};
  for (int g = 0; g < ngroups; ++g) {
    if (done[g]) continue;
    int ngr = 0;
    for (int g2 = g; g2 < ngroups; ++g2) if (!done[g2] && kh[g2] == kh[g]) { ++ngr; done[g2] = true; }
    int upw, slots;
    w6_partition(units_l, ngr, ngroups, w6_wgs_per_slot(H, W, Cin, Cout, kh[g], dtype, nullptr, nullptr), upw, slots, dtype == HDMOE_F32S, (W / TW) * (int)cdiv(H, TH), kh, kh[g]);
    const long b = (long)slots * kh[g] * kh[g] * Cout * Cin * 4;
    if (b > bytes) bytes = b;
  }
  return bytes >= (1l << 40) ? 0 : (int)((bytes + 1023) >> 10);
}

}  // extern "C"

// Launch geometry of the deferred, dual-class (3x3 + 5x5 experts) bf16 weight gradient of one layer (shared with bwd6.hip).
// 0 = planned, 1 = not applicable (the caller takes hdmoe_conv_wgrad6 / hdmoe_conv_wgrad).
int wgrad6_plan_dual(const void* x, const void* dy, float* const* G, const int* seg, int ngroups, int N, int H, int W, int Cin, int Cout,
                     const int* kh, const int* kw, const int* pt, const int* pl, void* ws, long ws_bytes, int dtype, W6DualPlan& p) {
  static const bool off = (getenv("HDMOE_WGRAD6") && atoi(getenv("HDMOE_WGRAD6")) == 0) || (getenv("HDMOE_W6_DUAL") && atoi(getenv("HDMOE_W6_DUAL")) == 0);
  if (off || dtype != HDMOE_BF16) return 1;
  const long need1 = 1024l * hdmoe_conv_wgrad6_ws_kib(ngroups, N, H, W, Cin, Cout, kh, kw, dtype);
  if (need1 == 0 || !ws || ws_bytes < 2 * need1 || !x || !dy || !G || N == 0) return 1;
  if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)ws) & 15) return 1;
  bool has3 = false, has5 = false;
  for (int g = 0; g < ngroups; ++g) {
    if (((uintptr_t)G[g] & 15) || pt[g] != (kh[g] - 1) / 2 || pl[g] != (kw[g] - 1) / 2) return 1;
    has3 = has3 || kh[g] == 3; has5 = has5 || kh[g] == 5;
  }
  if (!has3 && !has5) return 1;                             // (a single kernel-size class is fine: the other class gets no partition slots)
  const long xbytes = (long)N * H * W * Cin * 2, dybytes = (long)N * H * W * Cout * 2;
  if (xbytes >= (1l << 31) || dybytes >= (1l << 31)) return 1;
  const int TWS = W >= 32 ? 5 : 4, TW = 1 << TWS, TH = 256 / TW;
  const int OT = Cout % 64 == 0 ? 2 : 1;
  p.ibs = Cin / 32; p.obs = Cout / (32 * OT); p.TWS = TWS; p.OT = OT;
  W6Args a;
  a.x = x; a.dy = dy; a.ws = (float*)ws; a.seg = seg; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.tiles_x = W / TW; a.tpi = a.tiles_x * (int)cdiv(H, TH);
  a.xbytes = (int)xbytes; a.dybytes = (int)dybytes;
  a.in_scale = nullptr; a.in_shift = nullptr; a.in_relu = 0; a.hi_only = 0; a.icw = 1; a.ocw = 1; a.stamps = (unsigned long long*)hdmoe_debug_stamp_buffer();
  const long units_l = (long)N * a.tpi;
  for (int k = 0; k < 2; ++k) {
    const int ks = k == 0 ? 3 : 5;
    p.c[k] = a;
    p.c[k].ngr = 0;
    for (int g = 0; g < ngroups; ++g) if (kh[g] == ks) { p.c[k].groups[p.c[k].ngr++] = g; p.c[k].pt = pt[g]; p.c[k].pl = pl[g]; }
    p.c[k].ws_item = (long)ks * ks * Cout * Cin;
    int upw, slots;
    w6_partition(units_l, p.c[k].ngr, ngroups, w6_wgs_per_slot(H, W, Cin, Cout, ks, dtype, &p.c[k].icw, &p.c[k].ocw), upw, slots, false, a.tpi, kh, ks);
    p.c[k].upw = upw; p.c[k].chunks = p.c[k].ngr ? slots : 0;
  }
  // workspace regions in the order the classes appear in the group list (what hdmoe_conv_wgrad6_reduce_batch assumes)
  const int firstk = kh[0] == 3 ? 0 : 1;
  p.c[firstk].ws = (float*)ws; p.c[1 - firstk].ws = (float*)((char*)ws + need1);
  const int HP16 = ((TW + 4) * (TH + 4) + 15) / 16;
  p.lds = 2 * (size_t)(HP16 * 1024 + 16 * OT * 1024);
  return 0;
}

// Launch geometry of the deferred split-bf16 (fp32 tensors, 3x3 only) weight gradient of one layer.  0 = planned, 1 = not applicable.
int wgrad6_plan_split(const void* x, const void* dy, float* const* G, const int* seg, int ngroups, int N, int H, int W, int Cin, int Cout,
                      const int* kh, const int* kw, const int* pt, const int* pl, void* ws, long ws_bytes, W6DualPlan& p) {
  static const bool off = getenv("HDMOE_WGRAD6") && atoi(getenv("HDMOE_WGRAD6")) == 0;
  if (off) return 1;
  const long need1 = 1024l * hdmoe_conv_wgrad6_ws_kib(ngroups, N, H, W, Cin, Cout, kh, kw, HDMOE_F32S);
  if (need1 == 0 || !ws || ws_bytes < 2 * need1 || !x || !dy || !G || N == 0) return 1;
  if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)ws) & 15) return 1;
  for (int g = 0; g < ngroups; ++g)
    if (((uintptr_t)G[g] & 15) || kh[g] != 3 || kw[g] != 3 || pt[g] != 1 || pl[g] != 1) return 1;
  const long xbytes = (long)N * H * W * Cin * 4, dybytes = (long)N * H * W * Cout * 4;
  if (xbytes >= (1l << 31) || dybytes >= (1l << 31)) return 1;
  const int TWS = W >= 32 ? 5 : 4, TW = 1 << TWS, TH = 256 / TW;
  const int OT = Cout % 64 == 0 ? 2 : 1;
  p.ibs = Cin / 32; p.obs = Cout / (32 * OT); p.TWS = TWS; p.OT = OT;
  W6Args& a = p.c[0];
  a.x = x; a.dy = dy; a.ws = (float*)ws; a.seg = seg; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.tiles_x = W / TW; a.tpi = a.tiles_x * (int)cdiv(H, TH);
  a.xbytes = (int)xbytes; a.dybytes = (int)dybytes;
  a.in_scale = nullptr; a.in_shift = nullptr; a.in_relu = 0; a.hi_only = 0; a.icw = 1; a.ocw = 1; a.stamps = (unsigned long long*)hdmoe_debug_stamp_buffer();
  a.ngr = 0;
  for (int g = 0; g < ngroups; ++g) a.groups[a.ngr++] = g;
  a.pt = 1; a.pl = 1; a.ws_item = 9l * Cout * Cin;
  int upw, slots;
  w6_partition((long)N * a.tpi, a.ngr, ngroups, p.ibs * p.obs, upw, slots, true, a.tpi, kh, 3);
  a.upw = upw; a.chunks = slots;
  p.c[1] = a; p.c[1].chunks = 0;
  const int HP16 = ((TW + 2) * (TH + 2) + 15) / 16;
  p.lds = 2 * (size_t)(HP16 * 1024 + 16 * OT * 1024);
  return 0;
}

extern "C" {

// Same contract as hdmoe_conv_wgrad (G[g] += dW of group g, [tap][Cout][Cin] fp32) with a caller-provided workspace.
// Returns 1 when the shape is outside the domain (nothing launched).
int hdmoe_conv_wgrad6(const void* x, const void* dy, float* const* G, const int* seg, int ngroups, int N, int H, int W, int Cin,
                      int Cout, const int* kh, const int* kw, const int* pt, const int* pl, void* ws, long ws_bytes, int dtype,
                      int defer, hipStream_t stream) {
  static const bool off = getenv("HDMOE_WGRAD6") && atoi(getenv("HDMOE_WGRAD6")) == 0;
  if (off) return 1;
  // defer != 0: no reduction here (hdmoe_conv_wgrad6_reduce_batch does it later); the kernel-size classes then need their own
  // workspace regions, laid out one after the other (each `need` bytes at most)
  const long need1 = 1024l * hdmoe_conv_wgrad6_ws_kib(ngroups, N, H, W, Cin, Cout, kh, kw, dtype);
  const long need = defer ? 2 * need1 : need1;
  if (need1 == 0 || !ws || ws_bytes < need || !x || !dy || !G) return 1;
  if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)ws) & 15) return 1;
  for (int g = 0; g < ngroups; ++g) if (((uintptr_t)G[g] & 15) || pt[g] != (kh[g] - 1) / 2 || pl[g] != (kw[g] - 1) / 2) return 1;
  const int esz = dtype == HDMOE_F32S ? 4 : 2;
  const long xbytes = (long)N * H * W * Cin * esz, dybytes = (long)N * H * W * Cout * esz;
  if (xbytes >= (1l << 31) || dybytes >= (1l << 31)) return 1;
  if (N == 0) return HDMOE_OK;
  const int TWS = W >= 32 ? 5 : 4, TW = 1 << TWS, TH = 256 / TW;
  const int OT = Cout % 64 == 0 ? 2 : 1;
  const int ibs = Cin / 32, obs = Cout / (32 * OT);
  W6Args a;
  a.x = x; a.dy = dy; a.ws = (float*)ws; a.seg = seg; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.tiles_x = W / TW; a.tpi = a.tiles_x * (int)cdiv(H, TH);
  a.xbytes = (int)xbytes; a.dybytes = (int)dybytes;
  a.in_scale = nullptr; a.in_shift = nullptr; a.in_relu = 0; a.hi_only = 0; a.icw = 1; a.ocw = 1; a.stamps = (unsigned long long*)hdmoe_debug_stamp_buffer();
  const long units_l = (long)N * a.tpi;
  bool done[HDMOE_MAX_GROUPS] = {false};
  int cls = 0;
  if (defer) {
    W6DualPlan dp;
    if (wgrad6_plan_dual(x, dy, G, seg, ngroups, N, H, W, Cin, Cout, kh, kw, pt, pl, ws, ws_bytes, dtype, dp) == 0) {
      const dim3 grid(dp.ibs, dp.obs, dp.c[0].chunks + dp.c[1].chunks);
#define W6_DUAL(T, O)                                                                                                              \
      do {                                                                                                                         \
        static unsigned long long attr_ = 0;                                                                                                 \
        if (hdmoe_first_on_device(attr_)) { (void)hipFuncSetAttribute((const void*)wgrad6_dual_kernel<T, O>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); } \
        hipLaunchKernelGGL((wgrad6_dual_kernel<T, O>), grid, dim3(512), dp.lds, stream, dp.c[0], dp.c[1]);                         \
      } while (0)
      if (TWS == 5) { if (OT == 2) W6_DUAL(5, 2); else W6_DUAL(5, 1); } else { if (OT == 2) W6_DUAL(4, 2); else W6_DUAL(4, 1); }
      return hdmoe_launch_status();
    }
  }
  for (int g = 0; g < ngroups; ++g) {
    if (done[g]) continue;
    a.ngr = 0;
    for (int g2 = g; g2 < ngroups; ++g2)
      if (!done[g2] && kh[g2] == kh[g]) { a.groups[a.ngr++] = g2; done[g2] = true; }
    const int ks = kh[g];
    a.pt = pt[g]; a.pl = pl[g];
    if (defer) {
      if (cls >= 2) return HDMOE_EINVAL;                       // (the domain has two kernel sizes)
      a.ws = (float*)((char*)ws + (long)cls * need1);
    }
    ++cls;
    a.ws_item = (long)ks * ks * Cout * Cin;
    int upw, slots;
    w6_partition(units_l, a.ngr, ngroups, ibs * obs, upw, slots, dtype == HDMOE_F32S, a.tpi, kh, ks);
    a.upw = upw; a.chunks = slots;                          // (chunks = partition slots of this class)
    const W6Args& b = a;                                    // (kernel-size classes reuse the workspace: each class's reduce runs before the next class)
#define W6_LAUNCH(K, T, O) launch_w6<K, T, O, false>(b, ibs, obs, stream)
    if (dtype == HDMOE_F32S) {
      if (TWS == 5) { if (OT == 2) launch_w6<3, 5, 2, true>(b, ibs, obs, stream); else launch_w6<3, 5, 1, true>(b, ibs, obs, stream); }
      else { if (OT == 2) launch_w6<3, 4, 2, true>(b, ibs, obs, stream); else launch_w6<3, 4, 1, true>(b, ibs, obs, stream); }
    } else if (ks == 3) { if (TWS == 5) { if (OT == 2) W6_LAUNCH(3, 5, 2); else W6_LAUNCH(3, 5, 1); } else { if (OT == 2) W6_LAUNCH(3, 4, 2); else W6_LAUNCH(3, 4, 1); } }
    else { if (TWS == 5) { if (OT == 2) W6_LAUNCH(5, 5, 2); else W6_LAUNCH(5, 5, 1); } else { if (OT == 2) W6_LAUNCH(5, 4, 2); else W6_LAUNCH(5, 4, 1); } }
    if (defer) continue;
    W6Ptrs gp;
    for (int k = 0; k < HDMOE_MAX_GROUPS; ++k) gp.G[k] = k < b.ngr ? G[b.groups[k]] : nullptr;
    const long n4 = b.ws_item / 4;
    const long rblocks = (n4 + 31) / 32;
    hipLaunchKernelGGL(wgrad6_reduce_kernel, dim3((unsigned)(rblocks < 2048 ? rblocks : 2048), b.ngr), dim3(256), 0, stream, b, gp, n4);
  }
  return hdmoe_launch_status();
}

// Deferred reductions of `n` hdmoe_conv_wgrad6(..., defer = 1) calls, 16 (layer, class) items per launch.  Per call i: G + 8 i (its
// per-expert slabs), seg[i], ws[i] (the workspace it was given) and dims + 16 i = {ngroups, N, H, W, Cin, Cout, dtype, 0, kh[0..7]}.
int hdmoe_conv_wgrad6_reduce_batch(float* const* G, const int* const* seg, float* const* ws, const int* dims, int n, hipStream_t stream) {
  if (n < 0 || (n && (!G || !seg || !ws || !dims))) return HDMOE_EINVAL;
  W6RBatch batch;
  int fill = 0;
  long maxn4 = 0;
  auto flush = [&]() {
    if (!fill) return;
    for (int k = fill; k < 16; ++k) { batch.it[k] = batch.it[0]; batch.it[k].ngr = 0; }
    const long rblocks = (maxn4 + 31) / 32;
    hipLaunchKernelGGL(wgrad6_reduce_multi_kernel, dim3((unsigned)(rblocks < 256 ? rblocks : 256), HDMOE_MAX_GROUPS, fill), dim3(256), 0, stream, batch);
    fill = 0; maxn4 = 0;
  };
  for (int i = 0; i < n; ++i) {
    const int* d = dims + 16 * i;
    const int ngroups = d[0], N = d[1], H = d[2], W = d[3], Cin = d[4], Cout = d[5], dtype = d[6];
    const int* kh = d + 8;
    const long need1 = 1024l * hdmoe_conv_wgrad6_ws_kib(ngroups, N, H, W, Cin, Cout, kh, kh, dtype);
    if (need1 == 0 || !ws[i]) return HDMOE_EINVAL;
    const int TW = W >= 32 ? 32 : 16, TH = 256 / TW, OT = Cout % 64 == 0 ? 2 : 1;
    const int tpi = (W / TW) * (int)cdiv(H, TH);
    const long units_l = (long)N * tpi;
    bool done[HDMOE_MAX_GROUPS] = {false};
    int cls = 0;
    for (int g = 0; g < ngroups; ++g) {
      if (done[g]) continue;
      W6RItem& it = batch.it[fill];
      it.ngr = 0;
      for (int g2 = g; g2 < ngroups; ++g2)
        if (!done[g2] && kh[g2] == kh[g]) { it.groups[it.ngr] = g2; it.G[it.ngr] = G[8 * i + g2]; ++it.ngr; done[g2] = true; }
      for (int k = it.ngr; k < HDMOE_MAX_GROUPS; ++k) { it.groups[k] = 0; it.G[k] = nullptr; }
      int upw, slots;
      w6_partition(units_l, it.ngr, ngroups, w6_wgs_per_slot(H, W, Cin, Cout, kh[g], dtype, nullptr, nullptr), upw, slots, dtype == HDMOE_F32S, tpi, kh, kh[g]);
      it.ws = (const float*)((const char*)ws[i] + (long)cls * need1);
      it.seg = seg[i]; it.N = N; it.tpi = tpi; it.upw = upw;
      it.ws_item = (long)kh[g] * kh[g] * Cout * Cin;
      if (it.ws_item / 4 > maxn4) maxn4 = it.ws_item / 4;
      ++cls;
      if (++fill == 16) flush();
    }
  }
  flush();
  return hdmoe_launch_status();
}

}  // extern "C"
