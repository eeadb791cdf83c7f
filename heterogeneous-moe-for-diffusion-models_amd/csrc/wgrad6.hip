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
#include "hdmoe.h"

namespace {

typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

struct W6Args {
  const void* x; const void* dy; float* ws; const int* seg;
  int N, H, W, Cin, Cout;
  int groups[HDMOE_MAX_GROUPS]; int ngr;
  int pt, pl;
  int tiles_x, tpi;
  int upw, chunks;
  int xbytes, dybytes;
  long ws_item;                       // floats of one partial slab (taps * Cout * Cin)
};

// SPLIT: x and dy are fp32 (the router trunks); every operand is split into hi + lo bf16 while it is staged (through registers) and
// a product is three MFMAs, dy_hi x_hi + dy_hi x_lo + dy_lo x_hi -- see conv6s.hip.  LDS then holds a hi and a lo plane of each
// tile (single-buffered: the next tile waits in registers, loaded beside the current tile's loop).
template <int KS, int TWS, int OT, bool SPLIT>
DEVI void wgrad6_body(const W6Args& a, const int zslot) {
#if __HIP_DEVICE_COMPILE__
  constexpr int TW = 1 << TWS, TH = 256 >> TWS, HWp = TW + KS - 1, HHp = TH + KS - 1, NTAPS = KS * KS;
  constexpr int HP16 = (HWp * HHp + 15) / 16;               // x halo pieces (16 pixels x 64 B)
  constexpr int DYP = 16 * OT;                              // dy pieces (1 KB each) of a 256-pixel tile
  constexpr int DYROW = 64 * OT;                            // bytes per dy pixel row in LDS
  constexpr int NFULL = NTAPS / 8, REM = NTAPS % 8;         // full taps per wave / taps left over
  static_assert(REM * OT <= 8, "left-over tiles must fit one per wave");
  constexpr int XBUF = HP16 * 1024, DYBUF = DYP * 1024, BUF = XBUF + DYBUF;   // SPLIT: buffer 0 = hi planes, buffer 1 = lo planes
  constexpr int NXP = (HP16 + 7) / 8, NDP = DYP / 8;        // DMA pieces per wave
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5;
  const int q4 = (lane & 15) >> 2, col4 = (lane & 16) + 4 * (lane & 3);
  const int i0 = blockIdx.x * 32, o0 = blockIdx.y * 32 * OT;
  // partition slot blockIdx.z -> (expert of this class, pixel partition): experts take ceil(units / upw) consecutive slots each
  int gi = 0, chunk = zslot, row0 = 0, units = 0;
  for (; gi < a.ngr; ++gi) {
    const int g = a.groups[gi];
    row0 = a.seg ? a.seg[g] : 0;
    units = ((a.seg ? a.seg[g + 1] : a.N) - row0) * a.tpi;
    const int nch = (units + a.upw - 1) / a.upw;
    if (chunk < nch) break;
    chunk -= nch;
  }
  if (gi == a.ngr) return;                                   // slot beyond the partitions that exist for this routing
  const int u0 = chunk * a.upw, u1 = min(units, u0 + a.upw);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, a.dybytes, 0x00020000);
  // ---- SPLIT staging: 16-byte pieces = 4 fp32 channels of one pixel; thread t handles pieces t, t + 512, ..
  typedef __attribute__((ext_vector_type(4))) float f4;
  constexpr int XQ = 8, DYQ = 8 * OT;                        // pieces per pixel (32 / 32*OT channels)
  constexpr int NXS = (HP16 * 16 * XQ + 511) / 512, NDS = 256 * DYQ / 512;
  f4 sxr[SPLIT ? NXS : 1], sdr[SPLIT ? NDS : 1];
  auto split_load = [&](int n, int ty0, int tx0) {
#pragma unroll
    for (int k = 0; k < NXS; ++k) {
      const int e = tid + 512 * k;
      const int px = e / XQ, cqx = e - px * XQ;
      const int hy = px / HWp, hx = px - hy * HWp;
      const int iy = ty0 - a.pt + hy, ix = tx0 - a.pl + hx;
      const bool ok = px < HWp * HHp && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const unsigned off = ok ? (unsigned)((((n * a.H + iy) * a.W + ix) * a.Cin + i0) * 4 + cqx * 16) : 0xFFFFFFFFu;
      sxr[k] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
    }
#pragma unroll
    for (int k = 0; k < NDS; ++k) {
      const int e = tid + 512 * k;
      const int q = e / DYQ, cqd = e - q * DYQ;
      const int oy = ty0 + (q >> TWS), ox = tx0 + (q & (TW - 1));
      const bool ok = oy < a.H && ox < a.W;
      const unsigned off = ok ? (unsigned)((((n * a.H + oy) * a.W + ox) * a.Cout + o0) * 4 + cqd * 16) : 0xFFFFFFFFu;
      sdr[k] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rdy, off, 0, 0));
    }
  };
  auto split_store = [&]() {                                  // registers -> hi / lo bf16 planes (same images as the DMA path)
    auto put = [&](const f4& v, int off) {
      bf16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) { hi[e] = (bf16)v[e]; lo[e] = (bf16)(v[e] - (float)hi[e]); }
      *reinterpret_cast<bf16x4*>(lds + off) = hi;
      *reinterpret_cast<bf16x4*>(lds + BUF + off) = lo;
    };
#pragma unroll
    for (int k = 0; k < NXS; ++k) {
      const int e = tid + 512 * k;
      const int px = e / XQ, cqx = e - px * XQ;
      if (px < HP16 * 16) put(sxr[k], px * 64 + cqx * 8);
    }
#pragma unroll
    for (int k = 0; k < NDS; ++k) {
      const int e = tid + 512 * k;
      const int q = e / DYQ, c0 = 4 * (e - q * DYQ);
      int off;
      if (OT == 2) off = q * DYROW + ((((c0 >> 3) ^ (((q >> 1) & 1) << 2))) << 4) + ((c0 >> 2) & 1) * 8;
      else off = q * DYROW + c0 * 2;
      put(sdr[k], XBUF + off);
    }
  };

  // ---- DMA of one tile (unit u of this expert) into buffer `b`; piece k of this wave (k static at every call site)
  auto tile_origin = [&](int u, int& n, int& ty0, int& tx0) {
    const int img = u / a.tpi, ti = u - img * a.tpi;
    const int tyi = ti / a.tiles_x;
    n = row0 + img; ty0 = tyi * TH; tx0 = (ti - tyi * a.tiles_x) * TW;
  };
  auto issue_x = [&](int n, int ty0, int tx0, int k, int b) {
    const int piece = wave + 8 * k;
    if (piece >= HP16) return;
    const int px = 16 * piece + (lane >> 2);
    const int hy = px / HWp, hx = px - hy * HWp;
    const int iy = ty0 - a.pt + hy, ix = tx0 - a.pl + hx;
    const bool ok = px < HWp * HHp && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    const unsigned off = ok ? (unsigned)((((n * a.H + iy) * a.W + ix) * a.Cin + i0) * 2 + (lane & 3) * 16) : 0xFFFFFFFFu;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(lds + b * BUF + piece * 1024), 16, off, 0, 0, 0);
  };
  auto issue_dy = [&](int n, int ty0, int tx0, int k, int b) {
    const int piece = wave + 8 * k;                           // < DYP by construction (NDP * 8 == DYP)
    int q, slot;
    if (OT == 2) { q = 8 * piece + (lane >> 3); slot = (lane & 7) ^ (((lane >> 4) & 1) << 2); }   // 128-B rows: halves swapped on odd row pairs
    else { q = 16 * piece + (lane >> 2); slot = lane & 3; }
    const int oy = ty0 + (q >> TWS), ox = tx0 + (q & (TW - 1));
    const bool ok = oy < a.H && ox < a.W;
    const unsigned off = ok ? (unsigned)((((n * a.H + oy) * a.W + ox) * a.Cout + o0) * 2 + slot * 16) : 0xFFFFFFFFu;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lptr_t)(lds + b * BUF + XBUF + piece * 1024), 16, off, 0, 0, 0);
  };

  // ---- per-lane LDS read addresses (bytes, buffer 0): the 16-k-step loop adds instruction immediates only
  //  x fragment of tap slot s: rows = halo pixels (8h + q4 [+4]) + tap offset, 64-byte rows, columns col4
  int tapoff[NFULL + 1];
#pragma unroll
  for (int s = 0; s < NFULL; ++s) {
    const int tap = wave + 8 * s;
    tapoff[s] = ((tap / KS) * HWp + tap % KS) * 64;
  }
  const bool extra_ok = wave < REM * OT;
  const int etile = extra_ok ? wave : 0;
  const int etap = 8 * NFULL + etile / OT, eob = etile % OT;
  tapoff[NFULL] = ((etap / KS) * HWp + etap % KS) * 64;
  const int xlane = (8 * h + q4) * 64 + col4 * 2;
  //  dy fragment of channel half t: rows = tile pixels 8h + q4 [+4]; 128-byte rows are stored with their 64-byte halves swapped
  //  where bit 1 of the row index is set (the DMA source swizzle above), which makes a 4-row x 64-byte transposing read conflict-free
  int dylane[OT];
#pragma unroll
  for (int t = 0; t < OT; ++t) {
    const int c = 32 * t + col4;                              // first channel of this lane's 4-channel column group
    if (OT == 2) dylane[t] = (8 * h + q4) * DYROW + ((((c >> 3) ^ (((q4 >> 1) & 1) << 2))) << 4) + (c & 7) * 2;
    else dylane[t] = (8 * h + q4) * DYROW + c * 2;
  }
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  auto tr2 = [&](int addr, int rowstep4) -> bf16x8 {          // rows r..r+3 at addr, rows r+4..r+7 at addr + rowstep4
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds + addr));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds + addr + rowstep4));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  };

  f32x16 acc[NFULL][OT], acce;
#pragma unroll
  for (int s = 0; s < NFULL; ++s)
#pragma unroll
    for (int t = 0; t < OT; ++t) acc[s][t] = (f32x16)(0.f);
  acce = (f32x16)(0.f);

  if constexpr (SPLIT) {
    {
      int n, ty0, tx0;
      tile_origin(u0, n, ty0, tx0);
      split_load(n, ty0, tx0);
      split_store();
    }
    for (int u = u0; u < u1; ++u) {
      __syncthreads();                                        // tile u's planes are complete
      const bool more = u + 1 < u1;
      if (more) {                                             // next tile: fp32 registers, in flight beside the loop below
        int nn, nty0, ntx0;
        tile_origin(u + 1, nn, nty0, ntx0);
        split_load(nn, nty0, ntx0);
      }
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const int krow = (16 * ks) >> TWS, kcol = (16 * ks) & (TW - 1);
        const int kx_off = (krow * HWp + kcol) * 64;
        const int kdy_off = 16 * ks * DYROW;
        bf16x8 dh[OT], dl[OT], xh[NFULL + 1], xl[NFULL + 1];
#pragma unroll
        for (int t = 0; t < OT; ++t) { dh[t] = tr2(XBUF + dylane[t] + kdy_off, 4 * DYROW); dl[t] = tr2(BUF + XBUF + dylane[t] + kdy_off, 4 * DYROW); }
#pragma unroll
        for (int s2 = 0; s2 <= NFULL; ++s2) { xh[s2] = tr2(xlane + tapoff[s2] + kx_off, 4 * 64); xl[s2] = tr2(BUF + xlane + tapoff[s2] + kx_off, 4 * 64); }
#pragma unroll
        for (int s2 = 0; s2 < NFULL; ++s2)
#pragma unroll
          for (int t = 0; t < OT; ++t) {
            acc[s2][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dh[t], xh[s2], acc[s2][t], 0, 0, 0);
            acc[s2][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dh[t], xl[s2], acc[s2][t], 0, 0, 0);
            acc[s2][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dl[t], xh[s2], acc[s2][t], 0, 0, 0);
          }
        const bf16x8 deh = (OT == 2 && eob) ? dh[OT - 1] : dh[0], del = (OT == 2 && eob) ? dl[OT - 1] : dl[0];
        acce = __builtin_amdgcn_mfma_f32_32x32x16_bf16(deh, xh[NFULL], acce, 0, 0, 0);
        acce = __builtin_amdgcn_mfma_f32_32x32x16_bf16(deh, xl[NFULL], acce, 0, 0, 0);
        acce = __builtin_amdgcn_mfma_f32_32x32x16_bf16(del, xh[NFULL], acce, 0, 0, 0);
      }
      if (more) {
        __syncthreads();                                      // every wave is done reading tile u
        split_store();
      }
    }
  } else {
  // prologue: first tile into buffer 0
  {
    int n, ty0, tx0;
    tile_origin(u0, n, ty0, tx0);
#pragma unroll
    for (int k = 0; k < NXP; ++k) issue_x(n, ty0, tx0, k, 0);
#pragma unroll
    for (int k = 0; k < NDP; ++k) issue_dy(n, ty0, tx0, k, 0);
  }
  int par = 0;
  for (int u = u0; u < u1; ++u) {
    __syncthreads();                                          // tile u has landed in buffer par; buffer par^1 is free again
    const bool more = u + 1 < u1;
    int nn = 0, nty0 = 0, ntx0 = 0;
    if (more) tile_origin(u + 1, nn, nty0, ntx0);
    const int xb = par * BUF, dyb = par * BUF + XBUF;
    // pixels of this tile that lie inside the image take part (rows past the image bottom were DMA'd as zeros: no masking needed)
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      // halo offset of the k-step's first pixel (compile time): tile pixel 16 ks -> (row, column)
      const int krow = (16 * ks) >> TWS, kcol = (16 * ks) & (TW - 1);
      const int kx_off = (krow * HWp + kcol) * 64;            // folds into the instruction offset
      const int kdy_off = 16 * ks * DYROW;
      bf16x8 fdy[OT], fdye, fx[NFULL], fxe;
#pragma unroll
      for (int t = 0; t < OT; ++t) fdy[t] = tr2(dyb + dylane[t] + kdy_off, 4 * DYROW);
#pragma unroll
      for (int s = 0; s < NFULL; ++s) fx[s] = tr2(xb + xlane + tapoff[s] + kx_off, 4 * 64);
      fdye = (OT == 2 && eob) ? fdy[OT - 1] : fdy[0];           // wave-uniform select: the left-over tile's channel half
      fxe = tr2(xb + xlane + tapoff[NFULL] + kx_off, 4 * 64);
      // next tile's DMA pieces, one per k-step
      if (more) {
        if (ks < NXP) issue_x(nn, nty0, ntx0, ks, par ^ 1);
        else if (ks - NXP < NDP) issue_dy(nn, nty0, ntx0, ks - NXP, par ^ 1);
      }
#pragma unroll
      for (int s = 0; s < NFULL; ++s)
#pragma unroll
        for (int t = 0; t < OT; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fdy[t], fx[s], acc[s][t], 0, 0, 0);
      acce = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fdye, fxe, acce, 0, 0, 0);
    }
    par ^= 1;
  }
  }
  // ---- partial slab [tap][Cout][Cin] of this (expert, pixel partition): plain stores, 128-byte runs
  float* P = a.ws + (long)zslot * a.ws_item;
  const int col = lane & 31;
  auto store_tile = [&](const f32x16& v, int tap, int ob) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int o = o0 + 32 * ob + acc_row(reg, lane);
      P[((long)tap * a.Cout + o) * a.Cin + i0 + col] = v[reg];
    }
  };
#pragma unroll
  for (int s = 0; s < NFULL; ++s)
#pragma unroll
    for (int t = 0; t < OT; ++t) store_tile(acc[s][t], wave + 8 * s, t);
  if (extra_ok) store_tile(acce, etap, eob);
#endif
}

template <int KS, int TWS, int OT, bool SPLIT>
__global__ __launch_bounds__(512) void wgrad6_kernel(W6Args a) {
  wgrad6_body<KS, TWS, OT, SPLIT>(a, blockIdx.z);
}
// Both kernel-size classes of a layer (3x3 experts and 5x5 experts) in ONE launch: partition slots [0, a3.chunks) run the 3x3 program,
// the rest the 5x5 program.  As two launches the 3x3 class -- a third of the 5x5 class's FLOPs but latency-bound, hence just as long
// -- ran in front of the 5x5 class; side by side it disappears under it.
template <int TWS, int OT>
__global__ __launch_bounds__(512) void wgrad6_dual_kernel(W6Args a3, W6Args a5) {
  if ((int)blockIdx.z < a3.chunks) wgrad6_body<3, TWS, OT, false>(a3, blockIdx.z);
  else wgrad6_body<5, TWS, OT, false>(a5, blockIdx.z - a3.chunks);
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
void w6_partition(long units_l, int ngr, int ngroups, int ibs, int obs, int& upw, int& slots) {
  long parts = 256 / ((long)ibs * obs); if (parts < 1) parts = 1;         // ~256 workgroups alive for balanced routing
  const long class_units = (units_l * ngr + ngroups - 1) / ngroups;
  long u = (class_units + parts - 1) / parts; if (u < 1) u = 1;
  while (units_l / u + ngr > 1024) ++u;
  upw = (int)u; slots = (int)(units_l / u + ngr);
}

template <int KS, int TWS, int OT, bool SPLIT>
void launch_w6(const W6Args& a, int ibs, int obs, hipStream_t stream) {
  constexpr int TW = 1 << TWS, TH = 256 >> TWS, HP16 = ((TW + KS - 1) * (TH + KS - 1) + 15) / 16;
  const size_t lds = 2 * (size_t)(HP16 * 1024 + 16 * OT * 1024);     // two DMA buffers, or (SPLIT) a hi and a lo plane
  static bool attr = false;
  if (!attr) { attr = true; (void)hipFuncSetAttribute((const void*)wgrad6_kernel<KS, TWS, OT, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
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
    w6_partition(units_l, ngr, ngroups, Cin / 32, Cout / (32 * OT), upw, slots);
    const long b = (long)slots * kh[g] * kh[g] * Cout * Cin * 4;
    if (b > bytes) bytes = b;
  }
  return bytes >= (1l << 40) ? 0 : (int)((bytes + 1023) >> 10);
}

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
  const long units_l = (long)N * a.tpi;
  bool done[HDMOE_MAX_GROUPS] = {false};
  int cls = 0;
  static const bool dual_ok = !(getenv("HDMOE_W6_DUAL") && atoi(getenv("HDMOE_W6_DUAL")) == 0);
  bool has3 = false, has5 = false;
  for (int g = 0; g < ngroups; ++g) { has3 = has3 || kh[g] == 3; has5 = has5 || kh[g] == 5; }
  if (dual_ok && defer && has3 && has5 && dtype == HDMOE_BF16) {
    W6Args c[2];
    for (int k = 0; k < 2; ++k) {
      const int ks = k == 0 ? 3 : 5;
      c[k] = a;
      c[k].ngr = 0;
      for (int g = 0; g < ngroups; ++g) if (kh[g] == ks) { c[k].groups[c[k].ngr++] = g; c[k].pt = pt[g]; c[k].pl = pl[g]; }
      c[k].ws_item = (long)ks * ks * Cout * Cin;
      int upw, slots;
      w6_partition(units_l, c[k].ngr, ngroups, ibs, obs, upw, slots);
      c[k].upw = upw; c[k].chunks = slots;
    }
    // workspace regions in the order the classes appear in the group list (what hdmoe_conv_wgrad6_reduce_batch assumes)
    const int firstk = kh[0] == 3 ? 0 : 1;
    c[firstk].ws = (float*)ws; c[1 - firstk].ws = (float*)((char*)ws + need1);
    const dim3 grid(ibs, obs, c[0].chunks + c[1].chunks);
#define W6_DUAL(T, O)                                                                                                              \
    do {                                                                                                                           \
      constexpr int TW_ = 1 << T, TH_ = 256 >> T, HP16_ = ((TW_ + 4) * (TH_ + 4) + 15) / 16;                                        \
      const size_t lds_ = 2 * (size_t)(HP16_ * 1024 + 16 * O * 1024);                                                              \
      static bool attr_ = false;                                                                                                   \
      if (!attr_) { attr_ = true; (void)hipFuncSetAttribute((const void*)wgrad6_dual_kernel<T, O>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); } \
      hipLaunchKernelGGL((wgrad6_dual_kernel<T, O>), grid, dim3(512), lds_, stream, c[0], c[1]);                                   \
    } while (0)
    if (TWS == 5) { if (OT == 2) W6_DUAL(5, 2); else W6_DUAL(5, 1); } else { if (OT == 2) W6_DUAL(4, 2); else W6_DUAL(4, 1); }
    return hdmoe_launch_status();
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
    w6_partition(units_l, a.ngr, ngroups, ibs, obs, upw, slots);
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
      w6_partition(units_l, it.ngr, ngroups, Cin / 32, Cout / (32 * OT), upw, slots);
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
