// Device body of the wgrad6 kernel (see wgrad6.hip for the design notes): shared by the stand-alone kernels and by the fused backward
// launch of bwd6.hip.  (bx, by) = (input-channel block, output-channel block), zslot = pixel-partition slot of the kernel-size class.
#pragma once
#include "common.h"

struct W6Args {
  const void* x; const void* dy; float* ws; const int* seg;
  int N, H, W, Cin, Cout;
  int groups[HDMOE_MAX_GROUPS]; int ngr;
  int pt, pl;
  int tiles_x, tpi;
  int upw, chunks;
  int xbytes, dybytes;
  long ws_item;                       // floats of one partial slab (taps * Cout * Cin)
  // SPLIT only: the x operand is relu(x * in_scale[n][c] + in_shift[n][c]) (GroupNorm(1, C) + ReLU of the producing layer applied while
  // the tile is staged -- the normalised activation is never materialised; padding pixels stay zero).  Null = plain x.
  const float* in_scale; const float* in_shift; int in_relu;
  int hi_only;                        // SPLIT only: 1 = drop the two hi x lo correction products (bf16 operands, fp32 accumulation)
  int icw, ocw;                       // wgrad8 (3x3, 32 x 32 maps): 32-channel chunks per workgroup on the input / output side (1 or 2)
  unsigned long long* stamps;         // development: s_memtime stamps (wgrad7 program, first workgroup of each class), or null
};
struct W6DualPlan { W6Args c[2]; int ibs, obs, TWS, OT; size_t lds; };
// Launch geometry of the deferred dual-class bf16 weight gradient of one layer (wgrad6.hip).  0 = planned, 1 = not applicable.
int wgrad6_plan_dual(const void* x, const void* dy, float* const* G, const int* seg, int ngroups, int N, int H, int W, int Cin, int Cout,
                     const int* kh, const int* kw, const int* pt, const int* pl, void* ws, long ws_bytes, int dtype, W6DualPlan& p);

// The same for the split-bf16 (fp32 tensors, 3x3) weight gradient: one class in c[0].
int wgrad6_plan_split(const void* x, const void* dy, float* const* G, const int* seg, int ngroups, int N, int H, int W, int Cin, int Cout,
                      const int* kh, const int* kw, const int* pt, const int* pl, void* ws, long ws_bytes, W6DualPlan& p);

#include <type_traits>
#include <utility>

namespace {

template <typename F, int... I> DEVI void w6_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F> DEVI void w6_static_for(F&& f) { w6_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;


// SPLIT: x and dy are fp32 (the router trunks); every operand is split into hi + lo bf16 while it is staged (through registers) and
// a product is three MFMAs, dy_hi x_hi + dy_hi x_lo + dy_lo x_hi -- see conv6s.hip.  LDS then holds a hi and a lo plane of each
// tile (single-buffered: the next tile waits in registers, loaded beside the current tile's loop).
template <int KS, int TWS, int OT, bool SPLIT>
DEVI void wgrad6_body(const W6Args& a, const int bx, const int by, const int zslot) {
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
  const unsigned lds0 = lds_addr_of(lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5;
  const int q4 = (lane & 15) >> 2, col4 = (lane & 16) + 4 * (lane & 3);
  const int i0 = bx * 32, o0 = by * 32 * OT;
  // partition slot blockIdx.z -> (expert of this class, pixel partition): experts take ceil(units / upw) consecutive slots each
  int gi = 0, chunk = zslot, row0 = 0, units = 0;
  for (; gi < a.ngr; ++gi) {
    const int g = a.groups[gi];
    row0 = a.seg ? __builtin_amdgcn_readfirstlane(a.seg[g]) : 0;               // (scalar: see wgrad8_body.h w8_slot)
    units = ((a.seg ? __builtin_amdgcn_readfirstlane(a.seg[g + 1]) : a.N) - row0) * a.tpi;
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
  // (the input transform is applied in split_store, when the registers are consumed: applying it here would wait for the loads at
  //  once and take the prefetch away)
  f4 sx_sc = (f4)(1.f), sx_sh = (f4)(0.f);
  unsigned sx_ok = 0;                                         // bit k: piece k is an image pixel (padding must stay zero after the transform)
  auto split_load = [&](int n, int ty0, int tx0) {
    sx_ok = 0;
#pragma unroll
    for (int k = 0; k < NXS; ++k) {
      const int e = tid + 512 * k;
      const int px = e / XQ, cqx = e - px * XQ;
      const int hy = px / HWp, hx = px - hy * HWp;
      const int iy = ty0 - a.pt + hy, ix = tx0 - a.pl + hx;
      const bool ok = px < HWp * HHp && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const unsigned off = ok ? (unsigned)((((n * a.H + iy) * a.W + ix) * a.Cin + i0) * 4 + cqx * 16) : 0xFFFFFFFFu;
      sxr[k] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
      if (ok) sx_ok |= 1u << k;
    }
    if (a.in_scale) {                                         // (wave-uniform branch)  this thread's channel quad is the same for all its pieces
      const int cqx = tid & (XQ - 1);
      sx_sc = *reinterpret_cast<const f4*>(a.in_scale + (long)n * a.Cin + i0 + cqx * 4);
      sx_sh = *reinterpret_cast<const f4*>(a.in_shift + (long)n * a.Cin + i0 + cqx * 4);
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
      if (!a.hi_only) *reinterpret_cast<bf16x4*>(lds + BUF + off) = lo;
    };
#pragma unroll
    for (int k = 0; k < NXS; ++k) {
      const int e = tid + 512 * k;
      const int px = e / XQ, cqx = e - px * XQ;
      f4 v = sxr[k];
      if (a.in_scale) {
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          float t = v[e2] * sx_sc[e2] + sx_sh[e2];
          if (a.in_relu) t = fmaxf(t, 0.f);
          v[e2] = ((sx_ok >> k) & 1u) ? t : 0.f;
        }
      }
      if (px < HP16 * 16) put(v, px * 64 + cqx * 8);
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
        bf16x8 dh[OT], xh[NFULL + 1];
#pragma unroll
        for (int t = 0; t < OT; ++t) dh[t] = tr2(XBUF + dylane[t] + kdy_off, 4 * DYROW);
#pragma unroll
        for (int s2 = 0; s2 <= NFULL; ++s2) xh[s2] = tr2(xlane + tapoff[s2] + kx_off, 4 * 64);
#pragma unroll
        for (int s2 = 0; s2 < NFULL; ++s2)
#pragma unroll
          for (int t = 0; t < OT; ++t) acc[s2][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dh[t], xh[s2], acc[s2][t], 0, 0, 0);
        const bf16x8 deh = (OT == 2 && eob) ? dh[OT - 1] : dh[0];
        acce = __builtin_amdgcn_mfma_f32_32x32x16_bf16(deh, xh[NFULL], acce, 0, 0, 0);
        if (!a.hi_only) {                                       // (wave-uniform) the two correction products: dy_hi * x_lo + dy_lo * x_hi
          bf16x8 dl[OT], xl[NFULL + 1];
#pragma unroll
          for (int t = 0; t < OT; ++t) dl[t] = tr2(BUF + XBUF + dylane[t] + kdy_off, 4 * DYROW);
#pragma unroll
          for (int s2 = 0; s2 <= NFULL; ++s2) xl[s2] = tr2(BUF + xlane + tapoff[s2] + kx_off, 4 * 64);
#pragma unroll
          for (int s2 = 0; s2 < NFULL; ++s2)
#pragma unroll
            for (int t = 0; t < OT; ++t) {
              acc[s2][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dh[t], xl[s2], acc[s2][t], 0, 0, 0);
              acc[s2][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dl[t], xh[s2], acc[s2][t], 0, 0, 0);
            }
          const bf16x8 del = (OT == 2 && eob) ? dl[OT - 1] : dl[0];
          acce = __builtin_amdgcn_mfma_f32_32x32x16_bf16(deh, xl[NFULL], acce, 0, 0, 0);
          acce = __builtin_amdgcn_mfma_f32_32x32x16_bf16(del, xh[NFULL], acce, 0, 0, 0);
        }
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
    // Fragments through the asm transposing reads (common.h), requested one k-step ahead: the compiler's own reads wait for the DMA piece
    // issued in the previous k-step (a memory round trip per k-step) and sit right in front of their MFMAs.
    hd_s16x4 plo[2][OT + NFULL + 1], phi[2][OT + NFULL + 1];
    auto request = [&](auto ks_c, int set) {
      constexpr int ks = decltype(ks_c)::value;
      constexpr int krow = (16 * ks) >> TWS, kcol = (16 * ks) & (TW - 1);
      constexpr int kx_off = (krow * HWp + kcol) * 64, kdy_off = 16 * ks * DYROW;   // halo offset of the k-step's first pixel: tile pixel 16 ks -> (row, column)
#pragma unroll
      for (int t = 0; t < OT; ++t) { const unsigned ad = lds0 + dyb + dylane[t] + kdy_off; lds_tr2_issue(plo[set][t], phi[set][t], ad, ad + 4 * DYROW); }
#pragma unroll
      for (int s2 = 0; s2 <= NFULL; ++s2) { const unsigned ad = lds0 + xb + xlane + tapoff[s2] + kx_off; lds_tr2_issue(plo[set][OT + s2], phi[set][OT + s2], ad, ad + 4 * 64); }
    };
    request(std::integral_constant<int, 0>{}, 0);
    w6_static_for<16>([&](auto ks_c) {
      constexpr int ks = decltype(ks_c)::value;
      lds_tr_wait();
      bf16x8 fdy[OT], fdye, fx[NFULL], fxe;
#pragma unroll
      for (int t = 0; t < OT; ++t) fdy[t] = lds_tr2_take(plo[ks & 1][t], phi[ks & 1][t]);
#pragma unroll
      for (int s2 = 0; s2 < NFULL; ++s2) fx[s2] = lds_tr2_take(plo[ks & 1][OT + s2], phi[ks & 1][OT + s2]);
      fxe = lds_tr2_take(plo[ks & 1][OT + NFULL], phi[ks & 1][OT + NFULL]);
      fdye = (OT == 2 && eob) ? fdy[OT - 1] : fdy[0];           // wave-uniform select: the left-over tile's channel half
      if (ks + 1 < 16) request(std::integral_constant<int, (ks + 1 < 16 ? ks + 1 : 0)>{}, (ks + 1) & 1);
      // next tile's DMA pieces, one per k-step
      if (more) {
        if (ks < NXP) issue_x(nn, nty0, ntx0, ks, par ^ 1);
        else if (ks - NXP < NDP) issue_dy(nn, nty0, ntx0, ks - NXP, par ^ 1);
      }
#pragma unroll
      for (int s2 = 0; s2 < NFULL; ++s2)
#pragma unroll
        for (int t = 0; t < OT; ++t) acc[s2][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fdy[t], fx[s2], acc[s2][t], 0, 0, 0);
      acce = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fdye, fxe, acce, 0, 0, 0);
    });
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

}  // namespace
