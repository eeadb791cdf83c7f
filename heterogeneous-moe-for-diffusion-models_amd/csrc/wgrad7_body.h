// Device body of the wgrad7 program: weight gradient of a k x k (k = 3, 5) bf16 expert layer on 32 x 32 maps, streamed per half image
// (autograd of MP_Conv, reference models/model_internals.py:253-275 via F.conv2d):
//     dW[g][tap][o][i] = sum over the pixels p of expert g's rows of dy[p][o] * x[p + tap][i].
// Same partial-slab contract as wgrad6 (wgrad6_body.h: one [tap][Cout][Cin] fp32 slab per partition slot, summed in a fixed order by
// wgrad6_reduce_*), different work split -- wgrad6 gives every wave whole taps (taps w, w + 8, ..: 9 taps over 8 waves leave seven waves
// idle half of the time, 25 taps run at 25 / 32) and 256-pixel tiles with their halo.  Here
//   * a workgroup owns one (input chunk, output chunk, partition slot) and walks the slot's images in half-image units (16 dy rows + the
//     16 + k - 1 x rows they touch; rows outside the image arrive as zeros from the DMA), two units in flight in LDS (2 x 73 - 78 KB);
//   * the 8 waves split (tap group, pixel part): 3x3 -> every wave all 9 taps on 2 of the 16 rows; 5x5 -> 4 tap groups (7, 6, 6, 6 taps) x 2
//     pixel parts -- 100 % / 89 % of the issued MFMAs are useful -- and a dy fragment is read once per 16-pixel block for all of a wave's taps;
//   * tiles are plain [row][35 slots][64 B] / [row][32][64 B] images (no swizzle: only transposing reads touch them, and four consecutive
//     64-B pixels cover all banks); every LDS address is a per-lane base (one per tap, rebuilt per unit) plus an instruction immediate;
//   * the pixel parts of a tap meet in LDS once, at the end of the workgroup's life, and leave as 128-byte-run stores.
#pragma once
#include "common.h"
#include "wgrad6_body.h"

namespace {

template <int KS>
DEVI void wgrad7_body(const W6Args& a, const int bx, const int by, const int zslot) {
#if __HIP_DEVICE_COMPILE__
  constexpr int P = (KS - 1) / 2, Q = 3 - P, NTAPS = KS * KS;
  constexpr int NG = KS == 3 ? 1 : 4, NP = 8 / NG;            // tap groups x pixel parts = 8 waves
  constexpr int TPW = (NTAPS + NG - 1) / NG;                  // taps per wave (3x3: 9, 5x5: 7)
  constexpr int RPW = 16 / NP;                                // dy rows of a unit per wave (2 / 8)
  constexpr int XROWS = 16 + KS - 1, XROWB = 35 * 64;
  constexpr int XB = XROWS * XROWB + 3 * 64, DYB = 16 * 2048, STAGE = XB + DYB;
  constexpr int NXP = XROWS * 2, NDP = 32, NPIECE = (NXP + NDP + 7) / 8;      // DMA pieces of a unit (x rows x halves, dy rows x halves), per wave
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i0 = bx * 32, o0 = by * 32;
  int nstamp = 0;
  auto stamp = [&](int tag) {
    if (a.stamps && bx == 0 && by == 0 && zslot == 0 && lane == 0 && nstamp < 63) {
      a.stamps[(KS == 3 ? 512 : 1024) + wave * 64 + nstamp] = ((unsigned long long)tag << 56) | (__builtin_amdgcn_s_memtime() & 0x00FFFFFFFFFFFFFFull);
      ++nstamp;
    }
  };
  stamp(1);
  // partition slot -> (expert of this class, image range): experts take ceil(units / upw) consecutive slots each (upw: 256-pixel tiles, a
  // multiple of the 4 tiles of an image -- w6_partition)
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
  const int t0 = chunk * a.upw, t1 = min(units, t0 + a.upw);
  const int n0 = row0 + t0 / 4, n1 = row0 + (t1 + 3) / 4;     // images of this slot
  const int U = 2 * (n1 - n0);                                // half-image units
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, a.dybytes, 0x00020000);

  // ---- zero the pad slots of both x tiles (the DMA only writes data slots)
  for (int i = tid; i < (XROWS + 1) * 12; i += 512) {
    const int off = (i / 12) * XROWB + (i % 12) * 16;
    *reinterpret_cast<uint4*>(lds + off) = make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(lds + STAGE + off) = make_uint4(0, 0, 0, 0);
  }

  // ---- DMA: piece k of this wave; pieces [0, NXP) are x (tile row k >> 1, half k & 1), the rest dy
  const unsigned xl = (unsigned)((lane >> 2) * a.Cin * 2 + (lane & 3) * 16), dyl = (unsigned)((lane >> 2) * a.Cout * 2 + (lane & 3) * 16);
  auto issue_unit = [&](int u, int sb) {
    const int n = n0 + (u >> 1), r0 = (u & 1) * 16;
#pragma unroll
    for (int k = 0; k < NPIECE; ++k) {
      const int pc = wave + 8 * k;
      if (pc < NXP) {
        const int j = pc >> 1, half = pc & 1, row = r0 - P + j;
        const bool ok = (unsigned)row < 32u;
        const int so = ((n * 32 + row) * 32 + half * 16) * a.Cin * 2 + i0 * 2;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(lds + sb + j * XROWB + (3 + half * 16) * 64), 16, ok ? xl : 0xFFFFFFFFu, ok ? so : 0, 0, 0);
      } else if (pc < NXP + NDP) {
        const int d = pc - NXP, j = d >> 1, half = d & 1;
        const int so = ((n * 32 + r0 + j) * 32 + half * 16) * a.Cout * 2 + o0 * 2;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lptr_t)(lds + sb + XB + j * 2048 + half * 1024), 16, dyl, so, 0, 0);
      }
    }
  };

  // ---- per-lane read addresses (transposing reads: lane 4q + p of a 16-lane group supplies pixel q, channels 4p .. of the group's block)
  const int h = lane >> 5, cb = (lane >> 4) & 1, q = (lane & 15) >> 2, p4 = lane & 3;
  const int lbase = (8 * h + q) * 64 + cb * 32 + p4 * 8;       // pixel 8h + q of a 16-pixel block, channels 16 cb + 4 p4 ..
  const int tg = wave % NG, pp = wave / NG;
  int xoff[TPW];                                               // per tap of this wave: (ky, kx) displacement inside the x tile
  bool tvalid[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int t = tg + NG * j;
    tvalid[j] = t < NTAPS;
    const int tt = tvalid[j] ? t : 0;
    xoff[j] = (tt / KS) * XROWB + (tt % KS + Q) * 64;
  }
  const unsigned lds0 = lds_addr_of(lds);

  f32x16 acc[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) acc[j] = (f32x16)(0.f);

  stamp(2);
  if (U > 0) issue_unit(0, 0);
  stamp(3);
  for (int u = 0; u < U; ++u) {
    stamp(4);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // unit u has landed; the other stage is free again
    stamp(5);
    const int sb = (u & 1) * STAGE;
    if (u + 1 < U) issue_unit(u + 1, STAGE - sb);
    const int dyb = sb + XB + lbase + pp * RPW * 2048;
    int xb[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) xb[j] = sb + lbase + xoff[j] + pp * RPW * XROWB;
    // 16-pixel blocks of this wave's rows: (row b >> 1, half b & 1).  Fragments through the asm reads (common.h: the compiler's own transposing
    // reads would wait for the next unit's DMA first), requested one block ahead of their MFMAs (two register sets)
    hd_s16x4 dlo[2], dhi[2], xlo[2][TPW], xhi[2][TPW];
    auto request = [&](int b, int set) {
      const int rr = b >> 1, hh = b & 1;
      lds_tr2_issue(dlo[set], dhi[set], lds0 + dyb + rr * 2048 + hh * 1024, lds0 + dyb + rr * 2048 + hh * 1024 + 256);
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        const unsigned ad = lds0 + xb[j] + rr * XROWB + hh * 1024;
        lds_tr2_issue(xlo[set][j], xhi[set][j], ad, ad + 256);
      }
    };
    request(0, 0);
#pragma unroll
    for (int b = 0; b < RPW * 2; ++b) {
      lds_tr_wait();
      const bf16x8 fdy = lds_tr2_take(dlo[b & 1], dhi[b & 1]);
      bf16x8 fx[TPW];
#pragma unroll
      for (int j = 0; j < TPW; ++j) fx[j] = lds_tr2_take(xlo[b & 1][j], xhi[b & 1][j]);
      if (b + 1 < RPW * 2) request(b + 1, (b + 1) & 1);
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        if (NTAPS % NG == 0 || j + 1 < TPW || tvalid[j])       // (only a group's last tap can be missing)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fdy, fx[j], acc[j], 0, 0, 0);
      }
    }
  }
  stamp(6);
  // ---- the NP pixel parts of every tap meet in LDS (fixed order), then leave as 128-byte-run stores into the slot's partial slab
  float* Pw = a.ws + (long)zslot * a.ws_item;
  float* red = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the tiles (first round) / the previous round's sums are consumed (no vmcnt: the DMA was drained at the last unit's barrier, the stores stay in flight)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) red[wave * 1024 + reg * 64 + lane] = acc[j][reg];
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int m = 0; m < 2 * NG; ++m) {
      const int e = tid + 512 * m;                             // element of tap-group tile e >> 10
      const int tgi = e >> 10, el = e & 1023;
      const int t = tgi + NG * j;
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < NP; ++k) s += red[(tgi + NG * k) * 1024 + el];
      if (t < NTAPS) {
        const int reg = el >> 6, ln = el & 63;
        Pw[((long)t * a.Cout + o0 + acc_row(reg, ln)) * a.Cin + i0 + (ln & 31)] = s;
      }
    }
  }
  stamp(7);
#endif
}

}  // namespace
