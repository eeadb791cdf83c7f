// Device bodies of the wgrad8 programs: weight gradient of a k x k (k = 3, 5) bf16 layer on 32 x 32 maps (autograd of MP_Conv, reference
// models/model_internals.py:253-275 via F.conv2d):
//     dW[g][tap][o][i] = sum over the pixels p of expert g's rows of dy[p][o] * x[p + tap][i].
// Same partial-slab contract as wgrad6 / wgrad7 (one [tap][Cout][Cin] fp32 slab per partition slot, summed in a fixed order by
// wgrad6_reduce_*).  What the in-kernel stamps of wgrad7 showed (tools/conv7_check.py --stamps-bwd, 32 -> 32, B = 256): a 3x3 unit ran at
// 40 % of its MFMA time and a 5x5 unit at 65 % -- one LDS fragment read per MFMA (the LDS pipe is busy 8 cycles per fragment, 4 SIMDs x one
// MFMA per 32 cycles need 32 of 32), the next unit's DMA issued in one burst in front of the MFMAs (the last wave through the address unit
// started ~1000 cycles late, every unit) -- and the closing cross-wave reduction took 11-12 k cycles of a 65 k-cycle workgroup (one round
// and two barriers per tap).  And a 64 -> 64 layer read x and dy twice each (one 32 x 32 channel chunk pair per workgroup), a 128 -> 128
// trunk layer four times: 604 MB per launch for 2 x 67 MB of operands.  Here
//   * a fragment stays in registers for every MFMA that needs it: a wave walks its dy rows once per 16-pixel column block and holds them;
//     an x row fragment of kernel column kx is read ONCE and meets the dy rows r, r - 1, .. of the kernel rows ky = 0, 1, .. (a sliding
//     window down the image): 3x3: 14 reads per 18 MFMAs (2 rows per wave) / 22 per 36 (4 rows), 5x5: ~29 per 52;
//   * 3x3: a workgroup owns IC x OC channel chunks of 32 (2 x 2 where the layer has them): the 8 waves are (chunk pair) x (pixel part), x
//     and dy cross the fabric once per layer instead of Cout / 32 and Cin / 32 times; units are quarter images then (8 dy rows + halo:
//     two units of 2 x 22 + 2 x 16 KB in flight);
//   * the next unit's DMA pieces are issued between the MFMA groups, two at a time;
//   * the reduction moves four taps per round (128 KB of LDS: the tiles are dead by then) with 16-byte reads and stores.
#pragma once
#include "common.h"
#include "wgrad6_body.h"

namespace {

// Slot -> (first image, image count) of this class: experts take ceil(units / upw) consecutive slots each (w6_partition: upw is a
// multiple of the 4 tiles of an image).  False: the slot does not exist for this routing.
DEVI bool w8_slot(const W6Args& a, const int zslot, int& n0, int& n1) {
  int gi = 0, chunk = zslot, row0 = 0, units = 0;
  for (; gi < a.ngr; ++gi) {
    const int g = a.groups[gi];
    // (readfirstlane: loads behind a store of the same kernel are vector loads, and a "divergent" image index turns every DMA issue
    // into a waterfall loop)
    row0 = a.seg ? __builtin_amdgcn_readfirstlane(a.seg[g]) : 0;
    units = ((a.seg ? __builtin_amdgcn_readfirstlane(a.seg[g + 1]) : a.N) - row0) * a.tpi;
    const int nch = (units + a.upw - 1) / a.upw;
    if (chunk < nch) break;
    chunk -= nch;
  }
  if (gi == a.ngr) return false;
  const int t0 = chunk * a.upw, t1 = min(units, t0 + a.upw);
  n0 = row0 + t0 / 4; n1 = row0 + (t1 + 3) / 4;
  return true;
}

// The closing reduction: every wave holds NACC accumulator tiles (32 x 32 fp32); tile j of wave w belongs to output (group w % NGR, tap
// tap_of(j)) and is one of NP = 8 / NGR pixel-part partial sums of it.  Rounds of four tiles per wave through LDS (fixed summation order).
template <int NACC, typename TapOf, typename Store>
DEVI void w8_reduce(unsigned char* lds, const f32x16 (&acc)[NACC], const int NGR, const int wave, const int lane, const int tid, TapOf tap_of, Store store) {
  float* red = reinterpret_cast<float*>(lds);                  // [wave][4][16 regs][64 lanes]
  const int NP = 8 / NGR;
#pragma unroll
  for (int j0 = 0; j0 < NACC; j0 += 4) {
    const int nr = NACC - j0 < 4 ? NACC - j0 : 4;              // (compile-time after unrolling)
    // the tiles / the previous round's sums are consumed (the DMA was drained at the last unit's barrier: no vmcnt wait, the previous round's
    // stores stay in flight)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      if (jj < nr) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) red[((wave * 4 + jj) * 16 + reg) * 64 + lane] = acc[j0 + jj][reg];
      }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int ntask = NGR * nr * 256;                          // float4 tasks: (group, tile of the round, 4 consecutive lanes of a register)
    for (int task = tid; task < ntask; task += 512) {
      const int e4 = task & 255, gj = task >> 8, jj = gj % nr, g = gj / nr;
      f32x4 s = (f32x4)(0.f);
      for (int k = 0; k < NP; ++k) s += *reinterpret_cast<const f32x4*>(red + (((g + NGR * k) * 4 + jj) * 1024 + e4 * 4));
      const int el = e4 * 4, reg = el >> 6, ln = el & 63;
      store(g, tap_of(g, j0 + jj), acc_row(reg, ln), ln & 31, s);
    }
  }
}

// ---- 3 x 3.  RPW dy rows per wave and unit, RU dy rows per unit: <2, 16> one chunk pair (8 pixel parts), <2, 8> two chunk pairs (4 parts),
// <4, 8> four chunk pairs (2 parts).  a.icw x a.ocw = chunks per workgroup (wave-uniform, 1 or 2 each).
template <int RPW, int RU>
DEVI void wgrad8_body3(const W6Args& a, const int bx, const int by, const int zslot) {
#if __HIP_DEVICE_COMPILE__
  constexpr int Q = 2;                                        // row slot of pixel column c under kernel column kx: c + kx + Q (3 pad slots - pad 1)
  constexpr int XROWS = RU + 2, XROWB = 35 * 64, XB = XROWS * XROWB + 3 * 64, DYB = RU * 2048, UPI = 32 / RU;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = lds_addr_of(lds);
  const int IC = a.icw, OC = a.ocw, NGR = IC * OC, NP = 8 / NGR;
  const int STAGE = IC * XB + OC * DYB;
  const int i0 = bx * 32 * IC, o0 = by * 32 * OC;
  int nstamp = 0;
  auto stamp = [&](int tag) {
    if (a.stamps && bx == 0 && by == 0 && zslot == 0 && lane == 0 && nstamp < 63) {
      a.stamps[512 + wave * 64 + nstamp] = ((unsigned long long)tag << 56) | (__builtin_amdgcn_s_memtime() & 0x00FFFFFFFFFFFFFFull);
      ++nstamp;
    }
  };
  stamp(1);
  int n0, n1;
  if (!w8_slot(a, zslot, n0, n1)) return;
  const int U = UPI * (n1 - n0);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, a.dybytes, 0x00020000);

  // ---- zero the pad slots of every x tile of both stages (the DMA only writes data slots; the right halo of a row is the left pad of the next)
  for (int i = tid; i < 2 * IC * (XROWS + 1) * 12; i += 512) {
    const int til = i / ((XROWS + 1) * 12), k = i % ((XROWS + 1) * 12);
    const int off = (til / IC) * STAGE + (til % IC) * XB + (k / 12) * XROWB + (k % 12) * 16;
    *reinterpret_cast<uint4*>(lds + off) = make_uint4(0, 0, 0, 0);
  }

  // ---- DMA: piece pc of a unit: [0, NXP) x (chunk, tile row, 16-pixel half), then dy (chunk, row, half)
  const int NXP = IC * XROWS * 2, NPC = NXP + OC * RU * 2;
  const unsigned xl = (unsigned)((lane >> 2) * a.Cin * 2 + (lane & 3) * 16), dyl = (unsigned)((lane >> 2) * a.Cout * 2 + (lane & 3) * 16);
  auto issue_piece = [&](int u, int sb, int k) {
    const int pc = wave + 8 * k;
    if (pc >= NPC) return;
    const int n = n0 + u / UPI, r0 = (u % UPI) * RU;
    if (pc < NXP) {
      const int icp = pc >= XROWS * 2 ? 1 : 0, rem = pc - icp * XROWS * 2, j = rem >> 1, half = rem & 1, row = r0 - 1 + j;
      const bool ok = (unsigned)row < 32u;
      const int so = ((n * 32 + row) * 32 + half * 16) * a.Cin * 2 + (i0 + icp * 32) * 2;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(lds + sb + icp * XB + j * XROWB + (3 + half * 16) * 64), 16, ok ? xl : 0xFFFFFFFFu, ok ? so : 0, 0, 0);
    } else {
      const int d = pc - NXP, ocp = d >= RU * 2 ? 1 : 0, rem = d - ocp * RU * 2, j = rem >> 1, half = rem & 1;
      const int so = ((n * 32 + r0 + j) * 32 + half * 16) * a.Cout * 2 + (o0 + ocp * 32) * 2;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lptr_t)(lds + sb + IC * XB + ocp * DYB + j * 2048 + half * 1024), 16, dyl, so, 0, 0);
    }
  };
  constexpr int NPIECE = 10;                                   // pieces per wave, at most: (2 x 10 x 2 + 2 x 8 x 2) / 8 = 9, (18 x 2 + 32) / 8 = 8.5

  // ---- per-lane read address (transposing reads: lane 4q + p of a 16-lane group supplies pixel q, channels 4p .. of the group's block)
  const int h = lane >> 5, cb = (lane >> 4) & 1, q = (lane & 15) >> 2, p4 = lane & 3;
  const int lbase = (8 * h + q) * 64 + cb * 32 + p4 * 8;
  const int grp = wave % NGR, pp = wave / NGR, ic = grp % IC, oc = grp / IC;
  const int xoff = ic * XB + lbase + pp * RPW * XROWB + Q * 64, dyoff = IC * XB + oc * DYB + lbase + pp * RPW * 2048;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = (f32x16)(0.f);

  stamp(2);
#pragma unroll
  for (int k = 0; k < NPIECE; ++k) issue_piece(0, 0, k);
  stamp(3);
  for (int u = 0; u < U; ++u) {
    stamp(4);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // unit u has landed; the other stage is free again
    stamp(5);
    const int sb = (u & 1) * STAGE;
    const bool more = u + 1 < U;
    const int xb = sb + xoff, dyb = sb + dyoff;
    // Six column steps c = (half hh, kernel column kx).  The fragments of step c + 1 are requested BEFORE the MFMAs of step c (two register
    // sets): left to itself the compiler puts every read right in front of its first use and a wave then sits out the LDS latency once per
    // fragment -- 28 times per unit, more than the unit's MFMA time.
    hd_s16x4 dl[2][RPW], dh[2][RPW], xl_[2][RPW + 2], xh_[2][RPW + 2];   // fragment halves in flight (asm reads: common.h)
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) lds_tr2_issue(dl[0][rr], dh[0][rr], lds0 + dyb + rr * 2048, lds0 + dyb + rr * 2048 + 256);
#pragma unroll
    for (int j = 0; j < RPW + 2; ++j) lds_tr2_issue(xl_[0][j], xh_[0][j], lds0 + xb + j * XROWB, lds0 + xb + j * XROWB + 256);
    bf16x8 fdy[RPW];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const int hh = c / 3, kx = c % 3;
      lds_tr_wait();                                           // step c's fragments (requested one step ago) are here
      bf16x8 fx[RPW + 2];
#pragma unroll
      for (int j = 0; j < RPW + 2; ++j) fx[j] = lds_tr2_take(xl_[c & 1][j], xh_[c & 1][j]);
      if (kx == 0) {
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) fdy[rr] = lds_tr2_take(dl[hh][rr], dh[hh][rr]);
      }
      if (c + 1 < 6) {
        const int h2 = (c + 1) / 3, k2 = (c + 1) % 3;
#pragma unroll
        for (int j = 0; j < RPW + 2; ++j) {
          const unsigned ad = lds0 + xb + j * XROWB + h2 * 1024 + k2 * 64;
          lds_tr2_issue(xl_[(c + 1) & 1][j], xh_[(c + 1) & 1][j], ad, ad + 256);
        }
        if (c == 1) {
#pragma unroll
          for (int rr = 0; rr < RPW; ++rr) lds_tr2_issue(dl[1][rr], dh[1][rr], lds0 + dyb + rr * 2048 + 1024, lds0 + dyb + rr * 2048 + 1024 + 256);
        }
      }
      if (more && c < 5) { issue_piece(u + 1, STAGE - sb, 2 * c); issue_piece(u + 1, STAGE - sb, 2 * c + 1); }   // (10 slots for <= 9 pieces)
#pragma unroll
      for (int j = 0; j < RPW + 2; ++j) {                      // x rows of this wave's window; row j meets dy row j - ky under kernel row ky
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int rr = j - ky;
          if (rr >= 0 && rr < RPW) acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fdy[rr], fx[j], acc[ky * 3 + kx], 0, 0, 0);
        }
      }
    }
  }
  stamp(6);
  float* Pw = a.ws + (long)zslot * a.ws_item;
  w8_reduce<9>(lds, acc, NGR, wave, lane, tid,
               [](int, int j) { return j; },
               [&](int g, int tap, int o, int i, f32x4 s) {
                 *reinterpret_cast<f32x4*>(Pw + ((long)tap * a.Cout + o0 + (g / IC) * 32 + o) * a.Cin + i0 + (g % IC) * 32 + i) = s;
               });
  stamp(7);
#endif
}

}  // namespace
