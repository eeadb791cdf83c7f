// Device body of the fused Unet_block main branch (blk6.hip has the design notes and the launch code):
//   y = alpha * convB( mid( convA(x) ) ) + beta * res        with the intermediate tile kept in LDS.
// Forward  (mode 0): convA = conv_res1, mid = dropout_p(mp_silu(u * e[n][c])), convB = conv_res2 (+ mp_sum with the residual).
// Backward (mode 1): convA = dgrad of conv_res2, mid = FiLM / mp_silu / dropout backward, convB = dgrad of conv_res1.
// The LDS images, the DMA pieces, the swizzle and the MFMA tap loop are those of conv6_body.h.
#pragma once
#include "common.h"
#include "conv6_common.h"

namespace {

struct B6Args {
  const void* x;                      // [N][H][W][Ca] bf16: input of conv A
  const void* wa;                     // [g][tap][Cm][Ca] bf16
  const void* wb;                     // [g][tap][Cb][Cm] bf16
  void* y;                            // [N][H][W][Cb]
  const void* res;                    // optional [N][H][W][Cb]
  const int* seg;
  long wa_stride, wb_stride;          // elements per group
  int N, H, W, Ca, Cm, Cb, ngroups;
  int ks[HDMOE_MAX_GROUPS], order[HDMOE_MAX_GROUPS];
  float alpha, beta;                  // epilogue of conv B
  float alpha_mid;                    // scale of conv A's accumulator
  int TH, tpi;                        // tile rows (TH * W = 256 output pixels), tiles per image
  int T;                              // taps per weight stage
  int xb_bytes, hb_plane, wb_bytes;   // one x-chunk buffer / one 32-channel plane of the intermediate / one weight stage buffer
  int nxp;                            // x pieces every wave issues per chunk (xb_bytes / 1024 / waves)
  int xbytes, wabytes, wbbytes;       // buffer-descriptor extents
  unsigned m_tpi, m_T;                // 2^32 / d + 1 reciprocals
  int mode;                           // 0: FiLM forward, 1: FiLM backward
  const float* e;                     // [N][Cm] fp32 FiLM vector 1 + emb_layer(e) * gain
  void* u;                            // [N][H][W][Cm]: mode 0 OUT conv A's output (pre-activation); mode 1 IN the saved pre-activation
  void* hmid;                         // [N][H][W][Cm]: mode 0 OUT the activation (input of conv B, saved for its weight gradient); mode 1 OUT du
  float* de;                          // mode 1: [N][Cm] += sum over pixels of d(mid)/d(e)
  const unsigned long long* seed_dev; unsigned seed_lo, seed_hi; float p;
  unsigned long long* stamps;         // development: s_memtime stamps of workgroup 0 ([wave][64] slots), or null
  int dbg;                            // development ablations: 1 no MFMA loops, 2 no DMA inside the loops, 4 no global stores, 8 no middle op
  int desync;                         // 4-wave variant: how the two workgroups of a CU are kept out of phase (0 off, 1 priority by parity, 2 priority by grid half, 3 / 4: start delay)
};

struct B6Unit { int g, ks, n, ty0, rend; };     // (everything else follows from ks: kept out of the record, scalar registers are scarce here)
struct B6Geo { int pd, ntaps, ntg, WXp, HX, HM, ppt, nblkA; };

// NW = 8: one 8-wave workgroup per CU, two x-chunk buffers.  NW = 4: 4-wave workgroups, TWO per CU (<= 80 KB of LDS each, one x-chunk
// buffer): the phases of a unit -- conv A, middle op, conv B, stores -- are serial inside a workgroup, so a lone workgroup leaves the
// matrix pipe idle during its middle op / epilogues and the VALU idle during its MFMA stages; two independent workgroups interleave.
template <int NW, int NTM, int NTB, int MODE>
DEVI void blk6_body(const B6Args& a, const int bid, const int G) {
#if __HIP_DEVICE_COMPILE__
  constexpr bool XS = NW == 4;                  // single x buffer: the next chunk / unit is fetched once every wave is done with the current one
  constexpr int NBM = 32 * NTM, NBB = 32 * NTB;
  constexpr int PPTA = NBM / 16, PPTB = NBB / 16;
  constexpr int NPW = NW == 8 ? 6 : 9;          // x pieces per wave (<= 48 / <= 36 pieces of 16 pixels per chunk)
  constexpr int NWP = 40 / NW;                  // weight pieces per wave per stage (T * NB / 16 <= 40)
  constexpr int MBA = NW == 8 ? 2 : (NTM == 2 ? 2 : 3);   // pixel blocks per wave: conv A over <= 16 (NW = 8) / <= 12 or <= 8 (NW = 4) blocks ...
  constexpr int MBB = 8 / NW;                   // ... conv B over the tile's 8 blocks (4-wave variant: 2 each, or 1 each on a 128-pixel tile)
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int csl = ((lane & 3) ^ ((lane >> 4) & 3)) << 4;
  const int prow = lane >> 2;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwa = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wa), 0, a.wabytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wb), 0, a.wbbytes, 0x00020000);

  uint32_t seed_lo = a.seed_lo, seed_hi = a.seed_hi;
  if (a.p > 0.f) mix_seed(seed_lo, seed_hi, a.seed_dev);
  const float drop_inv = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;

  int nstamp = 0;
  auto stamp = [&](int tag) {
    if (a.stamps && bid == 0 && lane == 0 && nstamp < 63) {
      a.stamps[wave * 64 + nstamp] = ((unsigned long long)tag << 56) | (__builtin_amdgcn_s_memtime() & 0x00FFFFFFFFFFFFFFull);
      ++nstamp;
    }
  };
  stamp(1);
  if (NW == 4) {
    // Two workgroups share a CU.  Left alone they run in lockstep -- same program, same unit sizes: both in their MFMA stages, then both
    // in their middle ops -- and nothing overlaps.  A priority difference breaks the tie: the favoured workgroup takes the matrix pipe,
    // the other one falls half a unit behind and from then on computes while its partner stores, and vice versa.
    const bool second = (a.desync == 1 || a.desync == 3) ? (bid & 1) : (bid >= (G >> 1));
    if ((a.desync == 1 || a.desync == 2) && second) __builtin_amdgcn_s_setprio(1);
    if ((a.desync == 3 || a.desync == 4) && second) { for (int i = 0; i < 64; ++i) __builtin_amdgcn_s_sleep(100); }
  }
  // ---- unit list (as conv6_body.h): groups in descending kernel size; a unit = one TH x W tile of one routed row
  const int oi_l = lane & 7;
  int v_g = 0, v_ks = 0;
#pragma unroll
  for (int oi = 0; oi < HDMOE_MAX_GROUPS; ++oi) v_g = (oi_l == oi) ? a.order[oi] : v_g;
#pragma unroll
  for (int g = 0; g < HDMOE_MAX_GROUPS; ++g) v_ks = (v_g == g) ? a.ks[g] : v_ks;
  const bool slot_ok = lane < a.ngroups;
  const int v_row0 = (a.seg && slot_ok) ? a.seg[v_g] : 0;
  const int v_rows = !slot_ok ? 0 : (a.seg ? a.seg[v_g + 1] - v_row0 : a.N);
  const int v_units = v_rows * a.tpi;
  int v_ustart = v_units;
#pragma unroll
  for (int d = 1; d < 8; d <<= 1) {
    const int o = __shfl_up(v_ustart, d, 8);
    if (oi_l >= d) v_ustart += o;
  }
  const int total = __builtin_amdgcn_readlane(v_ustart, 7);
  v_ustart -= v_units;
  auto udiv = [](int x, unsigned magic, int d) {
    int q = (int)(((unsigned long long)(unsigned)x * magic) >> 32);
    if (q * d > x) --q;
    if ((q + 1) * d <= x) ++q;
    return q;
  };
  auto decode = [&](int j, B6Unit& u) {
    const unsigned long long hit = __ballot(lane < 8 && j >= v_ustart && j < v_ustart + v_units);
    const int slot = (int)__builtin_ctzll(hit | (1ull << 7));
    const int uu = j - __builtin_amdgcn_readlane(v_ustart, slot);
    const int row0 = __builtin_amdgcn_readlane(v_row0, slot);
    u.rend = row0 + __builtin_amdgcn_readlane(v_rows, slot);
    u.g = __builtin_amdgcn_readlane(v_g, slot); u.ks = __builtin_amdgcn_readlane(v_ks, slot);
    const int img = udiv(uu, a.m_tpi, a.tpi);
    u.n = row0 + img; u.ty0 = (uu - img * a.tpi) * a.TH;
  };
  auto geo_of = [&](int ks) {
    B6Geo q;
    q.pd = (ks - 1) >> 1; q.ntaps = ks * ks; q.ntg = udiv(q.ntaps + a.T - 1, a.m_T, a.T);
    q.WXp = a.W + ks - 1; q.HX = a.TH + 2 * (ks - 1); q.HM = a.TH + ks - 1;
    q.ppt = (q.WXp * q.HX + 15) >> 4;
    q.nblkA = (q.HM * a.W) >> 5;
    return q;
  };
  auto ppt_of = [&](int ks) { return ((a.W + ks - 1) * (a.TH + 2 * (ks - 1)) + 15) >> 4; };

  // ---- x pieces (16 pixels x 32 channels of one chunk) of the unit's (HX x WXp) input region; ~0 = padding (the DMA writes zeros)
  const int ca2 = a.Ca * 2;
  auto plan_piece = [&](const B6Unit& u, int k) -> unsigned {      // (only when a workgroup starts, or its next unit belongs to another expert)
    const int WXp = a.W + u.ks - 1, HX = a.TH + 2 * (u.ks - 1), ppt = (WXp * HX + 15) >> 4;
    const int pi = wave + NW * k;
    const int px = 16 * pi + prow;
    const int magic = (1 << 20) / WXp + 1;
    int hy = (int)(((unsigned)px * (unsigned)magic) >> 20);
    if (hy * WXp > px) --hy;
    const int hx = px - hy * WXp;
    const int pd = (u.ks - 1) >> 1;
    const int iy = u.ty0 - 2 * pd + hy, ix = hx - pd;
    const bool ok = pi < ppt && px < WXp * HX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    return ok ? (unsigned)(((u.n * a.H + iy) * a.W + ix) * ca2 + csl) : 0xFFFFFFFFu;
  };
  auto issue_xpiece = [&](unsigned off, int k, int c, int xbo) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(lds + xbo + (wave + NW * k) * 1024), 16, off, c * 64, 0, 0);
  };
  auto xpieces = [&](int ppt) { return (ppt + NW - 1) / NW; };
  // ---- weight stages: [tap][NB rows][64 B]; piece pi = wave + 8 k is tap pi / PPT, rows 16 (pi % PPT) ..
  const unsigned wloA = (unsigned)((((wave / PPTA) * a.Cm + (wave % PPTA) * 16 + prow) * a.Ca) * 2 + csl);
  const int wkA = (NW / PPTA) * a.Cm * a.Ca * 2;
  const unsigned wloB = (unsigned)((((wave / PPTB) * a.Cb + (wave % PPTB) * 16 + prow) * a.Cm) * 2 + csl);
  const int wkB = (NW / PPTB) * a.Cb * a.Cm * 2;
  auto wpieces = [&](int ntl, int ppt) { const int per = NW / ppt; return max(0, (ntl - wave / ppt + per - 1) / per); };
  auto baseA = [&](const B6Unit& u, int c, int t0) { return (int)(((long)u.g * a.wa_stride + (long)t0 * a.Cm * a.Ca + c * 32) * 2); };
  auto baseB = [&](const B6Unit& u, int nb, int c, int t0) {
    return (int)(((long)u.g * a.wb_stride + ((long)t0 * a.Cb + nb * NBB) * a.Cm + c * 32) * 2);
  };
  // a weight stage still to be fetched: which image, its first byte, this wave's share of it
  struct WNext { int isB, sb, np; };
  auto issue_wstage = [&](const WNext& w, int wbo) {            // this wave's pieces of one stage, back to back
    if (a.dbg & 2) return;
    if (w.isB) {
#pragma unroll
      for (int k = 0; k < NWP; ++k)
        if (k < w.np) __builtin_amdgcn_raw_ptr_buffer_load_lds(rwb, (lptr_t)(lds + wbo + (wave + NW * k) * 1024), 16, wloB, w.sb + k * wkB, 0, 0);
    } else {
#pragma unroll
      for (int k = 0; k < NWP; ++k)
        if (k < w.np) __builtin_amdgcn_raw_ptr_buffer_load_lds(rwa, (lptr_t)(lds + wbo + (wave + NW * k) * 1024), 16, wloA, w.sb + k * wkA, 0, 0);
    }
  };

  // Every LDS-DMA piece is written by ONE wave and read by all of them after the next barrier.  The compiler orders a wave's own LDS reads
  // behind its own DMA, but puts no vmcnt wait in front of a barrier whose successor code has none of them (seen in the ISA of the conv B
  // loop: `s_waitcnt lgkmcnt(0); s_barrier`; with 2-tap stages the pieces then land after another wave has read them): wait explicitly.
  // vm = false: the first stage behind an epilogue.  Its DMA pieces were issued before the epilogue and every wave waited for its own
  // (drain_dma) before issuing the epilogue's global stores, so the barrier must not wait for those stores (vmcnt counts them too, in order).
  // (a raw s_barrier: __syncthreads() carries a workgroup-scope release fence, i.e. an s_waitcnt vmcnt(0) of its own for the global stores)
  auto stage_barrier = [&](bool vm = true) {
    if (vm) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  auto drain_dma = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
  int j = bid;
  if (j >= total) return;
  B6Unit cur, nu;
  decode(j, cur);
  nu = cur;
  unsigned hoc[NPW];                                            // this wave's DMA source offsets of the unit whose x chunks are (still) to be fetched
  int jn = j + G;
  bool has_next = jn < total;

  const int XB0 = 0, HB0 = (XS ? 1 : 2) * a.xb_bytes, WB0 = HB0 + NTM * a.hb_plane, EB0 = WB0 + 2 * a.wb_bytes;     // EB: this unit's FiLM vector e[n][0 .. Cm)
  const int nchA = a.Ca >> 5;
  const int nblkB = a.Cb / NBB;
  const int wl = r * 64 + ((h << 4) ^ (((r >> 2) & 3) << 4));
  const int tws = a.W == 32 ? 5 : 4;
  const int nblkT = (a.TH * a.W) >> 5;                          // 32-pixel blocks of the output tile (8, or 4)

  // zero the intermediate image once (its padding columns are never written; re-done when the kernel size -- the image's row pitch -- changes)
  auto zero_hb = [&]() {
    const int n16 = (NTM * a.hb_plane) >> 4;
    for (int i = tid; i < n16; i += 64 * NW) *reinterpret_cast<uint4*>(lds + HB0 + (i << 4)) = make_uint4(0u, 0u, 0u, 0u);
  };
  zero_hb();
  int hb_ks = cur.ks;
  B6Geo cg = geo_of(cur.ks);
  // this wave's x pieces of one chunk: always a.nxp of them (pieces past the region are zero fills inside the buffer), so that the count of
  // DMA instructions in flight behind a point is known for counted waits
  auto issue_xchunk = [&](const unsigned (&off)[NPW], int c, int xbo) {
#pragma unroll
    for (int k = 0; k < NPW; ++k) if (k < a.nxp) issue_xpiece(off[k], k, c, xbo);
  };
  // prologue: first weight stage of conv A, first x chunk
  {
    WNext w0{0, baseA(cur, 0, 0), wpieces(min(a.T, cg.ntaps), PPTA)};
    issue_wstage(w0, WB0);
#pragma unroll
    for (int k = 0; k < NPW; ++k) hoc[k] = plan_piece(cur, k);
    issue_xchunk(hoc, 0, XB0);
  }
  int par = 0, sp = 0;

  // ---- one weight stage of MFMAs: MB pixel blocks x NT 32-channel blocks, operands from the pixel image at `bufpx` and the stage buffer `wbuf`
  // (the tap cursor goes in and out BY VALUE, packed ky << 8 | kx: as reference parameters of this generic lambda the two counters were
  //  kept in scratch memory and every tap stored them back -- VMEM traffic inside the MFMA loop that the stage barriers then waited for)
  auto mma_stage = [&](auto MBt, auto NTt, f32x16 (&acc)[3][2], const int (&P0)[3], int bufpx, int HWp, int ks, int cursor,
                       const unsigned char* wbuf, int ntl, int bt, auto&& burst) -> int {
    constexpr int MB = decltype(MBt)::value, NT = decltype(NTt)::value, NB = 32 * NT;
    // Fragment lookahead in taps.  A wave with one or two MFMAs per k-step (1 x 1, 2 x 1, 1 x 2 tiles) issues a tap's MFMAs in 64-128
    // cycles but waits ~250 for the fragments of the next one: two taps ahead (three register sets) the LDS latency is covered; the 2 x 2
    // tile (8 MFMAs per tap, no registers to spare) relies on its SIMD partner instead.
    constexpr int D = (MB * NT >= 4) ? 0 : ((NW == 8 && NTM == 1) ? 2 : 1);   // (registers: the 64-channel and the 4-wave variants get one tap)
    if (a.dbg & 1) { burst(); return cursor; }
    int ky = cursor >> 8, kx = cursor & 255;
    bf16x8 fx[D + 1][2][MB], fw[D + 1][2][NT];
    auto load_tap = [&](bf16x8 (&gx)[2][MB], bf16x8 (&gw)[2][NT], int tl) {
      const int toff = ky * HWp + kx + bufpx;
      const unsigned char* wt = wbuf + tl * NB * 64;
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const int px = P0[m] + toff;
        const int ad = (px << 6) + (((px << 2) & 0x30) ^ (h << 4));
        gx[0][m] = *reinterpret_cast<const bf16x8*>(lds + ad);
        gx[1][m] = *reinterpret_cast<const bf16x8*>(lds + (ad ^ 32));
      }
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        gw[0][b] = *reinterpret_cast<const bf16x8*>(wt + b * 2048 + wl);
        gw[1][b] = *reinterpret_cast<const bf16x8*>(wt + b * 2048 + (wl ^ 32));
      }
      if (++kx == ks) { kx = 0; ++ky; }
    };
    // (the tap cursor may run up to D taps past the stage's last one: such fragments are read inside LDS but never used)
#pragma unroll
    for (int d = 0; d < D; ++d) load_tap(fx[d], fw[d], min(d, ntl - 1));
#pragma unroll
    for (int tl = 0; tl < C6_MAXT; ++tl) {
      if (tl < ntl) {
        __builtin_amdgcn_sched_barrier(0);
        constexpr int NM = 2 * MB * NT, NR = 2 * (MB + NT);
        load_tap(fx[(tl + D) % (D + 1)], fw[(tl + D) % (D + 1)], min(tl + D, ntl - 1));
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int b = 0; b < NT; ++b)
              acc[m][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[tl % (D + 1)][s2][b], fx[tl % (D + 1)][s2][m], acc[m][b], 0, 0, 0);
        if constexpr (D > 0) {
#pragma unroll
          for (int i = 0; i < NM; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < NR - NM) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            else if (i < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x004, 1, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (tl == bt) burst();
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) { if (kx == 0) { kx = ks - 1; --ky; } else --kx; }     // undo the cursor's run-ahead
    return (ky << 8) | kx;
  };
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  // s_waitcnt vmcnt(n) for a run-time n <= 12: everything but the n youngest VMEM operations of this wave has completed
  auto wait_vm_all_but = [&](int n) {
    switch (n) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
      case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
      case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    }
  };

  bool first_unit = true;
  float e_reg = (wave == 0 && lane < a.Cm) ? a.e[(long)cur.n * a.Cm + lane] : 0.f;
  while (true) {
    stamp(2);
    // this unit's FiLM vector -> LDS (read in the middle op; visible behind the conv A stage barriers); fetched one unit ahead
    if (wave == 0 && lane < a.Cm) reinterpret_cast<float*>(lds + EB0)[lane] = e_reg;
    if (cur.ks != hb_ks) {                                      // (wave-uniform; every wave is past its reads of the old image: the previous unit ended on a barrier)
      stage_barrier();
      zero_hb();
      hb_ks = cur.ks;
    }
    // =========================================== conv A over the (HM x W) region the second conv needs ===========================================
    // this wave's blocks: wave, wave + NW, ... (32 pixels each; rows of 32, or pairs of rows of 16)
    const int nvA = min(MBA, (cg.nblkA - wave + NW - 1) / NW);
    int P0A[3], mrA[3], mcA[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int blk = min(wave + NW * m, cg.nblkA - 1);
      const int q = blk * 32 + r;
      mrA[m] = q >> tws; mcA[m] = q & (a.W - 1);
      P0A[m] = mrA[m] * cg.WXp + mcA[m];
    }
    f32x16 acc[3][2];
#pragma unroll
    for (int m = 0; m < MBA; ++m)
#pragma unroll
      for (int b = 0; b < NTM; ++b) acc[m][b] = (f32x16)(0.f);

    for (int c = 0; c < nchA; ++c) {
      const bool last_chunk = c == nchA - 1;
      int hmode = 0;
      if (!last_chunk) hmode = 1;
      else if (has_next) {
        // Units are dealt round robin (j, j + G, ...): with G a multiple of the tiles per image the successor is the SAME tile of row
        // n + G / tpi -- same expert, kernel size and padding pattern unless that row belongs to the next expert: no decode, the DMA
        // source offsets just move by whole images.
        const int dn = udiv(G, a.m_tpi, a.tpi);
        if (dn * a.tpi == G && cur.n + dn < cur.rend) {
          nu = cur; nu.n = cur.n + dn;
          const unsigned step_b = (unsigned)(dn * a.H * a.W * ca2);
#pragma unroll
          for (int k = 0; k < NPW; ++k) hoc[k] = hoc[k] == 0xFFFFFFFFu ? 0xFFFFFFFFu : hoc[k] + step_b;
        } else {
          decode(jn, nu);
#pragma unroll
          for (int k = 0; k < NPW; ++k) hoc[k] = plan_piece(nu, k);
        }                                                       // (the current unit's last chunk is in LDS or on its way: hoc now describes the successor)
        hmode = 2;
        if (wave == 0 && lane < a.Cm) e_reg = a.e[(long)nu.n * a.Cm + lane];
      }
      const int xbn = XS ? XB0 : XB0 + (par ^ 1) * a.xb_bytes;
      int cursor = 0;
      for (int tg = 0; tg < cg.ntg; ++tg) {
        const int t0 = tg * a.T;
        const int ntl = min(a.T, cg.ntaps - t0);
        stamp(3);
        stage_barrier(first_unit || c > 0 || tg > 0);
        stamp(4);
        const int wbn = WB0 + (sp ^ 1) * a.wb_bytes;
        WNext wn{0, 0, 0};
        if (tg + 1 < cg.ntg) wn = WNext{0, baseA(cur, c, t0 + a.T), wpieces(min(a.T, cg.ntaps - t0 - a.T), PPTA)};
        else if (!last_chunk) wn = WNext{0, baseA(cur, c + 1, 0), wpieces(min(a.T, cg.ntaps), PPTA)};
        else wn = WNext{1, baseB(cur, 0, 0, 0), wpieces(min(a.T, cg.ntaps), PPTB)};
        const bool xnow = !XS && tg == 0 && hmode != 0;         // two x buffers: the next chunk / unit rides on this chunk's first stage
        auto burst = [&]() {                                    // the next stage's weights, the next chunk's / unit's x pieces
          issue_wstage(wn, wbn);
          if ((a.dbg & 2) || !xnow) return;
          issue_xchunk(hoc, hmode == 2 ? 0 : c + 1, xbn);
        };
        const int bt = min(NW == 8 ? wave >> 2 : wave & 1, ntl - 1);     // (SIMD partners / alternate waves one tap apart)
        const int bufpx = (XS ? XB0 : XB0 + par * a.xb_bytes) >> 6;
        const unsigned char* wbuf = lds + WB0 + sp * a.wb_bytes;
        if (MBA >= 3 && nvA >= 3) cursor = mma_stage(I3{}, std::integral_constant<int, NTM>{}, acc, P0A, bufpx, cg.WXp, cur.ks, cursor, wbuf, ntl, bt, burst);
        else if (nvA == 2) cursor = mma_stage(I2{}, std::integral_constant<int, NTM>{}, acc, P0A, bufpx, cg.WXp, cur.ks, cursor, wbuf, ntl, bt, burst);
        else cursor = mma_stage(I1{}, std::integral_constant<int, NTM>{}, acc, P0A, bufpx, cg.WXp, cur.ks, cursor, wbuf, ntl, bt, burst);
        sp ^= 1;
      }
      if (XS) {
        // one x buffer: every wave is done with the chunk -> fetch the next one (this unit's, exposed: the other workgroup of the CU covers
        // it; or the next unit's, which lands under the middle op and conv B)
        stage_barrier(false);
        if (hmode != 0 && !(a.dbg & 2)) issue_xchunk(hoc, hmode == 2 ? 0 : c + 1, XB0);
      }
      par ^= 1;
    }
    stamp(5);
    // conv B's first weight stage (issued beside the last stage above) has landed; with one x buffer the next unit's a.nxp x pieces were
    // issued after it and stay in flight
    if (XS && has_next && !(a.dbg & 2)) wait_vm_all_but(a.nxp); else drain_dma();
    stamp(6);
    // ---- middle op on conv A's accumulators -> intermediate image in LDS (every position of the region: activation, or 0 outside the image)
    {
      typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
      typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
      bf16* U = (bf16*)a.u;
      bf16* HM = (bf16*)a.hmid;
      const int WMp = cg.WXp;
#pragma unroll
      for (int m = 0; m < MBA; ++m) {
        if (m < nvA && !(a.dbg & 8)) {
          const int mr = mrA[m], mc = mcA[m];
          const int iy = cur.ty0 - cg.pd + mr;
          const bool inimg = (unsigned)iy < (unsigned)a.H;
          const bool owned = mr >= cg.pd && mr < cg.pd + a.TH;
          const long pix = (((long)cur.n * a.H + iy) * a.W + mc) * a.Cm;
          const int hpx = mr * WMp + mc + cg.pd;
          const int hsw = (hpx >> 2) & 3;
#pragma unroll
          for (int b = 0; b < NTM; ++b)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
              float v[8];
#pragma unroll
              for (int q = 0; q < 8; ++q) v[q] = a.alpha_mid * acc[m][b][8 * p + q];
              const unsigned A0 = __builtin_bit_cast(unsigned, (bf2){(bf16)v[0], (bf16)v[1]}), A1 = __builtin_bit_cast(unsigned, (bf2){(bf16)v[2], (bf16)v[3]});
              const unsigned B0 = __builtin_bit_cast(unsigned, (bf2){(bf16)v[4], (bf16)v[5]}), B1 = __builtin_bit_cast(unsigned, (bf2){(bf16)v[6], (bf16)v[7]});
              const u32x2 s0 = __builtin_amdgcn_permlane32_swap(A0, B0, false, false);
              const u32x2 s1 = __builtin_amdgcn_permlane32_swap(A1, B1, false, false);
              // this lane: channels c0 .. c0 + 7 of pixel (mr, mc), rounded to bf16 (as the unfused conv writes them)
              const int c0 = 32 * b + 16 * p + 8 * h;
              const unsigned pk[4] = {s0[0], s1[0], s0[1], s1[1]};
              unsigned ho[4] = {0u, 0u, 0u, 0u};
              if (inimg) {
                const long eo = pix + c0;
                const float* ep = reinterpret_cast<const float*>(lds + EB0) + c0;
                uint32_t r4[8];
                if (a.p > 0.f) {
                  const long q0 = eo >> 2;
                  philox((uint32_t)q0, (uint32_t)(q0 >> 32), seed_lo, seed_hi, r4);
                  philox((uint32_t)(q0 + 1), (uint32_t)((q0 + 1) >> 32), seed_lo, seed_hi, r4 + 4);
                }
                if constexpr (MODE == 0) {
                  if (owned && U && !(a.dbg & 4)) *reinterpret_cast<uint4*>(U + eo) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
#pragma unroll
                  for (int j2 = 0; j2 < 4; ++j2) {
                    const bf2 yv = __builtin_bit_cast(bf2, pk[j2]);
                    float f0 = mp_silu_f((float)yv[0] * ep[2 * j2]);
                    float f1 = mp_silu_f((float)yv[1] * ep[2 * j2 + 1]);
                    if (a.p > 0.f) {
                      f0 = u01(r4[2 * j2]) >= a.p ? f0 * drop_inv : 0.f;
                      f1 = u01(r4[2 * j2 + 1]) >= a.p ? f1 * drop_inv : 0.f;
                    }
                    ho[j2] = __builtin_bit_cast(unsigned, (bf2){(bf16)f0, (bf16)f1});
                  }
                  if (owned && HM && !(a.dbg & 4)) *reinterpret_cast<uint4*>(HM + eo) = make_uint4(ho[0], ho[1], ho[2], ho[3]);
                } else {
                  // FiLM / mp_silu / dropout backward on the bf16-rounded d(activation): du = g * silu'(u e) * e, de += g * silu'(u e) * u
                  const uint4 uq = *reinterpret_cast<const uint4*>(U + eo);
                  const unsigned uk[4] = {uq.x, uq.y, uq.z, uq.w};
                  float dsum[8];
#pragma unroll
                  for (int j2 = 0; j2 < 4; ++j2) {
                    const bf2 gv = __builtin_bit_cast(bf2, pk[j2]);
                    const bf2 uv = __builtin_bit_cast(bf2, uk[j2]);
                    float g0 = (float)gv[0], g1 = (float)gv[1];
                    if (a.p > 0.f) {
                      g0 = u01(r4[2 * j2]) >= a.p ? g0 * drop_inv : 0.f;
                      g1 = u01(r4[2 * j2 + 1]) >= a.p ? g1 * drop_inv : 0.f;
                    }
                    const float u0 = (float)uv[0], u1 = (float)uv[1];
                    g0 *= mp_silu_grad_f(u0 * ep[2 * j2]); g1 *= mp_silu_grad_f(u1 * ep[2 * j2 + 1]);
                    dsum[2 * j2] = g0 * u0; dsum[2 * j2 + 1] = g1 * u1;
                    g0 *= ep[2 * j2]; g1 *= ep[2 * j2 + 1];
                    ho[j2] = __builtin_bit_cast(unsigned, (bf2){(bf16)g0, (bf16)g1});
                  }
                  if (owned) {
                    *reinterpret_cast<uint4*>(HM + eo) = make_uint4(ho[0], ho[1], ho[2], ho[3]);
                    if (a.de) {
                      // the 32 lanes of a half-wave hold the same 8 channels of 32 different pixels
#pragma unroll
                      for (int q = 0; q < 8; ++q) {
                        float s = dsum[q];
#pragma unroll
                        for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
                        if (r == 0) atomicAdd(a.de + (long)cur.n * a.Cm + c0 + q, s);
                      }
                    }
                  }
                }
              }
              *reinterpret_cast<uint4*>(lds + HB0 + b * a.hb_plane + (hpx << 6) + ((((16 * p + 8 * h) >> 3) ^ hsw) << 4)) = make_uint4(ho[0], ho[1], ho[2], ho[3]);
            }
        }
      }
    }
    stamp(7);
    // =========================================== conv B over the tile, intermediate read from LDS ===========================================
    {
      // this wave's blocks of the output tile: wave, wave + NW (4-wave variant on a 256-pixel tile)
      const int nvB = min(MBB, (nblkT - wave + NW - 1) / NW);
      int P0B[3], orowB[MBB], ocB[MBB];
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const int q = min(wave + NW * m, nblkT - 1) * 32 + r;
        if (m < MBB) { orowB[m] = q >> tws; ocB[m] = q & (a.W - 1); }
        P0B[m] = (q >> tws) * cg.WXp + (q & (a.W - 1));
      }
      for (int nb = 0; nb < nblkB; ++nb) {
        f32x16 accb[3][2];
#pragma unroll
        for (int m = 0; m < MBB; ++m)
#pragma unroll
          for (int b = 0; b < NTB; ++b) accb[m][b] = (f32x16)(0.f);
        // the residual's quads for the epilogue: loaded now, they arrive while the taps below run
        const bf16* R = (const bf16*)a.res;
        long pixB[MBB];
#pragma unroll
        for (int m = 0; m < MBB; ++m) pixB[m] = (((long)cur.n * a.H + cur.ty0 + orowB[m]) * a.W + ocB[m]) * a.Cb + nb * NBB;
        constexpr bool RPF = NTB == 1;                        // (the 64-channel variant has no registers left for it: it loads in the epilogue)
        bf16x4 rq[MBB][NTB][2][2];
        if (RPF && R) {
#pragma unroll
          for (int m = 0; m < MBB; ++m)
#pragma unroll
            for (int b = 0; b < NTB; ++b)
#pragma unroll
              for (int p = 0; p < 2; ++p) {
                const long o0 = pixB[m] + 32 * b + 16 * p + 4 * h;
                rq[m][b][p][0] = *reinterpret_cast<const bf16x4*>(R + o0);
                rq[m][b][p][1] = *reinterpret_cast<const bf16x4*>(R + o0 + 8);
              }
        }
        for (int c = 0; c < NTM; ++c) {
          int cursor = 0;
          for (int tg = 0; tg < cg.ntg; ++tg) {
            const int t0 = tg * a.T;
            const int ntl = min(a.T, cg.ntaps - t0);
            stamp(8);
            stage_barrier(c > 0 || tg > 0);
            stamp(9);
            const int wbn = WB0 + (sp ^ 1) * a.wb_bytes;
            WNext wn{0, 0, 0};
            if (tg + 1 < cg.ntg) wn = WNext{1, baseB(cur, nb, c, t0 + a.T), wpieces(min(a.T, cg.ntaps - t0 - a.T), PPTB)};
            else if (c + 1 < NTM) wn = WNext{1, baseB(cur, nb, c + 1, 0), wpieces(min(a.T, cg.ntaps), PPTB)};
            else if (nb + 1 < nblkB) wn = WNext{1, baseB(cur, nb + 1, 0, 0), wpieces(min(a.T, cg.ntaps), PPTB)};
            else if (has_next) wn = WNext{0, baseA(nu, 0, 0), wpieces(min(a.T, nu.ks * nu.ks), PPTA)};
            auto burst = [&]() { issue_wstage(wn, wbn); };
            const int bt = min(NW == 8 ? wave >> 2 : wave & 1, ntl - 1);
            const int bufpx = (HB0 + c * a.hb_plane) >> 6;
            const unsigned char* wbuf = lds + WB0 + sp * a.wb_bytes;
            if (MBB == 2 && nvB == 2) cursor = mma_stage(I2{}, std::integral_constant<int, NTB>{}, accb, P0B, bufpx, cg.WXp, cur.ks, cursor, wbuf, ntl, bt, burst);
            else cursor = mma_stage(I1{}, std::integral_constant<int, NTB>{}, accb, P0B, bufpx, cg.WXp, cur.ks, cursor, wbuf, ntl, bt, burst);
            sp ^= 1;
          }
        }
        stamp(10);
        drain_dma();                                          // the next stage's weights (and, long since, the next unit's x chunk) have landed
        stamp(11);
        // epilogue: y = alpha * acc + beta * res (fp32, one rounding), 16-byte stores
        bf16* Y = (bf16*)a.y;
#pragma unroll
        for (int m = 0; m < MBB; ++m) {
          if (m < nvB) {
#pragma unroll
            for (int b = 0; b < NTB; ++b)
#pragma unroll
              for (int p = 0; p < 2; ++p) {
                float v[8];
#pragma unroll
                for (int q2 = 0; q2 < 8; ++q2) v[q2] = a.alpha * accb[m][b][8 * p + q2];
                if (R) {
                  if (!RPF) {
                    const long o0 = pixB[m] + 32 * b + 16 * p + 4 * h;
                    rq[m][b][p][0] = *reinterpret_cast<const bf16x4*>(R + o0);
                    rq[m][b][p][1] = *reinterpret_cast<const bf16x4*>(R + o0 + 8);
                  }
#pragma unroll
                  for (int q2 = 0; q2 < 4; ++q2) { v[q2] += a.beta * (float)rq[m][b][p][0][q2]; v[4 + q2] += a.beta * (float)rq[m][b][p][1][q2]; }
                }
                typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                const unsigned A0 = __builtin_bit_cast(unsigned, (bf2){(bf16)v[0], (bf16)v[1]}), A1 = __builtin_bit_cast(unsigned, (bf2){(bf16)v[2], (bf16)v[3]});
                const unsigned B0 = __builtin_bit_cast(unsigned, (bf2){(bf16)v[4], (bf16)v[5]}), B1 = __builtin_bit_cast(unsigned, (bf2){(bf16)v[6], (bf16)v[7]});
                const u32x2 s0 = __builtin_amdgcn_permlane32_swap(A0, B0, false, false);
                const u32x2 s1 = __builtin_amdgcn_permlane32_swap(A1, B1, false, false);
                if (!(a.dbg & 4)) *reinterpret_cast<uint4*>(Y + pixB[m] + 32 * b + 16 * p + 8 * h) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
              }
          }
        }
      }
    }
    stamp(12);
    if (!has_next) break;
    first_unit = false;
    cur = nu;
    if (cg.pd != ((cur.ks - 1) >> 1)) cg = geo_of(cur.ks);
    jn += G;
    has_next = jn < total;
  }
#endif
}

}  // namespace
