// Device body of the split-bf16 conv kernel (see conv6s.hip for the design notes): shared by the stand-alone kernel and by the fused
// backward launch of bwd6.hip.
#pragma once
#include "common.h"
#include "conv6_common.h"

struct C6SArgs {
  C6Args c;
  int wplane;                         // bytes between the hi and the lo weight image
  const float* in_scale; const float* in_shift; int in_relu;   // fused input transform (or null)
  float* stats;                       // [N][tpi * nblk * 4][2] fp32 partial (sum, sum of squares) of the outputs per sample, or null
  int nprod;                          // 3: split-bf16 (fp32-equivalent); 1: only x_hi * w_hi = bf16 operands, fp32 accumulation
};
struct C6SPlan { C6SArgs sa; int NT; unsigned G; size_t lds; };
struct ConvFuse;
// Launch geometry of the split-bf16 conv for one layer (conv6s.hip).  0 = planned, 1 = outside its domain.
int conv6s_plan(const ConvArgs& c, long wplane_elems, const ConvFuse* fuse, C6SPlan& plan, bool p3 = false);

namespace {

// P3: the three products of a split-bf16 step share ONE pass over the weight stages -- a stage holds the hi AND the lo weight image of
// its taps, a k-step loads x_hi, x_lo, w_hi, w_lo once and issues w_hi x_hi + w_lo x_hi + w_hi x_lo back to back: 8 fragment reads per
// 12 MFMAs instead of 12, a third of the stage barriers (round 3; the forward trunk convs).  Without P3 the products are separate
// passes (sa.nprod of them: the bf16-operand backward runs ONE).
template <int NT, bool P3 = false>
DEVI void conv6s_body(const C6SArgs& sa, const int bid, const int G) {
#if __HIP_DEVICE_COMPILE__
  const C6Args& a = sa.c;
  constexpr int MT = 2, NB = 32 * NT, PPT = NB / 16, NW = C6_NW, MB = 8 * MT / NW;
  constexpr int NHP = 11;                      // 8-pixel half pieces per wave: 2 tiles x 22 pieces x 2 / 8 waves (3x3 halo of a 256-pixel tile)
  constexpr int NWP = 40 / NW;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int csl = ((lane & 3) ^ ((lane >> 4) & 3)) << 4;     // weight DMA: swizzled 16-B slot of this lane
  const int prow = lane >> 2;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.wbytes, 0x00020000);
  // y through a buffer descriptor too: a store whose pixel is outside the image gets an out-of-range offset and is dropped by the
  // hardware, so every wave issues the SAME number of store instructions per unit -- which lets the next unit's first barrier wait
  // with a counted vmcnt for the weight DMA issued BEFORE those stores instead of draining them (ybytes == 0: plain stores, full drain)
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.ybytes, 0x00020000);
  int pending = 0;                                                   // store instructions of the previous unit's epilogue still in flight

  // ---- unit list (see conv6.hip): slot oi of the descending-kernel-size group list lives in lane oi
  const int oi_l = lane & 7;
  int v_g = 0, v_ks = 0, v_pt = 0, v_pl = 0;
#pragma unroll
  for (int oi = 0; oi < HDMOE_MAX_GROUPS; ++oi) v_g = oi_l == oi ? a.order[oi] : v_g;
#pragma unroll
  for (int g = 0; g < HDMOE_MAX_GROUPS; ++g) {
    const bool me = v_g == g;
    v_ks = me ? a.ks[g] : v_ks; v_pt = me ? a.pt[g] : v_pt; v_pl = me ? a.pl[g] : v_pl;
  }
  const bool slot_ok = lane < a.ngroups;
  const int v_row0 = (a.seg && slot_ok) ? a.seg[v_g] : 0;
  const int v_rows = !slot_ok ? 0 : (a.seg ? a.seg[v_g + 1] - v_row0 : a.N);
  const int v_tiles = v_rows * a.tpi;
  const int v_units = (v_tiles + MT - 1) / MT;
  int v_ustart = v_units;
#pragma unroll
  for (int d = 1; d < 8; d <<= 1) {
    const int o = __shfl_up(v_ustart, d, 8);
    if (oi_l >= d) v_ustart += o;
  }
  const int total = __builtin_amdgcn_readlane(v_ustart, 7) * a.nblk;
  v_ustart -= v_units;
  auto udiv = [](int x, unsigned magic, int d) {
    int q = (int)(((unsigned long long)(unsigned)x * magic) >> 32);
    if (q * d > x) --q;
    if ((q + 1) * d <= x) ++q;
    return q;
  };
  auto decode = [&](int j, C6Unit<MT>& u) {
    const int uu0 = udiv(j, a.m_nblk, a.nblk);
    u.nbk = j - uu0 * a.nblk;
    const unsigned long long hit = __ballot(lane < 8 && uu0 >= v_ustart && uu0 < v_ustart + v_units);
    const int slot = (int)__builtin_ctzll(hit | (1ull << 7));
    const int uu = uu0 - __builtin_amdgcn_readlane(v_ustart, slot);
    const int row0 = __builtin_amdgcn_readlane(v_row0, slot), tiles = __builtin_amdgcn_readlane(v_tiles, slot);
    u.g = __builtin_amdgcn_readlane(v_g, slot); u.ks = __builtin_amdgcn_readlane(v_ks, slot);
    u.pt = __builtin_amdgcn_readlane(v_pt, slot); u.pl = __builtin_amdgcn_readlane(v_pl, slot);
    u.ntaps = u.ks * u.ks; u.ntg = udiv(u.ntaps + a.T - 1, a.m_T, a.T);
    u.HWp = a.TW + u.ks - 1; u.HHp = a.TH + u.ks - 1; u.ppt = (u.HWp * u.HHp + 15) >> 4;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int tt = uu * MT + m;
      u.valid[m] = tt < tiles;
      const int ttc = u.valid[m] ? tt : tiles - 1;
      const int img = udiv(ttc, a.m_tpi, a.tpi), ti = ttc - img * a.tpi;
      const int tyi = udiv(ti, a.m_tx, a.tiles_x);
      u.n[m] = row0 + img; u.ty0[m] = tyi * a.TH; u.tx0[m] = (ti - tyi * a.tiles_x) * a.TW;
    }
  };
  auto wbase_of = [&](const C6Unit<MT>& u) { return (int)(((long)u.g * a.wstride + (long)u.nbk * NB * a.Cin) * 2); };

  // ---- halo staging through registers.  Half piece hp = wave + 8k = 8 pixels x 32 fp32 channels (1 KB): lane = (pixel hp*8 + lane/8,
  // channels 4*(lane%8) ..+3).  ho[k]: byte offset of the lane's 16 B inside x at chunk 0 (~0 = padding / beyond the unit: reads 0).
  const int cq = lane & 7;
  auto plan = [&](const C6Unit<MT>& u, unsigned (&ho)[NHP]) {
    const int magic = (1 << 20) / u.HWp + 1;
    const int npx = u.HWp * u.HHp;
    const int cin4 = a.Cin * 4;
#pragma unroll
    for (int k = 0; k < NHP; ++k) {
      const int hp = wave + NW * k;
      const int pxu = 8 * hp + (lane >> 3);                   // pixel inside the unit's two-tile halo image
      const bool t1 = pxu >= u.ppt * 16;
      const int px = pxu - (t1 ? u.ppt * 16 : 0);
      int hy = (int)(((unsigned)px * (unsigned)magic) >> 20);
      if (hy * u.HWp > px) --hy;
      const int hx = px - hy * u.HWp;
      const int n = t1 ? u.n[1] : u.n[0];
      const int iy = (t1 ? u.ty0[1] : u.ty0[0]) - u.pt + hy, ix = (t1 ? u.tx0[1] : u.tx0[0]) - u.pl + hx;
      const bool ok = pxu < 2 * u.ppt * 16 && px < npx && (t1 ? u.valid[1] : u.valid[0]) && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      ho[k] = ok ? (unsigned)(((n * a.H + iy) * a.W + ix) * cin4 + cq * 16) : 0xFFFFFFFFu;
    }
  };
  typedef __attribute__((ext_vector_type(4))) float f4;
  auto nhp_of = [&](int ppt) { return (2 * ppt * 2 + NW - 1) / NW; };     // half pieces per wave (2 tiles x ppt pieces x 2)
  auto halo_load = [&](const unsigned (&ho)[NHP], int c, f4 (&rh)[NHP], int nhp) {
#pragma unroll
    for (int k = 0; k < NHP; ++k)
      if (k < nhp) rh[k] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rx, ho[k], c * 128, 0));
  };
  // Fused input transform: scale / shift [Cin] of the unit's two samples live in a small LDS table [scale | shift][tile][Cin] behind the
  // weight buffers.  Thread q < Cin fetches one 16-byte piece of it together with the unit's first halo chunk (tab_load, inside the
  // MFMA loop) and writes it right before the barrier that precedes the unit's first halo_store (tab_store); loading the values
  // inside halo_store put an L2 round trip into every staging step.  One buffer is enough: between a unit's last halo_store and
  // the next unit's tab_store lie the stage barriers of a whole chunk.
  const int tab_off = 2 * a.hb_bytes + 2 * a.wb_bytes;
  f4 tabreg = (f4)(0.f);
  const int tq_kind = tid >= (a.Cin >> 1) ? 1 : 0, tq_rem = tid - tq_kind * (a.Cin >> 1);
  const int tq_t = tq_rem >= (a.Cin >> 2) ? 1 : 0, tq_c = (tq_rem - tq_t * (a.Cin >> 2)) * 4;
  auto tab_load = [&](const C6Unit<MT>& u) {
    if (sa.in_scale && tid < a.Cin)
      tabreg = *reinterpret_cast<const f4*>((tq_kind ? sa.in_shift : sa.in_scale) + (long)(tq_t ? u.n[1] : u.n[0]) * a.Cin + tq_c);
  };
  auto tab_store = [&]() {
    if (sa.in_scale && tid < a.Cin) *reinterpret_cast<f4*>(lds + tab_off + ((tq_kind * 2 + tq_t) * a.Cin + tq_c) * 4) = tabreg;
  };
  // registers -> (input transform) -> hi / lo bf16 -> LDS images (XHI at 0, XLO at hb_bytes), conv6's swizzled [pixel][64 B] layout
  auto halo_store = [&](const C6Unit<MT>& u, const unsigned (&ho)[NHP], int c, const f4 (&rh)[NHP]) {
    const int nhp = nhp_of(u.ppt), npxu = 32 * u.ppt;
#pragma unroll
    for (int k = 0; k < NHP; ++k) {
      if (k >= nhp) continue;
      const int hp = wave + NW * k;
      const int pxu = 8 * hp + (lane >> 3);
      if (pxu >= npxu) continue;                              // (the last half piece may reach past the two-tile image)
      f4 v = rh[k];
      if (sa.in_scale && ho[k] != 0xFFFFFFFFu) {              // padding stays exactly zero (the reference pads AFTER the norm + ReLU)
        const int t1 = 8 * hp >= u.ppt * 16 ? 1 : 0;            // (a half piece never straddles the two tiles: wave-uniform)
        const f4 sc = *reinterpret_cast<const f4*>(lds + tab_off + (t1 * a.Cin + c * 32 + cq * 4) * 4);
        const f4 sh = *reinterpret_cast<const f4*>(lds + tab_off + ((2 + t1) * a.Cin + c * 32 + cq * 4) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = v[e] * sc[e] + sh[e]; if (sa.in_relu) v[e] = fmaxf(v[e], 0.f); }
      }
      bf16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) { hi[e] = (bf16)v[e]; lo[e] = (bf16)(v[e] - (float)hi[e]); }
      const int off = pxu * 64 + ((((cq >> 1) ^ ((pxu >> 2) & 3))) << 4) + ((cq & 1) << 3);
      *reinterpret_cast<bf16x4*>(lds + off) = hi;
      if (sa.nprod == 3) *reinterpret_cast<bf16x4*>(lds + a.hb_bytes + off) = lo;
    }
  };
  // ---- weights (as conv6.hip; plane = 0 hi / 1 lo)
  const unsigned wlo = (unsigned)((((wave / PPT) * a.Cout + (wave % PPT) * 16 + prow) * a.Cin) * 2 + csl);
  const int wkstep = (NW / PPT) * a.Cout * a.Cin * 2;
  const int wb_half = a.wb_bytes >> 1;                               // P3: [hi taps | lo taps] inside one weight buffer
  auto issue_wpiece = [&](int sb, int k, int wbo) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(lds + wbo + (wave + NW * k) * 1024), 16, wlo, sb + k * wkstep, 0, 0);
    if (P3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(lds + wbo + wb_half + (wave + NW * k) * 1024), 16, wlo, sb + sa.wplane + k * wkstep, 0, 0);
  };
  auto wpieces = [&](int ntl) { return max(0, (ntl - wave / PPT + (NW / PPT) - 1) / (NW / PPT)); };
  auto stage_base = [&](int wbase, int plane, int c, int t0) { return wbase + plane * sa.wplane + (t0 * a.Cout * a.Cin + c * 32) * 2; };

  int j = bid;
  if (j >= total) return;
  C6Unit<MT> cur, nu;
  decode(j, cur);
  nu = cur;
  unsigned hoc[NHP];                           // halo offsets of the unit whose chunks are being LOADED (cur, or nu from cur's last stage on)
  f4 rh[NHP];
  int wbase_cur = wbase_of(cur), wbase_nxt = 0;
  int jn = j + G;
  bool has_next = jn < total;
  const int WB0 = 2 * a.hb_bytes;
  const int nchunks = a.Cin >> 5;
  const int wl = r * 64 + ((h << 4) ^ (((r >> 2) & 3) << 4));
  const int mb0 = MB * wave;
  const int tile_w = mb0 >> 3;
  // prologue: first weight stage (DMA), first halo chunk (registers -> LDS)
  {
    const int sb = stage_base(wbase_cur, 0, 0, 0), np = wpieces(min(a.T, cur.ntaps));
    for (int k = 0; k < np; ++k) issue_wpiece(sb, k, WB0);
    plan(cur, hoc);
    halo_load(hoc, 0, rh, nhp_of(cur.ppt));
    if (sa.in_scale) { tab_load(cur); tab_store(); __syncthreads(); }
    halo_store(cur, hoc, 0, rh);
  }
  int sp = 0;

  while (true) {
    int P0[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const int q = ((mb0 + m) & 7) * 32 + r;
      P0[m] = tile_w * cur.ppt * 16 + (q >> a.tws) * cur.HWp + (q & (a.TW - 1));
    }
    f32x16 acc[MB][NT];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int b = 0; b < NT; ++b) acc[m][b] = (f32x16)(0.f);

    for (int c = 0; c < nchunks; ++c) {
      const bool last_chunk = c == nchunks - 1;
      const bool more = !last_chunk || has_next;                    // another halo chunk follows (this unit's or the next unit's)
      const int lastp = P3 ? 0 : sa.nprod - 1;                      // nprod = 1: x_hi * w_hi only (plain bf16 operands, the trunk BACKWARD)
      for (int prod = 0; prod <= lastp; ++prod) {                   // x_hi * w_hi, x_hi * w_lo, x_lo * w_hi
        const int xoff = prod == 2 ? a.hb_bytes : 0;
        int ky = 0, kx = 0;
        for (int tg = 0; tg < cur.ntg; ++tg) {
          const int t0 = tg * a.T;
          const int ntl = min(a.T, cur.ntaps - t0);
          const bool last_stage = prod == lastp && tg + 1 == cur.ntg;
          // vmcnt counts in issue order: the DMA of this stage is older than the epilogue stores of the previous unit
          if (pending == 16) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");
          else __syncthreads();
          pending = 0;
          // ---- next stage's weights
          const int wbn = WB0 + (sp ^ 1) * a.wb_bytes;
          int nsb = 0, nwp = 0;
          if (tg + 1 < cur.ntg) { nsb = stage_base(wbase_cur, prod == 1, c, t0 + a.T); nwp = wpieces(min(a.T, cur.ntaps - t0 - a.T)); }
          else if (prod < lastp) { nsb = stage_base(wbase_cur, prod == 0, c, 0); nwp = wpieces(min(a.T, cur.ntaps)); }
          else if (!last_chunk) { nsb = stage_base(wbase_cur, 0, c + 1, 0); nwp = wpieces(min(a.T, cur.ntaps)); }
          else if (has_next) {
            decode(jn, nu);
            wbase_nxt = wbase_of(nu);
            nsb = stage_base(wbase_nxt, 0, 0, 0); nwp = wpieces(min(a.T, nu.ntaps));
          }
          // ---- the next halo chunk's fp32 registers: loaded beside the chunk's last stage
          if (last_stage && more) {
            if (last_chunk) { plan(nu, hoc); halo_load(hoc, 0, rh, nhp_of(nu.ppt)); tab_load(nu); }   // (cur's own offsets are no longer needed)
            else halo_load(hoc, c + 1, rh, nhp_of(cur.ppt));
          }
          auto side = [&](int k) { if (k < NWP && k < nwp && !(a.dbg & 2)) issue_wpiece(nsb, k, wbn); };
          // ---- MFMA over the stage's taps (structure of conv6.hip)
          const int hbpx = xoff >> 6;
          const unsigned char* wb = lds + WB0 + sp * a.wb_bytes;
          // software pipeline at k-step granularity (two per tap; a tap-deep pipeline as in conv6.hip needs 32 more registers than
          // this kernel has beside the fp32 halo registers): set A always holds a tap's first 16 channels, set B its second
          bf16x8 fxA[MB], fwA[NT], fxB[MB], fwB[NT];
          bf16x8 lxA[P3 ? MB : 1], lwA[P3 ? NT : 1], lxB[P3 ? MB : 1], lwB[P3 ? NT : 1];   // P3: the lo fragments of the same k-step
          int ad[MB];
          auto tap_addr = [&]() {
            const int toff = ky * cur.HWp + kx + hbpx;
#pragma unroll
            for (int m = 0; m < MB; ++m) {
              const int px = P0[m] + toff;
              ad[m] = (px << 6) + (((px << 2) & 0x30) ^ (h << 4));
            }
            if (++kx == cur.ks) { kx = 0; ++ky; }
          };
          auto load_k = [&](bf16x8 (&fx)[MB], bf16x8 (&fw)[NT], bf16x8 (&lx)[P3 ? MB : 1], bf16x8 (&lw)[P3 ? NT : 1], int tl, int s2) {
            const unsigned char* wt = wb + tl * NB * 64;
#pragma unroll
            for (int m = 0; m < MB; ++m) {
              fx[m] = *reinterpret_cast<const bf16x8*>(lds + (ad[m] ^ (s2 << 5)));
              if (P3) lx[m] = *reinterpret_cast<const bf16x8*>(lds + a.hb_bytes + (ad[m] ^ (s2 << 5)));
            }
#pragma unroll
            for (int b = 0; b < NT; ++b) {
              fw[b] = *reinterpret_cast<const bf16x8*>(wt + b * 2048 + (wl ^ (s2 << 5)));
              if (P3) lw[b] = *reinterpret_cast<const bf16x8*>(wt + wb_half + b * 2048 + (wl ^ (s2 << 5)));
            }
          };
          auto mm = [&](const bf16x8 (&fx)[MB], const bf16x8 (&fw)[NT], const bf16x8 (&lx)[P3 ? MB : 1], const bf16x8 (&lw)[P3 ? NT : 1], bool first) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
              for (int b = 0; b < NT; ++b)
                if ((m + b == 0) == first && !(a.dbg & 1)) {
                  acc[m][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[b], fx[m], acc[m][b], 0, 0, 0);
                  if (P3) {
                    acc[m][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lw[b], fx[m], acc[m][b], 0, 0, 0);
                    acc[m][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[b], lx[m], acc[m][b], 0, 0, 0);
                  }
                }
          };
          tap_addr();
          load_k(fxA, fwA, lxA, lwA, 0, 0);
#pragma unroll
          for (int tl = 0; tl < C6_MAXT; ++tl) {
            if (tl < ntl) {
              mm(fxA, fwA, lxA, lwA, true);
              __builtin_amdgcn_sched_barrier(0);
              load_k(fxB, fwB, lxB, lwB, tl, 1);
              side(tl);
              __builtin_amdgcn_sched_barrier(0);
              mm(fxA, fwA, lxA, lwA, false);
              mm(fxB, fwB, lxB, lwB, true);
              __builtin_amdgcn_sched_barrier(0);
              tap_addr();                                           // (runs one tap past the stage's last: read inside the buffer, never used)
              load_k(fxA, fwA, lxA, lwA, min(tl + 1, ntl - 1), 0);
              __builtin_amdgcn_sched_barrier(0);
              mm(fxB, fwB, lxB, lwB, false);
            } else {
              side(tl);
            }
          }
          if (kx == 0) { kx = cur.ks - 1; --ky; } else --kx;
          sp ^= 1;
        }
      }
      if (more) {
        if (last_chunk) tab_store();                                // the next unit's scale / shift table (visible after the barrier)
        __syncthreads();                                            // every wave is done reading this chunk's hi / lo images
        if (!(a.dbg & 8)) {
          if (last_chunk) halo_store(nu, hoc, 0, rh);
          else halo_store(cur, hoc, c + 1, rh);
        }
      }
    }
    // ---- epilogue: fp32, one 16-byte store per accumulator quad (4 consecutive channels of the lane's pixel)
    {
      float* Y = (float*)a.y;
      const float* R = (const float*)a.res;
      const int n = tile_w ? cur.n[1] : cur.n[0];
      const int ty0 = tile_w ? cur.ty0[1] : cur.ty0[0], tx0 = tile_w ? cur.tx0[1] : cur.tx0[0];
      const bool tv = tile_w ? cur.valid[1] : cur.valid[0];
      float s1 = 0.f, s2 = 0.f;
      typedef __attribute__((ext_vector_type(4))) unsigned u4;
      const bool counted = a.ybytes != 0 && !R && !(a.dbg & 4) && MB * NT * 4 == 16;
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const int q = ((mb0 + m) & 7) * 32 + r;
        const int yy = ty0 + (q >> a.tws), xx = tx0 + (q & (a.TW - 1));
        const bool ok = tv && yy < a.H && xx < a.W;
        const long pix = (((long)n * a.H + yy) * a.W + xx) * a.Cout + cur.nbk * NB + 4 * h;
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            f4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = a.alpha * acc[m][b][4 * i + e];
            if (counted) {                                         // (no residual on this path)
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), ry, ok ? (unsigned)((pix + 32 * b + 8 * i) * 4) : 0xFFFFFFFFu, 0, 0);
              if (ok) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { s1 += v[e]; s2 += v[e] * v[e]; }
              }
            } else if (ok && !(a.dbg & 4)) {
              if (R) { const f4 rv = *reinterpret_cast<const f4*>(R + pix + 32 * b + 8 * i); v += a.beta * rv; }
              *reinterpret_cast<f4*>(Y + pix + 32 * b + 8 * i) = v;
#pragma unroll
              for (int e = 0; e < 4; ++e) { s1 += v[e]; s2 += v[e] * v[e]; }
            }
          }
      }
      if (sa.stats) {
        // both tiles of a wave's blocks belong to one sample.  One (sum, sum of squares) slot per (tile of the image, channel block,
        // wave of the tile): every slot is written exactly once per launch, the consumer (hdmoe_gn1_finalize) adds the slots of a
        // sample in a fixed order -- no atomics, a sample's statistics do not depend on its batch.
        s1 = wave_sum(s1); s2 = wave_sum(s2);
        if (lane == 0 && tv) {
          const int ti = (ty0 >> (8 - a.tws)) * a.tiles_x + (tx0 >> a.tws);
          const long slot = ((long)n * a.tpi + ti) * a.nblk * 4 + cur.nbk * 4 + (wave & 3);
          sa.stats[2 * slot] = s1; sa.stats[2 * slot + 1] = s2;
        }
      }
      // (a LOWER bound of the instructions issued after the DMA is what the counted wait needs: the statistics stores are not counted)
      pending = counted ? 16 : 0;
    }
    if (!has_next) break;
    cur = nu;
    wbase_cur = wbase_nxt;
    jn += G;
    has_next = jn < total;
  }
#endif
}

}  // namespace
