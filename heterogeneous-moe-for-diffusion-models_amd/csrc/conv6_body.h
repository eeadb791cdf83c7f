// Device body of the conv6 kernel (see conv6.hip for the design notes): shared by the stand-alone kernel and by the fused backward
// launch of bwd6.hip, where the same program runs in the first `G` workgroups of a larger grid.
#pragma once
#include "common.h"
#include "conv6_common.h"

namespace {

template <int MT, int NT, bool FILM = false>
DEVI void conv6_body(const C6Args& a, const int bid, const int G) {
#if __HIP_DEVICE_COMPILE__                     // (the buffer-descriptor builtins exist in the device pass only; the host pass needs just the stub)
  constexpr int NB = 32 * NT;
  constexpr int PPT = NB / 16;                 // weight DMA pieces per tap (16 rows x 64 B each)
  constexpr int NW = C6_NW;
  constexpr int MB = 8 * MT / NW;              // 32-pixel blocks per wave
  constexpr int NPW = (MT * 34 + NW - 1) / NW; // halo DMA pieces per wave (<= MT * 34 pieces of 16 pixels)
  constexpr int NWP = 40 / NW;                 // weight DMA pieces per wave per stage (T * PPT <= 40)
  constexpr int SPT = 2;                       // DMA pieces of either kind issued per tap inside the MFMA loop
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int csl = ((lane & 3) ^ ((lane >> 4) & 3)) << 4;     // byte offset of the (swizzled) 16-B channel slot this lane fetches
  const int prow = lane >> 2;                                // its row inside a 16-row DMA piece
  // Buffer descriptors: DMA lanes address their operand by a 32-bit byte offset; an out-of-range offset (~0: padding pixels)
  // makes the hardware write zeros, so the halo needs neither a select nor a zero page.
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.wbytes, 0x00020000);

  uint32_t film_lo = a.film_seed_lo, film_hi = a.film_seed_hi;
  if (FILM && a.film_e && a.film_p > 0.f) mix_seed(film_lo, film_hi, a.film_seed_dev);
  const float film_inv = a.film_p > 0.f ? 1.f / (1.f - a.film_p) : 1.f;
  int nstamp = 0;
  auto stamp = [&](int tag) {
    if (a.stamps && bid == 0 && lane == 0 && nstamp < 63) {
      a.stamps[wave * 64 + nstamp] = ((unsigned long long)tag << 56) | (__builtin_amdgcn_s_memtime() & 0x00FFFFFFFFFFFFFFull);
      ++nstamp;
    }
  };
  stamp(1);
  // ---- unit list: groups in descending kernel size; unit = MT tiles x one channel block.
  // Lane oi < ngroups of every wave keeps slot oi of the list in registers (group, kernel size, pads, first row, tile count, first
  // unit); decoding a unit index is then a ballot + a few v_readlane -- no memory access (seg / kernarg reads in the decode loop cost
  // ~5 us of scalar-load latency per unit in the first version).
  const int oi_l = lane & 7;
  int v_g = 0, v_ks = 0, v_pt = 0, v_pl = 0;
#pragma unroll
  for (int oi = 0; oi < HDMOE_MAX_GROUPS; ++oi) {
    const bool me = oi_l == oi;
    v_g = me ? a.order[oi] : v_g;
  }
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
  int v_ustart = v_units;                                    // inclusive prefix over the 8 slots (lanes 0..7), then made exclusive
#pragma unroll
  for (int d = 1; d < 8; d <<= 1) {
    const int o = __shfl_up(v_ustart, d, 8);
    if (oi_l >= d) v_ustart += o;
  }
  const int total = __builtin_amdgcn_readlane(v_ustart, 7) * a.nblk;
  v_ustart -= v_units;
  auto udiv = [](int x, unsigned magic, int d) {            // x / d for small non-negative x: 32.32 reciprocal + fix-up
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
  auto wbase_of = [&](const C6Unit<MT>& u) { return (int)(((long)u.g * a.wstride + (long)u.nbk * NB * a.w_rowpitch) * 2); };

  // ---- per-lane source offsets (bytes from x, channel chunk 0) of this wave's halo pieces of a unit; ~0 = padding (reads as zero).
  // The (row, column) of a piece's pixel inside the halo tile depends on the kernel size only: kept packed in hyx[] and recomputed
  // when the kernel size changes (units are sorted by kernel size); per unit that leaves ~8 VALU instructions per piece.
  unsigned hyx[NPW];
  int hyx_ks = -1;
  auto hyx_update = [&](const C6Unit<MT>& u) {
    if (u.ks == hyx_ks) return;
    hyx_ks = u.ks;
    const int magic = (1 << 20) / u.HWp + 1;
    const int npx = u.HWp * u.HHp;
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
      const int pi = wave + NW * k;
      const int tile = (MT == 2 && pi >= u.ppt) ? 1 : 0;
      const int px = 16 * (pi - tile * u.ppt) + prow;
      int hy = (int)(((unsigned)px * (unsigned)magic) >> 20);
      if (hy * u.HWp > px) --hy;
      const int hx = px - hy * u.HWp;
      hyx[k] = (pi < MT * u.ppt && px < npx) ? (unsigned)((hy << 8) | hx) : 0xFFFFFFFFu;
    }
  };
  const int cin2 = a.Cin * 2;
  auto plan_piece = [&](const C6Unit<MT>& u, int k) -> unsigned {      // k is a compile-time constant at every call site
    const int pi = wave + NW * k;
    const bool t1 = MT == 2 && pi >= u.ppt;                   // wave-uniform: which of the unit's tiles this piece belongs to
    const int n = t1 ? u.n[MT - 1] : u.n[0];
    const int y0 = (t1 ? u.ty0[MT - 1] : u.ty0[0]) - u.pt, x0 = (t1 ? u.tx0[MT - 1] : u.tx0[0]) - u.pl;
    const int vld = t1 ? u.valid[MT - 1] : u.valid[0];
    const int hy = (int)(hyx[k] >> 8), hx = (int)(hyx[k] & 255u);
    const int iy = y0 + hy, ix = x0 + hx;
    const bool ok = hyx[k] != 0xFFFFFFFFu && vld && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    return ok ? (unsigned)(((n * a.H + iy) * a.W + ix) * cin2 + csl) : 0xFFFFFFFFu;
  };
  // halo piece k of this wave: 16 pixels x 32 channels of chunk c -> LDS piece wave + NW k of the halo buffer at hbo
  auto issue_hpiece = [&](unsigned off, int k, int c, int hbo) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(lds + hbo + (wave + NW * k) * 1024), 16, off, c * 64, 0, 0);
  };
  auto hpieces = [&](int ppt) { return (MT * ppt + NW - 1) / NW; }; // pieces per wave (the buffer is padded to NW x that many pieces)
  // weights of one stage: taps [t0, t0 + ntl) x 32 channels of chunk c -> [tap][NB rows][64 B] at wbo.  Piece pi = wave + NW k is
  // tap pi / PPT, rows 16 * (pi % PPT) ..: with NW % PPT == 0 only the tap depends on k.
  const unsigned wlo = (unsigned)(((wave / PPT) * a.w_tapstride + ((wave % PPT) * 16 + prow) * a.w_rowpitch) * 2 + csl);
  const int wkstep = (NW / PPT) * a.w_tapstride * 2;           // bytes between a wave's consecutive pieces
  // one piece (k-th of this wave) of the stage whose first byte inside the weight image is sb
  auto issue_wpiece = [&](int sb, int k, int wbo) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(lds + wbo + (wave + NW * k) * 1024), 16, wlo, sb + k * wkstep, 0, 0);
  };
  auto wpieces = [&](int ntl) { return max(0, (ntl - wave / PPT + (NW / PPT) - 1) / (NW / PPT)); };   // this wave's share of a stage
  auto stage_base = [&](int wbase, int c, int t0) { return wbase + (t0 * a.w_tapstride + c * 32) * 2; };

  int j = bid;
  if (j >= total) return;
  stamp(2);
  C6Unit<MT> cur, nu;
  decode(j, cur);
  nu = cur;
  stamp(3);
  unsigned hoc[NPW], hon[NPW];
  hyx_update(cur);
  int wbase_cur = wbase_of(cur);
  int wbase_nxt = 0;
  int jn = j + G;
  bool has_next = jn < total;

  const int HB0 = 0, WB0 = 2 * a.hb_bytes;
  const int nchunks = a.Cin >> 5;
  const int wl = r * 64 + ((h << 4) ^ (((r >> 2) & 3) << 4));       // this lane's weight-fragment byte offset inside a tap block
  const int mb0 = MB * wave;                                        // first 32-pixel block of this wave (of MT * 8)
  const int tile_w = (MT == 2) ? (mb0 >> 3) : 0;                    // the tile all blocks of this wave belong to
  // prologue: first weight stage, then the first unit's first chunk
  {
    const int sb = stage_base(wbase_cur, 0, 0), np = wpieces(min(a.T, cur.ntaps));
    for (int k = 0; k < np; ++k) issue_wpiece(sb, k, WB0);
    const int nh = hpieces(cur.ppt);
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
      hoc[k] = plan_piece(cur, k);
      hon[k] = 0xFFFFFFFFu;
      if (k < nh) issue_hpiece(hoc[k], k, 0, HB0);
    }
  }
  stamp(4);
  int par = 0, sp = 0;                                              // halo-buffer parity / weight-buffer parity

  while (true) {
    // pixel bases of this wave's MT blocks inside the halo image (pixels), for tap (0, 0)
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
      // halo tile fetched beside this chunk's first stage: the unit's next chunk (mode 1), or chunk 0 of the NEXT unit (mode 2; its
      // per-lane source offsets are computed piece by piece inside the MFMA loop, where their VALU work is free)
      int hmode = 0, nh = 0;
      if (!last_chunk) { hmode = 1; nh = hpieces(cur.ppt); }
      else if (has_next) {
        stamp(5);
        decode(jn, nu);
        hyx_update(nu);
        wbase_nxt = wbase_of(nu);
        hmode = 2; nh = hpieces(nu.ppt);
        stamp(6);
      }
      if (a.dbg & 2) nh = 0;
      const int hbn = HB0 + (par ^ 1) * a.hb_bytes;
      int ky = 0, kx = 0;                                           // tap cursor, carried across the chunk's stages
      for (int tg = 0; tg < cur.ntg; ++tg) {
        const int t0 = tg * a.T;
        const int ntl = min(a.T, cur.ntaps - t0);
        stamp(7);
        __syncthreads();                                            // stage (c, tg) has landed; the buffers of the previous stage are free
        stamp(8);
        // ---- the next stage's weights: this wave's pieces are issued one per tap inside the MFMA loop (a DMA instruction occupies
        //      the wave for 100-200 cycles when all eight waves issue theirs together right behind the barrier)
        const int wbn = WB0 + (sp ^ 1) * a.wb_bytes;
        int nsb = 0, nwp = 0;
        if (tg + 1 < cur.ntg) { nsb = stage_base(wbase_cur, c, t0 + a.T); nwp = wpieces(min(a.T, cur.ntaps - t0 - a.T)); }
        else if (!last_chunk) { nsb = stage_base(wbase_cur, c + 1, 0); nwp = wpieces(min(a.T, cur.ntaps)); }
        else if (has_next) { nsb = stage_base(wbase_nxt, 0, 0); nwp = wpieces(min(a.T, nu.ntaps)); }
        if (a.dbg & 2) nwp = 0;
        const int nhs = tg == 0 ? nh : 0;                           // halo pieces ride on the chunk's first stage
        // per-tap side work (k compile-time): weight piece k of the next stage, halo piece k of the next chunk / unit
        auto side = [&](int k) {
          if (k < NWP && k < nwp) issue_wpiece(nsb, k, wbn);
          if (k < NPW && k < nhs) {
            if (hmode == 2) { hon[k] = plan_piece(nu, k); issue_hpiece(hon[k], k, 0, hbn); }
            else issue_hpiece(hoc[k], k, c + 1, hbn);
          }
        };
        stamp(9);
        // ---- MFMA over the stage's taps, one tap (two 16-channel k-steps) per step; the fragments of tap t+1 are read from LDS
        //      right after the first MFMA of tap t has been issued (that MFMA is where the wait for tap t's own fragments sits)
        const int hbpx = (HB0 + par * a.hb_bytes) >> 6;             // halo buffer base in pixel units (multiple of 16: swizzle-neutral)
        const unsigned char* wb = lds + WB0 + sp * a.wb_bytes;
        bf16x8 fxa[2][MB], fwa[2][NT], fxb[2][MB], fwb[2][NT];
        auto load_tap = [&](bf16x8 (&fx)[2][MB], bf16x8 (&fw)[2][NT], int tl) {
          const int toff = ky * cur.HWp + kx + hbpx;
          const unsigned char* wt = wb + tl * NB * 64;
#pragma unroll
          for (int m = 0; m < MB; ++m) {
            const int px = P0[m] + toff;
            const int ad = (px << 6) + (((px << 2) & 0x30) ^ (h << 4));
            fx[0][m] = *reinterpret_cast<const bf16x8*>(lds + ad);
            fx[1][m] = *reinterpret_cast<const bf16x8*>(lds + (ad ^ 32));
          }
#pragma unroll
          for (int b = 0; b < NT; ++b) {
            fw[0][b] = *reinterpret_cast<const bf16x8*>(wt + b * 2048 + wl);
            fw[1][b] = *reinterpret_cast<const bf16x8*>(wt + b * 2048 + (wl ^ 32));
          }
          if (++kx == cur.ks) { kx = 0; ++ky; }
        };
        // One tap = one scheduling region: its 2 * MB * NT MFMAs (fragment set f) and the LDS reads + address arithmetic of the NEXT
        // tap's fragments (set g) are interleaved by the scheduler -- one wave per SIMD has nobody else to cover its issue gaps, so
        // every non-MFMA instruction has to sit in the 32-cycle shadow of an MFMA.  The DMA pieces (uniform branches) follow the region.
        auto step = [&](bf16x8 (&fx)[2][MB], bf16x8 (&fw)[2][NT], bf16x8 (&gx)[2][MB], bf16x8 (&gw)[2][NT], int tl) {
          __builtin_amdgcn_sched_barrier(0);
          // (the tap cursor may run one tap past the stage's last one: such fragments are read inside the buffer but never used)
          load_tap(gx, gw, min(tl + 1, ntl - 1));
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
              for (int b = 0; b < NT; ++b)
                acc[m][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[s2][b], fx[s2][m], acc[m][b], 0, 0, 0);
          constexpr int NM = 2 * MB * NT, NR = 2 * (MB + NT);       // MFMAs / LDS reads of the region
#pragma unroll
          for (int i = 0; i < NM; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
            if (i < NR - NM) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);       // its share of the reads
            else if (i < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);      // ... of the address VALU
            __builtin_amdgcn_sched_group_barrier(0x004, 1, 0);      // ... of the scalar bookkeeping
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < SPT; ++q) side(SPT * tl + q);
        };
        if (!(a.dbg & 1)) {
          load_tap(fxa, fwa, 0);
#pragma unroll
          for (int tl = 0; tl < C6_MAXT; ++tl) {                    // fully unrolled: tl (and with it every DMA / plan index) is static
            if (tl < ntl) {
              if (tl & 1) step(fxb, fwb, fxa, fwa, tl); else step(fxa, fwa, fxb, fwb, tl);
            } else {
#pragma unroll
              for (int q = 0; q < SPT; ++q) side(SPT * tl + q);     // side work left over when the stage has fewer taps than pieces
            }
          }
          // undo the cursor's run-ahead (load_tap advanced it once more than there were taps)
          if (kx == 0) { kx = cur.ks - 1; --ky; } else --kx;
        } else {
#pragma unroll
          for (int k = 0; k < SPT * C6_MAXT; ++k) side(k);
        }
        sp ^= 1;
      }
      par ^= 1;
    }
    // ---- epilogue: y = alpha * acc + beta * res, 16-byte stores (two register quads paired across the half-waves)
    stamp(10);
    if (!(a.dbg & 4)) {
      bf16* Y = (bf16*)a.y;
      const bf16* R = (const bf16*)a.res;
      const int n = (MT == 2 && tile_w) ? cur.n[MT - 1] : cur.n[0];
      const int ty0 = (MT == 2 && tile_w) ? cur.ty0[MT - 1] : cur.ty0[0];
      const int tx0 = (MT == 2 && tile_w) ? cur.tx0[MT - 1] : cur.tx0[0];
      const bool tv = (MT == 2 && tile_w) ? cur.valid[MT - 1] : cur.valid[0];
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const int q = ((mb0 + m) & 7) * 32 + r;
        const int yy = ty0 + (q >> a.tws), xx = tx0 + (q & (a.TW - 1));
        const bool ok = tv && yy < a.H && xx < a.W;
        const long pix = (((long)n * a.H + yy) * a.W + xx) * a.Cout + cur.nbk * NB;
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            // quads 2p (channels 16p + 4h ..) and 2p+1 (channels 16p + 8 + 4h ..) of this lane
            float v[8];
            if (a.gbias && ok) {                                    // per-expert bias map (the folded ones channel), inside the alpha scale
              const float* gb = a.gbias + (((long)cur.g * a.H + yy) * a.W + xx) * a.Cout + cur.nbk * NB + 32 * b + 16 * p + 4 * h;
              const float4 g0 = *reinterpret_cast<const float4*>(gb), g1 = *reinterpret_cast<const float4*>(gb + 8);
              v[0] = a.alpha * (acc[m][b][8 * p] + g0.x); v[1] = a.alpha * (acc[m][b][8 * p + 1] + g0.y);
              v[2] = a.alpha * (acc[m][b][8 * p + 2] + g0.z); v[3] = a.alpha * (acc[m][b][8 * p + 3] + g0.w);
              v[4] = a.alpha * (acc[m][b][8 * p + 4] + g1.x); v[5] = a.alpha * (acc[m][b][8 * p + 5] + g1.y);
              v[6] = a.alpha * (acc[m][b][8 * p + 6] + g1.z); v[7] = a.alpha * (acc[m][b][8 * p + 7] + g1.w);
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = a.alpha * acc[m][b][8 * p + e];
            }
            if (R && ok) {                                          // residual added in fp32, before the one rounding to bf16
              const long o0 = pix + 32 * b + 16 * p + 4 * h;
              const bf16x4 r0 = *reinterpret_cast<const bf16x4*>(R + o0);
              const bf16x4 r1 = *reinterpret_cast<const bf16x4*>(R + o0 + 8);
#pragma unroll
              for (int e = 0; e < 4; ++e) { v[e] += a.beta * (float)r0[e]; v[4 + e] += a.beta * (float)r1[e]; }
            }
            typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
            const unsigned A0 = __builtin_bit_cast(unsigned, (bf2){(bf16)v[0], (bf16)v[1]}), A1 = __builtin_bit_cast(unsigned, (bf2){(bf16)v[2], (bf16)v[3]});
            const unsigned B0 = __builtin_bit_cast(unsigned, (bf2){(bf16)v[4], (bf16)v[5]}), B1 = __builtin_bit_cast(unsigned, (bf2){(bf16)v[6], (bf16)v[7]});
            // after the swap the lower half-wave holds channels 16p .. 16p+7 of its pixel, the upper half 16p+8 .. 16p+15
            const u32x2 s0 = __builtin_amdgcn_permlane32_swap(A0, B0, false, false);
            const u32x2 s1 = __builtin_amdgcn_permlane32_swap(A1, B1, false, false);
            if (ok) *reinterpret_cast<uint4*>(Y + pix + 32 * b + 16 * p + 8 * h) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            if (FILM && a.film_e && ok) {
              // FiLM + mp_silu + dropout of the 8 channels this lane just stored (from the bf16-ROUNDED values, as the separate pass reads them)
              const long eo = pix + 32 * b + 16 * p + 8 * h;                  // element index of the first of the 8 channels
              const int c0 = cur.nbk * NB + 32 * b + 16 * p + 8 * h;
              const unsigned pk[4] = {s0[0], s1[0], s0[1], s1[1]};
              uint32_t r4[8];
              if (a.film_p > 0.f) {
                const long q0 = eo >> 2;
                philox((uint32_t)q0, (uint32_t)(q0 >> 32), film_lo, film_hi, r4);
                philox((uint32_t)(q0 + 1), (uint32_t)((q0 + 1) >> 32), film_lo, film_hi, r4 + 4);
              }
              unsigned ho[4];
#pragma unroll
              for (int j2 = 0; j2 < 4; ++j2) {
                const bf2 yv = __builtin_bit_cast(bf2, pk[j2]);
                float f0 = mp_silu_f((float)yv[0] * a.film_e[(long)n * a.Cout + c0 + 2 * j2]);
                float f1 = mp_silu_f((float)yv[1] * a.film_e[(long)n * a.Cout + c0 + 2 * j2 + 1]);
                if (a.film_p > 0.f) {
                  f0 = u01(r4[2 * j2]) >= a.film_p ? f0 * film_inv : 0.f;
                  f1 = u01(r4[2 * j2 + 1]) >= a.film_p ? f1 * film_inv : 0.f;
                }
                ho[j2] = __builtin_bit_cast(unsigned, (bf2){(bf16)f0, (bf16)f1});
              }
              *reinterpret_cast<uint4*>(reinterpret_cast<bf16*>(a.film_h) + eo) = make_uint4(ho[0], ho[1], ho[2], ho[3]);
            }
          }
      }
    }
    stamp(11);
    if (!has_next) break;
    cur = nu;
#pragma unroll
    for (int k = 0; k < NPW; ++k) hoc[k] = hon[k];
    wbase_cur = wbase_nxt;
    jn += G;
    has_next = jn < total;
  }
#endif
}

}  // namespace
