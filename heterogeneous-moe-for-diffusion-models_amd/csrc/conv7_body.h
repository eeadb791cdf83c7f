// Device body of the conv7 kernel: whole-image streaming implicit-GEMM convolution for the k x k expert layers on 32 x 32 maps
// (forward and dgrad of MP_Conv, reference models/model_internals.py:253-275, grouped by expert as in models/model_config1.py:25-37).
//
// Why a third kernel next to conv6 ("Round 4"): on this model's 32 / 64-channel layers conv6's 256-pixel work
// units spend two thirds of their time outside the matrix pipe -- 1.5 LDS fragment reads per MFMA with the address arithmetic of a
// run-time tap cursor, a unit decode and a halo plan per unit, a barrier per stage of a 512-pixel tile.  Here
//   * a work unit is a WHOLE image: one 8-wave workgroup per CU keeps the image's current 32-channel chunk (32 rows x 35 pixel slots x
//     64 B, pad slots shared between neighbouring rows and zeroed once) in LDS, double-buffered, filled by LDS-DMA with one scalar
//     offset per 1-KB piece (no per-unit halo plan: the tile IS the image); units are dealt to the workgroups in a snake over the
//     cost-sorted image list, so a workgroup with a 5x5 image also gets a 3x3 one;
//   * wave w owns output rows 4w .. 4w+3 (four 32-pixel blocks) and all output channels: a weight fragment is reused by four (eight
//     with 64 output channels) MFMAs and an input-row fragment by every kernel row of a stage that touches it --
//     0.56 - 0.69 ds_read_b128 per MFMA instead of 1.0 - 1.5;
//   * the tap schedule (kernel column outer, kernel rows in stages of <= 4 / CO taps) is a compile-time table per kernel size: every
//     LDS address is a per-lane constant (one of seven column shifts) plus a scalar, rows outside the image read a zero row;
//   * weights stream through a 2 x 8 KB ring, one 1-KB DMA piece per wave and stage, counted vmcnt waits keep the next image's tile
//     and the epilogue stores in flight across the stage barriers.
#pragma once
#include <type_traits>
#include <type_traits>
#include <utility>
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void* c7_lptr_t;

struct C7Args {
  const void* x; const void* w; void* y; const void* res; const int* seg;
  long wstride;                        // elements per group in the weight image [g][tap][Cout][Cin]
  int N, Cin, Cout, ngroups;           // H = W = 32 (or 16: conv7_body<.., .., true>)
  int ks[HDMOE_MAX_GROUPS], order[HDMOE_MAX_GROUPS];   // kernel size per group; groups in descending kernel size
  float alpha, beta;
  int xbytes, wbytes;
  int dbg;                             // development ablations: 1 skip the MFMAs, 2 skip the tile DMA of later units, 4 skip the stores
  unsigned long long* stamps;          // development: s_memtime stamps of workgroup 0 ([wave][64] slots; hdmoe_conv6_debug_stamps), or null
};

// Tile geometry.  32 x 32 maps: a tile is one image, [32 rows][35 pixel slots][64 B] (32 pixels + 3 pad slots shared with the next row), wave w
// owns rows 4w .. 4w+3.  16 x 16 maps: a tile is TWO images of one expert stacked as 32 "virtual" rows of [19 slots][64 B] (rows 0-15 image A,
// 16-31 image B; 16 * 19 slots = 304 = 0 mod 16, so a 32-lane fragment read -- lanes 0-15 a row of A, lanes 16-31 the same row of B --
// keeps the bank pattern of 32 consecutive slots), wave w owns rows 2w, 2w+1 of both images.
template <bool W16> struct C7Geo;
template <> struct C7Geo<false> {
  static constexpr int IMG = 32, MB = 4, ROWB = 35 * 64, TILE = (32 * 35 + 3) * 64, WBUF = 8192, TCO = 4, PPW = 8, HBOFF = 0;
  static constexpr int ZBYTES = 38 * 64;
};
template <> struct C7Geo<true> {
  static constexpr int IMG = 16, MB = 2, ROWB = 19 * 64, TILE = (32 * 19 + 3) * 64, WBUF = 28672, TCO = 14, PPW = 4, HBOFF = 16 * 19 * 64;
  static constexpr int ZBYTES = HBOFF + 22 * 64;
};
template <bool W16> struct C7Lds {
  using GEO = C7Geo<W16>;
  static constexpr int T0 = 2 * GEO::WBUF;                     // tiles start behind the weight ring (keeps every row address >= 0)
  static constexpr int ZROW = T0 + 2 * GEO::TILE;              // zero pixel slots: what a row outside the image reads
  static constexpr int BYTES = ZROW + GEO::ZBYTES;             // 162,560 B / 156,416 B
};

// One stage = up to T = TCO / CO consecutive kernel rows of one kernel column (TCO: 2-KB weight blocks per ring buffer).
template <int KS, int CO, int TCO> struct C7Sched {
  static constexpr int T = TCO / CO;
  static constexpr int SPC = (KS + T - 1) / T;                 // stages per kernel column
  static constexpr int NS = KS * SPC;                          // stages per channel chunk
  // kernel rows of stage i of a column, balanced (5 -> 2 + 3, 7 -> 3 + 4 / 1 + 2 + 2 + 2)
  static constexpr int ky0(int i) { return (KS * i) / SPC; }
  static constexpr int nt(int i) { return (KS * (i + 1)) / SPC - (KS * i) / SPC; }
};

struct C7Unit { int g, n, n2, ks; };

template <typename F, int... I> DEVI void c7_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F> DEVI void c7_static_for(F&& f) { c7_static_for_impl(f, std::make_integer_sequence<int, N>{}); }
template <int N> DEVI void c7_wait_barrier() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(N) : "memory"); }

template <int CO, int KMASK, bool W16>
DEVI void conv7_body(const C7Args& a, const int bid, const int G) {
#if __HIP_DEVICE_COMPILE__
  using GEO = C7Geo<W16>;
  using L = C7Lds<W16>;
  constexpr int MB = GEO::MB, IMG = GEO::IMG, ROWB = GEO::ROWB, TILE = GEO::TILE, WBUF = GEO::WBUF, TCO = GEO::TCO, PPW = GEO::PPW;
  constexpr int T0 = L::T0, ZROW = L::ZROW;
  constexpr int NSTORE = MB * CO * 2;                           // epilogue stores per wave
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.wbytes, 0x00020000);
  const int CI = a.Cin >> 5;
  const int cin2 = a.Cin * 2;
  int nstamp = 0;
  auto stamp = [&](int tag) {
    if (a.stamps && bid == 0 && lane == 0 && nstamp < 63) {
      a.stamps[wave * 64 + nstamp] = ((unsigned long long)tag << 56) | (__builtin_amdgcn_s_memtime() & 0x00FFFFFFFFFFFFFFull);
      ++nstamp;
    }
  };
  stamp(1);

  // ---- unit list: images (pairs of images on 16 x 16 maps) of the groups in descending kernel size, dealt in a snake over the workgroups
  int cum[HDMOE_MAX_GROUPS + 1];
  cum[0] = 0;
#pragma unroll
  for (int i = 0; i < HDMOE_MAX_GROUPS; ++i) {
    int cnt = 0;
    if (i < a.ngroups) { const int g = a.order[i]; cnt = a.seg ? a.seg[g + 1] - a.seg[g] : a.N; }
    cum[i + 1] = cum[i] + (W16 ? (cnt + 1) >> 1 : cnt);
  }
  const int total = cum[HDMOE_MAX_GROUPS];
  auto unit_at = [&](int q, C7Unit& u) -> bool {
    const int pos = q * G + ((q & 1) ? G - 1 - bid : bid);
    if (pos >= total) return false;
    int slot = 0;
#pragma unroll
    for (int i = 1; i < HDMOE_MAX_GROUPS; ++i) slot += (i < a.ngroups && pos >= cum[i]) ? 1 : 0;
    int base = 0, g = 0;
#pragma unroll
    for (int i = 0; i < HDMOE_MAX_GROUPS; ++i) if (i == slot) { base = cum[i]; g = a.order[i]; }
    int ks = 3;
#pragma unroll
    for (int i = 0; i < HDMOE_MAX_GROUPS; ++i) if (i == g) ks = a.ks[i];
    const int s0 = a.seg ? a.seg[g] : 0, s1 = a.seg ? a.seg[g + 1] : a.N;
    u.g = g; u.ks = ks;
    if (W16) { u.n = s0 + 2 * (pos - base); u.n2 = u.n + 1 < s1 ? u.n + 1 : -1; }
    else { u.n = s0 + pos - base; u.n2 = -1; }
    return true;
  };

  // ---- zero the pad slots of both tiles and the zero rows (never written again: the DMA only touches data slots)
  for (int i = tid; i < 33 * 12; i += 512) {
    const int off = (i / 12) * ROWB + (i % 12) * 16;
    *reinterpret_cast<uint4*>(lds + T0 + off) = make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(lds + T0 + TILE + off) = make_uint4(0, 0, 0, 0);
  }
  for (int i = tid; i < GEO::ZBYTES / 16; i += 512) *reinterpret_cast<uint4*>(lds + ZROW + i * 16) = make_uint4(0, 0, 0, 0);

  // ---- per-lane constants
  // tile DMA: lane i of a piece holds pixel i >> 2 (of 16) at row slot 3 + pixel (+ 16 in the right half of a 32-pixel row: same key),
  // LDS chunk i & 3; it fetches the source chunk (i & 3) ^ key(slot)
  const int dpx = lane >> 2;
  const unsigned xlane = (unsigned)(dpx * cin2 + (((lane & 3) ^ (((3 + dpx) >> 2) & 3)) << 4));
  // weight DMA: lane i holds output row i >> 2 (of 16), chunk i & 3
  const unsigned wlane = (unsigned)(dpx * cin2 + (((lane & 3) ^ ((dpx >> 2) & 3)) << 4));
  // fragment reads: row slot (lane's pixel column) + cc (cc = kernel column + 3 - pad, 0 .. 6), 16-channel k-step 0; k-step 1 = ^ 32
  int acol[7];
#pragma unroll
  for (int cc = 0; cc < 7; ++cc) {
    const int col = (W16 ? (r & 15) : r) + cc;
    acol[cc] = (col << 6) + ((h ^ ((col >> 2) & 3)) << 4) + (W16 ? (r >> 4) * GEO::HBOFF : 0);
  }
  const int wl = (r << 6) + ((h ^ ((r >> 2) & 3)) << 4);        // weight fragment of output row r, k-step 0

  // issue this wave's pieces of chunk c of a unit's image(s) into tile buffer tb: 32 x 32: rows 4 wave .. + 3, two 16-pixel halves each;
  // 16 x 16: rows 2 wave, 2 wave + 1 of both images (an absent second image reads as zeros: out-of-range offset)
  auto issue_tile = [&](const C7Unit& u, int c, int tb) {
    const int lbase = T0 + tb * TILE + 3 * 64;
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
      if (W16) {
        const int img = k >> 1, row = 2 * wave + (k & 1);
        const int n = img ? u.n2 : u.n;
        const int so = (n * 16 + row) * 16 * cin2 + c * 64;
        const unsigned vo = n < 0 ? 0xFFFFFFFFu : xlane;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (c7_lptr_t)(lds + lbase + (img * 16 + row) * ROWB), 16, vo, n < 0 ? 0 : so, 0, 0);
      } else {
        const int row = 4 * wave + (k >> 1), half = k & 1;
        const int so = ((u.n * 32 + row) * 32 + half * 16) * cin2 + c * 64;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (c7_lptr_t)(lds + lbase + row * ROWB + half * 1024), 16, xlane, so, 0, 0);
      }
    }
  };
  // this wave's pieces of the weight stage (kernel column kx, rows ky0 .. ky0 + nt) of chunk c, group g, kernel size ks -> ring buffer sp
  auto issue_wstage = [&](int g, int ks, int c, int blk, int kx, int ky0, int nt, int sp) {
#pragma unroll
    for (int j = 0; j < (2 * TCO + 7) / 8; ++j) {
      const int pi = wave + 8 * j;
      const int ts = pi / (2 * CO), pc = pi % (2 * CO);         // tap slot, 16-row piece inside the tap
      if (ts < nt) {
        const int tap = (ky0 + ts) * ks + kx;
        const int so = (int)(((long)g * a.wstride + (long)(tap * a.Cout + blk * 32 * CO + pc * 16) * a.Cin) * 2) + c * 64;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (c7_lptr_t)(lds + sp * WBUF + pi * 1024), 16, wlane, so, 0, 0);
      }
    }
  };

  C7Unit cur, nxt;
  int q = 0;
  if (!unit_at(q, cur)) return;
  bool has_next = unit_at(q + 1, nxt);
  int tp = 0, sp = 0;                                           // tile / weight ring parity
  auto nt0_of = [](int ks) { return ks == 3 ? C7Sched<3, CO, GEO::TCO>::nt(0) : (ks == 5 ? C7Sched<5, CO, GEO::TCO>::nt(0) : C7Sched<7, CO, GEO::TCO>::nt(0)); };
  const int nblk = a.Cout / (32 * CO);                          // output-channel blocks of 32 CO: walked one after the other over the same image
  issue_wstage(cur.g, cur.ks, 0, 0, 0, 0, nt0_of(cur.ks), 0);    // first stage of the unit's schedule: column 0, rows 0 .. nt0
  issue_tile(cur, 0, 0);
  bool first = true;

  // ---- one unit: all chunks, all stages; prefetches the next unit's first tile and first weight stage
  auto run_unit = [&](auto ks_tag) {
    constexpr int KS = decltype(ks_tag)::value;
    using S = C7Sched<KS, CO, TCO>;
    constexpr int P = (KS - 1) / 2, Q = 3 - P;
    for (int blk = 0; blk < nblk; ++blk) {
    const bool last_blk = blk == nblk - 1;
    f32x16 acc[MB][CO];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int b = 0; b < CO; ++b) acc[m][b] = (f32x16)(0.f);

    for (int c = 0; c < CI; ++c) {
      const bool last_chunk = c == CI - 1;
      // the tile fetched beside this chunk's first stage: the unit's next chunk; behind the last chunk the first chunk again for the next
      // output block (a one-chunk image simply stays where it is), or the next unit's first chunk
      const bool same_tile = last_chunk && !last_blk && CI == 1;
      const bool tile_next = (!last_chunk || (!last_blk && CI > 1) || (last_blk && has_next)) && !((a.dbg & 2) && last_chunk);
      const int tbase = T0 + tp * TILE;
      c7_static_for<S::NS>([&](auto s_c) {
        constexpr int s = decltype(s_c)::value;
        constexpr int kx = s / S::SPC, si = s % S::SPC;
        constexpr int ky0 = S::ky0(si), nt = S::nt(si);
        // ---- this stage's weights (and, at s == 0, this chunk's tile) have landed; the previous stage's buffers are free.
        //      vmcnt counts in issue order: what may stay in flight is whatever this wave issued AFTER the pieces it needs now
        if (s == 0) {
          stamp(2);
          if (c == 0 && !first) c7_wait_barrier<NSTORE>();      // the previous block's / unit's epilogue stores
          else c7_wait_barrier<0>();
          stamp(3);
        } else if (s == 1 && tile_next) {
          c7_wait_barrier<PPW>();                               // the tile pieces issued behind this stage's weight pieces
        } else {
          c7_wait_barrier<0>();
        }
        // ---- next stage's weights, next tile
        if (s + 1 < S::NS) {
          const int s1 = s + 1;
          issue_wstage(cur.g, KS, c, blk, s1 / S::SPC, S::ky0(s1 % S::SPC), S::nt(s1 % S::SPC), sp ^ 1);
        } else if (!last_chunk) {
          issue_wstage(cur.g, KS, c + 1, blk, 0, S::ky0(0), S::nt(0), sp ^ 1);
        } else if (!last_blk) {
          issue_wstage(cur.g, KS, 0, blk + 1, 0, S::ky0(0), S::nt(0), sp ^ 1);
        } else if (has_next) {
          issue_wstage(nxt.g, nxt.ks, 0, 0, 0, 0, nt0_of(nxt.ks), sp ^ 1);
        }
        if (s == 0 && tile_next) {
          if (!last_chunk) issue_tile(cur, c + 1, tp ^ 1); else if (!last_blk) issue_tile(cur, 0, tp ^ 1); else issue_tile(nxt, 0, tp ^ 1);
        }
        // ---- MFMAs of the stage: per 16-channel k-step, the nt + MB - 1 input rows it touches, then per kernel row its weight fragment(s)
        if (!(a.dbg & 1)) {
          const int wb = sp * WBUF;
          const int row0 = MB * (tid >> 6) - P + ky0;           // image row of fragment j = 0 (per-lane arithmetic on purpose: the row bases then live in VGPRs, not in spilled SGPRs)
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2) {
            bf16x8 xf[7 + MB - 1];
#pragma unroll
            for (int j = 0; j < nt + MB - 1; ++j) {
              const int row = row0 + j;
              const int sb = ((unsigned)row < (unsigned)IMG) ? tbase + row * ROWB : ZROW;
              xf[j] = *reinterpret_cast<const bf16x8*>(lds + ((acol[kx + Q] ^ (k2 << 5)) + sb));
            }
#pragma unroll
            for (int i = 0; i < nt; ++i) {
              bf16x8 wf[CO];
#pragma unroll
              for (int b = 0; b < CO; ++b)
                wf[b] = *reinterpret_cast<const bf16x8*>(lds + wb + (i * CO + b) * 2048 + (wl ^ (k2 << 5)));
#pragma unroll
              for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int b = 0; b < CO; ++b)
                  acc[m][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[b], xf[m + i], acc[m][b], 0, 0, 0);
            }
          }
        }
        // schedule of the stage's block: a few fragment reads ahead, then one read per MFMA until the reads run out (left alone the compiler
        // puts every ds_read right in front of its first MFMA and a wave sits out the LDS latency once per tap)
        if ((CO == 1 || W16) && !(a.dbg & 16)) {                // (two output blocks on 32 x 32 maps: 128 accumulator registers, the interleave spills)
          constexpr int NRD = 2 * ((nt + MB - 1) + nt * CO), NMF = 2 * nt * MB * CO, LEAD = 4, R = NMF / NRD > 0 ? NMF / NRD : 1;
          __builtin_amdgcn_sched_group_barrier(0x100, LEAD < NRD ? LEAD : NRD, 0);
#pragma unroll
          for (int q = 0; q < NMF; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if ((q + 1) % R == 0 && LEAD + (q + 1) / R - 1 < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // (one read per R MFMAs: the reads end with the MFMAs)
          }
        }
        sp ^= 1;
      });
      if (!same_tile) tp ^= 1;
    }
    const int cobase = blk * 32 * CO;
    stamp(4);
    // ---- epilogue: y = alpha * acc + beta * res, 16-byte stores (register quads paired across the half-waves)
    if (!(a.dbg & 4)) {
      bf16* Y = (bf16*)a.y;
      const bf16* R = (const bf16*)a.res;
      const int nimg = W16 ? ((r >> 4) ? cur.n2 : cur.n) : cur.n;
      const bool live = !W16 || nimg >= 0;
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const long pix = W16 ? (((long)nimg * 16 + 2 * wave + m) * 16 + (r & 15)) * a.Cout
                             : (((long)nimg * 32 + 4 * wave + m) * 32 + r) * a.Cout;
#pragma unroll
        for (int b = 0; b < CO; ++b)
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = a.alpha * acc[m][b][8 * p + e];
            if (R && live) {
              const long o0 = pix + cobase + 32 * b + 16 * p + 4 * h;
              const bf16x4 r0 = *reinterpret_cast<const bf16x4*>(R + o0);
              const bf16x4 r1 = *reinterpret_cast<const bf16x4*>(R + o0 + 8);
#pragma unroll
              for (int e = 0; e < 4; ++e) { v[e] += a.beta * (float)r0[e]; v[4 + e] += a.beta * (float)r1[e]; }
            }
            typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
            const unsigned A0 = __builtin_bit_cast(unsigned, (bf2){(bf16)v[0], (bf16)v[1]}), A1 = __builtin_bit_cast(unsigned, (bf2){(bf16)v[2], (bf16)v[3]});
            const unsigned B0 = __builtin_bit_cast(unsigned, (bf2){(bf16)v[4], (bf16)v[5]}), B1 = __builtin_bit_cast(unsigned, (bf2){(bf16)v[6], (bf16)v[7]});
            const u32x2 s0 = __builtin_amdgcn_permlane32_swap(A0, B0, false, false);
            const u32x2 s1 = __builtin_amdgcn_permlane32_swap(A1, B1, false, false);
            if (live) *reinterpret_cast<uint4*>(Y + pix + cobase + 32 * b + 16 * p + 8 * h) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
          }
      }
    } else {
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int b = 0; b < CO; ++b) asm volatile("" :: "v"(acc[m][b]));
    }
    stamp(5);
    first = false;
    }                                                           // output blocks
  };

  while (true) {
    if ((KMASK & 4) && cur.ks == 7) run_unit(std::integral_constant<int, 7>{});
    else if ((KMASK & 2) && cur.ks == 5) run_unit(std::integral_constant<int, 5>{});
    else run_unit(std::integral_constant<int, 3>{});
    first = false;
    if (!has_next) break;
    cur = nxt;
    ++q;
    has_next = unit_at(q + 1, nxt);
  }
#endif
}

}  // namespace

struct ConvArgs;
struct C7Plan { C7Args a; unsigned G; int CO, kmask, w16; size_t lds; };
// Launch geometry of conv7 for one layer (conv7.hip; shared with the fused backward launch).  0 = planned, 1 = outside conv7's domain.
int conv7_plan(const ConvArgs& c, int dtype, C7Plan& plan);
void conv7_launch(const C7Plan& p, hipStream_t stream);
