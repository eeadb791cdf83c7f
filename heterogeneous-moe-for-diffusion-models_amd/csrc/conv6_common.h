// Shared by conv6.hip (bf16 kernel) and conv6s.hip (split-bf16 kernel for fp32 tensors): launch arguments and the work-unit record.
#pragma once
#include "conv_args.h"

namespace {

typedef __attribute__((address_space(3))) void* lptr_t;

struct C6Args {
  const void* x; const void* w; void* y; const void* res; const int* seg;
  long wstride;                       // elements per group in the weight image [g][tap][Cout][Cin]
  int N, H, W, Cin, Cout, ngroups;
  int ks[HDMOE_MAX_GROUPS], pt[HDMOE_MAX_GROUPS], pl[HDMOE_MAX_GROUPS], order[HDMOE_MAX_GROUPS];
  float alpha, beta;
  int TH, TW, tws, tiles_x, tpi;      // tile geometry (TW = 1 << tws), tiles per image
  int T;                              // taps per weight stage
  int nblk;                           // output-channel blocks
  int hb_bytes, wb_bytes;             // bytes of one halo buffer / one weight buffer
  int xbytes, wbytes;                 // extents of x and of the weight image (buffer descriptors; < 4 GB)
  unsigned ybytes;                    // extent of y for buffer-descriptor stores with a counted drain (conv6_body.h epilogue); 0: plain stores
  unsigned m_nblk, m_T, m_tpi, m_tx;  // 2^32 / d + 1 reciprocals of nblk, T, tpi, tiles_x
  int w_rowpitch, w_tapstride;        // weight image geometry in elements: between consecutive output rows / consecutive taps (default Cin, Cout * Cin);
                                      // larger values read a [tap][rows][pitch] image with more channels / rows than this conv uses
                                      // (the ones-channel layer of Unet_expert: hdmoe_conv6_ones_fwd / _bwd)
  const float* gbias;                 // optional fp32 [group][H][W][Cout]: y = alpha * (acc + gbias[g][yy][xx][:]) + beta * res
  int dbg;                            // development ablations: 1 skip the MFMA loop, 2 skip the in-loop DMA, 4 skip the stores
  unsigned long long* stamps;         // development: s_memtime stamps of workgroup 0 ([wave][64] slots), or null
  // fused FiLM epilogue (Unet_block, reference model_components.py:242-246): besides y the kernel writes
  // film_h = dropout_p(mp_silu(y * film_e[n][c])) -- the same arithmetic and the same Philox bits as film_silu_fwd_vec_kernel
  // (elementwise.hip) on the bf16-rounded y, so the separate pass (and its launch on the U-Net branch's serial chain) disappears.
  const float* film_e; void* film_h; const unsigned long long* film_seed_dev; unsigned film_seed_lo, film_seed_hi; float film_p;
};

template <int MT>
struct C6Unit {                       // one work unit: MT 256-pixel tiles of one expert x one output-channel block (all wave-uniform)
  int g, ks, ntaps, ntg, pt, pl, HWp, HHp, ppt, nbk;
  int n[MT], ty0[MT], tx0[MT], valid[MT];
};

constexpr int C6_MAXT = 9;            // taps per weight stage (<= 9: the stage's tap loop is fully unrolled)
constexpr int C6_NW = 8;              // waves per workgroup: two per SIMD.  (One per SIMD with twice the tile per wave measured 20 % slower: a lone wave
                                      // issues its ~4.5 LDS / VALU / scalar instructions per MFMA in the open, a partner wave hides them.)


}  // namespace

struct ConvArgs;
struct C6Plan { C6Args a; int MT, NT; unsigned G; size_t lds; };
// Launch geometry of conv6 for one layer (conv6.hip).  0 = planned, 1 = outside conv6's domain.
int conv6_plan(const ConvArgs& c, int dtype, C6Plan& plan);
