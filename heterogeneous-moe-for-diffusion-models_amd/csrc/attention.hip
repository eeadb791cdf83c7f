// K5: multi-head attention core of MP_Attention (reference models/model_internals.py:374-404), flash
// style: the S_q x S_kv score map is never materialised (the reference allocates B x H x S x S).
//   q [B][Sq][E], k/v [B][Skv][E], head h = channels [h*D, (h+1)*D), scores/sqrt(D) (+ rel_pos_bias[h][i][j]),
//   softmax over kv, out [B][Sq][E]; lse [B][H][Sq] is kept for the backward.
// Head dim is tiny here (D = E/H = 4): QK^T has K = 4, so this is exp/VALU work, not MFMA work.  One thread owns
// one query row (fwd, dq) or one key row (dk/dv); the other side streams through LDS in fp32 tiles and is
// read by every lane at the same address (LDS broadcast, conflict-free).
#include <stdlib.h>
#include "common.h"
#include "hdmoe.h"

namespace {

constexpr int TQ = 256;     // rows owned by a block (one per thread)
constexpr int TK = 128;     // streamed rows per LDS tile
constexpr int CH = 16;      // online-softmax chunk

template <typename T, int D, bool BIAS>
__global__ __launch_bounds__(TQ) void attn_fwd_kernel(T* out, float* lse, const T* q, const T* k, const T* v, const float* bias,
                                                     int Sq, int Skv, int H, int Sb, float scale) {
  __shared__ float sk[TK * D], sv[TK * D];
  const int b = blockIdx.z, h = blockIdx.y, i = blockIdx.x * TQ + threadIdx.x;
  const int E = H * D;
  const bool act = i < Sq;
  float qv[D], o[D];
#pragma unroll
  for (int d = 0; d < D; ++d) { qv[d] = act ? to_f(q[((long)b * Sq + i) * E + h * D + d]) * scale : 0.f; o[d] = 0.f; }
  float m = -INFINITY, l = 0.f;
  const float* brow = (BIAS && act) ? bias + ((long)h * Sb + i) * Sb : nullptr;
  for (int j0 = 0; j0 < Skv; j0 += TK) {
    const int nj = min(TK, Skv - j0);
    __syncthreads();
    for (int e = threadIdx.x; e < nj * D; e += TQ) {
      const long src = ((long)b * Skv + j0 + e / D) * E + h * D + e % D;
      sk[e] = to_f(k[src]); sv[e] = to_f(v[src]);
    }
    __syncthreads();
    if (!act) continue;
    for (int c0 = 0; c0 < nj; c0 += CH) {
      const int nc = min(CH, nj - c0);
      float s[CH];
      float cm = -INFINITY;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        float a = -INFINITY;
        if (c < nc) {
          a = 0.f;
#pragma unroll
          for (int d = 0; d < D; ++d) a += qv[d] * sk[(c0 + c) * D + d];
          if (BIAS) a += brow[j0 + c0 + c];
        }
        s[c] = a; cm = fmaxf(cm, a);
      }
      const float mn = fmaxf(m, cm);
      const float corr = __expf(m - mn);        // m = -inf on the first chunk -> 0
      l *= corr;
#pragma unroll
      for (int d = 0; d < D; ++d) o[d] *= corr;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (c < nc) {
          const float p = __expf(s[c] - mn);
          l += p;
#pragma unroll
          for (int d = 0; d < D; ++d) o[d] += p * sv[(c0 + c) * D + d];
        }
      }
      m = mn;
    }
  }
  if (act) {
    const float il = 1.f / l;
#pragma unroll
    for (int d = 0; d < D; ++d) out[((long)b * Sq + i) * E + h * D + d] = from_f<T>(o[d] * il);
    lse[((long)b * H + h) * Sq + i] = m + __logf(l);
  }
}

// dq (+ delta, + dbias): thread = query row
template <typename T, int D, bool BIAS>
__global__ __launch_bounds__(TQ) void attn_bwd_dq_kernel(T* dq, float* delta, float* dbias, const T* dout, const T* out, const T* q,
                                                        const T* k, const T* v, const float* lse, const float* bias, int Sq, int Skv,
                                                        int H, int Sb, float scale) {
  __shared__ float sk[TK * D], sv[TK * D];
  const int b = blockIdx.z, h = blockIdx.y, i = blockIdx.x * TQ + threadIdx.x;
  const int E = H * D;
  const bool act = i < Sq;
  float qv[D], dov[D], acc[D];
  float dl = 0.f, ls = 0.f;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const long idx = ((long)b * Sq + i) * E + h * D + d;
    qv[d] = act ? to_f(q[idx]) * scale : 0.f;
    dov[d] = act ? to_f(dout[idx]) : 0.f;
    if (act) dl += dov[d] * to_f(out[idx]);
    acc[d] = 0.f;
  }
  if (act) { ls = lse[((long)b * H + h) * Sq + i]; delta[((long)b * H + h) * Sq + i] = dl; }
  const long boff = ((long)h * Sb + i) * Sb;
  for (int j0 = 0; j0 < Skv; j0 += TK) {
    const int nj = min(TK, Skv - j0);
    __syncthreads();
    for (int e = threadIdx.x; e < nj * D; e += TQ) {
      const long src = ((long)b * Skv + j0 + e / D) * E + h * D + e % D;
      sk[e] = to_f(k[src]); sv[e] = to_f(v[src]);
    }
    __syncthreads();
    if (!act) continue;
#pragma unroll 8
    for (int j = 0; j < nj; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) { s += qv[d] * sk[j * D + d]; dp += dov[d] * sv[j * D + d]; }
      if (BIAS) s += bias[boff + j0 + j];
      const float ds = __expf(s - ls) * (dp - dl);
#pragma unroll
      for (int d = 0; d < D; ++d) acc[d] += ds * sk[j * D + d];
    }
  }
  if (act) {
#pragma unroll
    for (int d = 0; d < D; ++d) dq[((long)b * Sq + i) * E + h * D + d] = from_f<T>(acc[d] * scale);
  }
}

// dk, dv: thread = key row; queries stream through LDS
template <typename T, int D, bool BIAS>
__global__ __launch_bounds__(TQ) void attn_bwd_dkv_kernel(T* dk, T* dv, const T* dout, const T* q, const T* k, const T* v,
                                                         const float* lse, const float* delta, const float* bias, int Sq, int Skv,
                                                         int H, int Sb, float scale) {
  __shared__ float sq[TK * D], sdo[TK * D], sl[TK], sd[TK];
  const int b = blockIdx.z, h = blockIdx.y, j = blockIdx.x * TQ + threadIdx.x;
  const int E = H * D;
  const bool act = j < Skv;
  float kv[D], vv[D], ak[D], av[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const long idx = ((long)b * Skv + j) * E + h * D + d;
    kv[d] = act ? to_f(k[idx]) : 0.f;
    vv[d] = act ? to_f(v[idx]) : 0.f;
    ak[d] = 0.f; av[d] = 0.f;
  }
  for (int i0 = 0; i0 < Sq; i0 += TK) {
    const int ni = min(TK, Sq - i0);
    __syncthreads();
    for (int e = threadIdx.x; e < ni * D; e += TQ) {
      const long src = ((long)b * Sq + i0 + e / D) * E + h * D + e % D;
      sq[e] = to_f(q[src]) * scale; sdo[e] = to_f(dout[src]);
    }
    for (int e = threadIdx.x; e < ni; e += TQ) {
      sl[e] = lse[((long)b * H + h) * Sq + i0 + e];
      sd[e] = delta[((long)b * H + h) * Sq + i0 + e];
    }
    __syncthreads();
    if (!act) continue;
#pragma unroll 8
    for (int i = 0; i < ni; ++i) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) { s += sq[i * D + d] * kv[d]; dp += sdo[i * D + d] * vv[d]; }
      if (BIAS) s += bias[((long)h * Sb + i0 + i) * Sb + j];
      const float p = __expf(s - sl[i]);
      const float ds = p * (dp - sd[i]);
#pragma unroll
      for (int d = 0; d < D; ++d) { av[d] += p * sdo[i * D + d]; ak[d] += ds * sq[i * D + d]; }
    }
  }
  if (act) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const long idx = ((long)b * Skv + j) * E + h * D + d;
      dk[idx] = from_f<T>(ak[d]);           // sq already carries `scale`
      dv[idx] = from_f<T>(av[d]);
    }
  }
}

// d(rel_pos_bias)[h][i][j] = sum_b ds[b][h][i][j].  One thread per (h, i, j) walks the batch (no atomics: with B samples
// adding into the same H*S*S words the atomic version was 256-way contended and cost ~100 us per call on S = 64).
template <typename T, int D>
__global__ void attn_dbias_kernel(float* dbias, const T* dout, const T* q, const T* k, const T* v, const float* lse,
                                  const float* delta, const float* bias, int B, int Sq, int Skv, int H, int Sb, float scale) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)H * Sq * Skv) return;
  const int j = (int)(idx % Skv);
  const int i = (int)((idx / Skv) % Sq);
  const int h = (int)(idx / ((long)Skv * Sq));
  const int E = H * D;
  const float bv = bias[((long)h * Sb + i) * Sb + j];
  float acc = 0.f;
  for (int b = 0; b < B; ++b) {
    const T* qp = q + ((long)b * Sq + i) * E + h * D;
    const T* dop = dout + ((long)b * Sq + i) * E + h * D;
    const T* kp = k + ((long)b * Skv + j) * E + h * D;
    const T* vp = v + ((long)b * Skv + j) * E + h * D;
    float s = 0.f, dp = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) { s += to_f(qp[d]) * to_f(kp[d]); dp += to_f(dop[d]) * to_f(vp[d]); }
    const long li = ((long)b * H + h) * Sq + i;
    acc += __expf(s * scale + bv - lse[li]) * (dp - delta[li]);
  }
  dbias[((long)h * Sb + i) * Sb + j] = acc;
}

// =====================================================================================================================
// MFMA path: bf16, D = 4, no positional bias (the fusion cross-attention S = 1024 and the text cross-attention S_kv = 77).
// The one-row-per-thread kernels above spend ~15 VALU instructions per (q, k) pair (4 cycles each on a wave64); here the
// three contractions of a 32 x 32 tile run on the matrix pipe with the head dim zero-padded 4 -> 16 and only exp2 and the
// bf16 conversion of the probabilities stay on the vector pipe, which is the bound (v_exp_f32 issues every 8 cycles):
//   S^T[k][q]  = K[k][:] . Q[q][:]            A = K rows (LDS),        B = Q (registers, fixed per wave)
//   O^T[d][q] += V^T[d][k] * P^T[k][q]        A = V^T (LDS, permuted), B = P straight from the S^T accumulator registers
// An accumulator lane (c, hh) holds column q = c and rows k = (reg&3) + 8(reg>>2) + 4hh, i.e. for the 16-deep slice s of the
// second product its registers 8s..8s+7 ARE a B-operand fragment if K-slot (hh, j) is defined as k = 16s + (j&3) + 8(j>>2) + 4hh;
// the A operand (V^T, or K^T / Q^T / dO^T in the backward kernels) comes out of LDS in exactly that slot order through the
// transposing read, so no cross-lane movement is needed.  Row 4 of the V^T tile is all ones: O^T row 4 accumulates the softmax
// denominator on the matrix pipe for free.
//
// The 12 padding slots of the first product carry the softmax's scalar arithmetic (round 3; before, each score cost a v_fma for
// the scale and the shift, the online maximum 15 v_max + a cross-lane step per tile, and the backward a v_sub per score):
//   slots 0-3   K side k_d        Q side hi(c * q_d)        c = log2(e) / sqrt(D): the product is the score in log2 units,
//   slots 4-7   K side k_d        Q side lo(c * q_d)        hi + lo = 16 significant bits of c * q_d
//   slots 8-10  K side 1          Q side -shift (a 1-3 term bf16 expansion)
// so the accumulator IS  c * q.k - shift  and exp2 applies to it directly.  Forward: shift = the row maximum over the FIRST key tile,
// rounded to bf16 (any shift gives the same softmax; lse = shift + log2(sum)); if a later key beats it by more than 2^127 -- or all
// of them fall below 2^-126 of it -- the sums come out non-finite or zero, which the wave detects at the end of the head and then
// repeats the sweep with the online maximum (`attn_fwd_sweep<false>`), so the result never depends on the heuristic.  Backward:
// shift = the saved lse (p = exp2(S') exactly as the forward normalised it), and the dP product carries -delta the same way.
//
// Work split (round 3): a workgroup = 8 waves = 8 consecutive 32-row blocks of one sample; the heads are taken ONE AFTER THE OTHER and
// for each head the whole other side (up to ATT_HS = 1024 rows of it) is staged in LDS as the raw 8-byte rows [x0 x1 x2 x3] --
// double-buffered, so a head costs one barrier and the 32-tile sweep of a wave has none.  Before, a workgroup was one 32-row block x
// 8 heads with a stage of 64 rows of all heads between two barriers: the barriers (cooperative whole-row loads) or, with wave-private
// staging, the 8-byte loads 64 B apart (one L1 line pass per lane) cost 220 of the forward kernel's 350 us; staging per head moves
// the same strided loads but 8x fewer of them, off the sweep.  What the 16-byte padded rows of the old image supplied -- the copy of
// k for the hi / lo pair, the 1 of V's fifth column, zeros -- now comes from registers (a duplicated 8-byte read) and from a constant
// 256-byte block the lanes of the transposing read that address columns 4..15 point to instead of the row.
#ifndef HDMOE_ATTN_DBG
#define HDMOE_ATTN_DBG 0
#endif
constexpr int ATT_HS = 1024;                                 // rows of the streamed side staged per head image
constexpr int ATT_IMG = ATT_HS * 8;                          // bytes of one raw head image
constexpr int ATT_CONST = 1024;                              // constant block, period 64 B: 8-byte word 0 = [1 0 0 0], word 2 = [1 1 1 0], else zeros
constexpr int CB_ONE = 0, CB_ZERO = 8, CB_ONE3 = 16;
DEVI f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
DEVI bf16x8 pack8(const f32x16& v, int s) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16)v[8 * s + j];
  return o;
}
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
DEVI bf16x8 frag_of(unsigned a, unsigned b, unsigned c2, unsigned d) { return __builtin_bit_cast(bf16x8, (u32x4){a, b, c2, d}); }
// fragment [x0 x1 x2 x3 0 0 0 0] (head dim 4 padded to the 16-deep K of the MFMA; the caller keeps it on lanes hh == 0)
DEVI bf16x8 head_frag(const bf16* row, bool valid) {
  bf16x8 f = (bf16x8)(0);
  if (valid) {
    const bf16x4 t = *reinterpret_cast<const bf16x4*>(row);
    f[0] = t[0]; f[1] = t[1]; f[2] = t[2]; f[3] = t[3];
  }
  return f;
}
// fragment [hi(c x0..x3) | lo(c x0..x3)]
DEVI bf16x8 scaled_frag(const bf16* row, bool valid, float c) {
  bf16x8 f = (bf16x8)(0);
  if (valid) {
    const bf16x4 t = *reinterpret_cast<const bf16x4*>(row);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const float x = (float)t[d] * c;
      f[d] = (bf16)x;
      f[4 + d] = (bf16)(x - (float)f[d]);
    }
  }
  return f;
}
// slots j0 .. j0+2 = a three-term bf16 expansion of x (exact to fp32), the other slots zero
DEVI bf16x8 shift_frag(float x, int j0) {
  bf16x8 f = (bf16x8)(0);
  const bf16 a = (bf16)x;
  const float r1 = x - (float)a;
  const bf16 b2 = (bf16)r1;
  f[j0] = a; f[j0 + 1] = b2; f[j0 + 2] = (bf16)(r1 - (float)b2);
  return f;
}
DEVI bf16x8 ones_frag(int j0, int n) {                        // slots j0 .. j0+n-1 = 1
  bf16x8 f = (bf16x8)(0);
#pragma unroll
  for (int j = 0; j < 8; ++j) if (j >= j0 && j < j0 + n) f[j] = (bf16)1.f;
  return f;
}
DEVI bool finite_f(float x) { return fabsf(x) <= 3.0e38f; }
DEVI void const_block(unsigned char* cb, int tid) {           // (first ATT_CONST / 8 threads)
  if (tid < ATT_CONST / 8) {
    const int w = tid & 7;
    *reinterpret_cast<u32x2*>(cb + tid * 8) = (u32x2){w == 0 ? 0x3F80u : (w == 2 ? 0x3F803F80u : 0u), w == 2 ? 0x3F80u : 0u};
  }
}
// cooperative staging of one head of `src` [rows][E]: rows [row0, row0 + n) -> raw image, two rows per thread (512 threads).
struct HeadRegs { uint2 v[ATT_HS / 512]; };
DEVI void head_load(HeadRegs& rg, const bf16* src, long row0, int n, int E, int h, int tid) {
#pragma unroll
  for (int i = 0; i < ATT_HS / 512; ++i) {
    const int rr = tid + i * 512;
    rg.v[i] = rr < n ? *reinterpret_cast<const uint2*>(src + (row0 + rr) * (long)E + h * 4) : make_uint2(0, 0);
  }
}
DEVI void head_write(const HeadRegs& rg, unsigned char* img, int n, int tid) {
#pragma unroll
  for (int i = 0; i < ATT_HS / 512; ++i) {
    const int rr = tid + i * 512;
    if (rr < ((n + 31) & ~31)) *reinterpret_cast<uint2*>(img + rr * 8) = rg.v[i];      // (whole tiles: rows past n are zeros)
  }
}
// A-operand fragment of X^T (rows = head-dim index, 8 K-slots of slice s) straight from the row image with the hardware
// transposing read (cdna_hip_programming.md T10): within a 16-lane group lane 4q+p addresses row q, columns 4p..4p+3 and lane i
// receives column i of the four rows.  Lane (r, hh) so gets X[16s + j' + 8*(second read) + 4hh][r] -- the K-slot order of the
// accumulator registers.  Lanes p == 0 address the row (8 bytes = columns 0..3), lanes p >= 1 the constant block (column 4 = 1 for V
// in the forward kernel, everything else 0): tr_ptr().  +64 B = 8 rows for the second read, +128 B the second 16-row slice, +256 B
// the next tile; the constant block repeats every 64 B over 1 KB, so the same immediate offsets serve both kinds of lane for four
// tiles, after which a per-lane step (1 KB for row lanes, 0 for constant lanes) moves the pointers: the sweep's address arithmetic is
// one v_add per pointer per FOUR tiles, and the operand halves that are constants (the 1 that meets the shift, zero padding) cost
// LDS reads instead of vector instructions -- the vector pipe is the bound (SQ_ACTIVE_INST_VALU 68 % of the SIMD cycles).
DEVI int tr_row(int lane) { return ((lane & 15) >> 2) + 4 * (lane >> 5); }
DEVI const unsigned char* tr_ptr(const unsigned char* img, const unsigned char* cb, int lane, bool ones) {
  const int p = lane & 3;
  return p == 0 ? img + tr_row(lane) * 8 : cb + ((p == 1 && ones) ? CB_ONE : CB_ZERO);
}
DEVI bf16x8 tr_frag(const unsigned char* p) {
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p + 64));
  return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}
DEVI u32x2 row8(const unsigned char* p) { return *reinterpret_cast<const u32x2*>(p); }
DEVI bf16x8 frag2(const unsigned char* lo, const unsigned char* hi) {   // two 8-byte reads = one fragment
  const u32x2 a = row8(lo), b = row8(hi);
  return frag_of(a.x, a.y, b.x, b.y);
}

// One sweep of a wave's 32 queries of one head over the staged keys [0, n).  fq: lanes hh == 0 [hi(c q) | lo(c q)], lanes hh == 1 zeros.
// FAST: fixed shift (see the header); else the online maximum.  o / m carry over between key chunks (first = chunk 0 of the head).
// Returns O^T (rows 0-3 on lanes hh == 0, the denominator = row 4 on lanes hh == 1, register 0) and the shift m in log2 units.
template <bool FAST>
DEVI void attn_fwd_sweep(f32x16& o, float& m, bf16x8& fqs, const bool first, const unsigned char* imgK, const unsigned char* imgV,
                         const unsigned char* cb, const bf16x8 fq, const int n, const int lane) {
  const int r = lane & 31, hh = lane >> 5;
  // K fragment = [k | k] on lanes hh == 0, [1 0 0 0 | 0] on lanes hh == 1 (slot 8 meets the shift): two 8-byte reads either way
  const unsigned char* aK0 = hh == 0 ? imgK + r * 8 : cb + CB_ONE;
  const unsigned char* aK1 = hh == 0 ? imgK + r * 8 : cb + CB_ZERO;
  const unsigned char* aV = tr_ptr(imgV, cb, lane, true);
  const int kstep = hh == 0 ? 256 : 0, vstep = (lane & 3) == 0 ? 256 : 0;
  if (first) {
    o = (f32x16)(0.f);
    m = FAST ? 0.f : -INFINITY;
    fqs = fq;
    if (FAST) {                                              // the shift: this query's maximum over the first 32 keys, as a bf16 number
      f32x16 s = mfma_bf16(frag2(aK0, aK1), fq, (f32x16)(0.f));
      if (n < 32) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          if (acc_row(reg, lane) >= n) s[reg] = -INFINITY;
      }
      float tm = s[0];
#pragma unroll
      for (int reg = 1; reg < 16; ++reg) tm = fmaxf(tm, s[reg]);
      tm = fmaxf(tm, __shfl_xor(tm, 32, 64));                // the other half-wave holds the other 16 keys of these queries
      const bf16 sh = (bf16)(-tm);
      m = -(float)sh;
      if (hh == 1) fqs[0] = sh;
    }
  }
  auto tile = [&](const int off, const int kb, const bool partial) {
    const bf16x8 fk = frag2(aK0 + off, aK1 + off);
    const bf16x8 fv0 = tr_frag(aV + off);
    const bf16x8 fv1 = tr_frag(aV + off + 128);
    f32x16 s = mfma_bf16(fk, fqs, (f32x16)(0.f));            // rows = keys, columns = queries; log2 units, minus the shift if FAST
    if (partial) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        if (kb + acc_row(reg, lane) >= n) s[reg] = -INFINITY;
    }
    if (FAST) {
#if HDMOE_ATTN_DBG != 1 && HDMOE_ATTN_DBG != 3
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) s[reg] = __builtin_amdgcn_exp2f(s[reg]);
#endif
    } else {
      float tm = s[0];
#pragma unroll
      for (int reg = 1; reg < 16; ++reg) tm = fmaxf(tm, s[reg]);
      tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
      const float mn = fmaxf(m, tm);
      const float alpha = __builtin_amdgcn_exp2f(m - mn);
      m = mn;
      o[0] *= alpha; o[1] *= alpha; o[2] *= alpha; o[3] *= alpha;       // rows 0-3 (hh = 0) and the denominator row 4 (hh = 1)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) s[reg] = __builtin_amdgcn_exp2f(s[reg] - mn);
    }
#if HDMOE_ATTN_DBG == 2 || HDMOE_ATTN_DBG == 3
    o[0] += s[0] + s[5] + s[10] + s[15]; o[1] += s[1] + s[4] + s[11] + s[14]; o[2] += s[2] + s[7] + s[8] + s[13]; o[3] += s[3] + s[6] + s[9] + s[12];
#else
    o = mfma_bf16(fv0, pack8(s, 0), o);
    o = mfma_bf16(fv1, pack8(s, 1), o);
#endif
  };
  int kb = 0;
  for (; kb + 128 <= n; kb += 128) {                          // four full tiles per pointer step
    tile(0, kb, false); tile(256, kb + 32, false); tile(512, kb + 64, false); tile(768, kb + 96, false);
    aK0 += 4 * kstep; aK1 += 4 * kstep; aV += 4 * vstep;
  }
  for (; kb < n; kb += 32) {
    tile(0, kb, kb + 32 > n);
    aK0 += kstep; aK1 += kstep; aV += vstep;
  }
}

// grid (ceil(Sq / 256), B), 512 threads: wave w = queries [blockIdx.x * 256 + 32 w, +32) of sample blockIdx.y, all heads in turn.
__global__ __launch_bounds__(512) void attn_fwd_mfma_kernel(bf16* out, float* lse, const bf16* q, const bf16* k, const bf16* v, int Sq,
                                                           int Skv, int H, float c, int force_slow) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_attn[];
  unsigned char* cb = smem_attn + 4 * ATT_IMG;                // [2 buffers][K | V] raw images, then the constant block
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int b = blockIdx.y, q0 = blockIdx.x * 256 + wave * 32, E = H * 4;
  const bool qok = q0 + r < Sq;
  const long qrow = ((long)b * Sq + q0 + (qok ? r : 0)) * E;
  const int nch = (Skv + ATT_HS - 1) / ATT_HS, nst = H * nch;
  const bool slow = force_slow != 0 || nch > 1;               // (the repeat-on-overflow needs the head's keys in ONE image)
  const_block(cb, tid);
  HeadRegs rk, rv;
  head_load(rk, k, (long)b * Skv, min(ATT_HS, Skv), E, 0, tid);
  head_load(rv, v, (long)b * Skv, min(ATT_HS, Skv), E, 0, tid);
  head_write(rk, smem_attn, min(ATT_HS, Skv), tid);
  head_write(rv, smem_attn + ATT_IMG, min(ATT_HS, Skv), tid);
  __syncthreads();
  f32x16 o;
  float m;
  bf16x8 fq, fqs;
  for (int st = 0; st < nst; ++st) {
    const int h = st / nch, ch = st - h * nch, n = min(ATT_HS, Skv - ch * ATT_HS);
    const unsigned char* imgK = smem_attn + (st & 1) * 2 * ATT_IMG;
    const unsigned char* imgV = imgK + ATT_IMG;
    int nn = 0;
    if (st + 1 < nst) {                                       // the next head / chunk: global -> registers beside this sweep
      const int h2 = (st + 1) / nch, ch2 = st + 1 - h2 * nch;
      nn = min(ATT_HS, Skv - ch2 * ATT_HS);
      head_load(rk, k, (long)b * Skv + ch2 * ATT_HS, nn, E, h2, tid);
      head_load(rv, v, (long)b * Skv + ch2 * ATT_HS, nn, E, h2, tid);
    }
    if (ch == 0) fq = scaled_frag(q + qrow + h * 4, hh == 0 && qok, c);
    if (slow) attn_fwd_sweep<false>(o, m, fqs, ch == 0, imgK, imgV, cb, fq, n, lane);
    else {
      attn_fwd_sweep<true>(o, m, fqs, true, imgK, imgV, cb, fq, n, lane);
      const float l0 = __shfl_xor(o[0], 32, 64);
      const bool bad = HDMOE_ATTN_DBG == 0 && hh == 0 && !(finite_f(o[0]) && finite_f(o[1]) && finite_f(o[2]) && finite_f(o[3]) && finite_f(l0) && l0 > 0.f);
      if (__builtin_amdgcn_ballot_w64(bad) != 0)              // (never on the model's activations; tests force it)
        attn_fwd_sweep<false>(o, m, fqs, true, imgK, imgV, cb, fq, n, lane);
    }
    if (ch == nch - 1) {
      const float l = __shfl_xor(o[0], 32, 64);               // row 4 of O^T (lane + 32, register 0) = sum of the probabilities
      if (hh == 0 && qok) {
        const float il = 1.f / l;
        bf16x4 ov;
        ov[0] = (bf16)(o[0] * il); ov[1] = (bf16)(o[1] * il); ov[2] = (bf16)(o[2] * il); ov[3] = (bf16)(o[3] * il);
        *reinterpret_cast<bf16x4*>(out + qrow + h * 4) = ov;
        lse[((long)b * H + h) * Sq + q0 + r] = (m + __builtin_amdgcn_logf(l)) * 0.6931471805599453f;
      }
    }
    if (st + 1 < nst) {
      unsigned char* nxt = smem_attn + ((st + 1) & 1) * 2 * ATT_IMG;   // last read in stage st - 1: every wave has passed that stage's barrier
      head_write(rk, nxt, nn, tid);
      head_write(rv, nxt + ATT_IMG, nn, tid);
      __syncthreads();
    }
  }
}

// dq (+ delta): wave = 32 queries, heads in turn; the head's keys / values staged as in the forward kernel.
__global__ __launch_bounds__(512) void attn_bwd_dq_mfma_kernel(bf16* dq, float* delta, const bf16* dout, const bf16* out, const bf16* q,
                                                              const bf16* k, const bf16* v, const float* lse, int Sq, int Skv, int H,
                                                              float scale, float c) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_attn[];
  unsigned char* cb = smem_attn + 4 * ATT_IMG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int b = blockIdx.y, q0 = blockIdx.x * 256 + wave * 32, E = H * 4;
  const bool qok = q0 + r < Sq;
  const long qrow = ((long)b * Sq + q0 + (qok ? r : 0)) * E;
  const int nch = (Skv + ATT_HS - 1) / ATT_HS, nst = H * nch;
  const_block(cb, tid);
  HeadRegs rk, rv;
  head_load(rk, k, (long)b * Skv, min(ATT_HS, Skv), E, 0, tid);
  head_load(rv, v, (long)b * Skv, min(ATT_HS, Skv), E, 0, tid);
  head_write(rk, smem_attn, min(ATT_HS, Skv), tid);
  head_write(rv, smem_attn + ATT_IMG, min(ATT_HS, Skv), tid);
  __syncthreads();
  const int kstep = (lane & 3) == 0 ? 256 : 0, rstep = hh == 0 ? 256 : 0;
  f32x16 acc;
  bf16x8 fq, fdo;
  for (int st = 0; st < nst; ++st) {
    const int h = st / nch, ch = st - h * nch, n = min(ATT_HS, Skv - ch * ATT_HS);
    const unsigned char* imgK = smem_attn + (st & 1) * 2 * ATT_IMG;
    const unsigned char* imgV = imgK + ATT_IMG;
    int nn = 0;
    if (st + 1 < nst) {
      const int h2 = (st + 1) / nch, ch2 = st + 1 - h2 * nch;
      nn = min(ATT_HS, Skv - ch2 * ATT_HS);
      head_load(rk, k, (long)b * Skv + ch2 * ATT_HS, nn, E, h2, tid);
      head_load(rv, v, (long)b * Skv + ch2 * ATT_HS, nn, E, h2, tid);
    }
    if (ch == 0) {
      const long qoff = qrow + h * 4;
      float dl = 0.f, ls = 0.f;
      if (qok) {
        const bf16x4 a = *reinterpret_cast<const bf16x4*>(dout + qoff), o4 = *reinterpret_cast<const bf16x4*>(out + qoff);
        dl = (float)a[0] * (float)o4[0] + (float)a[1] * (float)o4[1] + (float)a[2] * (float)o4[2] + (float)a[3] * (float)o4[3];
        ls = lse[((long)b * H + h) * Sq + q0 + r] * 1.4426950408889634f;
        if (hh == 0) delta[((long)b * H + h) * Sq + q0 + r] = dl;
      }
      // S'^T = c q.k - lse (log2 units) and dP'^T = V.dO - delta come out of the matrix pipe: slots 8-10 carry the two shifts
      fq = hh == 0 ? scaled_frag(q + qoff, qok, c) : shift_frag(-ls, 0);
      fdo = hh == 0 ? head_frag(dout + qoff, qok) : shift_frag(-dl, 0);
      acc = (f32x16)(0.f);
    }
    // K fragment [k | k] / V fragment [v | 0] on lanes hh == 0, [1 1 1 0 | 0] on lanes hh == 1 (slots 8-10 meet the shifts)
    const unsigned char* aK0 = hh == 0 ? imgK + r * 8 : cb + CB_ONE3;
    const unsigned char* aK1 = hh == 0 ? imgK + r * 8 : cb + CB_ZERO;
    const unsigned char* aV0 = hh == 0 ? imgV + r * 8 : cb + CB_ONE3;
    const unsigned char* aZ = cb + CB_ZERO;
    const unsigned char* aKt = tr_ptr(imgK, cb, lane, false);
    auto tile = [&](const int off) {
      const bf16x8 fk = frag2(aK0 + off, aK1 + off);
      const bf16x8 fv = frag2(aV0 + off, aZ + off);
      const bf16x8 fk0 = tr_frag(aKt + off);
      const bf16x8 fk1 = tr_frag(aKt + off + 128);
      f32x16 s = mfma_bf16(fk, fq, (f32x16)(0.f));           // S'^T[k][q]
      const f32x16 dp = mfma_bf16(fv, fdo, (f32x16)(0.f));   // dP'^T[k][q] = V[k] . dO[q] - delta[q]
      // keys past Skv are zero rows: their ds is finite and meets a zero K^T column, so no masking is needed
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) s[reg] = __builtin_amdgcn_exp2f(s[reg]) * dp[reg];
      acc = mfma_bf16(fk0, pack8(s, 0), acc);                // dQ^T[d][q] += K^T[d][k] dS^T[k][q]
      acc = mfma_bf16(fk1, pack8(s, 1), acc);
    };
    int kb = 0;
    for (; kb + 128 <= n; kb += 128) {
      tile(0); tile(256); tile(512); tile(768);
      aK0 += 4 * rstep; aK1 += 4 * rstep; aV0 += 4 * rstep; aKt += 4 * kstep;
    }
    for (; kb < n; kb += 32) {
      tile(0);
      aK0 += rstep; aK1 += rstep; aV0 += rstep; aKt += kstep;
    }
    if (ch == nch - 1 && hh == 0 && qok) {
      bf16x4 ov;
      ov[0] = (bf16)(acc[0] * scale); ov[1] = (bf16)(acc[1] * scale); ov[2] = (bf16)(acc[2] * scale); ov[3] = (bf16)(acc[3] * scale);
      *reinterpret_cast<bf16x4*>(dq + qrow + h * 4) = ov;
    }
    if (st + 1 < nst) {
      unsigned char* nxt = smem_attn + ((st + 1) & 1) * 2 * ATT_IMG;
      head_write(rk, nxt, nn, tid);
      head_write(rv, nxt + ATT_IMG, nn, tid);
      __syncthreads();
    }
  }
}

// dk, dv: wave = 32 keys, heads in turn; the head's queries and dO rows staged raw, plus a 16-byte row of shifts per query,
// [-lse hi, lo, 0, 0 | -delta hi, mid, lo, 0]: the A fragment of the half-wave hh == 1 in BOTH first products (the K operand holds ones in
// slots 8-9, the V operand in slots 12-14).
constexpr int ATT_KV_BUF = 2 * ATT_IMG + ATT_HS * 16;        // Q | dO | shifts
__global__ __launch_bounds__(512, 4) void attn_bwd_dkv_mfma_kernel(bf16* dk, bf16* dv, const bf16* dout, const bf16* q, const bf16* k,
                                                               const bf16* v, const float* lse, const float* delta, int Sq, int Skv,
                                                               int H, float scale, float c, int nkb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_attn[];
  unsigned char* cb = smem_attn + 2 * ATT_KV_BUF;
  float* red = reinterpret_cast<float*>(cb + ATT_CONST);      // [8 waves][32 keys][8] partial dK | dV of the query splits
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  // nkb (1, 2, 4 or 8) key blocks per workgroup; with fewer than 8 the waves of a key block split the queries 8 / nkb ways (the text
  // cross-attention has 77 keys: 3 blocks) and the partial sums meet in LDS at the end of a head
  const int kbw = wave & (nkb - 1), split = wave / nkb, nsplit = 8 / nkb;
  const int b = blockIdx.y, k0 = (blockIdx.x * nkb + kbw) * 32, E = H * 4;
  const bool kok = k0 + r < Skv;
  const long krow = ((long)b * Skv + k0 + (kok ? r : 0)) * E;
  const int nch = (Sq + ATT_HS - 1) / ATT_HS, nst = H * nch;
  const_block(cb, tid);
  HeadRegs rq, rd;
  float rl[ATT_HS / 512], rdl[ATT_HS / 512];
  auto stage_load = [&](int st) {
    const int h2 = st / nch, ch2 = st - h2 * nch, i0 = ch2 * ATT_HS, nv = min(ATT_HS, Sq - i0);
    head_load(rq, q, (long)b * Sq + i0, nv, E, h2, tid);
    head_load(rd, dout, (long)b * Sq + i0, nv, E, h2, tid);
#pragma unroll
    for (int i = 0; i < ATT_HS / 512; ++i) {
      const int rr = tid + i * 512;
      rl[i] = rr < nv ? lse[((long)b * H + h2) * Sq + i0 + rr] * 1.4426950408889634f : 1.0e4f;   // queries past Sq: p = exp2(-1e4) = 0
      rdl[i] = rr < nv ? delta[((long)b * H + h2) * Sq + i0 + rr] : 0.f;
    }
    return nv;
  };
  auto stage_write = [&](int st, int nv) {
    unsigned char* buf = smem_attn + (st & 1) * ATT_KV_BUF;
    head_write(rq, buf, nv, tid);
    head_write(rd, buf + ATT_IMG, nv, tid);
#pragma unroll
    for (int i = 0; i < ATT_HS / 512; ++i) {
      const int rr = tid + i * 512;
      if (rr < ((nv + 31) & ~31)) {
        const bf16x8 a = shift_frag(-rl[i], 0);
        bf16x8 w = shift_frag(-rdl[i], 4);
        w[0] = a[0]; w[1] = a[1];
        *reinterpret_cast<bf16x8*>(buf + 2 * ATT_IMG + rr * 16) = w;
      }
    }
  };
  int nv0 = stage_load(0);
  stage_write(0, nv0);
  __syncthreads();
  const bf16x8 onesK = ones_frag(0, 2), onesV = ones_frag(4, 3);
  const int tstep = (lane & 3) == 0 ? 256 : 0, astep = hh == 0 ? 256 : 512;
  f32x16 ak, av;
  bf16x8 fkB, fvB;
  for (int st = 0; st < nst; ++st) {
    const int h = st / nch, ch = st - h * nch, n = min(ATT_HS, Sq - ch * ATT_HS);
    const unsigned char* imgQ = smem_attn + (st & 1) * ATT_KV_BUF;
    const unsigned char* imgD = imgQ + ATT_IMG;
    const unsigned char* imgS = imgD + ATT_IMG;
    int nn = 0;
    if (st + 1 < nst) nn = stage_load(st + 1);
    if (ch == 0) {
      fkB = hh == 0 ? scaled_frag(k + krow + h * 4, kok, c) : onesK;
      fvB = hh == 0 ? head_frag(v + krow + h * 4, kok) : onesV;
      ak = (f32x16)(0.f); av = (f32x16)(0.f);
    }
    // A fragments: lanes hh == 0 [q | q] and [dO | 0], lanes hh == 1 the query's shift row (both products)
    const unsigned char* aQ0 = hh == 0 ? imgQ + r * 8 : imgS + r * 16;
    const unsigned char* aQ1 = hh == 0 ? imgQ + r * 8 : imgS + r * 16 + 8;
    const unsigned char* aD0 = hh == 0 ? imgD + r * 8 : imgS + r * 16;
    const unsigned char* aD1 = hh == 0 ? cb + CB_ZERO : imgS + r * 16 + 8;
    const unsigned char* aQt = tr_ptr(imgQ, cb, lane, false);
    const unsigned char* aDt = tr_ptr(imgD, cb, lane, false);
    const int d1step = hh == 0 ? 0 : 512;
    auto tile = [&](const int off) {                          // (off counts 256-byte row tiles; the shift rows are twice as wide)
      const int offa = hh == 0 ? off : 2 * off;
      const bf16x8 fqr = frag2(aQ0 + offa, aQ1 + offa);
      const bf16x8 fdr = frag2(aD0 + offa, aD1 + offa);
      const bf16x8 fd0 = tr_frag(aDt + off), fd1 = tr_frag(aDt + off + 128);
      const bf16x8 fq0 = tr_frag(aQt + off), fq1 = tr_frag(aQt + off + 128);
      f32x16 s = mfma_bf16(fqr, fkB, (f32x16)(0.f));          // S'[q][k] = c q.k - lse[q]: rows = queries, columns = keys
      f32x16 dp = mfma_bf16(fdr, fvB, (f32x16)(0.f));         // dP'[q][k] = dO[q] . V[k] - delta[q]
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        s[reg] = __builtin_amdgcn_exp2f(s[reg]);
        dp[reg] *= s[reg];
      }
      av = mfma_bf16(fd0, pack8(s, 0), av);                   // dV^T[d][k] += dO^T[d][q] P[q][k]
      av = mfma_bf16(fd1, pack8(s, 1), av);
      ak = mfma_bf16(fq0, pack8(dp, 0), ak);                  // dK^T[d][k] += Q^T[d][q] dS[q][k]
      ak = mfma_bf16(fq1, pack8(dp, 1), ak);
    };
    const int ntq = (n + 31) >> 5, t0 = split * ntq / nsplit, t1 = (split + 1) * ntq / nsplit;
    aQ0 += t0 * astep; aQ1 += t0 * astep; aD0 += t0 * astep; aD1 += t0 * d1step; aQt += t0 * tstep; aDt += t0 * tstep;
    for (int t = t0; t < t1; ++t) {
      tile(0);
      aQ0 += astep; aQ1 += astep; aD0 += astep; aD1 += d1step; aQt += tstep; aDt += tstep;
    }
    if (nsplit > 1 && ch == nch - 1) {                        // (uniform over the workgroup)
      float* mine = red + ((long)wave * 32 + r) * 8;
      if (split > 0 && hh == 0) {
        *reinterpret_cast<f32x4*>(mine) = (f32x4){ak[0], ak[1], ak[2], ak[3]};
        *reinterpret_cast<f32x4*>(mine + 4) = (f32x4){av[0], av[1], av[2], av[3]};
      }
      __syncthreads();
      if (split == 0 && hh == 0) {
        for (int sp = 1; sp < nsplit; ++sp) {
          const float* o2 = red + ((long)(sp * nkb + kbw) * 32 + r) * 8;
          const f32x4 pk = *reinterpret_cast<const f32x4*>(o2), pv = *reinterpret_cast<const f32x4*>(o2 + 4);
          ak[0] += pk[0]; ak[1] += pk[1]; ak[2] += pk[2]; ak[3] += pk[3];
          av[0] += pv[0]; av[1] += pv[1]; av[2] += pv[2]; av[3] += pv[3];
        }
      }
    }
    if (ch == nch - 1 && hh == 0 && kok && split == 0) {
      bf16x4 ok4, ov4;
      ok4[0] = (bf16)(ak[0] * scale); ok4[1] = (bf16)(ak[1] * scale); ok4[2] = (bf16)(ak[2] * scale); ok4[3] = (bf16)(ak[3] * scale);
      ov4[0] = (bf16)av[0]; ov4[1] = (bf16)av[1]; ov4[2] = (bf16)av[2]; ov4[3] = (bf16)av[3];
      *reinterpret_cast<bf16x4*>(dk + krow + h * 4) = ok4;
      *reinterpret_cast<bf16x4*>(dv + krow + h * 4) = ov4;
    }
    if (st + 1 < nst) {
      stage_write(st + 1, nn);
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Merged backward: dq, dk, dv of a sample's heads from ONE evaluation of the probabilities.  The two kernels above each recompute
// p = exp2(S') (16 v_exp_f32 per tile and wave: the instruction that bounds them).  Here a tile (32 queries x 32 keys) is evaluated once,
// in the dk / dv kernel's orientation (rows = queries, columns = keys), whose accumulator layout puts the contraction index of the dQ
// product -- the keys -- across the LANES: the packed bf16 dS tile takes a round trip through a wave-private 2-KB LDS tile (written in
// the pack order the dK product already produced, XOR-swizzled 8-byte chunks, double-buffered so that the read-back of tile t - 1 runs
// beside the exps of tile t) and returns through the transposing read as the B operand of  dQ^T[d][q] += K^T[d][k] dS^T[k][q];  K^T of
// the pass's 32 keys is a constant fragment pair.
// Who owns what: the four waves of a workgroup walk the key blocks TOGETHER (one 32-key block per pass) and split the QUERIES -- wave w
// owns query tiles 8w .. 8w + 7 for every key block, so its dQ^T tiles accumulate in registers (8 x 4 VGPRs, the tile loop is unrolled)
// and are stored once per head; dK / dV of a pass are the sum of the four waves' partial sums, which meet in LDS (two barriers per pass).
// (A first version gave each wave its own key block and added the dQ^T tiles into an LDS accumulator: ds_add_f32 costs ~64 cycles per
// wave instruction, 4 per tile -- 1.66 ms instead of 0.68 without the adds.)  delta is computed while the head's dO / out rows are staged.
constexpr int MG_WAVES = 4, MG_TPW = ATT_HS / 32 / MG_WAVES;  // query tiles per wave
constexpr int MG_IMG = 2 * ATT_IMG + ATT_HS * 16;            // Q | dO | shifts of one head
constexpr int MG_RED = 2 * (MG_WAVES - 1) * 32 * 8 * 4;      // dK | dV partial sums of waves 1-3, two buffers
constexpr int MG_LDS = MG_IMG + MG_RED + MG_WAVES * 4096 + MG_WAVES * 256 + ATT_CONST;
DEVI void wave_lds_fence() {                                 // (same wave: the LDS keeps the order; this only stops the compiler)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
DEVI bf16x8 tr_pair(const unsigned char* p0, const unsigned char* p1) {
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p1));
  return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

__global__ __launch_bounds__(256, 2) void attn_bwd_merged_kernel(bf16* dq, bf16* dk, bf16* dv, const bf16* dout, const bf16* out,
                                                                const bf16* q, const bf16* k, const bf16* v, const float* lse, int Sq,
                                                                int Skv, int H, int hpw, float scale, float c) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_attn[];
  unsigned char* imgQ = smem_attn;
  unsigned char* imgD = imgQ + ATT_IMG;
  unsigned char* imgS = imgD + ATT_IMG;
  float* red = reinterpret_cast<float*>(smem_attn + MG_IMG);  // [2][3 waves][32 keys][8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  unsigned char* dst = smem_attn + MG_IMG + MG_RED + wave * 4096;          // this wave's two dS^T tiles [32 keys][32 queries] bf16
  unsigned char* kim = smem_attn + MG_IMG + MG_RED + MG_WAVES * 4096 + wave * 256;  // this wave's copy of the pass's K rows [32][8 B]
  unsigned char* cb = smem_attn + MG_IMG + MG_RED + MG_WAVES * 4096 + MG_WAVES * 256;
  const int b = blockIdx.y, E = H * 4;
  const_block(cb, tid);
  const int Sq32 = (Sq + 31) & ~31, ntq = Sq32 >> 5;
  const int t0w = wave * MG_TPW;                              // this wave's query tiles [t0w, t0w + MG_TPW) (those below ntq)
  // per-lane constants of the transposed path
  const int p4 = lane & 3, qp = (lane & 15) >> 2, g16 = lane >> 4;
  // write side: this lane = key column r, rows (queries) 8 i + 4 hh + 0..3 -> chunk 2 i + hh of row r, swizzled with (r & 7)
  int woff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) woff[i] = r * 64 + (((2 * i + hh) ^ (r & 7)) << 3);
  // read side (B operand of slice s): rows k = 16 s + 8 hh + qp (+4), query chunk 4 (g16 & 1) + p4
  int roff[2][2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = 16 * s2 + 8 * hh + qp + 4 * j, chunk = 4 * (g16 & 1) + p4;
      roff[s2][j] = row * 64 + ((chunk ^ (row & 7)) << 3);
    }
  // A-fragment pointers of this wave's FIRST tile (lanes hh == 0: the Q / dO row; lanes hh == 1: the query's shift row)
  const unsigned char* bQ0 = (hh == 0 ? imgQ + r * 8 : imgS + r * 16) + t0w * (hh == 0 ? 256 : 512);
  const unsigned char* bQ1 = (hh == 0 ? imgQ + r * 8 : imgS + r * 16 + 8) + t0w * (hh == 0 ? 256 : 512);
  const unsigned char* bD0 = (hh == 0 ? imgD + r * 8 : imgS + r * 16) + t0w * (hh == 0 ? 256 : 512);
  const unsigned char* bD1 = hh == 0 ? cb + CB_ZERO : imgS + r * 16 + 8 + t0w * 512;
  const unsigned char* bQt = tr_ptr(imgQ, cb, lane, false) + (p4 == 0 ? t0w * 256 : 0);
  const unsigned char* bDt = tr_ptr(imgD, cb, lane, false) + (p4 == 0 ? t0w * 256 : 0);
  const int astep = hh == 0 ? 256 : 512, d1step = hh == 0 ? 0 : 512, tstep = p4 == 0 ? 256 : 0;
  int parity = 0;
  for (int hi = 0; hi < hpw; ++hi) {
    const int h = blockIdx.x * hpw + hi;
    if (h >= H) break;                                        // (uniform)
    __syncthreads();                                          // the previous head's sweeps are done
    for (int rr = tid; rr < Sq32; rr += 256) {
      uint2 wq = make_uint2(0, 0), wd = make_uint2(0, 0);
      float ls = 1.0e4f, dl = 0.f;                            // queries past Sq: p = exp2(-1e4) = 0
      if (rr < Sq) {
        const long ro = ((long)b * Sq + rr) * E + h * 4;
        wq = *reinterpret_cast<const uint2*>(q + ro);
        wd = *reinterpret_cast<const uint2*>(dout + ro);
        const bf16x4 a = __builtin_bit_cast(bf16x4, wd), o4 = *reinterpret_cast<const bf16x4*>(out + ro);
        dl = (float)a[0] * (float)o4[0] + (float)a[1] * (float)o4[1] + (float)a[2] * (float)o4[2] + (float)a[3] * (float)o4[3];
        ls = lse[((long)b * H + h) * Sq + rr] * 1.4426950408889634f;
      }
      *reinterpret_cast<uint2*>(imgQ + rr * 8) = wq;
      *reinterpret_cast<uint2*>(imgD + rr * 8) = wd;
      const bf16x8 sa = shift_frag(-ls, 0);
      bf16x8 w = shift_frag(-dl, 4);
      w[0] = sa[0]; w[1] = sa[1];
      *reinterpret_cast<bf16x8*>(imgS + rr * 16) = w;
    }
    __syncthreads();
    float dqa[MG_TPW][4];
#pragma unroll
    for (int j = 0; j < MG_TPW; ++j) { dqa[j][0] = 0.f; dqa[j][1] = 0.f; dqa[j][2] = 0.f; dqa[j][3] = 0.f; }
    for (int k0 = 0; k0 < Skv; k0 += 32) {                    // passes: one 32-key block, all four waves
      const bool kok = k0 + r < Skv;
      const long krow = ((long)b * Skv + k0 + (kok ? r : 0)) * E + h * 4;
      const bf16x8 fkB = hh == 0 ? scaled_frag(k + krow, kok, c) : ones_frag(0, 2);
      const bf16x8 fvB = hh == 0 ? head_frag(v + krow, kok) : ones_frag(4, 3);
      // K^T of the pass's keys in the NATURAL slot order k = 16 s + 8 hh + j (the order the transposed dS^T comes back in)
      wave_lds_fence();
      if (hh == 0) *reinterpret_cast<uint2*>(kim + r * 8) = kok ? *reinterpret_cast<const uint2*>(k + krow) : make_uint2(0, 0);
      wave_lds_fence();
      bf16x8 kt[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const unsigned char* pa = p4 == 0 ? kim + (16 * s2 + 8 * hh + qp) * 8 : cb + CB_ZERO;
        const unsigned char* pb = p4 == 0 ? kim + (16 * s2 + 8 * hh + qp + 4) * 8 : cb + CB_ZERO;
        kt[s2] = tr_pair(pa, pb);
      }
      f32x16 ak = (f32x16)(0.f), av = (f32x16)(0.f);
      // the transposed half of tile j - 1 runs inside iteration j, beside that tile's exps (its LDS round trip is long over by then)
      auto dq_part = [&](const int jp) {
        const unsigned char* src = dst + (jp & 1) * 2048;
        const bf16x8 b0 = tr_pair(src + roff[0][0], src + roff[0][1]);
        const bf16x8 b1 = tr_pair(src + roff[1][0], src + roff[1][1]);
        f32x16 dqt = mfma_bf16(kt[0], b0, (f32x16)(0.f));     // dQ^T[d][q] of that tile and this key block
        dqt = mfma_bf16(kt[1], b1, dqt);
        dqa[jp][0] += dqt[0]; dqa[jp][1] += dqt[1]; dqa[jp][2] += dqt[2]; dqa[jp][3] += dqt[3];
      };
#pragma unroll
      for (int j = 0; j < MG_TPW; ++j) {
        if (t0w + j < ntq) {                                  // (uniform over the wave)
          wave_lds_fence();                                   // iteration j - 1 wrote buffer (j - 1) & 1 and read buffer j & 1
          const bf16x8 fqr = frag2(bQ0 + j * astep, bQ1 + j * astep);
          const bf16x8 fdr = frag2(bD0 + j * astep, bD1 + j * d1step);
          const bf16x8 fd0 = tr_frag(bDt + j * tstep), fd1 = tr_frag(bDt + j * tstep + 128);
          const bf16x8 fq0 = tr_frag(bQt + j * tstep), fq1 = tr_frag(bQt + j * tstep + 128);
          f32x16 s = mfma_bf16(fqr, fkB, (f32x16)(0.f));      // S'[q][k] = c q.k - lse[q]: rows = queries, columns = keys
          f32x16 dp = mfma_bf16(fdr, fvB, (f32x16)(0.f));     // dP'[q][k] = dO[q] . V[k] - delta[q]
          if (j > 0) dq_part(j - 1);
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            s[reg] = __builtin_amdgcn_exp2f(s[reg]);
            dp[reg] *= s[reg];
          }
          const bf16x8 ds0 = pack8(dp, 0), ds1 = pack8(dp, 1);
          av = mfma_bf16(fd0, pack8(s, 0), av);               // dV^T[d][k] += dO^T[d][q] P[q][k]
          av = mfma_bf16(fd1, pack8(s, 1), av);
          ak = mfma_bf16(fq0, ds0, ak);                       // dK^T[d][k] += Q^T[d][q] dS[q][k]
          ak = mfma_bf16(fq1, ds1, ak);
          const u32x4 w0 = __builtin_bit_cast(u32x4, ds0), w1 = __builtin_bit_cast(u32x4, ds1);
          unsigned char* dw = dst + (j & 1) * 2048;           // dS -> LDS (key-major), read back transposed in the next iteration
          *reinterpret_cast<u32x2*>(dw + woff[0]) = (u32x2){w0.x, w0.y};
          *reinterpret_cast<u32x2*>(dw + woff[1]) = (u32x2){w0.z, w0.w};
          *reinterpret_cast<u32x2*>(dw + woff[2]) = (u32x2){w1.x, w1.y};
          *reinterpret_cast<u32x2*>(dw + woff[3]) = (u32x2){w1.z, w1.w};
        }
      }
      wave_lds_fence();
      {
        const int nj = min(MG_TPW, ntq - t0w);                // tiles this wave had (<= 0: none)
#pragma unroll
        for (int j = 0; j < MG_TPW; ++j) if (j == nj - 1) dq_part(j);
      }
      // dK / dV of the pass: the four waves' partial sums meet in LDS (buffer `parity`: the previous pass's may still be read)
      float* mine = red + ((long)(parity * (MG_WAVES - 1) + wave - 1) * 32 + r) * 8;
      if (wave > 0 && hh == 0) {
        *reinterpret_cast<f32x4*>(mine) = (f32x4){ak[0], ak[1], ak[2], ak[3]};
        *reinterpret_cast<f32x4*>(mine + 4) = (f32x4){av[0], av[1], av[2], av[3]};
      }
      __syncthreads();
      if (wave == 0 && hh == 0 && kok) {
#pragma unroll
        for (int w2 = 0; w2 < MG_WAVES - 1; ++w2) {
          const float* o2 = red + ((long)(parity * (MG_WAVES - 1) + w2) * 32 + r) * 8;
          const f32x4 pk = *reinterpret_cast<const f32x4*>(o2), pv = *reinterpret_cast<const f32x4*>(o2 + 4);
          ak[0] += pk[0]; ak[1] += pk[1]; ak[2] += pk[2]; ak[3] += pk[3];
          av[0] += pv[0]; av[1] += pv[1]; av[2] += pv[2]; av[3] += pv[3];
        }
        bf16x4 ok4, ov4;
        ok4[0] = (bf16)(ak[0] * scale); ok4[1] = (bf16)(ak[1] * scale); ok4[2] = (bf16)(ak[2] * scale); ok4[3] = (bf16)(ak[3] * scale);
        ov4[0] = (bf16)av[0]; ov4[1] = (bf16)av[1]; ov4[2] = (bf16)av[2]; ov4[3] = (bf16)av[3];
        *reinterpret_cast<bf16x4*>(dk + krow) = ok4;
        *reinterpret_cast<bf16x4*>(dv + krow) = ov4;
      }
      parity ^= 1;                                            // (one barrier per pass: the other buffer is free again two passes later)
    }
    // dq of this wave's query tiles, straight from the registers
#pragma unroll
    for (int j = 0; j < MG_TPW; ++j) {
      const int qq = (t0w + j) * 32 + r;
      if (hh == 0 && qq < Sq) {
        bf16x4 o4;
        o4[0] = (bf16)(dqa[j][0] * scale); o4[1] = (bf16)(dqa[j][1] * scale); o4[2] = (bf16)(dqa[j][2] * scale); o4[3] = (bf16)(dqa[j][3] * scale);
        *reinterpret_cast<bf16x4*>(dq + ((long)b * Sq + qq) * E + h * 4) = o4;
      }
    }
  }
}

static inline bool attn_mfma_ok(int H, const void* a, const void* b2, const void* c2, const void* d2) {
  static const bool off = getenv("HDMOE_ATTN_VALU") != nullptr;
  return !off && H >= 1 && H <= 4096 && (((uintptr_t)a | (uintptr_t)b2 | (uintptr_t)c2 | (uintptr_t)d2) & 7) == 0;
}

static int attn_force_slow() {                                // (tests: the online-maximum sweep of the forward kernel for every block)
  const char* e = getenv("HDMOE_ATTN_SLOW");
  return e && atoi(e) != 0;
}

static void attn_mfma_attrs() {                              // (the dk / dv kernel's two buffers are 64 KB + the constant block)
  static unsigned long long done = 0;
  if (!hdmoe_first_on_device(done)) return;
  (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ATT_KV_BUF + ATT_CONST + 8 * 32 * 8 * 4);
}

template <typename T, int D>
int attn_fwd_launch(void* out, float* lse, const void* q, const void* k, const void* v, const float* bias, int B, int Sq, int Skv,
                    int H, int Sb, hipStream_t st) {
  if constexpr (sizeof(T) == 2 && D == 4) {
    if (!bias && attn_mfma_ok(H, out, q, k, v)) {
      const size_t lds = 4 * ATT_IMG + ATT_CONST;
      attn_mfma_attrs();
      hipLaunchKernelGGL(attn_fwd_mfma_kernel, dim3(cdiv(Sq, 256), B), dim3(512), lds, st, (bf16*)out, lse, (const bf16*)q, (const bf16*)k,
                         (const bf16*)v, Sq, Skv, H, 1.4426950408889634f / sqrtf((float)D), attn_force_slow());
      return hdmoe_launch_status();
    }
  }
  dim3 grid(cdiv(Sq, TQ), H, B);
  if (bias) hipLaunchKernelGGL((attn_fwd_kernel<T, D, true>), grid, dim3(TQ), 0, st, (T*)out, lse, (const T*)q, (const T*)k, (const T*)v, bias, Sq,
                     Skv, H, Sb, 1.f / sqrtf((float)D));
  else hipLaunchKernelGGL((attn_fwd_kernel<T, D, false>), grid, dim3(TQ), 0, st, (T*)out, lse, (const T*)q, (const T*)k, (const T*)v, bias, Sq,
                     Skv, H, Sb, 1.f / sqrtf((float)D));
  return hdmoe_launch_status();
}
template <typename T, int D>
int attn_bwd_launch(void* dq, void* dk, void* dv, float* dbias, float* delta, const void* dout, const void* out, const void* q,
                    const void* k, const void* v, const float* lse, const float* bias, int B, int Sq, int Skv, int H, int Sb,
                    hipStream_t st) {
  const float scale = 1.f / sqrtf((float)D);
  if constexpr (sizeof(T) == 2 && D == 4) {
    if (!bias && attn_mfma_ok(H, dq, dk, dv, dout) && attn_mfma_ok(H, out, q, k, v)) {
      const float c = scale * 1.4426950408889634f;
      static const bool merged = !(getenv("HDMOE_ATTN_BWD_MERGED") && atoi(getenv("HDMOE_ATTN_BWD_MERGED")) == 0);
      if (merged && Sq <= ATT_HS) {                           // one evaluation of the probabilities for dq, dk and dv
        static unsigned long long attr = 0;
        if (hdmoe_first_on_device(attr)) { (void)hipFuncSetAttribute((const void*)attn_bwd_merged_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MG_LDS); }
        int hg = (int)cdiv(512, B); if (hg > H) hg = H; if (hg < 1) hg = 1;
        const int hpw = (int)cdiv(H, hg);
        hipLaunchKernelGGL(attn_bwd_merged_kernel, dim3(cdiv(H, hpw), B), dim3(64 * MG_WAVES), MG_LDS, st, (bf16*)dq, (bf16*)dk, (bf16*)dv,
                           (const bf16*)dout, (const bf16*)out, (const bf16*)q, (const bf16*)k, (const bf16*)v, lse, Sq, Skv, H, hpw, scale, c);
        return hdmoe_launch_status();
      }
      const size_t lds_q = 4 * ATT_IMG + ATT_CONST, lds_kv = 2 * ATT_KV_BUF + ATT_CONST + 8 * 32 * 8 * 4;
      const int nb32 = (int)cdiv(Skv, 32), nkb = nb32 >= 5 ? 8 : (nb32 >= 3 ? 4 : nb32);
      attn_mfma_attrs();
      hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel, dim3(cdiv(Sq, 256), B), dim3(512), lds_q, st, (bf16*)dq, delta, (const bf16*)dout,
                         (const bf16*)out, (const bf16*)q, (const bf16*)k, (const bf16*)v, lse, Sq, Skv, H, scale, c);
      hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel, dim3(cdiv(Skv, 32 * nkb), B), dim3(512), lds_kv, st, (bf16*)dk, (bf16*)dv, (const bf16*)dout,
                         (const bf16*)q, (const bf16*)k, (const bf16*)v, lse, delta, Sq, Skv, H, scale, c, nkb);
      return hdmoe_launch_status();
    }
  }
#define ATTN_BWD(BB)                                                                                                                  \
  hipLaunchKernelGGL((attn_bwd_dq_kernel<T, D, BB>), dim3(cdiv(Sq, TQ), H, B), dim3(TQ), 0, st, (T*)dq, delta, dbias, (const T*)dout,  \
                     (const T*)out, (const T*)q, (const T*)k, (const T*)v, lse, bias, Sq, Skv, H, Sb, scale);                          \
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, D, BB>), dim3(cdiv(Skv, TQ), H, B), dim3(TQ), 0, st, (T*)dk, (T*)dv, (const T*)dout,      \
                     (const T*)q, (const T*)k, (const T*)v, lse, delta, bias, Sq, Skv, H, Sb, scale);
  if (bias) { ATTN_BWD(true) } else { ATTN_BWD(false) }
  if (bias && dbias) {
    const long nthreads = (long)H * Sq * Skv;
    hipLaunchKernelGGL((attn_dbias_kernel<T, D>), dim3(cdiv(nthreads, 256)), dim3(256), 0, st, dbias, (const T*)dout, (const T*)q, (const T*)k,
                       (const T*)v, lse, delta, bias, B, Sq, Skv, H, Sb, scale);
  }
  return hdmoe_launch_status();
}

// ------------------------------------------------------------------ rel_pos_bias resize (S > S0)
// Reference model_internals.py:388-397: F.interpolate(bias[None], size=(S,S), mode='bicubic', align_corners=False).
// Cubic convolution, A = -0.75, source coordinate (dst + 0.5) * S0/S - 0.5 (not clamped), taps clamped to the table.
struct CubicTaps { int i[4]; float c[4]; };
DEVI CubicTaps cubic_taps(int dst, int S0, float scale) {
  const float A = -0.75f;
  const float real = scale * (dst + 0.5f) - 0.5f;
  const float fl = floorf(real);
  const float t = real - fl;
  const int i0 = (int)fl;
  CubicTaps r;
  const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
  r.c[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
  r.c[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
  r.c[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
  r.c[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
#pragma unroll
  for (int k = 0; k < 4; ++k) r.i[k] = min(max(i0 - 1 + k, 0), S0 - 1);
  return r;
}

template <bool BWD>
__global__ __launch_bounds__(256) void bicubic_kernel(float* dst, const float* src, int H, int S0, int S) {
  // forward: dst = out [H][S][S], src = table [H][S0][S0];  backward: dst = dtable (pre-zeroed, atomics), src = dout
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)H * S * S) return;
  const int x = (int)(e % S), y = (int)((e / S) % S), h = (int)(e / ((long)S * S));
  const float scale = (float)S0 / (float)S;
  const CubicTaps ty = cubic_taps(y, S0, scale), tx = cubic_taps(x, S0, scale);
  if (!BWD) {
    const float* t = src + (long)h * S0 * S0;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      float row = 0.f;
#pragma unroll
      for (int b = 0; b < 4; ++b) row += tx.c[b] * t[(long)ty.i[a] * S0 + tx.i[b]];
      acc += ty.c[a] * row;
    }
    dst[e] = acc;
  } else {
    float* t = dst + (long)h * S0 * S0;
    const float g = src[e];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) atomicAdd(&t[(long)ty.i[a] * S0 + tx.i[b]], g * ty.c[a] * tx.c[b]);
  }
}

// =====================================================================================================================
// Ragged self-attention of the ViT expert bank (see ragged.hip): rows [seg[g], seg[g+1]) of the padded [R][Sp][E] tensors belong to
// expert g, hold len[g] real tokens and use that expert's rel_pos_bias table.  One 64-lane workgroup per (row, head): the head's
// K / V (and in the backward Q / dO) rows sit in LDS as fp32, thread = query row (forward, dq) or key row (dk, dv).  Padding
// tokens get zero outputs and zero gradients.
struct RagA {
  const int* seg; int ng;
  int len[HDMOE_MAX_GROUPS], sb[HDMOE_MAX_GROUPS];
  const float* bias[HDMOE_MAX_GROUPS];
  float* dbias[HDMOE_MAX_GROUPS];
  long off[HDMOE_MAX_GROUPS + 1];                           // dbias walker: first thread of each expert
};
DEVI int raga_group(const RagA& r, int row) {
  int g = -1;
  for (int i = 0; i < r.ng; ++i)
    if (row >= r.seg[i] && row < r.seg[i + 1]) g = i;
  return g;
}

template <typename T, int D>
__global__ __launch_bounds__(64) void attn_rag_fwd_kernel(T* out, float* lse, const T* q, const T* k, const T* v, RagA rg, int Sp, int H,
                                                         float scale) {
  extern __shared__ float sm_rag[];
  const int r = blockIdx.x, h = blockIdx.y, E = H * D;
  const int g = raga_group(rg, r);
  const int len = g >= 0 ? rg.len[g] : 0;
  float* sk = sm_rag;
  float* sv = sm_rag + len * D;
  for (int e = threadIdx.x; e < len * D; e += 64) {
    const long src = ((long)r * Sp + e / D) * E + h * D + e % D;
    sk[e] = to_f(k[src]); sv[e] = to_f(v[src]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Sp; i += 64) {
    const long row = ((long)r * Sp + i) * E + h * D;
    if (i >= len) {
#pragma unroll
      for (int d = 0; d < D; ++d) out[row + d] = from_f<T>(0.f);
      lse[((long)r * H + h) * Sp + i] = 0.f;
      continue;
    }
    float qv[D], o[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { qv[d] = to_f(q[row + d]) * scale; o[d] = 0.f; }
    const float* brow = rg.bias[g] + ((long)h * rg.sb[g] + i) * rg.sb[g];
    float m = -INFINITY, l = 0.f;
    for (int j = 0; j < len; ++j) {
      float a = brow[j];
#pragma unroll
      for (int d = 0; d < D; ++d) a += qv[d] * sk[j * D + d];
      const float mn = fmaxf(m, a);
      const float corr = __expf(m - mn), p = __expf(a - mn);
      l = l * corr + p;
#pragma unroll
      for (int d = 0; d < D; ++d) o[d] = o[d] * corr + p * sv[j * D + d];
      m = mn;
    }
    const float il = 1.f / l;
#pragma unroll
    for (int d = 0; d < D; ++d) out[row + d] = from_f<T>(o[d] * il);
    lse[((long)r * H + h) * Sp + i] = m + __logf(l);
  }
}

// dq, dk, dv (and delta for the dbias walker) of one (row, head)
template <typename T, int D>
__global__ __launch_bounds__(64) void attn_rag_bwd_kernel(T* dq, T* dk, T* dv, float* delta, const T* dout, const T* out, const T* q,
                                                         const T* k, const T* v, const float* lse, RagA rg, int Sp, int H, float scale) {
  extern __shared__ float sm_rag[];
  const int r = blockIdx.x, h = blockIdx.y, E = H * D;
  const int g = raga_group(rg, r);
  const int len = g >= 0 ? rg.len[g] : 0;
  float* sq = sm_rag;                                        // q * scale
  float* sk = sq + len * D;
  float* sv = sk + len * D;
  float* sdo = sv + len * D;
  float* sl = sdo + len * D;                                 // lse
  float* sd = sl + len;                                      // delta
  for (int e = threadIdx.x; e < len * D; e += 64) {
    const long src = ((long)r * Sp + e / D) * E + h * D + e % D;
    sq[e] = to_f(q[src]) * scale; sk[e] = to_f(k[src]); sv[e] = to_f(v[src]); sdo[e] = to_f(dout[src]);
  }
  for (int i = threadIdx.x; i < len; i += 64) {
    const long row = ((long)r * Sp + i) * E + h * D;
    float dl = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) dl += to_f(dout[row + d]) * to_f(out[row + d]);
    sd[i] = dl; sl[i] = lse[((long)r * H + h) * Sp + i];
    delta[((long)r * H + h) * Sp + i] = dl;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Sp; i += 64) {               // thread = query row: dq
    const long row = ((long)r * Sp + i) * E + h * D;
    float acc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = 0.f;
    if (i < len) {
      const float* brow = rg.bias[g] + ((long)h * rg.sb[g] + i) * rg.sb[g];
      for (int j = 0; j < len; ++j) {
        float s = brow[j], dp = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) { s += sq[i * D + d] * sk[j * D + d]; dp += sdo[i * D + d] * sv[j * D + d]; }
        const float ds = __expf(s - sl[i]) * (dp - sd[i]);
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] += ds * sk[j * D + d];
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) dq[row + d] = from_f<T>(acc[d] * scale);
  }
  for (int j = threadIdx.x; j < Sp; j += 64) {               // thread = key row: dk, dv
    const long row = ((long)r * Sp + j) * E + h * D;
    float ak[D], av[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { ak[d] = 0.f; av[d] = 0.f; }
    if (j < len) {
      const float* bcol = rg.bias[g] + (long)h * rg.sb[g] * rg.sb[g] + j;
      for (int i = 0; i < len; ++i) {
        float s = bcol[(long)i * rg.sb[g]], dp = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) { s += sq[i * D + d] * sk[j * D + d]; dp += sdo[i * D + d] * sv[j * D + d]; }
        const float p = __expf(s - sl[i]);
        const float ds = p * (dp - sd[i]);
#pragma unroll
        for (int d = 0; d < D; ++d) { av[d] += p * sdo[i * D + d]; ak[d] += ds * sq[i * D + d]; }
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) { dk[row + d] = from_f<T>(ak[d]); dv[row + d] = from_f<T>(av[d]); }
  }
}

// d(rel_pos_bias_g)[h][i][j] += sum over the rows of expert g of ds: one thread per (g, h, i, j) walks that expert's rows
template <typename T, int D>
__global__ void attn_rag_dbias_kernel(const T* dout, const T* q, const T* k, const T* v, const float* lse, const float* delta, RagA rg,
                                      int Sp, int H, float scale) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rg.off[rg.ng]) return;
  int g = 0;
  while (g + 1 < rg.ng && idx >= rg.off[g + 1]) ++g;
  const long e = idx - rg.off[g];
  const int len = rg.len[g], Sb = rg.sb[g], E = H * D;
  const int j = (int)(e % len), i = (int)((e / len) % len), h = (int)(e / ((long)len * len));
  if (!rg.dbias[g]) return;
  const float bv = rg.bias[g][((long)h * Sb + i) * Sb + j];
  float acc = 0.f;
  for (int r = rg.seg[g] + (int)blockIdx.y; r < rg.seg[g + 1]; r += (int)gridDim.y) {     // gridDim.y row slices per element (atomic merge)
    const T* qp = q + ((long)r * Sp + i) * E + h * D;
    const T* dop = dout + ((long)r * Sp + i) * E + h * D;
    const T* kp = k + ((long)r * Sp + j) * E + h * D;
    const T* vp = v + ((long)r * Sp + j) * E + h * D;
    float s = 0.f, dp = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) { s += to_f(qp[d]) * to_f(kp[d]); dp += to_f(dop[d]) * to_f(vp[d]); }
    const long li = ((long)r * H + h) * Sp + i;
    acc += __expf(s * scale + bv - lse[li]) * (dp - delta[li]);
  }
  atomicAdd(&rg.dbias[g][((long)h * Sb + i) * Sb + j], acc);
}

static inline bool mk_raga(RagA& rg, const int* seg, const int* lens, const int* sb, const float* const* bias, float* const* dbias,
                           int ng, int Sp, int H) {
  if (!seg || !lens || !sb || !bias || ng < 1 || ng > HDMOE_MAX_GROUPS) return false;
  rg.seg = seg; rg.ng = ng; rg.off[0] = 0;
  for (int g = 0; g < HDMOE_MAX_GROUPS; ++g) {
    const bool in = g < ng;
    rg.len[g] = in ? lens[g] : 0; rg.sb[g] = in ? sb[g] : 0;
    rg.bias[g] = in ? bias[g] : nullptr; rg.dbias[g] = (in && dbias) ? dbias[g] : nullptr;
    if (in && (lens[g] < 1 || lens[g] > Sp || sb[g] < lens[g] || !bias[g])) return false;
    rg.off[g + 1] = rg.off[g] + (in ? (long)H * lens[g] * lens[g] : 0);
  }
  return true;
}
template <typename T, int D>
int attn_rag_fwd_launch(void* out, float* lse, const void* q, const void* k, const void* v, const RagA& rg, int R, int Sp, int H,
                        hipStream_t st) {
  int ml = 0;
  for (int g = 0; g < rg.ng; ++g) ml = max(ml, rg.len[g]);
  const size_t lds = (size_t)2 * ml * D * sizeof(float);
  if (lds > 60 * 1024) return HDMOE_EINVAL;
  hipLaunchKernelGGL((attn_rag_fwd_kernel<T, D>), dim3(R, H), dim3(64), lds, st, (T*)out, lse, (const T*)q, (const T*)k, (const T*)v, rg, Sp, H,
                     1.f / sqrtf((float)D));
  return hdmoe_launch_status();
}
template <typename T, int D>
int attn_rag_bwd_launch(void* dq, void* dk, void* dv, float* delta, const void* dout, const void* out, const void* q, const void* k,
                        const void* v, const float* lse, const RagA& rg, bool want_dbias, int R, int Sp, int H, hipStream_t st) {
  int ml = 0;
  for (int g = 0; g < rg.ng; ++g) ml = max(ml, rg.len[g]);
  const size_t lds = (size_t)(4 * ml * D + 2 * ml) * sizeof(float);
  if (lds > 60 * 1024) return HDMOE_EINVAL;
  const float scale = 1.f / sqrtf((float)D);
  hipLaunchKernelGGL((attn_rag_bwd_kernel<T, D>), dim3(R, H), dim3(64), lds, st, (T*)dq, (T*)dk, (T*)dv, delta, (const T*)dout, (const T*)out,
                     (const T*)q, (const T*)k, (const T*)v, lse, rg, Sp, H, scale);
  if (want_dbias)
    hipLaunchKernelGGL((attn_rag_dbias_kernel<T, D>), dim3(cdiv(rg.off[rg.ng], 256), 8), dim3(256), 0, st, (const T*)dout, (const T*)q, (const T*)k,
                       (const T*)v, lse, delta, rg, Sp, H, scale);
  return hdmoe_launch_status();
}

}  // namespace

#define D_SWITCH(D, CALL)                       \
  switch (D) {                                  \
    case 1: { constexpr int DD = 1; CALL; }     \
    case 2: { constexpr int DD = 2; CALL; }     \
    case 4: { constexpr int DD = 4; CALL; }     \
    case 8: { constexpr int DD = 8; CALL; }     \
    case 16: { constexpr int DD = 16; CALL; }   \
    case 32: { constexpr int DD = 32; CALL; }   \
    default: return HDMOE_EINVAL;               \
  }

extern "C" {

int hdmoe_attn_fwd(void* out, float* lse, const void* q, const void* k, const void* v, const float* bias, int B, int Sq, int Skv,
                   int H, int D, int Sb, int dtype, hipStream_t stream) {
  if (B < 1 || B > 65535 || H < 1 || H > 65535 || Sq < 1 || Skv < 1 || (bias && (Sb < Sq || Sb < Skv))) return HDMOE_EINVAL;
  if (dtype == HDMOE_F32) { D_SWITCH(D, return (attn_fwd_launch<float, DD>(out, lse, q, k, v, bias, B, Sq, Skv, H, Sb, stream))) }
  if (dtype == HDMOE_BF16) { D_SWITCH(D, return (attn_fwd_launch<bf16, DD>(out, lse, q, k, v, bias, B, Sq, Skv, H, Sb, stream))) }
  return HDMOE_EDTYPE;
}

// delta: [B][H][Sq] fp32 scratch; dbias: [H][Sb][Sb] fp32 accumulator (caller zeroes) or null
int hdmoe_attn_bwd(void* dq, void* dk, void* dv, float* dbias, float* delta, const void* dout, const void* out, const void* q,
                   const void* k, const void* v, const float* lse, const float* bias, int B, int Sq, int Skv, int H, int D, int Sb,
                   int dtype, hipStream_t stream) {
  if (B < 1 || B > 65535 || H < 1 || H > 65535 || Sq < 1 || Skv < 1 || (bias && (Sb < Sq || Sb < Skv)) || (dbias && !bias)) return HDMOE_EINVAL;
  if (dtype == HDMOE_F32) { D_SWITCH(D, return (attn_bwd_launch<float, DD>(dq, dk, dv, dbias, delta, dout, out, q, k, v, lse, bias, B, Sq, Skv, H, Sb, stream))) }
  if (dtype == HDMOE_BF16) { D_SWITCH(D, return (attn_bwd_launch<bf16, DD>(dq, dk, dv, dbias, delta, dout, out, q, k, v, lse, bias, B, Sq, Skv, H, Sb, stream))) }
  return HDMOE_EDTYPE;
}

/* Ragged self-attention of the ViT expert bank: q/k/v/out [R][Sp][H*D]; rows [seg[g], seg[g+1]) belong to expert g with lens[g]
 * real tokens and the table bias[g] [H][sb[g]][sb[g]]; lse / delta [R][H][Sp].  dbias[g] (or null) accumulate (+=). */
int hdmoe_attn_rag_fwd(void* out, float* lse, const void* q, const void* k, const void* v, const float* const* bias, const int* seg,
                       const int* lens, const int* sb, int ngroups, int R, int Sp, int H, int D, int dtype, hipStream_t stream) {
  RagA rg;
  if (R < 1 || H < 1 || H > 65535 || !mk_raga(rg, seg, lens, sb, bias, nullptr, ngroups, Sp, H)) return HDMOE_EINVAL;
  if (dtype == HDMOE_F32) { D_SWITCH(D, return (attn_rag_fwd_launch<float, DD>(out, lse, q, k, v, rg, R, Sp, H, stream))) }
  if (dtype == HDMOE_BF16) { D_SWITCH(D, return (attn_rag_fwd_launch<bf16, DD>(out, lse, q, k, v, rg, R, Sp, H, stream))) }
  return HDMOE_EDTYPE;
}
int hdmoe_attn_rag_bwd(void* dq, void* dk, void* dv, float* const* dbias, float* delta, const void* dout, const void* out,
                       const void* q, const void* k, const void* v, const float* lse, const float* const* bias, const int* seg,
                       const int* lens, const int* sb, int ngroups, int R, int Sp, int H, int D, int dtype, hipStream_t stream) {
  RagA rg;
  if (R < 1 || H < 1 || H > 65535 || !mk_raga(rg, seg, lens, sb, bias, dbias, ngroups, Sp, H)) return HDMOE_EINVAL;
  const bool wd = dbias != nullptr;
  if (dtype == HDMOE_F32) { D_SWITCH(D, return (attn_rag_bwd_launch<float, DD>(dq, dk, dv, delta, dout, out, q, k, v, lse, rg, wd, R, Sp, H, stream))) }
  if (dtype == HDMOE_BF16) { D_SWITCH(D, return (attn_rag_bwd_launch<bf16, DD>(dq, dk, dv, delta, dout, out, q, k, v, lse, rg, wd, R, Sp, H, stream))) }
  return HDMOE_EDTYPE;
}

// out [H][S][S] <- table [H][S0][S0] (fwd);  dtable [H][S0][S0] (caller zeroes) += resize^T(dout [H][S][S]) (bwd)
int hdmoe_bicubic_fwd(float* out, const float* table, int H, int S0, int S, hipStream_t stream) {
  if (!out || !table || H < 1 || S0 < 1 || S < 1) return HDMOE_EINVAL;
  hipLaunchKernelGGL(bicubic_kernel<false>, dim3(cdiv((long)H * S * S, 256)), dim3(256), 0, stream, out, table, H, S0, S);
  return hdmoe_launch_status();
}
int hdmoe_bicubic_bwd(float* dtable, const float* dout, int H, int S0, int S, hipStream_t stream) {
  if (!dtable || !dout || H < 1 || S0 < 1 || S < 1) return HDMOE_EINVAL;
  hipLaunchKernelGGL(bicubic_kernel<true>, dim3(cdiv((long)H * S * S, 256)), dim3(256), 0, stream, dtable, dout, H, S0, S);
  return hdmoe_launch_status();
}

}  // extern "C"
