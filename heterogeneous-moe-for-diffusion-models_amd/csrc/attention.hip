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
// three contractions of a 32 x 32 tile run on the matrix pipe with the head dim zero-padded 4 -> 16 (75 % of each MFMA is
// padding, still ~8x cheaper than the VALU form) and only the softmax arithmetic stays on the vector pipe.
//   S^T[k][q]  = K[k][:] . Q[q][:]            A = K rows (LDS),        B = Q (registers, fixed per wave)
//   O^T[d][q] += V^T[d][k] * P^T[k][q]        A = V^T (LDS, permuted), B = P straight from the S^T accumulator registers
// An accumulator lane (c, hh) holds column q = c and rows k = (reg&3) + 8(reg>>2) + 4hh, i.e. for the 16-deep slice s of the
// second product its registers 8s..8s+7 ARE a B-operand fragment if K-slot (hh, j) is defined as k = 16s + (j&3) + 8(j>>2) + 4hh;
// the A operand (V^T, or K^T / Q^T / dO^T in the backward kernels) is staged in LDS in exactly that slot order (kslot()), so
// no cross-lane movement is needed.  Row 4 of the padded V^T tile is all ones: O^T row 4 accumulates the softmax denominator
// on the matrix pipe for free.  One workgroup = one 32-row block x all heads (wave w = head w), so K/V rows are loaded whole.
#ifndef HDMOE_ATTN_MK
#define HDMOE_ATTN_MK 64
#endif
constexpr int MK = HDMOE_ATTN_MK;                            // rows staged per barrier (a multiple of 64)
constexpr int MKR = MK / 64;                                 // 8-byte pieces of a staged tensor per thread
DEVI f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
DEVI bf16x8 pack8(const f32x16& v, int s) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16)v[8 * s + j];
  return o;
}
// fragment [x0 x1 x2 x3 0 0 0 0] on lanes hh == 0, zeros on hh == 1 (head dim 4 padded to the 16-deep K of the MFMA)
DEVI bf16x8 head_frag(const bf16* row, bool valid) {
  bf16x8 f = (bf16x8)(0);
  if (valid) {
    const bf16x4 t = *reinterpret_cast<const bf16x4*>(row);
    f[0] = t[0]; f[1] = t[1]; f[2] = t[2]; f[3] = t[3];
  }
  return f;
}
// LDS image of a staged [rows][E] bf16 tensor: per head MK + 1 rows of 16 bytes, [x0 x1 x2 x3 e4 0 0 0] (e4 = 1 for V in the
// forward kernel, else 0); row MK is all zeros.  Consecutive threads take consecutive heads of one row: whole global rows,
// and the 16-byte LDS writes of the 8 heads land 2064 B apart = on disjoint bank quads (conflict-free).
struct StageRegs { uint2 v[MKR]; };
DEVI void stage_load(StageRegs& rg, const bf16* src, long row0, int nrows_valid, int H, int tid, int nthr) {
#pragma unroll
  for (int i = 0; i < MKR; ++i) {
    const int idx = tid + i * nthr;
    const int rr = idx / H, hd = idx - rr * H;
    uint2 t = make_uint2(0, 0);
    if (rr < nrows_valid) t = *reinterpret_cast<const uint2*>(src + (row0 + rr) * (long)(H * 4) + hd * 4);
    rg.v[i] = t;
  }
}
DEVI void stage_write(const StageRegs& rg, bf16* rowL, unsigned e4, int H, int tid, int nthr) {
#pragma unroll
  for (int i = 0; i < MKR; ++i) {
    const int idx = tid + i * nthr;
    const int rr = idx / H, hd = idx - rr * H;
    *reinterpret_cast<uint4*>(rowL + ((long)hd * (MK + 1) + rr) * 8) = make_uint4(rg.v[i].x, rg.v[i].y, e4, 0);
  }
}
// A-operand fragment of X^T (rows = head-dim index, 8 K-slots of slice s) straight from the row image X[k][8] with the
// hardware transposing read (cdna_hip_programming.md T10): within a 16-lane group lane 4q+p addresses row q, columns 4p..4p+3
// and lane i receives column i of the four rows.  Lane (r, hh) so gets X[16s + j' + 8*(second read) + 4hh][r] -- the K-slot
// order of the accumulator registers.  Columns 8..15 of a "row" are the next row's bytes: they only feed MFMA output rows
// 8..31, which nobody reads.  The per-lane address is tr_base(); +64 elements = 8 rows for the second read.
DEVI int tr_base(int lane) { return ((((lane & 15) >> 2) + 4 * (lane >> 5)) * 8) + 4 * (lane & 3); }
DEVI bf16x8 tr_frag(const bf16* p) {
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p + 64));
  return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}
DEVI void zero_row(bf16* rowL, int H, int tid, int nthr) {
  for (int e = tid; e < H * 8; e += nthr) rowL[((long)(e >> 3) * (MK + 1) + MK) * 8 + (e & 7)] = (bf16)0.f;
}

__global__ __launch_bounds__(512) void attn_fwd_mfma_kernel(bf16* out, float* lse, const bf16* q, const bf16* k, const bf16* v, int Sq,
                                                           int Skv, int H, float c) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_attn[];
  bf16* rowK = reinterpret_cast<bf16*>(smem_attn);           // [H][MK+1][8]
  bf16* rowV = rowK + (long)H * (MK + 1) * 8;                 // [H][MK+1][8], element 4 of every row = 1
  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int b = blockIdx.y, q0 = blockIdx.x * 32, E = H * 4, nthr = H * 64;
  zero_row(rowK, H, tid, nthr);
  zero_row(rowV, H, tid, nthr);
  const bool qok = q0 + r < Sq;
  const bf16x8 fq = head_frag(q + ((long)b * Sq + q0 + (qok ? r : 0)) * E + h * 4, hh == 0 && qok);
  const bf16* aK = rowK + ((long)h * (MK + 1) + (hh == 0 ? r : MK)) * 8;
  const int aKstep = hh == 0 ? 32 * 8 : 0;
  const bf16* aV = rowV + (long)h * (MK + 1) * 8 + tr_base(lane);
  f32x16 o = (f32x16)(0.f);
  float m = -INFINITY;
  StageRegs rk, rv;
  stage_load(rk, k, (long)b * Skv, min(MK, Skv), H, tid, nthr);
  stage_load(rv, v, (long)b * Skv, min(MK, Skv), H, tid, nthr);
  for (int j0 = 0; j0 < Skv; j0 += MK) {
    __syncthreads();                                         // readers of the previous stage are done
    stage_write(rk, rowK, 0u, H, tid, nthr);
    stage_write(rv, rowV, 0x3F80u, H, tid, nthr);            // bf16 1.0 in column 4: O^T row 4 = sum_k P = softmax denominator
    __syncthreads();
    if (j0 + MK < Skv) {
      stage_load(rk, k, (long)b * Skv + j0 + MK, min(MK, Skv - j0 - MK), H, tid, nthr);
      stage_load(rv, v, (long)b * Skv + j0 + MK, min(MK, Skv - j0 - MK), H, tid, nthr);
    }
    const int nt = min(MK / 32, (Skv - j0 + 31) >> 5);
    for (int t = 0; t < nt; ++t) {
      const bf16x8 fk = *reinterpret_cast<const bf16x8*>(aK + t * aKstep);
      const bf16x8 fv0 = tr_frag(aV + t * 256);
      const bf16x8 fv1 = tr_frag(aV + t * 256 + 128);
      f32x16 s = mfma_bf16(fk, fq, (f32x16)(0.f));           // rows = keys, columns = queries
      const int kb = j0 + t * 32;
      if (kb + 32 > Skv) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          if (kb + acc_row(reg, lane) >= Skv) s[reg] = -INFINITY;
      }
      float tm = s[0];
#pragma unroll
      for (int reg = 1; reg < 16; ++reg) tm = fmaxf(tm, s[reg]);
      tm = fmaxf(tm, __shfl_xor(tm, 32, 64));                // the other half-wave holds the other 16 keys of these queries
      const float mn = fmaxf(m, tm);
      const float alpha = __builtin_amdgcn_exp2f((m - mn) * c);
      m = mn;
      const float nm = -mn * c;
      o[0] *= alpha; o[1] *= alpha; o[2] *= alpha; o[3] *= alpha;     // rows 0-3 (hh = 0) and the denominator row 4 (hh = 1)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) s[reg] = __builtin_amdgcn_exp2f(fmaf(s[reg], c, nm));
      o = mfma_bf16(fv0, pack8(s, 0), o);
      o = mfma_bf16(fv1, pack8(s, 1), o);
    }
  }
  const float l = __shfl_xor(o[0], 32, 64);                  // row 4 of O^T (lane + 32, register 0) = sum of the probabilities
  if (hh == 0 && qok) {
    const float il = 1.f / l;
    bf16x4 ov;
    ov[0] = (bf16)(o[0] * il); ov[1] = (bf16)(o[1] * il); ov[2] = (bf16)(o[2] * il); ov[3] = (bf16)(o[3] * il);
    *reinterpret_cast<bf16x4*>(out + ((long)b * Sq + q0 + r) * E + h * 4) = ov;
    lse[((long)b * H + h) * Sq + q0 + r] = (m * c + __builtin_amdgcn_logf(l)) * 0.6931471805599453f;
  }
}

// dq (+ delta): wave = (head, 32 queries); keys stream through LDS.
__global__ __launch_bounds__(512) void attn_bwd_dq_mfma_kernel(bf16* dq, float* delta, const bf16* dout, const bf16* out, const bf16* q,
                                                              const bf16* k, const bf16* v, const float* lse, int Sq, int Skv, int H,
                                                              float scale, float c) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_attn[];
  bf16* rowK = reinterpret_cast<bf16*>(smem_attn);           // [H][MK+1][8]
  bf16* rowV = rowK + (long)H * (MK + 1) * 8;                 // [H][MK+1][8]
  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int b = blockIdx.y, q0 = blockIdx.x * 32, E = H * 4, nthr = H * 64;
  zero_row(rowK, H, tid, nthr);
  zero_row(rowV, H, tid, nthr);
  const bool qok = q0 + r < Sq;
  const long qoff = ((long)b * Sq + q0 + (qok ? r : 0)) * E + h * 4;
  const bf16x8 fq = head_frag(q + qoff, hh == 0 && qok);
  const bf16x8 fdo = head_frag(dout + qoff, hh == 0 && qok);
  float dl = 0.f, ls = 0.f;
  if (qok) {
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(dout + qoff), o4 = *reinterpret_cast<const bf16x4*>(out + qoff);
    dl = (float)a[0] * (float)o4[0] + (float)a[1] * (float)o4[1] + (float)a[2] * (float)o4[2] + (float)a[3] * (float)o4[3];
    ls = lse[((long)b * H + h) * Sq + q0 + r] * 1.4426950408889634f;
    if (hh == 0) delta[((long)b * H + h) * Sq + q0 + r] = dl;
  }
  const long lstep = hh == 0 ? 32 * 8 : 0;
  const bf16* aK = rowK + ((long)h * (MK + 1) + (hh == 0 ? r : MK)) * 8;
  const bf16* aVr = rowV + ((long)h * (MK + 1) + (hh == 0 ? r : MK)) * 8;
  const bf16* aKt = rowK + (long)h * (MK + 1) * 8 + tr_base(lane);
  f32x16 acc = (f32x16)(0.f);
  StageRegs rk, rv;
  stage_load(rk, k, (long)b * Skv, min(MK, Skv), H, tid, nthr);
  stage_load(rv, v, (long)b * Skv, min(MK, Skv), H, tid, nthr);
  for (int j0 = 0; j0 < Skv; j0 += MK) {
    __syncthreads();
    stage_write(rk, rowK, 0u, H, tid, nthr);
    stage_write(rv, rowV, 0u, H, tid, nthr);
    __syncthreads();
    if (j0 + MK < Skv) {
      stage_load(rk, k, (long)b * Skv + j0 + MK, min(MK, Skv - j0 - MK), H, tid, nthr);
      stage_load(rv, v, (long)b * Skv + j0 + MK, min(MK, Skv - j0 - MK), H, tid, nthr);
    }
    const int nt = min(MK / 32, (Skv - j0 + 31) >> 5);
    for (int t = 0; t < nt; ++t) {
      const bf16x8 fk = *reinterpret_cast<const bf16x8*>(aK + t * lstep);
      const bf16x8 fv = *reinterpret_cast<const bf16x8*>(aVr + t * lstep);
      const bf16x8 fk0 = tr_frag(aKt + t * 256);
      const bf16x8 fk1 = tr_frag(aKt + t * 256 + 128);
      f32x16 s = mfma_bf16(fk, fq, (f32x16)(0.f));           // S^T[k][q]
      const f32x16 dp = mfma_bf16(fv, fdo, (f32x16)(0.f));   // dP^T[k][q] = V[k] . dO[q]
      // keys past Skv are zero rows: their ds is finite and meets a zero K^T column, so no masking is needed
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) s[reg] = __builtin_amdgcn_exp2f(fmaf(s[reg], c, -ls)) * (dp[reg] - dl);
      acc = mfma_bf16(fk0, pack8(s, 0), acc);                // dQ^T[d][q] += K^T[d][k] dS^T[k][q]
      acc = mfma_bf16(fk1, pack8(s, 1), acc);
    }
  }
  if (hh == 0 && qok) {
    bf16x4 ov;
    ov[0] = (bf16)(acc[0] * scale); ov[1] = (bf16)(acc[1] * scale); ov[2] = (bf16)(acc[2] * scale); ov[3] = (bf16)(acc[3] * scale);
    *reinterpret_cast<bf16x4*>(dq + qoff) = ov;
  }
}

// dk, dv: wave = (head, 32 keys); queries (+ dO, lse, delta) stream through LDS.
__global__ __launch_bounds__(512, 4) void attn_bwd_dkv_mfma_kernel(bf16* dk, bf16* dv, const bf16* dout, const bf16* q, const bf16* k,
                                                               const bf16* v, const float* lse, const float* delta, int Sq, int Skv,
                                                               int H, float scale, float c) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_attn[];
  bf16* rowQ = reinterpret_cast<bf16*>(smem_attn);           // [H][MK+1][8]
  bf16* rowD = rowQ + (long)H * (MK + 1) * 8;                 // [H][MK+1][8]
  float* sl = reinterpret_cast<float*>(rowD + (long)H * (MK + 1) * 8);   // [H][MK] lse * log2(e)  (+inf past Sq)
  float* sd = sl + (long)H * MK;                              // [H][MK] delta
  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int b = blockIdx.y, k0 = blockIdx.x * 32, E = H * 4, nthr = H * 64;
  zero_row(rowQ, H, tid, nthr);
  zero_row(rowD, H, tid, nthr);
  const bool kok = k0 + r < Skv;
  const long koff = ((long)b * Skv + k0 + (kok ? r : 0)) * E + h * 4;
  const bf16x8 fkB = head_frag(k + koff, hh == 0 && kok);
  const bf16x8 fvB = head_frag(v + koff, hh == 0 && kok);
  const long lstep = hh == 0 ? 32 * 8 : 0;
  const bf16* aQ = rowQ + ((long)h * (MK + 1) + (hh == 0 ? r : MK)) * 8;
  const bf16* aD = rowD + ((long)h * (MK + 1) + (hh == 0 ? r : MK)) * 8;
  const bf16* aQt = rowQ + (long)h * (MK + 1) * 8 + tr_base(lane);
  const bf16* aDt = rowD + (long)h * (MK + 1) * 8 + tr_base(lane);
  const float* pl = sl + (long)h * MK + 4 * hh;
  const float* pd = sd + (long)h * MK + 4 * hh;
  f32x16 ak = (f32x16)(0.f), av = (f32x16)(0.f);
  StageRegs rq, rd;
  float rl[MKR], rdl[MKR];
  auto load_ld = [&](int i0) {
#pragma unroll
    for (int i = 0; i < MKR; ++i) {
      const int idx = tid + i * nthr;
      const int rr = idx / H, hd = idx - rr * H;
      const bool ok = i0 + rr < Sq;
      rl[i] = ok ? lse[((long)b * H + hd) * Sq + i0 + rr] * 1.4426950408889634f : INFINITY;
      rdl[i] = ok ? delta[((long)b * H + hd) * Sq + i0 + rr] : 0.f;
    }
  };
  stage_load(rq, q, (long)b * Sq, min(MK, Sq), H, tid, nthr);
  stage_load(rd, dout, (long)b * Sq, min(MK, Sq), H, tid, nthr);
  load_ld(0);
  for (int i0 = 0; i0 < Sq; i0 += MK) {
    __syncthreads();
    stage_write(rq, rowQ, 0u, H, tid, nthr);
    stage_write(rd, rowD, 0u, H, tid, nthr);
#pragma unroll
    for (int i = 0; i < MKR; ++i) {
      const int idx = tid + i * nthr;
      const int rr = idx / H, hd = idx - rr * H;
      sl[(long)hd * MK + rr] = rl[i];
      sd[(long)hd * MK + rr] = rdl[i];
    }
    __syncthreads();
    if (i0 + MK < Sq) {
      stage_load(rq, q, (long)b * Sq + i0 + MK, min(MK, Sq - i0 - MK), H, tid, nthr);
      stage_load(rd, dout, (long)b * Sq + i0 + MK, min(MK, Sq - i0 - MK), H, tid, nthr);
      load_ld(i0 + MK);
    }
    const int nt = min(MK / 32, (Sq - i0 + 31) >> 5);
    for (int t = 0; t < nt; ++t) {
      const bf16x8 fqr = *reinterpret_cast<const bf16x8*>(aQ + t * lstep);
      const bf16x8 fdr = *reinterpret_cast<const bf16x8*>(aD + t * lstep);
      const bf16x8 fd0 = tr_frag(aDt + t * 256), fd1 = tr_frag(aDt + t * 256 + 128);
      const bf16x8 fq0 = tr_frag(aQt + t * 256), fq1 = tr_frag(aQt + t * 256 + 128);
      f32x16 s = mfma_bf16(fqr, fkB, (f32x16)(0.f));          // S[q][k]: rows = queries, columns = keys
      f32x16 dp = mfma_bf16(fdr, fvB, (f32x16)(0.f));         // dP[q][k] = dO[q] . V[k]
#pragma unroll
      for (int i = 0; i < 4; ++i) {                           // register quad i = queries 8i + 4hh .. +3 of the tile
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(pl + t * 32 + 8 * i);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(pd + t * 32 + 8 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float p = __builtin_amdgcn_exp2f(fmaf(s[4 * i + j], c, -l4[j]));    // 0 for queries past Sq (lse = +inf)
          s[4 * i + j] = p;
          dp[4 * i + j] = p * (dp[4 * i + j] - d4[j]);
        }
      }
      av = mfma_bf16(fd0, pack8(s, 0), av);                   // dV^T[d][k] += dO^T[d][q] P[q][k]
      av = mfma_bf16(fd1, pack8(s, 1), av);
      ak = mfma_bf16(fq0, pack8(dp, 0), ak);                  // dK^T[d][k] += Q^T[d][q] dS[q][k]
      ak = mfma_bf16(fq1, pack8(dp, 1), ak);
    }
  }
  if (hh == 0 && kok) {
    bf16x4 ok4, ov4;
    ok4[0] = (bf16)(ak[0] * scale); ok4[1] = (bf16)(ak[1] * scale); ok4[2] = (bf16)(ak[2] * scale); ok4[3] = (bf16)(ak[3] * scale);
    ov4[0] = (bf16)av[0]; ov4[1] = (bf16)av[1]; ov4[2] = (bf16)av[2]; ov4[3] = (bf16)av[3];
    *reinterpret_cast<bf16x4*>(dk + koff) = ok4;
    *reinterpret_cast<bf16x4*>(dv + koff) = ov4;
  }
}

static inline bool attn_mfma_ok(int H, const void* a, const void* b2, const void* c2, const void* d2) {
  static const bool off = getenv("HDMOE_ATTN_VALU") != nullptr;
  return !off && H >= 1 && H <= 8 && (((uintptr_t)a | (uintptr_t)b2 | (uintptr_t)c2 | (uintptr_t)d2) & 7) == 0;
}

static void attn_mfma_attrs() {                              // (the staged images exceed the 64 KB default with MK = 256)
  static bool done = false;
  if (done) return;
  done = true;
  (void)hipFuncSetAttribute((const void*)attn_fwd_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute((const void*)attn_bwd_dq_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

template <typename T, int D>
int attn_fwd_launch(void* out, float* lse, const void* q, const void* k, const void* v, const float* bias, int B, int Sq, int Skv,
                    int H, int Sb, hipStream_t st) {
  if constexpr (sizeof(T) == 2 && D == 4) {
    if (!bias && attn_mfma_ok(H, out, q, k, v)) {
      const size_t lds = (size_t)H * 2 * (MK + 1) * 16;
      attn_mfma_attrs();
      hipLaunchKernelGGL(attn_fwd_mfma_kernel, dim3(cdiv(Sq, 32), B), dim3(64 * H), lds, st, (bf16*)out, lse, (const bf16*)q, (const bf16*)k,
                         (const bf16*)v, Sq, Skv, H, 1.4426950408889634f / sqrtf((float)D));
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
      const size_t lds_q = (size_t)H * 2 * (MK + 1) * 16;
      const size_t lds_kv = (size_t)H * (2 * (MK + 1) * 16 + 2 * MK * 4);
      attn_mfma_attrs();
      hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel, dim3(cdiv(Sq, 32), B), dim3(64 * H), lds_q, st, (bf16*)dq, delta, (const bf16*)dout,
                         (const bf16*)out, (const bf16*)q, (const bf16*)k, (const bf16*)v, lse, Sq, Skv, H, scale, c);
      hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel, dim3(cdiv(Skv, 32), B), dim3(64 * H), lds_kv, st, (bf16*)dk, (bf16*)dv, (const bf16*)dout,
                         (const bf16*)q, (const bf16*)k, (const bf16*)v, lse, delta, Sq, Skv, H, scale, c);
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
