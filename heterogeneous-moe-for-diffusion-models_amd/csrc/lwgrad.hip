// Weight gradient of the POINTWISE layers (linear / 1x1 conv, stride 1): dW[g][o][i] += sum over the positions p of expert g's rows of
// dy[p][o] * x[p][i]  (autograd of MP_Conv with kernel () or (1,1), reference models/model_internals.py:253-275).
//
// A third of the step's launches are such layers (ViT linears, q/k/v/out projections, time / text embeddings of every block, 1x1 skips),
// and the tiled k x k kernel (conv.hip: conv_wgrad2) spent 20-40 us on each of them whatever their size: the contraction runs over
// positions and there is no halo to share, so the work is a tall-skinny GEMM that should cost its HBM read.  Here a wave streams
// 64-position slices of x and dy on its own:
//   bf16: 16-byte loads -> wave-private LDS rows -> both MFMA operands by transposing reads (ds_read_b64_tr_b16), 32x32x16 MFMAs;
//   fp32: no LDS at all -- for v_mfma_f32_32x32x2_f32 a lane supplies ONE element A[row = lane & 31][k = lane >> 5], which for
//         position-major tensors is exactly a coalesced 128-byte row load per half-wave (exact fp32 products; these layers are tiny).
// The four waves of a workgroup split the workgroup's slice range, add their accumulators through LDS and flush one [32*OT][32*IT]
// tile with float atomics (the slab is shared with the other position partitions, as in conv_wgrad2).
// Domain: Cin % 32 == 0, Cout % 32 == 0, no constant-one input channel; everything else stays with conv.hip.
#include <stdlib.h>
#include "common.h"
#include "conv_args.h"
#include "hdmoe.h"

namespace {

struct LwgArgs {
  const void* x; const void* dy; float* G[HDMOE_MAX_GROUPS]; const int* seg;
  int ngroups, N, I, O, upw;
  long HW;
};
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds_p4;

// partition slot -> (expert, slice range): experts take ceil(slices / upw) consecutive slots each
DEVI bool lwg_slot(const LwgArgs& a, int slot, int& g, long& p0, long& p1, long& u0, long& u1) {
  for (g = 0; g < a.ngroups; ++g) {
    const long r0 = a.seg ? a.seg[g] : 0, r1 = a.seg ? a.seg[g + 1] : a.N;
    p0 = r0 * a.HW; p1 = r1 * a.HW;
    const long units = (p1 - p0 + 63) >> 6;
    const long nch = (units + a.upw - 1) / a.upw;
    if (slot < nch) { u0 = (long)slot * a.upw; u1 = u0 + a.upw < units ? u0 + a.upw : units; return true; }
    slot -= (int)nch;
  }
  return false;
}

template <int OT, int IT>
DEVI void lwg_flush(f32x16 (&acc)[OT][IT], float* red, float* G, int o0, int i0, int I, int tid) {
  // red: [4 waves][OT * IT][16 regs][64 lanes] floats; lane-major so that both the write and the summing read are conflict-free
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int t = 0; t < OT; ++t)
#pragma unroll
    for (int u = 0; u < IT; ++u)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) red[((wave * OT * IT + t * IT + u) * 16 + reg) * 64 + lane] = acc[t][u][reg];
  __syncthreads();
  // every workgroup of the launch adds into the same [32*OT][32*IT] tile: start each one at a different element so that they do
  // not all queue on the same addresses at the same time
  const int rot = (int)((blockIdx.z * 131u + blockIdx.x * 17u) % (unsigned)(OT * IT * 4)) * 256;
  for (int e0 = tid; e0 < OT * IT * 1024; e0 += 256) {
    const int e = (e0 + rot) % (OT * IT * 1024);
    const int l = e & 63, reg = (e >> 6) & 15, tu = e >> 10;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += red[((w * OT * IT + tu) * 16 + reg) * 64 + l];
    const int o = o0 + 32 * (tu / IT) + acc_row(reg, l), i = i0 + 32 * (tu % IT) + (l & 31);
    atomicAdd(&G[(long)o * I + i], v);
  }
}

template <int OT, int IT>
__global__ __launch_bounds__(256) void lwg_bf16_kernel(LwgArgs a) {
  // per wave: dy sub-tiles [OT][64 positions][64 B], x sub-tiles [IT][64][64 B]; reused for the cross-wave reduction at the end
  constexpr int WB = (OT + IT) * 4096;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i0 = blockIdx.x * 32 * IT, o0 = blockIdx.y * 32 * OT;
  int g; long p0, p1, u0, u1;
  if (!lwg_slot(a, blockIdx.z, g, p0, p1, u0, u1)) return;
  const bf16* X = (const bf16*)a.x;
  const bf16* DY = (const bf16*)a.dy;
  unsigned char* mine = lds + wave * WB;
  uint4 sdy[4 * OT], sx[4 * IT];
  auto load = [&](long u) {
    const long pb = p0 + (u << 6);
#pragma unroll
    for (int k = 0; k < 4 * OT; ++k) {
      const int e = lane + 64 * k, q = e / (4 * OT), pc = e % (4 * OT);
      const long p = pb + q;
      sdy[k] = *reinterpret_cast<const uint4*>(DY + (p < p1 ? p : p0) * a.O + o0 + pc * 8);
    }
#pragma unroll
    for (int k = 0; k < 4 * IT; ++k) {
      const int e = lane + 64 * k, q = e / (4 * IT), pc = e % (4 * IT);
      const long p = pb + q;
      sx[k] = *reinterpret_cast<const uint4*>(X + (p < p1 ? p : p0) * a.I + i0 + pc * 8);
    }
    // (positions past the range were read from the range's first position; they are zeroed here, after all loads were issued)
#pragma unroll
    for (int k = 0; k < 4 * OT; ++k) if (pb + (lane + 64 * k) / (4 * OT) >= p1) sdy[k] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 4 * IT; ++k) if (pb + (lane + 64 * k) / (4 * IT) >= p1) sx[k] = make_uint4(0, 0, 0, 0);
  };
  auto store = [&]() {
#pragma unroll
    for (int k = 0; k < 4 * OT; ++k) {
      const int e = lane + 64 * k, q = e / (4 * OT), pc = e % (4 * OT);
      *reinterpret_cast<uint4*>(mine + (pc >> 2) * 4096 + q * 64 + (pc & 3) * 16) = sdy[k];
    }
#pragma unroll
    for (int k = 0; k < 4 * IT; ++k) {
      const int e = lane + 64 * k, q = e / (4 * IT), pc = e % (4 * IT);
      *reinterpret_cast<uint4*>(mine + OT * 4096 + (pc >> 2) * 4096 + q * 64 + (pc & 3) * 16) = sx[k];
    }
  };
  // transposing-read lane address inside a [rows][64 B] sub-tile: fragment = 8 consecutive positions (k) of channel lane & 31
  const int h = lane >> 5, q4 = (lane & 15) >> 2, col4 = (lane & 16) + 4 * (lane & 3);
  const int tlane = (8 * h + q4) * 64 + col4 * 2;
  auto tr2 = [&](const unsigned char* base) -> bf16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p4)(base));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p4)(base + 4 * 64));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  };
  f32x16 acc[OT][IT];
#pragma unroll
  for (int t = 0; t < OT; ++t)
#pragma unroll
    for (int u = 0; u < IT; ++u) acc[t][u] = (f32x16)(0.f);
  // the sub-tiles are WAVE-PRIVATE: the LDS keeps one wave's writes and reads in order, so a slice hand-over needs a compiler fence, not a
  // workgroup barrier (two per slice marched the four waves in lockstep until round 3); the one barrier is in front of the shared flush
  auto wfence = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  const long iters = (u1 - u0 + 3) >> 2;
  if (u0 + wave < u1) load(u0 + wave);
  for (long it = 0; it < iters; ++it) {
    const long u = u0 + wave + 4 * it;
    const bool valid = u < u1;
    wfence();
    if (valid) store();
    wfence();
    if (u + 4 < u1) load(u + 4);
    if (valid) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 fdy[OT], fx[IT];
#pragma unroll
        for (int t = 0; t < OT; ++t) fdy[t] = tr2(mine + t * 4096 + ks * 1024 + tlane);
#pragma unroll
        for (int v = 0; v < IT; ++v) fx[v] = tr2(mine + (OT + v) * 4096 + ks * 1024 + tlane);
#pragma unroll
        for (int t = 0; t < OT; ++t)
#pragma unroll
          for (int v = 0; v < IT; ++v) acc[t][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fdy[t], fx[v], acc[t][v], 0, 0, 0);
      }
    }
  }
  __syncthreads();                                            // every wave is done with its sub-tiles: the flush re-uses the whole buffer
  lwg_flush<OT, IT>(acc, reinterpret_cast<float*>(lds), a.G[g], o0, i0, a.I, tid);
}

template <int OT, int IT>
__global__ __launch_bounds__(256) void lwg_f32_kernel(LwgArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i0 = blockIdx.x * 32 * IT, o0 = blockIdx.y * 32 * OT;
  int g; long p0, p1, u0, u1;
  if (!lwg_slot(a, blockIdx.z, g, p0, p1, u0, u1)) return;
  const float* X = (const float*)a.x;
  const float* DY = (const float*)a.dy;
  const int r = lane & 31, h = lane >> 5;
  f32x16 acc[OT][IT];
#pragma unroll
  for (int t = 0; t < OT; ++t)
#pragma unroll
    for (int u = 0; u < IT; ++u) acc[t][u] = (f32x16)(0.f);
  // this wave's positions: the workgroup's range in 8-position steps, round-robin over the four waves
  const long pa = p0 + (u0 << 6), pe = (p0 + (u1 << 6)) < p1 ? (p0 + (u1 << 6)) : p1;
  for (long pb = pa + 8 * wave; pb < pe; pb += 32) {
    float dv[4][OT], xv[4][IT];
    bool okm[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {                              // loads from clamped addresses first, masks afterwards (a `cond ? load : 0`
      const long p = pb + 2 * s + h;                           //  form serialises the loads behind one another)
      okm[s] = p < pe;
      const long pc = okm[s] ? p : pa;
#pragma unroll
      for (int t = 0; t < OT; ++t) dv[s][t] = DY[pc * a.O + o0 + 32 * t + r];
#pragma unroll
      for (int v = 0; v < IT; ++v) xv[s][v] = X[pc * a.I + i0 + 32 * v + r];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int t = 0; t < OT; ++t) dv[s][t] = okm[s] ? dv[s][t] : 0.f;
#pragma unroll
      for (int v = 0; v < IT; ++v) xv[s][v] = okm[s] ? xv[s][v] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int v = 0; v < IT; ++v) acc[t][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(dv[s][t], xv[s][v], acc[t][v], 0, 0, 0);
  }
  lwg_flush<OT, IT>(acc, reinterpret_cast<float*>(lds), a.G[g], o0, i0, a.I, tid);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// k x k layers with a TINY input channel count (the stem: 4 latent channels -> 32, 3x3; reference model_config2.py:66-68 input_proj):
// taps * Cin <= 64 columns.  The tiled kernel pads Cin to 32 per tap (8x the work, 300 us); here the contraction over pixels runs on
// v_mfma_f32_32x32x2_f32 with A = dy[pixel][o] (a coalesced row load) and B = the im2col row of x built on the fly
// (column n = tap * Cin + i -> x[pixel + tap offset][i], zero outside the image): 2 MFMAs per 2 pixels per wave, HBM-bound on dy.
struct SwgArgs { const float* x; const float* dy; float* G; int N, H, W, Cin, Cout, k, pt, pl; long ppb; };

template <int NBLK>
__global__ __launch_bounds__(256) void swg_f32_kernel(SwgArgs a) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [4 waves][NBLK][16][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int o0 = blockIdx.y * 32;
  const long total = (long)a.N * a.H * a.W;
  const long p0 = (long)blockIdx.x * a.ppb, p1 = p0 + a.ppb < total ? p0 + a.ppb : total;
  const int ncols = a.k * a.k * a.Cin;
  int ky[NBLK], kx[NBLK], ci[NBLK]; bool colok[NBLK];
#pragma unroll
  for (int b = 0; b < NBLK; ++b) {
    const int n = 32 * b + r;
    colok[b] = n < ncols;
    const int tap = colok[b] ? n / a.Cin : 0;
    ci[b] = colok[b] ? n - tap * a.Cin : 0;
    ky[b] = tap / a.k - a.pt; kx[b] = tap % a.k - a.pl;
  }
  f32x16 acc[NBLK];
#pragma unroll
  for (int b = 0; b < NBLK; ++b) acc[b] = (f32x16)(0.f);
  // 8 k-steps (16 pixels of this wave) per trip: all their loads are issued before the first MFMA (one step at a time the loop ran at
  // one memory latency per step)
  for (long base = p0; base < p1; base += 64) {
    float dv[8], xv[8][NBLK];
    bool okm[8], inm[8][NBLK];
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const long p = base + 8 * st + 2 * wave + h;
      const bool ok = p < p1;
      const unsigned pc = (unsigned)(ok ? p : p0);               // 32-bit index arithmetic (64-bit div / mod cost ~500 instructions per step)
      const unsigned t = pc / (unsigned)a.W;
      const int xw = (int)(pc - t * (unsigned)a.W);
      const unsigned img = t / (unsigned)a.H;
      const int yh = (int)(t - img * (unsigned)a.H);
      // unconditional loads from clamped (always valid) addresses, masks applied after ALL loads of the trip are in flight: a
      // `cond ? load : 0` form made the compiler wait for every load before issuing the next (24 serial memory latencies per trip)
      dv[st] = a.dy[(long)pc * a.Cout + o0 + r];
      okm[st] = ok;
#pragma unroll
      for (int b = 0; b < NBLK; ++b) {
        const int yy = yh + ky[b], xx = xw + kx[b];
        const bool in = ok && colok[b] && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
        inm[st][b] = in;
        xv[st][b] = a.x[in ? (((long)img * a.H + yy) * a.W + xx) * a.Cin + ci[b] : 0];
      }
    }
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      dv[st] = okm[st] ? dv[st] : 0.f;
#pragma unroll
      for (int b = 0; b < NBLK; ++b) xv[st][b] = inm[st][b] ? xv[st][b] : 0.f;
    }
#pragma unroll
    for (int st = 0; st < 8; ++st)
#pragma unroll
      for (int b = 0; b < NBLK; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(dv[st], xv[st][b], acc[b], 0, 0, 0);
  }
#pragma unroll
  for (int b = 0; b < NBLK; ++b)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) red[((wave * NBLK + b) * 16 + reg) * 64 + lane] = acc[b][reg];
  __syncthreads();
  const int rot = (int)((blockIdx.x * 37u) % (unsigned)(NBLK * 4)) * 256;      // stagger the workgroups' atomic flushes (see lwg_flush)
  for (int e0 = tid; e0 < NBLK * 1024; e0 += 256) {
    const int e = (e0 + rot) % (NBLK * 1024);
    const int l = e & 63, reg = (e >> 6) & 15, b = e >> 10;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += red[((w * NBLK + b) * 16 + reg) * 64 + l];
    const int o = o0 + acc_row(reg, l), n = 32 * b + (l & 31);
    if (n < ncols) {
      const int tap = n / a.Cin, i = n - tap * a.Cin;
      atomicAdd(&a.G[((long)tap * a.Cout + o) * a.Cin + i], v);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// k x k (k = 1 or 3) bf16 layers with a TINY output channel count (Cout <= 4: the output head 32 -> IN_in_channels 3x3 and the 2-way
// gate 32 -> 2 1x1, reference model_config2.py output_proj / gate2).  The tiled kernel pads Cout to 32 rows per tap and took ~80 us
// per layer on the serial tail of the step for 0.6 GFLOP; here it is plain vector arithmetic: thread = (channel pair of x, one of 16
// pixel slices of an 8-row band), 9 x 4 x 2 fp32 accumulators, x straight from HBM (4 bytes per pixel-thread, 64 B per pixel over the
// 16 lanes of a channel chunk), dy with its halo converted to fp32 [pixel][4] in LDS (one 16-byte broadcast read per tap).  The
// slices meet by two cross-lane steps + LDS, one atomic flush of the <= 1152 words per workgroup.
struct TowgArgs { const bf16* x; const bf16* dy; float* G; int N, H, W, Cin, Cout, pad, TH, tiles_y, ntiles, wshift; };

template <int K>
__global__ __launch_bounds__(256) void towg_bf16_kernel(TowgArgs a) {
  constexpr int T = K * K;
  typedef __attribute__((ext_vector_type(2))) float f2;
  extern __shared__ __attribute__((aligned(16))) float tsm[];  // dy halo tile [(TH + K - 1)][(W + K - 1)][4], later the 4 waves' sums
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cp = tid & 15, sl = tid >> 4;
  const int c0 = blockIdx.y * 32 + 2 * cp;
  const int HWp = a.W + K - 1, HHp = a.TH + K - 1;
  f2 acc[T][4];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int o = 0; o < 4; ++o) acc[t][o] = (f2)(0.f);
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int n = tile / a.tiles_y, y0 = (tile - n * a.tiles_y) * a.TH;
    __syncthreads();                                          // the previous tile's readers are done
    for (int e = tid; e < HHp * HWp; e += 256) {
      const int hy = e / HWp, hx = e - hy * HWp;
      const int iy = y0 + hy - a.pad, ix = hx - a.pad;
      f32x4 v = (f32x4)(0.f);
      if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
        const bf16* d = a.dy + (((long)n * a.H + iy) * a.W + ix) * a.Cout;
        for (int o = 0; o < a.Cout; ++o) v[o] = (float)d[o];
      }
      *reinterpret_cast<f32x4*>(tsm + 4 * e) = v;
    }
    __syncthreads();
    const int rows = a.H - y0 < a.TH ? a.H - y0 : a.TH, npx = rows * a.W;
    const bf16* xb = a.x + ((long)n * a.H + y0) * a.W * a.Cin + c0;
#pragma unroll 2
    for (int p = sl; p < npx; p += 16) {
      const int y = a.wshift >= 0 ? p >> a.wshift : p / a.W, xx = p - y * a.W;
      const unsigned xw = *reinterpret_cast<const unsigned*>(xb + (long)p * a.Cin);
      const f2 xf = {__builtin_bit_cast(float, xw << 16), __builtin_bit_cast(float, xw & 0xFFFF0000u)};
      // dW[tap][o][i] = sum_p' x[p'][i] dy[p' - tap + pad][o]: halo coordinates (y + 2 pad - ty, x + 2 pad - tx)
      const float* dbase = tsm + 4 * ((y + K - 1) * HWp + xx + K - 1);
#pragma unroll
      for (int ty = 0; ty < K; ++ty)
#pragma unroll
        for (int tx = 0; tx < K; ++tx) {
          const f32x4 d = *reinterpret_cast<const f32x4*>(dbase - 4 * (ty * HWp + tx));
#pragma unroll
          for (int o = 0; o < 4; ++o) acc[ty * K + tx][o] += xf * d[o];
        }
    }
  }
  // the 4 pixel slices of a wave (lanes with equal cp), then the 4 waves
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float v = acc[t][o][j];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        acc[t][o][j] = v;
      }
  __syncthreads();
  if (lane < 16) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int j = 0; j < 2; ++j) tsm[((wave * T + t) * 4 + o) * 32 + 2 * lane + j] = acc[t][o][j];
  }
  __syncthreads();
  for (int e = tid; e < T * 4 * 32; e += 256) {
    const int i = e & 31, o = (e >> 5) & 3, t = e >> 7;
    if (o < a.Cout) {
      const float v = tsm[e] + tsm[T * 128 + e] + tsm[2 * T * 128 + e] + tsm[3 * T * 128 + e];
      atomicAdd(&a.G[((long)t * a.Cout + o) * a.Cin + blockIdx.y * 32 + i], v);
    }
  }
}

}  // namespace

// k x k bf16 weight gradient for Cout <= 4 (k = 1 or 3, one expert, stride 1, "same" geometry): G [tap][Cout][Cin] += .  Same return convention.
int towg_try_launch(const void* x, const void* dy, float* G, int N, int H, int W, int Cin, int Cout, int k, int pt, int pl, int dtype,
                    hipStream_t stream) {
  static const bool off = getenv("HDMOE_TOWG") && atoi(getenv("HDMOE_TOWG")) == 0;
  if (off || dtype != HDMOE_BF16 || Cout < 1 || Cout > 4 || Cin % 32 || (k != 1 && k != 3) || pt != (k - 1) / 2 || pl != (k - 1) / 2) return 1;
  if (!x || !dy || !G || ((uintptr_t)x & 3)) return 1;
  TowgArgs a;
  a.x = (const bf16*)x; a.dy = (const bf16*)dy; a.G = G; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.pad = (k - 1) / 2;
  if (k == 1) {                                              // pointwise: no halo, the image shape is irrelevant (ops flattens such tensors
    const long total = (long)N * H * W;                       // to ONE row of N * H * W pixels): re-tile as rows of 256 pixels
    if (total % 256 == 0 && total / 256 < (1l << 30)) { a.N = 1; a.H = (int)(total / 256); a.W = 256; N = 1; H = a.H; W = 256; }
  }
  a.TH = H < 8 ? H : 8;
  if (k == 1 && H >= 4 && (long)N * ((H + 7) / 8) < 256) a.TH = 4;   // (enough tiles for every CU)
  while (a.TH > 1 && (size_t)(a.TH + k - 1) * (W + k - 1) * 16 > 48 * 1024) a.TH >>= 1;
  if ((size_t)(a.TH + k - 1) * (W + k - 1) * 16 > 48 * 1024) return 1;
  a.tiles_y = (H + a.TH - 1) / a.TH;
  const long ntiles = (long)N * a.tiles_y;
  if (ntiles >= (1l << 30)) return 1;
  a.ntiles = (int)ntiles;
  a.wshift = -1;
  for (int sft = 0; sft < 9; ++sft) if ((1 << sft) == W) a.wshift = sft;
  const int cb = Cin / 32;
  long nb = 256 / cb; if (nb < 1) nb = 1;                    // ~one workgroup per CU: every workgroup ends with a flush of the same words
  if (nb > ntiles) nb = ntiles;
  const size_t halo = (size_t)(a.TH + k - 1) * (W + k - 1) * 16, red = (size_t)4 * k * k * 128 * 4;
  const size_t lds = halo > red ? halo : red;
  const dim3 grid((unsigned)nb, (unsigned)cb);
  if (k == 3) hipLaunchKernelGGL(towg_bf16_kernel<3>, grid, dim3(256), lds, stream, a);
  else hipLaunchKernelGGL(towg_bf16_kernel<1>, grid, dim3(256), lds, stream, a);
  return hdmoe_launch_status();
}

// k x k fp32 layer with taps * Cin <= 64 (one expert, stride 1, "same" geometry): G [tap][Cout][Cin] += .  Same return convention.
int swg_try_launch(const void* x, const void* dy, float* G, int N, int H, int W, int Cin, int Cout, int k, int pt, int pl, int dtype,
                   hipStream_t stream) {
  static const bool off = getenv("HDMOE_SWG") && atoi(getenv("HDMOE_SWG")) == 0;
  if (off || dtype != HDMOE_F32 || Cout % 32 || k * k * Cin > 64 || !x || !dy || !G || (long)N * H * W >= (1l << 31)) return 1;
  SwgArgs a;
  a.x = (const float*)x; a.dy = (const float*)dy; a.G = G; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.k = k; a.pt = pt; a.pl = pl;
  const long total = (long)N * H * W;
  long blocks = 256 / (Cout / 32); if (blocks < 1) blocks = 1;       // one workgroup per CU: more only adds atomic flushes of the same 1152 words
  long ppb = (total + blocks - 1) / blocks; ppb = (ppb + 63) / 64 * 64;
  a.ppb = ppb;
  const dim3 grid((unsigned)((total + ppb - 1) / ppb), Cout / 32);
  const int NBLK = (k * k * Cin + 31) / 32;
  if (NBLK == 1) hipLaunchKernelGGL(swg_f32_kernel<1>, grid, dim3(256), 4 * 1 * 4096, stream, a);
  else hipLaunchKernelGGL(swg_f32_kernel<2>, grid, dim3(256), 4 * 2 * 4096, stream, a);
  return hdmoe_launch_status();
}

// Returns HDMOE_OK after launching, a negative status on a launch error, or 1 when the layer is outside this file's domain.
int lwg_try_launch(const void* x, const void* dy, float* const* G, const int* seg, int ngroups, int N, long HW, int Cin, int Cout,
                   int dtype, hipStream_t stream) {
  static const bool off = getenv("HDMOE_LWG") && atoi(getenv("HDMOE_LWG")) == 0;
  if (off || Cin % 32 || Cout % 32 || (dtype != HDMOE_BF16 && dtype != HDMOE_F32)) return 1;
  if (((uintptr_t)x | (uintptr_t)dy) & 15) return 1;
  LwgArgs a;
  a.x = x; a.dy = dy; a.seg = seg; a.ngroups = ngroups; a.N = N; a.HW = HW; a.I = Cin; a.O = Cout;
  for (int g = 0; g < HDMOE_MAX_GROUPS; ++g) a.G[g] = G[g < ngroups ? g : 0];
  const int OT = Cout % 64 == 0 ? 2 : 1, IT = Cin % 64 == 0 ? 2 : 1;
  const long ib = Cin / (32 * IT), ob = Cout / (32 * OT);
  const long units = ((long)N * HW + 63) / 64 + ngroups;               // 64-position slices (upper bound over the experts' ragged ends)
  long upw = (units * ib * ob + 511) / 512;                              // ~512 workgroups per launch
  if (upw < 4) upw = 4;                                                  // at least one slice per wave
  if (upw > 4096) upw = 4096;
  a.upw = (int)upw;
  const long slots = units / upw + ngroups + 1;
  if (slots > 65535 || ob > 65535) return 1;
  dim3 grid((unsigned)ib, (unsigned)ob, (unsigned)slots);
  const size_t red = (size_t)4 * OT * IT * 4096;
#define LWG_GO(KERNEL, LDS)                                                                                   \
  do {                                                                                                        \
    if (OT == 2 && IT == 2) hipLaunchKernelGGL((KERNEL<2, 2>), grid, dim3(256), LDS, stream, a);               \
    else if (OT == 2) hipLaunchKernelGGL((KERNEL<2, 1>), grid, dim3(256), LDS, stream, a);                     \
    else if (IT == 2) hipLaunchKernelGGL((KERNEL<1, 2>), grid, dim3(256), LDS, stream, a);                     \
    else hipLaunchKernelGGL((KERNEL<1, 1>), grid, dim3(256), LDS, stream, a);                                  \
  } while (0)
  if (dtype == HDMOE_BF16) {
    const size_t stage = (size_t)4 * (OT + IT) * 4096;
    const size_t lds = stage > red ? stage : red;
    LWG_GO(lwg_bf16_kernel, lds);
  } else {
    LWG_GO(lwg_f32_kernel, red);
  }
  return hdmoe_launch_status();
}
