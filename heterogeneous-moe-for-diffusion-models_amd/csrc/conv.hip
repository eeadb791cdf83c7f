// K3: magnitude-preserving weight prep + grouped implicit-GEMM convolution (fwd / dgrad) + wgrad.
//
// Replaces MP_Conv.forward (reference models/model_internals.py:253-275: normalize -> gain/sqrt(fan_in)
// -> cast -> F.conv2d / F.linear) and its autograd.  Activations are NHWC ([N][H][W][C], C contiguous);
// "grouped" = the sample rows of one launch belong to up to 8 experts with their own weights and their
// own kernel size (heterogeneous 3x3 / 5x5 / 7x7 experts in one launch); seg[g]..seg[g+1] is the row
// range of group g (device memory, produced by the dispatch plan, so no host sync is needed).
//
// GEMM view (fwd and dgrad):  M = output pixels, N = output channels, K = taps * input channels.
//   conv_fwd5_kernel  (default)   256-pixel x 32/64-channel workgroup tiles, halo + weight rows staged in LDS per
//                                 (64-byte channel chunk, kernel-row group) stage, next stage prefetched into registers,
//                                 MFMA 32x32x16 bf16 / 8x 32x32x2 f32, epilogue through LDS as 16-byte stores
//   conv_fwd3_kernel              same walk with a per-tile staging plan; layers with a channel tail / implicit ones channel
//   conv_fwd2_kernel, conv_fwd_kernel   earlier generations, kept for odd shapes (tiny images, stride > 1) and A/B runs
// wgrad:  M = out channels, N = in channels, K = pixels.
//   conv_wgrad2_kernel            pixel tile + halo in LDS, taps owned per wave, transposing LDS reads (bf16), one fp32 atomic
//                                 flush per workgroup into a [tap][O][I] slab (contiguous I: full-rate atomic shape,
//                                 MI355X_MICROARCH "Global float atomics"); conv_wgrad_kernel for stride > 1
#include <stdlib.h>
#include "common.h"
#include "conv_args.h"
#include "hdmoe.h"

namespace {

// ------------------------------------------------------------------ weight prep
struct WprepArgs {
  float* w_raw[HDMOE_MAX_GROUPS];        // (O, I, kh, kw) fp32 (written when mutate)
  const float* gain_ptr[HDMOE_MAX_GROUPS];
  void* wf;                              // [g][tap][O][Ipad]
  void* wd;                              // [g][tap'][I][Opad]   (may be null)
  long wf_stride, wd_stride;             // elements per group
  int kh[HDMOE_MAX_GROUPS], kw[HDMOE_MAX_GROUPS];
  int O, I, Ipad, Opad;
  float gain_val;
  int normalize, mutate, flip;
  long wf_plane, wd_plane;               // > 0: split-bf16 images (T = bf16): the lo plane lies this many elements behind the hi plane
};

template <typename T>
__global__ __launch_bounds__(128) void wprep_fwd_kernel(WprepArgs a) {
  __shared__ float sm[16];
  const int o = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
  const int taps = a.kh[g] * a.kw[g];
  const int fan = a.I * taps;
  T* wf = (T*)a.wf + (long)g * a.wf_stride;
  T* wd = a.wd ? (T*)a.wd + (long)g * a.wd_stride : nullptr;
  if (o >= a.O) {                                   // pad rows of the dgrad layout
    if (wd && o < a.Opad)
      for (int e = tid; e < fan; e += blockDim.x) {
        const int i = e / taps, t = e % taps;
        const int td = a.flip ? taps - 1 - t : t;
        wd[((long)td * a.I + i) * a.Opad + o] = from_f<T>(0.f);
      }
    return;
  }
  float* w = a.w_raw[g] + (long)o * fan;
  float scale = 1.f;
  if (a.normalize) {
    const float c = rsqrtf((float)fan);
    float ss = 0.f;
    for (int e = tid; e < fan; e += blockDim.x) { const float v = w[e]; ss += v * v; }
    ss = block_sum(ss, sm);
    float inv = 1.f / (1e-4f + sqrtf(ss) * c);
    if (a.mutate) {                                 // reference: weights.copy_(normalize(w)) then normalize again
      for (int e = tid; e < fan; e += blockDim.x) w[e] = w[e] * inv;
      __syncthreads();
      float s2 = 0.f;
      for (int e = tid; e < fan; e += blockDim.x) { const float v = w[e]; s2 += v * v; }
      s2 = block_sum(s2, sm);
      inv = 1.f / (1e-4f + sqrtf(s2) * c);
    }
    const float gain = a.gain_val * (a.gain_ptr[g] ? *a.gain_ptr[g] : 1.f);
    scale = inv * gain * c;
  }
  for (int e = tid; e < fan; e += blockDim.x) {
    const int i = e / taps, t = e % taps;
    const float vf = w[e] * scale;
    const T v = from_f<T>(vf);
    const long fi = ((long)t * a.O + o) * a.Ipad + i;
    wf[fi] = v;
    const int td = a.flip ? taps - 1 - t : t;
    const long di = ((long)td * a.I + i) * a.Opad + o;
    if (wd) wd[di] = v;
    if (a.wf_plane > 0) {                                   // split-bf16: lo = bf16(v - hi)
      const T lo = from_f<T>(vf - to_f(v));
      wf[a.wf_plane + fi] = lo;
      if (wd) wd[a.wd_plane + di] = lo;
    }
  }
  for (int e = tid; e < (a.Ipad - a.I) * taps; e += blockDim.x) {
    const int t = e / (a.Ipad - a.I), i = a.I + e % (a.Ipad - a.I);
    wf[((long)t * a.O + o) * a.Ipad + i] = from_f<T>(0.f);
    if (a.wf_plane > 0) wf[a.wf_plane + ((long)t * a.O + o) * a.Ipad + i] = from_f<T>(0.f);
  }
}

struct WprepBwdArgs {
  const float* w_raw[HDMOE_MAX_GROUPS];
  const float* gain_ptr[HDMOE_MAX_GROUPS];
  const float* G[HDMOE_MAX_GROUPS];      // [tap][O][I] fp32
  float* dw[HDMOE_MAX_GROUPS];           // (O, I, kh, kw) fp32
  float* dgain[HDMOE_MAX_GROUPS];        // scalar accumulators (may be null)
  int kh[HDMOE_MAX_GROUPS], kw[HDMOE_MAX_GROUPS];
  int O, I;
  float gain_val;
  int normalize;
};

__global__ __launch_bounds__(128) void wprep_bwd_kernel(WprepBwdArgs a) {
  __shared__ float sm[16];
  const int o = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
  const int taps = a.kh[g] * a.kw[g];
  const int fan = a.I * taps;
  const float* w = a.w_raw[g] + (long)o * fan;
  const float* G = a.G[g];
  float* dw = a.dw[g] + (long)o * fan;
  if (!a.normalize) {
    for (int e = tid; e < fan; e += blockDim.x) {
      const int i = e / taps, t = e % taps;
      dw[e] = G[((long)t * a.O + o) * a.I + i];
    }
    return;
  }
  const float c = rsqrtf((float)fan);
  float ss = 0.f, gw = 0.f;
  for (int e = tid; e < fan; e += blockDim.x) {
    const int i = e / taps, t = e % taps;
    const float v = w[e];
    ss += v * v;
    gw += v * G[((long)t * a.O + o) * a.I + i];
  }
  ss = block_sum(ss, sm);
  gw = block_sum(gw, sm);
  const float n = sqrtf(ss);
  const float d = 1e-4f + n * c;
  const float gain = a.gain_val * (a.gain_ptr[g] ? *a.gain_ptr[g] : 1.f);
  const float s = gain * c;
  const float k1 = s / d;
  const float k2 = n > 0.f ? s * c * gw / (d * d * n) : 0.f;
  for (int e = tid; e < fan; e += blockDim.x) {
    const int i = e / taps, t = e % taps;
    dw[e] = k1 * G[((long)t * a.O + o) * a.I + i] - k2 * w[e];
  }
  if (a.dgain[g] && tid == 0) atomicAdd(a.dgain[g], a.gain_val * c * gw / d);   // d/dgain of sum(G * w*gain*c/d)
}

// ------------------------------------------------------------------ conv forward / dgrad
DEVI int find_group(const int* seg, int ngroups, int n) {
  if (!seg) return 0;
  int g = -1;
  for (int i = 0; i < ngroups; ++i)
    if (n >= seg[i] && n < seg[i + 1]) g = i;
  return g;
}

template <typename T, bool VEC>
DEVI void load_act8(Frag8<T>& f, const T* px, int c, int Cphys, bool inb, int ones) {
  // 8 channels c..c+7 of one pixel; channel Cphys is the implicit ones channel when `ones`
  if (VEC) {
    if (inb && c + 8 <= Cphys) { load8(f, px + c); return; }
  }
  f.zero();
  if (!inb) return;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int cc = c + j;
    if (cc < Cphys) f.set(j, to_f(px[cc]));
    else if (ones && cc == Cphys) f.set(j, 1.f);
  }
}

template <typename T, int NB, bool VEC>
__global__ __launch_bounds__(256) void conv_fwd_kernel(ConvArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n = blockIdx.y + a.n0;                       // rows beyond 65535 come as further launches (a.n0)
  const int g = find_group(a.seg, a.ngroups, n);
  if (g < 0) return;
  const int HWo = a.Ho * a.Wo;
  const int p0 = (blockIdx.x * 4 + wave) * 32;
  if (p0 >= HWo) return;
  const int nbase = blockIdx.z * (32 * NB);
  const int p = p0 + r;
  const bool pvalid = p < HWo;
  const int oy = pvalid ? p / a.Wo : 0, ox = pvalid ? p % a.Wo : 0;
  const int kh = a.kh[g], kw = a.kw[g];
  const T* x = (const T*)a.x + (long)n * a.H * a.W * a.Cphys;
  const T* w = (const T*)a.w + (long)g * a.wstride;

  f32x16 acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) acc[b] = (f32x16)(0.f);

  for (int ky = 0; ky < kh; ++ky) {
    const int iy = oy * a.stride + ky - a.pt[g];
    for (int kx = 0; kx < kw; ++kx) {
      const int ix = ox * a.stride + kx - a.pl[g];
      const bool inb = pvalid && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const T* px = x + ((long)iy * a.W + ix) * a.Cphys;
      const T* wt = w + (long)(ky * kw + kx) * a.Cout * a.Ipad;
      for (int c0 = 0; c0 < a.Ipad; c0 += 16) {
        Frag8<T> fa;
        load_act8<T, VEC>(fa, px, c0 + 8 * h, a.Cphys, inb, a.ones);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int co = nbase + 32 * b + r;
          Frag8<T> fb;
          if (co < a.Cout) load8(fb, wt + (long)co * a.Ipad + c0 + 8 * h);
          else fb.zero();
          mma32(acc[b], fa, fb);
        }
      }
    }
  }
  T* y = (T*)a.y + (long)n * HWo * a.Cstore;
  const T* res = a.res ? (const T*)a.res + (long)n * HWo * a.Cstore : nullptr;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int co = nbase + 32 * b + r;
    if (co >= a.Cstore) continue;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int pp = p0 + acc_row(reg, lane);
      if (pp < HWo) {
        float v = a.alpha * acc[b][reg];
        if (res) v += a.beta * to_f(res[(long)pp * a.Cstore + co]);
        y[(long)pp * a.Cstore + co] = from_f<T>(v);
      }
    }
  }
}

// ------------------------------------------------------------------ conv forward / dgrad v2 (stride 1): LDS-staged
// One workgroup = one (sample row, 256-pixel tile, 32*NT output channels).  K is walked in 64-byte channel chunks
// (32 bf16 / 16 fp32): per chunk the input halo tile [(TH+kh-1) x (TW+kw-1) px][chunk] is staged in LDS ONCE and then
// read kh*kw times (the reference's im2col re-reads become LDS reads); the weights of one kernel row (kw taps) are
// staged per ky and shared by the 4 waves.  Each wave owns 64 pixels x 32*NT channels (2 x NT MFMA 32x32 tiles), so
// every weight fragment is reused twice from registers.  LDS rows are padded 64 -> 80 bytes: the 16 lanes of a
// ds_read_b128 group then fall on 16 distinct 16-byte slots (5 is coprime to 16) => conflict-free.
constexpr int CV2_MT = 2;

template <typename T, int NT, bool VEC>
__global__ __launch_bounds__(256) void conv_fwd2_kernel(ConvArgs a, int TH, int TW, int tiles_x, int halo_cap) {
  constexpr int ESZ = sizeof(T), VW = 16 / ESZ, KC = 64 / ESZ, PSE = KC + VW, KS = KC / 16, NCH = KC / VW;
  constexpr int PT = 4 * CV2_MT * 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* sA = reinterpret_cast<T*>(smem_raw);                    // [halo px][PSE]
  T* sB = sA + (long)halo_cap * PSE;                         // [kw][32*NT][PSE]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n = blockIdx.y + a.n0;                       // rows beyond 65535 come as further launches (a.n0)
  const int g = find_group(a.seg, a.ngroups, n);
  if (g < 0) return;
  const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * TW;
  const int nbase = blockIdx.z * (32 * NT);
  const int kh = a.kh[g], kw = a.kw[g], pt = a.pt[g], pl = a.pl[g];
  const int HWp = TW + kw - 1, HHp = TH + kh - 1;
  const int npx = TH * TW;                                   // <= PT
  const T* x = (const T*)a.x + (long)n * a.H * a.W * a.Cphys;
  const T* w = (const T*)a.w + (long)g * a.wstride;
  const T zero = from_f<T>(0.f);

  // this lane's pixels (one per M-tile) and their halo base offsets
  int abase[CV2_MT];
  bool pval[CV2_MT];
  int oyv[CV2_MT], oxv[CV2_MT];
#pragma unroll
  for (int m = 0; m < CV2_MT; ++m) {
    const int q = wave * (32 * CV2_MT) + m * 32 + r;
    const int qc = q < npx ? q : npx - 1;
    const int ty = qc / TW, tx = qc - ty * TW;
    oyv[m] = ty0 + ty; oxv[m] = tx0 + tx;
    pval[m] = q < npx && oyv[m] < a.Ho && oxv[m] < a.Wo;
    abase[m] = (ty * HWp + tx) * PSE + 8 * h;
  }
  f32x16 acc[CV2_MT][NT];
#pragma unroll
  for (int m = 0; m < CV2_MT; ++m)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[m][b] = (f32x16)(0.f);

  for (int c0 = 0; c0 < a.Ipad; c0 += KC) {
    __syncthreads();                                         // previous chunk's readers done (sA and sB)
    // ---- stage the halo tile for channels [c0, c0+KC)
    for (int hy = 0; hy < HHp; ++hy) {
      const int iy = ty0 + hy - pt;
      const bool rowin = iy >= 0 && iy < a.H;
      for (int e = tid; e < HWp * NCH; e += 256) {
        const int hx = e / NCH, cc = (e - hx * NCH) * VW;
        const int ix = tx0 + hx - pl, ci = c0 + cc;
        T* dst = sA + (hy * HWp + hx) * PSE + cc;
        const bool inb = rowin && ix >= 0 && ix < a.W;
        const T* src = x + ((long)iy * a.W + ix) * a.Cphys + ci;
        if (VEC && inb && ci + VW <= a.Cphys) {
          *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
        } else {
#pragma unroll
          for (int j = 0; j < VW; ++j) {
            T v = zero;
            if (inb) {
              if (ci + j < a.Cphys) v = src[j];
              else if (a.ones && ci + j == a.Cphys) v = from_f<T>(1.f);
            }
            dst[j] = v;
          }
        }
      }
    }
    for (int ky = 0; ky < kh; ++ky) {
      if (ky) __syncthreads();                               // readers of the previous kernel row's weights done
      // ---- stage weights of taps (ky, 0..kw-1): [kx][co][KC]
      for (int e = tid; e < kw * 32 * NT * NCH; e += 256) {
        const int cc = (e % NCH) * VW; const int row = e / NCH;
        const int co = row % (32 * NT), kx = row / (32 * NT);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (nbase + co < a.Cout && c0 + cc < a.Ipad)
          v = *reinterpret_cast<const uint4*>(w + ((long)(ky * kw + kx) * a.Cout + nbase + co) * a.Ipad + c0 + cc);
        *reinterpret_cast<uint4*>(sB + row * PSE + cc) = v;
      }
      __syncthreads();
      for (int kx = 0; kx < kw; ++kx) {
        const int aoff = (ky * HWp + kx) * PSE;
        const T* bt = sB + (kx * 32 * NT + r) * PSE + 8 * h;
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) {
          Frag8<T> fa[CV2_MT], fb[NT];
#pragma unroll
          for (int m = 0; m < CV2_MT; ++m) load8(fa[m], sA + abase[m] + aoff + 16 * s2);
#pragma unroll
          for (int b = 0; b < NT; ++b) load8(fb[b], bt + b * 32 * PSE + 16 * s2);
#pragma unroll
          for (int m = 0; m < CV2_MT; ++m)
#pragma unroll
            for (int b = 0; b < NT; ++b) mma32(acc[m][b], fa[m], fb[b]);
        }
      }
    }
  }
  // ---- epilogue: lanes = channels; register rows = pixels of the M-tile
  const long img = (long)n * a.Ho * a.Wo;
  T* y = (T*)a.y;
  const T* res = (const T*)a.res;
#pragma unroll
  for (int m = 0; m < CV2_MT; ++m) {
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      const int co = nbase + 32 * b + r;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        // pixel of accumulator row `reg`: lanes only know their own pixel, so recompute from the tile-local index
        const int q = wave * (32 * CV2_MT) + m * 32 + acc_row(reg, lane);
        if (q < npx && co < a.Cstore) {
          const int ty = q / TW, tx = q - ty * TW;
          const int oy = ty0 + ty, ox = tx0 + tx;
          if (oy < a.Ho && ox < a.Wo) {
            const long idx = (img + (long)oy * a.Wo + ox) * a.Cstore + co;
            float v = a.alpha * acc[m][b][reg];
            if (res) v += a.beta * to_f(res[idx]);
            y[idx] = from_f<T>(v);
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------ conv forward / dgrad v3: v2 + register prefetch
// Same tiling as v2, but the K walk is a flat sequence of stages (64-byte channel chunk x group of `tg` kernel rows) and
// the global loads of stage s+1 (weights, plus the halo tile when a new chunk starts) are issued into registers right
// after stage s is published, so they fly under stage s's MFMAs (guide G15, register-staged "issue early / write late").
template <typename T, int NT>
__global__ __launch_bounds__(256) void conv_fwd3_kernel(ConvArgs a, int TH, int TW, int tiles_x, int halo_cap, int tg) {
  constexpr int ESZ = sizeof(T), VW = 16 / ESZ, KC = 64 / ESZ, PSE = KC + VW, KS = KC / 16, NCH = KC / VW;
  constexpr int NWR = 9, NHR = 7;                           // 16-B chunks per thread held in registers (weights / halo)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* sA = reinterpret_cast<T*>(smem_raw);                    // [halo px][PSE]
  T* sB = sA + (long)halo_cap * PSE;                         // [tg*kw][32*NT][PSE]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n = blockIdx.y + a.n0;                       // rows beyond 65535 come as further launches (a.n0)
  const int g = find_group(a.seg, a.ngroups, n);
  if (g < 0) return;
  const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * TW;
  const int nbase = blockIdx.z * (32 * NT);
  const int kh = a.kh[g], kw = a.kw[g], pt = a.pt[g], pl = a.pl[g];
  const int HWp = TW + kw - 1, HHp = TH + kh - 1;
  const int npx = TH * TW;
  const int nhal = HHp * HWp * NCH;
  const int magic_hw = (1 << 20) / HWp + 1;
  const T* x = (const T*)a.x + (long)n * a.H * a.W * a.Cphys;
  const T* w = (const T*)a.w + (long)g * a.wstride;
  const T zero = from_f<T>(0.f);

  int abase[CV2_MT];
#pragma unroll
  for (int m = 0; m < CV2_MT; ++m) {
    const int q = wave * (32 * CV2_MT) + m * 32 + r;
    const int qc = q < npx ? q : npx - 1;
    const int ty = qc / TW, tx = qc - ty * TW;
    abase[m] = (ty * HWp + tx) * PSE + 8 * h;
  }
  f32x16 acc[CV2_MT][NT];
#pragma unroll
  for (int m = 0; m < CV2_MT; ++m)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[m][b] = (f32x16)(0.f);

  const int spc = (kh + tg - 1) / tg;                        // stages per channel chunk
  const int nstages = ((a.Ipad + KC - 1) / KC) * spc;
  // ---- per-thread staging plan, computed ONCE per tile (the per-stage index arithmetic used to cost ~470 VALU instructions
  // per 72 MFMAs; plain VALU issues at 4 cycles per wave64 instruction, so it -- not the MFMA pipe -- set the pace)
  uint4 rw[NWR], rh[NHR];
  int woff[NWR], wlds[NWR];                                  // weight chunk: global offset (stage 0) / LDS offset; woff < 0 = unused
  int hoff[NHR], hlds[NHR], hcc[NHR];                        // halo chunk: global offset (c0 = 0) / LDS offset / channel in chunk
  const int rows_full = tg * kw * 32 * NT;
#pragma unroll
  for (int k = 0; k < NWR; ++k) {
    const int e = tid + k * 256;
    const int cc = (e % NCH) * VW, row = e / NCH;
    const int co = row % (32 * NT), tapl = row / (32 * NT);  // tap within the stage's group of kernel rows
    woff[k] = (row < rows_full && nbase + co < a.Cout) ? (tapl * a.Cout + nbase + co) * a.Ipad + cc : -1;
    wlds[k] = row * PSE + cc;
    if (woff[k] >= 0) woff[k] |= (tapl / kw) << 26;          // local kernel row in the top bits (a stage may hold fewer rows)
  }
#pragma unroll
  for (int k = 0; k < NHR; ++k) {
    const int e = tid + k * 256;
    const int px = e / NCH, cc = (e - px * NCH) * VW;
    const int hy = (int)(((unsigned)px * (unsigned)magic_hw) >> 20), hx = px - hy * HWp;
    const int iy = ty0 + hy - pt, ix = tx0 + hx - pl;
    const bool inb = e < nhal && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    hoff[k] = inb ? (iy * a.W + ix) * a.Cphys + cc : -1;
    hlds[k] = (e < nhal) ? px * PSE + cc : -1;
    hcc[k] = cc;
  }
  const int wtap_stride = kw * a.Cout * a.Ipad;              // one kernel row of the [tap][Cout][Ipad] image
  auto prefetch = [&](int st) {
    const int c0 = (st / spc) * KC, ky0 = (st % spc) * tg;
    const int tgr = min(tg, kh - ky0);
    const T* wst = w + (long)ky0 * wtap_stride + c0;
#pragma unroll
    for (int k = 0; k < NWR; ++k) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (woff[k] >= 0 && (woff[k] >> 26) < tgr && c0 + (wlds[k] % PSE) < a.Ipad)
        v = *reinterpret_cast<const uint4*>(wst + (woff[k] & 0x3FFFFFF));
      rw[k] = v;
    }
    if (ky0 == 0) {
#pragma unroll
      for (int k = 0; k < NHR; ++k) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (hoff[k] >= 0) {
          const int ci = c0 + hcc[k];
          const T* src = x + hoff[k] + c0;
          if (ci + VW <= a.Cphys) v = *reinterpret_cast<const uint4*>(src);
          else {
            T tmp[VW];
#pragma unroll
            for (int j = 0; j < VW; ++j) {
              T e2 = zero;
              if (ci + j < a.Cphys) e2 = src[j];
              else if (a.ones && ci + j == a.Cphys) e2 = from_f<T>(1.f);
              tmp[j] = e2;
            }
            v = *reinterpret_cast<const uint4*>(tmp);
          }
        }
        rh[k] = v;
      }
    }
  };

  prefetch(0);
  for (int st = 0; st < nstages; ++st) {
    const int ky0 = (st % spc) * tg;
    const int tgr = min(tg, kh - ky0);
    __syncthreads();                                         // readers of the previous stage are done
#pragma unroll
    for (int k = 0; k < NWR; ++k)
      if (woff[k] >= 0 && (woff[k] >> 26) < tgr) *reinterpret_cast<uint4*>(sB + wlds[k]) = rw[k];
    if (ky0 == 0) {
#pragma unroll
      for (int k = 0; k < NHR; ++k)
        if (hlds[k] >= 0) *reinterpret_cast<uint4*>(sA + hlds[k]) = rh[k];
    }
    __syncthreads();
    if (st + 1 < nstages) prefetch(st + 1);
    // strength-reduced operand addresses: one add per tap, immediate offsets for the k-steps / N-tiles
    const T* arow = sA + ky0 * HWp * PSE;
    const T* bt = sB + r * PSE + 8 * h;
    for (int kyl = 0; kyl < tgr; ++kyl) {
      const T* ap = arow;
      for (int kx = 0; kx < kw; ++kx) {
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) {
          Frag8<T> fa[CV2_MT], fb[NT];
#pragma unroll
          for (int m = 0; m < CV2_MT; ++m) load8(fa[m], ap + abase[m] + 16 * s2);
#pragma unroll
          for (int b = 0; b < NT; ++b) load8(fb[b], bt + b * 32 * PSE + 16 * s2);
#pragma unroll
          for (int m = 0; m < CV2_MT; ++m)
#pragma unroll
            for (int b = 0; b < NT; ++b) mma32(acc[m][b], fa[m], fb[b]);
        }
        ap += PSE;
        bt += 32 * NT * PSE;
      }
      arow += HWp * PSE;
    }
  }
  // ---- epilogue.  Tiles are either whole image rows (TW == Wo) or a slice of one row (TH == 1), so the tile-local pixel
  // q maps to the linear image position ty0*Wo + tx0 + q: no per-element division (the v2 epilogue spent ~25 VALU/element).
  const long HWo = (long)a.Ho * a.Wo;
  const long lin0 = (long)ty0 * a.Wo + tx0;
  T* y = (T*)a.y + (long)n * HWo * a.Cstore;
  const T* res = a.res ? (const T*)a.res + (long)n * HWo * a.Cstore : nullptr;
  const int qlim = (TH > 1) ? npx : min(npx, a.Wo - tx0);
#pragma unroll
  for (int m = 0; m < CV2_MT; ++m) {
    const int qb = wave * (32 * CV2_MT) + m * 32 + 4 * h;
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      const int co = nbase + 32 * b + r;
      if (co < a.Cstore) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int q = qb + (reg & 3) + 8 * (reg >> 2);
          const long lin = lin0 + q;
          if (q < qlim && lin < HWo) {
            const long idx = lin * a.Cstore + co;
            float v = a.alpha * acc[m][b][reg];
            if (res) v += a.beta * to_f(res[idx]);
            y[idx] = from_f<T>(v);
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------ conv forward / dgrad v5: lean staging + vector epilogue
// Same tile and stage walk as v3.  What changed, and why (PMC on 64->64 3x3 bf16: v3 executed ~15 VALU instructions per MFMA,
// a plain VALU instruction issues at 4 cycles per wave, so the vector pipe -- not the matrix pipe -- set the pace):
//   * MFMA operand roles are swapped (A = weights, B = activations): an accumulator lane then owns ONE pixel and four
//     consecutive channels per register quad, so the epilogue is one 8-byte (bf16) / 16-byte (f32) store per quad instead
//     of sixteen 2-byte stores with a bounds test and a 64-bit address each;
//   * weight staging needs no per-thread plan: with 32*NT rows per tap the (tap, channel-row, piece) of a thread's k-th
//     16-byte piece are compile-time functions of k; rows past Cout / taps past the stage are clamped to valid addresses
//     (their products are never stored) instead of branching;
//   * layers with a channel tail or the implicit ones channel stay on v3 (Cphys % (16/sizeof T) == 0 and !ones here), which
//     removes the scalar tail path from the hot kernel.
template <typename T> struct Quad;
template <> struct Quad<bf16> { typedef bf16x4 type; };
template <> struct Quad<float> { typedef f32x4 type; };

template <typename T, int NT, bool LEPI, int NHR>
__global__ __launch_bounds__(256) void conv_fwd5_kernel(ConvArgs a, int TH, int TW, int tiles_x, int halo_cap, int tg_flags, int ntiles) {
  const int tg = tg_flags & 0xFFFF;
  const bool tile2d = (tg_flags >> 16) & 1;                  // 8 x 32 block tile of a wider image (TW == 32)
  constexpr int ESZ = sizeof(T), VW = 16 / ESZ, KC = 64 / ESZ, PSE = KC + VW, KS = KC / 16;
  constexpr int NB = 32 * NT, LNB = NT == 1 ? 5 : (NT == 2 ? 6 : 7);
  constexpr int NWR = 9;                                    // 16-B weight pieces per thread held in registers (halo: NHR, 7 or 9)
  typedef typename Quad<T>::type QT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* sA = reinterpret_cast<T*>(smem_raw);                    // [halo px][PSE]
  T* sB = sA + (long)halo_cap * PSE;                         // [tg*kw][NB][PSE]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  // 1-D grid, XCD-aware: workgroup ids go round-robin over the 8 XCDs (each with its own L2), so the (image, tile) pairs are
  // dealt to XCDs by pair % 8 and the channel blocks of one pair run back to back on the SAME XCD (ids L and L + 8): the second
  // block finds the input tile the first one fetched in that L2 instead of reading it from HBM again.
  const int nblk = (tg_flags >> 17) & 0x3FFF;
  const int slot = blockIdx.x >> 3, pair = (slot / nblk) * 8 + (blockIdx.x & 7);
  const int n = pair / ntiles, tile = pair - n * ntiles;
  if (n >= a.N) return;
  const int g = find_group(a.seg, a.ngroups, n);
  if (g < 0) return;
  const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
  const int nbase = (slot % nblk) * NB;
  const int kh = a.kh[g], kw = a.kw[g], pt = a.pt[g], pl = a.pl[g];
  const int HWp = TW + kw - 1, HHp = TH + kh - 1;
  const int npx = TH * TW;
  const int nhal = HHp * HWp * 4;
  const int magic_hw = (1 << 20) / HWp + 1;
  const T* x = (const T*)a.x + (long)n * a.H * a.W * a.Cphys;
  const T* w = (const T*)a.w + (long)g * a.wstride;

  int abase[CV2_MT];
#pragma unroll
  for (int m = 0; m < CV2_MT; ++m) {
    const int q = wave * (32 * CV2_MT) + m * 32 + r;
    const int qc = q < npx ? q : npx - 1;
    const int ty = qc / TW, tx = qc - ty * TW;
    abase[m] = (ty * HWp + tx) * PSE + 8 * h;
  }
  f32x16 acc[CV2_MT][NT];
#pragma unroll
  for (int m = 0; m < CV2_MT; ++m)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[m][b] = (f32x16)(0.f);

  const int spc = (kh + tg - 1) / tg;                        // stages per channel chunk
  const int nstages = ((a.Ipad + KC - 1) / KC) * spc;
  const int ntaps = kh * kw;
  const int wrows = tg * kw * NB;                            // LDS weight rows of a full stage
  // halo plan (per tile): global offset at channel 0 / LDS offset of this thread's pieces
  uint4 rw[NWR], rh[NHR];
  int hoff[NHR], hlds[NHR];
#pragma unroll
  for (int k = 0; k < NHR; ++k) {
    const int e = tid + k * 256;
    const int px = e >> 2, cc = (e & 3) * VW;
    const int hy = (int)(((unsigned)px * (unsigned)magic_hw) >> 20), hx = px - hy * HWp;
    const int iy = ty0 + hy - pt, ix = tx0 + hx - pl;
    const bool inb = e < nhal && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    hoff[k] = inb ? (iy * a.W + ix) * a.Cphys + cc : -1;
    hlds[k] = (e < nhal) ? px * PSE + cc : -1;
  }
  // weight pieces: row = (tid >> 2) + 64 k, piece = tid & 3  ->  co = row & (NB-1), local tap = row >> LNB
  const int wcc = (tid & 3) * VW;
  const int wrow0 = tid >> 2;
  const long wtap = (long)a.Cout * a.Ipad;
  auto prefetch = [&](int st) {
    const int c0 = (st / spc) * KC, ky0 = (st % spc) * tg;
    const int chw = min(c0 + wcc, a.Ipad - VW);              // a piece past the padded row reads valid bytes (unused k-steps)
    const int tap0 = ky0 * kw;
#pragma unroll
    for (int k = 0; k < NWR; ++k) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (k * 64 < wrows) {                                  // uniform
        const int row = wrow0 + 64 * k;
        const int co = min(nbase + (row & (NB - 1)), a.Cout - 1);
        const int tap = min(tap0 + (row >> LNB), ntaps - 1);
        v = *reinterpret_cast<const uint4*>(w + tap * wtap + (long)co * a.Ipad + chw);
      }
      rw[k] = v;
    }
    if (ky0 == 0) {
#pragma unroll
      for (int k = 0; k < NHR; ++k) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (hoff[k] >= 0 && c0 + (((tid + k * 256) & 3) * VW) < a.Cphys) v = *reinterpret_cast<const uint4*>(x + hoff[k] + c0);
        rh[k] = v;
      }
    }
  };

  prefetch(0);
  for (int st = 0; st < nstages; ++st) {
    const int c0 = (st / spc) * KC, ky0 = (st % spc) * tg;
    const int tgr = min(tg, kh - ky0);
    const int ksteps = min(KS, (a.Ipad - c0) >> 4);
    __syncthreads();                                         // readers of the previous stage are done
#pragma unroll
    for (int k = 0; k < NWR; ++k)
      if (wrow0 + 64 * k < wrows) *reinterpret_cast<uint4*>(sB + (wrow0 + 64 * k) * PSE + wcc) = rw[k];
    if (ky0 == 0) {
#pragma unroll
      for (int k = 0; k < NHR; ++k)
        if (hlds[k] >= 0) *reinterpret_cast<uint4*>(sA + hlds[k]) = rh[k];
    }
    __syncthreads();
    if (st + 1 < nstages) prefetch(st + 1);
    const T* arow = sA + ky0 * HWp * PSE;
    const T* bt = sB + r * PSE + 8 * h;
    for (int kyl = 0; kyl < tgr; ++kyl) {
      const T* ap = arow;
      for (int kx = 0; kx < kw; ++kx) {
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) {
          if (s2 >= ksteps) break;
          Frag8<T> fx[CV2_MT], fw[NT];
#pragma unroll
          for (int m = 0; m < CV2_MT; ++m) load8(fx[m], ap + abase[m] + 16 * s2);
#pragma unroll
          for (int b = 0; b < NT; ++b) load8(fw[b], bt + b * 32 * PSE + 16 * s2);
#pragma unroll
          for (int m = 0; m < CV2_MT; ++m)
#pragma unroll
            for (int b = 0; b < NT; ++b) mma32(acc[m][b], fw[b], fx[m]);       // rows = channels, columns = pixels
        }
        ap += PSE;
        bt += NB * PSE;
      }
      arow += HWp * PSE;
    }
  }
  // ---- epilogue: lane = pixel (32m + r of the wave's 64), register quad i = channels 32b + 8i + 4h .. +3
  const long HWo = (long)a.Ho * a.Wo;
  const long lin0 = (long)ty0 * a.Wo + tx0;
  T* y = (T*)a.y + (long)n * HWo * a.Cstore;
  const T* res = a.res ? (const T*)a.res + (long)n * HWo * a.Cstore : nullptr;
  const int qlim = (TH > 1) ? npx : min(npx, a.Wo - tx0);
  // image position of tile pixel q: a linear run (whole rows, or a slice of one row) unless the tile is an 8 x 32 block
  auto pix_lin = [&](int q, bool& ok) -> long {
    if (!tile2d) { const long lin = lin0 + q; ok = q < qlim && lin < HWo; return lin; }
    const int ty = q >> 5, tx = q & 31;
    ok = q < npx && ty0 + ty < a.Ho && tx0 + tx < a.Wo;
    return lin0 + (long)ty * a.Wo + tx;
  };
  if constexpr (LEPI) {
    // Through LDS: a quad store straight from the accumulator layout hands the memory system 32 separate 8/16-byte
    // pieces per instruction (one per pixel row).  Each wave instead parks its 64 x NB tile in its own LDS slab
    // [64 px][NB + pad] and streams it out as 16 B per lane, consecutive lanes covering whole pixel rows.
    constexpr int ROW = NB + VW;                             // elements; +16 B keeps the rows off each other's banks
    constexpr int PPR = NB / VW;                             // 16-B pieces per pixel row
    constexpr int PXI = 64 / PPR;                            // pixel rows per store instruction
    __syncthreads();                                         // every wave is done reading the operand tiles
    T* sE = reinterpret_cast<T*>(smem_raw) + wave * 32 * ROW;   // one 32-pixel slab per wave, reused for each m-fragment
    const int piece = lane % PPR, prow = lane / PPR;
    const int cch = nbase + piece * VW;
#pragma unroll
    for (int m = 0; m < CV2_MT; ++m) {
      const int q = wave * (32 * CV2_MT) + m * 32 + r;
      bool ok;
      const long lin = pix_lin(q, ok);
      const long pix = lin * a.Cstore + nbase + 4 * h;
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int cl = 32 * b + 8 * i;
          float v0 = a.alpha * acc[m][b][4 * i], v1 = a.alpha * acc[m][b][4 * i + 1];
          float v2 = a.alpha * acc[m][b][4 * i + 2], v3 = a.alpha * acc[m][b][4 * i + 3];
          if (res && ok && nbase + cl + 4 * h < a.Cstore) {
            const QT rv = *reinterpret_cast<const QT*>(res + pix + cl);
            v0 += a.beta * (float)rv[0]; v1 += a.beta * (float)rv[1]; v2 += a.beta * (float)rv[2]; v3 += a.beta * (float)rv[3];
          }
          QT o;
          o[0] = from_f<T>(v0); o[1] = from_f<T>(v1); o[2] = from_f<T>(v2); o[3] = from_f<T>(v3);
          *reinterpret_cast<QT*>(sE + r * ROW + cl + 4 * h) = o;
        }
      // the slab is private to the wave: LDS operations of one wave complete in order, no workgroup barrier needed
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (also keeps the compiler from hoisting the reads over the writes)
#pragma unroll
      for (int j = 0; j < 32 / PXI; ++j) {
        const int pl = j * PXI + prow;                       // pixel within the 32 of this m-fragment
        const int q2 = wave * (32 * CV2_MT) + m * 32 + pl;
        bool ok2;
        const long lin2 = pix_lin(q2, ok2);
        if (ok2 && cch < a.Cstore) {
          const uint4 v = *reinterpret_cast<const uint4*>(sE + pl * ROW + piece * VW);
          *reinterpret_cast<uint4*>(y + lin2 * a.Cstore + cch) = v;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // reads done before the next fragment overwrites the slab
    }
  } else {
#pragma unroll
    for (int m = 0; m < CV2_MT; ++m) {
      const int q = wave * (32 * CV2_MT) + m * 32 + r;
      bool ok;
      const long lin = pix_lin(q, ok);
      if (ok) {
        const long pix = lin * a.Cstore + nbase + 4 * h;
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int cl = 32 * b + 8 * i;                   // + 4h + nbase = first channel of the quad
            if (nbase + cl + 4 * h < a.Cstore) {
              float v0 = a.alpha * acc[m][b][4 * i], v1 = a.alpha * acc[m][b][4 * i + 1];
              float v2 = a.alpha * acc[m][b][4 * i + 2], v3 = a.alpha * acc[m][b][4 * i + 3];
              if (res) {
                const QT rv = *reinterpret_cast<const QT*>(res + pix + cl);
                v0 += a.beta * (float)rv[0]; v1 += a.beta * (float)rv[1]; v2 += a.beta * (float)rv[2]; v3 += a.beta * (float)rv[3];
              }
              QT o;
              o[0] = from_f<T>(v0); o[1] = from_f<T>(v1); o[2] = from_f<T>(v2); o[3] = from_f<T>(v3);
              *reinterpret_cast<QT*>(y + pix + cl) = o;
            }
          }
      }
    }
  }
}

// ------------------------------------------------------------------ wgrad
struct WgradArgs {
  const void* x;       // [N][H][W][Cphys]
  const void* dy;      // [N][Ho][Wo][Cout]
  float* G[HDMOE_MAX_GROUPS];   // [tap][Cout][Cin] fp32, pre-zeroed
  const int* seg;
  int N, H, W, Ho, Wo, Cin, Cphys, Cout, stride, ones, ngroups, spw, ob_count, ib_count, tap_lo;   // tap_lo: first tap of this pass
  int kh[HDMOE_MAX_GROUPS], kw[HDMOE_MAX_GROUPS], pt[HDMOE_MAX_GROUPS], pl[HDMOE_MAX_GROUPS];
};

template <typename T> struct WgTraits;
template <> struct WgTraits<float> { static constexpr int CPL = 1; };
template <> struct WgTraits<bf16> { static constexpr int CPL = 2; };   // a lane owns a channel pair (one dword)

template <typename T, int CPL>
DEVI void flush_wgrad(f32x16 (&acc)[CPL][CPL], float* G, int tap, int o0, int i0, int Cout, int Cin, int lane) {
  const int col = lane & 31;
#pragma unroll
  for (int qa = 0; qa < CPL; ++qa)
#pragma unroll
    for (int qb = 0; qb < CPL; ++qb) {
      const int ci = i0 + CPL * col + qb;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int o = o0 + CPL * acc_row(reg, lane) + qa;
        if (o < Cout && ci < Cin) atomicAdd(&G[((long)tap * Cout + o) * Cin + ci], acc[qa][qb][reg]);
      }
      acc[qa][qb] = (f32x16)(0.f);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int CPL = WgTraits<T>::CPL;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  int t = blockIdx.x;
  const int ib = t % a.ib_count; t /= a.ib_count;
  const int ob = t % a.ob_count; t /= a.ob_count;
  const int tap = t;
  const int o0 = ob * 32 * CPL, i0 = ib * 32 * CPL;
  const int n_begin = (blockIdx.y * 4 + wave) * a.spw;
  const int HWo = a.Ho * a.Wo;
  const T* X = (const T*)a.x;
  const T* DY = (const T*)a.dy;

  f32x16 acc[CPL][CPL];
#pragma unroll
  for (int qa = 0; qa < CPL; ++qa)
#pragma unroll
    for (int qb = 0; qb < CPL; ++qb) acc[qa][qb] = (f32x16)(0.f);

  int cur_g = -1;
  for (int s = 0; s < a.spw; ++s) {
    const int n = n_begin + s;
    if (n >= a.N) break;
    const int g = find_group(a.seg, a.ngroups, n);
    if (g != cur_g) {
      if (cur_g >= 0 && tap < a.kh[cur_g] * a.kw[cur_g])
        flush_wgrad<T, CPL>(acc, a.G[cur_g], tap, o0, i0, a.Cout, a.Cin, lane);
      cur_g = g;
    }
    if (g < 0 || tap >= a.kh[g] * a.kw[g]) continue;
    const int ky = tap / a.kw[g] - a.pt[g], kx = tap % a.kw[g] - a.pl[g];
    const T* dy = DY + (long)n * HWo * a.Cout;
    const T* x = X + (long)n * a.H * a.W * a.Cphys;
    for (int p0 = 0; p0 < HWo; p0 += 16) {
      Frag8<T> fa[CPL], fb[CPL];
#pragma unroll
      for (int q = 0; q < CPL; ++q) { fa[q].zero(); fb[q].zero(); }
      int p = p0 + 8 * h;
      int oy = p / a.Wo, ox = p % a.Wo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (p < HWo) {
          const int iy = oy * a.stride + ky, ix = ox * a.stride + kx;
          const bool inb = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
          const T* dyp = dy + (long)p * a.Cout;
          const T* xp = x + ((long)iy * a.W + ix) * a.Cphys;
#pragma unroll
          for (int q = 0; q < CPL; ++q) {
            const int o = o0 + CPL * r + q;
            if (o < a.Cout) fa[q].set(j, to_f(dyp[o]));
            const int ci = i0 + CPL * r + q;
            if (inb) {
              if (ci < a.Cphys) fb[q].set(j, to_f(xp[ci]));
              else if (a.ones && ci == a.Cphys) fb[q].set(j, 1.f);
            }
          }
        }
        ++p; ++ox;
        if (ox == a.Wo) { ox = 0; ++oy; }
      }
#pragma unroll
      for (int qa = 0; qa < CPL; ++qa)
#pragma unroll
        for (int qb = 0; qb < CPL; ++qb) mma32(acc[qa][qb], fa[qa], fb[qb]);
    }
  }
  if (cur_g >= 0 && tap < a.kh[cur_g] * a.kw[cur_g])
    flush_wgrad<T, CPL>(acc, a.G[cur_g], tap, o0, i0, a.Cout, a.Cin, lane);
}


// ------------------------------------------------------------------ wgrad v2 (stride 1): LDS-staged tiles
// A workgroup walks `upw` consecutive (sample, pixel-tile) units.  Per unit it stages the dy tile [128 px][OB] and the
// x halo tile [(TH+kh-1)][(TW+kw-1)][32 ch] in LDS once (16-byte vectors); every tap then reads both from LDS.  Each of
// the 4 waves owns the taps t = wave (mod 4) (k-steps instead, for 1x1 / linear layers) and keeps their [OB x 32]
// accumulators in registers across all its units; one coalesced fp32 atomic flush per expert change / at the end.
// The contraction runs over PIXELS, so both MFMA operands need 8 consecutive pixels of one channel per lane while the
// tiles are stored [pixel][channel]: bf16 uses the hardware transposing read ds_read_b64_tr_b16 (two per fragment,
// cdna_hip_programming.md T10), fp32 reads its single value per 32x32x2 MFMA with conflict-free ds_read_b32.
constexpr int WG2_PT = 128;     // pixels per tile (bf16)
// f32 holds twice the bytes per staged pixel in registers (next unit's prefetch) and per accumulator operand: 64-pixel tiles
// keep the f32 instantiations under 256 VGPRs without spills (spill reloads in the k-loop wait on the prefetch loads).
template <typename T> struct Wg2PT { static constexpr int N = sizeof(T) == 4 ? 64 : 128; };
constexpr int WG2_IB = 32;      // input channels per workgroup

template <typename T> DEVI void frag_set_raw(Frag8<T>& f, int j, T v);
template <> DEVI void frag_set_raw<bf16>(Frag8<bf16>& f, int j, bf16 v) { f.v[j] = v; }
template <> DEVI void frag_set_raw<float>(Frag8<float>& f, int j, float v) { f.v[j] = v; }

typedef __attribute__((ext_vector_type(4))) short s16x4;
template <typename T> struct VecW;                         // elements per 16-byte vector
template <> struct VecW<bf16> { static constexpr int N = 8; };
template <> struct VecW<float> { static constexpr int N = 4; };

// bf16: fragment rows = pixels rowa (elements 0..3) and rowb (4..7) of this lane's 16-lane group; `col` = 16-bit column
DEVI void load_frag_tr(Frag8<bf16>& f, const bf16* rowa, const bf16* rowb, int col) {
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(rowa + col));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(rowb + col));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  f.v = __builtin_bit_cast(bf16x8, v);
}

struct Wg2Geom {                      // launch geometry of one kernel-size class
  int TH, TW, tw_shift, tiles_y, tiles_x, upw, chunks, ngr;
  int groups[HDMOE_MAX_GROUPS];       // the groups (experts) of this class
};

template <typename T, int OT, int MAXT, bool VEC>
__global__ __launch_bounds__(256, (MAXT <= 3 ? 2 : 1)) void conv_wgrad2_kernel(WgradArgs a, Wg2Geom gm) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int OB = 32 * OT;
  constexpr int OBP = (OT == 2 && sizeof(T) == 2) ? OB + 32 : OB;      // bank-conflict-free row stride for the tr reads
  constexpr int VW = VecW<T>::N;
  constexpr int CPP = OB / VW;                                          // 16-B chunks per dy pixel
  constexpr int XPP = WG2_IB / VW;                                      // 16-B chunks per x pixel
  constexpr int PT = Wg2PT<T>::N;
  T* sdy = reinterpret_cast<T*>(smem_raw);                              // [PT][OBP]
  T* sx = sdy + PT * OBP;                                           // [halo rows][halo cols][WG2_IB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int q4 = (lane & 15) >> 2, col4 = (lane & 16) + 4 * (lane & 3);  // tr-read row within the 4-row block / column
  const int i0 = blockIdx.x * WG2_IB;
  const int o0 = blockIdx.y * OB;
  const int TH = gm.TH, TW = gm.TW, tw_shift = gm.tw_shift;
  // ---- this workgroup: one chunk of the units of ONE group (expert); the group's row range comes from the device-side plan
  const int g = gm.groups[blockIdx.z / gm.chunks];
  const int chunk = blockIdx.z % gm.chunks;
  const int tpn = gm.tiles_y * gm.tiles_x;
  const int row0 = a.seg ? a.seg[g] : 0, row1 = a.seg ? a.seg[g + 1] : a.N;
  const int u0 = row0 * tpn + chunk * gm.upw;
  const int u1 = min(row1 * tpn, u0 + gm.upw);
  if (u0 >= u1) return;
  const int kh = a.kh[g], kw = a.kw[g], pt = a.pt[g], pl = a.pl[g];
  const int ntaps = kh * kw;
  const bool split_k = ntaps < 4;
  const int HWp = TW + kw - 1, HHp = TH + kh - 1;
  const int nh = HHp * HWp * XPP;
  const int magic_hw = (1 << 20) / HWp + 1, magic_tpn = (1 << 20) / tpn + 1, magic_tx = (1 << 20) / gm.tiles_x + 1;
  const T* X = (const T*)a.x;
  const T* DY = (const T*)a.dy;
  const T zero = from_f<T>(0.f);
  // integer division is ~35 VALU instructions on CDNA: tile widths are powers of two in every shipped config (shift path),
  // the other divisors get a 20-bit reciprocal (exact for the small indices used here)
  auto div_tw = [&](int q) { return tw_shift >= 0 ? (q >> tw_shift) : q / TW; };
  auto fast_div = [](int q, int magic) { return (int)(((unsigned)q * (unsigned)magic) >> 20); };

  for (int e = tid; e < (PT - TH * TW) * OBP; e += 256) sdy[TH * TW * OBP + e] = zero;   // rows no tile ever writes

  // Tap ownership.  Dealing 9 taps round-robin gives the waves 3/2/2/2 and three of them idle a third of every k-loop at the
  // barrier.  For 3x3 layers wave w instead owns taps 2w and 2w+1 outright and every 4th k-step (ks % 4 == w) of tap 8:
  // 9 tap-k-steps per 4 k-steps for every wave, still three accumulator sets per wave (tap 8 is flushed as four partials).
  const bool bal9 = ntaps == 9 && MAXT == 3 && sizeof(T) == 4;   // (bf16: measured slower, its k-loop is not MFMA-bound)
  // a.tap_lo: classes with more taps than 4 waves x MAXT accumulators run as several passes over tap ranges
  auto tap_of = [&](int t) { return split_k ? t : (bal9 ? (t < 2 ? 2 * wave + t : 8) : a.tap_lo + wave + 4 * t); };
  int toff[MAXT];                                             // LDS offset of this wave's taps inside the halo tile
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    const int tap = tap_of(t);
    const int ky = tap / kw, kx = tap - ky * kw;
    toff[t] = (ky * HWp + kx) * WG2_IB;
  }
  f32x16 acc[MAXT][OT];
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
#pragma unroll
    for (int q = 0; q < OT; ++q) acc[t][q] = (f32x16)(0.f);

  constexpr int NDY = (PT * CPP + 255) / 256;
  constexpr int NX = sizeof(T) == 2 ? 6 : 8;                  // halo chunks per thread held in registers (host checks the cap)
  uint4 rdy[VEC ? NDY : 1], rx[VEC ? NX : 1];
  // unit -> (sample, tile origin); large unit indices (flattened linear layers) take the exact division
  // (q * magic) >> 20 with magic = 2^20/d + 1 overshoots floor(q/d) by at most one for q < 2^20 and is exact only while
  // q * d < 2^20 -- not guaranteed here (a flattened 262144-pixel row has d = tpn = 2048 tiles): fix the quotient up.
  auto unit_origin = [&](int u, int& n, int& ty0, int& tx0) {
    if (u < (1 << 12)) { n = fast_div(u, magic_tpn); if (n * tpn > u) --n; } else n = u / tpn;
    const int tyx = u - n * tpn;
    int tyi;
    if (tyx < (1 << 12)) { tyi = fast_div(tyx, magic_tx); if (tyi * gm.tiles_x > tyx) --tyi; } else tyi = tyx / gm.tiles_x;
    ty0 = tyi * TH; tx0 = (tyx - tyi * gm.tiles_x) * TW;
  };
  // Branch-free staging (VEC path: Cphys and Cout are multiples of the 16-byte vector): every piece is loaded from a CLAMPED,
  // always-valid address and per-thread bit masks remember which pieces are real; the masks are applied when the registers
  // are published to LDS.  (Predicated loads cost ~100 spilled SGPRs of saved exec masks here, and spill reloads -- vector
  // memory operations themselves -- put an s_waitcnt vmcnt(0) in front of the loads that follow them.)
  unsigned dymask = 0, xmask = 0, xinb = 0;
  auto prefetch = [&](int u) {                                 // global -> registers (VEC path)
    int n, ty0, tx0;
    unit_origin(u, n, ty0, tx0);
    const int rows_valid = min(TH, a.Ho - ty0);
    const int cols_valid = min(TW, a.Wo - tx0);
    const T* dyn = DY + (((long)n * a.Ho + ty0) * a.Wo + tx0) * a.Cout + o0;
    const int cmax = a.Cout - o0 - VW;                         // last whole vector of this o-block that exists
    dymask = 0;
#pragma unroll
    for (int k = 0; k < NDY; ++k) {
      const int idx = tid + k * 256;
      const int pq = idx / CPP, c = (idx - pq * CPP) * VW;
      const int ty = div_tw(pq), tx = pq - ty * TW;
      const bool ok = pq < TH * TW && ty < rows_valid && tx < cols_valid && c <= cmax;
      const int tyc = min(ty, rows_valid - 1), txc = min(tx, cols_valid - 1), cc = max(min(c, cmax), -o0);
      rdy[k] = *reinterpret_cast<const uint4*>(dyn + ((long)tyc * a.Wo + txc) * a.Cout + cc);
      dymask |= (ok ? 1u : 0u) << k;
    }
    const T* xn = X + (long)n * a.H * a.W * a.Cphys;
    xmask = 0; xinb = 0;
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int idx = min(tid + k * 256, nh - 1);
      const int px = idx / XPP, c = (idx - px * XPP) * VW;
      const int hy = fast_div(px, magic_hw), hx = px - hy * HWp;
      const int iy = ty0 + hy - pt, ix = tx0 + hx - pl, ci = i0 + c;
      const bool inb = tid + k * 256 < nh && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const bool ok = inb && ci < a.Cphys;
      const int iyc = min(max(iy, 0), a.H - 1), ixc = min(max(ix, 0), a.W - 1), cic = min(ci, a.Cphys - VW);
      rx[k] = *reinterpret_cast<const uint4*>(xn + ((long)iyc * a.W + ixc) * a.Cphys + cic);
      xmask |= (ok ? 1u : 0u) << k;
      xinb |= (inb ? 1u : 0u) << k;
    }
  };
  // value of an x piece that was not loaded: zero (padding pixels, channels past Cin), or the implicit ones channel
  // (channel Cphys of an in-bounds pixel) in its first element
  auto const_piece = [&](int ci, bool inb) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (a.ones && inb && ci == a.Cphys) {
      const T one = from_f<T>(1.f);
      if constexpr (sizeof(T) == 2) v.x = (unsigned)__builtin_bit_cast(unsigned short, one);
      else v.x = __builtin_bit_cast(unsigned, one);
    }
    return v;
  };

  if (VEC) prefetch(u0);
  for (int u = u0; u < u1; ++u) {
    int n, ty0, tx0;
    unit_origin(u, n, ty0, tx0);
    const int pv = min(TH, a.Ho - ty0) * TW;                 // tile-local pixels q = ty*TW + tx below pv lie in image rows
    __syncthreads();                                         // previous unit's readers are done
    if (VEC) {
#pragma unroll
      for (int k = 0; k < NDY; ++k) {
        const int idx = tid + k * 256;
        const int pq = idx / CPP, c = (idx - pq * CPP) * VW;
        const uint4 v = (dymask >> k) & 1u ? rdy[k] : make_uint4(0, 0, 0, 0);
        if (pq < TH * TW) *reinterpret_cast<uint4*>(sdy + pq * OBP + c) = v;
      }
#pragma unroll
      for (int k = 0; k < NX; ++k) {
        const int idx = tid + k * 256;
        if (idx < nh) {
          const int c = (idx % XPP) * VW;
          const uint4 v = (xmask >> k) & 1u ? rx[k] : const_piece(i0 + c, (xinb >> k) & 1u);
          *reinterpret_cast<uint4*>(sx + (idx / XPP) * WG2_IB + c) = v;
        }
      }
    } else {
      const int rows_valid = min(TH, a.Ho - ty0);
      const T* dyn = DY + (((long)n * a.Ho + ty0) * a.Wo + tx0) * a.Cout + o0;
      for (int ty = 0; ty < TH; ++ty) {
        for (int e = tid; e < TW * OB; e += 256) {
          const int tx = e / OB, c = e - tx * OB;
          T v = zero;
          if (ty < rows_valid && tx0 + tx < a.Wo && o0 + c < a.Cout) v = dyn[((long)ty * a.Wo + tx) * a.Cout + c];
          sdy[(ty * TW + tx) * OBP + c] = v;
        }
      }
      const T* xn = X + (long)n * a.H * a.W * a.Cphys;
      for (int hy = 0; hy < HHp; ++hy) {
        const int iy = ty0 + hy - pt;
        const bool rowin = iy >= 0 && iy < a.H;
        for (int e = tid; e < HWp * WG2_IB; e += 256) {
          const int hx = e / WG2_IB, c = e - hx * WG2_IB;
          const int ix = tx0 + hx - pl, ci = i0 + c;
          T v = zero;
          if (rowin && ix >= 0 && ix < a.W) {
            if (ci < a.Cphys) v = xn[((long)iy * a.W + ix) * a.Cphys + ci];
            else if (a.ones && ci == a.Cphys) v = from_f<T>(1.f);
          }
          sx[(hy * HWp + hx) * WG2_IB + c] = v;
        }
      }
    }
    __syncthreads();
    if (VEC && u + 1 < u1) prefetch(u + 1);                  // next unit's loads fly while this unit computes
    const int nks = (pv + 15) >> 4;
    for (int ks = (split_k ? wave : 0); ks < nks; ks += (split_k ? 4 : 1)) {
      Frag8<T> fa[OT];
      if constexpr (sizeof(T) == 2) {
        // lane's two pixel rows of the 16-pixel k-step: q = 16 ks + 8 h + q4 (+4)
        const int qa = ks * 16 + 8 * h + q4, qb = qa + 4;
#pragma unroll
        for (int t = 0; t < OT; ++t) load_frag_tr(fa[t], sdy + qa * OBP + 32 * t, sdy + qb * OBP + 32 * t, col4);
        const int qac = min(qa, pv - 1), qbc = min(qb, pv - 1);           // dy is zero beyond pv; keep x reads in the halo
        const int tya = div_tw(qac), tyb = div_tw(qbc);
        const T* xa = sx + (tya * HWp + (qac - tya * TW)) * WG2_IB;
        const T* xb = sx + (tyb * HWp + (qbc - tyb * TW)) * WG2_IB;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
          const int tap = tap_of(t);
          if (tap < ntaps && !(bal9 && t == 2 && (ks & 3) != wave)) {
            Frag8<T> fb;
            load_frag_tr(fb, xa + toff[t], xb + toff[t], col4);
#pragma unroll
            for (int q = 0; q < OT; ++q) mma32(acc[t][q], fa[q], fb);
          }
        }
      } else {
        int qs[8];
        const int q0 = ks * 16 + 8 * h;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int q = q0 + j;
#pragma unroll
          for (int t = 0; t < OT; ++t) frag_set_raw<T>(fa[t], j, sdy[q * OBP + 32 * t + r]);
          const int qc = min(q, pv - 1);
          const int ty = div_tw(qc), tx = qc - ty * TW;
          qs[j] = (ty * HWp + tx) * WG2_IB + r;
        }
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
          const int tap = tap_of(t);
          if (tap < ntaps && !(bal9 && t == 2 && (ks & 3) != wave)) {
            Frag8<T> fb;
#pragma unroll
            for (int j = 0; j < 8; ++j) frag_set_raw<T>(fb, j, sx[qs[j] + toff[t]]);
#pragma unroll
            for (int q = 0; q < OT; ++q) mma32(acc[t][q], fa[q], fb);
          }
        }
      }
    }
  }
  // ---- one coalesced fp32 atomic flush: lanes = 32 consecutive input channels (128-byte runs)
  float* G = a.G[g];
  const int ci = i0 + r;
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    const int tap = tap_of(t);
    if (tap < ntaps) {
#pragma unroll
      for (int q = 0; q < OT; ++q) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int o = o0 + 32 * q + acc_row(reg, lane);
          if (o < a.Cout && ci < a.Cin) atomicAdd(&G[((long)tap * a.Cout + o) * a.Cin + ci], acc[t][q][reg]);
        }
      }
    }
  }
}

constexpr int ROWS_PER_LAUNCH = 65535;                      // gridDim.y limit of the row-per-blockIdx.y kernels

template <typename T, int NB>
void launch_conv(ConvArgs a, bool vec, hipStream_t st) {
  for (int n0 = 0; n0 < a.N; n0 += ROWS_PER_LAUNCH) {
    a.n0 = n0;
    dim3 grid(cdiv((long)a.Ho * a.Wo, 128), a.N - n0 < ROWS_PER_LAUNCH ? a.N - n0 : ROWS_PER_LAUNCH, cdiv(a.Cstore, 32 * NB));
    if (vec) hipLaunchKernelGGL((conv_fwd_kernel<T, NB, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_fwd_kernel<T, NB, false>), grid, dim3(256), 0, st, a);
  }
}

template <typename T>
void launch_conv_nb(const ConvArgs& a, bool vec, hipStream_t st) {
  if (a.Cstore <= 32) launch_conv<T, 1>(a, vec, st);
  else if (a.Cstore <= 64) launch_conv<T, 2>(a, vec, st);
  else launch_conv<T, 4>(a, vec, st);
}

}  // namespace

extern "C" {

int hdmoe_wprep_fwd(float* const* w_raw, const float* const* gain_ptr, float gain_val, const int* kh,
                    const int* kw, int ngroups, int O, int I, int Ipad, int Opad, void* wf, long wf_stride,
                    void* wd, long wd_stride, int normalize, int mutate, int flip, int dtype,
                    hipStream_t stream) {
  if (ngroups < 1 || ngroups > HDMOE_MAX_GROUPS || O < 1 || I < 1 || Ipad < I || Ipad % 16 || !wf) return HDMOE_EINVAL;
  if (wd && (Opad < O || Opad % 16)) return HDMOE_EINVAL;
  WprepArgs a;
  for (int g = 0; g < HDMOE_MAX_GROUPS; ++g) {
    const int s = g < ngroups ? g : 0;
    a.w_raw[g] = w_raw[s]; a.gain_ptr[g] = gain_ptr ? gain_ptr[s] : nullptr;
    a.kh[g] = kh[s]; a.kw[g] = kw[s];
  }
  a.wf = wf; a.wd = wd; a.wf_stride = wf_stride; a.wd_stride = wd_stride;
  a.O = O; a.I = I; a.Ipad = Ipad; a.Opad = wd ? Opad : O; a.gain_val = gain_val;
  a.normalize = normalize; a.mutate = mutate; a.flip = flip;
  a.wf_plane = a.wd_plane = 0;
  if (dtype == HDMOE_F32S) {                               // planes [hi | lo], each holding all groups
    if (Opad != O && wd) return HDMOE_EINVAL;              // (the split kernels take unpadded channel counts only)
    a.wf_plane = (long)ngroups * wf_stride; a.wd_plane = (long)ngroups * wd_stride;
  }
  dim3 grid(wd ? (Opad > O ? Opad : O) : O, ngroups);
  if (dtype == HDMOE_F32) hipLaunchKernelGGL(wprep_fwd_kernel<float>, grid, dim3(128), 0, stream, a);
  else if (dtype == HDMOE_BF16 || dtype == HDMOE_F32S) hipLaunchKernelGGL(wprep_fwd_kernel<bf16>, grid, dim3(128), 0, stream, a);
  else return HDMOE_EDTYPE;
  return hdmoe_launch_status();
}

int hdmoe_wprep_bwd(const float* const* w_raw, const float* const* gain_ptr, float gain_val,
                    const float* const* G, float* const* dw, float* const* dgain, const int* kh, const int* kw,
                    int ngroups, int O, int I, int normalize, hipStream_t stream) {
  if (ngroups < 1 || ngroups > HDMOE_MAX_GROUPS || O < 1 || I < 1) return HDMOE_EINVAL;
  WprepBwdArgs a;
  for (int g = 0; g < HDMOE_MAX_GROUPS; ++g) {
    const int s = g < ngroups ? g : 0;
    a.w_raw[g] = w_raw[s]; a.gain_ptr[g] = gain_ptr ? gain_ptr[s] : nullptr;
    a.G[g] = G[s]; a.dw[g] = dw[s]; a.dgain[g] = dgain ? dgain[s] : nullptr;
    a.kh[g] = kh[s]; a.kw[g] = kw[s];
  }
  a.O = O; a.I = I; a.gain_val = gain_val; a.normalize = normalize;
  hipLaunchKernelGGL(wprep_bwd_kernel, dim3(O, ngroups), dim3(128), 0, stream, a);
  return hdmoe_launch_status();
}

int hdmoe_conv_fwd(const void* x, const void* w, void* y, const void* res, float alpha, float beta,
                   const int* seg, int ngroups, long wstride, int N, int H, int W, int Ho, int Wo, int Cin,
                   int Cphys, int Ipad, int Cout, int Cstore, int stride, int ones, const int* kh, const int* kw,
                   const int* pt, const int* pl, int dtype, hipStream_t stream) {
  if (!x || !w || !y || N < 0 || ngroups < 1 || ngroups > HDMOE_MAX_GROUPS) return HDMOE_EINVAL;
  if (Ipad % 16 || Ipad < Cin || Cin != Cphys + (ones ? 1 : 0) || Cstore > Cout || stride < 1) return HDMOE_EINVAL;
  if (N == 0 || Ho * Wo == 0) return HDMOE_OK;
  ConvArgs a;
  a.x = x; a.w = w; a.y = y; a.res = res; a.seg = seg; a.wstride = wstride;
  a.N = N; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.Cin = Cin; a.Cphys = Cphys; a.Ipad = Ipad; a.Cout = Cout;
  a.Cstore = Cstore; a.stride = stride; a.ones = ones; a.ngroups = ngroups; a.alpha = alpha; a.beta = beta; a.n0 = 0;
  for (int g = 0; g < HDMOE_MAX_GROUPS; ++g) {
    const int s = g < ngroups ? g : 0;
    a.kh[g] = kh[s]; a.kw[g] = kw[s]; a.pt[g] = pt[s]; a.pl[g] = pl[s];
  }
  if (dtype == HDMOE_F32S) {                               // fp32 tensors on the bf16 pipe (conv6s.hip); callers check the domain first
    const int rc = conv6_split_try_launch(a, (long)ngroups * wstride, nullptr, stream);
    return rc == 1 ? HDMOE_EINVAL : rc;
  }
  if (dtype != HDMOE_F32 && dtype != HDMOE_BF16) return HDMOE_EDTYPE;
  {                                                        // k x k expert layers on 32 x 32 maps: whole-image streaming kernel (conv7.hip)
    const int rc = conv7_try_launch(a, dtype, stream);
    if (rc <= 0) return rc;
  }
  {                                                        // k x k layers of the experts / trunks: persistent LDS-DMA kernel (conv6.hip)
    const int rc = conv6_try_launch(a, nullptr, dtype, stream);
    if (rc <= 0) return rc;
  }
  {                                                        // pointwise, long contraction, few outputs (kgemm.hip)
    const int rc = kgemm_try_launch(a, dtype, stream);
    if (rc <= 0) return rc;
  }
  {                                                        // grouped fp32 linear on one-position rows, long input (mlinear.hip)
    const int rc = glin_try_launch(a, dtype, stream);
    if (rc <= 0) return rc;
  }
  if (stride == 1 && (long)Ho * Wo >= 64) {
    // ---- v2: LDS-staged 256-pixel tiles
    const int PT = 4 * CV2_MT * 32;
    int maxkh = 1, maxkw = 1;
    for (int g = 0; g < ngroups; ++g) { if (kh[g] > maxkh) maxkh = kh[g]; if (kw[g] > maxkw) maxkw = kw[g]; }
    const int esz = dtype == HDMOE_BF16 ? 2 : 4;
    const bool v5ok = !ones && Cstore % 4 == 0 && (uintptr_t)y % 16 == 0 && (uintptr_t)res % 16 == 0 && (uintptr_t)w % 16 == 0;
    // Tile = whole image rows, or (v5 only, images wider than 32 with a real kernel window) 8 x 32 blocks: a 4 x 64 tile of a
    // 7x7 layer needs a 10 x 70 halo, the 8 x 32 block 14 x 38.  tile2d != 0 tells the kernel the tile is not a linear pixel run.
    const int NTq = Cstore <= 32 ? 1 : 2;
    const bool vecq = Cphys % (16 / esz) == 0 && (uintptr_t)x % 16 == 0;
    // (only when the v5 launch below is certain: the older kernels assume linear tiles)
    const bool tile2d = v5ok && vecq && Ho >= 8 && Wo > 32 && Wo % 32 == 0 && maxkh * maxkw > 1 &&
                        (8 + maxkh - 1) * (32 + maxkw - 1) * 4 <= 9 * 256 && maxkw * 32 * NTq <= 576 &&
                        (long)maxkh * maxkw * Cout * Ipad < (1l << 26) && (long)H * W * Cphys < (1l << 30) &&
                        (size_t)80 * ((8 + maxkh - 1) * (32 + maxkw - 1) + maxkw * 32 * NTq) <= 80 * 1024;
    const int TW = tile2d ? 32 : (Wo < PT ? Wo : PT);
    int TH = PT / TW; if (TH > Ho) TH = Ho; if (TH < 1) TH = 1;
    const int tiles_y = cdiv(Ho, TH), tiles_x = cdiv(Wo, TW);
    const int NT = Cstore <= 32 ? 1 : 2;
    const int halo_cap = (TH + maxkh - 1) * (TW + maxkw - 1);
    const size_t lds = (size_t)80 * (halo_cap + maxkw * 32 * NT);
    const bool vec = Cphys % (16 / esz) == 0 && (uintptr_t)x % 16 == 0;
    dim3 grid(tiles_y * tiles_x, N, cdiv(Cstore, 32 * NT));
    // v5 / v3 (register prefetch): rows of weights per stage <= 576 (9 chunks/thread); halo <= 448 px (7 chunks/thread), or for
    // v5 <= 576 px (9 chunks/thread: 7x7 experts, whose halo of a 256-pixel tile is 14 x 38 or 22 x 22 pixels)
    const int halo_max = v5ok ? 9 * 256 : 7 * 256;
    if (vec && halo_cap * 4 <= halo_max && maxkw * 32 * NT <= 576 && (long)maxkh * maxkw * Cout * Ipad < (1l << 26) &&
        (long)H * W * Cphys < (1l << 30) &&
        (long)halo_cap * (TW + maxkw - 1) < (1l << 20)) {   // the kernels' (px * magic) >> 20 halo decode is exact while px * HWp < 2^20
      // a workgroup may use up to 80 KB of LDS under v5 (two still share a CU): needed by 7x7 experts with 64-channel tiles
      const size_t cap = v5ok ? 80 * 1024 : 64 * 1024;
      int tg = 576 / (maxkw * 32 * NT);
      if (tg > maxkh) tg = maxkh;
      while (tg > 1 && (size_t)80 * (halo_cap + tg * maxkw * 32 * NT) > 64 * 1024) --tg;       // prefer <= 64 KB
      const size_t lds3 = (size_t)80 * (halo_cap + tg * maxkw * 32 * NT);
      if (lds3 <= cap && v5ok) {
        static unsigned long long attr_set = 0;
        if (hdmoe_first_on_device(attr_set)) {               // opt every instantiation into > 64 KB of dynamic LDS, once per device
#define CV5_ATTR(TT, NTv, L, H) (void)hipFuncSetAttribute((const void*)conv_fwd5_kernel<TT, NTv, L, H>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)
#define CV5_ATTR4(TT, NTv) CV5_ATTR(TT, NTv, true, 7); CV5_ATTR(TT, NTv, false, 7); CV5_ATTR(TT, NTv, true, 9); CV5_ATTR(TT, NTv, false, 9)
          CV5_ATTR4(float, 1); CV5_ATTR4(float, 2); CV5_ATTR4(bf16, 1); CV5_ATTR4(bf16, 2);
        }
        // LDS-transposed epilogue needs whole 16-B pieces per pixel and a slab of 4 waves x 32 px x (32 NT + pad) elements
        const bool lepi = Cstore % (16 / esz) == 0 && (size_t)4 * 32 * (32 * NT + 16 / esz) * esz <= lds3;
        const bool big_halo = halo_cap * 4 > 7 * 256;
        const int nblk5 = (int)cdiv(Cstore, 32 * NT), ntiles5 = tiles_y * tiles_x;
        const int tg5 = tg | (tile2d ? 1 << 16 : 0) | (nblk5 << 17);     // bit 16: 8 x 32 block tiles; bits 17-30: channel blocks
        if (nblk5 > 0x3FFF) return HDMOE_EINVAL;
        const long npairs = (long)N * ntiles5;
        const dim3 grid5((unsigned)(8 * ((npairs + 7) / 8) * nblk5));
#define CV5_LAUNCH(TT, NTv)                                                                                                                        \
  do { if (lepi && !big_halo) hipLaunchKernelGGL((conv_fwd5_kernel<TT, NTv, true, 7>), grid5, dim3(256), lds3, stream, a, TH, TW, tiles_x, halo_cap, tg5, ntiles5);   \
       else if (!big_halo) hipLaunchKernelGGL((conv_fwd5_kernel<TT, NTv, false, 7>), grid5, dim3(256), lds3, stream, a, TH, TW, tiles_x, halo_cap, tg5, ntiles5); \
       else if (lepi) hipLaunchKernelGGL((conv_fwd5_kernel<TT, NTv, true, 9>), grid5, dim3(256), lds3, stream, a, TH, TW, tiles_x, halo_cap, tg5, ntiles5);       \
       else hipLaunchKernelGGL((conv_fwd5_kernel<TT, NTv, false, 9>), grid5, dim3(256), lds3, stream, a, TH, TW, tiles_x, halo_cap, tg5, ntiles5); } while (0)
        if (dtype == HDMOE_F32) { if (NT == 1) CV5_LAUNCH(float, 1); else CV5_LAUNCH(float, 2); }
        else { if (NT == 1) CV5_LAUNCH(bf16, 1); else CV5_LAUNCH(bf16, 2); }
        return hdmoe_launch_status();
      }
      if (lds3 <= 64 * 1024 && halo_cap * 4 <= 7 * 256) {
        for (int n0 = 0; n0 < N; n0 += ROWS_PER_LAUNCH) {
          a.n0 = n0; grid.y = N - n0 < ROWS_PER_LAUNCH ? N - n0 : ROWS_PER_LAUNCH;
          if (dtype == HDMOE_F32) {
            if (NT == 1) hipLaunchKernelGGL((conv_fwd3_kernel<float, 1>), grid, dim3(256), lds3, stream, a, TH, TW, tiles_x, halo_cap, tg);
            else hipLaunchKernelGGL((conv_fwd3_kernel<float, 2>), grid, dim3(256), lds3, stream, a, TH, TW, tiles_x, halo_cap, tg);
          } else {
            if (NT == 1) hipLaunchKernelGGL((conv_fwd3_kernel<bf16, 1>), grid, dim3(256), lds3, stream, a, TH, TW, tiles_x, halo_cap, tg);
            else hipLaunchKernelGGL((conv_fwd3_kernel<bf16, 2>), grid, dim3(256), lds3, stream, a, TH, TW, tiles_x, halo_cap, tg);
          }
        }
        return hdmoe_launch_status();
      }
    }
    if (lds <= 64 * 1024 && grid.x <= 65535 * 32) {
#define CV2_LAUNCH(TT, NTv)                                                                                                   \
  do { if (vec) hipLaunchKernelGGL((conv_fwd2_kernel<TT, NTv, true>), grid, dim3(256), lds, stream, a, TH, TW, tiles_x, halo_cap);  \
       else hipLaunchKernelGGL((conv_fwd2_kernel<TT, NTv, false>), grid, dim3(256), lds, stream, a, TH, TW, tiles_x, halo_cap); } while (0)
      for (int n0 = 0; n0 < N; n0 += ROWS_PER_LAUNCH) {
        a.n0 = n0; grid.y = N - n0 < ROWS_PER_LAUNCH ? N - n0 : ROWS_PER_LAUNCH;
        if (dtype == HDMOE_F32) { if (NT == 1) CV2_LAUNCH(float, 1); else CV2_LAUNCH(float, 2); }
        else { if (NT == 1) CV2_LAUNCH(bf16, 1); else CV2_LAUNCH(bf16, 2); }
      }
      return hdmoe_launch_status();
    }
  }
  if (dtype == HDMOE_F32) launch_conv_nb<float>(a, Cphys % 4 == 0 && ((uintptr_t)x % 16 == 0), stream);
  else launch_conv_nb<bf16>(a, Cphys % 8 == 0 && ((uintptr_t)x % 16 == 0), stream);
  return hdmoe_launch_status();
}

int hdmoe_conv_wgrad(const void* x, const void* dy, float* const* G, const int* seg, int ngroups, int N, int H,
                     int W, int Ho, int Wo, int Cin, int Cphys, int Cout, int stride, int ones, const int* kh,
                     const int* kw, const int* pt, const int* pl, int dtype, hipStream_t stream) {
  if (!x || !dy || !G || ngroups < 1 || ngroups > HDMOE_MAX_GROUPS || Cin != Cphys + (ones ? 1 : 0)) return HDMOE_EINVAL;
  if (N == 0 || Ho * Wo == 0) return HDMOE_OK;
  WgradArgs a;
  a.x = x; a.dy = dy; a.seg = seg; a.N = N; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.Cin = Cin; a.Cphys = Cphys;
  a.Cout = Cout; a.stride = stride; a.ones = ones; a.ngroups = ngroups;
  int maxtaps = 0;
  for (int g = 0; g < HDMOE_MAX_GROUPS; ++g) {
    const int s = g < ngroups ? g : 0;
    a.G[g] = G[s]; a.kh[g] = kh[s]; a.kw[g] = kw[s]; a.pt[g] = pt[s]; a.pl[g] = pl[s];
    if (kh[s] * kw[s] > maxtaps) maxtaps = kh[s] * kw[s];
  }
  a.tap_lo = 0;
  if (stride == 1 && !ones && ngroups == 1 && !seg && Ho == H && Wo == W && kh[0] == kw[0] && maxtaps > 1 && maxtaps * Cin <= 64 &&
      pt[0] == (kh[0] - 1) / 2 && pl[0] == (kw[0] - 1) / 2) {
    const int rc = swg_try_launch(x, dy, G[0], N, H, W, Cin, Cout, kh[0], pt[0], pl[0], dtype, stream);
    if (rc <= 0) return rc;
  }
  if (stride == 1 && !ones && ngroups == 1 && !seg && Ho == H && Wo == W && kh[0] == kw[0] && Cout <= 4) {
    const int rc = towg_try_launch(x, dy, G[0], N, H, W, Cin, Cout, kh[0], pt[0], pl[0], dtype, stream);
    if (rc <= 0) return rc;
  }
  if (stride == 1 && !ones && maxtaps == 1 && Ho == H && Wo == W) {
    bool plain = true;
    for (int g = 0; g < ngroups; ++g) plain = plain && kh[g] == 1 && kw[g] == 1 && pt[g] == 0 && pl[g] == 0;
    if (plain) {
      const int rc = lwg_try_launch(x, dy, G, seg, ngroups, N, (long)H * W, Cin, Cout, dtype, stream);
      if (rc <= 0) return rc;
    }
  }
  if (stride == 1) {
    // ---- v2: LDS-staged tiles of TH rows x TW columns (TH * TW <= 128 pixels); one launch per kernel-size class so the
    //      per-wave accumulator count (MAXT) matches the class (3x3 -> 3, 5x5 -> 7, 7x7 -> 13)
    Wg2Geom gm;
    const int PTpx = dtype == HDMOE_F32 ? Wg2PT<float>::N : Wg2PT<bf16>::N;
    // near-square tiles keep the halo small (a 2 x 64 tile of a 7x7 layer needs 8 x 70 halo pixels, a 4 x 32 tile 10 x 38);
    // single-row tensors (flattened token rows) take the whole tile width
    const int tw_cap = Ho == 1 ? PTpx : 32;
    gm.TW = Wo < tw_cap ? Wo : tw_cap;
    gm.TH = PTpx / gm.TW; if (gm.TH > Ho) gm.TH = Ho; if (gm.TH < 1) gm.TH = 1;
    gm.tw_shift = -1;
    for (int sft = 0; sft < 8; ++sft) if ((1 << sft) == gm.TW) gm.tw_shift = sft;
    gm.tiles_y = cdiv(Ho, gm.TH); gm.tiles_x = cdiv(Wo, gm.TW);
    const long units_l = (long)N * gm.tiles_y * gm.tiles_x;
    const int esz = dtype == HDMOE_BF16 ? 2 : 4;
    const int vw = 16 / esz;
    bool done[HDMOE_MAX_GROUPS] = {false};
    bool ok = units_l < (1l << 30) && (dtype == HDMOE_F32 || dtype == HDMOE_BF16);
    // feasibility of every class first (fall back to v1 as a whole otherwise)
    for (int g = 0; g < ngroups && ok; ++g) {
      const int taps = kh[g] * kw[g];
      const int passes = taps > 28 ? (taps + 27) / 28 : 1;
      const int mt = passes > 1 ? 7 : (taps < 4 ? taps : (taps + 3) / 4);
      const int OT = ((passes == 1 && mt > 7) || Cout <= 32) ? 1 : 2;
      const int OBP = (OT == 2 && esz == 2) ? 96 : 32 * OT;
      const size_t lds = (size_t)esz * (PTpx * OBP + (gm.TH + kh[g] - 1) * (gm.TW + kw[g] - 1) * WG2_IB);
      if (lds > 64 * 1024 || mt > 13) ok = false;
      // the kernel decodes halo pixels with (px * magic) >> 20, exact only while px * HWp < 2^20 (px < HHp * HWp)
      const long HWp = gm.TW + kw[g] - 1, HHp = gm.TH + kh[g] - 1;
      if (HHp * HWp * HWp >= (1l << 20)) ok = false;
    }
    if (ok) {
      for (int g = 0; g < ngroups; ++g) {
        if (done[g]) continue;
        gm.ngr = 0;
        for (int g2 = g; g2 < ngroups; ++g2)
          if (!done[g2] && kh[g2] == kh[g] && kw[g2] == kw[g]) { gm.groups[gm.ngr++] = g2; done[g2] = true; }
        const int taps = kh[g] * kw[g];
        // 13 accumulator sets per wave (7x7) do not fit the register file (the MAXT = 13 instantiation spills ~250 VGPRs):
        // such classes run as ceil(taps / 28) passes of the 7-set kernel over tap ranges (the tiles are staged once per pass)
        const int passes = taps > 28 ? (taps + 27) / 28 : 1;
        const int mt = passes > 1 ? 7 : (taps < 4 ? taps : (taps + 3) / 4);
        const int OT = ((passes == 1 && mt > 7) || Cout <= 32) ? 1 : 2;
        const int OB = 32 * OT, OBP = (OT == 2 && esz == 2) ? OB + 32 : OB;
        const int halo = (gm.TH + kh[g] - 1) * (gm.TW + kw[g] - 1);
        const size_t lds = (size_t)esz * (PTpx * OBP + halo * WG2_IB);
        const int nx_cap = (esz == 2 ? 6 : 8) * 256;                    // register-prefetch capacity (16-B halo chunks)
        const bool vec = Cout % vw == 0 && Cphys % vw == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)dy % 16 == 0 &&
                         halo * (WG2_IB / vw) <= nx_cap;
        const int ibs = cdiv(Cin, WG2_IB), obs = cdiv(Cout, OB);
        // every workgroup ends with an atomic flush of its [taps][OB][32] accumulators; the flush traffic is
        // (pixel partitions) x (weight bytes) at ~1.3 TB/s, so keep the partition count near 256 / (ibs * obs) per launch
        // Every workgroup ends with an fp32 atomic flush of its [taps][OB][32] accumulators (~1.3 TB/s chip-wide), so the
        // flush traffic is (pixel partitions) x (weight bytes).  fp32 units are long (64-cycle MFMAs): fill the chip exactly
        // once.  bf16 units are short and the flush dominates: fewer, longer workgroups measured best (launch_table sweeps).
        long upw;
        if (esz == 4) {
            static const long f32_parts = getenv("HDMOE_WG_PARTS_F32") ? atol(getenv("HDMOE_WG_PARTS_F32")) : 512;
            long parts = f32_parts / ((long)ibs * obs); if (parts < 8) parts = 8;
            const long class_units = (units_l * gm.ngr + ngroups - 1) / ngroups;   // assume balanced routing
            upw = (class_units + parts - 1) / parts;
        } else {
            static const long bf16_parts = getenv("HDMOE_WG_PARTS") ? atol(getenv("HDMOE_WG_PARTS")) : 384;
            // classes with more than 9 taps flush (taps x 64 x 32) floats per workgroup: fewer, longer workgroups (bench sweep:
            // 384 -> 256 partitions = -0.7 ms/step at B = 256)
            static const long bf16_parts_big = getenv("HDMOE_WG_PARTS_BIG") ? atol(getenv("HDMOE_WG_PARTS_BIG")) : 256;
            long parts = (taps > 9 ? bf16_parts_big : bf16_parts) / ((long)ibs * obs); if (parts < 8) parts = 8;
            upw = (units_l + parts - 1) / parts;
        }
        if (upw < 1) upw = 1;
        gm.upw = (int)upw;
        gm.chunks = (int)((units_l + upw - 1) / upw);
        dim3 grid(ibs, obs, gm.chunks * gm.ngr);
#define WG2_LAUNCH(TT, OTv, MT)                                                                                           \
  do { if (vec) hipLaunchKernelGGL((conv_wgrad2_kernel<TT, OTv, MT, true>), grid, dim3(256), lds, stream, a, gm);          \
       else hipLaunchKernelGGL((conv_wgrad2_kernel<TT, OTv, MT, false>), grid, dim3(256), lds, stream, a, gm); } while (0)
#define WG2_BY_MT(TT, OTv)                                          \
  if (mt <= 3) WG2_LAUNCH(TT, OTv, 3);                               \
  else if (mt <= 7) WG2_LAUNCH(TT, OTv, 7);                          \
  else WG2_LAUNCH(TT, OTv, 13);
        for (int pass = 0; pass < passes; ++pass) {
          a.tap_lo = 28 * pass;
          if (dtype == HDMOE_F32) { if (OT == 2) { WG2_BY_MT(float, 2) } else { WG2_BY_MT(float, 1) } }
          else { if (OT == 2) { WG2_BY_MT(bf16, 2) } else { WG2_BY_MT(bf16, 1) } }
        }
        a.tap_lo = 0;
      }
      return hdmoe_launch_status();
    }
  }
  const int cpl = dtype == HDMOE_BF16 ? 2 : 1;
  a.ob_count = cdiv(Cout, 32 * cpl);
  a.ib_count = cdiv(Cin, 32 * cpl);
  const long tiles = (long)maxtaps * a.ob_count * a.ib_count;
  // samples per wave: keep >= ~2048 waves in flight, but at least 1 and at most 16 samples per wave
  long spw = ((long)N * tiles) / 2048;
  if (spw < 1) spw = 1;
  if (spw > 16) spw = 16;
  a.spw = (int)spw;
  dim3 grid((unsigned)tiles, cdiv(N, 4 * spw));
  if (dtype == HDMOE_F32) hipLaunchKernelGGL(conv_wgrad_kernel<float>, grid, dim3(256), 0, stream, a);
  else if (dtype == HDMOE_BF16) hipLaunchKernelGGL(conv_wgrad_kernel<bf16>, grid, dim3(256), 0, stream, a);
  else return HDMOE_EDTYPE;
  return hdmoe_launch_status();
}


/* Grouped k x k bf16 conv (the conv6 domain: stride 1, "same" padding, square k in {3,5,7}, Cin % 32 == Cout % 32 == 0) with the FiLM of
 * Unet_block fused into its epilogue (reference model_components.py:242-246): y = alpha * conv(x, w) as hdmoe_conv_fwd, and
 * h = dropout_p(mp_silu(y * e[n][c])) -- bit for bit what hdmoe_film_silu_drop_fwd(h, y, e, ..., seed, seed_dev, p) writes.
 * Returns 1 without launching when the layer is outside the domain (the caller then issues the two launches). */
int hdmoe_conv_fwd_film(const void* x, const void* w, void* y, void* h, const float* e, unsigned long long seed,
                        const unsigned long long* seed_dev, float p, float alpha, const int* seg, int ngroups, long wstride, int N, int H, int W,
                        int Cin, int Cout, const int* kh, const int* kw, const int* pt, const int* pl, int dtype, hipStream_t stream) {
  if (!x || !w || !y || !h || !e || N < 0 || ngroups < 1 || ngroups > HDMOE_MAX_GROUPS || p < 0.f || p >= 1.f) return HDMOE_EINVAL;
  if (dtype != HDMOE_BF16) return 1;
  if (N == 0) return HDMOE_OK;
  ConvArgs a;
  a.x = x; a.w = w; a.y = y; a.res = nullptr; a.seg = seg; a.wstride = wstride;
  a.N = N; a.H = H; a.W = W; a.Ho = H; a.Wo = W; a.Cin = Cin; a.Cphys = Cin; a.Ipad = Cin; a.Cout = Cout; a.Cstore = Cout;
  a.stride = 1; a.ones = 0; a.ngroups = ngroups; a.alpha = alpha; a.beta = 0.f; a.n0 = 0;
  for (int g = 0; g < HDMOE_MAX_GROUPS; ++g) {
    const int s = g < ngroups ? g : 0;
    a.kh[g] = kh[s]; a.kw[g] = kw[s]; a.pt[g] = pt[s]; a.pl[g] = pl[s];
  }
  ConvFuse f;
  f.in_scale = nullptr; f.in_shift = nullptr; f.stats = nullptr; f.in_relu = 0;
  f.film_e = e; f.film_h = h; f.film_seed_dev = seed_dev; f.film_seed = seed; f.film_p = p;
  return conv6_try_launch(a, &f, dtype, stream);
}

/* 3x3 "same" conv of an fp32 tensor as split bf16 (conv6s) with GroupNorm(1, C) + ReLU of the PRODUCING layer applied to x while it is
 * staged (in_scale / in_shift [N][Cin] from hdmoe_gn1_finalize, or NULL) and per-sample partial statistics of y written to stats_ws
 * ([N][hdmoe_conv_split_stats_slots][2] floats, or NULL).  w: [hi | lo] bf16 image, wplane elements per plane.  Returns 1 when the shape
 * is outside the kernel's domain (nothing launched). */
int hdmoe_conv_split_stats_slots(int H, int W, int Cout) {
  if (!(W == 16 || W % 32 == 0) || H < 8 || Cout % 32) return 0;
  const int TW = W >= 32 ? 32 : 16, TH = 256 / TW;
  const int tpi = (W / TW) * (int)cdiv(H, TH);
  return tpi * (Cout / (Cout % 64 == 0 ? 64 : 32)) * 4;
}
int hdmoe_conv_fwd_split_gn(const void* x, const void* w, void* y, const float* in_scale, const float* in_shift, int in_relu, float* stats_ws,
                            long wstride, long wplane, int N, int H, int W, int Cin, int Cout, float alpha, hipStream_t stream) {
  if (!x || !w || !y || (in_scale == nullptr) != (in_shift == nullptr)) return HDMOE_EINVAL;
  ConvArgs a;
  a.x = x; a.w = w; a.y = y; a.res = nullptr; a.seg = nullptr; a.wstride = wstride;
  a.N = N; a.H = H; a.W = W; a.Ho = H; a.Wo = W; a.Cin = Cin; a.Cphys = Cin; a.Ipad = Cin; a.Cout = Cout; a.Cstore = Cout;
  a.stride = 1; a.ones = 0; a.ngroups = 1; a.alpha = alpha; a.beta = 0.f; a.n0 = 0;
  for (int g = 0; g < HDMOE_MAX_GROUPS; ++g) { a.kh[g] = 3; a.kw[g] = 3; a.pt[g] = 1; a.pl[g] = 1; }
  ConvFuse f;
  f.in_scale = in_scale; f.in_shift = in_shift; f.in_relu = in_relu; f.stats = stats_ws;
  return conv6_split_try_launch(a, wplane, &f, stream);
}

}  // extern "C"
