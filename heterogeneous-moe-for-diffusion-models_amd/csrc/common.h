// Shared device/host helpers for the HDMOE HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HDMOE_OK 0
#define HDMOE_EINVAL (-1)
#define HDMOE_EDTYPE (-2)
#define HDMOE_ELAUNCH (-3)

// dtype codes of the C ABI
#define HDMOE_F32 0
#define HDMOE_BF16 1
#define HDMOE_F32S 2     /* fp32 tensors, split-bf16 arithmetic: weight images are bf16 [hi | lo] planes */
#define HDMOE_F16 3      /* hdmoe_cast only: fp16 tensors at the module boundary (converted at ingest / egress) */

#define HDMOE_MAX_GROUPS 8
#define MP_SILU_DIV 0.596f

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define DEVI __device__ __forceinline__

// ---- transposing LDS reads through inline asm
// The compiler puts "s_waitcnt vmcnt(0)" in front of every ds_read_b64_tr_b16 it emits itself while an LDS-DMA (buffer_load ... lds) is in
// flight: it cannot tell that the DMA fills the OTHER buffer, so a kernel that requests the next tile and then reads fragments of this one
// waits out the whole memory round trip first -- DMA and MFMAs never overlap.  (It does not do this for plain ds_read_b128; found in the ISA of
// the weight-gradient kernels in round 4, where every k-step of wgrad6 and every unit of wgrad7 paid it.)  The asm forms are invisible to that
// bookkeeping; in exchange the waits are ours: issue -> lds_tr_wait() -> lds_tr_take() (the take ties the registers to the wait, so no MFMA
// can be scheduled in front of it).  Addresses are byte offsets into LDS (lds_addr_of).
typedef __attribute__((ext_vector_type(4))) short hd_s16x4;
DEVI unsigned lds_addr_of(const void* p) { return (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)(p); }
DEVI void lds_tr2_issue(hd_s16x4& lo, hd_s16x4& hi, unsigned a0, unsigned a1) {
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3" : "=&v"(lo), "=&v"(hi) : "v"(a0), "v"(a1));
}
DEVI void lds_tr_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
DEVI bf16x8 lds_tr2_take(hd_s16x4& lo, hd_s16x4& hi) {
  asm volatile("" : "+v"(lo), "+v"(hi));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

DEVI float to_f(float v) { return v; }
DEVI float to_f(bf16 v) { return (float)v; }
template <typename T> DEVI T from_f(float v);
template <> DEVI float from_f<float>(float v) { return v; }
template <> DEVI bf16 from_f<bf16>(float v) { return (bf16)v; }

// ---- 8-element fragments (one MFMA operand slice per lane) -----------------
template <typename T> struct Frag8;
template <> struct Frag8<bf16> {
  bf16x8 v;
  DEVI void zero() { v = (bf16x8)(0); }
  DEVI void set(int j, float f) { v[j] = (bf16)f; }
  DEVI float get(int j) const { return (float)v[j]; }
};
template <> struct Frag8<float> {
  float v[8];
  DEVI void zero() {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
  }
  DEVI void set(int j, float f) { v[j] = f; }
  DEVI float get(int j) const { return v[j]; }
};

// aligned 8-element load (16 B for bf16, 2x16 B for f32); p must be 16-B aligned
DEVI void load8(Frag8<bf16>& f, const bf16* p) { f.v = *reinterpret_cast<const bf16x8*>(p); }
DEVI void load8(Frag8<float>& f, const float* p) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
  f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
}

// D(32x32) += A(32x16) * B(16x32).  Lane l = (r = l & 31, h = l >> 5) holds
// A[r][8h + j] and B[8h + j][r] in element j.  For f32 the 16-deep step is
// eight 32x32x2 MFMAs; MFMA j contracts the k-slots {8*0 + j, 8*1 + j}, the
// same slot->k map on both operands, so the sum over k is unchanged.
DEVI void mma32(f32x16& acc, const Frag8<bf16>& a, const Frag8<bf16>& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
}
DEVI void mma32(f32x16& acc, const Frag8<float>& a, const Frag8<float>& b) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[j], b.v[j], acc, 0, 0, 0);
}
// accumulator element `reg` of lane l sits at row (reg&3) + 8*(reg>>2) + 4*(l>>5), col l&31
DEVI int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// ---- reductions -----------------------------------------------------------------
DEVI float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
DEVI float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block-wide sum for blockDim.x <= 1024 (multiple of 64); `sm` >= 16 floats; all threads get the result
DEVI float block_sum(float v, float* sm) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) sm[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += sm[i];
  return t;
}

// sigmoid through v_rcp_f32 (1 ulp) instead of an IEEE division (~10 instructions): the activation sits in the epilogue of the fused
// block kernel (blk6_body.h), 16 of them per pixel block and lane
DEVI float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
DEVI float silu_f(float x) { return x * sigmoid_f(x); }
DEVI float mp_silu_f(float x) { return silu_f(x) * (1.f / MP_SILU_DIV); }
// d/dx [silu(x)/0.596]
DEVI float mp_silu_grad_f(float x) {
  const float s = sigmoid_f(x);
  return s * (1.f + x * (1.f - s)) * (1.f / MP_SILU_DIV);
}

// ---- 16-byte vector access helpers (guide G13: bf16 as 8-wide, fp32 as 4-wide) --------------------------------------
template <typename T> struct VT;
template <> struct VT<bf16> { static constexpr int W = 8; };
template <> struct VT<float> { static constexpr int W = 4; };
template <typename T> DEVI void vload(float* f, const T* p);
template <> DEVI void vload<bf16>(float* f, const bf16* p) {
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (float)v[j];
}
template <> DEVI void vload<float>(float* f, const float* p) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
}
template <typename T> DEVI void vstore(T* p, const float* f);
template <> DEVI void vstore<bf16>(bf16* p, const float* f) {
  bf16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (bf16)f[j];
  *reinterpret_cast<bf16x8*>(p) = v;
}
template <> DEVI void vstore<float>(float* p, const float* f) { *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]); }
static inline bool al16(const void* p) { return p == nullptr || ((uintptr_t)p & 15) == 0; }

static inline int hdmoe_launch_status() {
  return hipGetLastError() == hipSuccess ? HDMOE_OK : HDMOE_ELAUNCH;
}
// hipFuncSetAttribute applies to the CURRENT device only: "done once" flags are kept per device (bit = device id), so a process that drives
// several devices raises the dynamic-LDS limit on each of them
static inline bool hdmoe_first_on_device(unsigned long long& mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (mask & bit) return false;
  mask |= bit;
  return true;
}
static inline unsigned cdiv(long a, long b) { return (unsigned)((a + b - 1) / b); }

// ---------------------------------------------------------------- counter RNG (Philox4x32-10)
DEVI void philox(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1, uint32_t* o) {
  uint32_t c[4] = {c0, c1, 0u, 0u};
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = c[3];
}
DEVI float u01(uint32_t v) { return ((float)(v >> 8) + 0.5f) * (1.f / 16777216.f); }

// the Philox key is (per-call salt) + (device step counter) * golden ratio: a captured hipGraph replays with fresh randomness
// because the counter lives in device memory and is advanced by hdmoe_seed_advance once per step
DEVI void mix_seed(uint32_t& lo, uint32_t& hi, const unsigned long long* seed_dev) {
  if (seed_dev) {
    const unsigned long long k = (((unsigned long long)hi << 32) | lo) + (*seed_dev) * 0x9E3779B97F4A7C15ull;
    lo = (uint32_t)k; hi = (uint32_t)(k >> 32);
  }
}
