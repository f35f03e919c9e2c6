// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the HRI-EMO fusion path.
// wave = 64 lanes everywhere; bf16 storage, fp32 arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define LDS_PTR(T) __attribute__((address_space(3))) T*
#define GLB_PTR(T) __attribute__((address_space(1))) T*

// ---------------------------------------------------------------- error plumbing (host)
void hriemo_set_error(const char* fmt, ...);
#define HRIEMO_CHECK(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      hriemo_set_error(__VA_ARGS__);       \
      return 1;                            \
    }                                      \
  } while (0)
#define HRIEMO_LAUNCH_CHECK(name)                                         \
  do {                                                                    \
    hipError_t e__ = hipGetLastError();                                   \
    if (e__ != hipSuccess) {                                              \
      hriemo_set_error("%s launch failed: %s", name, hipGetErrorString(e__)); \
      return 2;                                                           \
    }                                                                     \
  } while (0)

// optional per-kernel-class event timing (bench.py roofline leg); see prof.cpp
enum { HP_GEMM_NT = 0, HP_GEMM_NN, HP_GEMM_TN, HP_ATTN_FWD, HP_ATTN_BWD_DQ, HP_ATTN_BWD_DKV, HP_ROWOPS, HP_GEMM_MX8, HP_NCLASS };
void hriemo_prof_begin(int cls, hipStream_t s);
void hriemo_prof_end(int cls, hipStream_t s, double work);

// ---------------------------------------------------------------- wave reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---------------------------------------------------------------- dropout RNG
// Counter hash (murmur3 finaliser) -- one 16-bit uniform per element, replayable in the
// backward from (key, a, b) alone so no mask is ever stored.  key = site_key(seed, site, c).
__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ uint32_t site_key(uint64_t seed, uint32_t site, uint32_t c) {
  uint32_t k = fmix32((uint32_t)seed ^ 0x9e3779b9u);
  k = fmix32(k + (uint32_t)(seed >> 32) * 0x85ebca77u + site * 0x27d4eb2fu);
  return fmix32(k + c * 0x165667b1u);
}
// Per-element uniforms.  One hash serves TWO adjacent elements of the minor index b (b = key position in
// attention, column in the row kernels): x = mix24(key + a*CA + (b>>1)*CB); element b keeps the low 16 bits
// if b is even, the high 16 bits if odd.  mix24 uses 24-bit multiplies (full-rate v_mul_u32_u24; the 32-bit
// v_mul_lo_u32 of a murmur finaliser is quarter rate and made the hash the largest VALU cost of the
// attention kernels).  Statistics (drop rate, neighbour correlations) match murmur3's finaliser:
// scripts_dev/ rng study in DESIGN.md.  thr16 = round(p * 65536): P(drop) = thr16 / 65536.
#define DROP_CA 0x9e3779b1u
#define DROP_CB 0x85ebca77u
__host__ __device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) {
  return (a & 0xffffffu) * (b & 0xffffffu);          // low 32 bits of the 24x24 product: one v_mul_u32_u24
}
__host__ __device__ __forceinline__ uint32_t mix24(uint32_t x) {
  x ^= x >> 16; x = mul24(x, 0xB2AE35u); x ^= x >> 13; x = mul24(x, 0xEBCA6Bu); x ^= x >> 15;
  return x;
}
// base for (a, pair index bp); further pairs of the same a are base + j*DROP_CB
__host__ __device__ __forceinline__ uint32_t drop_base(uint32_t key, uint32_t a, uint32_t bp) {
  return key + a * DROP_CA + bp * DROP_CB;
}
__host__ __device__ __forceinline__ bool keep_lo(uint32_t x, uint32_t thr16) { return (x & 0xffffu) >= thr16; }
__host__ __device__ __forceinline__ bool keep_hi(uint32_t x, uint32_t thr16) { return (x >> 16) >= thr16; }
// generic single-element form (true = element is kept)
__host__ __device__ __forceinline__ bool keep16(uint32_t key, uint32_t a, uint32_t b, uint32_t thr16) {
  const uint32_t x = mix24(drop_base(key, a, b >> 1));
  return (b & 1u) ? keep_hi(x, thr16) : keep_lo(x, thr16);
}

struct DropCfg {
  uint64_t seed;
  uint32_t site;
  uint32_t thr16;     // 0 => dropout off
  float inv_keep;     // 1 / (1 - thr16/65536)
};
// Effective seed = immediate seed + *seed_dev (device word, may be NULL).  The device word lets a captured
// hipGraph draw fresh masks on every replay: the graph itself bumps it, the kernel arguments stay frozen.
__device__ __forceinline__ uint64_t eff_seed(uint64_t seed, const unsigned long long* seed_dev) {
  return seed + (seed_dev != nullptr ? (uint64_t)*seed_dev : 0ull);
}
static inline DropCfg make_drop(float p, uint64_t seed, uint32_t site) {
  DropCfg d;
  d.seed = seed; d.site = site;
  long t = (long)(p * 65536.0 + 0.5);
  if (t < 0) t = 0;
  if (t > 65535) t = 65535;
  d.thr16 = (uint32_t)t;
  d.inv_keep = (float)(1.0 / (1.0 - (double)t / 65536.0));
  return d;
}

// ---------------------------------------------------------------- MX-fp8 block quantiser (gemm_mx8.hip, rowops.hip)
// 8 consecutive fp32 values of this lane; the 32-element block is this lane and its three neighbours (lane ^ 1, ^ 2, ^ 3).
// Returns the 8 e4m3 bytes and the biased E8M0 scale exponent e: scale 2^(e-127) = the smallest power of two that brings
// the block maximum inside e4m3's finite range (448), computed exactly from the bits: amax = 1.m x 2^Ea fits under
// 1.75 x 2^(E+8) iff E >= Ea - 8 (1.m <= 1.75) or E >= Ea - 7 (1.m > 1.75).  All 64 lanes must call it (shuffles).
__device__ __forceinline__ __attribute__((ext_vector_type(2))) int mx8_block(const float (&f)[8], int& e_out) {
  float amax = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(f[j]));
  amax = fmaxf(amax, __shfl_xor(amax, 1));
  amax = fmaxf(amax, __shfl_xor(amax, 2));
  const unsigned ab = __float_as_uint(amax);
  int e = (int)(ab >> 23) - 8 + ((ab & 0x7fffffu) > 0x600000u ? 1 : 0);
  e = amax == 0.f ? 0 : (e < 1 ? 1 : (e > 253 ? 253 : e));
  const float inv = amax == 0.f ? 0.f : __uint_as_float((unsigned)(254 - e) << 23);     // 2^-(e-127)
  int w0 = 0, w1 = 0;
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, w0, false);
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, w0, true);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, w1, false);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, w1, true);
  e_out = e;
  return (__attribute__((ext_vector_type(2))) int){w0, w1};
}

// ---------------------------------------------------------------- small vector helpers
__device__ __forceinline__ void bf8_to_f32(const bf16x8& v, float* f) {
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
__device__ __forceinline__ bf16x8 f32_to_bf8(const float* f) {
  bf16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (bf16_t)f[i];
  return v;
}
