// Shared device helpers for the AVT gfx950 kernels (bf16 conversion, GELU variants, counter-based RNG,
// wave reductions) and the host-side error channel of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;                                               // raw bf16 bits in HBM
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;           // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

#define AVT_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {                      // round-to-nearest-even, NaN kept quiet
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
// two fp32 -> packed bf16x2 in ONE instruction: clang selects gfx950's v_cvt_pk_bf16_f32 (round-to-nearest-even) for the
// vector fptrunc, and -- unlike inline asm -- keeps track of the MFMA -> VALU wait states of its operands.
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bflo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfhi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// erf to ~1.5e-7 absolute (Abramowitz-Stegun 7.1.26): one v_exp + one v_rcp + 5 FMAs instead of the libm call --
// far below bf16 resolution, and it keeps the GEMM epilogue off the critical path.
__device__ __forceinline__ float fast_erf(float x) {
  float ax = fabsf(x);
  float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float poly = fmaf(fmaf(fmaf(fmaf(1.061405429f, t, -1.453152027f), t, 1.421413741f), t, -0.284496736f), t, 0.254829592f) * t;
  float r = 1.0f - poly * __expf(-ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float fast_tanh(float u) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * u)); }

// exact (erf) GELU -- timm Mlp act_layer=nn.GELU ; tanh GELU -- HF "gelu_new"
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_erf(float x) {
  return 0.5f * (1.0f + fast_erf(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
__device__ __forceinline__ float gelu_tanh(float x) {
  float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  return 0.5f * x * (1.0f + fast_tanh(u));
}
__device__ __forceinline__ float dgelu_tanh(float x) {
  float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  float t = fast_tanh(u);
  return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * x * x);
}

// value and derivative together (one erf / one exp shared): the forward epilogue stores GELU'(h) for the backward pass
__device__ __forceinline__ void gelu_erf_both(float x, float& y, float& dy) {
  float ax = fabsf(x) * 0.70710678118654752f;
  float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float ex = __expf(-ax * ax);                                   // = exp(-x^2/2)
  float poly = fmaf(fmaf(fmaf(fmaf(1.061405429f, t, -1.453152027f), t, 1.421413741f), t, -0.284496736f), t, 0.254829592f) * t;
  float cdf = 0.5f * (1.0f + copysignf(1.0f - poly * ex, x));
  y = x * cdf;
  dy = cdf + x * 0.3989422804014327f * ex;
}
// (the erf GELU of the GEMM epilogues is a table look-up since round 4: csrc/gemm.hip, "GELU by table"; the packed-polynomial form of
// rounds 2-3 -- degree-8 fit of Phi on |x| <= 3 sqrt 2 + one v_exp_f32 per element -- measured 4 % slower on the fc1-forward launches)
__device__ __forceinline__ void gelu_tanh_both(float x, float& y, float& dy) {
  float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  float t = fast_tanh(u);
  y = 0.5f * x * (1.0f + t);
  dy = 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * x * x);
}

// Counter-based RNG for dropout: keep(seed, idx) is a pure function, so backward recomputes the mask.
__device__ __forceinline__ uint32_t rng_u32(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 16);
}
__device__ __forceinline__ bool drop_keep(uint64_t seed, uint64_t idx, uint32_t thresh) { return rng_u32(seed, idx) >= thresh; }
// Indirect seeds (ABI 9, include/avt_hip.h "captured steps"): a seed argument with bit 63 set is not the seed but where to find it -- bits 0..47 = the
// device address of a uint64 holding a base seed, bits 48..62 = an offset added to it -- so that a step captured into a hipGraph draws fresh masks at
// every replay (the host rewrites the base seed between replays; the launches' arguments stay what they were).  Plain seeds have bit 63 clear.
__device__ __forceinline__ uint64_t resolve_seed(uint64_t s) {
  if (s >> 63) s = *(const uint64_t*)(uintptr_t)(s & 0x0000FFFFFFFFFFFFull) + ((s >> 48) & 0x7FFFull);
  return s;
}
static inline uint32_t drop_threshold(float p) {
  double t = (double)p * 4294967296.0;
  if (t < 0) t = 0;
  if (t > 4294967295.0) t = 4294967295.0;
  return (uint32_t)t;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- transposing LDS read issued through inline assembly -------------------------------------------------------------------
// hipcc cannot tell that __builtin_amdgcn_ds_read_tr16_b64 does not alias LDS-DMA (buffer_load ... lds) transfers still in flight
// and puts s_waitcnt vmcnt(0) in front of it; kernels that prefetch by LDS-DMA and place their own waits use this form instead.
// `addr` = the lane's LDS byte address, OFF = compile-time offset.  The result may only be used after an explicit s_waitcnt lgkmcnt.
// The two registers come back as 32-bit words and a fragment is put together from two reads with 32-bit vector operations only
// (tr_join: a register sequence, no instruction).  Going through 16-bit element vectors (a union with short4) made hipcc emit one
// `v_bfi_b32 d, 0xffff, s, s` per register to "merge" the halves -- 32 vector instructions per K-loop trip of the weight-gradient kernel,
// placed BEFORE the hand-written wait, i.e. reading registers whose LDS data the compiler had no reason to believe outstanding
// (tools/isa_async_check.py scans the generated ISA for exactly that).
template <int OFF>
__device__ __forceinline__ u32x2_t ds_read_tr_na(uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read immediate offset");
  u32x2_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ bf16x8_t tr_join(u32x2_t lo, u32x2_t hi) {
  const u32x4_t w = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bf16x8_t, w);
}
__device__ __forceinline__ uint32_t lds_addr32(const char* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }

// ---- streamed tensors (read or written exactly once by a kernel): non-temporal hint -----------------------------------------
// Measured per kernel (profiles/r04_cache_policy.txt, 2560 frames): LayerNorm forward 295 -> 283 us, backward 623 -> 602 us with nt
// loads and stores (adopted there); the fused SGD does not gain (1.83-2.14 vs 2.19 ms) and the attention kernels lose with nt on
// their 8-byte stores (forward 727 -> 945 us) and on their tile loads (708 -> 820 us): those keep the default policy.
#define AVT_LDG_NT(p) __builtin_nontemporal_load(p)
#define AVT_STG_NT(p, v) __builtin_nontemporal_store((v), (p))

// ---- host-side error channel -------------------------------------------------------------------------
void avt_set_error(const char* fmt, ...);
#define AVT_CHECK(cond, ...) do { if (!(cond)) { avt_set_error(__VA_ARGS__); return -1; } } while (0)
#define AVT_LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) { \
  avt_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); return (int)e__; } } while (0)
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// ---- run-to-run identical parameter gradients ----------------------------------------------------------
// Kernels that fold many workgroups into one fp32 vector (bias / gamma / beta / embedding gradients) either merge with fp32
// atomics (order = arrival order: the last bits differ from run to run) or, when the caller passes a partials workspace,
// store one partial vector per workgroup ("slot") with plain stores: part[(q * nslots + slot) * n + i], q = quantity.
// avt_reduce_partials then adds the slots in a FIXED tree (a function of nslots only) and does out_q[i] += sum.
int avt_reduce_partials(const float* part, int nslots, long n, float* const* outs, int nq, hipStream_t stream);
