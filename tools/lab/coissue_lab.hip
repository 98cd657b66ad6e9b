// Can the vector ALU and the matrix pipe of one SIMD work at the same time?
//   (a) two waves per SIMD: wave A issues only v_mfma_f32_32x32x16_bf16, wave B only v_pk_fma_f32 -- each alone, then together
//   (b) one wave per SIMD interleaving k packed FMAs after every MFMA (independent registers)
// Durations in s_memtime ticks per wave; "together ~ max(alone)" = the pipes overlap, "together ~ sum" = they serialise.
// This decides whether a GEMM epilogue (vector ALU) can hide under another tile's K loop (matrix pipe) on the same CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// mode bit 0: MFMA waves (0-3) work; bit 1: VALU waves (4-7) work
__global__ __launch_bounds__(512) void two_wave_kernel(long long* out, float* sink, int mode, int n_mfma, int n_valu, float seed) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __builtin_amdgcn_s_barrier();
  long long t0 = __builtin_readcyclecounter();
  if (wave < 4) {
    if (mode & 1) {
      f32x16 acc[8];
      for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
      bf16x8 a, b;
      for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + lane * 0.01f + e); b[e] = (__bf16)(seed * 0.5f - lane * 0.02f + e); }
      for (int it = 0; it < n_mfma; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
      }
      float s = 0.f;
      for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
      if (s == 123.456f) sink[tid] = s;
    }
  } else {
    if (mode & 2) {
      f32x2 v[8];
      for (int i = 0; i < 8; ++i) v[i] = (f32x2){seed + i + lane, seed - i};
      const f32x2 c1 = {0.999f, 1.001f}, c2 = {1e-3f, -1e-3f};
      for (int it = 0; it < n_valu; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = v[i] * c1 + c2;          // v_pk_fma_f32, 8 independent chains
      }
      float s = 0.f;
      for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
      if (s == 123.456f) sink[tid] = s;
    }
  }
  long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

// one wave per SIMD (256 threads): every MFMA followed by K independent packed FMAs
template <int K>
__global__ __launch_bounds__(256) void interleave_kernel(long long* out, float* sink, int n, float seed) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + lane * 0.01f + e); b[e] = (__bf16)(seed * 0.5f - lane * 0.02f + e); }
  f32x2 v[8];
  for (int i = 0; i < 8; ++i) v[i] = (f32x2){seed + i + lane, seed - i};
  const f32x2 c1 = {0.999f, 1.001f}, c2 = {1e-3f, -1e-3f};
  __builtin_amdgcn_s_barrier();
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) v[(i + k) & 7] = v[(i + k) & 7] * c1 + c2;
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15] + v[i][0] + v[i][1];
  if (s == 123.456f) sink[tid] = s;
  if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
}

int main() {
  long long* out; float* sink;
  CK(hipMalloc(&out, 8 * 4096)); CK(hipMalloc(&sink, 4 * 512));
  long long h[8];
  const int NM = 2000, NV = 8000;       // 16000 MFMAs (x 32 cycles at 8 passes x 4) vs 64000 packed FMAs (x 4 cycles)
  const char* names[] = {"", "MFMA waves alone", "VALU waves alone", "both together"};
  for (int mode = 1; mode <= 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(two_wave_kernel, dim3(256), dim3(512), 0, 0, out, sink, mode, NM, NV, 1.5f); CK(hipDeviceSynchronize()); }
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    printf("two waves per SIMD, %-18s: MFMA wave %9lld ticks (%.1f per MFMA)   VALU wave %9lld ticks (%.2f per v_pk_fma_f32)\n", names[mode], h[0], (double)h[0] / (NM * 8), h[4], (double)h[4] / (NV * 8));
  }
  auto run = [&](auto kern, int K) {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, sink, NM, 1.5f); CK(hipDeviceSynchronize()); }
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    printf("one wave per SIMD, %d packed FMAs after every MFMA: %9lld ticks = %.1f per MFMA (+%d VALU)\n", K, h[0], (double)h[0] / (NM * 8), K);
  };
  run(interleave_kernel<0>, 0); run(interleave_kernel<2>, 2); run(interleave_kernel<4>, 4); run(interleave_kernel<6>, 6); run(interleave_kernel<8>, 8); run(interleave_kernel<12>, 12);
  return 0;
}
