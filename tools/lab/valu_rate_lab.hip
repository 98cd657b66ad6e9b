// Issue cost of the vector instructions a GEMM epilogue is made of, per wave64 instruction, with 1 and 2 waves per SIMD:
// v_pk_fma_f32 / v_fma_f32 / v_pk_mul_f32 / v_exp_f32 / v_cvt_pk_bf16_f32 / v_med3_f32 / v_pk_add_f32 (8 independent chains each).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

template <int OP>
__global__ __launch_bounds__(512) void rate_kernel(long long* out, float* sink, int n, float seed) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f32x2 v[8];
  for (int i = 0; i < 8; ++i) v[i] = (f32x2){seed + i + lane * 0.001f, seed - i * 0.01f};
  f32x2 c1 = {0.999f + seed * 1e-6f, 1.001f}, c2 = {1e-3f * seed, -1e-3f};
  asm volatile("" : "+v"(c1), "+v"(c2));                    // constants in VGPRs
  __builtin_amdgcn_s_barrier();
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) v[i] = v[i] * c1 + c2;                                                        // v_pk_fma_f32
      if (OP == 1) { v[i][0] = __builtin_fmaf(v[i][0], c1[0], c2[0]); }                          // v_fma_f32
      if (OP == 2) v[i] = v[i] * c1;                                                             // v_pk_mul_f32
      if (OP == 3) v[i][0] = __builtin_amdgcn_exp2f(v[i][0]);                                    // v_exp_f32
      if (OP == 4) { bf16x2 b = __builtin_convertvector(v[i], bf16x2); v[i][0] = __builtin_bit_cast(float, b); }   // v_cvt_pk_bf16_f32
      if (OP == 5) v[i][0] = __builtin_amdgcn_fmed3f(v[i][0], c2[1], c1[1]);                     // v_med3_f32
      if (OP == 6) v[i] = v[i] + c2;                                                             // v_pk_add_f32
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
  if (s == 123.456f) sink[tid] = s;
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
  long long* out; float* sink;
  CK(hipMalloc(&out, 8 * 4096)); CK(hipMalloc(&sink, 4 * 512));
  long long h[8];
  const int N = 4000;
  const char* names[] = {"v_pk_fma_f32", "v_fma_f32", "v_pk_mul_f32", "v_exp_f32", "v_cvt_pk_bf16_f32", "v_med3_f32", "v_pk_add_f32"};
  auto run = [&](auto kern, int op) {
    for (int waves : {4, 8}) {
      for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), 0, 0, out, sink, N, 1.5f); CK(hipDeviceSynchronize()); }
      CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
      printf("%-20s %d wave(s)/SIMD: %6.2f cycles of wave time per instruction -> %5.2f cycles of SIMD time per instruction\n", names[op], waves / 4,
             (double)h[0] / (N * 8), (double)h[0] / (N * 8) / (waves / 4));
    }
  };
  run(rate_kernel<0>, 0); run(rate_kernel<1>, 1); run(rate_kernel<2>, 2); run(rate_kernel<3>, 3); run(rate_kernel<4>, 4); run(rate_kernel<5>, 5); run(rate_kernel<6>, 6);
  return 0;
}
