// Does a wave's own memory-instruction issue (LDS-DMA buffer_load ... lds, ds_read_b128) overlap with its own MFMAs?
// One wave per SIMD (256 threads): per iteration 8 MFMAs (32x32x16 bf16) plus D LDS-DMA instructions and R ds_read_b128,
// placed between the MFMAs.  Compare with the MFMA-only and memory-only loops: "sum" = serialised, "max" = overlapped.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int NWAVE, bool MFMA, int D, int R>
__global__ __launch_bounds__(64 * NWAVE) void k(const char* base, long long* out, float* sink, int n, float seed) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + lane * 0.01f + e); b[e] = (__bf16)(seed * 0.5f - lane * 0.02f + e); }
  u32x4 keep = {0, 0, 0, 0};
  unsigned off = (unsigned)((blockIdx.x * NWAVE + wave) * 65536 + lane * 16);
  __builtin_amdgcn_s_barrier();
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < n; ++it) {
    u32x4 rd[R > 0 ? R : 1];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MFMA) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
      if (i < D) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDSP(lds + wave * 16384 + i * 1024), 16, off + (unsigned)((it & 7) * 8192 + i * 1024), 0, 0, 0);
      if (i < R) rd[i] = *(const u32x4*)(lds + wave * 16384 + 8192 + i * 1024 + lane * 16);
    }
    if (D) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#pragma unroll
    for (int i = 0; i < R; ++i) keep += rd[i];
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  long long t1 = __builtin_readcyclecounter();
  float s = keep[0] * 1e-30f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
  if (s == 123.456f) sink[tid] = s;
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
  char* buf; CK(hipMalloc(&buf, (size_t)256 * 8 * 65536 + (1 << 20))); CK(hipMemset(buf, 1, (size_t)256 * 8 * 65536));
  long long* out; float* sink;
  CK(hipMalloc(&out, 8 * 4096)); CK(hipMalloc(&sink, 4 * 512));
  long long h[8];
  const int N = 2000;
  auto run = [&](auto kern, int nw, const char* what) {
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(256), dim3(64 * nw), 131072, 0, buf, out, sink, N, 1.5f); CK(hipDeviceSynchronize()); }
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    printf("%d wave(s)/SIMD  %-44s: %8.1f cycles per iteration\n", nw / 4, what, (double)h[0] / N);
  };
  run(k<4, true, 0, 0>, 4, "8 MFMA");
  run(k<4, false, 2, 0>, 4, "2 LDS-DMA");
  run(k<4, true, 2, 0>, 4, "8 MFMA + 2 LDS-DMA");
  run(k<4, false, 4, 0>, 4, "4 LDS-DMA");
  run(k<4, true, 4, 0>, 4, "8 MFMA + 4 LDS-DMA");
  run(k<4, false, 0, 6>, 4, "6 ds_read_b128");
  run(k<4, true, 0, 6>, 4, "8 MFMA + 6 ds_read_b128");
  run(k<4, true, 3, 6>, 4, "8 MFMA + 3 LDS-DMA + 6 ds_read_b128");
  run(k<8, true, 0, 0>, 8, "8 MFMA");
  run(k<8, false, 3, 6>, 8, "3 LDS-DMA + 6 ds_read_b128");
  run(k<8, true, 3, 6>, 8, "8 MFMA + 3 LDS-DMA + 6 ds_read_b128");
  return 0;
}
