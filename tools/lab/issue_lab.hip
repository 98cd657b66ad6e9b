// How long does ISSUING a memory instruction block a wave?  (a) buffer_load ... lds (LDS-DMA), (b) buffer_load_dwordx4 to
// VGPRs, (c) ds_write_b128.  W waves per CU issue 8 instructions back-to-back; s_memtime around the issue block only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE>
__global__ __launch_bounds__(512) void issue_kernel(const char* base, long long* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
  long long total = 0;
  u32x4 keep = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    unsigned off0 = (unsigned)(((blockIdx.x * 8 + wave) * 64 + it * 7) % 4096) * 4096u + lane * 16;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    long long t0 = __builtin_readcyclecounter();
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDSP(lds + wave * 8192 + j * 1024), 16, off0 + j * 1024, 0, 0, 0);
    } else if (MODE == 1) {
      u32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off0 + j * 1024, 0, 0);
      long long t1 = __builtin_readcyclecounter();
      total += t1 - t0;
#pragma unroll
      for (int j = 0; j < 8; ++j) keep += v[j];
      continue;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) *(u32x4*)(lds + wave * 8192 + j * 1024 + lane * 16) = keep + j;
    }
    long long t1 = __builtin_readcyclecounter();
    total += t1 - t0;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (keep[0] == 0x1234567 && MODE != 1) out[1000000] = 1;
  if (MODE == 1 && keep[1] == 0x7654321) out[1000000] = keep[0];
  if (lane == 0) out[blockIdx.x * 8 + wave] = total;
}

int main() {
  char* buf; CK(hipMalloc(&buf, 4096 * 4096 + (1 << 20))); CK(hipMemset(buf, 1, 4096 * 4096));
  long long* out; CK(hipMalloc(&out, 8 * 2000000));
  const int iters = 200;
  for (int waves : {1, 4, 8}) {
    for (int mode = 0; mode < 3; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) { CK(hipFuncSetAttribute((const void*)issue_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536)); hipLaunchKernelGGL(issue_kernel<0>, dim3(256), dim3(64 * waves), 65536, 0, buf, out, iters); }
        if (mode == 1) hipLaunchKernelGGL(issue_kernel<1>, dim3(256), dim3(64 * waves), 65536, 0, buf, out, iters);
        if (mode == 2) { CK(hipFuncSetAttribute((const void*)issue_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536)); hipLaunchKernelGGL(issue_kernel<2>, dim3(256), dim3(64 * waves), 65536, 0, buf, out, iters); }
        CK(hipDeviceSynchronize());
      }
      long long h[8]; CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
      const char* names[] = {"buffer_load ... lds (LDS-DMA)", "buffer_load_dwordx4 -> VGPR", "ds_write_b128"};
      printf("%d wave(s)/CU  %-32s : %6.1f cycles of wave time per instruction (issue only)\n", waves, names[mode], (double)h[0] / iters / 8);
    }
  }
  return 0;
}
