// Lab: occupy `nwg` CUs for `usec` microseconds (one workgroup per CU: 1024 threads and 128 KB of LDS keep anything else off it), to see how a
// GEMM launch behaves when the chip is not all its own -- the situation of the backward GEMMs while RCCL's kernels move gradient buckets.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/lab/cu_hog.hip -o tools/lab/libcu_hog.so
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(1024) void cu_hog_kernel(long long ticks, int* sink) {
  extern __shared__ char lds[];
  const long long t0 = wall_clock64();                     // 100 MHz
  long long n = 0;
  while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(64); ++n; }
  if (threadIdx.x == 0 && n == -1) { lds[0] = 1; sink[0] = lds[0]; }
}
extern "C" int cu_hog(int nwg, int usec, int* sink, void* stream) {
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)cu_hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024); attr = true; }
  hipLaunchKernelGGL(cu_hog_kernel, dim3(nwg), dim3(1024), 159 * 1024, (hipStream_t)stream, (long long)usec * 100, sink);
  return (int)hipGetLastError();
}
