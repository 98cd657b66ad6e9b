// LDS-DMA staging-rate lab: how fast can one workgroup per CU pull [256 x 64] bf16 tiles (128-B rows, row stride ld)
// from L2 / MALL / HBM into LDS?  Variants: bytes in flight per CU (depth), plain loads to registers for comparison.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// mode 0: LDS-DMA, wait for everything each iteration (one 64 KB stage in flight)
// mode 1: LDS-DMA, two stages in flight (counted vmcnt)
// mode 2: global_load_dwordx4 to registers (no LDS), 64 KB per iteration, wait each iteration
template <int MODE>
__global__ __launch_bounds__(512) void dma_kernel(const char* base, long ld_bytes, int rows_total, int ktiles, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
  // tile origin: block b owns rows [b*512 % rows_total ...): two 256-row panels (A-like and B-like)
  long row0 = ((long)blockIdx.x * 512) % rows_total;
  unsigned acc = 0;
  int kt = 0;
  auto issue = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {                      // 8 x 1 KB per wave = 64 KB per block per stage
      int r = j * 64 + wave * 8 + (lane >> 3);
      int c = (lane & 7) ^ ((r >> 1) & 7);
      long off = (row0 + r) % rows_total * ld_bytes + (long)kt * 128 + c * 16;
      if (MODE == 2) {
        u32x4 v = *(const u32x4*)(base + off);
        acc += v[0] ^ v[1] ^ v[2] ^ v[3];
      } else {
        char* dst = lds + buf * 65536 + (j * 64 + wave * 8) * 128;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDSP(dst), 16, (unsigned)off, 0, 0, 0);
      }
    }
    kt = (kt + 1) % ktiles;
  };
  if (MODE == 1) issue(0);
  for (int it = 0; it < iters; ++it) {
    issue((it + (MODE == 1)) & 1);
    if (MODE == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (MODE != 2) acc = ((unsigned*)lds)[tid];
  if (acc == 0x12345678) sink[0] = acc;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int nblk = 256, iters = 400;
  unsigned* sink; CK(hipMalloc(&sink, 64));
  struct Case { const char* name; int rows_total; int K; };
  // ld = K*2 bytes.  "L2": 4 MB total shared by all; "MALL": 128 MB; "HBM": 2 GB
  Case cases[] = {{"L2-resident  (rows 4096  x K 512)", 4096, 512}, {"MALL-resident(rows 16384 x K 4096)", 16384, 4096},
                  {"HBM-stream   (rows 131072 x K 8192)", 131072, 8192}, {"ViT-A-like   (rows 63040 x K 3072)", 63040, 3072}};
  for (auto& c : cases) {
    size_t bytes = (size_t)c.rows_total * c.K * 2;
    char* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 1, bytes));
    for (int mode = 0; mode < 3; ++mode) {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      auto launch = [&]() {
        if (mode == 0) { CK(hipFuncSetAttribute((const void*)dma_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); hipLaunchKernelGGL(dma_kernel<0>, dim3(nblk), dim3(512), 131072, 0, buf, (long)c.K * 2, c.rows_total, c.K / 64, iters, sink); }
        if (mode == 1) { CK(hipFuncSetAttribute((const void*)dma_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); hipLaunchKernelGGL(dma_kernel<1>, dim3(nblk), dim3(512), 131072, 0, buf, (long)c.K * 2, c.rows_total, c.K / 64, iters, sink); }
        if (mode == 2) { hipLaunchKernelGGL(dma_kernel<2>, dim3(nblk), dim3(512), 0, 0, buf, (long)c.K * 2, c.rows_total, c.K / 64, iters, sink); }
      };
      launch(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double total = (double)nblk * iters * 65536.0;
      printf("%s mode %d: %8.1f us  %6.2f TB/s aggregate  %6.1f B/clk/CU @2.1GHz  (%.0f ns per 64KB stage)\n", c.name, mode, ms * 1e3, total / (ms * 1e-3) / 1e12,
             total / (ms * 1e-3) / 256 / 2.1e9, ms * 1e6 / iters);
    }
    CK(hipFree(buf));
  }
  return 0;
}
