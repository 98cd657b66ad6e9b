// How fast can one workgroup (8 waves) push a 256x256 bf16 output tile (128 KB) to global memory, per CU, with every CU
// doing it at once?  Patterns: 0 = 8 rows x 128 B per wave instruction (the GEMM epilogue's: lane owns 16 B of a 64-column
// strip), 1 = 2 rows x 512 B per instruction (full tile rows), 2 = 1 KB fully contiguous per instruction (upper bound),
// 3 = pattern 0 with 8-B stores, 4 = pattern 0 but non-temporal (nt) stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

template <int MODE>
__global__ __launch_bounds__(512) void store_kernel(char* C, int ldc_bytes, int tiles_n, long long* out, int reps) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wn = wave & 3;
  long long total = 0;
  for (int rep = 0; rep < reps; ++rep) {
    const int t = blockIdx.x + rep * gridDim.x;
    char* tile = C + (size_t)(t / tiles_n) * 256 * ldc_bytes + (size_t)(t % tiles_n) * 512;
    u32x4 v = {(unsigned)tid, (unsigned)rep, 3u, 4u};
    __builtin_amdgcn_s_barrier();
    long long t0 = __builtin_readcyclecounter();
    if (MODE == 0 || MODE == 4) {
      // wave tile 128 rows x 64 cols (128 B): 16 instructions of 8 rows
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        char* p = tile + (size_t)(grp * 128 + j * 8 + (lane >> 3)) * ldc_bytes + wn * 128 + (lane & 7) * 16;
        if (MODE == 4) __builtin_nontemporal_store(v, (u32x4*)p); else *(u32x4*)p = v;
      }
    } else if (MODE == 1) {
      // the wave owns 32 full tile rows: 16 instructions of 2 rows x 512 B
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        char* p = tile + (size_t)(wave * 32 + j * 2 + (lane >> 5)) * ldc_bytes + (lane & 31) * 16;
        *(u32x4*)p = v;
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        char* p = C + ((size_t)t * 128 + wave * 16 + j) * 1024 + lane * 16;
        *(u32x4*)p = v;
      }
    } else if (MODE == 5) {
      // transposed-accumulator layout after v_permlane32_swap: 32 rows x 32 B per instruction
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        int i = j >> 2, cb = j & 3;                      // 32-row block, 16-column block of the 64-column wave strip
        char* p = tile + (size_t)(grp * 128 + i * 32 + (lane & 31)) * ldc_bytes + wn * 128 + cb * 32 + (lane >> 5) * 16;
        *(u32x4*)p = v;
      }
    } else if (MODE == 3) {
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        char* p = tile + (size_t)(grp * 128 + j * 4 + (lane >> 4)) * ldc_bytes + wn * 128 + (lane & 15) * 8;
        *(u32x2*)p = u32x2{v[0], v[1]};
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    total += __builtin_readcyclecounter() - t0;
  }
  if (tid == 0) out[blockIdx.x] = total;
}

int main() {
  const int M = 63040, N = 3072;                 // fc1 output at 32 clips
  const int tiles_m = M / 256, tiles_n = N / 256, reps = 8;
  char* C; long long* out;
  CK(hipMalloc(&C, (size_t)M * N * 2 + (1 << 20)));
  CK(hipMalloc(&out, 256 * 8));
  long long h[256];
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](auto kern, const char* name, int grid = 256) {
    for (int w = 0; w < 2; ++w) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, C, N * 2, tiles_n, out, reps);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    double avg = 0; for (int i = 0; i < grid; ++i) avg += h[i]; avg /= (double)grid * reps;
    printf("%-44s %8.0f cycles per 128 KB tile  = %5.1f B/clk/CU ; kernel %.1f us -> %.2f TB/s aggregate\n", name, avg, 131072.0 / avg, ms * 1e3,
           (double)grid * reps * 131072 / (ms * 1e-3) / 1e12);
  };
  (void)tiles_m;
  run(store_kernel<0>, "8 rows x 128 B per instr (16-B lanes)");
  run(store_kernel<1>, "2 rows x 512 B per instr");
  run(store_kernel<2>, "1 KB contiguous per instr");
  run(store_kernel<3>, "4 rows x 128 B per instr (8-B lanes)");
  run(store_kernel<4>, "8 rows x 128 B per instr, nontemporal");
  run(store_kernel<5>, "32 rows x 32 B per instr (permlane layout)");
  for (int g : {1, 32, 64}) { printf("grid %3d: ", g); run(store_kernel<5>, "32 rows x 32 B per instr", g); }
  for (int g : {1, 8, 32, 64, 128}) { printf("grid %3d: ", g); run(store_kernel<0>, "8 rows x 128 B per instr", g); }
  return 0;
}
