// Exhaustive search for a chunk swizzle of the attention tiles ([rows][64] bf16, 128-B rows, chunk c of row r at c ^ s((r >> 1) & 7)) that is
// conflict-free for BOTH access patterns: ds_read_b128 row fragments (16 rows x 4 chunks) and ds_read_b64_tr_b16 transposing reads.
// All 8! bijections s are timed on the hardware (8 waves, back-to-back reads); prints those within 3 % of the linear baselines.
//   hipcc --offload-arch=gfx950 -O3 -o tools/lab/lds_swizzle_search tools/lab/lds_swizzle_search.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

__device__ __forceinline__ int sw(unsigned perm, int r) { return (perm >> (3 * ((r >> 1) & 7))) & 7; }

template <bool TR>
__global__ __launch_bounds__(512) void k(long long* out, unsigned* sink, int iters, unsigned perm, int linear) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 40960 / 4; i += blockDim.x) ((unsigned*)lds)[i] = i * 2654435761u;
  __syncthreads();
  constexpr int NR = 16;
  unsigned a[NR];
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const int g = lane >> 4, i16 = lane & 15;
#pragma unroll
  for (int n = 0; n < NR; ++n) {
    unsigned off;
    if (!TR) {
      const int tile = n % 13, ks = (n / 13) & 1;
      const int r = tile * 16 + i16, c = ks * 4 + g;
      off = linear ? (unsigned)(lane * 16 + n * 1024) : (unsigned)(r * 128 + ((c ^ sw(perm, r)) << 4));
    } else {
      const int dt = n & 3, h = (n >> 2) & 1, t = (n >> 3) % 6;
      const int r = 4 * g + (i16 >> 2), c = dt * 2 + ((i16 & 3) >> 1);
      off = linear ? (unsigned)(lane * 8 + n * 512) : (unsigned)(r * 128 + ((c ^ sw(perm, r)) << 4) + (i16 & 1) * 8 + t * 4096 + h * 2048);
    }
    a[n] = base + off;
  }
  unsigned acc = 0;
  __builtin_amdgcn_s_barrier();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int n = 0; n < NR; ++n) {
      if (TR) { u32x2 v; asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(a[n])); asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(v)); acc ^= v[0]; }
      else { u32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a[n])); asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(v)); acc ^= v[0]; }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[wave] = t1 - t0;
  sink[tid] = acc;
}

template <bool TR>
double timeit(long long* d_out, unsigned* d_sink, unsigned perm, int linear) {
  const int iters = 100, waves = 8;
  hipLaunchKernelGGL((k<TR>), dim3(1), dim3(64 * waves), 65536, 0, d_out, d_sink, iters, perm, linear);
  CK(hipDeviceSynchronize());
  long long h[8];
  CK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
  long long mx = 0;
  for (int w = 0; w < waves; ++w) mx = std::max(mx, h[w]);
  return (double)mx / ((double)iters * 16 * waves);
}

int main() {
  long long* d_out; unsigned* d_sink;
  CK(hipMalloc(&d_out, 16 * sizeof(long long))); CK(hipMalloc(&d_sink, 1024 * sizeof(unsigned)));
  CK(hipFuncSetAttribute((const void*)k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CK(hipFuncSetAttribute((const void*)k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  timeit<false>(d_out, d_sink, 0, 1);
  const double b0 = timeit<false>(d_out, d_sink, 0, 1), t0 = timeit<true>(d_out, d_sink, 0, 1);
  printf("baselines: b128 %.2f, tr %.2f cycles per wave instruction\n", b0, t0);
  int p[8] = {0, 1, 2, 3, 4, 5, 6, 7};
  int good = 0, n = 0;
  std::vector<int> hist_b(40, 0), hist_t(40, 0);
  do {
    unsigned perm = 0;
    for (int i = 0; i < 8; ++i) perm |= (unsigned)p[i] << (3 * i);
    const double b = timeit<false>(d_out, d_sink, perm, 0);
    hist_b[std::min(39, (int)(b / b0 * 10))]++;
    ++n;
    if (b > b0 * 1.04) continue;
    const double t = timeit<true>(d_out, d_sink, perm, 0);
    hist_t[std::min(39, (int)(t / t0 * 10))]++;
    if (t <= t0 * 1.04) {
      if (good < 40) printf("GOOD s = {%d,%d,%d,%d,%d,%d,%d,%d}: b128 %.2f tr %.2f\n", p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], b, t);
      ++good;
    }
  } while (std::next_permutation(p, p + 8));
  printf("%d permutations, %d conflict-free for both\n", n, good);
  printf("b128 time / baseline histogram (x0.1):"); for (int i = 8; i < 40; ++i) if (hist_b[i]) printf(" %d:%d", i, hist_b[i]); printf("\n");
  printf("tr time / baseline histogram over the b128-clean ones (x0.1):"); for (int i = 8; i < 40; ++i) if (hist_t[i]) printf(" %d:%d", i, hist_t[i]); printf("\n");
  return 0;
}
