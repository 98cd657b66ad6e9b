// LDS read patterns of the attention and GEMM kernels, timed back to back (s_memtime) with 4 and 8 waves per CU: cycles of LDS-pipe time per
// wave instruction.  A conflict-free ds_read_b128 moves 1 KB, a ds_read_b64_tr_b16 512 B; anything above the linear baseline is bank conflicts.
//   hipcc --offload-arch=gfx950 -O3 -o tools/lab/lds_conflict_lab tools/lab/lds_conflict_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

__device__ __forceinline__ int swz8(int r) { const int p = (r >> 1) & 7; return ((p & 3) << 1) | (p >> 2); }
__device__ __forceinline__ int swz_plain(int r) { return (r >> 1) & 7; }

// byte address of read number n (0..NR-1) of pattern P for this lane
template <int P>
__device__ __forceinline__ unsigned addr_of(int lane, int n) {
  const int g = lane >> 4, i16 = lane & 15;
  if (P == 0) return (unsigned)(lane * 16 + (n & 15) * 1024);                                   // b128, linear
  if (P == 1 || P == 6) {                                                                        // b128, attention frag_rm (swz8 | plain swizzle)
    const int tile = n % 13, ks = (n / 13) & 1;
    const int r = tile * 16 + i16, c = ks * 4 + g;
    return (unsigned)(r * 128 + ((c ^ (P == 1 ? swz8(r) : swz_plain(r))) << 4));
  }
  if (P == 2) {                                                                                  // b128, GEMM frag_kmajor<64>
    const int tile = n & 3, ks = (n >> 2) & 3;
    const int r = tile * 32 + (lane & 31), c = (ks * 2 + (lane >> 5)) ^ ((r >> 1) & 7);
    return (unsigned)(r * 128 + c * 16);
  }
  if (P == 3 || P == 7) {                                                                        // tr b64, attention (swz8 | plain): dt = n & 3, t, h
    const int dt = n & 3, h = (n >> 2) & 1, t = (n >> 3) % 6;
    const int r = 4 * g + (i16 >> 2), c = dt * 2 + ((i16 & 3) >> 1);
    return (unsigned)(r * 128 + ((c ^ (P == 3 ? swz8(r) : swz_plain(r))) << 4) + (i16 & 1) * 8 + t * 4096 + h * 2048);
  }
  if (P == 4) return (unsigned)(lane * 8 + (n & 15) * 512);                                      // tr b64, linear
  {                                                                                              // P == 5: tr b64, GEMM frag_kstrided<128> (256-B rows)
    const int tile = n & 3, ks = (n >> 2) & 1, h = (n >> 3) & 1;
    const int col = tile * 32 + (g & 1) * 16 + (i16 & 3) * 4;
    const int r = ks * 16 + (g >> 1) * 8 + (i16 >> 2) + h * 4;
    const int c = (col >> 3) ^ ((r & 3) << 2);
    return (unsigned)(r * 256 + c * 16 + (col & 7) * 2);
  }
}

template <int P, bool TR>
__global__ __launch_bounds__(1024) void k(long long* out, unsigned* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 40960 / 4; i += blockDim.x) ((unsigned*)lds)[i] = i * 2654435761u;
  __syncthreads();
  constexpr int NR = 16;
  unsigned a[NR];
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
#pragma unroll
  for (int n = 0; n < NR; ++n) a[n] = base + addr_of<P>(lane, n);
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
  if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
  sink[blockIdx.x * blockDim.x + tid] = acc;
}

template <int P, bool TR>
void run(const char* name, long long* d_out, unsigned* d_sink) {
  for (int waves : {4, 8, 13}) {
    const int iters = 2000;
    CK(hipFuncSetAttribute((const void*)k<P, TR>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipLaunchKernelGGL((k<P, TR>), dim3(1), dim3(64 * waves), 65536, 0, d_out, d_sink, iters);
    CK(hipDeviceSynchronize());
    long long h[16];
    CK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
    long long mx = 0;
    for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
    printf("%-52s %2d waves: %6.2f cycles of CU time per wave instruction\n", name, waves, (double)mx / ((double)iters * 16 * waves));
  }
}

int main() {
  long long* d_out; unsigned* d_sink;
  CK(hipMalloc(&d_out, 16 * sizeof(long long))); CK(hipMalloc(&d_sink, 1024 * sizeof(unsigned)));
  run<0, false>("ds_read_b128 linear (baseline)", d_out, d_sink);
  run<2, false>("ds_read_b128 GEMM k-major fragment (32 rows x 2)", d_out, d_sink);
  run<1, false>("ds_read_b128 attention fragment, swz8", d_out, d_sink);
  run<6, false>("ds_read_b128 attention fragment, (r>>1)&7", d_out, d_sink);
  run<4, true>("ds_read_b64_tr_b16 linear (baseline)", d_out, d_sink);
  run<5, true>("ds_read_b64_tr_b16 GEMM k-strided fragment", d_out, d_sink);
  run<3, true>("ds_read_b64_tr_b16 attention, swz8", d_out, d_sink);
  run<7, true>("ds_read_b64_tr_b16 attention, (r>>1)&7", d_out, d_sink);
  return 0;
}
