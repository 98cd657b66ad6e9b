// Feed-rate lab for the next GEMM main loop: how many cycles does one 64-deep K step of a 256x256 tile take when the
// matrix pipe is fed by  (a) 8 waves of 128x64 (two per SIMD: today's 8-phase shape)  or  (b) 4 waves of 128x128 (one per SIMD,
// a third fewer fragment reads), with the real instruction mix -- MFMA 32x32x16 bf16 on fragments that really come from LDS
// (ds_read_b128, conflict-free), LDS-DMA refills of 64 KB per K step (buffer_load ... lds, L2-resident source), one barrier per K
// step -- and with parts of the mix removed?  Prints cycles per K step (s_memtime of wave 0) and the equivalent PFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_feed_lab mfma_feed_lab.hip && ./mfma_feed_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// NW waves; wave tile = TM x TN 32x32 blocks; per 16-deep k-step a wave reads TM + TN fragments and issues TM*TN MFMAs.
// READS: fragments come from LDS (else stay in registers).  DMA: the wave issues its share of the 64 one-KB LDS-DMA per K step,
// spread between the MFMAs of the first three k-steps.  BAR: one s_barrier (+ counted vmcnt) per K step.
template <int NW, int TM, int TN, bool READS, bool DMA, int BAR, bool ILV = false>      // BAR: barriers per K step (0, 1, 4 or 8); ILV: one fragment read after every other MFMA instead of a burst
__global__ __launch_bounds__(64 * NW) void feed_kernel(const char* src, float* out, long long* cyc, int ksteps, int random_bits) {
  extern __shared__ __attribute__((aligned(16))) char lds[];        // 128 KB: two 64 KB stages
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
  constexpr int NDMA = 64 / NW;                                      // LDS-DMA instructions per wave per K step
  constexpr int PER = (NDMA + 2) / 3;                                // per k-step over the first three
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // operand bits: zeros (RANDOM == 0) or pseudo-random bf16 in [-2, 2) -- the matrix pipe's power draw, and with it the clock the
  // chip sustains, depends on how many bits toggle
  for (int i = tid; i < 131072 / 4; i += 64 * NW) ((unsigned*)lds)[i] = random_bits ? ((i * 2654435761u) & 0x807f807fu) | 0x3f803f80u : 0u;
  __syncthreads();
  bf16x8 fa[2][TM], fb[2][TN];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[b][i] = *(const bf16x8*)(lds + (i * 64 + lane) * 16);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[b][j] = *(const bf16x8*)(lds + 32768 + (j * 64 + lane) * 16);
  }
  const unsigned lane_off = (unsigned)(((blockIdx.x * NW + wave) * 64 + lane) * 16);
  long long t0 = __builtin_readcyclecounter();
  for (int kt = 0; kt < ksteps; ++kt) {
    const char* stage = lds + (kt & 1) * 65536;
    char* fill = lds + ((kt + 1) & 1) * 65536;
    if (BAR) {
      if (DMA) { if (NDMA == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
    }
    int dma_done = 0;
    constexpr int WGN = NW == 8 ? 4 : 2;
    const int wm = wave / WGN, wn = wave % WGN;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      const char* s2 = (ks < 3) ? stage + (ks + 1) * 8192 : fill;
      if (READS && !ILV) {                                           // fragments of the next k-step (of the next stage's first at ks = 3)
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[nxt][i] = *(const bf16x8*)(s2 + (wm * TM + i) * 1024 + lane * 16);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[nxt][j] = *(const bf16x8*)(s2 + 32768 + (wn * TN + j) * 1024 + lane * 16);
      }
      __builtin_amdgcn_sched_barrier(0);
      int in_ks = 0;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
          if (READS && ILV) {                                        // spread: fragment (i*TN+j)/2 after every second MFMA
            constexpr int NF = TM + TN;
            const int m = i * TN + j;
            if (m % 2 == 1 && m / 2 < NF) {
              __builtin_amdgcn_sched_barrier(0);
              const int f = m / 2;
              if (f < TM) fa[nxt][f] = *(const bf16x8*)(s2 + (wm * TM + f) * 1024 + lane * 16);
              else fb[nxt][f - TM] = *(const bf16x8*)(s2 + 32768 + (wn * TN + (f - TM)) * 1024 + lane * 16);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          if (BAR >= 8 && i * TN + j == TM * TN / 2 - 1) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
          constexpr int STRIDE = (TM * TN) / PER > 0 ? (TM * TN) / PER : 1;
          if (DMA && ks < 3 && (i * TN + j) % STRIDE == 0 && in_ks < PER && dma_done < NDMA) {     // spread between the MFMAs
            __builtin_amdgcn_sched_barrier(0);
            const unsigned off = lane_off + (unsigned)((kt * 64 + dma_done) & 1023) * 4096u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDSP(fill + (wave * NDMA + dma_done) * 1024), 16, off, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            ++dma_done; ++in_ks;
          }
        }
      }
      if (READS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (BAR >= 4 && ks < 3) asm volatile("s_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 123.456f) out[tid] = s;
  if (blockIdx.x == 0 && tid == 0) cyc[0] = t1 - t0;
}

template <int NW, int TM, int TN, bool READS, bool DMA, int BAR, bool ILV = false>
void run(const char* name, const char* src, float* out, long long* cyc, int random_bits) {
  const int ksteps = 2000, nblk = 256;
  CK(hipFuncSetAttribute((const void*)feed_kernel<NW, TM, TN, READS, DMA, BAR, ILV>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((feed_kernel<NW, TM, TN, READS, DMA, BAR, ILV>), dim3(nblk), dim3(64 * NW), 131072, 0, src, out, cyc, ksteps, random_bits);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((feed_kernel<NW, TM, TN, READS, DMA, BAR, ILV>), dim3(nblk), dim3(64 * NW), 131072, 0, src, out, cyc, ksteps, random_bits);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  const double flop = (double)nblk * ksteps * 2.0 * 256 * 256 * 64;
  printf("%-58s %7.0f counter ticks / K step   %8.1f ns / K step   %6.3f PFLOP/s\n", name, (double)c / ksteps, ms * 1e6 / ksteps, flop / (ms * 1e-3) / 1e15);
}

int main() {
  char* src; CK(hipMalloc(&src, 64 << 20));
  unsigned* host = (unsigned*)malloc(64 << 20);
  float* out; CK(hipMalloc(&out, 4096 * 4));
  long long* cyc; CK(hipMalloc(&cyc, 64));
  printf("one workgroup per CU, 256x256x64 per K step (2048 cycles of MFMA per SIMD at the matrix pipe's rate)\n");
  for (int random_bits = 0; random_bits < 2; ++random_bits) {
  printf("---- operand bits: %s\n", random_bits ? "pseudo-random bf16 in [-2, 2)" : "all zero");
  for (size_t i = 0; i < (64u << 20) / 4; ++i) host[i] = random_bits ? (((unsigned)i * 2654435761u) & 0x807f807fu) | 0x3f803f80u : 0u;
  CK(hipMemcpy(src, host, 64 << 20, hipMemcpyHostToDevice));
  run<8, 4, 2, false, false, 0>("8 waves 128x64: MFMA only", src, out, cyc, random_bits);
  run<8, 4, 2, true, false, 0>("8 waves 128x64: + fragment reads (24 / wave)", src, out, cyc, random_bits);
  run<8, 4, 2, true, false, 1>("8 waves 128x64: + reads + barrier", src, out, cyc, random_bits);
  run<8, 4, 2, false, true, 1>("8 waves 128x64: + LDS-DMA (8 / wave) + barrier", src, out, cyc, random_bits);
  run<8, 4, 2, true, true, 1>("8 waves 128x64: + reads + LDS-DMA + barrier", src, out, cyc, random_bits);
  run<8, 4, 2, true, true, 4>("8 waves 128x64: + reads + LDS-DMA + 4 barriers", src, out, cyc, random_bits);
  run<8, 4, 2, true, true, 8>("8 waves 128x64: + reads + LDS-DMA + 8 barriers", src, out, cyc, random_bits);
  run<4, 4, 4, false, false, 0>("4 waves 128x128: MFMA only", src, out, cyc, random_bits);
  run<4, 4, 4, true, false, 0>("4 waves 128x128: + fragment reads (32 / wave)", src, out, cyc, random_bits);
  run<4, 4, 4, true, false, 1>("4 waves 128x128: + reads + barrier", src, out, cyc, random_bits);
  run<4, 4, 4, false, true, 1>("4 waves 128x128: + LDS-DMA (16 / wave) + barrier", src, out, cyc, random_bits);
  run<4, 4, 4, true, true, 1>("4 waves 128x128: + reads + LDS-DMA + barrier", src, out, cyc, random_bits);
  run<4, 4, 4, true, false, 0, true>("4 waves 128x128: + reads spread between the MFMAs", src, out, cyc, random_bits);
  run<4, 4, 4, true, true, 1, true>("4 waves 128x128: + spread reads + LDS-DMA + barrier", src, out, cyc, random_bits);
  }
  return 0;
}
