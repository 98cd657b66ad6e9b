// Probe: does gfx950 execute scalar memory atomics (s_atomic_add ... glc), are tickets unique per XCD counter, and what do a scalar
// and a vector (global_atomic_add, device scope) ticket cost in cycles?   hipcc --offload-arch=gfx950 -O2 satomic_probe.hip -o satomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void probe(int* counters, int* out, long long* cyc, int rounds, int use_vector) {
  uint32_t xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7;
  int* c = counters + xcc * 32;          // one 128-byte line per XCD
  long long tot = 0;
  for (int r = 0; r < rounds; ++r) {
    int t;
    long long t0 = __builtin_readcyclecounter();
    if (use_vector) {
      t = 0;
      if (threadIdx.x == 0) t = __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      t = __builtin_amdgcn_readfirstlane(t);
    } else {
      int one = 1;
      asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(one) : "s"(c) : "memory");
      t = one;
    }
    long long t1 = __builtin_readcyclecounter();
    tot += t1 - t0;
    if (threadIdx.x == 0) out[(blockIdx.x * rounds + r) * 2] = (int)xcc, out[(blockIdx.x * rounds + r) * 2 + 1] = t;
  }
  if (threadIdx.x == 0) cyc[blockIdx.x] = tot / rounds;
}
int main() {
  const int G = 256, R = 16;
  int *counters, *out; long long* cyc;
  hipMalloc(&counters, 8 * 32 * 4); hipMalloc(&out, G * R * 2 * 4); hipMalloc(&cyc, G * 8);
  for (int mode = 0; mode < 2; ++mode) {
    hipMemset(counters, 0, 8 * 32 * 4);
    hipLaunchKernelGGL(probe, dim3(G), dim3(64), 0, 0, counters, out, cyc, R, mode);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("mode %d: %s\n", mode, hipGetErrorString(e)); return 1; }
    std::vector<int> h(G * R * 2), hc(8 * 32); std::vector<long long> hcy(G);
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hc.data(), counters, hc.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hcy.data(), cyc, G * 8, hipMemcpyDeviceToHost);
    bool ok = true; long long s = 0;
    for (int x = 0; x < 8; ++x) {
      std::vector<int> t;
      for (int i = 0; i < G * R; ++i) if (h[2 * i] == x) t.push_back(h[2 * i + 1]);
      std::sort(t.begin(), t.end());
      for (size_t i = 0; i < t.size(); ++i) if (t[i] != (int)i) ok = false;
      if ((int)t.size() != hc[x * 32]) ok = false;
      printf("  xcd %d: %zu tickets, counter %d\n", x, t.size(), hc[x * 32]);
    }
    for (int b = 0; b < G; ++b) s += hcy[b];
    printf("%s atomics: tickets unique and dense per XCD: %s;  %lld cycles per ticket (mean over %d workgroups)\n", mode ? "vector" : "scalar", ok ? "yes" : "NO", s / G, G);
  }
  return 0;
}
