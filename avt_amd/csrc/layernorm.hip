// Fused LayerNorm forward / backward for bf16 activations (fp32 statistics), gfx950.
// One 64-lane wave per row; a row of D = 8*64*V bf16 is held in registers (D in {64..4096}, D % 8 == 0).
// HBM-bound: every element is read once and written once, 16 B per lane per access.
//
// forward : y = (x - mean) * rstd * gamma + beta ; saves mean, rstd (fp32) -- timm LayerNorm eps 1e-6,
//           HF GPT-2 eps 1e-5 (eps is an argument).  Input rows may be strided (CLS-row select before the
//           final ViT norm).
// backward: dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) [+ dres],  g = dy * gamma
//           dgamma += sum_rows dy * xhat ; dbeta += sum_rows dy ; optional colsum(dx_out) (the bias gradient
//           of the Linear that produced the residual stream) -- all accumulated with fp32 atomics after a
//           per-block reduction.
#include <cstdlib>
#include "common.hpp"
#include "../../include/avt_hip.h"

namespace {

constexpr int MAXV = 8;   // up to 8 x 16-B chunks per lane -> D <= 4096

template <int V>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ y, int ldy,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     int rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nchunk = D >> 3;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const bf16_t* xr = x + (size_t)row * ldx;
    float v[V][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      int c = lane + i * 64;
      if (c < nchunk) {
        u32x4_t w = AVT_LDG_NT((const u32x4_t*)(xr + c * 8));
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[i][2 * e] = bflo(w[e]); v[i][2 * e + 1] = bfhi(w[e]); }
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[i][e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
      }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      int c = lane + i * 64;
      if (c < nchunk) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { float d = v[i][e] - mean; q += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0) { if (mean_out) mean_out[row] = mean; if (rstd_out) rstd_out[row] = rstd; }
    bf16_t* yr = y + (size_t)row * ldy;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      int c = lane + i * 64;
      if (c < nchunk) {
        f32x4_t g0 = *(const f32x4_t*)(gamma + c * 8), g1 = *(const f32x4_t*)(gamma + c * 8 + 4);
        f32x4_t b0 = *(const f32x4_t*)(beta + c * 8), b1 = *(const f32x4_t*)(beta + c * 8 + 4);
        float o[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (v[i][e] - mean) * rstd * g0[e] + b0[e];
          o[4 + e] = (v[i][4 + e] - mean) * rstd * g1[e] + b1[e];
        }
        u32x4_t w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = pack2bf(o[2 * e], o[2 * e + 1]);
        AVT_STG_NT((u32x4_t*)(yr + c * 8), w);
      }
    }
  }
}

template <int V, int MINW = 1, bool PF = true>
__global__ __launch_bounds__(256, MINW) void ln_bwd_kernel(const bf16_t* __restrict__ dy, int lddy, const bf16_t* __restrict__ x, int ldx,
                                                     const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                     const float* __restrict__ gamma, const bf16_t* __restrict__ dres, int lddres,
                                                     bf16_t* __restrict__ dx, int lddx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, float* __restrict__ colsum, int rows, int D,
                                                     float* __restrict__ part) {
  __shared__ float red[3][4][64 * 8];    // [quantity][wave][lane*8+e] scratch for one chunk column at a time
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nchunk = D >> 3;
  float ag[V][8], ab[V][8], ac[V][8];
#pragma unroll
  for (int i = 0; i < V; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { ag[i][e] = 0.f; ab[i][e] = 0.f; ac[i][e] = 0.f; }
  float gam[V][8];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    int c = lane + i * 64;
    if (c < nchunk) {
      f32x4_t g0 = *(const f32x4_t*)(gamma + c * 8), g1 = *(const f32x4_t*)(gamma + c * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { gam[i][e] = g0[e]; gam[i][4 + e] = g1[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) gam[i][e] = 0.f;
    }
  }
  // software pipeline over this wave's rows: the 16-byte loads of row r+1 (x, dy, residual gradient, statistics) are in flight
  // while row r is reduced and written, so every wave keeps two rows of requests outstanding
  const int rstep = gridDim.x * 4;
  u32x4_t nx[V], nd[V], nr[V];
  float nmean = 0.f, nrstd = 0.f;
  auto fetch = [&](int row) __attribute__((always_inline)) {
    const bool ok = row < rows;
    const bf16_t* xr = x + (size_t)row * ldx;
    const bf16_t* dyr = dy + (size_t)row * lddy;
    const bf16_t* rr = dres + (size_t)row * lddres;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = lane + i * 64;
      nx[i] = (u32x4_t){0u, 0u, 0u, 0u}; nd[i] = nx[i]; nr[i] = nx[i];
      if (ok && c < nchunk) {
        nx[i] = AVT_LDG_NT((const u32x4_t*)(xr + c * 8));
        nd[i] = AVT_LDG_NT((const u32x4_t*)(dyr + c * 8));
        if (dres) nr[i] = AVT_LDG_NT((const u32x4_t*)(rr + c * 8));
      }
    }
    nmean = ok ? mean_in[row] : 0.f; nrstd = ok ? rstd_in[row] : 0.f;
  };
  int row = blockIdx.x * 4 + wave;
  if (PF) fetch(row);
  for (; row < rows; row += rstep) {
    if (!PF) fetch(row);
    u32x4_t cx[V], cd[V], cr[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { cx[i] = nx[i]; cd[i] = nd[i]; cr[i] = nr[i]; }
    const float mean = nmean, rstd = nrstd;
    if (PF) fetch(row + rstep);
    float xh[V][8], g[V][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      int c = lane + i * 64;
      if (c < nchunk) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x0 = (bflo(cx[i][e]) - mean) * rstd, x1 = (bfhi(cx[i][e]) - mean) * rstd;
          float d0 = bflo(cd[i][e]), d1 = bfhi(cd[i][e]);
          xh[i][2 * e] = x0; xh[i][2 * e + 1] = x1;
          ag[i][2 * e] += d0 * x0; ag[i][2 * e + 1] += d1 * x1;
          ab[i][2 * e] += d0; ab[i][2 * e + 1] += d1;
          g[i][2 * e] = d0 * gam[i][2 * e]; g[i][2 * e + 1] = d1 * gam[i][2 * e + 1];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1 += g[i][e]; s2 += g[i][e] * xh[i][e]; }
      }
    }
    const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
    bf16_t* dxr = dx + (size_t)row * lddx;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      int c = lane + i * 64;
      if (c < nchunk) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = rstd * (g[i][e] - m1 - xh[i][e] * m2);
        if (dres) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { o[2 * e] += bflo(cr[i][e]); o[2 * e + 1] += bfhi(cr[i][e]); }
        }
        u32x4_t w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = pack2bf(o[2 * e], o[2 * e + 1]);
        AVT_STG_NT((u32x4_t*)(dxr + c * 8), w);
        if (colsum) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { ac[i][2 * e] += bflo(w[e]); ac[i][2 * e + 1] += bfhi(w[e]); }   // sum what the consumer GEMM will see
        }
      }
    }
  }
  // block reduction of the per-lane column partials, one chunk column (64 lanes x 8) at a time
#pragma unroll
  for (int i = 0; i < V; ++i) {
    int c = lane + i * 64;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[0][wave][lane * 8 + e] = ag[i][e];
      red[1][wave][lane * 8 + e] = ab[i][e];
      red[2][wave][lane * 8 + e] = ac[i][e];
    }
    __syncthreads();
    // 256 threads fold 3 x 512 values: thread t handles elements t and t+256 of each quantity
    for (int idx = threadIdx.x; idx < 512; idx += 256) {
      int l = idx >> 3, e = idx & 7;
      int cc = l + i * 64;
      if (cc < nchunk) {
        float sg = red[0][0][idx] + red[0][1][idx] + red[0][2][idx] + red[0][3][idx];
        float sb = red[1][0][idx] + red[1][1][idx] + red[1][2][idx] + red[1][3][idx];
        float sc = colsum ? red[2][0][idx] + red[2][1][idx] + red[2][2][idx] + red[2][3][idx] : 0.f;
        if (part) {                                    // one partial vector per workgroup and quantity, merged in fixed order afterwards
          const size_t qs = (size_t)gridDim.x * D;
          float* dst = part + (size_t)blockIdx.x * D + cc * 8 + e;
          dst[0] = sg; dst[qs] = sb;
          if (colsum) dst[2 * qs] = sc;
        } else {
          if (dgamma) unsafeAtomicAdd(&dgamma[cc * 8 + e], sg);
          if (dbeta) unsafeAtomicAdd(&dbeta[cc * 8 + e], sb);
          if (colsum) unsafeAtomicAdd(&colsum[cc * 8 + e], sc);
        }
      }
    }
    (void)c;
  }
}


// ---- D = 768 (ViT-B): 96 16-byte chunks per row would leave half of every second load instruction idle with one row per
// wave, so a wave takes TWO rows: 192 chunks = 3 full wave loads.  Chunk k = lane + 64 i belongs to row k / 96, column chunk
// k % 96 (i = 0: row 0; i = 1: lanes 0-31 row 0, lanes 32-63 row 1; i = 2: row 1).
constexpr int CPR96 = 96;
__device__ __forceinline__ void split2(float s, bool r1, float& a0, float& a1) { a0 += r1 ? 0.f : s; a1 += r1 ? s : 0.f; }

__global__ __launch_bounds__(256) void ln_fwd2_kernel(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, bf16_t* __restrict__ y, int ldy,
                                                      float* __restrict__ mean_out, float* __restrict__ rstd_out, int rows, float eps) {
  constexpr int D = CPR96 * 8;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  int cc[3]; bool r1[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) { const int k = lane + 64 * i; r1[i] = k >= CPR96; cc[i] = r1[i] ? k - CPR96 : k; }
  float gam[3][8], bet[3][8];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    f32x4_t g0 = *(const f32x4_t*)(gamma + cc[i] * 8), g1 = *(const f32x4_t*)(gamma + cc[i] * 8 + 4);
    f32x4_t b0 = *(const f32x4_t*)(beta + cc[i] * 8), b1 = *(const f32x4_t*)(beta + cc[i] * 8 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { gam[i][e] = g0[e]; gam[i][4 + e] = g1[e]; bet[i][e] = b0[e]; bet[i][4 + e] = b1[e]; }
  }
  const int npair = (rows + 1) >> 1;
  for (int pair = blockIdx.x * 4 + wave; pair < npair; pair += gridDim.x * 4) {
    const int row0 = pair * 2;
    float v[3][8];
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int row = row0 + (r1[i] ? 1 : 0);
      u32x4_t w = (u32x4_t){0u, 0u, 0u, 0u};
      if (row < rows) w = AVT_LDG_NT((const u32x4_t*)(x + (size_t)row * ldx + cc[i] * 8));
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[i][2 * e] = bflo(w[e]); v[i][2 * e + 1] = bfhi(w[e]); s += v[i][2 * e] + v[i][2 * e + 1]; }
      split2(s, r1[i], s0, s1);
    }
    const float m0 = wave_sum(s0) * (1.f / D), m1 = wave_sum(s1) * (1.f / D);
    float q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float m = r1[i] ? m1 : m0;
      float q = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - m; q += d * d; }
      split2(q, r1[i], q0, q1);
    }
    const float rs0 = rsqrtf(wave_sum(q0) * (1.f / D) + eps), rs1 = rsqrtf(wave_sum(q1) * (1.f / D) + eps);
    if (lane == 0) {
      if (mean_out) { mean_out[row0] = m0; if (row0 + 1 < rows) mean_out[row0 + 1] = m1; }
      if (rstd_out) { rstd_out[row0] = rs0; if (row0 + 1 < rows) rstd_out[row0 + 1] = rs1; }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int row = row0 + (r1[i] ? 1 : 0);
      const float m = r1[i] ? m1 : m0, rs = r1[i] ? rs1 : rs0;
      u32x4_t w;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        w[e] = pack2bf((v[i][2 * e] - m) * rs * gam[i][2 * e] + bet[i][2 * e], (v[i][2 * e + 1] - m) * rs * gam[i][2 * e + 1] + bet[i][2 * e + 1]);
      if (row < rows) AVT_STG_NT((u32x4_t*)(y + (size_t)row * ldy + cc[i] * 8), w);
    }
  }
}


constexpr int LN_BWD_MINW = 3;      // three waves per SIMD without the one-row-ahead prefetch: 364 -> 300 us per call (round 2)
int pick_v(int D) { int nchunk = D / 8; return (nchunk + 63) / 64; }

}  // namespace

extern "C" int avt_layernorm_fwd(const void* x, int ldx, const float* gamma, const float* beta, void* y, int ldy,
                                 float* mean, float* rstd, int rows, int D, float eps, void* stream) {
  AVT_CHECK(x && gamma && beta && y, "avt_layernorm_fwd: null argument");
  AVT_CHECK(rows > 0 && D > 0 && D % 8 == 0 && D <= 64 * 8 * MAXV, "avt_layernorm_fwd: D must be a multiple of 8 and <= 4096 (D=%d)", D);
  AVT_CHECK(ldx % 8 == 0 && ldy % 8 == 0 && aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta),
            "avt_layernorm_fwd: 16-byte alignment required");
  int grid = (rows + 3) / 4; if (grid > 4096) grid = 4096;
  hipStream_t s = (hipStream_t)stream;
  if (D == 768 && rows >= 64) {           // two rows per wave: full 16-byte lanes (see ln_fwd2_kernel)
    int g2 = ((rows + 1) / 2 + 3) / 4; if (g2 > 4096) g2 = 4096;
    hipLaunchKernelGGL(ln_fwd2_kernel, dim3(g2), dim3(256), 0, s, (const bf16_t*)x, ldx, gamma, beta, (bf16_t*)y, ldy, mean, rstd, rows, eps);
    AVT_LAUNCH_CHECK();
    return 0;
  }
#define LN_FWD(V) hipLaunchKernelGGL((ln_fwd_kernel<V>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, ldx, gamma, beta, (bf16_t*)y, ldy, mean, rstd, rows, D, eps)
  switch (pick_v(D)) {
    case 1: LN_FWD(1); break; case 2: LN_FWD(2); break; case 3: LN_FWD(3); break; case 4: LN_FWD(4); break;
    case 5: LN_FWD(5); break; case 6: LN_FWD(6); break; case 7: LN_FWD(7); break; default: LN_FWD(8); break;
  }
#undef LN_FWD
  AVT_LAUNCH_CHECK();
  return 0;
}

extern "C" int avt_layernorm_bwd(const void* dy, int lddy, const void* x, int ldx, const float* mean, const float* rstd,
                                 const float* gamma, const void* dres, int lddres, void* dx, int lddx,
                                 float* dgamma, float* dbeta, float* colsum, int rows, int D,
                                 float* part, size_t part_bytes, void* stream) {
  AVT_CHECK(dy && x && mean && rstd && gamma && dx, "avt_layernorm_bwd: null argument");
  AVT_CHECK(rows > 0 && D > 0 && D % 8 == 0 && D <= 64 * 8 * MAXV, "avt_layernorm_bwd: D must be a multiple of 8 and <= 4096 (D=%d)", D);
  AVT_CHECK(lddy % 8 == 0 && ldx % 8 == 0 && lddx % 8 == 0 && (!dres || lddres % 8 == 0), "avt_layernorm_bwd: leading dims must be multiples of 8");
  AVT_CHECK(aligned16(dy) && aligned16(x) && aligned16(dx) && aligned16(gamma) && (!dres || aligned16(dres)), "avt_layernorm_bwd: 16-byte alignment required");
  int grid = (rows + 3) / 4; if (grid > 512) grid = 512;          // 2 blocks of 4 waves per CU (register-limited), persistent over rows
  hipStream_t s = (hipStream_t)stream;
  // (a two-rows-per-wave variant like ln_fwd2_kernel was measured for backward: 349-407 us against this kernel's 346 us at
  //  252160 x 768 -- three input streams per row leave no registers for it to win; not kept)
  if (pick_v(D) == 2) {
    // D = 768 / 1024: three waves per SIMD without the software prefetch (150 VGPRs) beat two waves with it (175 VGPRs):
    // 300 vs 364 us at 252160 x 768 -- more rows in flight per CU than the one-row-ahead pipeline gave
    int g = (rows + 3) / 4; if (g > 768) g = 768;
    grid = g;
    AVT_CHECK(!part || part_bytes >= (size_t)3 * grid * D * 4, "avt_layernorm_bwd: partials workspace too small");
    hipLaunchKernelGGL((ln_bwd_kernel<2, LN_BWD_MINW, false>), dim3(g), dim3(256), 0, s, (const bf16_t*)dy, lddy, (const bf16_t*)x, ldx, mean, rstd, gamma,
                       (const bf16_t*)dres, lddres, (bf16_t*)dx, lddx, dgamma, dbeta, colsum, rows, D, part);
    AVT_LAUNCH_CHECK();
  } else {
    AVT_CHECK(!part || part_bytes >= (size_t)3 * grid * D * 4, "avt_layernorm_bwd: partials workspace too small");
#define LN_BWD(V) hipLaunchKernelGGL((ln_bwd_kernel<V>), dim3(grid), dim3(256), 0, s, (const bf16_t*)dy, lddy, (const bf16_t*)x, ldx, mean, rstd, gamma, (const bf16_t*)dres, lddres, (bf16_t*)dx, lddx, dgamma, dbeta, colsum, rows, D, part)
    switch (pick_v(D)) {
      case 1: LN_BWD(1); break; case 3: LN_BWD(3); break; case 4: LN_BWD(4); break;
      case 5: LN_BWD(5); break; case 6: LN_BWD(6); break; case 7: LN_BWD(7); break; default: LN_BWD(8); break;
    }
#undef LN_BWD
    AVT_LAUNCH_CHECK();
  }
  if (part) {
    float* outs[3] = {dgamma, dbeta, colsum};
    return avt_reduce_partials(part, grid, D, outs, colsum ? 3 : 2, s);
  }
  return 0;
}
extern "C" size_t avt_layernorm_bwd_workspace_bytes(int rows, int D) {
  (void)rows;
  return (size_t)3 * 768 * (size_t)D * 4;                          // at most 768 workgroups x {dgamma, dbeta, colsum}
}
