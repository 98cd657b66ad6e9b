// LayerNorm folded into the GEMMs around it (ViT blocks: norm1 -> qkv, norm2 -> fc1; [timm] Block.forward via
// models/video_classification.py:224).  The normalised copy of the residual stream is never written:
//
//   forward   y = LN(x) W^T + b = rstd o (x G^T) - (rstd o mean) c^T + b',   G = gamma o W (bf16), c = G 1, b' = b + W beta
//             -- the GEMM reads the residual stream x itself; its epilogue applies two per-row scalars (avt_gemm_ln_bf16, "fold").
//             The rows' mean / rstd come from the epilogue of the GEMM that PRODUCED x (per-32-column partial sums, "stat_part")
//             and avt_ln_stats_finalize below.
//   backward  dY' = rstd o dY (written by the producer of dY: avt_gemm_ln_bf16 "scale", avt_vit_attn_bwd_scaled)
//             d xhat' = dY' G                      (the data-gradient GEMM on G^T, unchanged kernels)
//             dx     = d xhat' - mean_k(d xhat') - xhat o mean_k(d xhat' o xhat) [+ dres]           (avt_layernorm_bwd_folded: no gamma, no rstd factor)
//             T      = dY'^T x                     (the weight-gradient GEMM on x itself, unchanged kernels)
//             dG     = T - rowmean_k(T) 1^T        (= dY'^T (x - mean 1^T): sum_m dY'[m,n] mean[m] IS the row mean of T)
//             dW = gamma o dG + db beta^T,  dgamma = colsum(W o dG),  dbeta = W^T db,  db = colsum(dY)      (avt_ln_fold_wgrad)
//
// All four kernels here are small (statistics: 16 B per row; weights: 49.5 M elements per step for ViT-B) and HBM-bound.
#include <cstdlib>
#include "common.hpp"
#include "../../include/avt_hip.h"

int avt_reduce_partials(const float* part, int nslots, long n, float* const* outs, int nq, hipStream_t stream);   // elementwise.hip

namespace {

// ---- row statistics from the producer's partial sums ---------------------------------------------------------------------
// part [ceil(nslots / 2)][rows][2][2] = (sum, sum of squares) of each row over one 32-column slot (slot s at [s / 2][row][s % 2]).  The slots are added in slot order, in double
// (the partials are fp32 sums of 32 products; E[x^2] - mean^2 in double keeps the relative error of the variance at 6e-8 (mean^2 + var) / var).
__global__ __launch_bounds__(256) void ln_stats_finalize_kernel(const float* __restrict__ part, int nslots, int rows, float invD, float eps,
                                                                float* __restrict__ stat_fwd, float* __restrict__ stat_bwd) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  double s1 = 0.0, s2 = 0.0;
  for (int s = 0; s < nslots; ++s) {
    const f32x2_t v = *(const f32x2_t*)(part + ((size_t)(s >> 1) * rows + row) * 4 + (s & 1) * 2);
    s1 += (double)v[0]; s2 += (double)v[1];
  }
  const double mean = s1 * invD;
  double var = s2 * invD - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = rsqrtf((float)var + eps);
  if (stat_fwd) *(f32x2_t*)(stat_fwd + 2 * (size_t)row) = (f32x2_t){rstd, -(float)mean * rstd};
  if (stat_bwd) *(f32x2_t*)(stat_bwd + 2 * (size_t)row) = (f32x2_t){rstd, 1.0f / rstd};
}

// ---- folded weights: G = bf16(gamma o W), c = G 1 (over the ROUNDED values: it cancels the mean component the GEMM accumulates with them),
// b' = b + W beta.  One workgroup per output row n; fixed-order block reduction.
__global__ __launch_bounds__(256) void ln_fold_weights_kernel(const float* __restrict__ W, int ldw, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ bias,
                                                              bf16_t* __restrict__ G, int ldg, float* __restrict__ c, float* __restrict__ b2, int K) {
  __shared__ float red[2][256];
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* wr = W + (size_t)n * ldw;
  bf16_t* gr = G + (size_t)n * ldg;
  float sc = 0.f, sb = 0.f;
  for (int k = tid * 4; k < K; k += 1024) {
    const f32x4_t w = *(const f32x4_t*)(wr + k), g = *(const f32x4_t*)(gamma + k), b = *(const f32x4_t*)(beta + k);
    const uint32_t p0 = pack2bf(w[0] * g[0], w[1] * g[1]), p1 = pack2bf(w[2] * g[2], w[3] * g[3]);
    *(u32x2_t*)(gr + k) = (u32x2_t){p0, p1};
    sc += (bflo(p0) + bfhi(p0)) + (bflo(p1) + bfhi(p1));
    sb += (w[0] * b[0] + w[1] * b[1]) + (w[2] * b[2] + w[3] * b[3]);
  }
  red[0][tid] = sc; red[1][tid] = sb;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; }
    __syncthreads();
  }
  if (tid == 0) { c[n] = red[0][0]; b2[n] = (bias ? bias[n] : 0.f) + red[1][0]; }
}

// ---- backward of the folded LayerNorm: dx = g - mean(g) - xhat * mean(g * xhat) [+ dres], g = dy (= rstd o d xhat: the scale travelled
// with dY'), xhat = x * rstd + (-mean * rstd) from stat_fwd.  One wave per row, V 16-byte chunks per lane; optional column sums of the
// bf16-rounded output (the bias gradient of the Linear that produced the residual stream).  No gamma, no dgamma / dbeta: a third fewer
// registers than ln_bwd_kernel, three waves per SIMD with the next row's loads in flight.
template <int V>
__global__ __launch_bounds__(256) void ln_bwd_folded_kernel(const bf16_t* __restrict__ dy, int lddy, const bf16_t* __restrict__ x, int ldx,
                                                            const float* __restrict__ stat, const bf16_t* __restrict__ dres, int lddres,
                                                            bf16_t* __restrict__ dx, int lddx, float* __restrict__ colsum, int rows, int D,
                                                            float* __restrict__ part) {
  __shared__ float red[4][64 * 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = D >> 3;
  const float invD = 1.0f / (float)D;
  float ac[V][8];
#pragma unroll
  for (int i = 0; i < V; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) ac[i][e] = 0.f;
  const int rstep = gridDim.x * 4;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += rstep) {
    const bf16_t* xr = x + (size_t)row * ldx;
    const bf16_t* dyr = dy + (size_t)row * lddy;
    const bf16_t* rr = dres + (size_t)row * lddres;
    u32x4_t cx[V], cd[V], cr[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = lane + i * 64;
      cx[i] = (u32x4_t){0u, 0u, 0u, 0u}; cd[i] = cx[i]; cr[i] = cx[i];
      if (c < nchunk) {
        cx[i] = AVT_LDG_NT((const u32x4_t*)(xr + c * 8));
        cd[i] = AVT_LDG_NT((const u32x4_t*)(dyr + c * 8));
        if (dres) cr[i] = AVT_LDG_NT((const u32x4_t*)(rr + c * 8));
      }
    }
    const f32x2_t st = *(const f32x2_t*)(stat + 2 * (size_t)row);
    float xh[V][8], g[V][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[i][2 * e] = fmaf(bflo(cx[i][e]), st[0], st[1]); xh[i][2 * e + 1] = fmaf(bfhi(cx[i][e]), st[0], st[1]);
          g[i][2 * e] = bflo(cd[i][e]); g[i][2 * e + 1] = bfhi(cd[i][e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1 += g[i][e]; s2 += g[i][e] * xh[i][e]; }
      }
    }
    const float m1 = wave_sum(s1) * invD, m2 = wave_sum(s2) * invD;
    bf16_t* dxr = dx + (size_t)row * lddx;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = g[i][e] - m1 - xh[i][e] * m2;
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
  if (!colsum) return;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave][lane * 8 + e] = ac[i][e];
    __syncthreads();
    for (int idx = threadIdx.x; idx < 512; idx += 256) {
      const int l = idx >> 3, e = idx & 7, cc = l + i * 64;
      if (cc < nchunk) {
        const float sc = red[0][idx] + red[1][idx] + red[2][idx] + red[3][idx];
        if (part) part[(size_t)blockIdx.x * D + cc * 8 + e] = sc;
        else unsafeAtomicAdd(&colsum[cc * 8 + e], sc);
      }
    }
  }
}

// (Round 5, measured and removed: two rows per wave for D = 768, as ln_fwd2_kernel does -- 148 registers, three waves per SIMD: 569 us against this
// kernel's 558 us at 504320 x 768 (5.45 vs 5.55 TB/s), the step 951 vs 953 clips/s; profiles/r05h_ln_bwd_and_attention_scaling.txt.)

// ---- weight-side backward of the fold.  T [N][K] fp32 = dY'^T x (accumulated by avt_gemm_accum_bf16 into a zeroed scratch), dbt [N] = this
// backward's colsum(dY) (unscaled).  Per row n: dG = T[n,:] - mean_k(T[n,:]);  dW[n,:] += gamma o dG + dbt[n] beta;  dbias[n] += dbt[n];
// per column: dgamma[k] += sum_n W[n,k] dG[n,k], dbeta[k] += sum_n W[n,k] dbt[n] (per-workgroup partial vectors, merged in fixed order).
// T and dbt are re-zeroed on the way (the next accumulate starts from zero).  One wave per row, 16 rows per workgroup; K <= 256 V (V <= 8), K % 4 == 0.
template <int V>
__global__ __launch_bounds__(256) void ln_fold_wgrad_kernel(float* __restrict__ T, int ldt, const float* __restrict__ W, int ldw,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ dbt,
                                                            float* __restrict__ dW, int lddw, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            float* __restrict__ dbias, int N, int K, float* __restrict__ part) {
  __shared__ float red[2][4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float invK = 1.0f / (float)K;
  f32x4_t gam[V], bet[V], ag[V], ab[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int k = (lane + 64 * i) * 4;
    gam[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; bet[i] = gam[i];
    if (k < K) { gam[i] = *(const f32x4_t*)(gamma + k); bet[i] = *(const f32x4_t*)(beta + k); }
    ag[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; ab[i] = ag[i];
  }
  const int row0 = blockIdx.x * 16;
  for (int r = wave; r < 16; r += 4) {
    const int n = row0 + r;
    if (n >= N) break;
    float* tr = T + (size_t)n * ldt;
    const float* wr = W + (size_t)n * ldw;
    float* dwr = dW + (size_t)n * lddw;
    f32x4_t t[V], w[V];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int k = (lane + 64 * i) * 4;
      t[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; w[i] = t[i];
      if (k < K) { t[i] = *(const f32x4_t*)(tr + k); w[i] = *(const f32x4_t*)(wr + k); }
      s += (t[i][0] + t[i][1]) + (t[i][2] + t[i][3]);
    }
    const float mean = wave_sum(s) * invK;
    const float db = dbt[n];
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int k = (lane + 64 * i) * 4;
      if (k < K) {
        const f32x4_t dg = t[i] - mean;
        f32x4_t o = *(const f32x4_t*)(dwr + k);
        o += gam[i] * dg + bet[i] * db;
        *(f32x4_t*)(dwr + k) = o;
        ag[i] += w[i] * dg; ab[i] += w[i] * db;
        *(f32x4_t*)(tr + k) = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      }
    }
    if (lane == 0) { if (dbias) dbias[n] += db; dbt[n] = 0.f; }
  }
  // fold the four waves' column partials in wave order, one 256-column slab at a time
#pragma unroll
  for (int i = 0; i < V; ++i) {
    __syncthreads();
    *(f32x4_t*)(&red[0][wave][lane * 4]) = ag[i];
    *(f32x4_t*)(&red[1][wave][lane * 4]) = ab[i];
    __syncthreads();
    const int tcol = threadIdx.x;                       // 256 threads = the slab's 256 columns
    const int k = tcol + 256 * i;
    const float sg = red[0][0][tcol] + red[0][1][tcol] + red[0][2][tcol] + red[0][3][tcol];
    const float sb = red[1][0][tcol] + red[1][1][tcol] + red[1][2][tcol] + red[1][3][tcol];
    if (k >= K) continue;
    if (part) {
      const size_t qs = (size_t)gridDim.x * K;
      part[(size_t)blockIdx.x * K + k] = sg; part[qs + (size_t)blockIdx.x * K + k] = sb;
    } else {
      unsafeAtomicAdd(&dgamma[k], sg); unsafeAtomicAdd(&dbeta[k], sb);
    }
  }
}

int pick_v8(int D) { return (D / 8 + 63) / 64; }

}  // namespace

extern "C" int avt_ln_stats_finalize(const float* stat_part, int nslots, int rows, int D, float eps, float* stat_fwd, float* stat_bwd, void* stream) {
  AVT_CHECK(stat_part && nslots > 0 && rows > 0 && D > 0 && (stat_fwd || stat_bwd), "avt_ln_stats_finalize: bad argument");
  AVT_CHECK(nslots * 32 >= D, "avt_ln_stats_finalize: %d slots of 32 columns do not cover D = %d", nslots, D);
  AVT_CHECK(((((uintptr_t)stat_part) | ((uintptr_t)stat_fwd) | ((uintptr_t)stat_bwd)) & 7) == 0, "avt_ln_stats_finalize: 8-byte alignment required");
  hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, stat_part, nslots, rows, 1.0f / (float)D, eps,
                     stat_fwd, stat_bwd);
  AVT_LAUNCH_CHECK();
  return 0;
}

extern "C" int avt_ln_fold_weights(const float* W, int ldw, const float* gamma, const float* beta, const float* bias, void* G, int ldg,
                                   float* c, float* b2, int N, int K, void* stream) {
  AVT_CHECK(W && gamma && beta && G && c && b2 && N > 0 && K > 0, "avt_ln_fold_weights: null argument");
  AVT_CHECK(K % 4 == 0 && ldw % 4 == 0 && ldg % 4 == 0 && aligned16(W) && aligned16(gamma) && aligned16(beta) && (((uintptr_t)G) & 7) == 0,
            "avt_ln_fold_weights: K and the leading dimensions must be multiples of 4, pointers 16-byte aligned");
  hipLaunchKernelGGL(ln_fold_weights_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, W, ldw, gamma, beta, bias, (bf16_t*)G, ldg, c, b2, K);
  AVT_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t avt_layernorm_bwd_folded_workspace_bytes(int rows, int D) {
  (void)rows;
  return (size_t)768 * (size_t)D * 4;                              // at most 768 workgroups x colsum
}
extern "C" int avt_layernorm_bwd_folded(const void* dy, int lddy, const void* x, int ldx, const float* stat_fwd, const void* dres, int lddres,
                                        void* dx, int lddx, float* colsum, int rows, int D, float* part, size_t part_bytes, void* stream) {
  AVT_CHECK(dy && x && stat_fwd && dx, "avt_layernorm_bwd_folded: null argument");
  AVT_CHECK(rows > 0 && D > 0 && D % 8 == 0 && D <= 4096, "avt_layernorm_bwd_folded: D must be a multiple of 8 and <= 4096 (D=%d)", D);
  AVT_CHECK(lddy % 8 == 0 && ldx % 8 == 0 && lddx % 8 == 0 && (!dres || lddres % 8 == 0), "avt_layernorm_bwd_folded: leading dims must be multiples of 8");
  AVT_CHECK(aligned16(dy) && aligned16(x) && aligned16(dx) && (!dres || aligned16(dres)) && (((uintptr_t)stat_fwd) & 7) == 0, "avt_layernorm_bwd_folded: alignment");
  int grid = (rows + 3) / 4; if (grid > 768) grid = 768;
  if (!colsum) part = nullptr;
  AVT_CHECK(!part || part_bytes >= (size_t)grid * D * 4, "avt_layernorm_bwd_folded: partials workspace too small");
  hipStream_t s = (hipStream_t)stream;
#define LNF(V) hipLaunchKernelGGL((ln_bwd_folded_kernel<V>), dim3(grid), dim3(256), 0, s, (const bf16_t*)dy, lddy, (const bf16_t*)x, ldx, stat_fwd, (const bf16_t*)dres, lddres, (bf16_t*)dx, lddx, colsum, rows, D, part)
  switch (pick_v8(D)) {
    case 1: LNF(1); break; case 2: LNF(2); break; case 3: LNF(3); break; case 4: LNF(4); break;
    case 5: LNF(5); break; case 6: LNF(6); break; case 7: LNF(7); break; default: LNF(8); break;
  }
#undef LNF
  AVT_LAUNCH_CHECK();
  if (part) { float* outs[1] = {colsum}; return avt_reduce_partials(part, grid, D, outs, 1, s); }
  return 0;
}

extern "C" size_t avt_ln_fold_wgrad_workspace_bytes(int N, int K) { return (size_t)2 * (size_t)((N + 15) / 16) * (size_t)K * 4; }
extern "C" int avt_ln_fold_wgrad(float* T, int ldt, const float* W, int ldw, const float* gamma, const float* beta, float* dbias_tmp,
                                 float* dW, int lddw, float* dgamma, float* dbeta, float* dbias, int N, int K,
                                 float* part, size_t part_bytes, void* stream) {
  AVT_CHECK(T && W && gamma && beta && dbias_tmp && dW && dgamma && dbeta && N > 0, "avt_ln_fold_wgrad: null argument");
  AVT_CHECK(K % 4 == 0 && K > 0 && K <= 2048, "avt_ln_fold_wgrad: K must be a multiple of 4, at most 2048 (K=%d)", K);
  AVT_CHECK(ldt % 4 == 0 && ldw % 4 == 0 && lddw % 4 == 0 && aligned16(T) && aligned16(W) && aligned16(dW) && aligned16(gamma) && aligned16(beta),
            "avt_ln_fold_wgrad: leading dimensions must be multiples of 4, pointers 16-byte aligned");
  const int grid = (N + 15) / 16;
  AVT_CHECK(!part || (aligned16(part) && part_bytes >= (size_t)2 * grid * K * 4), "avt_ln_fold_wgrad: partials workspace too small or misaligned");
  hipStream_t s = (hipStream_t)stream;
#define LFW(V) hipLaunchKernelGGL((ln_fold_wgrad_kernel<V>), dim3(grid), dim3(256), 0, s, T, ldt, W, ldw, gamma, beta, dbias_tmp, dW, lddw, dgamma, dbeta, dbias, N, K, part)
  switch ((K + 255) / 256) {
    case 1: LFW(1); break; case 2: LFW(2); break; case 3: LFW(3); break; case 4: LFW(4); break;
    case 5: LFW(5); break; case 6: LFW(6); break; case 7: LFW(7); break; default: LFW(8); break;
  }
#undef LFW
  AVT_LAUNCH_CHECK();
  if (part) { float* outs[2] = {dgamma, dbeta}; return avt_reduce_partials(part, grid, K, outs, 2, s); }
  return 0;
}
