// Single-query attention kernels (HBM-bound, vector ALU, fp32 math, bf16 I/O).
//
// (1) CLS-query attention of the LAST ViT block.  Only token 0 of that block's output is consumed
//     (timm VisionTransformer.forward_features returns x[:, 0] after the final norm -- reached through
//     models/video_classification.py:224 -> models/base_model.py:157), so its attention is needed for ONE query per
//     (frame, head) against all S keys: out[n, h] = softmax(q_cls k^T * scale) v.  One wave per (frame, head).
//       q    [N, H*64]       the CLS rows' queries (compact)
//       kv   [N*S, 2*H*64]   columns [k | v], head-major (the k|v two thirds of timm's qkv Linear)
//       probs fp32 [N, H, S] saved for backward
//     Backward: dv_j = p_j do, dp_j = do . v_j, ds_j = p_j (dp_j - sum_i p_i dp_i), dq = scale sum_j ds_j k_j,
//     dk_j = scale ds_j q.  Bias gradients need no extra pass: colsum(dk) == 0 (softmax shift invariance), colsum(dv) ==
//     colsum(do), colsum(dq) is taken by the caller over the compact [N, D] tensor.
// (2) KV-cache decode attention of the AVT-h roll-out (models/future_prediction.py:168-202, HF GPT-2 `past_key_values`):
//     one new token per clip attends over the cached keys 0..pos (causal by construction).
#include "common.hpp"
#include "../../include/avt_hip.h"

namespace {
constexpr int CLS_SMAX = 256;

// A wave walks its item's keys 8 rows at a time: lane = (row r8 = lane >> 3, chunk ch = lane & 7) loads 16 bytes = dims
// [8 ch, 8 ch + 8) of key 8 g + r8, so one load instruction covers eight full 128-byte rows (the earlier one-row-per-lane layout
// touched 64 rows per instruction, 16 bytes each, and lived off L1 reuse: 2.6-2.7 TB/s).  Per-key scalars go through LDS.
__device__ __forceinline__ void unpack8(u32x4_t w, float (&r)[8]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[2 * e] = bflo(w[e]); r[2 * e + 1] = bfhi(w[e]); }
}
__device__ __forceinline__ float dot8(u32x4_t w, const float (&r)[8]) {
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) { s0 = fmaf(bflo(w[e]), r[2 * e], s0); s1 = fmaf(bfhi(w[e]), r[2 * e + 1], s1); }
  return s0 + s1;
}
__device__ __forceinline__ float row8_sum(float v) {          // over the 8 lanes of a row group
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
  return v;
}
__device__ __forceinline__ float col8_sum(float v) {          // over the 8 row groups
  v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
  return v;
}
constexpr int CLS_UNROLL = 5;                                  // key groups (of 8 rows) in flight per wave

__global__ __launch_bounds__(256) void cls_attn_fwd_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ kv, int ldkv,
                                                           bf16_t* __restrict__ out, int ldo, float* __restrict__ probs,
                                                           int S, int H, float scale, int items) {
  __shared__ float P[4][CLS_SMAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int item = blockIdx.x * 4 + wave;
  if (item >= items) return;                       // whole wave leaves; no block-level barrier below
  const int n = item / H, h = item % H;
  const int D = H * 64;
  const int r8 = lane >> 3, ch = lane & 7;
  float qr[8];
  unpack8(*(const u32x4_t*)(q + (size_t)n * ldq + h * 64 + ch * 8), qr);
  const bf16_t* kbase = kv + (size_t)n * S * ldkv + h * 64 + ch * 8;
  const int ngroups = (S + 7) / 8;
  for (int g0 = 0; g0 < ngroups; g0 += CLS_UNROLL) {
    u32x4_t w[CLS_UNROLL];
#pragma unroll
    for (int u = 0; u < CLS_UNROLL; ++u) {
      const int j = (g0 + u) * 8 + r8;
      w[u] = (u32x4_t){0u, 0u, 0u, 0u};
      if (j < S) w[u] = *(const u32x4_t*)(kbase + (size_t)j * ldkv);
    }
#pragma unroll
    for (int u = 0; u < CLS_UNROLL; ++u) {
      const int j = (g0 + u) * 8 + r8;
      const float sc = row8_sum(dot8(w[u], qr)) * scale;
      if (ch == 0 && j < S) P[wave][j] = sc;
    }
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float s[4], mx = -3.0e38f;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) { const int j = jj * 64 + lane; s[jj] = (j < S) ? P[wave][j] : -3.0e38f; mx = fmaxf(mx, s[jj]); }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) { const int j = jj * 64 + lane; s[jj] = (j < S) ? __expf(s[jj] - mx) : 0.f; sum += s[jj]; }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  float* prow = probs + (size_t)item * S;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int j = jj * 64 + lane;
    if (j < S) { const float p = s[jj] * inv; prow[j] = p; P[wave][j] = p; }
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // out[d] = sum_j p_j v[j][d]: every lane accumulates its 8 dims over the rows of its row group
  const bf16_t* vbase = kbase + D;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int g0 = 0; g0 < ngroups; g0 += CLS_UNROLL) {
    u32x4_t w[CLS_UNROLL];
    float pj[CLS_UNROLL];
#pragma unroll
    for (int u = 0; u < CLS_UNROLL; ++u) {
      const int j = (g0 + u) * 8 + r8;
      w[u] = (u32x4_t){0u, 0u, 0u, 0u}; pj[u] = 0.f;
      if (j < S) { w[u] = *(const u32x4_t*)(vbase + (size_t)j * ldkv); pj[u] = P[wave][j]; }
    }
#pragma unroll
    for (int u = 0; u < CLS_UNROLL; ++u) {
      float v[8];
      unpack8(w[u], v);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(pj[u], v[e], acc[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = col8_sum(acc[e]);
  if (r8 == 0) {
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2bf(acc[2 * e], acc[2 * e + 1]);
    *(u32x4_t*)(out + (size_t)n * ldo + h * 64 + ch * 8) = o;
  }
}

__global__ __launch_bounds__(256) void cls_attn_bwd_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ kv, int ldkv,
                                                           const float* __restrict__ probs, const bf16_t* __restrict__ dout, int lddo,
                                                           bf16_t* __restrict__ dq, int lddq, bf16_t* __restrict__ dkv, int lddkv,
                                                           int S, int H, float scale, int items) {
  __shared__ float PP[4][CLS_SMAX];     // p_j
  __shared__ float DP[4][CLS_SMAX];     // dp_j = do . v_j
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int item = blockIdx.x * 4 + wave;
  if (item >= items) return;
  const int n = item / H, h = item % H;
  const int D = H * 64;
  const int r8 = lane >> 3, ch = lane & 7;
  float dor[8], qr[8];
  unpack8(*(const u32x4_t*)(dout + (size_t)n * lddo + h * 64 + ch * 8), dor);
  unpack8(*(const u32x4_t*)(q + (size_t)n * ldq + h * 64 + ch * 8), qr);
  const bf16_t* kbase = kv + (size_t)n * S * ldkv + h * 64 + ch * 8;
  bf16_t* dkbase = dkv + (size_t)n * S * lddkv + h * 64 + ch * 8;
  const float* prow = probs + (size_t)item * S;
  const int ngroups = (S + 7) / 8;
  // pass 1 over V: dp_j, delta = sum_j p_j dp_j, dv_j = p_j do
  float delta = 0.f;
  for (int g0 = 0; g0 < ngroups; g0 += CLS_UNROLL) {
    u32x4_t w[CLS_UNROLL];
    float pj[CLS_UNROLL];
#pragma unroll
    for (int u = 0; u < CLS_UNROLL; ++u) {
      const int j = (g0 + u) * 8 + r8;
      w[u] = (u32x4_t){0u, 0u, 0u, 0u}; pj[u] = 0.f;
      if (j < S) { w[u] = *(const u32x4_t*)(kbase + D + (size_t)j * ldkv); pj[u] = prow[j]; }
    }
#pragma unroll
    for (int u = 0; u < CLS_UNROLL; ++u) {
      const int j = (g0 + u) * 8 + r8;
      const float dpj = row8_sum(dot8(w[u], dor));
      if (j < S) {
        if (ch == 0) { PP[wave][j] = pj[u]; DP[wave][j] = dpj; delta = fmaf(pj[u], dpj, delta); }
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack2bf(pj[u] * dor[2 * e], pj[u] * dor[2 * e + 1]);
        *(u32x4_t*)(dkbase + D + (size_t)j * lddkv) = o;
      }
    }
  }
  delta = wave_sum(delta);
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // pass 2 over K: ds_j = scale p_j (dp_j - delta), dq = sum_j ds_j k_j, dk_j = ds_j q
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int g0 = 0; g0 < ngroups; g0 += CLS_UNROLL) {
    u32x4_t w[CLS_UNROLL];
    float ds[CLS_UNROLL];
#pragma unroll
    for (int u = 0; u < CLS_UNROLL; ++u) {
      const int j = (g0 + u) * 8 + r8;
      w[u] = (u32x4_t){0u, 0u, 0u, 0u}; ds[u] = 0.f;
      if (j < S) { w[u] = *(const u32x4_t*)(kbase + (size_t)j * ldkv); ds[u] = PP[wave][j] * (DP[wave][j] - delta) * scale; }
    }
#pragma unroll
    for (int u = 0; u < CLS_UNROLL; ++u) {
      const int j = (g0 + u) * 8 + r8;
      float k8[8];
      unpack8(w[u], k8);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(ds[u], k8[e], acc[e]);
      if (j < S) {
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack2bf(ds[u] * qr[2 * e], ds[u] * qr[2 * e + 1]);
        *(u32x4_t*)(dkbase + (size_t)j * lddkv) = o;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = col8_sum(acc[e]);
  if (r8 == 0) {
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2bf(acc[2 * e], acc[2 * e + 1]);
    *(u32x4_t*)(dq + (size_t)n * lddq + h * 64 + ch * 8) = o;
  }
}

// ---- KV-cache decode attention: one new token per clip ------------------------------------------------------------------
// block = (clip b, head h), 256 threads.  The token's k / v are appended to the caches at row `pos` first.
constexpr int DEC_TMAX = 1024;
__global__ __launch_bounds__(256) void causal_decode_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ kc, bf16_t* __restrict__ vc,
                                                            bf16_t* __restrict__ out, int H, int hd, int pos, int tmax, float scale) {
  __shared__ float P[DEC_TMAX];
  __shared__ float red[4];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int E = H * hd;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bf16_t* qrow = qkv + (size_t)b * 3 * E + h * hd;
  bf16_t* kb = kc + (size_t)b * tmax * E + h * hd;
  bf16_t* vb = vc + (size_t)b * tmax * E + h * hd;
  for (int c = tid; c < hd / 8; c += 256) {
    *(u32x4_t*)(kb + (size_t)pos * E + c * 8) = *(const u32x4_t*)(qrow + E + c * 8);
    *(u32x4_t*)(vb + (size_t)pos * E + c * 8) = *(const u32x4_t*)(qrow + 2 * E + c * 8);
  }
  __syncthreads();                                   // the appended row is read back through this block's own stores
  for (int j = wave; j <= pos; j += 4) {
    float s = 0.f;
    for (int c = lane; c < hd / 8; c += 64) {
      u32x4_t x = *(const u32x4_t*)(qrow + c * 8), y = *(const u32x4_t*)(kb + (size_t)j * E + c * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) s += bflo(x[e]) * bflo(y[e]) + bfhi(x[e]) * bfhi(y[e]);
    }
    s = wave_sum(s) * scale;
    if (lane == 0) P[j] = s;
  }
  __syncthreads();
  float mx = -3.0e38f;
  for (int j = tid; j <= pos; j += 256) mx = fmaxf(mx, P[j]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int j = tid; j <= pos; j += 256) { float e = __expf(P[j] - mx); P[j] = e; sum += e; }
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
  for (int c = tid; c < hd / 8; c += 256) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j <= pos; ++j) {
      const float pj = P[j] * inv;
      u32x4_t w = *(const u32x4_t*)(vb + (size_t)j * E + c * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[2 * e] = fmaf(pj, bflo(w[e]), acc[2 * e]); acc[2 * e + 1] = fmaf(pj, bfhi(w[e]), acc[2 * e + 1]); }
    }
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2bf(acc[2 * e], acc[2 * e + 1]);
    *(u32x4_t*)(out + (size_t)b * E + h * hd + c * 8) = o;
  }
}
}  // namespace

extern "C" int avt_cls_attn_fwd(const void* q, int ldq, const void* kv, int ldkv, void* out, int ldo, float* probs,
                                int frames, int S, int H, int head_dim, float scale, void* stream) {
  AVT_CHECK(q && kv && out && probs, "avt_cls_attn_fwd: null argument");
  AVT_CHECK(head_dim == 64, "avt_cls_attn_fwd: head_dim must be 64 (got %d)", head_dim);
  AVT_CHECK(frames > 0 && H > 0 && S > 0 && S <= CLS_SMAX, "avt_cls_attn_fwd: need 0 < S <= %d (got %d)", CLS_SMAX, S);
  AVT_CHECK(ldq % 8 == 0 && ldkv % 8 == 0 && ldo % 8 == 0 && aligned16(q) && aligned16(kv) && aligned16(out), "avt_cls_attn_fwd: 16-byte aligned rows required");
  const int items = frames * H;
  hipLaunchKernelGGL(cls_attn_fwd_kernel, dim3((items + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q, ldq,
                     (const bf16_t*)kv, ldkv, (bf16_t*)out, ldo, probs, S, H, scale, items);
  AVT_LAUNCH_CHECK();
  return 0;
}

extern "C" int avt_cls_attn_bwd(const void* q, int ldq, const void* kv, int ldkv, const float* probs, const void* dout, int lddo,
                                void* dq, int lddq, void* dkv, int lddkv, int frames, int S, int H, int head_dim, float scale,
                                void* stream) {
  AVT_CHECK(q && kv && probs && dout && dq && dkv, "avt_cls_attn_bwd: null argument");
  AVT_CHECK(head_dim == 64, "avt_cls_attn_bwd: head_dim must be 64 (got %d)", head_dim);
  AVT_CHECK(frames > 0 && H > 0 && S > 0 && S <= CLS_SMAX, "avt_cls_attn_bwd: need 0 < S <= %d (got %d)", CLS_SMAX, S);
  AVT_CHECK(ldq % 8 == 0 && ldkv % 8 == 0 && lddo % 8 == 0 && lddkv % 8 == 0 && lddq % 8 == 0 && aligned16(q) && aligned16(kv) &&
            aligned16(dout) && aligned16(dkv) && aligned16(dq), "avt_cls_attn_bwd: 16-byte aligned rows required");
  const int items = frames * H;
  hipLaunchKernelGGL(cls_attn_bwd_kernel, dim3((items + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q, ldq,
                     (const bf16_t*)kv, ldkv, probs, (const bf16_t*)dout, lddo, (bf16_t*)dq, lddq, (bf16_t*)dkv, lddkv, S, H, scale, items);
  AVT_LAUNCH_CHECK();
  return 0;
}

extern "C" int avt_causal_attn_decode(const void* qkv, void* kcache, void* vcache, void* out, int B, int H, int head_dim, int pos,
                                      int tmax, float scale, void* stream) {
  AVT_CHECK(qkv && kcache && vcache && out, "avt_causal_attn_decode: null argument");
  AVT_CHECK(B > 0 && H > 0 && head_dim > 0 && head_dim % 8 == 0, "avt_causal_attn_decode: head_dim must be a positive multiple of 8");
  AVT_CHECK(pos >= 0 && pos < tmax && tmax <= DEC_TMAX, "avt_causal_attn_decode: need 0 <= pos < tmax <= %d (pos %d, tmax %d)", DEC_TMAX, pos, tmax);
  hipLaunchKernelGGL(causal_decode_kernel, dim3(B * H), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, (bf16_t*)kcache,
                     (bf16_t*)vcache, (bf16_t*)out, H, head_dim, pos, tmax, scale);
  AVT_LAUNCH_CHECK();
  return 0;
}
