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

__device__ __forceinline__ void load_row64(const bf16_t* p, float (&r)[64]) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    u32x4_t w = *(const u32x4_t*)(p + c * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) { r[c * 8 + 2 * e] = bflo(w[e]); r[c * 8 + 2 * e + 1] = bfhi(w[e]); }
  }
}
__device__ __forceinline__ float dot_row64(const bf16_t* p, const float (&r)[64]) {
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    u32x4_t w = *(const u32x4_t*)(p + c * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) { s0 = fmaf(bflo(w[e]), r[c * 8 + 2 * e], s0); s1 = fmaf(bfhi(w[e]), r[c * 8 + 2 * e + 1], s1); }
  }
  return s0 + s1;
}

__global__ __launch_bounds__(256) void cls_attn_fwd_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ kv, int ldkv,
                                                           bf16_t* __restrict__ out, int ldo, float* __restrict__ probs,
                                                           int S, int H, float scale, int items) {
  __shared__ float P[4][CLS_SMAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int item = blockIdx.x * 4 + wave;
  if (item >= items) return;                       // whole wave leaves; no block-level barrier below
  const int n = item / H, h = item % H;
  const int D = H * 64;
  float qr[64];
  load_row64(q + (size_t)n * ldq + h * 64, qr);
  const bf16_t* kbase = kv + (size_t)n * S * ldkv + h * 64;
  float s[4], mx = -3.0e38f;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int j = jj * 64 + lane;
    s[jj] = -3.0e38f;
    if (j < S) { s[jj] = dot_row64(kbase + (size_t)j * ldkv, qr) * scale; mx = fmaxf(mx, s[jj]); }
  }
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
  // out[d] = sum_j p_j v[j][d]: lane = (key parity, dim pair); row reads are 128 contiguous bytes per key
  const int dp = lane & 31, par = lane >> 5;
  const bf16_t* vbase = kbase + D + dp * 2;
  float a0 = 0.f, a1 = 0.f;
  for (int j = par; j < S; j += 2) {
    const uint32_t w = *(const uint32_t*)(vbase + (size_t)j * ldkv);
    const float p = P[wave][j];
    a0 = fmaf(p, bflo(w), a0); a1 = fmaf(p, bfhi(w), a1);
  }
  a0 += __shfl_xor(a0, 32, 64); a1 += __shfl_xor(a1, 32, 64);
  if (par == 0) *(uint32_t*)(out + (size_t)n * ldo + h * 64 + dp * 2) = pack2bf(a0, a1);
}

__global__ __launch_bounds__(256) void cls_attn_bwd_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ kv, int ldkv,
                                                           const float* __restrict__ probs, const bf16_t* __restrict__ dout, int lddo,
                                                           bf16_t* __restrict__ dq, int lddq, bf16_t* __restrict__ dkv, int lddkv,
                                                           int S, int H, float scale, int items) {
  __shared__ float DS[4][CLS_SMAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int item = blockIdx.x * 4 + wave;
  if (item >= items) return;
  const int n = item / H, h = item % H;
  const int D = H * 64;
  float dor[64];
  load_row64(dout + (size_t)n * lddo + h * 64, dor);
  const bf16_t* kbase = kv + (size_t)n * S * ldkv + h * 64;
  const float* prow = probs + (size_t)item * S;
  float p[4], dpv[4], delta = 0.f;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int j = jj * 64 + lane;
    p[jj] = 0.f; dpv[jj] = 0.f;
    if (j < S) { p[jj] = prow[j]; dpv[jj] = dot_row64(kbase + D + (size_t)j * ldkv, dor); delta = fmaf(p[jj], dpv[jj], delta); }
  }
  delta = wave_sum(delta);
  // dv_j = p_j * do (this lane's keys, whole 64-wide rows from registers)
  bf16_t* dkbase = dkv + (size_t)n * S * lddkv + h * 64;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int j = jj * 64 + lane;
    if (j < S) {
      bf16_t* row = dkbase + D + (size_t)j * lddkv;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack2bf(p[jj] * dor[c * 8 + 2 * e], p[jj] * dor[c * 8 + 2 * e + 1]);
        *(u32x4_t*)(row + c * 8) = o;
      }
    }
  }
  float ds[4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int j = jj * 64 + lane;
    ds[jj] = p[jj] * (dpv[jj] - delta) * scale;
    if (j < S) DS[wave][j] = ds[jj];
  }
  // dk_j = scale ds_j q: reuse the register row for q
  load_row64(q + (size_t)n * ldq + h * 64, dor);
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int j = jj * 64 + lane;
    if (j < S) {
      bf16_t* row = dkbase + (size_t)j * lddkv;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack2bf(ds[jj] * dor[c * 8 + 2 * e], ds[jj] * dor[c * 8 + 2 * e + 1]);
        *(u32x4_t*)(row + c * 8) = o;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // dq[d] = sum_j (scale ds_j) k[j][d]
  const int dp = lane & 31, par = lane >> 5;
  const bf16_t* kcol = kbase + dp * 2;
  float a0 = 0.f, a1 = 0.f;
  for (int j = par; j < S; j += 2) {
    const uint32_t w = *(const uint32_t*)(kcol + (size_t)j * ldkv);
    const float d = DS[wave][j];
    a0 = fmaf(d, bflo(w), a0); a1 = fmaf(d, bfhi(w), a1);
  }
  a0 += __shfl_xor(a0, 32, 64); a1 += __shfl_xor(a1, 32, 64);
  if (par == 0) *(uint32_t*)(dq + (size_t)n * lddq + h * 64 + dp * 2) = pack2bf(a0, a1);
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
  AVT_CHECK(ldq % 8 == 0 && ldkv % 8 == 0 && ldo % 2 == 0 && aligned16(q) && aligned16(kv), "avt_cls_attn_fwd: 16-byte aligned rows required");
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
  AVT_CHECK(ldq % 8 == 0 && ldkv % 8 == 0 && lddo % 8 == 0 && lddkv % 8 == 0 && lddq % 2 == 0 && aligned16(q) && aligned16(kv) &&
            aligned16(dout) && aligned16(dkv), "avt_cls_attn_bwd: 16-byte aligned rows required");
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
