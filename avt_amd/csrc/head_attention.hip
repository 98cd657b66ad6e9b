// AVT-h causal self-attention core (HF GPT-2 attention: softmax(q k^T / sqrt(hd) + causal mask), attention
// dropout, times v) for the temporal head: T <= 32 frame tokens, 4 heads x 512 (any hd % 8 == 0).
// The problem is tiny (B*heads workgroups of T x T scores), so it runs on the vector ALU in fp32 with bf16 I/O:
// one workgroup per (clip, head), dot products spread over the 64 lanes of a wave.
//   qkv layout: [B*T, 3*E] with columns [q | k | v], each head-major (HF split(E, dim=2) + view(B,T,H,hd)).
// Forward saves the pre-dropout probabilities (fp32, [B,H,T,T]); the dropout mask is recomputed from (seed, idx).
#include "common.hpp"
#include "../../include/avt_hip.h"

namespace {
constexpr int TMAX = 32;

__device__ __forceinline__ float dot_wave(const bf16_t* a, const bf16_t* b, int hd, int lane) {
  float s = 0.f;
  for (int c = lane; c < hd / 8; c += 64) {
    u32x4_t x = *(const u32x4_t*)(a + c * 8), y = *(const u32x4_t*)(b + c * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) s += bflo(x[e]) * bflo(y[e]) + bfhi(x[e]) * bfhi(y[e]);
  }
  return wave_sum(s);
}

__global__ __launch_bounds__(256) void causal_attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                              float* __restrict__ probs, int T, int H, int hd, float scale,
                                                              uint32_t drop_thresh, float drop_scale, uint64_t seed, int causal) {
  __shared__ float P[TMAX][TMAX + 1];
  seed = resolve_seed(seed);
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int E = H * hd, ld = 3 * E;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bf16_t* base = qkv + (size_t)b * T * ld + h * hd;
  for (int pr = wave; pr < T * T; pr += 4) {
    int i = pr / T, j = pr % T;
    if (j <= i || !causal) {
      float s = dot_wave(base + (size_t)i * ld, base + (size_t)j * ld + E, hd, lane) * scale;
      if (lane == 0) P[i][j] = s;
    }
  }
  __syncthreads();
  if (tid < T) {
    int i = tid;
    const int lim = causal ? i : T - 1;                   // last visible key of query i
    float mx = -3.0e38f;
    for (int j = 0; j <= lim; ++j) mx = fmaxf(mx, P[i][j]);
    float sum = 0.f;
    for (int j = 0; j <= lim; ++j) { float e = __expf(P[i][j] - mx); P[i][j] = e; sum += e; }
    float inv = 1.f / sum;
    float* prow = probs + (((size_t)b * H + h) * T + i) * T;
    for (int j = 0; j < T; ++j) {
      float pv = (j <= lim) ? P[i][j] * inv : 0.f;
      prow[j] = pv;
      if (drop_thresh && j <= lim) {
        uint64_t idx = (((uint64_t)b * H + h) * T + i) * T + j;
        pv = drop_keep(seed, idx, drop_thresh) ? pv * drop_scale : 0.f;
      }
      P[i][j] = pv;
    }
  }
  __syncthreads();
  const int nch = hd / 8;
  for (int task = tid; task < T * nch; task += 256) {
    int i = task / nch, c = task % nch;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int lim = causal ? i : T - 1;
    for (int j = 0; j <= lim; ++j) {
      float pv = P[i][j];
      u32x4_t v = *(const u32x4_t*)(base + (size_t)j * ld + 2 * E + c * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[2 * e] += pv * bflo(v[e]); acc[2 * e + 1] += pv * bfhi(v[e]); }
    }
    u32x4_t w;
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = pack2bf(acc[2 * e], acc[2 * e + 1]);
    *(u32x4_t*)(out + ((size_t)b * T + i) * E + h * hd + c * 8) = w;
  }
}

__global__ __launch_bounds__(256) void causal_attn_bwd_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ probs,
                                                              const bf16_t* __restrict__ dout, bf16_t* __restrict__ dqkv,
                                                              int T, int H, int hd, float scale,
                                                              uint32_t drop_thresh, float drop_scale, uint64_t seed, int causal) {
  __shared__ float P[TMAX][TMAX + 1];    // pre-dropout probabilities
  __shared__ float Pd[TMAX][TMAX + 1];   // dropped probabilities (what multiplied v)
  __shared__ float dS[TMAX][TMAX + 1];
  seed = resolve_seed(seed);
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int E = H * hd, ld = 3 * E;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bf16_t* base = qkv + (size_t)b * T * ld + h * hd;
  const bf16_t* dob = dout + (size_t)b * T * E + h * hd;
  bf16_t* dbase = dqkv + (size_t)b * T * ld + h * hd;
  for (int pr = tid; pr < T * T; pr += 256) {
    int i = pr / T, j = pr % T;
    float pv = probs[(((size_t)b * H + h) * T + i) * T + j];
    float m = 1.f;
    if (drop_thresh && (j <= i || !causal)) {
      uint64_t idx = (((uint64_t)b * H + h) * T + i) * T + j;
      m = drop_keep(seed, idx, drop_thresh) ? drop_scale : 0.f;
    }
    P[i][j] = pv;
    Pd[i][j] = pv * m;
  }
  __syncthreads();
  // dPd[i][j] = dout[i] . v[j]  ->  dP = dPd * mask
  for (int pr = wave; pr < T * T; pr += 4) {
    int i = pr / T, j = pr % T;
    if (j <= i || !causal) {
      float d = dot_wave(dob + (size_t)i * E, base + (size_t)j * ld + 2 * E, hd, lane);
      if (lane == 0) {
        float m = (P[i][j] != 0.f) ? Pd[i][j] / P[i][j] : 0.f;   // mask factor (0 or 1/(1-p))
        dS[i][j] = d * m;
      }
    } else if (lane == 0) dS[i][j] = 0.f;
  }
  __syncthreads();
  if (tid < T) {
    int i = tid;
    const int lim = causal ? i : T - 1;
    float dotp = 0.f;
    for (int j = 0; j <= lim; ++j) dotp += dS[i][j] * P[i][j];
    for (int j = 0; j <= lim; ++j) dS[i][j] = P[i][j] * (dS[i][j] - dotp) * scale;
  }
  __syncthreads();
  const int nch = hd / 8;
  for (int task = tid; task < T * nch; task += 256) {
    int i = task / nch, c = task % nch;
    float aq[8], ak[8], av[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { aq[e] = 0.f; ak[e] = 0.f; av[e] = 0.f; }
    const int lim = causal ? i : T - 1;
    for (int j = 0; j <= lim; ++j) {          // dq[i] = sum_{j visible} dS[i][j] k[j]
      float w = dS[i][j];
      u32x4_t k = *(const u32x4_t*)(base + (size_t)j * ld + E + c * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) { aq[2 * e] += w * bflo(k[e]); aq[2 * e + 1] += w * bfhi(k[e]); }
    }
    for (int r = causal ? i : 0; r < T; ++r) { // dk[i] = sum_{r sees i} dS[r][i] q[r] ; dv[i] = sum_{r sees i} Pd[r][i] dout[r]
      float w = dS[r][i], pw = Pd[r][i];
      u32x4_t q = *(const u32x4_t*)(base + (size_t)r * ld + c * 8);
      u32x4_t d = *(const u32x4_t*)(dob + (size_t)r * E + c * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ak[2 * e] += w * bflo(q[e]); ak[2 * e + 1] += w * bfhi(q[e]);
        av[2 * e] += pw * bflo(d[e]); av[2 * e + 1] += pw * bfhi(d[e]);
      }
    }
    u32x4_t wq, wk, wv;
#pragma unroll
    for (int e = 0; e < 4; ++e) { wq[e] = pack2bf(aq[2 * e], aq[2 * e + 1]); wk[e] = pack2bf(ak[2 * e], ak[2 * e + 1]); wv[e] = pack2bf(av[2 * e], av[2 * e + 1]); }
    *(u32x4_t*)(dbase + (size_t)i * ld + c * 8) = wq;
    *(u32x4_t*)(dbase + (size_t)i * ld + E + c * 8) = wk;
    *(u32x4_t*)(dbase + (size_t)i * ld + 2 * E + c * 8) = wv;
  }
}
}  // namespace

extern "C" int avt_head_attn_fwd(const void* qkv, void* out, float* probs, int B, int T, int H, int head_dim, float scale,
                                 float drop_p, uint64_t seed, int causal, void* stream) {
  AVT_CHECK(qkv && out && probs, "avt_head_attn_fwd: null argument");
  AVT_CHECK(T >= 1 && T <= TMAX, "avt_head_attn_fwd: T must be in [1, %d] (got %d)", TMAX, T);
  AVT_CHECK(head_dim % 8 == 0 && B > 0 && H > 0, "avt_head_attn_fwd: head_dim must be a multiple of 8");
  AVT_CHECK(aligned16(qkv) && aligned16(out), "avt_head_attn_fwd: 16-byte alignment required");
  AVT_CHECK(drop_p >= 0.f && drop_p < 1.f, "avt_head_attn_fwd: bad dropout p");
  hipLaunchKernelGGL(causal_attn_fwd_kernel, dim3(B * H), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, (bf16_t*)out, probs,
                     T, H, head_dim, scale, drop_threshold(drop_p), 1.f / (1.f - drop_p), seed, causal);
  AVT_LAUNCH_CHECK();
  return 0;
}

extern "C" int avt_head_attn_bwd(const void* qkv, const float* probs, const void* dout, void* dqkv, int B, int T, int H,
                                 int head_dim, float scale, float drop_p, uint64_t seed, int causal, void* stream) {
  AVT_CHECK(qkv && probs && dout && dqkv, "avt_head_attn_bwd: null argument");
  AVT_CHECK(T >= 1 && T <= TMAX, "avt_head_attn_bwd: T must be in [1, %d] (got %d)", TMAX, T);
  AVT_CHECK(head_dim % 8 == 0 && B > 0 && H > 0, "avt_head_attn_bwd: head_dim must be a multiple of 8");
  AVT_CHECK(aligned16(qkv) && aligned16(dout) && aligned16(dqkv), "avt_head_attn_bwd: 16-byte alignment required");
  hipLaunchKernelGGL(causal_attn_bwd_kernel, dim3(B * H), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, probs, (const bf16_t*)dout,
                     (bf16_t*)dqkv, T, H, head_dim, scale, drop_threshold(drop_p), 1.f / (1.f - drop_p), seed, causal);
  AVT_LAUNCH_CHECK();
  return 0;
}

extern "C" int avt_causal_attn_fwd(const void* qkv, void* out, float* probs, int B, int T, int H, int head_dim, float scale,
                                   float drop_p, uint64_t seed, void* stream) {
  return avt_head_attn_fwd(qkv, out, probs, B, T, H, head_dim, scale, drop_p, seed, 1, stream);
}

extern "C" int avt_causal_attn_bwd(const void* qkv, const float* probs, const void* dout, void* dqkv, int B, int T, int H,
                                   int head_dim, float scale, float drop_p, uint64_t seed, void* stream) {
  return avt_head_attn_bwd(qkv, probs, dout, dqkv, B, T, H, head_dim, scale, drop_p, seed, 1, stream);
}
