// Softmax cross-entropy over C classes with ignore_index (loss_fn/multidim_xentropy.py:11-25 ->
// torch.nn.CrossEntropyLoss(ignore_index=-1, reduction='none')) plus the rank of the target logit (top-1 / top-5
// accuracy, common/utils.py:17-44).  One workgroup per row; logits fp32 [R, ld] (columns >= C are padding).
//   forward : loss[r] = lse[r] - logit[r, t]  (0 when t == ignore_index); rank[r] = #{c : logit[c] > logit[t]}
//   backward: dlogits[r, c] = (exp(logit - lse) - [c == t]) * gout[r]  (0 for ignored rows and padding columns),
//             written in bf16 for the classifier's dgrad / wgrad GEMMs.
#include "common.hpp"
#include "../../include/avt_hip.h"

namespace {
__device__ __forceinline__ float block_reduce(float v, bool is_max, float* sh) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float r = sh[0];
  for (int w = 1; w < 4; ++w) r = is_max ? fmaxf(r, sh[w]) : r + sh[w];
  return r;
}

__global__ __launch_bounds__(256) void xent_fwd_kernel(const float* __restrict__ logits, int ld, const long* __restrict__ target,
                                                       float* __restrict__ loss, float* __restrict__ lse_out, int* __restrict__ rank,
                                                       int C, long ignore_index) {
  __shared__ float sh[4];
  const int r = blockIdx.x;
  const float* row = logits + (size_t)r * ld;
  const long t = target[r];
  float mx = -3.0e38f;
  for (int c = threadIdx.x; c < C; c += 256) mx = fmaxf(mx, row[c]);
  mx = block_reduce(mx, true, sh);
  const bool valid = (t != ignore_index) && t >= 0 && t < C;
  const float lt = valid ? row[t] : 0.f;
  float sum = 0.f, gt = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    float v = row[c];
    sum += __expf(v - mx);
    gt += (valid && v > lt) ? 1.f : 0.f;
  }
  sum = block_reduce(sum, false, sh);
  gt = block_reduce(gt, false, sh);
  if (threadIdx.x == 0) {
    float l = mx + __logf(sum);
    lse_out[r] = l;
    loss[r] = valid ? (l - lt) : 0.f;
    if (rank) rank[r] = valid ? (int)gt : -1;
  }
}

__global__ __launch_bounds__(256) void xent_bwd_kernel(const float* __restrict__ logits, int ld, const long* __restrict__ target,
                                                       const float* __restrict__ lse, const float* __restrict__ gout,
                                                       bf16_t* __restrict__ dlogits, int ldd, int C, long ignore_index) {
  const int r = blockIdx.x;
  const float* row = logits + (size_t)r * ld;
  const long t = target[r];
  const bool valid = (t != ignore_index) && t >= 0 && t < C;
  const float g = valid ? gout[r] : 0.f, l = lse[r];
  bf16_t* drow = dlogits + (size_t)r * ldd;
  for (int c = threadIdx.x; c < ldd; c += 256) {
    float v = 0.f;
    if (valid && c < C) v = (__expf(row[c] - l) - ((long)c == t ? 1.f : 0.f)) * g;
    drow[c] = f2bf(v);
  }
}
}  // namespace

extern "C" int avt_xent_fwd(const float* logits, int ld, const long* target, float* loss, float* lse, int* rank, int R, int C,
                            long ignore_index, void* stream) {
  AVT_CHECK(logits && target && loss && lse && R > 0 && C > 0 && ld >= C, "avt_xent_fwd: bad argument");
  hipLaunchKernelGGL(xent_fwd_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, logits, ld, target, loss, lse, rank, C, ignore_index);
  AVT_LAUNCH_CHECK();
  return 0;
}
extern "C" int avt_xent_bwd(const float* logits, int ld, const long* target, const float* lse, const float* gout, void* dlogits,
                            int ldd, int R, int C, long ignore_index, void* stream) {
  AVT_CHECK(logits && target && lse && gout && dlogits && R > 0 && C > 0 && ld >= C && ldd >= C, "avt_xent_bwd: bad argument");
  hipLaunchKernelGGL(xent_bwd_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, logits, ld, target, lse, gout, (bf16_t*)dlogits, ldd, C, ignore_index);
  AVT_LAUNCH_CHECK();
  return 0;
}
