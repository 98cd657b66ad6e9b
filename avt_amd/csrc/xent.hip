// Softmax cross-entropy over C classes with ignore_index (loss_fn/multidim_xentropy.py:11-25 ->
// torch.nn.CrossEntropyLoss(ignore_index=-1, reduction='none')) plus the rank of the target logit (top-1 / top-5
// accuracy, common/utils.py:17-44).  One workgroup per row; logits fp32 [R, ld] (columns >= C are padding).
//   forward : loss[r] = lse[r] - logit[r, t]  (0 when t == ignore_index); rank[r] = #{c : logit[c] > logit[t]}
//   backward: dlogits[r, c] = (exp(logit - lse) - [c == t]) * gout[r]  (0 for ignored rows and padding columns)
//             [+ gextra[r, c]: a gradient that reached the logits from another consumer], written in bf16 for the classifier's
//             dgrad / wgrad GEMMs.
// The classifier + cross-entropy operator of the C ABI (SURVEY 8b linear_softmax_xent_{fwd,bwd}) lives here too: it is what the
// drop-in model's classifier node runs when the training operator hands it the targets (models/classifiers.py).
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
                                                       bf16_t* __restrict__ dlogits, int ldd, int C, long ignore_index,
                                                       const float* __restrict__ gextra, int ldg) {
  const int r = blockIdx.x;
  const float* row = logits + (size_t)r * ld;
  const long t = target[r];
  const bool valid = (t != ignore_index) && t >= 0 && t < C;
  const float g = valid ? gout[r] : 0.f, l = lse[r];
  bf16_t* drow = dlogits + (size_t)r * ldd;
  for (int c = threadIdx.x; c < ldd; c += 256) {
    float v = 0.f;
    if (valid && c < C) v = (__expf(row[c] - l) - ((long)c == t ? 1.f : 0.f)) * g;
    if (gextra && c < C) v += gextra[(size_t)r * ldg + c];
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
  hipLaunchKernelGGL(xent_bwd_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, logits, ld, target, lse, gout, (bf16_t*)dlogits, ldd, C, ignore_index,
                     (const float*)nullptr, 0);
  AVT_LAUNCH_CHECK();
  return 0;
}

// ---- classifier + softmax cross-entropy as ONE operator ---------------------------------------------------------------------
// What the reference computes with torch.nn.Linear (models/base_model.py:203-216) followed by MultiDimCrossEntropy
// (loss_fn/multidim_xentropy.py:11-25, ignore_index, reduction 'none') and, in backward, their two autograd nodes.
// Forward: logits = x W^T + b (fp32, kept: they are a model output) -> loss, log-sum-exp, rank of the target.
// Backward: (softmax - onehot) * gloss (+ any gradient that reached the logits themselves) is produced ONCE, in the bf16
// class-padded layout the three GEMM-side consumers read (dW = dlogits^T x into the fp32 gradient, db = column sums,
// dx = dlogits W): the fp32 dlogits tensor of the two-node path, its slice / pad / add and its re-cast never exist.
// Runs on the caller's stream; allocates nothing.
extern "C" int avt_linear_softmax_xent_fwd(const void* x, int ldx, const void* w, int ldw, const float* bias, const long* target,
                                           float* logits, int ldl, float* loss, float* lse, int* rank,
                                           int R, int C, int Cpad, int K, long ignore_index, void* stream) {
  AVT_CHECK(x && w && target && logits && loss && lse, "avt_linear_softmax_xent_fwd: null argument");
  AVT_CHECK(R > 0 && C > 0 && Cpad >= C && Cpad % 8 == 0 && ldl >= Cpad && K > 0, "avt_linear_softmax_xent_fwd: bad shape (R=%d C=%d Cpad=%d K=%d ldl=%d)", R, C, Cpad, K, ldl);
  int rc = avt_gemm_bf16(x, 1, ldx, w, 1, ldw, logits, ldl, R, Cpad, K, bias, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, 0.f, 0, nullptr,
                         /*out_mode fp32*/ 1, 0, 0, nullptr, 0, stream);
  if (rc) return rc;
  return avt_xent_fwd(logits, ldl, target, loss, lse, rank, R, C, ignore_index, stream);
}

extern "C" int avt_linear_softmax_xent_bwd(const float* logits, int ldl, const long* target, const float* lse, const float* gloss,
                                           const float* glogits, int ldg,
                                           const void* x, int ldx, const void* w, int ldw, void* dlogits_bf16,
                                           float* dw, int lddw, float* dbias, void* dx, int lddx, int dx_f32,
                                           float dx_drop_p, uint64_t dx_drop_seed,
                                           int R, int C, int Cpad, int K, long ignore_index,
                                           void* workspace, size_t workspace_bytes, float* partials, size_t partials_bytes, void* stream) {
  AVT_CHECK(logits && target && lse && gloss && x && w && dlogits_bf16, "avt_linear_softmax_xent_bwd: null argument");
  AVT_CHECK(R > 0 && C > 0 && Cpad >= C && Cpad % 8 == 0 && K > 0 && ldl >= C && (!glogits || ldg >= C), "avt_linear_softmax_xent_bwd: bad shape");
  // dlogits[R, Cpad] bf16, padding columns written as zeros
  hipLaunchKernelGGL(xent_bwd_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, logits, ldl, target, lse, gloss, (bf16_t*)dlogits_bf16, Cpad, C,
                     ignore_index, glogits, ldg);
  AVT_LAUNCH_CHECK();
  int rc = 0;
  if (dw) {          // dW[Cpad, K] += dlogits^T x : both operands stored reduction-index (row) major
    AVT_CHECK(workspace, "avt_linear_softmax_xent_bwd: the weight gradient needs the split-K workspace (avt_gemm_accum_workspace_bytes(Cpad, K, R))");
    rc = avt_gemm_accum_bf16(dlogits_bf16, Cpad, x, ldx, dw, lddw, Cpad, K, R, 0, 0, workspace, workspace_bytes, stream);
    if (rc) return rc;
  }
  if (dbias) {
    rc = avt_colsum_bf16(dlogits_bf16, Cpad, dbias, R, Cpad, partials, partials_bytes, stream);
    if (rc) return rc;
  }
  if (dx)            // dx[R, K] = dlogits[R, Cpad] W[Cpad, K] (W stored with the reduction index as its row index), optionally through the
                     // mask of the dropout that preceded the classifier (counter-based: keep(seed, r * K + k), common.hpp)
    rc = avt_gemm_bf16(dlogits_bf16, 1, Cpad, w, 0, ldw, dx, lddx, R, K, Cpad, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0,
                       dx_drop_p, dx_drop_seed, nullptr, dx_f32 ? 1 : 0, 0, 0, nullptr, 0, stream);
  return rc;
}
