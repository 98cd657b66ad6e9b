// Classifier + softmax cross-entropy as ONE operator of the C ABI (SURVEY 8b `linear_softmax_xent_{fwd,bwd}`): what the reference
// computes with torch.nn.Linear (models/base_model.py:203-216) followed by MultiDimCrossEntropy (loss_fn/multidim_xentropy.py:11-25,
// ignore_index, reduction 'none') and, in backward, their two autograd nodes.  Forward: logits = x W^T + b (fp32, kept: they are a
// model output) -> loss, log-sum-exp, rank of the target.  Backward: (softmax - onehot) * gloss is produced directly in the bf16,
// class-padded layout the three GEMM-side consumers read (dW = dlogits^T x into the fp32 gradient, db = column sums, dx = dlogits W):
// the fp32 dlogits tensor and its re-cast of the two-node path never exist.  Sequences the kernels of gemm.hip / xent.hip /
// elementwise.hip on the caller's stream; allocates nothing.
#include "common.hpp"
#include "../../include/avt_hip.h"

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
                                           const void* x, int ldx, const void* w, int ldw, void* dlogits_bf16,
                                           float* dw, int lddw, float* dbias, void* dx, int lddx, int dx_f32,
                                           int R, int C, int Cpad, int K, long ignore_index,
                                           void* workspace, size_t workspace_bytes, float* partials, size_t partials_bytes, void* stream) {
  AVT_CHECK(logits && target && lse && gloss && x && w && dlogits_bf16, "avt_linear_softmax_xent_bwd: null argument");
  AVT_CHECK(R > 0 && C > 0 && Cpad >= C && Cpad % 8 == 0 && K > 0, "avt_linear_softmax_xent_bwd: bad shape");
  // dlogits[R, Cpad] bf16, padding columns written as zeros
  int rc = avt_xent_bwd(logits, ldl, target, lse, gloss, dlogits_bf16, Cpad, R, C, ignore_index, stream);
  if (rc) return rc;
  if (dw) {          // dW[Cpad, K] += dlogits^T x : both operands stored reduction-index (row) major
    AVT_CHECK(workspace, "avt_linear_softmax_xent_bwd: the weight gradient needs the split-K workspace (avt_gemm_accum_workspace_bytes(Cpad, K, R))");
    rc = avt_gemm_accum_bf16(dlogits_bf16, Cpad, x, ldx, dw, lddw, Cpad, K, R, 0, 0, workspace, workspace_bytes, stream);
    if (rc) return rc;
  }
  if (dbias) {
    rc = avt_colsum_bf16(dlogits_bf16, Cpad, dbias, R, Cpad, partials, partials_bytes, stream);
    if (rc) return rc;
  }
  if (dx)            // dx[R, K] = dlogits[R, Cpad] W[Cpad, K]: W is stored with the reduction index as its row index
    rc = avt_gemm_bf16(dlogits_bf16, 1, Cpad, w, 0, ldw, dx, lddx, R, K, Cpad, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, 0.f, 0, nullptr,
                       dx_f32 ? 1 : 0, 0, 0, nullptr, 0, stream);
  return rc;
}
