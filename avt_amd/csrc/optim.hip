// Fused SGD (momentum, Nesterov, weight decay) over flat fp32 buffers -- torch.optim.SGD semantics
// (conf/opt/optimizer/sgd.yaml, expts/01_ek100_avt.txt:26-28):
//     g = grad * grad_scale + wd * p ;  buf = first ? g : mom * buf + g ;  p -= lr * (nesterov ? g + mom * buf : buf)
// One pass emits the updated fp32 master weights, the momentum buffer, the bf16 shadow copy the GEMMs read, and
// (optionally) re-zeroes the gradient buffer for the next step's atomic accumulation.
// HBM-bound: 4+4+4 B read, 4+4+2(+4) B written per parameter.
#include "common.hpp"
#include "../../include/avt_hip.h"

namespace {
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ buf,
                                                  bf16_t* __restrict__ shadow, long n, float lr, float mom, float wd,
                                                  float grad_scale, int nesterov, int first, int zero_grad, const float* __restrict__ lr_dev) {
  if (lr_dev) lr = *lr_dev;               // captured steps: the learning rate lives in device memory, rewritten between replays (avt_sgd_step_dev)
  long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * 256 * 4;
  for (; i + 4 <= n; i += stride) {
    f32x4_t pv = *(f32x4_t*)(p + i), gv = *(f32x4_t*)(g + i), bv = {0.f, 0.f, 0.f, 0.f};
    if (!first) bv = *(f32x4_t*)(buf + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float gg = gv[e] * grad_scale + wd * pv[e];
      float b = first ? gg : mom * bv[e] + gg;
      float step = nesterov ? gg + mom * b : b;
      bv[e] = b;
      pv[e] = pv[e] - lr * step;
    }
    *(f32x4_t*)(p + i) = pv;
    *(f32x4_t*)(buf + i) = bv;
    if (shadow) { u32x2_t w; w[0] = pack2bf(pv[0], pv[1]); w[1] = pack2bf(pv[2], pv[3]); *(u32x2_t*)(shadow + i) = w; }
    if (zero_grad) *(f32x4_t*)(g + i) = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  if (i < n) {
    for (long j = i; j < n; ++j) {
      float gg = g[j] * grad_scale + wd * p[j];
      float b = first ? gg : mom * buf[j] + gg;
      float step = nesterov ? gg + mom * b : b;
      buf[j] = b; p[j] -= lr * step;
      if (shadow) shadow[j] = f2bf(p[j]);
      if (zero_grad) g[j] = 0.f;
    }
  }
}
}  // namespace

static int sgd_launch(float* param, float* grad, float* momentum_buf, void* shadow_bf16, long n, float lr, const float* lr_dev, float momentum,
                      float weight_decay, float grad_scale, int nesterov, int first_step, int zero_grad, void* stream) {
  AVT_CHECK(param && grad && momentum_buf && n > 0, "avt_sgd_step: null argument");
  AVT_CHECK(aligned16(param) && aligned16(grad) && aligned16(momentum_buf) && (!shadow_bf16 || (((uintptr_t)shadow_bf16) & 7) == 0),
            "avt_sgd_step: buffers must be 16-byte aligned");
  long g = (n / 4 + 255) / 256; if (g > 4096) g = 4096; if (g < 1) g = 1;
  hipLaunchKernelGGL(sgd_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, param, grad, momentum_buf, (bf16_t*)shadow_bf16, n, lr,
                     momentum, weight_decay, grad_scale, nesterov, first_step, zero_grad, lr_dev);
  AVT_LAUNCH_CHECK();
  return 0;
}

extern "C" int avt_sgd_step(float* param, float* grad, float* momentum_buf, void* shadow_bf16, long n, float lr, float momentum,
                            float weight_decay, float grad_scale, int nesterov, int first_step, int zero_grad, void* stream) {
  return sgd_launch(param, grad, momentum_buf, shadow_bf16, n, lr, nullptr, momentum, weight_decay, grad_scale, nesterov, first_step, zero_grad, stream);
}

extern "C" int avt_sgd_step_dev(float* param, float* grad, float* momentum_buf, void* shadow_bf16, long n, const float* lr_dev, float momentum,
                                float weight_decay, float grad_scale, int nesterov, int first_step, int zero_grad, void* stream) {
  AVT_CHECK(lr_dev && (((uintptr_t)lr_dev) & 3) == 0, "avt_sgd_step_dev: lr_dev must be a device pointer to one float");
  return sgd_launch(param, grad, momentum_buf, shadow_bf16, n, 0.f, lr_dev, momentum, weight_decay, grad_scale, nesterov, first_step, zero_grad, stream);
}
