// GPU input pipeline of the training loop: the reference's per-clip CPU transform chain
//   ToTensorVideo (uint8 (T,H,W,C) -> float (C,T,H,W) / 255)  -> Resize (bilinear, align_corners = False, shorter side to a
//   per-clip random size)  -> RandomHorizontalFlipVideo  -> x scale_pix_val  -> (reverse channels)  -> NormalizeVideo  ->
//   RandomCropVideo / CenterCropVideo                         (func/train.py:550-569, common/transforms.py:60-91, 124-170)
// as ONE kernel: every output pixel of the crop is traced back through crop, flip and resize to its four source pixels, so the
// resized intermediate never exists.  One thread per output pixel (all three channels: the source is channel-interleaved),
// writes are coalesced along x into the (B, T, 3, 1, OH, OW) fp32 batch the backbone's patch embedding reads.  HBM-bound and
// tiny next to the model (0.95 MB per frame).  The random draws stay on the host (params), as in the reference.
// Every OUTPUT clip has its own parameter row {new_h, new_w, flip, crop_i, crop_j, source clip}: the evaluation transform
// MultiCropVideo (common/transforms.py:254-296: top-left / centre / bottom-right crops, optionally followed by their mirror
// images) is 3 or 6 output clips reading one source clip -- a mirrored crop at column j is the crop at new_w - crop_w - j of
// the mirrored frame.
// quantize_u8: the training chain runs ColorJitterVideo between the flip and the scaling (func/train.py:554-557) with all four
// strengths 0 in every AVT experiment (conf/data/default.yaml:37-40).  torchvision's ColorJitter then changes nothing, but the
// wrapper (common/transforms.py:399-421) still converts the resized float clip to a PIL image and back: torchvision 0.8.2
// to_pil_image does pic.mul(255).byte() (truncation), to_tensor divides by 255 -- the resized pixels are cut to 8 bits.
#include "common.hpp"
#include "../../include/avt_hip.h"

namespace {
__global__ __launch_bounds__(256) void video_preproc_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst,
                                                            const int* __restrict__ params, int T, int H, int W, int OH, int OW,
                                                            float scale_pix, float m0, float m1, float m2, float is0, float is1,
                                                            float is2, int reverse, int quantize, long total) {
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int x = (int)(idx % OW);
    long r = idx / OW;
    const int y = (int)(r % OH); r /= OH;
    const int t = (int)(r % T);
    const int b = (int)(r / T);
    const int* pp = params + b * 6;
    const int new_h = pp[0], new_w = pp[1], flip = pp[2], ci = pp[3], cj = pp[4], sb = pp[5];      // sb: source clip (several crops may share one)
    const int yr = y + ci;
    int xr = x + cj;
    if (flip) xr = new_w - 1 - xr;
    // torch.nn.functional.interpolate(mode='bilinear', align_corners=False) with an explicit size: scale = in / out,
    // source = scale * (dst + 0.5) - 0.5 clamped at 0, upper neighbour clamped at the edge
    const float sy = fmaxf(((float)H / (float)new_h) * ((float)yr + 0.5f) - 0.5f, 0.f);
    const float sx = fmaxf(((float)W / (float)new_w) * ((float)xr + 0.5f) - 0.5f, 0.f);
    const int y0 = min((int)sy, H - 1), x0 = min((int)sx, W - 1);
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const uint8_t* f = src + ((size_t)sb * T + t) * (size_t)H * W * 3;
    const uint8_t* p00 = f + ((size_t)y0 * W + x0) * 3;
    const uint8_t* p01 = f + ((size_t)y0 * W + x1) * 3;
    const uint8_t* p10 = f + ((size_t)y1 * W + x0) * 3;
    const uint8_t* p11 = f + ((size_t)y1 * W + x1) * 3;
    float* o = dst + (((size_t)b * T + t) * 3) * (size_t)OH * OW + (size_t)y * OW + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int sc = reverse ? 2 - c : c;
      const float k = 1.f / 255.f;
      float v = hy * (hx * (p00[sc] * k) + lx * (p01[sc] * k)) + ly * (hx * (p10[sc] * k) + lx * (p11[sc] * k));
      if (quantize) v = floorf(v * 255.f) / 255.f;
      const float m = c == 0 ? m0 : (c == 1 ? m1 : m2), is = c == 0 ? is0 : (c == 1 ? is1 : is2);
      o[(size_t)c * OH * OW] = (v * scale_pix - m) * is;
    }
  }
}
}  // namespace

extern "C" int avt_video_preproc_u8(const void* src, float* dst, const int* params, int B, int T, int H, int W, int OH, int OW,
                                    float scale_pix, const float* mean3, const float* std3, int reverse_channels, int quantize_u8,
                                    void* stream) {
  AVT_CHECK(src && dst && params && mean3 && std3, "avt_video_preproc_u8: null argument");
  AVT_CHECK(B > 0 && T > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "avt_video_preproc_u8: bad shape");
  AVT_CHECK(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, "avt_video_preproc_u8: zero std");
  const long total = (long)B * T * OH * OW;
  long g = (total + 255) / 256; if (g > 16384) g = 16384;
  hipLaunchKernelGGL(video_preproc_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src, dst, params, T, H, W,
                     OH, OW, scale_pix, mean3[0], mean3[1], mean3[2], 1.f / std3[0], 1.f / std3[1], 1.f / std3[2], reverse_channels, quantize_u8, total);
  AVT_LAUNCH_CHECK();
  return 0;
}
