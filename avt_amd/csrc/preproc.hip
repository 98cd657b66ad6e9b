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
// No floating-point contraction anywhere in this file: every expression below restates a CPU computation (torch's bilinear kernel,
// Pillow's C) whose 8-bit cuts are sensitive to the last bit; the fused multiply-adds that ARE part of those computations are
// written out (__fmaf_rn).  hipcc contracts by default, through the __f*_rn intrinsics as well.
#pragma clang fp contract(off)

namespace {
// torch.nn.functional.interpolate(mode='bilinear', align_corners=False) with an explicit size, restated bit for bit as torch's CPU
// kernel evaluates it for float tensors (ATen UpSampleKernel.cpp, the separable Interpolate<2> loop as built with contraction on;
// pinned by tools/lab/interp_order.py against torch 2.10 on the build host: 0 differing floats over 7 M pixels):
//   source index  r = max(fma(in / out, dst + 0.5, -0.5), 0);  i0 = min(floor(r), in - 1);  w1 = clamp(r - i0, 0, 1);  w0 = 1 - w1
//   pixel         top = fma(p00, wx0, p01 * wx1);  bot = fma(p10, wx0, p11 * wx1);  out = fma(top, wy0, bot * wy1)
// on pixels that went through to_tensor's TRUE division by 255.  The 8-bit cut of the colour-jitter round trip
// (floor(v * 255)) sits right behind this, so a different association order puts pixels near an integer level one level off.
__device__ __forceinline__ void src_index(int in_size, int out_size, int dst, int& i0, int& i1, float& w0, float& w1) {
  const float r = fmaxf(__fmaf_rn((float)in_size / (float)out_size, (float)dst + 0.5f, -0.5f), 0.f);
  i0 = min((int)r, in_size - 1);
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  w1 = fminf(fmaxf(r - (float)i0, 0.f), 1.f);
  w0 = 1.f - w1;
}
__device__ __forceinline__ float bilerp_u8(int a00, int a01, int a10, int a11, float wx0, float wx1, float wy0, float wy1) {
  const float p00 = (float)a00 / 255.f, p01 = (float)a01 / 255.f, p10 = (float)a10 / 255.f, p11 = (float)a11 / 255.f;
  const float top = __fmaf_rn(p00, wx0, __fmul_rn(p01, wx1));
  const float bot = __fmaf_rn(p10, wx0, __fmul_rn(p11, wx1));
  return __fmaf_rn(top, wy0, __fmul_rn(bot, wy1));
}

// one output pixel (b, t, y, x) of the crop, all three channels: traced back through crop, flip and resize to its four source pixels
struct PreprocArgs { int T, H, W, OH, OW; float scale_pix, m0, m1, m2, is0, is1, is2; int reverse, quantize; };
__device__ __forceinline__ void preproc_pixel(const uint8_t* __restrict__ src, const int* __restrict__ pp, const PreprocArgs& a, int t, int y, int x, float (&out)[3]) {
  const int new_h = pp[0], new_w = pp[1], flip = pp[2], ci = pp[3], cj = pp[4], sb = pp[5];      // sb: source clip (several crops may share one)
  const int yr = y + ci;
  int xr = x + cj;
  if (flip) xr = new_w - 1 - xr;
  int y0, y1, x0, x1; float hy, ly, hx, lx;
  src_index(a.H, new_h, yr, y0, y1, hy, ly);
  src_index(a.W, new_w, xr, x0, x1, hx, lx);
  const uint8_t* f = src + ((size_t)sb * a.T + t) * (size_t)a.H * a.W * 3;
  const uint8_t* p00 = f + ((size_t)y0 * a.W + x0) * 3;
  const uint8_t* p01 = f + ((size_t)y0 * a.W + x1) * 3;
  const uint8_t* p10 = f + ((size_t)y1 * a.W + x0) * 3;
  const uint8_t* p11 = f + ((size_t)y1 * a.W + x1) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int sc = a.reverse ? 2 - c : c;
    float v = bilerp_u8(p00[sc], p01[sc], p10[sc], p11[sc], hx, lx, hy, ly);
    if (a.quantize) v = floorf(__fmul_rn(v, 255.f)) / 255.f;
    const float m = c == 0 ? a.m0 : (c == 1 ? a.m1 : a.m2), is = c == 0 ? a.is0 : (c == 1 ? a.is1 : a.is2);
    out[c] = (v * a.scale_pix - m) * is;
  }
}

__global__ __launch_bounds__(256) void video_preproc_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst,
                                                            const int* __restrict__ params, PreprocArgs a, long total) {
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int x = (int)(idx % a.OW);
    long r = idx / a.OW;
    const int y = (int)(r % a.OH); r /= a.OH;
    const int t = (int)(r % a.T);
    const int b = (int)(r / a.T);
    float v[3];
    preproc_pixel(src, params + b * 6, a, t, y, x, v);
    float* o = dst + (((size_t)b * a.T + t) * 3) * (size_t)a.OH * a.OW + (size_t)y * a.OW + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[(size_t)c * a.OH * a.OW] = v[c];
  }
}

// The same pixels written straight as the patch-embedding GEMM's rows (round 6): bf16 [frames * (P + 1)][768], row n (P + 1) = the zero CLS slot,
// row n (P + 1) + 1 + py PW + px = patch (py, px) flattened as k = c 256 + ky 16 + kx (the Conv2d weight's (3, 16, 16) order) -- exactly what
// avt_im2col_patch16 makes of the fp32 frames (same fp32 value, same round-to-nearest-even), without the 0.95 MB per frame of fp32 in between.
// One thread = 8 consecutive pixels of one patch row (ky) x 3 channels: three 16-byte stores; a patch's 16 ky x 2 halves are 32 neighbouring
// threads, so each channel's stores of a patch cover 512 contiguous bytes.  PIX(t, y, x, v[3]) yields the normalised fp32 pixel.
template <typename PIX>
__device__ __forceinline__ void emit_patch_rows(bf16_t* __restrict__ patches, int T, int OH, int OW, long total, PIX pix) {
  const int PW = OW / 16, P = PW * (OH / 16);
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int half = (int)(idx & 1), ky = (int)((idx >> 1) & 15);
    const long row = idx >> 5;                                  // frame * (P + 1) + s
    const int s = (int)(row % (P + 1));
    const long n = row / (P + 1);                               // frame = b * T + t
    u32x4_t w[3] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    if (s > 0) {
      const int p = s - 1, py = p / PW, px = p % PW;
      const int b = (int)(n / T), t = (int)(n % T);
      const int y = py * 16 + ky, x0 = px * 16 + half * 8;
      float v[8][3];
#pragma unroll
      for (int i = 0; i < 8; ++i) pix(b, t, y, x0 + i, v[i]);
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) w[c][i] = pack2bf(v[2 * i][c], v[2 * i + 1][c]);
    }
    bf16_t* dst = patches + (size_t)row * 768 + ky * 16 + half * 8;
#pragma unroll
    for (int c = 0; c < 3; ++c) *(u32x4_t*)(dst + c * 256) = w[c];
  }
}
__global__ __launch_bounds__(256) void video_preproc_patches_kernel(const uint8_t* __restrict__ src, bf16_t* __restrict__ patches,
                                                                    const int* __restrict__ params, PreprocArgs a, long total) {
  emit_patch_rows(patches, a.T, a.OH, a.OW, total, [&](int b, int t, int y, int x, float (&v)[3]) { preproc_pixel(src, params + b * 6, a, t, y, x, v); });
}
// ---- ColorJitterVideo with non-zero strengths (common/transforms.py:399-421) ------------------------------------------------------
// The reference converts the flipped, resized clip -- all frames stacked into one tall image -- to an 8-bit PIL image, runs
// torchvision 0.8.2's ColorJitter on it (four Pillow operations in a drawn order with drawn factors) and converts back.  The
// contrast step blends with the mean grey of the WHOLE stacked image, so the resized clip has to exist: three stages instead of
// the single fused kernel --
//   1. resize + flip + floor(v * 255)            -> 8-bit scratch clip [clip][T][new_h][new_w][3] (pitch SH x SW)
//   2. up to four in-place passes, op and factor per clip: brightness / contrast / saturation = Pillow's Image.blend(degenerate,
//      image, factor) (Blend.c: float32, truncation inside [0, 1], clipping outside) with degenerate = 0 | round(mean luma of the
//      clip) | the pixel's luma ((19595 R + 38470 G + 7471 B + 0x8000) >> 16); hue = Pillow's rgb2hsv_row / hsv2rgb_row
//      (Convert.c) around an 8-bit wrap-around add of (int)(factor * 255) & 255 (handed in as the op's factor).  The clip's luma sum is an
//      exact 64-bit integer sum.
//   3. / 255, x scale_pix, (reverse channels), normalise, crop -> fp32.
// (No floating-point contraction, see the top of the file: a fused multiply-add rounds differently from Pillow's C -- measured:
// saturation was one level off in 0.1 % of the pixels.)
__device__ __forceinline__ int luma_u8(int r, int g, int b) { return (int)(((unsigned)r * 19595u + (unsigned)g * 38470u + (unsigned)b * 7471u + 0x8000u) >> 16); }
__device__ __forceinline__ int blend_u8(int d, int im, float f) {
  const float t = ((float)d + (f * (float)(im - d)));
  if (f >= 0.f && f <= 1.f) return (int)t;                        // (UINT8) of a value inside [0, 255]
  return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}
__device__ __forceinline__ void rgb2hsv_u8(int r, int g, int b, int& uh, int& us, int& uv) {
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  uv = maxc;
  if (minc == maxc) { uh = 0; us = 0; return; }
  const float cr = (float)(maxc - minc);
  const float s = (cr / (float)maxc);
  const float rc = ((float)(maxc - r) / cr), gc = ((float)(maxc - g) / cr), bc = ((float)(maxc - b) / cr);
  float h;
  if (r == maxc) h = (bc - gc);
  else if (g == maxc) h = (float)((2.0 + (double)rc) - (double)bc);
  else h = (float)((4.0 + (double)gc) - (double)rc);
  h = (float)fmod((((double)h / 6.0) + 1.0), 1.0);
  uh = min(max((int)((double)h * 255.0), 0), 255);
  us = min(max((int)((double)s * 255.0), 0), 255);
}
// hsv2rgb_row with C's promotions spelled out (round-3 advisor finding: an all-fp32 version with rintf was one level off on 2 of the
// 2^24 triples): (float)h * 6.0 / 255.0 is double arithmetic, f and fs are stored as floats, fs * f is a float product, the
// arguments of round() are doubles and round() is half away from zero
__device__ __forceinline__ int round_u8(double x) { return min(max((int)round(x), 0), 255); }
__device__ __forceinline__ void hsv2rgb_u8(int uh, int us, int uv, int& r, int& g, int& b) {
  if (us == 0) { r = g = b = uv; return; }
  const double hd = (((double)(float)uh * 6.0) / 255.0);
  const int i = (int)floor(hd);
  const float f = (float)(hd - (double)(float)i);
  const float fs = (float)((double)(float)us / 255.0);
  const double fv = (double)(float)uv;
  const int p = round_u8((fv * (1.0 - (double)fs)));
  const int q = round_u8((fv * (1.0 - (double)(fs * f))));
  const int t = round_u8((fv * (1.0 - ((double)fs * (1.0 - (double)f)))));
  switch (i % 6) {
    case 0: r = uv; g = t; b = p; break;
    case 1: r = q; g = uv; b = p; break;
    case 2: r = p; g = uv; b = t; break;
    case 3: r = p; g = q; b = uv; break;
    case 4: r = t; g = p; b = uv; break;
    default: r = uv; g = p; b = q; break;
  }
}

__global__ __launch_bounds__(256) void resize_flip_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ scratch,
                                                             const int* __restrict__ params, int T, int H, int W, int SH, int SW, long total) {
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int x = (int)(idx % SW);
    long r = idx / SW;
    const int y = (int)(r % SH); r /= SH;
    const int t = (int)(r % T);
    const int b = (int)(r / T);
    const int* pp = params + b * 6;
    const int new_h = pp[0], new_w = pp[1], flip = pp[2], sb = pp[5];
    if (y >= new_h || x >= new_w) continue;
    const int xr = flip ? new_w - 1 - x : x;
    int y0, y1, x0, x1; float hy, ly, hx, lx;
    src_index(H, new_h, y, y0, y1, hy, ly);
    src_index(W, new_w, xr, x0, x1, hx, lx);
    const uint8_t* f = src + ((size_t)sb * T + t) * (size_t)H * W * 3;
    const uint8_t* p00 = f + ((size_t)y0 * W + x0) * 3;
    const uint8_t* p01 = f + ((size_t)y0 * W + x1) * 3;
    const uint8_t* p10 = f + ((size_t)y1 * W + x0) * 3;
    const uint8_t* p11 = f + ((size_t)y1 * W + x1) * 3;
    uint8_t* o = scratch + ((((size_t)b * T + t) * SH + y) * SW + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = bilerp_u8(p00[c], p01[c], p10[c], p11[c], hx, lx, hy, ly);
      o[c] = (uint8_t)(int)floorf(__fmul_rn(v, 255.f));
    }
  }
}
// exact per-clip sum of the luma of every pixel (for the clips whose op of this slot is contrast)
__global__ __launch_bounds__(256) void luma_sum_kernel(const uint8_t* __restrict__ scratch, const int* __restrict__ params, const int* __restrict__ ops,
                                                       int slot, unsigned long long* __restrict__ sums, int T, int SH, int SW, int blocks_per_clip) {
  const int b = blockIdx.x / blocks_per_clip, part = blockIdx.x % blocks_per_clip;
  if (ops[b * 4 + slot] != 1) return;
  const int new_h = params[b * 6], new_w = params[b * 6 + 1];
  const long n = (long)T * new_h * new_w;
  unsigned long long acc = 0;
  for (long i = (long)part * 256 + threadIdx.x; i < n; i += (long)blocks_per_clip * 256) {
    const int x = (int)(i % new_w);
    const long r = i / new_w;
    const int y = (int)(r % new_h), t = (int)(r / new_h);
    const uint8_t* px = scratch + ((((size_t)b * T + t) * SH + y) * SW + x) * 3;
    acc += (unsigned long long)luma_u8(px[0], px[1], px[2]);
  }
  __shared__ unsigned long long sh[256];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) atomicAdd(&sums[b], sh[0]);               // integer atomics: exact, order-independent
}
// op ids: 0 brightness, 1 contrast, 2 saturation, 3 hue, < 0 none
__global__ __launch_bounds__(256) void jitter_op_kernel(uint8_t* __restrict__ scratch, const int* __restrict__ params, const int* __restrict__ ops,
                                                        const float* __restrict__ factors, int slot, const unsigned long long* __restrict__ sums,
                                                        int T, int SH, int SW, long total) {
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int x = (int)(idx % SW);
    long r = idx / SW;
    const int y = (int)(r % SH); r /= SH;
    const int t = (int)(r % T);
    const int b = (int)(r / T);
    const int op = ops[b * 4 + slot];
    const int new_h = params[b * 6], new_w = params[b * 6 + 1];
    if (op < 0 || y >= new_h || x >= new_w) continue;
    const float f = factors[b * 4 + slot];
    uint8_t* px = scratch + ((((size_t)b * T + t) * SH + y) * SW + x) * 3;
    int R = px[0], G = px[1], B = px[2];
    if (op == 0) { R = blend_u8(0, R, f); G = blend_u8(0, G, f); B = blend_u8(0, B, f); }
    else if (op == 1) {
      const double n = (double)T * (double)new_h * (double)new_w;
      const int mean = (int)((double)sums[b] / n + 0.5);
      R = blend_u8(mean, R, f); G = blend_u8(mean, G, f); B = blend_u8(mean, B, f);
    } else if (op == 2) {
      const int l = luma_u8(R, G, B);
      R = blend_u8(l, R, f); G = blend_u8(l, G, f); B = blend_u8(l, B, f);
    } else {
      int uh, us, uv;
      rgb2hsv_u8(R, G, B, uh, us, uv);
      uh = (uh + (int)f) & 255;                                    // np_h += np.uint8(hue_factor * 255), 8-bit wrap-around: the caller passes the shift
      hsv2rgb_u8(uh, us, uv, R, G, B);
    }
    px[0] = (uint8_t)R; px[1] = (uint8_t)G; px[2] = (uint8_t)B;
  }
}
__device__ __forceinline__ void crop_norm_pixel(const uint8_t* __restrict__ scratch, const int* __restrict__ pp, const PreprocArgs& a, int SH, int SW,
                                                int b, int t, int y, int x, float (&out)[3]) {
  const int ci = pp[3], cj = pp[4];
  const uint8_t* px = scratch + ((((size_t)b * a.T + t) * SH + (y + ci)) * SW + (x + cj)) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = (float)px[a.reverse ? 2 - c : c] / 255.f;
    const float m = c == 0 ? a.m0 : (c == 1 ? a.m1 : a.m2), is = c == 0 ? a.is0 : (c == 1 ? a.is1 : a.is2);
    out[c] = (v * a.scale_pix - m) * is;
  }
}
__global__ __launch_bounds__(256) void crop_norm_u8_kernel(const uint8_t* __restrict__ scratch, float* __restrict__ dst, const int* __restrict__ params,
                                                           PreprocArgs a, int SH, int SW, long total) {
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int x = (int)(idx % a.OW);
    long r = idx / a.OW;
    const int y = (int)(r % a.OH); r /= a.OH;
    const int t = (int)(r % a.T);
    const int b = (int)(r / a.T);
    float v[3];
    crop_norm_pixel(scratch, params + b * 6, a, SH, SW, b, t, y, x, v);
    float* o = dst + (((size_t)b * a.T + t) * 3) * (size_t)a.OH * a.OW + (size_t)y * a.OW + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[(size_t)c * a.OH * a.OW] = v[c];
  }
}
__global__ __launch_bounds__(256) void crop_norm_u8_patches_kernel(const uint8_t* __restrict__ scratch, bf16_t* __restrict__ patches, const int* __restrict__ params,
                                                                   PreprocArgs a, int SH, int SW, long total) {
  emit_patch_rows(patches, a.T, a.OH, a.OW, total, [&](int b, int t, int y, int x, float (&v)[3]) { crop_norm_pixel(scratch, params + b * 6, a, SH, SW, b, t, y, x, v); });
}
}  // namespace

extern "C" size_t avt_video_jitter_scratch_bytes(int B, int T, int max_h, int max_w) { return (size_t)B * T * max_h * max_w * 3 + 256; }

extern "C" int avt_video_preproc_jitter_u8(const void* src, float* dst, void* patches, const int* params, const int* jitter_ops, const float* jitter_factors,
                                           int B, int T, int H, int W, int OH, int OW, int max_h, int max_w, float scale_pix,
                                           const float* mean3, const float* std3, int reverse_channels, int slot_mask,
                                           void* scratch, size_t scratch_bytes, unsigned long long* luma_sums, void* stream) {
  AVT_CHECK(src && (dst || patches) && params && jitter_ops && jitter_factors && mean3 && std3 && scratch && luma_sums, "avt_video_preproc_jitter_u8: null argument");
  AVT_CHECK(B > 0 && T > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && max_h >= OH && max_w >= OW, "avt_video_preproc_jitter_u8: bad shape");
  AVT_CHECK(!patches || (OH % 16 == 0 && OW % 16 == 0 && aligned16(patches)), "avt_video_preproc_jitter_u8: patch rows need a crop that is a multiple of 16 and a 16-byte aligned buffer");
  AVT_CHECK(scratch_bytes >= (size_t)B * T * max_h * max_w * 3, "avt_video_preproc_jitter_u8: scratch too small (%zu bytes needed)", (size_t)B * T * max_h * max_w * 3);
  AVT_CHECK(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, "avt_video_preproc_jitter_u8: zero std");
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)B * T * max_h * max_w;
  long g = (total + 255) / 256; if (g > 16384) g = 16384;
  hipLaunchKernelGGL(resize_flip_u8_kernel, dim3((int)g), dim3(256), 0, s, (const uint8_t*)src, (uint8_t*)scratch, params, T, H, W, max_h, max_w, total);
  for (int slot = 0; slot < 4; ++slot) {
    // slot_mask (host knowledge of the device-side op table): bit s = some clip has an operation in slot s, bit 4 + s = some clip's
    // operation in slot s is contrast (needs the clip's mean luma).  A caller that does not know passes 0xff.
    if (!((slot_mask >> slot) & 1)) continue;
    if ((slot_mask >> (4 + slot)) & 1) {
      (void)hipMemsetAsync(luma_sums, 0, (size_t)B * sizeof(unsigned long long), s);
      const int bpc = 64;
      hipLaunchKernelGGL(luma_sum_kernel, dim3(B * bpc), dim3(256), 0, s, (const uint8_t*)scratch, params, jitter_ops, slot, luma_sums, T, max_h, max_w, bpc);
    }
    hipLaunchKernelGGL(jitter_op_kernel, dim3((int)g), dim3(256), 0, s, (uint8_t*)scratch, params, jitter_ops, jitter_factors, slot, luma_sums, T, max_h, max_w, total);
  }
  const PreprocArgs a{T, H, W, OH, OW, scale_pix, mean3[0], mean3[1], mean3[2], 1.f / std3[0], 1.f / std3[1], 1.f / std3[2], reverse_channels, 0};
  if (dst) {
    const long tot_out = (long)B * T * OH * OW;
    long g2 = (tot_out + 255) / 256; if (g2 > 16384) g2 = 16384;
    hipLaunchKernelGGL(crop_norm_u8_kernel, dim3((int)g2), dim3(256), 0, s, (const uint8_t*)scratch, dst, params, a, max_h, max_w, tot_out);
  }
  if (patches) {
    const long tot_p = (long)B * T * ((OH / 16) * (OW / 16) + 1) * 32;
    long g3 = (tot_p + 255) / 256; if (g3 > 16384) g3 = 16384;
    hipLaunchKernelGGL(crop_norm_u8_patches_kernel, dim3((int)g3), dim3(256), 0, s, (const uint8_t*)scratch, (bf16_t*)patches, params, a, max_h, max_w, tot_p);
  }
  AVT_LAUNCH_CHECK();
  return 0;
}


extern "C" int avt_video_preproc_u8(const void* src, float* dst, void* patches, const int* params, int B, int T, int H, int W, int OH, int OW,
                                    float scale_pix, const float* mean3, const float* std3, int reverse_channels, int quantize_u8,
                                    void* stream) {
  AVT_CHECK(src && (dst || patches) && params && mean3 && std3, "avt_video_preproc_u8: null argument");
  AVT_CHECK(B > 0 && T > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "avt_video_preproc_u8: bad shape");
  AVT_CHECK(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, "avt_video_preproc_u8: zero std");
  AVT_CHECK(!patches || (OH % 16 == 0 && OW % 16 == 0 && aligned16(patches)), "avt_video_preproc_u8: patch rows need a crop that is a multiple of 16 and a 16-byte aligned buffer");
  const PreprocArgs a{T, H, W, OH, OW, scale_pix, mean3[0], mean3[1], mean3[2], 1.f / std3[0], 1.f / std3[1], 1.f / std3[2], reverse_channels, quantize_u8};
  if (dst) {
    const long total = (long)B * T * OH * OW;
    long g = (total + 255) / 256; if (g > 16384) g = 16384;
    hipLaunchKernelGGL(video_preproc_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src, dst, params, a, total);
  }
  if (patches) {
    const long tot_p = (long)B * T * ((OH / 16) * (OW / 16) + 1) * 32;
    long g = (tot_p + 255) / 256; if (g > 16384) g = 16384;
    hipLaunchKernelGGL(video_preproc_patches_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src, (bf16_t*)patches, params, a, tot_p);
  }
  AVT_LAUNCH_CHECK();
  return 0;
}
