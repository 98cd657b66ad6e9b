// HBM-bound helper kernels of the AVT step: patch extraction (im2col for the 16x16/stride-16 patch-embed conv),
// positional/CLS residual table, fp32->bf16 casts, dropout, GPT-2 position-embedding add, reductions that feed
// parameter gradients (pos_embed / cls_token / patch bias / wpe), shifted-MSE feature loss.
// All of them move 8-16 B per lane per access and touch every byte once.
#include "common.hpp"
#include "../../include/avt_hip.h"

namespace {

// video fp32 [N,3,Hi,Wi] -> patches bf16 [N*(P+1), 768]; row n*(P+1) is the (zero) CLS slot, row n*(P+1)+1+p is
// patch p = py*(Wi/16)+px flattened as k = c*256 + ky*16 + kx (the Conv2d weight's (3,16,16) order).
__global__ __launch_bounds__(256) void im2col16_kernel(const float* __restrict__ video, bf16_t* __restrict__ patches,
                                                       int N, int Hi, int Wi) {
  const int PW = Wi / 16, PH = Hi / 16, P = PW * PH;
  const long total = (long)N * (P + 1) * 48;           // 48 = 3 channels x 16 ky: one 16-pixel run each
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    int run = (int)(idx % 48);
    long row = idx / 48;
    int s = (int)(row % (P + 1));
    int n = (int)(row / (P + 1));
    u32x4_t w0 = {0u, 0u, 0u, 0u}, w1 = {0u, 0u, 0u, 0u};
    if (s > 0) {
      int p = s - 1, py = p / PW, px = p % PW, c = run >> 4, ky = run & 15;
      const float* src = video + (((size_t)n * 3 + c) * Hi + (py * 16 + ky)) * Wi + px * 16;
      f32x4_t a = *(const f32x4_t*)(src), b = *(const f32x4_t*)(src + 4), cc = *(const f32x4_t*)(src + 8), d = *(const f32x4_t*)(src + 12);
      w0[0] = pack2bf(a[0], a[1]); w0[1] = pack2bf(a[2], a[3]); w0[2] = pack2bf(b[0], b[1]); w0[3] = pack2bf(b[2], b[3]);
      w1[0] = pack2bf(cc[0], cc[1]); w1[1] = pack2bf(cc[2], cc[3]); w1[2] = pack2bf(d[0], d[1]); w1[3] = pack2bf(d[2], d[3]);
    }
    bf16_t* dst = patches + (size_t)row * 768 + run * 16;
    *(u32x4_t*)(dst) = w0;
    *(u32x4_t*)(dst + 8) = w1;
  }
}

// R[s] = pos[s] + (s == 0 ? cls : conv_bias)  (bf16) -- the row-periodic residual of the patch-embed GEMM
__global__ void posres_kernel(const float* __restrict__ pos, const float* __restrict__ cls, const float* __restrict__ bias,
                              bf16_t* __restrict__ R, int S, int D) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S * D) return;
  int s = i / D, d = i % D;
  R[i] = f2bf(pos[i] + (s == 0 ? cls[d] : bias[d]));
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
  long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
  const long stride = (long)gridDim.x * 256 * 8;
  for (; i + 8 <= n; i += stride) {
    f32x4_t a = *(const f32x4_t*)(src + i), b = *(const f32x4_t*)(src + i + 4);
    u32x4_t w; w[0] = pack2bf(a[0], a[1]); w[1] = pack2bf(a[2], a[3]); w[2] = pack2bf(b[0], b[1]); w[3] = pack2bf(b[2], b[3]);
    *(u32x4_t*)(dst + i) = w;
  }
  if (i < n) for (long j = i; j < n; ++j) dst[j] = f2bf(src[j]);
}

__global__ __launch_bounds__(256) void cast_back_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = bf2f(src[i]);
}

// y = keep(seed, idx) ? x / (1-p) : 0, bf16 -> bf16 (also its own backward on gradients, same seed)
__global__ __launch_bounds__(256) void dropout_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long n,
                                                      uint32_t thresh, float scale, uint64_t seed) {
  seed = resolve_seed(seed);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    y[i] = drop_keep(seed, (uint64_t)i, thresh) ? f2bf(bf2f(x[i]) * scale) : (bf16_t)0;
}

// h[b,t,:] = dropout(enc[b,t,:] + wpe[t,:])   (HF GPT2Model: inputs_embeds + position_embeds, then drop)
__global__ __launch_bounds__(256) void embed_pos_kernel(const bf16_t* __restrict__ enc, const float* __restrict__ wpe,
                                                        bf16_t* __restrict__ h, int B, int T, int E,
                                                        uint32_t thresh, float scale, uint64_t seed) {
  seed = resolve_seed(seed);
  long n = (long)B * T * E;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    int e = (int)(i % E);
    int t = (int)((i / E) % T);
    float v = bf2f(enc[i]) + wpe[(size_t)t * E + e];
    if (thresh) v = drop_keep(seed, (uint64_t)i, thresh) ? v * scale : 0.f;
    h[i] = f2bf(v);
  }
}
// backward: denc = dh * mask ; dwpe[t] += sum_b denc[b,t]
__global__ __launch_bounds__(256) void embed_pos_bwd_kernel(const bf16_t* __restrict__ dh, bf16_t* __restrict__ denc,
                                                            float* __restrict__ dwpe, int B, int T, int E,
                                                            uint32_t thresh, float scale, uint64_t seed) {
  seed = resolve_seed(seed);
  long n = (long)T * E;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float acc = 0.f;
    for (int b = 0; b < B; ++b) {
      long idx = (long)b * n + i;
      float v = bf2f(dh[idx]);
      if (thresh) v = drop_keep(seed, (uint64_t)idx, thresh) ? v * scale : 0.f;
      bf16_t o = f2bf(v);
      denc[idx] = o;
      acc += bf2f(o);
    }
    dwpe[i] += acc;
  }
}

// dx0 bf16 [N,S,D] -> dpos[S,D] += sum_n ; dcls[D] += row s=0 ; dbias[D] += sum_{s>=1}.  blockIdx.y splits the frames so that
// the 19 k (s, chunk) columns of a ViT-B still give a few thousand workgroups of work in flight (fp32 atomics merge the parts)
__global__ __launch_bounds__(256) void patch_bwd_reduce_kernel(const bf16_t* __restrict__ dx, float* __restrict__ dpos,
                                                               float* __restrict__ dcls, float* __restrict__ dbias,
                                                               int N, int S, int D, int frames_per_block, float* __restrict__ part) {
  const int nch = D / 8;
  const long total = (long)S * nch;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int n0 = blockIdx.y * frames_per_block;
  int n1 = n0 + frames_per_block; if (n1 > N) n1 = N;
  const int c = (int)(idx % nch), s = (int)(idx / nch);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bf16_t* src = dx + (size_t)s * D + c * 8;
  const size_t fstride = (size_t)S * D;
  int n = n0;
  for (; n + 4 <= n1; n += 4) {                       // four independent 16-byte loads in flight per thread
    u32x4_t w0 = *(const u32x4_t*)(src + (size_t)n * fstride), w1 = *(const u32x4_t*)(src + (size_t)(n + 1) * fstride);
    u32x4_t w2 = *(const u32x4_t*)(src + (size_t)(n + 2) * fstride), w3 = *(const u32x4_t*)(src + (size_t)(n + 3) * fstride);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[2 * e] += (bflo(w0[e]) + bflo(w1[e])) + (bflo(w2[e]) + bflo(w3[e]));
      acc[2 * e + 1] += (bfhi(w0[e]) + bfhi(w1[e])) + (bfhi(w2[e]) + bfhi(w3[e]));
    }
  }
  for (; n < n1; ++n) {
    u32x4_t w = *(const u32x4_t*)(src + (size_t)n * fstride);
#pragma unroll
    for (int e = 0; e < 4; ++e) { acc[2 * e] += bflo(w[e]); acc[2 * e + 1] += bfhi(w[e]); }
  }
  if (part) {                                            // slot = frame group: [gy][S*D]; dcls / dbias are folded from dpos afterwards
    float* dst = part + (size_t)blockIdx.y * ((size_t)S * D) + (size_t)s * D + c * 8;
    *(f32x4_t*)dst = (f32x4_t){acc[0], acc[1], acc[2], acc[3]};
    *(f32x4_t*)(dst + 4) = (f32x4_t){acc[4], acc[5], acc[6], acc[7]};
    return;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    unsafeAtomicAdd(&dpos[(size_t)s * D + c * 8 + e], acc[e]);
    if (s == 0) unsafeAtomicAdd(&dcls[c * 8 + e], acc[e]);
    else unsafeAtomicAdd(&dbias[c * 8 + e], acc[e]);
  }
}

// fixed-order merge of per-workgroup partial vectors (common.hpp): block = 32 slot groups x 32 columns; a thread adds its
// group's contiguous slots with four interleaved accumulators, thread 0 of a column adds the 32 group sums in order.
struct ReduceOuts { float* o[4]; float* keep; };     // keep: the sums of quantity 0 are also STORED there (may alias slot 0 of part)
__global__ __launch_bounds__(1024) void partials_reduce_kernel(const float* part, int nslots, long n, ReduceOuts outs) {
  __shared__ float red[32][33];
  const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
  const long col = (long)blockIdx.x * 32 + c;
  const int q = blockIdx.y;
  const int per = (nslots + 31) / 32;
  const int s0 = g * per;
  int s1 = s0 + per; if (s1 > nslots) s1 = nslots;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (col < n) {
    const float* src = part + ((size_t)q * nslots) * (size_t)n + col;
    int s = s0;
    for (; s + 4 <= s1; s += 4) {
      a0 += src[(size_t)s * n]; a1 += src[(size_t)(s + 1) * n]; a2 += src[(size_t)(s + 2) * n]; a3 += src[(size_t)(s + 3) * n];
    }
    for (; s < s1; ++s) a0 += src[(size_t)s * n];
  }
  red[g][c] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (g == 0 && col < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += red[k][c];
    if (outs.o[q]) outs.o[q][col] += t;
    if (q == 0 && outs.keep) outs.keep[col] = t;
  }
}

// column sums of a bf16 matrix: out[n] += sum_m x[m][n]   (bias gradients that have no producer to fuse into)
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x, int ld, float* __restrict__ out, int M, int N, int rows_per_block,
                                                     float* __restrict__ part) {
  const int nch = N / 8;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= nch) return;
  const int m0 = blockIdx.y * rows_per_block;
  int m1 = m0 + rows_per_block; if (m1 > M) m1 = M;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int m = m0; m < m1; ++m) {
    u32x4_t w = *(const u32x4_t*)(x + (size_t)m * ld + c * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) { acc[2 * e] += bflo(w[e]); acc[2 * e + 1] += bfhi(w[e]); }
  }
  if (part) {
    float* dst = part + (size_t)blockIdx.y * N + c * 8;
    *(f32x4_t*)dst = (f32x4_t){acc[0], acc[1], acc[2], acc[3]};
    *(f32x4_t*)(dst + 4) = (f32x4_t){acc[4], acc[5], acc[6], acc[7]};
    return;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) unsafeAtomicAdd(&out[c * 8 + e], acc[e]);
}

// feat loss (models/future_prediction.py:207-215, torch.nn.MSELoss(reduction='none')): loss[b,t,:] = (dec[b,t,:] - x[b,t+1,:])^2, t < T-1
__global__ __launch_bounds__(256) void mse_shift_fwd_kernel(const float* __restrict__ dec, const float* __restrict__ x,
                                                            float* __restrict__ loss, int B, int T, int F) {
  long n = (long)B * (T - 1) * F;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    int f = (int)(i % F); long r = i / F; int t = (int)(r % (T - 1)); int b = (int)(r / (T - 1));
    float d = dec[((size_t)b * T + t) * F + f] - x[((size_t)b * T + t + 1) * F + f];
    loss[i] = d * d;
  }
}
// its backward: ddec[b,t,:] = 2 (dec[b,t] - x[b,t+1]) g[b,t] for t < T-1 (row T-1: 0);  dx[b,t,:] = -ddec[b,t-1,:] for t >= 1 (row 0: 0)
__global__ __launch_bounds__(256) void mse_shift_bwd_kernel(const float* __restrict__ dec, const float* __restrict__ x,
                                                            const float* __restrict__ g, float* __restrict__ ddec,
                                                            float* __restrict__ dx, int B, int T, int F) {
  long n = (long)B * T * F;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    int f = (int)(i % F); long r = i / F; int t = (int)(r % T); int b = (int)(r / T);
    float a = 0.f, c = 0.f;
    if (t < T - 1) a = 2.f * (dec[i] - x[i + F]) * g[((size_t)b * (T - 1) + t) * F + f];
    if (t >= 1) c = -2.f * (dec[i - F] - x[i]) * g[((size_t)b * (T - 1) + t - 1) * F + f];
    ddec[i] = a; dx[i] = c;
  }
}
// fp32 [rows, cols] (row stride lds) -> bf16 [rows, ldd] with columns >= cols zero (classifier gradient padding)
__global__ __launch_bounds__(256) void pad_cast_kernel(const float* __restrict__ src, int lds, bf16_t* __restrict__ dst, int ldd,
                                                       int rows, int cols) {
  long n = (long)rows * ldd;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    int c = (int)(i % ldd); long r = i / ldd;
    dst[i] = c < cols ? f2bf(src[(size_t)r * lds + c]) : (bf16_t)0;
  }
}
// ReLU with its derivative mask (torch.nn.TransformerEncoderLayer's activation, models/temporal_aggregation.py:87): y = max(x, 0),
// mask = x > 0 ? 1 : 0 (bf16), so that backward is the GEMM epilogue's "multiply by the saved derivative" (act 3)
__global__ __launch_bounds__(256) void relu_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, bf16_t* __restrict__ mask, long n8) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    u32x4_t w = *(const u32x4_t*)(x + i * 8), o, m;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = bflo(w[e]), b = bfhi(w[e]);
      o[e] = pack2bf(fmaxf(a, 0.f), fmaxf(b, 0.f));
      m[e] = pack2bf(a > 0.f ? 1.f : 0.f, b > 0.f ? 1.f : 0.f);
    }
    *(u32x4_t*)(y + i * 8) = o;
    *(u32x4_t*)(mask + i * 8) = m;
  }
}
// dst[c][r] = src[r][c] (bf16), 64 x 64 tiles through LDS: the transposed bf16 shadow of a Linear weight, refreshed once per
// optimizer step, lets the data-gradient GEMM read the weight k-major (ds_read_b128) instead of through transposing LDS reads
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ src, long lds_, bf16_t* __restrict__ dst, long ldd, int rows, int cols) {
  __shared__ bf16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < rows && c0 + c < cols) ? src[(size_t)(r0 + r) * lds_ + c0 + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (c0 + c < cols && r0 + r < rows) dst[(size_t)(c0 + c) * ldd + r0 + r] = tile[r][c];
  }
}
// the same for a whole table of matrices in ONE launch (blockIdx.y = matrix, blockIdx.x = 64x64 tile of it): the ~40 transposed
// weight shadows of a ViT are refreshed after every optimizer step, and forty 10-us launches of a few hundred workgroups each
// cost more than the 340 MB they move.  16-byte global loads and stores when the shapes allow it.
struct TransposeJob { const bf16_t* src; bf16_t* dst; long ld_src, ld_dst; int rows, cols; };
__global__ __launch_bounds__(256) void transpose_batch_kernel(const TransposeJob* __restrict__ jobs) {
  __shared__ bf16_t tile[64][66];                    // row pitch 132 B: rows 8 apart land 8 banks apart (2-way conflicts on the column reads)
  const TransposeJob jb = jobs[blockIdx.y];
  const int tc = (jb.cols + 63) / 64, tr = (jb.rows + 63) / 64;
  if ((int)blockIdx.x >= tc * tr) return;
  const int r0 = (blockIdx.x / tc) * 64, c0 = (blockIdx.x % tc) * 64;
  const bool vec = (jb.rows % 8 == 0) && (jb.cols % 8 == 0) && (jb.ld_src % 8 == 0) && (jb.ld_dst % 8 == 0) &&
                   ((((uintptr_t)jb.src) | ((uintptr_t)jb.dst)) & 15u) == 0;
  if (vec) {
    for (int i = threadIdx.x; i < 64 * 8; i += 256) {            // 8 chunks of 8 columns per row
      const int r = i >> 3, c = (i & 7) * 8;
      u32x4_t w = {0u, 0u, 0u, 0u};
      if (r0 + r < jb.rows && c0 + c < jb.cols) w = *(const u32x4_t*)(jb.src + (size_t)(r0 + r) * jb.ld_src + c0 + c);
      uint32_t* t32 = (uint32_t*)&tile[r][c];
      t32[0] = w[0]; t32[1] = w[1]; t32[2] = w[2]; t32[3] = w[3];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 8; i += 256) {            // output row = source column c, 8 source rows per chunk
      const int c = i >> 3, r = (i & 7) * 8;
      if (c0 + c < jb.cols && r0 + r < jb.rows) {
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (uint32_t)tile[r + 2 * e][c] | ((uint32_t)tile[r + 2 * e + 1][c] << 16);
        *(u32x4_t*)(jb.dst + (size_t)(c0 + c) * jb.ld_dst + r0 + r) = o;
      }
    }
  } else {
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
      const int r = i >> 6, c = i & 63;
      tile[r][c] = (r0 + r < jb.rows && c0 + c < jb.cols) ? jb.src[(size_t)(r0 + r) * jb.ld_src + c0 + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
      const int c = i >> 6, r = i & 63;
      if (c0 + c < jb.cols && r0 + r < jb.rows) jb.dst[(size_t)(c0 + c) * jb.ld_dst + r0 + r] = tile[r][c];
    }
  }
}
// dst[r, :] += src[r, :] for strided bf16 rows (the CLS rows of a [frames*S, D] tensor: ldd = S*D)
__global__ __launch_bounds__(256) void add_rows_kernel(bf16_t* __restrict__ dst, long ldd, const bf16_t* __restrict__ src, long lds,
                                                       int rows, int D) {
  const int nch = D / 8;
  long n = (long)rows * nch;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    int c = (int)(i % nch); long r = i / nch;
    u32x4_t a = *(const u32x4_t*)(dst + r * ldd + c * 8), b = *(const u32x4_t*)(src + r * lds + c * 8), o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2bf(bflo(a[e]) + bflo(b[e]), bfhi(a[e]) + bfhi(b[e]));
    *(u32x4_t*)(dst + r * ldd + c * 8) = o;
  }
}
}  // namespace

#define GRID_FOR(n, per) ({ long g__ = ((n) + (per) - 1) / (per); if (g__ > 8192) g__ = 8192; if (g__ < 1) g__ = 1; (int)g__; })

extern "C" int avt_im2col_patch16(const float* video, void* patches, int N, int Himg, int Wimg, void* stream) {
  AVT_CHECK(video && patches && N > 0, "avt_im2col_patch16: null argument");
  AVT_CHECK(Himg % 16 == 0 && Wimg % 16 == 0 && Himg > 0 && Wimg > 0, "avt_im2col_patch16: image size must be a multiple of 16");
  AVT_CHECK(aligned16(video) && aligned16(patches), "avt_im2col_patch16: 16-byte alignment required");
  long total = (long)N * ((Himg / 16) * (Wimg / 16) + 1) * 48;
  hipLaunchKernelGGL(im2col16_kernel, dim3(GRID_FOR(total, 256)), dim3(256), 0, (hipStream_t)stream, video, (bf16_t*)patches, N, Himg, Wimg);
  AVT_LAUNCH_CHECK();
  return 0;
}
extern "C" int avt_posres_prep(const float* pos, const float* cls, const float* bias, void* R, int S, int D, void* stream) {
  AVT_CHECK(pos && cls && bias && R && S > 0 && D > 0, "avt_posres_prep: null argument");
  hipLaunchKernelGGL(posres_kernel, dim3((S * D + 255) / 256), dim3(256), 0, (hipStream_t)stream, pos, cls, bias, (bf16_t*)R, S, D);
  AVT_LAUNCH_CHECK();
  return 0;
}
extern "C" int avt_cast_f32_to_bf16(const float* src, void* dst, long n, void* stream) {
  AVT_CHECK(src && dst && n > 0, "avt_cast_f32_to_bf16: null argument");
  AVT_CHECK(aligned16(src) && aligned16(dst), "avt_cast_f32_to_bf16: 16-byte alignment required");
  hipLaunchKernelGGL(cast_kernel, dim3(GRID_FOR(n, 2048)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n);
  AVT_LAUNCH_CHECK();
  return 0;
}
extern "C" int avt_cast_bf16_to_f32(const void* src, float* dst, long n, void* stream) {
  AVT_CHECK(src && dst && n > 0, "avt_cast_bf16_to_f32: null argument");
  hipLaunchKernelGGL(cast_back_kernel, dim3(GRID_FOR(n, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, n);
  AVT_LAUNCH_CHECK();
  return 0;
}
extern "C" int avt_dropout_bf16(const void* x, void* y, long n, float p, uint64_t seed, void* stream) {
  AVT_CHECK(x && y && n > 0, "avt_dropout_bf16: null argument");
  AVT_CHECK(p >= 0.f && p < 1.f, "avt_dropout_bf16: bad p");
  hipLaunchKernelGGL(dropout_kernel, dim3(GRID_FOR(n, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, n,
                     drop_threshold(p), 1.f / (1.f - p), seed);
  AVT_LAUNCH_CHECK();
  return 0;
}
extern "C" int avt_embed_pos_fwd(const void* enc, const float* wpe, void* h, int B, int T, int E, float p, uint64_t seed, void* stream) {
  AVT_CHECK(enc && wpe && h && B > 0 && T > 0 && E > 0, "avt_embed_pos_fwd: null argument");
  AVT_CHECK(p >= 0.f && p < 1.f, "avt_embed_pos_fwd: bad p");
  long n = (long)B * T * E;
  hipLaunchKernelGGL(embed_pos_kernel, dim3(GRID_FOR(n, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)enc, wpe, (bf16_t*)h, B, T, E,
                     drop_threshold(p), 1.f / (1.f - p), seed);
  AVT_LAUNCH_CHECK();
  return 0;
}
extern "C" int avt_embed_pos_bwd(const void* dh, void* denc, float* dwpe, int B, int T, int E, float p, uint64_t seed, void* stream) {
  AVT_CHECK(dh && denc && dwpe && B > 0 && T > 0 && E > 0, "avt_embed_pos_bwd: null argument");
  long n = (long)T * E;
  hipLaunchKernelGGL(embed_pos_bwd_kernel, dim3(GRID_FOR(n, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dh, (bf16_t*)denc, dwpe, B, T, E,
                     drop_threshold(p), 1.f / (1.f - p), seed);
  AVT_LAUNCH_CHECK();
  return 0;
}
static int reduce_partials_keep(const float* part, int nslots, long n, float* const* outs, int nq, float* keep, hipStream_t stream) {
  if (nslots <= 0 || n <= 0 || nq < 1 || nq > 4) { avt_set_error("avt_reduce_partials: bad shape"); return -1; }
  ReduceOuts o{};
  for (int q = 0; q < nq; ++q) o.o[q] = outs[q];
  o.keep = keep;
  hipLaunchKernelGGL(partials_reduce_kernel, dim3((unsigned)((n + 31) / 32), nq), dim3(1024), 0, stream, part, nslots, n, o);
  AVT_LAUNCH_CHECK();
  return 0;
}
int avt_reduce_partials(const float* part, int nslots, long n, float* const* outs, int nq, hipStream_t stream) {
  return reduce_partials_keep(part, nslots, n, outs, nq, nullptr, stream);
}

extern "C" size_t avt_patch_embed_bwd_reduce_workspace_bytes(int N, int S, int D) {
  (void)N;
  return (size_t)16 * (size_t)S * (size_t)D * 4;                 // at most 16 frame groups
}
extern "C" size_t avt_colsum_workspace_bytes(int M, int N) {
  (void)M;
  return (size_t)2048 * (size_t)N * 4;                           // at most 2048 row groups
}

extern "C" int avt_patch_embed_bwd_reduce(const void* dx, float* dpos, float* dcls, float* dbias, int N, int S, int D,
                                          float* part, size_t part_bytes, void* stream) {
  AVT_CHECK(dx && dpos && dcls && dbias && N > 0 && S > 0 && D > 0 && D % 8 == 0, "avt_patch_embed_bwd_reduce: bad argument");
  AVT_CHECK(aligned16(dx), "avt_patch_embed_bwd_reduce: 16-byte alignment required");
  long total = (long)S * (D / 8);
  int gx = (int)((total + 255) / 256);
  int gy = 1024 / gx; if (gy < 1) gy = 1; if (gy > 16) gy = 16; if (gy > (N + 7) / 8) gy = (N + 7) / 8;   // few frame groups: the fp32 atomics of the groups collide
  int fpb = (N + gy - 1) / gy; gy = (N + fpb - 1) / fpb;
  AVT_CHECK(!part || (aligned16(part) && part_bytes >= (size_t)gy * S * D * 4), "avt_patch_embed_bwd_reduce: partials workspace too small or misaligned");
  hipLaunchKernelGGL(patch_bwd_reduce_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dx, dpos, dcls, dbias, N, S, D, fpb, part);
  AVT_LAUNCH_CHECK();
  if (part) {
    // frame groups -> dpos (the per-position sums stay in slot 0), then row 0 -> dcls and rows 1.. -> dbias in row order
    float* o1[1] = {dpos};
    int rc = reduce_partials_keep(part, gy, (long)S * D, o1, 1, part, (hipStream_t)stream);
    if (rc) return rc;
    float* o2[1] = {dcls};
    rc = avt_reduce_partials(part, 1, D, o2, 1, (hipStream_t)stream);
    if (rc) return rc;
    if (S > 1) { float* o3[1] = {dbias}; rc = avt_reduce_partials(part + D, S - 1, D, o3, 1, (hipStream_t)stream); }
    return rc;
  }
  return 0;
}
extern "C" int avt_colsum_bf16(const void* x, int ld, float* out, int M, int N, float* part, size_t part_bytes, void* stream) {
  AVT_CHECK(x && out && M > 0 && N > 0 && N % 8 == 0 && ld % 8 == 0, "avt_colsum_bf16: N and ld must be multiples of 8");
  AVT_CHECK(aligned16(x), "avt_colsum_bf16: 16-byte alignment required");
  int nch = N / 8, gx = (nch + 255) / 256;
  int gy = 2048 / gx; if (gy < 1) gy = 1; if (gy > (M + 15) / 16) gy = (M + 15) / 16;
  int rpb = (M + gy - 1) / gy;
  gy = (M + rpb - 1) / rpb;
  AVT_CHECK(!part || (aligned16(part) && part_bytes >= (size_t)gy * N * 4), "avt_colsum_bf16: partials workspace too small or misaligned");
  hipLaunchKernelGGL(colsum_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ld, out, M, N, rpb, part);
  AVT_LAUNCH_CHECK();
  if (part) { float* o1[1] = {out}; return avt_reduce_partials(part, gy, N, o1, 1, (hipStream_t)stream); }
  return 0;
}
extern "C" int avt_mse_shift_fwd(const float* dec, const float* x, float* loss, int B, int T, int F, void* stream) {
  AVT_CHECK(dec && x && loss && B > 0 && T > 1 && F > 0, "avt_mse_shift_fwd: bad argument");
  long n = (long)B * (T - 1) * F;
  hipLaunchKernelGGL(mse_shift_fwd_kernel, dim3(GRID_FOR(n, 256)), dim3(256), 0, (hipStream_t)stream, dec, x, loss, B, T, F);
  AVT_LAUNCH_CHECK();
  return 0;
}
extern "C" int avt_mse_shift_bwd(const float* dec, const float* x, const float* gloss, float* ddec, float* dx, int B, int T, int F,
                                 void* stream) {
  AVT_CHECK(dec && x && gloss && ddec && dx && B > 0 && T > 1 && F > 0, "avt_mse_shift_bwd: bad argument");
  long n = (long)B * T * F;
  hipLaunchKernelGGL(mse_shift_bwd_kernel, dim3(GRID_FOR(n, 256)), dim3(256), 0, (hipStream_t)stream, dec, x, gloss, ddec, dx, B, T, F);
  AVT_LAUNCH_CHECK();
  return 0;
}
extern "C" int avt_pad_cast_f32_to_bf16(const float* src, int lds, void* dst, int ldd, int rows, int cols, void* stream) {
  AVT_CHECK(src && dst && rows > 0 && cols > 0 && ldd >= cols && lds >= cols, "avt_pad_cast_f32_to_bf16: bad argument");
  long n = (long)rows * ldd;
  hipLaunchKernelGGL(pad_cast_kernel, dim3(GRID_FOR(n, 256)), dim3(256), 0, (hipStream_t)stream, src, lds, (bf16_t*)dst, ldd, rows, cols);
  AVT_LAUNCH_CHECK();
  return 0;
}
extern "C" int avt_add_rows_bf16(void* dst, long ldd, const void* src, long lds, int rows, int D, void* stream) {
  AVT_CHECK(dst && src && rows > 0 && D > 0, "avt_add_rows_bf16: bad argument");
  AVT_CHECK(D % 8 == 0 && ldd % 8 == 0 && lds % 8 == 0 && aligned16(dst) && aligned16(src), "avt_add_rows_bf16: D and strides must be multiples of 8, pointers 16-byte aligned");
  long n = (long)rows * (D / 8);
  hipLaunchKernelGGL(add_rows_kernel, dim3(GRID_FOR(n, 256)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)dst, ldd, (const bf16_t*)src, lds, rows, D);
  AVT_LAUNCH_CHECK();
  return 0;
}
extern "C" int avt_relu_bf16(const void* x, void* y, void* mask, long n, void* stream) {
  AVT_CHECK(x && y && mask && n > 0 && n % 8 == 0, "avt_relu_bf16: n must be a positive multiple of 8");
  AVT_CHECK(aligned16(x) && aligned16(y) && aligned16(mask), "avt_relu_bf16: 16-byte alignment required");
  hipLaunchKernelGGL(relu_kernel, dim3(GRID_FOR(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, (bf16_t*)mask, n / 8);
  AVT_LAUNCH_CHECK();
  return 0;
}
extern "C" int avt_transpose_batch_bf16(const void* jobs, int njobs, int max_tiles, void* stream) {
  AVT_CHECK(jobs && njobs > 0 && njobs <= 65535 && max_tiles > 0, "avt_transpose_batch_bf16: bad argument");
  AVT_CHECK((((uintptr_t)jobs) & 7u) == 0, "avt_transpose_batch_bf16: the job table must be 8-byte aligned");
  hipLaunchKernelGGL(transpose_batch_kernel, dim3(max_tiles, njobs), dim3(256), 0, (hipStream_t)stream, (const TransposeJob*)jobs);
  AVT_LAUNCH_CHECK();
  return 0;
}
extern "C" int avt_transpose_bf16(const void* src, long ld_src, void* dst, long ld_dst, int rows, int cols, void* stream) {
  AVT_CHECK(src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= rows, "avt_transpose_bf16: bad argument");
  hipLaunchKernelGGL(transpose_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, ld_src,
                     (bf16_t*)dst, ld_dst, rows, cols);
  AVT_LAUNCH_CHECK();
  return 0;
}
