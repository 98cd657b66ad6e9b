// bf16 MFMA GEMM for gfx950 (MI355X): C[M,N] = epilogue( sum_k opA[m,k] * opB[n,k] ), fp32 accumulate.
//
// One kernel template covers every contraction of the AVT training step:
//   * Linear (weight (out,in), timm ViT / encoder / decoder / classifier):  fwd  A=x[M][K] k-major, B=W[N][K] k-major
//                                                                          dgrad A=dy k-major,  B=W stored [K][N]
//                                                                          wgrad A=dy stored [K][M], B=x stored [K][N]
//   * HF Conv1D (weight (in,out), GPT-2):                                   fwd  A=x k-major,  B=W stored [K][N]
//                                                                          dgrad A=dy k-major, B=W[N][K] k-major
//                                                                          wgrad A=x stored [K][M], B=dy stored [K][N]
// "k-major" operands are read from LDS with ds_read_b128; operands stored with the reduction index as the
// ROW index are read with gfx950's transposing ds_read_b64_tr_b16, so no transposed copy of any tensor ever
// exists in HBM.
//
// Structure (v1): 256 threads = 4 waves (2x2), block tile BMxBNx64, v_mfma_f32_32x32x16_bf16, operands staged
// global->LDS by LDS-DMA (buffer_load ... lds, 16 B/lane; out-of-range rows arrive as zeros through the buffer
// descriptor's bounds check), XOR-swizzled through the per-lane SOURCE address so the LDS image stays
// lane-linear, double-buffered with one barrier per K tile, XCD-aware tile order.
// Epilogue 0 (activations): accumulators are staged through LDS so every lane owns 4 consecutive columns of
//   one row: + bias, GELU (erf|tanh) with optional pre-activation second output, multiply by GELU'(aux),
//   dropout, + residual (optionally row-periodic), per-column sums (bias gradients), bf16 or fp32 store.
// Epilogue 1 (weight gradients): fp32 atomic accumulation straight from the accumulator layout (split-K over
//   the reduction axis fills the chip when the output has few tiles).
#include "common.hpp"
#include "../../include/avt_hip.h"

namespace {

struct GemmParams {
  const bf16_t* A; const bf16_t* B; void* C; bf16_t* C2;
  const float* bias; const bf16_t* res; const bf16_t* aux; float* colsum;
  int M, N, K;
  int lda, ldb, ldc, ldc2, ldres, ldaux;
  int res_period;
  int act;          // 0 none | 1 gelu_erf | 2 gelu_tanh | 3 *= gelu_erf'(aux) | 4 *= gelu_tanh'(aux)
  int out_f32;
  int splitk;
  int tiles_m, tiles_n;
  uint32_t a_bytes, b_bytes;     // buffer-descriptor bounds
  uint32_t drop_thresh; float drop_scale; uint64_t drop_seed;
};

constexpr int BK = 64;

// XCD-aware bijective remap of the linear block id: XCD x (= id % 8 by dispatch order) owns a contiguous
// range of logical tiles, so tiles sharing an A row-panel sit behind the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  int q = nblk >> 3, r = nblk & 7;
  int xcd = bid & 7, idx = bid >> 3;
  int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + idx;
}

// ---- operand tile loaders (LDS-DMA, swizzle on the source address) ------------------------------------
// k-major operand: LDS tile [BR][64] bf16 (128-B rows); 16-B chunk c of row r lives at chunk c ^ ((r>>1)&7).
template <int BR>
__device__ __forceinline__ void stage_kmajor(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int row0, int k0, int ld,
                                             int K, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < BR / 32; ++j) {
    int r = j * 32 + wave * 8 + (lane >> 3);
    int c = (lane & 7) ^ ((r >> 1) & 7);
    int kcol = k0 + c * 8;
    uint32_t off = (uint32_t)(((size_t)(row0 + r) * (size_t)ld + (size_t)kcol) * 2);
    if (kcol >= K) off = 0xFFFFFFF0u;                     // forces the bounds check -> zeros
    char* dst = lds_tile + (j * 32 + wave * 8) * 128;     // wave-uniform base; lane l lands at +16*l
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, AVT_LDS_PTR(dst), 16, off, 0, 0, 0);
  }
}
// reduction-index-as-row operand: LDS tile [64][BR] bf16; chunk swizzle keeps the 4 rows of a tr-read on
// distinct 64-B bank segments (BR=128: c ^ ((r&3)<<2); BR=64: c ^ (((r>>1)&1)<<2)).
template <int BR>
__device__ __forceinline__ int kstrided_swz(int r) { return BR == 128 ? ((r & 3) << 2) : (((r >> 1) & 1) << 2); }
template <int BR>
__device__ __forceinline__ void stage_kstrided(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int col0, int k0, int ld,
                                               int ncols, int wave, int lane) {
  constexpr int CPR = BR / 8;            // 16-B chunks per row
  constexpr int RPI = 64 / CPR;          // rows per wave instruction
#pragma unroll
  for (int j = 0; j < 64 / (4 * RPI); ++j) {
    int r = j * 4 * RPI + wave * RPI + lane / CPR;
    int c = (lane % CPR) ^ kstrided_swz<BR>(r);
    int col = col0 + c * 8;
    uint32_t off = (uint32_t)(((size_t)(k0 + r) * (size_t)ld + (size_t)col) * 2);
    if (col >= ncols) off = 0xFFFFFFF0u;
    char* dst = lds_tile + (j * 4 * RPI + wave * RPI) * (BR * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, AVT_LDS_PTR(dst), 16, off, 0, 0, 0);
  }
}

// ---- fragment reads --------------------------------------------------------------------------------------
// Both forms give lane l the 8 values k = ks*16 + (l>>5)*8 + e, e = 0..7, of operand row (tile*32 + (l&31)).
__device__ __forceinline__ bf16x8_t frag_kmajor(const char* lds_tile, int tile, int ks, int lane) {
  int r = tile * 32 + (lane & 31);
  int c = (ks * 2 + (lane >> 5)) ^ ((r >> 1) & 7);
  return *(const bf16x8_t*)(lds_tile + r * 128 + c * 16);
}
template <int BR>
__device__ __forceinline__ bf16x8_t frag_kstrided(const char* lds_tile, int tile, int ks, int lane) {
  int g = lane >> 4, i16 = lane & 15;
  int col = tile * 32 + (g & 1) * 16 + (i16 & 3) * 4;
  int rbase = ks * 16 + (g >> 1) * 8 + (i16 >> 2);
  union { bf16x8_t v; s16x4_t h[2]; } u;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    int r = rbase + h * 4;
    int c = (col >> 3) ^ kstrided_swz<BR>(r);
    const char* p = lds_tile + r * (BR * 2) + c * 16 + (col & 7) * 2;
    u.h[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
  }
  return u.v;
}

template <int BM, int BN, bool A_KMAJOR, bool B_KMAJOR, int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
  constexpr int WM = BM / 2, WN = BN / 2;          // wave tile
  constexpr int TM = WM / 32, TN = WN / 32;        // 32x32 MFMA tiles per wave
  constexpr int A_TILE = BM * BK * 2, B_TILE = BN * BK * 2;
  constexpr int STAGE = A_TILE + B_TILE;
  constexpr int EPI_BYTES = (EPI == 0) ? 4 * WM * WN * 4 : 0;
  constexpr int LDS_BYTES = (2 * STAGE > EPI_BYTES) ? 2 * STAGE : EPI_BYTES;
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int ntile = p.tiles_m * p.tiles_n;
  const int bid = blockIdx.x;
  const int split = bid / ntile;
  const int t = xcd_remap(bid - split * ntile, ntile);
  const int tm0 = (t / p.tiles_n) * BM;
  const int tn0 = (t % p.tiles_n) * BN;

  const int nk_total = (p.K + BK - 1) / BK;
  const int kt_begin = (int)(((long)nk_total * split) / p.splitk);
  const int kt_end = (int)(((long)nk_total * (split + 1)) / p.splitk);
  const int nk = kt_end - kt_begin;

  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.b_bytes, 0x00020000);

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto stage = [&](int buf, int kt) {
    char* base = lds + buf * STAGE;
    int k0 = kt * BK;
    if (A_KMAJOR) stage_kmajor<BM>(ra, base, tm0, k0, p.lda, p.K, wave, lane);
    else stage_kstrided<BM>(ra, base, tm0, k0, p.lda, p.M, wave, lane);
    if (B_KMAJOR) stage_kmajor<BN>(rb, base + A_TILE, tn0, k0, p.ldb, p.K, wave, lane);
    else stage_kstrided<BN>(rb, base + A_TILE, tn0, k0, p.ldb, p.N, wave, lane);
  };

  if (nk > 0) {
    stage(0, kt_begin);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  for (int it = 0; it < nk; ++it) {
    const int cur = it & 1;
    if (it + 1 < nk) stage(cur ^ 1, kt_begin + it + 1);
    const char* la = lds + cur * STAGE;
    const char* lb = la + A_TILE;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8_t af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = A_KMAJOR ? frag_kmajor(la, wm * TM + i, ks, lane) : frag_kstrided<BM>(la, wm * TM + i, ks, lane);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bfr[j] = B_KMAJOR ? frag_kmajor(lb, wn * TN + j, ks, lane) : frag_kstrided<BN>(lb, wn * TN + j, ks, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  if (EPI == 1) {
    // weight-gradient epilogue: fp32 accumulate into C (atomics; C is pre-zeroed or holds the running sum)
    float* C = (float*)p.C;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        int n = tn0 + wn * WN + j * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int m = tm0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (m < p.M && n < p.N) unsafeAtomicAdd(&C[(size_t)m * p.ldc + n], acc[i][j][r]);
        }
      }
    return;
  } else {
    // activation epilogue: registers -> wave-private fp32 LDS patch -> row-major 4-column strips per lane
    float* patch = (float*)(lds) + wave * (WM * WN);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int ml = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          int nl = j * 32 + (lane & 31);
          patch[ml * WN + nl] = acc[i][j][r];
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    constexpr int LPR = WN / 4;          // lanes per row
    constexpr int RPI = 64 / LPR;        // rows per iteration
    const int cl = (lane % LPR) * 4;
    const int n = tn0 + wn * WN + cl;
    const bool ncol_ok = n < p.N;        // N % 4 == 0 is a host-checked precondition
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && ncol_ok) { f32x4_t b = *(const f32x4_t*)(p.bias + n); bias4[0] = b[0]; bias4[1] = b[1]; bias4[2] = b[2]; bias4[3] = b[3]; }
    float csum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int itr = 0; itr < WM / RPI; ++itr) {
      int ml = itr * RPI + lane / LPR;
      int m = tm0 + wm * WM + ml;
      if (m < p.M && ncol_ok) {
        f32x4_t v4 = *(const f32x4_t*)(patch + ml * WN + cl);
        float v[4] = {v4[0] + bias4[0], v4[1] + bias4[1], v4[2] + bias4[2], v4[3] + bias4[3]};
        if (p.act >= 3) {
          u32x2_t a = *(const u32x2_t*)(p.aux + (size_t)m * p.ldaux + n);
          float h[4] = {bflo(a[0]), bfhi(a[0]), bflo(a[1]), bfhi(a[1])};
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= (p.act == 3) ? dgelu_erf(h[e]) : dgelu_tanh(h[e]);
        }
        if (p.C2) {
          u32x2_t o; o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
          *(u32x2_t*)(p.C2 + (size_t)m * p.ldc2 + n) = o;
        }
        if (p.act == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        } else if (p.act == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(v[e]);
        }
        if (p.drop_thresh) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = drop_keep(p.drop_seed, (uint64_t)m * (uint64_t)p.N + (uint64_t)(n + e), p.drop_thresh) ? v[e] * p.drop_scale : 0.f;
        }
        if (p.res) {
          int mr = p.res_period ? (m % p.res_period) : m;
          u32x2_t a = *(const u32x2_t*)(p.res + (size_t)mr * p.ldres + n);
          v[0] += bflo(a[0]); v[1] += bfhi(a[0]); v[2] += bflo(a[1]); v[3] += bfhi(a[1]);
        }
        if (p.colsum) { csum[0] += v[0]; csum[1] += v[1]; csum[2] += v[2]; csum[3] += v[3]; }
        if (p.out_f32) {
          f32x4_t o = {v[0], v[1], v[2], v[3]};
          *(f32x4_t*)((float*)p.C + (size_t)m * p.ldc + n) = o;
        } else {
          u32x2_t o; o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
          *(u32x2_t*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = o;
        }
      }
    }
    if (p.colsum) {
      // lanes sharing (lane % LPR) own the same 4 columns: fold the RPI row groups, one atomic per column
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) csum[e] += __shfl_xor(csum[e], o, 64);
      }
      if (lane < LPR && ncol_ok) {
#pragma unroll
        for (int e = 0; e < 4; ++e) unsafeAtomicAdd(&p.colsum[n + e], csum[e]);
      }
    }
  }
}

template <int BM, int BN, bool AK, bool BK_, int EPI>
int launch(const GemmParams& p, hipStream_t s) {
  int grid = p.tiles_m * p.tiles_n * p.splitk;
  hipLaunchKernelGGL((gemm_kernel<BM, BN, AK, BK_, EPI>), dim3(grid), dim3(256), 0, s, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { avt_set_error("avt_gemm: launch failed: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

template <int BM, int BN, int EPI>
int dispatch_layout(const GemmParams& p, int a_kmajor, int b_kmajor, hipStream_t s) {
  if (a_kmajor && b_kmajor) return launch<BM, BN, true, true, EPI>(p, s);
  if (a_kmajor && !b_kmajor) return launch<BM, BN, true, false, EPI>(p, s);
  if (!a_kmajor && !b_kmajor) return launch<BM, BN, false, false, EPI>(p, s);
  return launch<BM, BN, false, true, EPI>(p, s);
}

}  // namespace

extern "C" int avt_gemm_bf16(const void* A, int a_kmajor, int lda, const void* B, int b_kmajor, int ldb,
                             void* C, int ldc, int M, int N, int K,
                             const float* bias, int act, const void* aux, int ldaux,
                             void* C2, int ldc2, const void* res, int ldres, int res_period,
                             float drop_p, uint64_t drop_seed, float* colsum,
                             int out_mode, int splitk, int tile, void* stream) {
  AVT_CHECK(A && B && C, "avt_gemm_bf16: null operand");
  AVT_CHECK(M > 0 && N > 0 && K > 0, "avt_gemm_bf16: bad dims M=%d N=%d K=%d", M, N, K);
  AVT_CHECK(aligned16(A) && aligned16(B) && aligned16(C), "avt_gemm_bf16: operands must be 16-byte aligned");
  AVT_CHECK(lda % 8 == 0 && ldb % 8 == 0, "avt_gemm_bf16: lda/ldb must be multiples of 8 (got %d, %d)", lda, ldb);
  AVT_CHECK(K % 8 == 0 || (!a_kmajor && !b_kmajor), "avt_gemm_bf16: K must be a multiple of 8 for k-major operands (K=%d)", K);
  AVT_CHECK(out_mode >= 0 && out_mode <= 2, "avt_gemm_bf16: out_mode must be 0 (bf16), 1 (fp32) or 2 (fp32 atomic accumulate)");
  AVT_CHECK(act >= 0 && act <= 4, "avt_gemm_bf16: bad act %d", act);
  AVT_CHECK(act < 3 || aux, "avt_gemm_bf16: act %d needs aux", act);
  AVT_CHECK(drop_p >= 0.f && drop_p < 1.f, "avt_gemm_bf16: bad dropout p");
  GemmParams p{};
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.C2 = (bf16_t*)C2;
  p.bias = bias; p.res = (const bf16_t*)res; p.aux = (const bf16_t*)aux; p.colsum = colsum;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldc2 = ldc2; p.ldres = ldres; p.ldaux = ldaux;
  p.res_period = res_period; p.act = act; p.out_f32 = (out_mode == 1);
  p.drop_thresh = drop_threshold(drop_p); p.drop_scale = 1.0f / (1.0f - drop_p); p.drop_seed = drop_seed;
  size_t a_rows = a_kmajor ? (size_t)M : (size_t)K, b_rows = b_kmajor ? (size_t)N : (size_t)K;
  size_t ab = a_rows * (size_t)lda * 2, bb = b_rows * (size_t)ldb * 2;
  AVT_CHECK(ab < 0xFFFFFFF0ull && bb < 0xFFFFFFF0ull, "avt_gemm_bf16: operand larger than 4 GiB");
  p.a_bytes = (uint32_t)ab; p.b_bytes = (uint32_t)bb;
  hipStream_t s = (hipStream_t)stream;
  const int epi = (out_mode == 2) ? 1 : 0;
  if (epi == 0) {
    AVT_CHECK(N % 4 == 0 && ldc % 4 == 0, "avt_gemm_bf16: N and ldc must be multiples of 4 for the activation epilogue");
    AVT_CHECK(splitk <= 1, "avt_gemm_bf16: split-K needs out_mode 2");
  } else {
    AVT_CHECK(!bias && !act && !C2 && !res && !colsum && drop_p == 0.f, "avt_gemm_bf16: accumulate mode has no fused epilogue");
  }
  int bm = 128;
  if (tile == 64 || (tile == 0 && epi == 0 && (long)((M + 127) / 128) * ((N + 127) / 128) < 192)) bm = 64;
  if (epi == 1) bm = (tile == 64) ? 64 : 128;
  p.tiles_m = (M + bm - 1) / bm; p.tiles_n = (N + bm - 1) / bm;
  int nk = (K + BK - 1) / BK;
  if (splitk <= 0) {                 // auto: fill ~2 waves of the 256 CUs x 2 blocks
    splitk = 1;
    if (epi == 1) {
      long tiles = (long)p.tiles_m * p.tiles_n;
      while (tiles * splitk < 768 && splitk * 2 <= nk / 4 && splitk < 64) splitk *= 2;
    }
  }
  if (splitk > nk) splitk = nk;
  p.splitk = splitk;
  if (bm == 128) return epi ? dispatch_layout<128, 128, 1>(p, a_kmajor, b_kmajor, s) : dispatch_layout<128, 128, 0>(p, a_kmajor, b_kmajor, s);
  return epi ? dispatch_layout<64, 64, 1>(p, a_kmajor, b_kmajor, s) : dispatch_layout<64, 64, 0>(p, a_kmajor, b_kmajor, s);
}
