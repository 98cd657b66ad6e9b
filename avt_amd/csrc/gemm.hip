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
// Structure: block tile BMxBNx64 computed by WGM x WGN waves (256x256 by 2x4 waves = 512 threads, one block per CU,
// for the big ViT GEMMs; 128x128 and 64x64 by 2x2 waves for small outputs), v_mfma_f32_32x32x16_bf16, operands
// staged global->LDS by LDS-DMA (buffer_load ... lds, 16 B/lane; out-of-range rows arrive as zeros through the
// buffer descriptor's bounds check), XOR-swizzled through the per-lane SOURCE address so the LDS image stays
// lane-linear, double-buffered with one barrier per K tile, XCD-aware tile order.
// Epilogue 0 (activations): accumulators are staged through LDS so every lane owns 4 consecutive columns of
//   one row: + bias, GELU (erf|tanh) with GELU'(pre-activation) as optional second output, multiply by aux,
//   dropout, + residual (optionally row-periodic), per-column sums (bias gradients), bf16 or fp32 store.
// Epilogue 1 (weight gradients): fp32 atomic accumulation straight from the accumulator layout (split-K over
//   the reduction axis fills the chip when the output has few tiles).
#include "gemm_tile.hpp"

namespace {

template <int BM, int BN, int WGM, int WGN, int BK, int NSTAGE, bool A_KMAJOR, bool B_KMAJOR, int EPI, bool SPREAD = false, int PR = 0, int MINW = 1, int NWL = 0>
__global__ __launch_bounds__(64 * WGM * WGN, MINW) void gemm_kernel(GemmParams p) {
  constexpr int NW = WGM * WGN;
  constexpr int WM = BM / WGM, WN = BN / WGN;      // wave tile
  constexpr int TM = WM / 32, TN = WN / 32;        // 32x32 MFMA tiles per wave
  constexpr int A_TILE = BM * BK * 2, B_TILE = BN * BK * 2;
  constexpr int STAGE = A_TILE + B_TILE;
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  const int ntile = p.tiles_m * p.tiles_n;
  const int bid = blockIdx.x;
  const int lb = xcd_remap(bid, ntile * p.splitk);          // an XCD owns a contiguous range of (split, tile): see xcd_remap
  const int split = lb / ntile;
  const int t = lb - split * ntile;
  const int tm0 = (t / p.tiles_n) * BM;
  const int tn0 = (t % p.tiles_n) * BN;

  const int nk_total = (p.K + BK - 1) / BK;
  // (32-bit arithmetic: nk_total * splitk < 2^31 for every operand below 4 GiB; the 64-bit form expands to ~200 scalar
  // instructions at the start of every workgroup)
  const int kt_begin = p.splitk == 1 ? 0 : (int)((unsigned)nk_total * (unsigned)split / (unsigned)p.splitk);
  const int kt_end = p.splitk == 1 ? nk_total : (int)((unsigned)nk_total * (unsigned)(split + 1) / (unsigned)p.splitk);
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

  constexpr int NL = (BM + BN) * BK * 2 / (NW * 1024);   // LDS-DMA instructions per stage per wave
  auto stage = [&](int buf, int kt) {
    char* base = lds + buf * STAGE;
    int k0 = kt * BK;
    if (A_KMAJOR) stage_kmajor<BM, NW, BK>(ra, base, tm0, k0, p.lda, p.K, wave, lane);
    else stage_kstrided<BM, NW, BK>(ra, base, tm0, k0, p.lda, p.M, wave, lane);
    if (B_KMAJOR) stage_kmajor<BN, NW, BK>(rb, base + A_TILE, tn0, k0, p.ldb, p.K, wave, lane);
    else stage_kstrided<BN, NW, BK>(rb, base + A_TILE, tn0, k0, p.ldb, p.N, wave, lane);
  };

  // NSTAGE-deep LDS ring, one barrier per K tile: iteration `it` waits (counted vmcnt) until its own tile has
  // landed while up to NSTAGE-2 younger tiles stay in flight across the barrier, then refills the slot that was
  // consumed in iteration it-1 with tile it+NSTAGE-1, then computes.
  // NL_W waves issue the in-loop LDS-DMA (all of them by default; with NWL = 4 only waves 0-3 -- one per SIMD -- so
  // that on every SIMD one wave is never stalled in the texture-address queue while its partner feeds the MFMA pipe)
  constexpr int NL_W = NWL ? NWL : NW;
  constexpr int RA = A_KMAJOR ? BM / (NL_W * (64 / (BK / 8))) : BK / (NL_W * (64 / (BM / 8)));   // A LDS-DMA rounds per loading wave
  constexpr int RB = B_KMAJOR ? BN / (NL_W * (64 / (BK / 8))) : BK / (NL_W * (64 / (BN / 8)));   // B rounds
  uint32_t voffA[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, voffB[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};   // fixed size: a dependent-size array makes hipcc drop the host stubs
  static_assert(RA <= 8 && RB <= 8, "voff arrays");
  const bool loader = wave < NL_W;
  const bool hoist = SPREAD && (NWL != 0 || p.K % BK == 0 || (!A_KMAJOR && !B_KMAJOR));   // NWL variants are only dispatched when K % BK == 0
  if (SPREAD) {
#pragma unroll
    for (int q = 0; q < RA; ++q) {
      if (A_KMAJOR) {
        constexpr int CPR = BK / 8, RPI = 64 / CPR;
        int r = q * (NL_W * RPI) + wave * RPI + lane / CPR;
        int c = (lane % CPR) ^ kmajor_swz<BK>(r);
        voffA[q] = (uint32_t)(((size_t)(tm0 + r) * (size_t)p.lda + (size_t)c * 8) * 2);
      } else {
        constexpr int CPR = BM / 8, RPI = 64 / CPR;
        int r = q * NL_W * RPI + wave * RPI + lane / CPR;
        int c = (lane % CPR) ^ kstrided_swz_fwd<BM>(r);
        int col = tm0 + c * 8;
        voffA[q] = (col >= p.M) ? 0xFFFFFFF0u : (uint32_t)(((size_t)r * (size_t)p.lda + (size_t)col) * 2);
      }
    }
#pragma unroll
    for (int q = 0; q < RB; ++q) {
      if (B_KMAJOR) {
        constexpr int CPR = BK / 8, RPI = 64 / CPR;
        int r = q * (NL_W * RPI) + wave * RPI + lane / CPR;
        int c = (lane % CPR) ^ kmajor_swz<BK>(r);
        voffB[q] = (uint32_t)(((size_t)(tn0 + r) * (size_t)p.ldb + (size_t)c * 8) * 2);
      } else {
        constexpr int CPR = BN / 8, RPI = 64 / CPR;
        int r = q * NL_W * RPI + wave * RPI + lane / CPR;
        int c = (lane % CPR) ^ kstrided_swz_fwd<BN>(r);
        int col = tn0 + c * 8;
        voffB[q] = (col >= p.N) ? 0xFFFFFFF0u : (uint32_t)(((size_t)r * (size_t)p.ldb + (size_t)col) * 2);
      }
    }
  }
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nk) stage(s, kt_begin + s);
  int slot = 0;
  for (int it = 0; it < nk; ++it) {
    if (NSTAGE >= 3 && it + NSTAGE - 2 < nk) wait_vmcnt<(NSTAGE - 2) * NL>();
    else if (NSTAGE >= 4 && it + NSTAGE - 3 < nk) wait_vmcnt<(NSTAGE >= 4 ? (NSTAGE - 3) * NL : 0)>();
    else wait_vmcnt<0>();
    asm volatile("s_barrier" ::: "memory");
    int fill = slot + NSTAGE - 1; if (fill >= NSTAGE) fill -= NSTAGE;
    const bool more = (it + NSTAGE - 1 < nk);
    if (!SPREAD && more) stage(fill, kt_begin + it + NSTAGE - 1);
    const char* la = lds + slot * STAGE;
    const char* lb = la + A_TILE;
    if (!SPREAD) {
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        bf16x8_t af[TM], bfr[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[i] = A_KMAJOR ? frag_kmajor<BK>(la, wm * TM + i, ks, lane) : frag_kstrided_na<BM>(la, wm * TM + i, ks, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bfr[j] = B_KMAJOR ? frag_kmajor<BK>(lb, wn * TN + j, ks, lane) : frag_kstrided_na<BN>(lb, wn * TN + j, ks, lane);
        if (!A_KMAJOR || !B_KMAJOR) frag_wait<0>(af, bfr);          // inline-assembly reads: the compiler does not wait for them
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = mma<EPI>(af[i], bfr[j], acc[i][j]);
      }
    } else {
      // software-pipelined k-steps: fragments of step ks+1 are requested before the MFMAs of step ks, and the LDS-DMA
      // refill of the other stage is dribbled out between the MFMA groups (3+3+2 of the 8 instructions in steps 0..2,
      // leaving step 3 as landing time) instead of a burst that stalls every wave at the top of the tile.
      constexpr int KS = BK / 16;
      char* fbase = lds + fill * STAGE;
      const int fkt = kt_begin + it + NSTAGE - 1;
      const int fk0 = fkt * BK;
      // K advance lives in the (scalar) buffer descriptor: base += advance, bound -= advance, so the per-lane offsets
      // (voffA/voffB, computed once before the loop) never change and the bounds check still zero-fills the M / K tails.
      const size_t advA = A_KMAJOR ? (size_t)fk0 * 2 : (size_t)fk0 * (size_t)p.lda * 2;
      const size_t advB = B_KMAJOR ? (size_t)fk0 * 2 : (size_t)fk0 * (size_t)p.ldb * 2;
      __amdgpu_buffer_rsrc_t ra_t = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.A + advA), 0,
                                                                      advA < p.a_bytes ? (uint32_t)(p.a_bytes - advA) : 0u, 0x00020000);
      __amdgpu_buffer_rsrc_t rb_t = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.B + advB), 0,
                                                                      advB < p.b_bytes ? (uint32_t)(p.b_bytes - advB) : 0u, 0x00020000);
      auto dma_part = [&](int q) {
        if (!more) return;
        if (q < RA) {
          if (A_KMAJOR) stage_kmajor_part<BM, NW, BK>(ra, fbase, tm0, fk0, p.lda, p.K, wave, lane, q);
          else stage_kstrided_part<BM, NW, BK>(ra, fbase, tm0, fk0, p.lda, p.M, wave, lane, q);
        } else if (q < RA + RB) {
          if (B_KMAJOR) stage_kmajor_part<BN, NW, BK>(rb, fbase + A_TILE, tn0, fk0, p.ldb, p.K, wave, lane, q - RA);
          else stage_kstrided_part<BN, NW, BK>(rb, fbase + A_TILE, tn0, fk0, p.ldb, p.N, wave, lane, q - RA);
        }
      };
      bf16x8_t af[2][TM], bfr[2][TN];
      auto ldf = [&](int ks, int b) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[b][i] = A_KMAJOR ? frag_kmajor<BK>(la, wm * TM + i, ks, lane) : frag_kstrided_na<BM>(la, wm * TM + i, ks, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bfr[b][j] = B_KMAJOR ? frag_kmajor<BK>(lb, wn * TN + j, ks, lane) : frag_kstrided_na<BN>(lb, wn * TN + j, ks, lane);
      };
      constexpr int NPART = RA + RB;
      constexpr int PER = (NPART + KS - 2) / (KS - 1);        // parts per k-step over the first KS-1 steps
      constexpr int NRD = (A_KMAJOR ? TM : 2 * TM) + (B_KMAJOR ? TN : 2 * TN);     // LDS reads of one k-step's fragments
      static_assert(NRD <= 15, "lgkmcnt is a 4-bit counter");
      ldf(0, 0);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) ldf(ks + 1, (ks + 1) & 1);
        if (!A_KMAJOR || !B_KMAJOR) {                          // fragments of step ks: older than the NRD reads just requested
          if (ks + 1 < KS) frag_wait<NRD>(af[ks & 1], bfr[ks & 1]); else frag_wait<0>(af[ks & 1], bfr[ks & 1]);
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
          const int part = ks * PER + q;
          if (ks < KS - 1 && part < NPART) {
            if (!hoist) dma_part(part);
            else if (more && loader) {
              if (part < RA) {
                constexpr int RPIA = A_KMAJOR ? 64 / (BK / 8) : 64 / (BM / 8);
                char* dst = fbase + (part * NL_W * RPIA + wave * RPIA) * (A_KMAJOR ? BK * 2 : BM * 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_t, AVT_LDS_PTR(dst), 16, voffA[part < RA ? part : 0], 0, 0, 0);
              } else {
                constexpr int RPIB = B_KMAJOR ? 64 / (BK / 8) : 64 / (BN / 8);
                const int qb = part - RA;
                char* dst = fbase + A_TILE + (qb * NL_W * RPIB + wave * RPIB) * (B_KMAJOR ? BK * 2 : BN * 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_t, AVT_LDS_PTR(dst), 16, voffB[qb >= 0 && qb < RB ? qb : 0], 0, 0, 0);
              }
            }
          }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = mma<EPI>(af[ks & 1][i], bfr[ks & 1][j], acc[i][j]);
      }
    }
    if (++slot == NSTAGE) slot = 0;
  }
  asm volatile("s_barrier" ::: "memory");   // every wave is done reading the ring before the epilogue reuses it

  gemm_epilogue<TM, TN, WM, WN, EPI, 0, false, A_KMAJOR && B_KMAJOR>(p, acc, lds, wave, lane, tm0 + wm * WM, tn0 + wn * WN);
}

// ---- skinny kernel: at most 64 output rows (the head at the reference's own 3 clips per GPU: 30 rows at T = 10, 45 at T = 15; the CLS-only last ViT block) ----
// C[M <= 64, N] = epilogue(A[M,K] . op(B)), A k-major.  Such a GEMM is a stream of the weight matrix (2048 x 8192 bf16 = 33.5 MB against 0.5 MB of
// activations) and the 64 x 64 kernel runs it as 32 workgroups of N / 64 -- one eighth of the chip, each with two 16-KB stages in flight: 33 us = 1 TB/s
// (profiles/r06h_kernel_trace_B3.txt).  Split-K would fill the chip but change the fp32 summation order, and every tile route of this library gives the
// same bits for a shape (the batch-invariance tests).  So the reduction stays ONE ordered chain of v_mfma_f32_32x32x16_bf16 per 32 x 32 output tile, and
// the parallelism comes from (a) 32-column tiles -- N / 32 workgroups -- and (b) memory-level parallelism inside a workgroup: four waves issue the LDS-DMA
// of an 18-stage ring of [32 x 64] A + [32 x 64] B tiles (144 KB; 17 stages = 68 KB of weights in flight per workgroup, a wave's own counter sees 34
// requests), wave 0 alone runs the chain (4 MFMAs per stage, the next stage's fragments requested before them).  Stages past the end of the reduction
// are requested all the same (out of range -> zeros, no traffic): the counted wait stays one constant.
template <bool B_KMAJOR, int TMR = 1>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmParams p) {
  // TMR = row tiles of 32 (1: M <= 32, 18 stages; 2: M <= 64 -- the head at 3 clips x 15 frames = 45 rows, BASELINE config 4 at the reference's batch -- two
  // independent chains per wave, 13 stages of 12 KB)
  constexpr int BM = 32 * TMR, BN = 32, BK = 64, NST = TMR == 1 ? 18 : 13, NW = 4;
  constexpr int A_TILE = BM * BK * 2, B_TILE = BN * BK * 2, STAGE = A_TILE + B_TILE;
  constexpr int PER = TMR + 1;                             // LDS-DMA requests per wave and stage
  static_assert(NST * STAGE <= 160 * 1024 && PER * (NST - 1) <= 63, "skinny kernel: LDS / request counter");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // neighbouring column tiles on the same XCD: with B stored [K][N] a k row of the tile is 64 bytes, half a cache line -- the other half belongs to the
  // next tile, and one L2 then fetches the line once for both
  const int tn0 = xcd_remap(blockIdx.x, p.tiles_n) * BN;
  const int nk = (p.K + BK - 1) / BK;
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.b_bytes, 0x00020000);
  auto stage = [&](int buf, int kt) __attribute__((always_inline)) {       // PER requests per wave
    char* base = lds + buf * STAGE;
    const int k0 = kt * BK;
    stage_kmajor<BM, NW, BK>(ra, base, 0, k0, p.lda, p.K, wave, lane);
    if (B_KMAJOR) stage_kmajor<BN, NW, BK>(rb, base + A_TILE, tn0, k0, p.ldb, p.K, wave, lane);
    else stage_kstrided<BN, NW, BK>(rb, base + A_TILE, tn0, k0 < p.K ? k0 : p.K, p.ldb, p.N, wave, lane);
  };
  f32x16_t acc[TMR][1];
#pragma unroll
  for (int i = 0; i < TMR; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
  bf16x8_t af[2][TMR][4], bfr[2][4];
  auto frags = [&](int slot, int b) __attribute__((always_inline)) {
    const char* la = lds + slot * STAGE;
    const char* lb = la + A_TILE;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int i = 0; i < TMR; ++i) af[b][i][ks] = frag_kmajor<BK>(la, i, ks, lane);
      bfr[b][ks] = B_KMAJOR ? frag_kmajor<BK>(lb, 0, ks, lane) : frag_kstrided_na<BN>(lb, 0, ks, lane);
    }
  };
  auto chain = [&](int b) __attribute__((always_inline)) {
    if (!B_KMAJOR) {                                       // (inline-assembly reads: the compiler does not wait for them)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(bfr[b][ks]));
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < TMR; ++i) acc[i][0] = mma<0>(af[b][i][ks], bfr[b][ks], acc[i][0]);
  };
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) stage(s, s);
  int slot = 0;
#pragma nounroll
  for (int it = 0; it < nk; it += 2) {
    // even stage: its fragments go into set 0 while the chain of the previous (odd) stage runs on set 1
    // (lgkmcnt(0): wave 0's fragment reads of the slot refilled below have retired -- they were issued a whole chain ago)
    wait_vmcnt<PER * (NST - 2)>();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    { int fill = slot + NST - 1; if (fill >= NST) fill -= NST; stage(fill, it + NST - 1); }
    if (wave == 0) { frags(slot, 0); if (it) chain(1); }
    if (++slot == NST) slot = 0;
    // odd stage (past the end when nk is odd: fetched as zeros, never multiplied)
    wait_vmcnt<PER * (NST - 2)>();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    { int fill = slot + NST - 1; if (fill >= NST) fill -= NST; stage(fill, it + NST); }
    if (wave == 0) { frags(slot, 1); chain(0); }
    if (++slot == NST) slot = 0;
  }
  if (wave == 0 && !(nk & 1)) chain(1);
  wait_vmcnt<0>();                                        // the zero stages past the end have landed: the ring is quiet
  asm volatile("s_barrier" ::: "memory");
  if (wave == 0) gemm_epilogue<TMR, 1, 32 * TMR, 32, 0, 0, false, B_KMAJOR>(p, acc, lds, 0, lane, 0, tn0);
}

int dispatch_skinny(GemmParams& p, int a_kmajor, int b_kmajor, hipStream_t s) {
  if (!a_kmajor || p.M > 64) { avt_set_error("avt_gemm_bf16: tile 32 is the skinny kernel: A k-major, M <= 64 (got M = %d)", p.M); return -1; }
  // (a B stored [K][N] is addressed up to one stage past its last row: that offset must not wrap)
  if (!b_kmajor && (uint64_t)p.b_bytes + 64ull * (uint64_t)p.ldb * 2ull >= (1ull << 32)) { avt_set_error("avt_gemm_bf16: tile 32: B too large"); return -1; }
  p.tiles_m = 1; p.tiles_n = (p.N + 31) / 32; p.splitk = 1;
  constexpr int smem1 = 18 * 8192, smem2 = 13 * 12288;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem1);
    (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem1);
    (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem2);
    (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem2);
    attr_set = true;
  }
  if (p.M <= 32) {
    if (b_kmajor) hipLaunchKernelGGL((gemm_skinny_kernel<true, 1>), dim3(p.tiles_n), dim3(256), smem1, s, p);
    else hipLaunchKernelGGL((gemm_skinny_kernel<false, 1>), dim3(p.tiles_n), dim3(256), smem1, s, p);
  } else {
    if (b_kmajor) hipLaunchKernelGGL((gemm_skinny_kernel<true, 2>), dim3(p.tiles_n), dim3(256), smem2, s, p);
    else hipLaunchKernelGGL((gemm_skinny_kernel<false, 2>), dim3(p.tiles_n), dim3(256), smem2, s, p);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { avt_set_error("avt_gemm: launch failed: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

// Second pass of the deterministic split-K accumulate: one wave per 1-KB chunk (producing wave w, block (i, j), register
// quad q) mirrors the producer's register layout, sums the chunk over the splits IN ORDER and adds the result to C (every C
// element has exactly one owner, so plain read-modify-write; C keeps the running sum of earlier GEMMs into the same gradient).
template <int TM, int TN, int WGM, int WGN>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, int ldc, int M, int N,
                                                            int tiles_n, int ntile, int splitk, int assign) {
  constexpr int NW = WGM * WGN, WM = TM * 32, WN = TN * 32, BM = WM * WGM, BN = WN * WGN;
  constexpr int CHUNKS = NW * TM * TN * 4;                 // 1-KB chunks per tile
  const int lane = threadIdx.x & 63;
  const int chunk_id = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int t = chunk_id / CHUNKS, c = chunk_id % CHUNKS;
  if (t >= ntile) return;
  const int wave = c / (TM * TN * 4), rem = c % (TM * TN * 4);
  const int i = rem / (TN * 4), j = (rem / 4) % TN, q = rem % 4;
  const int wm = wave / WGN, wn = wave % WGN;
  const size_t slab = (size_t)BM * BN;
  const float* src = ws + (size_t)t * slab + (size_t)c * 256 + lane * 4;
  const size_t sstride = (size_t)ntile * slab;
  f32x4_t v = *(const f32x4_t*)src;
  int s_ = 1;
  for (; s_ + 3 < splitk; s_ += 4) {                       // four independent loads in flight, added in split order
    const f32x4_t a = *(const f32x4_t*)(src + (size_t)s_ * sstride), b = *(const f32x4_t*)(src + (size_t)(s_ + 1) * sstride);
    const f32x4_t d = *(const f32x4_t*)(src + (size_t)(s_ + 2) * sstride), e = *(const f32x4_t*)(src + (size_t)(s_ + 3) * sstride);
    v += a; v += b; v += d; v += e;
  }
  for (; s_ < splitk; ++s_) v += *(const f32x4_t*)(src + (size_t)s_ * sstride);
  const int n = (t % tiles_n) * BN + wn * WN + j * 32 + (lane & 31);
  const int m0 = (t / tiles_n) * BM + wm * WM + i * 32 + 8 * q + 4 * (lane >> 5);
  if (n < N) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (m0 + k < M) { if (assign) C[(size_t)(m0 + k) * ldc + n] = v[k]; else C[(size_t)(m0 + k) * ldc + n] += v[k]; }
  }
}
template <int TM, int TN, int WGM, int WGN>
int launch_reduce(const GemmParams& p, hipStream_t s) {
  const int ntile = p.tiles_m * p.tiles_n;
  constexpr int CHUNKS = WGM * WGN * TM * TN * 4;
  hipLaunchKernelGGL((splitk_reduce_kernel<TM, TN, WGM, WGN>), dim3((ntile * CHUNKS + 3) / 4), dim3(256), 0, s, (const float*)p.ws, (float*)p.C, p.ldc,
                     p.M, p.N, p.tiles_n, ntile, p.splitk, p.c_assign);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { avt_set_error("avt_gemm: reduce launch failed: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

// Split-K factor for the accumulate (weight-gradient) epilogue: one workgroup per CU, so pick the factor whose block count
// best fills whole rounds of the 256 CUs (576 blocks = 2.25 rounds wastes a quarter of the chip; 504 = 1.97 rounds does
// not), keeping at least `kmin` K tiles per split.
static int pick_splitk(long tiles, int nk, int blocks_per_cu, int kmin) {
  const long slots = 256L * blocks_per_cu;
  int best = 1; double best_score = -1.0;
  int smax = nk / kmin; if (smax < 1) smax = 1; if (smax > 96) smax = 96;
  for (int s = 1; s <= smax; ++s) {
    long blocks = tiles * s;
    long rounds = (blocks + slots - 1) / slots;
    double eff = (double)blocks / (double)(rounds * slots);
    double score = eff - 0.004 * s;            // prefer fewer splits (less atomic traffic) at equal fill
    if (blocks < slots / 2) score -= 0.5;      // never leave more than half the chip idle
    if (score > best_score) { best_score = score; best = s; }
  }
  return best;
}

template <int BM, int BN, int WGM, int WGN, int BK, int NSTAGE, int PR = 0>
constexpr int lds_bytes(int epi) {
  constexpr int ring = NSTAGE * (BM + BN) * BK * 2;
  constexpr int patch = WGM * WGN * epi_wave_lds<BN / WGN>();
  return (epi == 0 && patch > ring) ? patch : ring;
}

template <int BM, int BN, int WGM, int WGN, int BK, int NSTAGE, bool AK, bool BK_, int EPI, bool SPREAD = false, int PR = 0, int MINW = 1, int NWL = 0>
int launch(const GemmParams& p, hipStream_t s) {
  int grid = p.tiles_m * p.tiles_n * p.splitk;
  constexpr int smem = lds_bytes<BM, BN, WGM, WGN, BK, NSTAGE, PR>(EPI);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, WGM, WGN, BK, NSTAGE, AK, BK_, EPI, SPREAD, PR, MINW, NWL>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_kernel<BM, BN, WGM, WGN, BK, NSTAGE, AK, BK_, EPI, SPREAD, PR, MINW, NWL>), dim3(grid), dim3(64 * WGM * WGN), smem, s, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { avt_set_error("avt_gemm: launch failed: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

template <int BM, int BN, int WGM, int WGN, int BK, int NSTAGE, int EPI, bool SPREAD = false, int PR = 0, int MINW = 1, int NWL = 0>
int dispatch_layout(const GemmParams& p, int a_kmajor, int b_kmajor, hipStream_t s) {
  if (a_kmajor && b_kmajor) return launch<BM, BN, WGM, WGN, BK, NSTAGE, true, true, EPI, SPREAD, PR, MINW, NWL>(p, s);
  if (a_kmajor && !b_kmajor) return launch<BM, BN, WGM, WGN, BK, NSTAGE, true, false, EPI, SPREAD, PR, MINW, NWL>(p, s);
  if (!a_kmajor && !b_kmajor) return launch<BM, BN, WGM, WGN, BK, NSTAGE, false, false, EPI, SPREAD, PR, MINW, NWL>(p, s);
  return launch<BM, BN, WGM, WGN, BK, NSTAGE, false, true, EPI, SPREAD, PR, MINW, NWL>(p, s);
}

template <int BM, int BN, int WGM, int WGN, int BK, int NSTAGE, bool SPREAD = false, int PR = 0, int MINW = 1, int NWL = 0>
int dispatch_epi(GemmParams& p, int epi, int a_kmajor, int b_kmajor, int splitk, hipStream_t s) {
  p.tiles_m = (p.M + BM - 1) / BM; p.tiles_n = (p.N + BN - 1) / BN;
  const int nk = (p.K + BK - 1) / BK;
  if (splitk <= 0) {                 // auto: about two blocks' worth of work per CU slot
    splitk = 1;
    if (epi >= 1) splitk = pick_splitk((long)p.tiles_m * p.tiles_n, nk, (BM * BN >= 256 * 256) ? 1 : 2, 512 / BK);
  }
  if (splitk > nk) splitk = nk;
  p.splitk = splitk;
  if (epi == 2) {                    // deterministic accumulate: only the weight-gradient layout (both operands reduction-major)
    if (a_kmajor || b_kmajor) { avt_set_error("avt_gemm_accum_bf16: operands must both be stored reduction-index-major"); return -1; }
    if ((size_t)p.tiles_m * p.tiles_n * splitk * BM * BN * 4 > p.ws_bytes) { avt_set_error("avt_gemm_accum_bf16: workspace too small (%zu bytes needed)", (size_t)p.tiles_m * p.tiles_n * splitk * BM * BN * 4); return -2; }
    int rc = launch<BM, BN, WGM, WGN, BK, NSTAGE, false, false, 2, SPREAD, PR, MINW, NWL>(p, s);
    return rc ? rc : launch_reduce<BM / WGM / 32, BN / WGN / 32, WGM, WGN>(p, s);
  }
  return epi ? dispatch_layout<BM, BN, WGM, WGN, BK, NSTAGE, 1, SPREAD, PR, MINW, NWL>(p, a_kmajor, b_kmajor, s)
             : dispatch_layout<BM, BN, WGM, WGN, BK, NSTAGE, 0, SPREAD, PR, MINW, NWL>(p, a_kmajor, b_kmajor, s);
}


// ---- 8-phase kernel: 256x256x64 tile, two wave groups half a phase apart, half-tile ring 1.5 K tiles deep ---------
// The K tile is consumed in four phases, one 64x32 quadrant of the 128x64 wave tile each; every phase is
//     L: ds_read the operand sub-tiles the quadrant still needs, issue 2 LDS-DMA instructions (1/8 of one 16-KB
//        half-tile), s_waitcnt vmcnt(8), s_barrier
//     M: s_waitcnt lgkmcnt(0), 8 x v_mfma_f32_32x32x16_bf16 at raised priority, s_barrier
// Group 1 (waves 4-7, the partners of waves 0-3 on the four SIMDs) runs one barrier behind group 0, so on every SIMD
// one wave is in M (matrix pipe) while the other is in L (LDS / texture-address pipes).
// LDS = 8 half-tile slots of 16 KB (kind x K-tile parity).  Half-tile kinds are cut so that need order == stage order:
//     A0h = rows {g*128 + 0..63},   A1h = rows {g*128 + 64..127}   (g = wave group, 128 rows each)
//     B0h = cols {w*64 + 0..31},    B1h = cols {w*64 + 32..63}     (w = wave column 0..3, 128 cols each)
// Quadrant order (A0,B0) (A0,B1) (A1,B1) (A1,B0); the phases of tile t read {A0h(t), B0h(t)}, B1h(t), A1h(t), nothing -- or,
// in the balanced variant, A0h(t), B1h(t), A1h(t), B0h(t+1) (two alternating B0 register sets) -- and phase P stages
// S(P+6) of the sequence S = A0h(0), B0h(0), B1h(0), A1h(0), A0h(1), ... -- every half-tile is in flight for 4-6 phases
// and the slot it lands in was last read >= 2 phases earlier.  The counted vmcnt before the L barrier of phase P retires
// this wave's share of everything phase P+1 reads (vmcnt(8) = 4 younger stages x 2 instructions stay in flight; 6 before
// the phase that reads the next tile's B0h, which was staged only 4 phases earlier); the M barrier that follows publishes
// it to both groups before anyone reads it.
// Stages past the end of the reduction are still issued, with an out-of-range source (zero fill), so the counts hold.
template <bool A_KMAJOR, bool B_KMAJOR, int EPI, bool GTAB = false>
__global__ __launch_bounds__(512) void gemm_8p_kernel(GemmParams p) {
  constexpr int BM = 256, BN = 256, BK = 64, WM = 128, WN = 64, TM = 4, TN = 2;
  constexpr int HALF = 128 * BK * 2;                       // 16 KB
  extern __shared__ __attribute__((aligned(16))) char smem8[];
  // GTAB (fc1 forward): the first 24 KB of the LDS hold the {GELU, GELU'} table for the whole kernel (its byte offsets then fit the
  // 16-bit halves the epilogue computes them in); the operand ring and the epilogue patches start behind it
  char* const lds = smem8 + (GTAB ? GELU_TAB_BYTES : 0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wn = wave & 3;
  if constexpr (GTAB) {       // 24 LDS-DMA instructions of 1 KB, three per wave, issued before the operand prologue (so its counted waits cover them)
    __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)g_gelu_tab, 0, GELU_TAB_BYTES, 0x00020000);
#pragma unroll
    for (int i = 0; i < GELU_TAB_BYTES / (8 * 1024); ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, AVT_LDS_PTR(smem8 + (wave * (GELU_TAB_BYTES / 8192) + i) * 1024), 16,
                                               (uint32_t)((wave * (GELU_TAB_BYTES / 8192) + i) * 1024 + lane * 16), 0, 0, 0);
  }

  const int ntile = p.tiles_m * p.tiles_n;
  const int bid = blockIdx.x;
  const int lb = xcd_remap(bid, ntile * p.splitk);          // an XCD owns a contiguous range of (split, tile): see xcd_remap
  const int split = lb / ntile;
  const int t_ = lb - split * ntile;
  // tile order inside the XCD's contiguous range.  Row-major: the ~32 tiles an XCD has in flight cover all column tiles, so
  // the whole B operand cycles through its 4-MB L2; when B is larger than that (N = 3072, K = 768: 4.7 MB) every tile re-fetches
  // its B block from the Infinity Cache (measured 6.2 GB of fabric reads for 0.78 GB of operands, profiles/r03_strip_order.txt).
  // Column strips: all row panels for strip_w column tiles, then the next strip -- the strip of B stays L2-resident and A is
  // streamed once per strip.
  int tm_i, tn_i;
  if (p.strip_w > 0) {
    const int per = p.tiles_m * p.strip_w;
    const int strip = t_ / per, r_ = t_ - strip * per;
    const int w_ = min(p.strip_w, p.tiles_n - strip * p.strip_w);
    tm_i = r_ / w_; tn_i = strip * p.strip_w + (r_ - tm_i * w_);
  } else { tm_i = t_ / p.tiles_n; tn_i = t_ - tm_i * p.tiles_n; }
  const int tm0 = tm_i * BM;
  const int tn0 = tn_i * BN;
  const int nk_total = (p.K + BK - 1) / BK;
  // (32-bit arithmetic: nk_total * splitk < 2^31 for every operand below 4 GiB; the 64-bit form expands to ~200 scalar
  // instructions at the start of every workgroup)
  const int kt_begin = p.splitk == 1 ? 0 : (int)((unsigned)nk_total * (unsigned)split / (unsigned)p.splitk);
  const int kt_end = p.splitk == 1 ? nk_total : (int)((unsigned)nk_total * (unsigned)(split + 1) / (unsigned)p.splitk);
  const int nk = kt_end - kt_begin;

  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.b_bytes, 0x00020000);

  // per-lane source offsets (bytes) of this wave's two DMA instructions per half-tile, K tile 0.  Rows / columns past the
  // matrix edge need no zero fill (they only feed output rows / columns that are never stored): a k-major row past the end is
  // out of the descriptor's range anyway, a column chunk of a k-strided operand that lies entirely past the edge is clamped to the
  // last chunk that still holds a valid column (lda / ldb are multiples of 8, so a partial chunk stays inside its row).
  // Only the K tail must read as zero: tiles >= nk use a descriptor with num_records = 0, so the in-loop address work is
  // one scalar select of the descriptor and one vector add per instruction (no per-lane masks, no branches).
  uint32_t offA[2][2], offB[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (A_KMAJOR) {
        int r = j * 64 + wave * 8 + (lane >> 3);                         // LDS row of the half-tile
        int c = (lane & 7) ^ kmajor_swz<BK>(r);
        int row = tm0 + j * 128 + (r & 63) + h * 64;
        offA[h][j] = (uint32_t)(((size_t)row * (size_t)p.lda + (size_t)(kt_begin * BK + c * 8)) * 2);
      } else {
        int r = j * 32 + wave * 4 + (lane >> 4);                         // k row
        int cl = ((lane & 15) ^ kstrided_swz<128>(r)) * 8;               // LDS column of the half-tile
        int col = tm0 + (cl >> 6) * 128 + (cl & 63) + h * 64;
        if (col >= p.M) col = (p.M - 1) & ~7;                            // chunk entirely past the edge -> re-read the last (possibly partial) one
        offA[h][j] = (uint32_t)(((size_t)(kt_begin * BK + r) * (size_t)p.lda + (size_t)col) * 2);
      }
      if (B_KMAJOR) {
        int r = j * 64 + wave * 8 + (lane >> 3);
        int c = (lane & 7) ^ kmajor_swz<BK>(r);
        int row = tn0 + (r >> 5) * 64 + (r & 31) + h * 32;
        offB[h][j] = (uint32_t)(((size_t)row * (size_t)p.ldb + (size_t)(kt_begin * BK + c * 8)) * 2);
      } else {
        int r = j * 32 + wave * 4 + (lane >> 4);
        int cl = ((lane & 15) ^ kstrided_swz<128>(r)) * 8;
        int col = tn0 + (cl >> 5) * 64 + (cl & 31) + h * 32;
        if (col >= p.N) col = (p.N - 1) & ~7;
        offB[h][j] = (uint32_t)(((size_t)(kt_begin * BK + r) * (size_t)p.ldb + (size_t)col) * 2);
      }
    }
  const uint32_t kstepA = A_KMAJOR ? (uint32_t)(BK * 2) : (uint32_t)((size_t)BK * p.lda * 2);
  const uint32_t kstepB = B_KMAJOR ? (uint32_t)(BK * 2) : (uint32_t)((size_t)BK * p.ldb * 2);
  char* const dstA = lds + (A_KMAJOR ? wave * 8 * (BK * 2) : wave * 4 * 256);
  char* const dstB = lds + (B_KMAJOR ? wave * 8 * (BK * 2) : wave * 4 * 256);
  constexpr int JSTEP = 8192;                              // second DMA instruction lands 64 rows x 128 B (or 32 k rows x 256 B) further
  __amdgpu_buffer_rsrc_t ra_null = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0, 0x00020000);
  __amdgpu_buffer_rsrc_t rb_null = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, 0, 0x00020000);

  // slot index = kind * 2 + (tile & 1); kinds 0 = A0h, 1 = B0h, 2 = B1h, 3 = A1h
  auto stage_a = [&](int h, int tile) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t r = (tile < nk) ? ra : ra_null;
    char* d = dstA + ((h ? 3 : 0) * 2 + (tile & 1)) * HALF;
    const uint32_t adv = (uint32_t)tile * kstepA;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, AVT_LDS_PTR(d + j * JSTEP), 16, offA[h][j] + adv, 0, 0, 0);
  };
  auto stage_b = [&](int h, int tile) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t r = (tile < nk) ? rb : rb_null;
    char* d = dstB + ((h ? 2 : 1) * 2 + (tile & 1)) * HALF;
    const uint32_t adv = (uint32_t)tile * kstepB;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, AVT_LDS_PTR(d + j * JSTEP), 16, offB[h][j] + adv, 0, 0, 0);
  };

  bf16x8_t fa[2][4], fb0[4], fb1[4];
  // k-strided operands: per-lane LDS addresses of the lane's element in k row (g>>1)*8 + (i>>2) of the 32-column blocks it reads
  // (see frag_kstrided: chunk = ((block ^ (i>>2)) << 2) | (g&1)*2 | ((i&3)>>1)); the slot, the k-step and the half are
  // immediates of the (inline-assembly) reads; slots 4-7 go through a second base 64 KB further
  uint32_t trA[2][2] = {{0u, 0u}, {0u, 0u}}, trB[2] = {0u, 0u};
  {
    const int g = lane >> 4, i16 = lane & 15;
    const int lane_off = ((g >> 1) * 8 + (i16 >> 2)) * 256 + (i16 & 1) * 8;
    const int x = (g & 1) * 2 + ((i16 & 3) >> 1);
    if (!A_KMAJOR) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        trA[0][i] = lds_addr32(lds) + (uint32_t)(lane_off + (((((grp * 2 + i) ^ (i16 >> 2)) & 3) << 2) | x) * 16);
        trA[1][i] = trA[0][i] + 65536u;
      }
    }
    if (!B_KMAJOR) {
      trB[0] = lds_addr32(lds) + (uint32_t)(lane_off + ((((wn ^ (i16 >> 2)) & 3) << 2) | x) * 16);
      trB[1] = trB[0] + 65536u;
    }
  }
  auto read_a = [&](bf16x8_t (&f)[2][4], int h, int par) __attribute__((always_inline)) {
    if (A_KMAJOR) {
      const char* slot = lds + ((h ? 3 : 0) * 2 + par) * HALF;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) f[i][ks] = frag_kmajor<BK>(slot + grp * 64 * (BK * 2), i, ks, lane);
    } else {
      switch ((h ? 3 : 0) * 2 + par) {          // slots 0, 1, 6, 7 (h and par are literals at every call site)
        case 0: frag4_tr_na<0>(f[0], trA[0][0]); frag4_tr_na<0>(f[1], trA[0][1]); break;
        case 1: frag4_tr_na<HALF>(f[0], trA[0][0]); frag4_tr_na<HALF>(f[1], trA[0][1]); break;
        case 6: frag4_tr_na<6 * HALF - 65536>(f[0], trA[1][0]); frag4_tr_na<6 * HALF - 65536>(f[1], trA[1][1]); break;
        default: frag4_tr_na<7 * HALF - 65536>(f[0], trA[1][0]); frag4_tr_na<7 * HALF - 65536>(f[1], trA[1][1]); break;
      }
    }
  };
  auto read_b = [&](bf16x8_t (&f)[4], int h, int par) __attribute__((always_inline)) {
    if (B_KMAJOR) {
      const char* slot = lds + ((h ? 2 : 1) * 2 + par) * HALF;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) f[ks] = frag_kmajor<BK>(slot + wn * 32 * (BK * 2), 0, ks, lane);
    } else {
      switch ((h ? 2 : 1) * 2 + par) {          // slots 2, 3, 4, 5
        case 2: frag4_tr_na<2 * HALF>(f, trB[0]); break;
        case 3: frag4_tr_na<3 * HALF>(f, trB[0]); break;
        case 4: frag4_tr_na<4 * HALF - 65536>(f, trB[1]); break;
        default: frag4_tr_na<5 * HALF - 65536>(f, trB[1]); break;
      }
    }
  };

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define P8_BARRIER() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
// fragment reads first, LDS-DMA second: an LDS-DMA blocks the issuing wave for ~100 cycles, the reads only queue
#define P8_PIN() __builtin_amdgcn_sched_barrier(0)
#define P8_MFMA(FA, FB, I0, J)                                                                          \
  do {                                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    __builtin_amdgcn_s_setprio(1);                                                                       \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                     \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                      \
        acc[(I0) + i][J] = mma<EPI>(FA[i][ks], FB[ks], acc[(I0) + i][J]);                                \
    __builtin_amdgcn_s_setprio(0);                                                                       \
  } while (0)

  // prologue: S(0..5)
  stage_a(0, 0); stage_b(0, 0); stage_b(1, 0); stage_a(1, 0); stage_a(0, 1); stage_b(0, 1);
  wait_vmcnt<8>();
  P8_BARRIER();
  if (grp == 1) P8_BARRIER();
  // two schedules of the fragment reads: with both operands k-major (ds_read_b128 fragments) the reads are spread 8/4/8/4
  // over the phases (B0 of the next tile fetched in phase 3 into a second register set: +2-3 %); with a transposing-read
  // operand (twice the LDS instructions per fragment) the plain 12/4/8/0 order measured 2-5 % faster
  if constexpr (A_KMAJOR && B_KMAJOR) {
    bf16x8_t fb0n[4];
    read_b(fb0, 0, 0);
    for (int t = 0; t < nk; t += 2) {
      // ---- even tile t (slot parity 0): B0 in fb0, next tile's B0 -> fb0n ----
      read_a(fa, 0, 0); P8_PIN(); stage_b(1, t + 1); wait_vmcnt<8>(); P8_BARRIER();
      P8_MFMA(fa, fb0, 0, 0); P8_BARRIER();
      read_b(fb1, 1, 0); P8_PIN(); stage_a(1, t + 1); wait_vmcnt<8>(); P8_BARRIER();
      P8_MFMA(fa, fb1, 0, 1); P8_BARRIER();
      read_a(fa, 1, 0); P8_PIN(); stage_a(0, t + 2); wait_vmcnt<6>(); P8_BARRIER();
      P8_MFMA(fa, fb1, 2, 1); P8_BARRIER();
      read_b(fb0n, 0, 1); P8_PIN(); stage_b(0, t + 2); wait_vmcnt<8>(); P8_BARRIER();
      P8_MFMA(fa, fb0, 2, 0); P8_BARRIER();
      // ---- odd tile t+1 (slot parity 1): B0 in fb0n, next tile's B0 -> fb0; when nk is odd this runs once on zero-filled
      //      slots (no mid-loop exit: it would split the accumulators' live ranges and cost a copy of all of them per trip) ----
      read_a(fa, 0, 1); P8_PIN(); stage_b(1, t + 2); wait_vmcnt<8>(); P8_BARRIER();
      P8_MFMA(fa, fb0n, 0, 0); P8_BARRIER();
      read_b(fb1, 1, 1); P8_PIN(); stage_a(1, t + 2); wait_vmcnt<8>(); P8_BARRIER();
      P8_MFMA(fa, fb1, 0, 1); P8_BARRIER();
      read_a(fa, 1, 1); P8_PIN(); stage_a(0, t + 3); wait_vmcnt<6>(); P8_BARRIER();
      P8_MFMA(fa, fb1, 2, 1); P8_BARRIER();
      read_b(fb0, 0, 0); P8_PIN(); stage_b(0, t + 3); wait_vmcnt<8>(); P8_BARRIER();
      P8_MFMA(fa, fb0n, 2, 0); P8_BARRIER();
    }
  } else {
    for (int t = 0; t < nk; t += 2) {
      // ---- even tile t (slot parity 0) ----
      read_a(fa, 0, 0); read_b(fb0, 0, 0); P8_PIN(); stage_b(1, t + 1); wait_vmcnt<8>(); P8_BARRIER();
      P8_MFMA(fa, fb0, 0, 0); P8_BARRIER();
      read_b(fb1, 1, 0); P8_PIN(); stage_a(1, t + 1); wait_vmcnt<8>(); P8_BARRIER();
      P8_MFMA(fa, fb1, 0, 1); P8_BARRIER();
      read_a(fa, 1, 0); P8_PIN(); stage_a(0, t + 2); wait_vmcnt<8>(); P8_BARRIER();
      P8_MFMA(fa, fb1, 2, 1); P8_BARRIER();
      stage_b(0, t + 2); wait_vmcnt<8>(); P8_BARRIER();
      P8_MFMA(fa, fb0, 2, 0); P8_BARRIER();
      // ---- odd tile t+1 (slot parity 1) ----
      read_a(fa, 0, 1); read_b(fb0, 0, 1); P8_PIN(); stage_b(1, t + 2); wait_vmcnt<8>(); P8_BARRIER();
      P8_MFMA(fa, fb0, 0, 0); P8_BARRIER();
      read_b(fb1, 1, 1); P8_PIN(); stage_a(1, t + 2); wait_vmcnt<8>(); P8_BARRIER();
      P8_MFMA(fa, fb1, 0, 1); P8_BARRIER();
      read_a(fa, 1, 1); P8_PIN(); stage_a(0, t + 3); wait_vmcnt<8>(); P8_BARRIER();
      P8_MFMA(fa, fb1, 2, 1); P8_BARRIER();
      stage_b(0, t + 3); wait_vmcnt<8>(); P8_BARRIER();
      P8_MFMA(fa, fb0, 2, 0); P8_BARRIER();
    }
  }
  wait_vmcnt<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (grp == 0) P8_BARRIER();
  P8_BARRIER();                                          // every LDS-DMA has landed and every fragment read retired: LDS is free
  int lane_e = lane, m0_e = tm0 + grp * WM, n0_e = tn0 + wn * WN;
  asm volatile("" : "+v"(lane_e), "+s"(m0_e), "+s"(n0_e));   // keep the epilogue's address arithmetic out of the K loop's register budget
  gemm_epilogue<TM, TN, WM, WN, EPI, 0, GTAB, A_KMAJOR && B_KMAJOR>(p, acc, lds, wave, lane_e, m0_e, n0_e, smem8);
#undef P8_PIN
#undef P8_MFMA
#undef P8_BARRIER
}

template <bool AK, bool BK_, int EPI, bool GTAB = false>
int launch_8p(const GemmParams& p, hipStream_t s) {
  int grid = p.tiles_m * p.tiles_n * p.splitk;
  constexpr int smem = ((EPI == 0 && 8 * epi_wave_lds<64>() > 8 * 128 * 64 * 2) ? 8 * epi_wave_lds<64>() : 8 * 128 * 64 * 2)   // 128 KiB ring | 134 KiB epilogue
                       + (GTAB ? GELU_TAB_BYTES : 0);
  static_assert(smem <= 160 * 1024, "8-phase kernel: LDS");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_8p_kernel<AK, BK_, EPI, GTAB>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_8p_kernel<AK, BK_, EPI, GTAB>), dim3(grid), dim3(512), smem, s, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { avt_set_error("avt_gemm: launch failed: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

// persist: 0 = never (tile 808: gemm_8p_kernel itself), 1 = where gemm_persist.hip covers the shape and measures faster (tile 0), 2 = forced (tile 809)
int dispatch_8p(GemmParams& p, int epi, int a_kmajor, int b_kmajor, int splitk, hipStream_t s, int persist = 0) {
  p.tiles_m = (p.M + 255) / 256; p.tiles_n = (p.N + 255) / 256;
  const int nk = (p.K + 63) / 64;
  if (splitk <= 0) {
    splitk = 1;
    if (epi >= 1) splitk = pick_splitk((long)p.tiles_m * p.tiles_n, nk, 1, 8);
  }
  if (splitk > nk) splitk = nk;
  p.splitk = splitk;
  p.strip_w = 0;
  if (epi == 0 && p.tiles_m >= 64) {
    // column strips when the B operand does not fit next to the A working set in an XCD's 4-MB L2: as many column tiles per
    // strip as keep the strip of B under ~2.5 MB (the last strip takes the remainder: when the remainder is narrow -- a
    // column strip narrower than 3 tiles re-reads A too often -- the strips are evened out)
    const size_t b_bytes = (size_t)p.N * p.K * 2;
    if (b_bytes > (size_t)4 << 20) {       // (3.5 MB -- the qkv weight -- still lives in L2: strips cost +3 % there)
      int w = (int)(((size_t)5 << 19) / ((size_t)256 * p.K * 2)); if (w < 1) w = 1;
      const int nstrip = (p.tiles_n + w - 1) / w;
      w = (p.tiles_n + nstrip - 1) / nstrip;
      if (w >= 3 && w < p.tiles_n) p.strip_w = w;
    }
  }
  if (epi == 2) {
    if (a_kmajor || b_kmajor) { avt_set_error("avt_gemm_accum_bf16: operands must both be stored reduction-index-major"); return -1; }
    if ((size_t)p.tiles_m * p.tiles_n * splitk * 65536 * 4 > p.ws_bytes) { avt_set_error("avt_gemm_accum_bf16: workspace too small (%zu bytes needed)", (size_t)p.tiles_m * p.tiles_n * splitk * 65536 * 4); return -2; }
    int rc = launch_8p<false, false, 2>(p, s);
    return rc ? rc : launch_reduce<4, 2, 2, 4>(p, s);
  }
  if (!persist && (p.c2_frag || p.aux_frag)) { avt_set_error("avt_gemm_bf16: a fragment-major C2 / aux (ldc2 == 0 / ldaux == 0) needs the persistent kernel (tile 0 or 809)"); return -1; }
  if (persist) {
    // the persistent form (gemm_persist.hip) where it covers the shape and the epilogue: 0 = not covered
    int mask = 0xF;
    const int rc = (epi == 0 && a_kmajor && b_kmajor && splitk == 1) ? avt_gemm_persist(p, mask, persist == 2, s) : 0;
    if (rc != 0) return rc < 0 ? rc : 0;
    if (p.c2_frag || p.aux_frag) { avt_set_error("avt_gemm_bf16: a fragment-major C2 / aux (ldc2 == 0 / ldaux == 0) needs the persistent kernel: ask avt_gemm_frag_ok(M, N, K) first"); return -1; }
    if (persist == 2) { avt_set_error("avt_gemm_bf16: tile 809 (persistent 8-phase kernel) covers bf16 outputs of k-major operands with N %% 256 == 0, K %% 128 == 0, >= 512 tiles and a bias / GELU / residual / saved-derivative epilogue"); return -1; }
  }
  if (epi == 0) {
    // fc1 forward (erf GELU, with or without the derivative output, nothing else in the epilogue): activation by LDS table
    // (measured against the polynomial + exponential form of rounds 2-3, 256 clips: 2853 vs 2975 us per launch, 900.4 vs 896.2 clips/s)
    if (a_kmajor && b_kmajor && p.act == 1 && !p.out_f32 && p.wide_ok && p.N % 8 == 0 && !p.res && !p.colsum && !p.drop_thresh)
      return launch_8p<true, true, 0, true>(p, s);
    if (a_kmajor && b_kmajor) return launch_8p<true, true, 0>(p, s);
    if (a_kmajor && !b_kmajor) return launch_8p<true, false, 0>(p, s);
    if (!a_kmajor && !b_kmajor) return launch_8p<false, false, 0>(p, s);
    return launch_8p<false, true, 0>(p, s);
  }
  if (a_kmajor && b_kmajor) return launch_8p<true, true, 1>(p, s);
  if (a_kmajor && !b_kmajor) return launch_8p<true, false, 1>(p, s);
  if (!a_kmajor && !b_kmajor) return launch_8p<false, false, 1>(p, s);
  return launch_8p<false, true, 1>(p, s);
}

// (Round 3, measured and removed: a 4-wave 256x128x32 kernel, TWO workgroups per CU, so that one's epilogue runs under the other's K loop -- bit-identical
// results, 65-86 % of every epilogue under the partner's K loop, and slower: fc1 forward 3648 vs 3274 us; profiles/r03_tile_timeline.txt, DESIGN.md section 4.
// The source is part of tools/lab/avt_lab_hooks.diff.)

// ---- 4-wave weight-gradient kernel: 256x256x32 tile, wave tile 128x128, accumulators in the AGPR half of the file ----------
// C[m, n] += sum_k A[k, m] B[k, n], both operands stored reduction-index-major (dy^T x), split-K slabs (EPI 2).
// The 8-phase kernel spends 24 KB of transposing LDS reads per wave and K tile (wave tile 128x64: 4 + 2 fragments per 8 MFMAs)
// and its matrix pipe is busy 0.48-0.51 of the time on these launches (profiles/r03a_pmc_sq.txt) although neither LDS nor HBM
// is near a limit: the L phases (twice the LDS instructions of a k-major operand) are longer than the partner's M phases.
// With one wave per SIMD (4 waves, 512 registers each: 256 accumulators as AGPRs) the wave tile is 128x128 -- 4 + 4 fragments per
// 16 MFMAs, a third fewer LDS bytes per flop -- and, since a wave's own LDS reads and LDS-DMA issue overlap its MFMAs
// (profiles/r03_issue_rules.txt), no second wave is needed to keep the pipe fed as long as the K loop holds no vector-ALU work:
// fragment addresses and DMA offsets are per-lane constants, the K advance lives in the scalar buffer descriptor.
// Ring: 5 stages of [32 k][256 + 256] bf16 = 160 KB (the whole LDS; this epilogue needs none), 4 stages in flight.
// fragment F of a k-step (0: A0, 1..4: B0..B3, 5..7: A1..A3): its per-lane LDS address and its place in the register arrays, selected at
// compile time (the macro form with ?: indexed the arrays out of bounds in its dead branches: 700 -Warray-bounds warnings per build)
template <int F>
__device__ __forceinline__ uint32_t w4_frag_addr(const uint32_t (&a)[4], const uint32_t (&b)[4]) {
  if constexpr (F == 0) return a[0]; else if constexpr (F <= 4) return b[F - 1]; else return a[F - 4];
}
template <int F>
__device__ __forceinline__ void w4_frag_put(bf16x8_t (&af)[4], bf16x8_t (&bf)[4], bf16x8_t v) {
  if constexpr (F == 0) af[0] = v; else if constexpr (F <= 4) bf[F - 1] = v; else af[F - 4] = v;
}
// (cache policy of the two operand streams: default on both -- nt on the dY stream / the x stream / both measured 0 / -0.7 / -0.9 % on the step, profiles/r05e_w4_cache_policy.txt)
template <int EPI, int DIRECT = 0>
__global__ __launch_bounds__(256) void gemm_w4_kernel(GemmParams p) {
  static_assert(EPI == 2, "4-wave weight-gradient kernel: split-K slab epilogue only");
  constexpr int BM = 256, BN = 256, BK = 32, NST = 5, WM = 128, WN = 128, TM = 4, TN = 4;
  constexpr int ROWB = 512;                                  // bytes of one k row of a [32][256] operand tile
  constexpr int OP_T = BK * ROWB, STAGE = 2 * OP_T;          // 16 KB + 16 KB
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntile = p.tiles_m * p.tiles_n;
  const int lb = accum_slab(p);                               // split * tiles + row-major tile (walk order: gemm_tile.hpp)
  const int split = lb / ntile;
  const int t_ = lb - split * ntile;
  const int tm0 = (t_ / p.tiles_n) * BM;
  const int tn0 = (t_ % p.tiles_n) * BN;
  const int nk_total = (p.K + BK - 1) / BK;
  // (32-bit arithmetic: nk_total * splitk < 2^31 for every operand below 4 GiB; the 64-bit form expands to ~200 scalar
  // instructions at the start of every workgroup)
  const int kt_begin = p.splitk == 1 ? 0 : (int)((unsigned)nk_total * (unsigned)split / (unsigned)p.splitk);
  const int kt_end = p.splitk == 1 ? nk_total : (int)((unsigned)nk_total * (unsigned)(split + 1) / (unsigned)p.splitk);
  const int nk = kt_end - kt_begin;

  // LDS-DMA: one instruction = 2 k rows x 512 B; wave w stages rows q*8 + w*2 .. +1 (q = 0..3) of A and of B.  A column
  // chunk that lies entirely past the matrix edge is clamped to the last chunk holding a valid column (never stored).
  uint32_t voffA[4], voffB[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = q * 8 + wave * 2 + (lane >> 5);
    const int c = (lane & 31) ^ kstrided_swz<256>(r);
    int ca = tm0 + c * 8, cb = tn0 + c * 8;
    if (ca >= p.M) ca = (p.M - 1) & ~7;
    if (cb >= p.N) cb = (p.N - 1) & ~7;
    voffA[q] = (uint32_t)(((size_t)r * (size_t)p.lda + (size_t)ca) * 2);
    voffB[q] = (uint32_t)(((size_t)r * (size_t)p.ldb + (size_t)cb) * 2);
  }
  char* const dst = lds + wave * 2 * ROWB;
  // K advance through the descriptor (scalar arithmetic only): a running base pointer and a running byte bound per operand, moved
  // one stage forward after each stage's eight DMA instructions; rows past the end of the reduction -- and whole stages past
  // this split's range -- read as zeros (bound 0)
  // (everything here is wave-uniform; readfirstlane makes that provable, otherwise the descriptors end up in vector registers
  // and every LDS-DMA instruction turns into a waterfall loop)
  auto sgpr = [](uint32_t v) __attribute__((always_inline)) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
  const uint32_t stepA = sgpr((uint32_t)BK * (uint32_t)p.lda * 2u), stepB = sgpr((uint32_t)BK * (uint32_t)p.ldb * 2u);
  const uint64_t pa0 = (uint64_t)(uintptr_t)p.A + (uint64_t)kt_begin * stepA, pb0 = (uint64_t)(uintptr_t)p.B + (uint64_t)kt_begin * stepB;
  uint32_t pa_lo = sgpr((uint32_t)pa0), pa_hi = sgpr((uint32_t)(pa0 >> 32)), pb_lo = sgpr((uint32_t)pb0), pb_hi = sgpr((uint32_t)(pb0 >> 32));
  uint32_t remA = sgpr((uint64_t)kt_begin * stepA < p.a_bytes ? (uint32_t)(p.a_bytes - (uint64_t)kt_begin * stepA) : 0u);
  uint32_t remB = sgpr((uint64_t)kt_begin * stepB < p.b_bytes ? (uint32_t)(p.b_bytes - (uint64_t)kt_begin * stepB) : 0u);
  int left = __builtin_amdgcn_readfirstlane(nk);                      // stages of this split not yet handed to the DMA
  __amdgpu_buffer_rsrc_t ra_t, rb_t;
  auto next_stage = [&]() __attribute__((always_inline)) {            // descriptors of the next stage to fetch, then advance
    const bool live = left > 0;                                       // a live stage starts inside the operand: no underflow below
    ra_t = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((uint64_t)sgpr(pa_hi) << 32) | sgpr(pa_lo)), 0, sgpr(live ? remA : 0u), 0x00020000);
    rb_t = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((uint64_t)sgpr(pb_hi) << 32) | sgpr(pb_lo)), 0, sgpr(live ? remB : 0u), 0x00020000);
    const uint32_t na = pa_lo + stepA, nb = pb_lo + stepB;
    pa_hi += (na < pa_lo) ? 1u : 0u; pb_hi += (nb < pb_lo) ? 1u : 0u; pa_lo = na; pb_lo = nb;
    remA -= stepA; remB -= stepB;
    --left;
  };
  auto dma = [&](int q, int slot) __attribute__((always_inline)) {    // q = 0..3: A, 4..7: B, of the stage next_stage() prepared
    char* d = dst + slot * STAGE;
    if (q < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_t, AVT_LDS_PTR(d + q * 8 * ROWB), 16, voffA[q], 0, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_t, AVT_LDS_PTR(d + OP_T + (q - 4) * 8 * ROWB), 16, voffB[q - 4], 0, 0, 0);
  };
  // transposing fragment reads (frag_kstrided<256>): lane (g = l >> 4, i = l & 15) reads k rows ks*16 + (g>>1)*8 + (i>>2) + 4h
  // (h = 0, 1), columns tile*32 + (g&1)*16 + (i&3)*4 .. +3; the 16-B chunk index is swizzled with (row & 3) << 2 = (i>>2) << 2,
  // i.e. chunk = ((tile ^ (i>>2)) << 2) | (g&1)*2 | ((i&3)>>1): one per-lane address per 32-column block, rows as immediates.
  // Slots 0-1, 2-3 and 4 get their own base registers (the ds_read immediate reaches 64 KB).  The addresses are made opaque
  // so that they stay in registers instead of being re-added inside the loop (a vector-ALU instruction there costs MFMA time).
  uint32_t adA[3][TM], adB[3][TN];
  {
    const int g = lane >> 4, i16 = lane & 15;
    const int rsub = (g >> 1) * 8 + (i16 >> 2);
    const int x = (g & 1) * 2 + ((i16 & 3) >> 1);
    const int lane_off = rsub * ROWB + (i16 & 1) * 8;
    const uint32_t base = lds_addr32(lds);
#pragma unroll
    for (int b_ = 0; b_ < 3; ++b_)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int ta = wm * 4 + t, tb = wn * 4 + t;
        uint32_t oa = base + (uint32_t)(b_ * 2 * STAGE + lane_off + (((ta ^ (i16 >> 2)) << 2) | x) * 16);
        uint32_t ob = base + (uint32_t)(b_ * 2 * STAGE + OP_T + lane_off + (((tb ^ (i16 >> 2)) << 2) | x) * 16);
        asm volatile("" : "+v"(oa), "+v"(ob));
        adA[b_][t] = oa; adB[b_][t] = ob;
      }
  }
  bf16x8_t af[2][TM], bfr[2][TN];

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define W4_PIN() __builtin_amdgcn_sched_barrier(0)
  // fragment f of the next k-step: f = 0: A0, 1..4: B0..B3, 5..7: A1..A3 (the order the MFMAs need them)
#define W4_RDF(NB, SLOT, KS, F)                                                                                       \
  do {                                                                                                                \
    constexpr int imm_ = ((SLOT) & 1) * STAGE + (KS) * 16 * ROWB;                                                     \
    const uint32_t a_ = w4_frag_addr<(F)>(adA[(SLOT) >> 1], adB[(SLOT) >> 1]);                                        \
    const u32x2_t lo_ = ds_read_tr_na<imm_>(a_), hi_ = ds_read_tr_na<imm_ + 4 * ROWB>(a_);   /* inline asm: no compiler-placed vmcnt(0) */ \
    w4_frag_put<(F)>(af[NB], bfr[NB], tr_join(lo_, hi_));                                                             \
  } while (0)
  // 16 MFMAs of buffer CB; between them the 8 fragments of (SLOT, KS) into buffer NB and the DMA instructions Q0..Q0+3 of stage ST
#define W4_STEP(CB, NB, SLOT, KS, Q0, ST, READ)                                                                       \
  do {                                                                                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      /* buffer CB's fragments (inline-assembly reads) have landed */ \
    W4_PIN();                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                                    \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                                \
        acc[i][j] = mma<EPI>(af[CB][i], bfr[CB][j], acc[i][j]);                                                       \
        W4_PIN();                                                                                                     \
        if (READ) {                                                                                                   \
          if (i * 4 + j == 0) W4_RDF(NB, SLOT, KS, 0);  if (i * 4 + j == 1) W4_RDF(NB, SLOT, KS, 1);                  \
          if (i * 4 + j == 2) W4_RDF(NB, SLOT, KS, 2);  if (i * 4 + j == 3) W4_RDF(NB, SLOT, KS, 3);                  \
          if (i * 4 + j == 4) W4_RDF(NB, SLOT, KS, 4);  if (i * 4 + j == 6) W4_RDF(NB, SLOT, KS, 5);                  \
          if (i * 4 + j == 8) W4_RDF(NB, SLOT, KS, 6);  if (i * 4 + j == 10) W4_RDF(NB, SLOT, KS, 7);                 \
        }                                                                                                             \
        if ((i * 4 + j) % 4 == 1 && (Q0) >= 0) dma((Q0) + (i * 4 + j) / 4, (ST));            /* ST = destination slot */   \
        W4_PIN();                                                                                                     \
      }                                                                                                               \
  } while (0)
  // stage S in slot SLOT: wait for its DMA (3 younger stages = 24 instructions stay in flight), barrier, then
  //   fragments (S, k-step 0) under the MFMAs of (S-1, k-step 1), fragments (S, k-step 1) under the MFMAs of (S, k-step 0);
  //   the 8 DMA instructions of stage S+4 (slot of stage S-1, free since the barrier) are spread over both halves
#define W4_STAGE(S, SLOT)                                                                                             \
  do {                                                                                                                \
    wait_vmcnt<24>();                                                                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                \
    W4_PIN(); asm volatile("s_barrier" ::: "memory"); W4_PIN();                                                       \
    next_stage();                                                                                                     \
    W4_STEP(1, 0, SLOT, 0, 0, ((SLOT) + 4) % NST, true);                                                              \
    W4_STEP(0, 1, SLOT, 1, 4, ((SLOT) + 4) % NST, true);                                                              \
  } while (0)

  // prologue: stages 0..3 in flight; the first stage has no previous k-step to multiply: its k-step 0 fragments are read plainly
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    next_stage();
#pragma unroll
    for (int q = 0; q < 8; ++q) dma(q, st);
  }
  wait_vmcnt<24>();
  W4_PIN(); asm volatile("s_barrier" ::: "memory"); W4_PIN();
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    if (f == 0) W4_RDF(0, 0, 0, 0); if (f == 1) W4_RDF(0, 0, 0, 1); if (f == 2) W4_RDF(0, 0, 0, 2); if (f == 3) W4_RDF(0, 0, 0, 3);
    if (f == 4) W4_RDF(0, 0, 0, 4); if (f == 5) W4_RDF(0, 0, 0, 5); if (f == 6) W4_RDF(0, 0, 0, 6); if (f == 7) W4_RDF(0, 0, 0, 7);
  }
  next_stage();
#pragma unroll
  for (int q = 0; q < 4; ++q) dma(q, 4);
  W4_PIN();
  W4_STEP(0, 1, 0, 1, 4, 4, true);                        // MFMAs of (0, k-step 0), fragments of (0, k-step 1), second half of stage 4's DMA
  int s = 1;
  for (; s + 4 < nk; s += 5) {                            // slots 1, 2, 3, 4, 0
    W4_STAGE(s, 1); W4_STAGE(s + 1, 2); W4_STAGE(s + 2, 3); W4_STAGE(s + 3, 4); W4_STAGE(s + 4, 0);
  }
  if (s < nk) { W4_STAGE(s, 1); ++s; }
  if (s < nk) { W4_STAGE(s, 2); ++s; }
  if (s < nk) { W4_STAGE(s, 3); ++s; }
  if (s < nk) { W4_STAGE(s, 4); ++s; }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  W4_PIN();
  W4_STEP(1, 0, 0, 0, -1, 0, false);                      // k-step 1 of the last stage
  wait_vmcnt<0>();
#undef W4_STAGE
#undef W4_STEP
#undef W4_RDF
#undef W4_PIN
  gemm_epilogue<TM, TN, WM, WN, EPI, 0, false, false, DIRECT>(p, acc, lds, wave, lane, tm0 + wm * WM, tn0 + wn * WN);
}

int dispatch_w4(GemmParams& p, int a_kmajor, int b_kmajor, int splitk, hipStream_t s) {
  if (a_kmajor || b_kmajor) { avt_set_error("avt_gemm_accum_bf16: operands must both be stored reduction-index-major"); return -1; }
  p.tiles_m = (p.M + 255) / 256; p.tiles_n = (p.N + 255) / 256;
  const int nk64 = (p.K + 63) / 64;
  // (at least 16 K tiles = 1024 rows per split: at the reference's 3 clips per GPU -- 5910 rows -- 7-9 splits of 650-850 rows are mostly prologue, slab and
  //  reduce: 5 splits are 3-4 us per call faster, profiles/r06s_wgrad_split_sweep.txt; from 8 clips on the choice is what it was.  The workspace query
  //  assumes 8 tiles per split: an upper bound)
  if (splitk <= 0) splitk = pick_splitk((long)p.tiles_m * p.tiles_n, nk64, 1, 16);
  if (splitk > nk64) splitk = nk64;
  p.splitk = splitk;
  p.tile_cm = p.tiles_n > p.tiles_m;
  if (splitk > 1 && (size_t)p.tiles_m * p.tiles_n * splitk * 65536 * 4 > p.ws_bytes) { avt_set_error("avt_gemm_accum_bf16: workspace too small (%zu bytes needed)", (size_t)p.tiles_m * p.tiles_n * splitk * 65536 * 4); return -2; }
  constexpr int smem = 5 * 2 * 32 * 512;               // 160 KB
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_w4_kernel<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute((const void*)gemm_w4_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute((const void*)gemm_w4_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  if (splitk == 1 && p.c_assign) hipLaunchKernelGGL((gemm_w4_kernel<2, 2>), dim3(p.tiles_m * p.tiles_n), dim3(256), smem, s, p);      // writes its tiles into C itself
  else if (splitk == 1) hipLaunchKernelGGL((gemm_w4_kernel<2, 1>), dim3(p.tiles_m * p.tiles_n), dim3(256), smem, s, p);             // adds its tiles into C itself
  else hipLaunchKernelGGL((gemm_w4_kernel<2, 0>), dim3(p.tiles_m * p.tiles_n * splitk), dim3(256), smem, s, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { avt_set_error("avt_gemm: launch failed: %s", hipGetErrorString(e)); return (int)e; }
  return splitk == 1 ? 0 : launch_reduce<4, 4, 2, 2>(p, s);            // (splitk == 1: the kernel added its tiles into C itself)
}

}  // namespace

static int gemm_dispatch(GemmParams& p, int bm, int epi, int a_kmajor, int b_kmajor, int splitk, int K, hipStream_t s);
static int gemm_impl(const void* A, int a_kmajor, int lda, const void* B, int b_kmajor, int ldb,
                     void* C, int ldc, int M, int N, int K,
                     const float* bias, int act, const void* aux, int ldaux,
                     void* C2, int ldc2, const void* res, int ldres, int res_period,
                     float drop_p, uint64_t drop_seed, float* colsum,
                     int out_mode, int splitk, int tile, void* ws, size_t ws_bytes, float* part, size_t part_bytes, void* stream,
                     const float* ln_stat = nullptr, const float* ln_c = nullptr, float* stat_part = nullptr, int c_assign = 0) {
  AVT_CHECK(A && B && C, "avt_gemm_bf16: null operand");
  AVT_CHECK(M > 0 && N > 0 && K > 0, "avt_gemm_bf16: bad dims M=%d N=%d K=%d", M, N, K);
  AVT_CHECK(aligned16(A) && aligned16(B) && aligned16(C), "avt_gemm_bf16: operands must be 16-byte aligned");
  AVT_CHECK(lda % 8 == 0 && ldb % 8 == 0, "avt_gemm_bf16: lda/ldb must be multiples of 8 (got %d, %d)", lda, ldb);
  AVT_CHECK(K % 8 == 0 || (!a_kmajor && !b_kmajor), "avt_gemm_bf16: K must be a multiple of 8 for k-major operands (K=%d)", K);
  AVT_CHECK(out_mode >= 0 && out_mode <= 3, "avt_gemm_bf16: out_mode must be 0 (bf16), 1 (fp32) or 2 (fp32 atomic accumulate)");
  AVT_CHECK(out_mode != 3 || (ws && aligned16(ws)), "avt_gemm_accum_bf16: needs a 16-byte aligned workspace");
  AVT_CHECK(out_mode != 3 || (size_t)M * (size_t)ldc * 4 < 0xFFFFFFF0ull, "avt_gemm_accum_bf16: C larger than 4 GiB");      // (the one-split epilogue addresses C through a buffer descriptor)
  AVT_CHECK(act >= 0 && act <= 3, "avt_gemm_bf16: bad act %d", act);
  AVT_CHECK(act < 3 || aux, "avt_gemm_bf16: act %d needs aux", act);
  AVT_CHECK(drop_p >= 0.f && drop_p < 1.f, "avt_gemm_bf16: bad dropout p");
  GemmParams p{};
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.C2 = (bf16_t*)C2; p.ws = (float*)ws; p.ws_bytes = ws_bytes;
  p.bias = bias; p.res = (const bf16_t*)res; p.aux = (const bf16_t*)aux; p.colsum = colsum;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldc2 = ldc2; p.ldres = ldres; p.ldaux = ldaux;
  p.res_period = res_period; p.act = act; p.out_f32 = (out_mode == 1);
  p.drop_thresh = drop_threshold(drop_p); p.drop_scale = 1.0f / (1.0f - drop_p); p.drop_seed = drop_seed;
  p.ln_stat = ln_stat; p.ln_c = ln_c; p.stat_part = stat_part;
  p.c_assign = (c_assign && out_mode == 3) ? 1 : 0;
  // ldc2 == 0 / ldaux == 0: the fragment-major private layout of the persistent kernel (include/avt_hip.h, gemm_persist.hip)
  p.c2_frag = (C2 && ldc2 == 0) ? 1 : 0; p.aux_frag = (aux && ldaux == 0) ? 1 : 0;
  if (p.c2_frag || p.aux_frag) {
    AVT_CHECK(out_mode == 0 && a_kmajor && b_kmajor && (tile == 0 || tile == 809), "avt_gemm_bf16: a fragment-major C2 / aux needs a bf16 output, k-major operands and tile 0 | 809");
    AVT_CHECK(!p.c2_frag || act == 1, "avt_gemm_bf16: a fragment-major C2 goes with the erf-GELU epilogue (act 1)");
    AVT_CHECK(!p.aux_frag || act == 3, "avt_gemm_bf16: a fragment-major aux goes with the saved-derivative epilogue (act 3)");
    AVT_CHECK(avt_gemm_frag_bytes(M, N) < 0xFFFFFFF0ull && aligned16(C2) && aligned16(aux), "avt_gemm_bf16: fragment-major tensor misaligned or larger than 4 GiB");
  }
  if (ln_stat || ln_c || stat_part) {
    // LayerNorm folded into the GEMMs around it (avt_gemm_ln_bf16): activation epilogue of k-major operands only
    AVT_CHECK(out_mode <= 1 && a_kmajor && b_kmajor, "avt_gemm_ln_bf16: the LayerNorm-fold modes need out_mode 0 | 1 and both operands k-major");
    AVT_CHECK(!ln_c || ln_stat, "avt_gemm_ln_bf16: ln_c needs ln_stat");
    AVT_CHECK(!ln_c || ((act == 0 || act == 1) && !res && !colsum && !stat_part && drop_p == 0.f), "avt_gemm_ln_bf16: the fold (ln_c) goes with a bias | erf-GELU epilogue only");
    AVT_CHECK(ln_c || !ln_stat || (act == 3 && !stat_part), "avt_gemm_ln_bf16: ln_stat without ln_c scales the rows of a saved-derivative epilogue (act 3)");
    AVT_CHECK(!stat_part || (act == 0 && !colsum && !C2 && drop_p == 0.f && out_mode == 0 && N % 32 == 0 && ldc % 8 == 0 && (!res || ldres % 8 == 0)),
              "avt_gemm_ln_bf16: row statistics (stat_part) go with a bf16 bias (+ residual) epilogue, N %% 32 == 0 and 16-byte row strides");
    AVT_CHECK((!ln_stat || (((uintptr_t)ln_stat) & 7) == 0) && (!ln_c || aligned16(ln_c)) && (!stat_part || aligned16(stat_part)), "avt_gemm_ln_bf16: misaligned statistics");
  }
  size_t a_rows = a_kmajor ? (size_t)M : (size_t)K, b_rows = b_kmajor ? (size_t)N : (size_t)K;
  size_t ab = a_rows * (size_t)lda * 2, bb = b_rows * (size_t)ldb * 2;
  AVT_CHECK(ab < 0xFFFFFFF0ull && bb < 0xFFFFFFF0ull, "avt_gemm_bf16: operand larger than 4 GiB");
  p.a_bytes = (uint32_t)ab; p.b_bytes = (uint32_t)bb;
  hipStream_t s = (hipStream_t)stream;
  const int epi = (out_mode == 2) ? 1 : (out_mode == 3 ? 2 : 0);
  if (epi == 0) {
    AVT_CHECK(N % 4 == 0 && ldc % 4 == 0 && (!C2 || ldc2 % 4 == 0) && (!res || ldres % 4 == 0) && (!aux || ldaux % 4 == 0),
              "avt_gemm_bf16: N and ldc/ldc2/ldres/ldaux must be multiples of 4 for the activation epilogue");
    p.wide_ok = (ldc % 8 == 0) && (!C2 || ldc2 % 8 == 0) && (!res || ldres % 8 == 0) && (!aux || ldaux % 8 == 0);
    AVT_CHECK(splitk <= 1, "avt_gemm_bf16: split-K needs out_mode 2");
    // the second operand of the epilogue is fetched through a buffer descriptor with 32-bit byte offsets
    AVT_CHECK(!aux || (size_t)M * (size_t)ldaux * 2 < 0xFFFFFFF0ull, "avt_gemm_bf16: aux larger than 4 GiB");
    AVT_CHECK(!res || (size_t)(res_period ? res_period : M) * (size_t)ldres * 2 < 0xFFFFFFF0ull, "avt_gemm_bf16: res larger than 4 GiB");
  } else {
    AVT_CHECK(!bias && !act && !C2 && !res && !colsum && drop_p == 0.f, "avt_gemm_bf16: accumulate mode has no fused epilogue");
  }
  int bm = tile;
  if (bm == 0) {
    long t256 = (long)((M + 255) / 256) * ((N + 255) / 256);
    long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (epi == 0) {                                                       // epi 1 | 2: below
      // Small token counts (late round 5, profiles/r05z_small_batch_gemm_sweep.txt: the ViT GEMMs at 3 / 8 / 16 clips per GPU, the head's at 2560 rows).
      // All-k-major contractions: from 96 output tiles of 256 x 256 on the 8-phase kernel wins even though it leaves CUs idle (186 tiles: 27 / 66 us
      // against 43 / 96 us on 744 tiles of 128 x 128); below that, with K <= 3072, the 3-deep 64 x 64 ring beats the 128 x 128 tiles whose 1.1-1.25
      // rounds waste most of a second round (72 / 282 / 1116 tiles: 23.4 / 27.2 / 17.7 us at K = 768).  Longer reductions and the other layouts keep
      // the 128 x 128 tiles there (head, 2560 x 2048 x 8192: 115 us against 125 on the 8-phase kernel and 191 on 64 x 64).
      const bool kk = a_kmajor && b_kmajor;
      if (t256 >= 200 || (kk && K % 64 == 0 && t256 >= 96)) bm = 256;
      else if (t128 >= 192 && !(kk && K <= 3072)) bm = 128;
      else bm = 64;
    }
    else {
      long sk = ((K + 63) / 64) / 4; if (sk < 1) sk = 1; if (sk > 64) sk = 64;
      bm = (t256 * sk >= 256 && t256 < 4096) ? 256 : 128;
    }
  }
  if (tile == 0 && bm == 256 && (K % 64 == 0 || (!a_kmajor && !b_kmajor))) bm = 8080;      // default big-tile kernel: the 8-phase schedule, persistent where that is faster
  if (tile == 0 && bm == 8080 && epi == 2) bm = 2565;                                          // weight gradients: 4 waves of 128x128 (+2-3 % over the 8-phase kernel)
  // at most 64 output rows of k-major A rows (the head at the reference's 3 clips per GPU -- 30 rows at T = 10, 45 at T = 15 --, the CLS-only last ViT block): the skinny kernel -- N / 32
  // workgroups, one ordered MFMA chain each, an 18-stage ring: same bits, 2-3x the weight stream of the 64 x 64 tiles (profiles/r06j_skinny_gemm.txt).  33-64 rows (two
  // row tiles, 13 stages): only with k-major weights -- with B stored [K][N] the 64 x 64 ring is as fast or faster there (profiles/r06w_skinny64.txt)
  if (tile == 0 && bm == 64 && epi == 0 && a_kmajor && (M <= 32 || (M <= 64 && b_kmajor)) && (b_kmajor || (uint64_t)bb + 64ull * (uint64_t)ldb * 2ull < (1ull << 32))) bm = 32;
  if (tile == 0 && bm == 64 && epi == 0 && a_kmajor && (b_kmajor || (long)((M + 63) / 64) * ((N + 63) / 64) <= 256)) bm = 643;                                // small outputs of k-major rows: 3-deep ring (+15-25 % on the head's data gradients; late round 5: also with B stored [K][N] while the tiles fit one round -- the head's forward at 30 .. 160 rows: 60 -> 46 us at K = 8192; at 2560 rows the 2-deep ring is the faster one there)
  AVT_CHECK(!(p.c2_frag || p.aux_frag) || bm == 8080 || bm == 809,
            "avt_gemm_bf16: a fragment-major C2 / aux (ldc2 == 0 / ldaux == 0) needs the persistent kernel, which does not take this shape: ask avt_gemm_frag_ok(M, N, K) first");
  int nslots = 0;
  if (colsum && part) {
    // one partial row per wave row of the grid: every tile shape here has two wave rows per tile
    const int BM = (bm == 64 || bm == 643) ? 64 : (bm == 128 ? 128 : 256);
    nslots = bm == 32 ? 1 : ((M + BM - 1) / BM) * 2;             // (the skinny kernel: one wave row in all)
    AVT_CHECK(aligned16(part) && part_bytes >= (size_t)nslots * N * 4, "avt_gemm_bf16: partials workspace too small or misaligned (%zu bytes needed)", (size_t)nslots * N * 4);
    p.colsum_part = part;
  }
  // (Round 6, measured and removed: "tail round on small tiles" -- a big-tile GEMM whose last round of 256 workgroups is at most half full cut along M, the
  // whole rounds on the 8-phase kernel, the remaining rows on 128 x 128 tiles in a second launch.  Same bits, and slower at every batch: 16 clips per GPU
  // 25.3 -> 26.1 ms, 256 clips 261.2 -> 262.9 ms (profiles/r06m_tail_split.txt).  The dispatcher hands a finished CU its next tile by itself, a half-empty
  // last round runs at a higher clock, and the small-tile kernel needs about a big-tile round for those rows anyway.)
  int rc = gemm_dispatch(p, bm, epi, a_kmajor, b_kmajor, splitk, K, s);
  if (rc == 0 && nslots) { float* outs[1] = {colsum}; rc = avt_reduce_partials(part, nslots, N, outs, 1, s); }
  return rc;
}

extern "C" size_t avt_gemm_colsum_workspace_bytes(int M, int N, int tile) {
  // mirrors gemm_impl's automatic tile choice for the activation epilogue; two wave rows per tile
  int BM = (tile == 64 || tile == 643) ? 64 : (tile == 128 ? 128 : 256);
  if (tile == 0) {
    // (the choice below 200 tiles depends on K and the layouts, which this query does not see: the 64-row bound covers every tile there)
    const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256);
    BM = (t256 >= 200) ? 256 : 64;
  }
  return (size_t)(((M + BM - 1) / BM) * 2) * (size_t)N * 4;
}

static int gemm_dispatch(GemmParams& p, int bm, int epi, int a_kmajor, int b_kmajor, int splitk, int K, hipStream_t s) {
  switch (bm) {
    case 64:  return dispatch_epi<64, 64, 2, 2, 64, 2>(p, epi, a_kmajor, b_kmajor, splitk, s);
    case 128: return dispatch_epi<128, 128, 2, 2, 64, 2>(p, epi, a_kmajor, b_kmajor, splitk, s);
    case 643: return dispatch_epi<64, 64, 2, 2, 64, 3>(p, epi, a_kmajor, b_kmajor, splitk, s);     // 3-deep ring
    case 32:
      if (epi != 0 || splitk > 1) { avt_set_error("avt_gemm_bf16: tile 32 (skinny kernel) has the activation epilogue only"); return -1; }
      return dispatch_skinny(p, a_kmajor, b_kmajor, s);
    // (4- and 6-deep rings measured no better than the 3-deep one on the head's 30 .. 160-row GEMMs: profiles/r05zc_tiny_m_gemm_sweep.txt)
    case 256:                                                                                     // one barrier per K tile, dribbled LDS-DMA issued by 4 loader waves
      if (K % 64 == 0 || (!a_kmajor && !b_kmajor)) return dispatch_epi<256, 256, 2, 4, 64, 2, true, 0, 1, 4>(p, epi, a_kmajor, b_kmajor, splitk, s);
      return dispatch_epi<256, 256, 2, 4, 64, 2, true>(p, epi, a_kmajor, b_kmajor, splitk, s);
    case 2568: return dispatch_epi<256, 256, 2, 4, 64, 2, true>(p, epi, a_kmajor, b_kmajor, splitk, s);       // all 8 waves issue LDS-DMA
    case 2565:                                                                                    // 4-wave weight-gradient kernel (128x128 wave tiles, AGPR accumulators)
      if (epi != 2) { avt_set_error("avt_gemm: tile 2565 is the deterministic weight-gradient kernel (avt_gemm_accum_bf16 only)"); return -1; }
      return dispatch_w4(p, a_kmajor, b_kmajor, splitk, s);                                  // 4 waves, 256x128x32, two workgroups per CU
    case 808: case 8080: case 809:                                                                // 8-phase schedule (needs K % 64 == 0 for k-major operands)
      if (K % 64 == 0 || (!a_kmajor && !b_kmajor)) return dispatch_8p(p, epi, a_kmajor, b_kmajor, splitk, s, bm == 808 ? 0 : (bm == 809 ? 2 : 1));
      if (bm == 809) { avt_set_error("avt_gemm_bf16: tile 809 needs K %% 64 == 0"); return -1; }
      return dispatch_epi<256, 256, 2, 4, 64, 2, true>(p, epi, a_kmajor, b_kmajor, splitk, s);
    default: break;
  }
  avt_set_error("avt_gemm_bf16: tile must be 0 (choose), 32 (M <= 64), 64, 128, 643, 256 / 2568 (one barrier per K tile), 808 (8-phase) or 809 (8-phase, persistent) (got %d)", bm);
  return -1;
}

extern "C" int avt_gemm_bf16(const void* A, int a_kmajor, int lda, const void* B, int b_kmajor, int ldb,
                             void* C, int ldc, int M, int N, int K,
                             const float* bias, int act, const void* aux, int ldaux,
                             void* C2, int ldc2, const void* res, int ldres, int res_period,
                             float drop_p, uint64_t drop_seed, float* colsum,
                             int out_mode, int splitk, int tile, float* part, size_t part_bytes, void* stream) {
  AVT_CHECK(out_mode != 3, "avt_gemm_bf16: out_mode must be 0 (bf16), 1 (fp32) or 2 (fp32 atomic accumulate)");
  return gemm_impl(A, a_kmajor, lda, B, b_kmajor, ldb, C, ldc, M, N, K, bias, act, aux, ldaux, C2, ldc2, res, ldres, res_period,
                   drop_p, drop_seed, colsum, out_mode, splitk, tile, nullptr, 0, part, part_bytes, stream);
}

extern "C" int avt_gemm_ln_bf16(const void* A, int a_kmajor, int lda, const void* B, int b_kmajor, int ldb,
                                void* C, int ldc, int M, int N, int K,
                                const float* bias, int act, const void* aux, int ldaux,
                                void* C2, int ldc2, const void* res, int ldres, int res_period,
                                float drop_p, uint64_t drop_seed, float* colsum,
                                int out_mode, int splitk, int tile, float* part, size_t part_bytes,
                                const float* ln_stat, const float* ln_c, float* stat_part, void* stream) {
  AVT_CHECK(out_mode != 3, "avt_gemm_ln_bf16: out_mode must be 0 (bf16) or 1 (fp32)");
  return gemm_impl(A, a_kmajor, lda, B, b_kmajor, ldb, C, ldc, M, N, K, bias, act, aux, ldaux, C2, ldc2, res, ldres, res_period,
                   drop_p, drop_seed, colsum, out_mode, splitk, tile, nullptr, 0, part, part_bytes, stream, ln_stat, ln_c, stat_part);
}

extern "C" int avt_gemm_accum_bf16(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K,
                                   int splitk, int tile, void* workspace, size_t workspace_bytes, void* stream) {
  return gemm_impl(A, 0, lda, B, 0, ldb, C, ldc, M, N, K, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, 0.f, 0, nullptr,
                   3, splitk, tile, workspace, workspace_bytes, nullptr, 0, stream);
}

extern "C" int avt_gemm_assign_bf16(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K,
                                    int splitk, int tile, void* workspace, size_t workspace_bytes, void* stream) {
  return gemm_impl(A, 0, lda, B, 0, ldb, C, ldc, M, N, K, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, 0.f, 0, nullptr,
                   3, splitk, tile, workspace, workspace_bytes, nullptr, 0, stream, nullptr, nullptr, nullptr, 1);
}

extern "C" size_t avt_gemm_accum_workspace_bytes(int M, int N, int K) {
  // mirrors the automatic (tile = 0, splitk = 0) choice of gemm_impl for the accumulate epilogues
  const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256), t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
  const int nk = (K + 63) / 64;
  long sk = nk / 4; if (sk < 1) sk = 1; if (sk > 64) sk = 64;
  const bool big = (t256 * sk >= 256 && t256 < 4096);
  const long tiles = big ? t256 : t128;
  int s = pick_splitk(tiles, nk, big ? 1 : 2, 8); if (s > nk) s = nk;
  return (size_t)tiles * (size_t)s * (big ? 65536u : 16384u) * 4u;
}
