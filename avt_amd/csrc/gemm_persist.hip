// Persistent form of the 8-phase bf16 GEMM (gemm.hip: gemm_8p_kernel) for the large ViT contractions with both operands
// k-major: C[M,N] = epilogue(A[M,K] . B[N,K]^T), 256x256x64 tiles, eight waves, one workgroup per CU.
//
// Why: with K = 768 an output tile is 12 K tiles = 24.6 k cycles of MFMA work per SIMD, and gemm_8p_kernel spends another
// 14-33 k cycles per tile with the matrix pipe idle (tools/gemm_timeline.py, profiles/r03_tile_timeline.txt): operand fill
// after the workgroup starts, the epilogue, the drain of its stores before the workgroup can retire, the launch of the next
// workgroup on the CU.  Here a workgroup stays on its CU and walks a list of output tiles; the reduction stream is continuous
// across tiles:
//   * the LDS-DMA stages that gemm_8p_kernel issues past the end of the reduction (zero fill, to keep its counted waits uniform)
//     fetch the NEXT output tile's first K tile instead (A0h, B0h, B1h, A1h -> the four parity-0 slots of the ring);
//   * the epilogue then runs in the four parity-1 slots (64 KB) plus the 32 KB (8 KB next to the GELU table) behind the ring,
//     while those 64 KB of operands arrive; its patches are 128-byte-pitch, XOR-swizzled (4 KB per 32x64 block instead of
//     4.25 KB) so that two patches -- or one patch and one second-operand buffer -- fit a wave's 8-KB share of a slot;
//   * the stores of a tile drain under the next tile's first phases (the first counted wait that has to see them retired
//     is the one of phase 2), nothing waits for them at a workgroup boundary, and the kernel arguments, descriptors, per-lane
//     offsets and the GELU table are set up once per workgroup instead of once per tile.
// The second operand of the epilogue (residual / saved derivative) of a tile's first 32-row block is requested two phases
// before the reduction ends (into the space behind the ring, idle during the K loop); the bias strip is fetched at the start
// of the last iteration.
// Results are bit-identical to gemm_8p_kernel's (same MFMA order, same epilogue arithmetic in the same order).
//
// Tile order and scheduling: the XCD-aware order of gemm.hip (an XCD owns a contiguous range of tiles, column strips when B does
// not fit an L2), handed out dynamically: one ticket counter per XCD in device memory, a workgroup (which reads its XCC_ID) draws the
// next tile of its XCD's range.  A workgroup that finds a CU only late -- the collectives of a data-parallel step occupy CUs while
// the backward GEMMs run -- draws what is left or nothing; a static assignment would make such a launch take twice as long.
// When its own range is exhausted a workgroup goes on with the ranges of XCDs xcc + 1, xcc + 2, ... (mod 8): every tile is computed
// wherever the workgroups land -- a partitioned device, a CU mask that empties an XCD, or any part whose XCC_ID does not span 0..7
// (round-4 advisor finding: the static split left such ranges unwritten).  In the normal case that is the launch's tail balancing itself.
// The ticket for the tile after next is requested by wave 0 during the epilogue (a returning atomic issued before the tile's stores,
// so that in-order retirement has it back by the counted wait at the epilogue's end) and reaches the other waves through LDS at the
// barrier that ends the epilogue; the last workgroup to leave zeroes the counters for the next launch on the stream.
// (Built with -mllvm -amdgpu-atomic-optimizer-strategy=None: the optimizer would turn the one-lane atomic into a wave reduction
// followed by an immediate s_waitcnt vmcnt(0).)
#include <mutex>
#include <unordered_map>
#include <vector>
#include "gemm_tile.hpp"

namespace {

// (Measured and removed, round 4: running the epilogues without a second operand in two slots + the space behind the ring, so that the next tile's
// K tile 1 halves A0h / B0h are fetched before the epilogue as well -- six stages resident when a tile starts instead of four: 906.9 clips/s against
// 911.5 with the four-slot layout; a tile's first K-loop iteration stays 1.4x as long as the others either way, 6.7 k against 4.9 k cycles, so that time
// is not the wait for those two stages.  profiles/r04_persistent_gemm.txt; the switch is part of tools/lab/avt_lab_hooks.diff.)
// Fragment-major second output / second operand (late round 5; EPK 8 / 9 write it, EPK 10 / 11 read it; chosen by ldc2 == 0 / ldaux == 0 in the C ABI):
// the saved GELU' of fc1 forward (C2) is only ever read back by the fc2 data gradient (aux), a launch with the same M, N and the same 128 x 64 wave
// tiles -- a private tensor between two such kernels needs no row-major form.  Layout: per (128-row strip, 64-column group) 4 blocks x 4 KB, block i =
// [4 stores][64 lanes][16 B] straight from the accumulator layout (store st of lane l = rows' (j, q) pieces (st / 2, 2 (st % 2)) and (st / 2, 2 (st % 2) + 1)):
// no LDS patch round trip for the writer, 1-KB contiguous requests and conflict-free 16-byte LDS reads for the reader.  Same-box A/B
// (profiles/r05s_auxfrag.txt): fc1 forward -0.7 %, fc2 data gradient -2.5 % per launch, the step +0.3 %.
constexpr bool epk_gelu(int e) { return e == 1 || e == 6 || e == 8 || e == 9; }
constexpr bool epk_fold(int e) { return e == 5 || e == 6 || e == 9; }
constexpr bool epk_aux(int e) { return e == 3 || e == 7 || e == 10 || e == 11; }
constexpr bool epk_scale(int e) { return e == 7 || e == 11; }
constexpr bool epk_fragw(int e) { return e == 8 || e == 9; }
constexpr bool epk_fragr(int e) { return e == 10 || e == 11; }
constexpr int PK_HALF = 128 * 64 * 2;                      // one half-tile slot of the ring: 16 KB
// Longer reductions stay with gemm_8p_kernel unless the caller forces tile 809: what the persistent form removes is per-TILE time (fill,
// store drain, workgroup turn-over: 12-20 % of a K = 768 tile, 2-4 % of a K = 3072 tile), and the step runs at the board's power limit --
// at K >= 2304 the busier matrix pipe costs as much clock as the removed idle time is worth (measured, 256 clips, same box:
// 907.4 clips/s without, 920.1 with every shape persistent, 924.1 with K <= 1024 only; profiles/r04_persistent_gemm.txt).
// Round 5, after the epilogue stores lost their waterfall loops: every shape persistent 963.5 / 962.5 against 960.2 / 958.6 clips/s with K <= 1024 only
// (same box, profiles/r05j_persistent_all_k.txt): the limit now covers every reduction of ViT-B / ViT-L (K <= 4096).
constexpr int PK_KMAX = 4096;

// swizzled wave-private patch [32 rows][128 B]: the 8-byte position q8 (0..15) of row r lives at position q8 ^ (r & 15).
// Writes (accumulator layout: lane = row, 8 B per (j, q)): the 32 lanes of a half-wave hit 16 positions x 2 rows each = every
// bank twice (256 B in two cycles); reads (row strips: 8 lanes per row, two 8-byte halves): 4 rows x 8 positions, same.
__device__ __forceinline__ int patch_wr(int ml, int h, int j, int q) { return ml * 128 + (((8 * j + 2 * q + h) ^ (ml & 15)) << 3); }
__device__ __forceinline__ int patch_rd(int row, int pc, int half) { return row * 128 + (((2 * pc + half) ^ (row & 15)) << 3); }

// the lane id, re-derived where it is needed (volatile: not merged with an earlier copy, so no register holds it across the K loop)
__device__ __forceinline__ int pk_lane_id() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
// `tkt` != nullptr in wave 0 while tiles may be left: lane 0 draws the ticket for the tile after next (the value is back when the
// counted wait at the end of the epilogue has passed)
__device__ __forceinline__ int pk_ticket(int* tkt, int lane) {
  int tk = 0x7fffffff;
  if (tkt && lane == 0) tk = __hip_atomic_fetch_add(tkt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return tk;
}

// one synchronous device-scope ticket draw, issued through inline assembly: the compiler's wait-count bookkeeping never sees it (a returning
// atomic that it tracks, inside control flow, made it put an s_waitcnt vmcnt(0) behind the stores and LDS-DMA requests of every path: 13 -> 78
// such waits in this file).  Only used where the wave has nothing else to do: the launch's first draw and the move to another XCD's range.
__device__ __forceinline__ int pk_draw_sync(int* ctr) {
  int r;
  asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(ctr), "v"(1) : "memory");
  return r;
}

// A buffer descriptor whose words the compiler can PROVE wave-uniform (cdna_hip_programming.md T20): the tile origin is uniform in fact, but it
// is derived from values the register allocator keeps in vector registers across the K loop, and a descriptor in vector registers makes hipcc wrap
// EVERY buffer operation that uses it in a waterfall loop (4 v_readfirstlane, 2 compares, s_and_saveexec, the operation, a branch: the 16 / 32
// stores of a tile's epilogue ran serialised like that through round 4).  Passing the base and the size through readfirstlane once per tile
// puts the descriptor into scalar registers.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pk_uniform_rsrc(const void* base, uint32_t nrec) {
  const uint64_t a = (uint64_t)base;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(nrec), 0x00020000);
}

// 16-byte row-strip stores of a wave tile (lane = row rl = lane / 8 of an 8-row group, columns (lane & 7) * 8 ..): the per-lane part of the
// address is one register for the whole tile, the row group a scalar added per store (no 64-bit multiply-add, whose don't-care high half
// the register allocator once paired with a register still waiting for memory: an s_waitcnt vmcnt(0) after every store)
struct PkStore {
  __amdgpu_buffer_rsrc_t r; uint32_t voff, ld2;
  // `rows` = rows of the wave's 128-row strip that exist (>= 128 everywhere but in the last row tile): the descriptor ends behind them, so
  // a store to a row past M is dropped by the bounds check -- no predicate, no branch, and every wave issues the same number of stores
  // (the counted waits of the epilogue rely on that)
  __device__ __forceinline__ void init(bf16_t* origin, int ld, int lane, int rows) {
    ld2 = (uint32_t)ld * 2u;
    const uint32_t nrec = (uint32_t)(rows < 0 ? 0 : (rows > 128 ? 128 : rows)) * ld2;
    r = pk_uniform_rsrc(origin, nrec);
    voff = (uint32_t)(lane >> 3) * ld2 + (uint32_t)(lane & 7) * 16u;
  }
  __device__ __forceinline__ void st(int row8, u32x4_t v) const {      // row8 = first row of the 8-row group (wave-uniform)
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff + (uint32_t)row8 * ld2, 0, AVT_ST_AUX);
  }
};

// row-strip side of a block: read the lane's 16 bytes of rows it*8 + rl back and store them
template <bool TWO>
__device__ __forceinline__ void pk_store_block(const char* patch_c, const char* patch_d, int lane, int i32, const PkStore& sc, const PkStore& sd) {
  const int rl = lane >> 3, pc = lane & 7;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = it * 8 + rl;
    const u32x2_t lo = *(const u32x2_t*)(patch_c + patch_rd(row, pc, 0)), hi = *(const u32x2_t*)(patch_c + patch_rd(row, pc, 1));
    u32x2_t lo2 = lo, hi2 = hi;
    if (TWO) { lo2 = *(const u32x2_t*)(patch_d + patch_rd(row, pc, 0)); hi2 = *(const u32x2_t*)(patch_d + patch_rd(row, pc, 1)); }
    sc.st(i32 + it * 8, (u32x4_t){lo[0], lo[1], hi[0], hi[1]});
    if (TWO) sd.st(i32 + it * 8, (u32x4_t){lo2[0], lo2[1], hi2[0], hi2[1]});
  }
}

// (Round 5, measured and removed: stores straight from the accumulator layout -- one v_permlane32_swap per register pairs the half-waves' column
// groups into 16-byte pieces, cdna_hip_programming.md T21, no LDS patch -- write 32 bytes per row and instruction instead of the row strips'
// 128-byte lines: qkv forward 1579 -> 1927 us, fc1 forward with GELU + GELU' 2782 -> 4994 us, proj forward 595 -> 656 us, the step 909 -> 835
// clips/s (profiles/r05b_epilogue_stores.txt).  Full-line stores are worth their LDS round trip.)
// EPK 0: C = bf16(acc + bias)
// FOLD (EPK 5): the LayerNorm fold -- v = rstd[m] * acc + (bias[n] + (-mean[m] rstd[m]) * c[n]); `rst[i]` = the lane's row statistics of block i,
// `bias_l + 64` = the c strip
template <bool FOLD>
__device__ __forceinline__ void pk_epi_plain(const GemmParams& p, const f32x16_t (&acc)[4][2], char* patch, const float* bias_l,
                                             int lane, int row0, int col0, int mrem, const f32x2_t (&rst)[4]) {
  const int ml = lane & 31, h = lane >> 5;
  PkStore sc;
  sc.init((bf16_t*)p.C + (size_t)row0 * p.ldc + col0, p.ldc, lane, mrem);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2_t rr = (f32x2_t){rst[i][0], rst[i][0]}, tt = (f32x2_t){rst[i][1], rst[i][1]};
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nl = j * 32 + 8 * q + 4 * h;
        const f32x4_t b = *(const f32x4_t*)(bias_l + nl);
        f32x2_t v0, v1;
        if (FOLD) {
          const f32x4_t c = *(const f32x4_t*)(bias_l + 64 + nl);
          v0 = (f32x2_t){acc[i][j][4 * q], acc[i][j][4 * q + 1]} * rr + ((f32x2_t){c[0], c[1]} * tt + (f32x2_t){b[0], b[1]});
          v1 = (f32x2_t){acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]} * rr + ((f32x2_t){c[2], c[3]} * tt + (f32x2_t){b[2], b[3]});
        } else {
          v0 = (f32x2_t){acc[i][j][4 * q], acc[i][j][4 * q + 1]} + (f32x2_t){b[0], b[1]};
          v1 = (f32x2_t){acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]} + (f32x2_t){b[2], b[3]};
        }
        *(u32x2_t*)(patch + patch_wr(ml, h, j, q)) = (u32x2_t){pack2bf(v0[0], v0[1]), pack2bf(v1[0], v1[1])};
      }
    pk_store_block<false>(patch, patch, lane, i * 32, sc, sc);
  }
}

// exact values for the elements of a block that lie above the table (|x| >= 2^8, inf, NaN): WHICH = 0 writes GELU, 1 writes GELU'
template <int WHICH, bool FOLD>
__device__ __forceinline__ void pk_gelu_fix(const f32x16_t* blk, const float* bias_l, char* patch, int ml, int h, f32x2_t rs) {
#pragma unroll 1
  for (int j = 0; j < 2; ++j)
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
      const int nl = j * 32 + 8 * q + 4 * h;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pre = FOLD ? fmaf(blk[j][4 * q + e], rs[0], fmaf(bias_l[64 + nl + e], rs[1], bias_l[nl + e])) : blk[j][4 * q + e] + bias_l[nl + e];
        const float x = bf2f(f2bf(pre));                                        // the bf16-rounded pre-activation, as in the look-up
        if (!(fabsf(x) < GELU_TAB_TOP)) {
          bf16_t hb, db; gelu_big(x, hb, db);
          *(bf16_t*)(patch + patch_wr(ml, h, j, q) + e * 2) = WHICH ? db : hb;
        }
      }
    }
}
// EPK 1: C = GELU(acc + bias), C2 = GELU'(acc + bias), both by the LDS table (see gemm_tile.hpp: epi_fast_block, TAB)
// (one patch, used twice: GELU rows out, then GELU' rows out)
template <bool FOLD, bool FRAGW>
__device__ __forceinline__ void pk_epi_gelu_block(const GemmParams& p, const EpiBlk<2> blk_, char* patch, const float* bias_l,
                                                  int lane, int i32, const PkStore& sc, const PkStore& sd, const char* tab, f32x2_t rs) {
  const f32x2_t rr = (f32x2_t){rs[0], rs[0]}, tt = (f32x2_t){rs[1], rs[1]};
  const f32x16_t* blk = blk_.t;
  const int ml = lane & 31, h = lane >> 5;
  u16x2_t mx = {0, 0};
  u32x2_t dd[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    uint32_t off[8], e[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int nl = j * 32 + 8 * q + 4 * h;
      const f32x4_t b = *(const f32x4_t*)(bias_l + nl);
      f32x2_t v0, v1;
      if (FOLD) {
        const f32x4_t c = *(const f32x4_t*)(bias_l + 64 + nl);
        v0 = (f32x2_t){blk[j][4 * q], blk[j][4 * q + 1]} * rr + ((f32x2_t){c[0], c[1]} * tt + (f32x2_t){b[0], b[1]});
        v1 = (f32x2_t){blk[j][4 * q + 2], blk[j][4 * q + 3]} * rr + ((f32x2_t){c[2], c[3]} * tt + (f32x2_t){b[2], b[3]});
      } else {
        v0 = (f32x2_t){blk[j][4 * q], blk[j][4 * q + 1]} + (f32x2_t){b[0], b[1]};
        v1 = (f32x2_t){blk[j][4 * q + 2], blk[j][4 * q + 3]} + (f32x2_t){b[2], b[3]};
      }
      off[2 * q] = gelu_tab_offsets(pack2bf(v0[0], v0[1]), mx);
      off[2 * q + 1] = gelu_tab_offsets(pack2bf(v1[0], v1[1]), mx);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      e[2 * k] = *(const uint32_t*)(tab + (off[k] & 0xffffu));
      e[2 * k + 1] = *(const uint32_t*)(tab + (off[k] >> 16));
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t h0 = __builtin_amdgcn_perm(e[4 * q + 1], e[4 * q], 0x05040100u), d0 = __builtin_amdgcn_perm(e[4 * q + 1], e[4 * q], 0x07060302u);
      const uint32_t h1 = __builtin_amdgcn_perm(e[4 * q + 3], e[4 * q + 2], 0x05040100u), d1 = __builtin_amdgcn_perm(e[4 * q + 3], e[4 * q + 2], 0x07060302u);
      dd[j][q] = (u32x2_t){d0, d1};
      *(u32x2_t*)(patch + patch_wr(ml, h, j, q)) = (u32x2_t){h0, h1};
    }
  }
  constexpr unsigned short HI = ((GELU_TAB_ELO + GELU_TAB_NEXP) << 7) - 1;
  const bool big = __any((mx[0] > HI) | (mx[1] > HI));          // some value of this block lies above the table: patch those elements
  if (__builtin_expect(big, 0)) pk_gelu_fix<0, FOLD>(blk, bias_l, patch, ml, h, rs);
  pk_store_block<false>(patch, patch, lane, i32, sc, sc);
  if constexpr (FRAGW) {                                         // fragment-major derivative: straight from the registers (file comment)
    if (__builtin_expect(big, 0)) {                              // (rare: the exact values of the elements above the table, through the lane's own patch positions)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) *(u32x2_t*)(patch + patch_wr(ml, h, j, q)) = dd[j][q];
      pk_gelu_fix<1, FOLD>(blk, bias_l, patch, ml, h, rs);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) dd[j][q] = *(const u32x2_t*)(patch + patch_wr(ml, h, j, q));
    }
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int j = st >> 1, q0 = 2 * (st & 1);
      __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){dd[j][q0][0], dd[j][q0][1], dd[j][q0 + 1][0], dd[j][q0 + 1][1]}, sd.r,
                                             sd.voff + (uint32_t)(i32 * 128 + st * 1024), 0, AVT_ST_AUX);
    }
  } else if (p.C2) {                                                    // the same patch again for the derivative (the LDS pipe runs a wave's operations in order)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) *(u32x2_t*)(patch + patch_wr(ml, h, j, q)) = dd[j][q];
    if (__builtin_expect(big, 0)) pk_gelu_fix<1, FOLD>(blk, bias_l, patch, ml, h, rs);
    pk_store_block<false>(patch, patch, lane, i32, sd, sd);
  }
}
template <bool FOLD, bool FRAGW>
__device__ __forceinline__ void pk_epi_gelu(const GemmParams& p, const f32x16_t (&acc)[4][2], char* patch, const float* bias_l,
                                            int lane, int row0, int col0, const char* tab, int mrem, const f32x2_t (&rst)[4]) {
  PkStore sc, sd;
  sc.init((bf16_t*)p.C + (size_t)row0 * p.ldc + col0, p.ldc, lane, mrem);
  if constexpr (FRAGW) {
    // the wave's 16 KB of the fragment-major tensor; a strip wholly past M writes nothing (empty descriptor)
    sd.r = pk_uniform_rsrc((const char*)p.C2 + ((size_t)(row0 >> 7) * (size_t)(p.N >> 6) + (size_t)(col0 >> 6)) * 16384, mrem > 0 ? 16384u : 0u);
    sd.voff = (uint32_t)lane * 16u; sd.ld2 = 0u;
  } else {
    sd.init(p.C2 ? p.C2 + (size_t)row0 * p.ldc2 + col0 : (bf16_t*)p.C, p.ldc2, lane, mrem);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {          // (unrolled: a single copy of the block's code would need the block moved into place -- 32 more registers)
    EpiBlk<2> b;
    b.t[0] = acc[i][0]; b.t[1] = acc[i][1];
    pk_epi_gelu_block<FOLD, FRAGW>(p, b, patch, bias_l, lane, i * 32, sc, sd, tab, rst[i]);
  }
}

// second operand of the epilogue (EPK 2: residual, EPK 3: saved derivative): 32 rows x 64 columns of bf16 per block, LDS-DMA, the
// layout of gemm_tile.hpp's epi_fast_ext (position (row r, chunk pc) holds chunk pc ^ ((r >> 1) & 7) of row r)
struct PkOperand {
  __amdgpu_buffer_rsrc_t r; int ld;
  __device__ __forceinline__ void init(const bf16_t* ptr, int ld_) { r = __builtin_amdgcn_make_buffer_rsrc((void*)ptr, 0, 0xFFFFFFF0u, 0x00020000); ld = ld_; }
  // FRAG: the operand is stored fragment-major (file comment): block i of the wave's (strip, column group) is 4 KB, instruction itr takes 1 KB of it
  int nq;                                                      // N / 64 (FRAG)
  template <bool FRAG = false>
  __device__ __forceinline__ void dma_block(char* buf, int lane, int row0, int col0, int i, int mrem) const {
    if constexpr (FRAG) {
      uint32_t off = (uint32_t)((((size_t)(row0 >> 7) * (size_t)nq + (size_t)(col0 >> 6)) * 4 + (size_t)i) * 4096) + (uint32_t)lane * 16u;
      if (mrem <= 0) off = 0xFFFFFFF0u;                        // a strip wholly past M: zeros
#pragma unroll
      for (int itr = 0; itr < 4; ++itr)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, AVT_LDS_PTR(buf + itr * 1024), 16, off == 0xFFFFFFF0u ? off : off + (uint32_t)(itr * 1024), 0, 0, 0);
      return;
    }
    const int rl = lane >> 3, pc = lane & 7;
#pragma unroll
    for (int itr = 0; itr < 4; ++itr) {
      const int r_ = itr * 8 + rl;
      const int m = row0 + i * 32 + r_;
      const int n = col0 + ((pc ^ ((r_ >> 1) & 7)) * 8);
      uint32_t off = (uint32_t)(((size_t)m * (size_t)ld + (size_t)n) * 2);
      if (i * 32 + r_ >= mrem) off = 0xFFFFFFF0u;            // a row past M: out of the descriptor's range (zeros)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, AVT_LDS_PTR(buf + itr * 1024), 16, off, 0, 0, 0);
    }
  }
};

// EPK 2: C = bf16(acc + bias + res);  EPK 3: C = bf16((acc + 0) * aux), column sums of the rounded outputs.
// EPK 4 = EPK 2 + the rows' partial (sum, sum of squares) per 32-column slot -> stat_part (the next LayerNorm's statistics, from the fp32 values
//         before their bf16 rounding): one more 16-byte store per block and wave -- through a descriptor that ends behind the strip's last row,
//         no predicate, so every wave issues the same number of operations (the counted waits).
// EPK 7 = EPK 3 with the output rows multiplied by rstd[m] (dY' = rstd o dY of the folded LayerNorm backward) and the column sums weighted by
//         1 / rstd[m], i.e. taken over the UNscaled values: ln_stat[m] = {rstd, 1 / rstd}.
// Block 0's operand is already on its way into `buf0` (requested during the last K iteration); blocks 1 and 3 use `buf1`.
template <int EPK>
__device__ __forceinline__ int pk_epi_ext(const GemmParams& p, const f32x16_t (&acc)[4][2], char* patch, char* buf0, char* buf1,
                                          const PkOperand& op, float bias_v, int lane, int row0, int col0, int* tkt, int mrem) {
  constexpr bool RES = (EPK == 2 || EPK == 4), STATS = (EPK == 4), SCALE = epk_scale(EPK), FRAGR = epk_fragr(EPK);
  constexpr int NST = STATS ? 5 : 4;       // stores per block
  int tk = 0x7fffffff;
  const int ml = lane & 31, h = lane >> 5;
  const int rl = lane >> 3, pc = lane & 7;
  // (scale: the lane's row statistics of the four blocks, requested BEFORE DMA(1): older than everything the counted waits below count)
  f32x2_t rst[SCALE ? 4 : 1];
  if constexpr (SCALE) {
    // (unconditional loads of a clamped row: a branch around them would make the compiler wait for them on the spot)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int mrow = row0 + i * 32 + ml;
      rst[i] = *(const f32x2_t*)(p.ln_stat + 2 * (size_t)(mrow < p.M ? mrow : p.M - 1));
    }
  }
  op.template dma_block<FRAGR>(buf1, lane, row0, col0, 1, mrem);
  f32x4_t bb[2][4];
  if (RES) {          // the bias strip goes through the (still unused) patch once: lane l holds bias[col0 + l]
    ((float*)patch)[lane] = bias_v;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) bb[j][q] = *(const f32x4_t*)(patch + (j * 32 + 8 * q + 4 * h) * 4);
  } else {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) bb[j][q] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  PkStore sc;
  sc.init((bf16_t*)p.C + (size_t)row0 * p.ldc + col0, p.ldc, lane, mrem);
  // row statistics: the wave's two 32-column slots are slot pair col0 / 64 of [N / 64][M][2 slots][2] fp32 -- 16 bytes per row, one store per
  // block; descriptor over this strip's rows
  __amdgpu_buffer_rsrc_t rstat = sc.r;
  if constexpr (STATS) {
    const uint32_t nrec = (uint32_t)(mrem < 0 ? 0 : (mrem > 128 ? 128 : mrem)) * 16u;
    rstat = pk_uniform_rsrc(p.stat_part + ((size_t)(col0 >> 6) * (size_t)p.M + (size_t)row0) * 4, nrec);
  }
  float cs[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) cs[k] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // VMEM operations younger than DMA(i) (NST stores per block; as in epi_fast_ext):
    //   i = 0: DMA(1) = 4;  i = 1: [ticket], DMA(2), stores(0) = 4 + NST (+ 1);  i = 2: stores(0), DMA(3), stores(1) = 4 + 2 NST;  i = 3: stores(1), stores(2) = 2 NST
    // The wave that draws a ticket does so right after the wait of block 0 (the ticket has two blocks' time to return before the wait of
    // block 2 needs it retired)
    switch (i) {
      case 0:
        wait_vmcnt<4>();
        // (the statistics are older than DMA(1): they are here.  Consumed "by" this empty statement, before the ticket's branch hides the count of
        // younger operations from the compiler -- it would wait for ALL outstanding operations at their first real use otherwise)
        if constexpr (SCALE) asm volatile("" : "+v"(rst[0]), "+v"(rst[SCALE ? 1 : 0]), "+v"(rst[SCALE ? 2 : 0]), "+v"(rst[SCALE ? 3 : 0]));
        tk = pk_ticket(tkt, lane);
        break;
      case 1: if (tkt) wait_vmcnt<4 + NST + 1>(); else wait_vmcnt<4 + NST>(); break;
      case 2: wait_vmcnt<4 + 2 * NST>(); break;
      default: wait_vmcnt<2 * NST>(); break;
    }
    const char* buf = (i & 1) ? buf1 : buf0;
    u32x2_t opv[2][4];
    if constexpr (FRAGR) {
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const u32x4_t v = *(const u32x4_t*)(buf + st * 1024 + lane * 16);
        opv[st >> 1][2 * (st & 1)] = (u32x2_t){v[0], v[1]}; opv[st >> 1][2 * (st & 1) + 1] = (u32x2_t){v[2], v[3]};
      }
    } else {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        opv[j][q] = *(const u32x2_t*)(buf + ml * 128 + (((j * 4 + q) ^ ((ml >> 1) & 7)) * 16) + h * 8);
    }
    if (i + 2 < 4) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the buffer's previous contents are in registers
      op.template dma_block<FRAGR>((i & 1) ? buf1 : buf0, lane, row0, col0, i + 2, mrem);
    }
    const f32x2_t rr = SCALE ? (f32x2_t){rst[SCALE ? i : 0][0], rst[SCALE ? i : 0][0]} : (f32x2_t){1.f, 1.f};
    u32x4_t stw = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x2_t s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x2_t v0 = (f32x2_t){acc[i][j][4 * q], acc[i][j][4 * q + 1]} + (f32x2_t){bb[j][q][0], bb[j][q][1]};
        f32x2_t v1 = (f32x2_t){acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]} + (f32x2_t){bb[j][q][2], bb[j][q][3]};
        const f32x2_t o0 = (f32x2_t){bflo(opv[j][q][0]), bfhi(opv[j][q][0])}, o1 = (f32x2_t){bflo(opv[j][q][1]), bfhi(opv[j][q][1])};
        if (!RES) { v0 *= o0; v1 *= o1; if (SCALE) { v0 *= rr; v1 *= rr; } } else { v0 += o0; v1 += o1; }
        if (STATS) { s1 += v0; s1 += v1; s2 += v0 * v0; s2 += v1 * v1; }
        *(u32x2_t*)(patch + patch_wr(ml, h, j, q)) = (u32x2_t){pack2bf(v0[0], v0[1]), pack2bf(v1[0], v1[1])};
      }
      if constexpr (STATS) {
        // the two half-waves hold the two halves of the slot's columns of one row; both store the (same) total
        float a1 = s1[0] + s1[1], a2 = s2[0] + s2[1];
        a1 += __shfl_xor(a1, 32, 64); a2 += __shfl_xor(a2, 32, 64);
        stw[2 * j] = __builtin_bit_cast(uint32_t, a1); stw[2 * j + 1] = __builtin_bit_cast(uint32_t, a2);
      }
    }
    if constexpr (STATS) __builtin_amdgcn_raw_buffer_store_b128(stw, rstat, (uint32_t)(i * 32 + ml) * 16u, 0, 0);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + rl;
      const u32x2_t lo = *(const u32x2_t*)(patch + patch_rd(row, pc, 0)), hi = *(const u32x2_t*)(patch + patch_rd(row, pc, 1));
      sc.st(i * 32 + it * 8, (u32x4_t){lo[0], lo[1], hi[0], hi[1]});
      if (!RES && p.colsum) {                                // (a row past M adds nothing: selected, not branched around)
        float live = (i * 32 + row < mrem) ? 1.f : 0.f;
        if (SCALE) live *= __shfl(rst[SCALE ? i : 0][1], row, 64);      // 1 / rstd of the strip's row (held by lane `row` in the accumulator layout)
        cs[0] += live * bflo(lo[0]); cs[1] += live * bfhi(lo[0]); cs[2] += live * bflo(lo[1]); cs[3] += live * bfhi(lo[1]);
        cs[4] += live * bflo(hi[0]); cs[5] += live * bfhi(hi[0]); cs[6] += live * bflo(hi[1]); cs[7] += live * bfhi(hi[1]);
      }
    }
  }
  if (!RES && p.colsum) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) cs[q] += __shfl_xor(cs[q], o, 64);
    }
    if (lane < 8) {
      if (p.colsum_part) {
        float* dst = p.colsum_part + (size_t)(row0 / 128) * p.N + col0 + pc * 8;
        *(f32x4_t*)dst = (f32x4_t){cs[0], cs[1], cs[2], cs[3]};
        *(f32x4_t*)(dst + 4) = (f32x4_t){cs[4], cs[5], cs[6], cs[7]};
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) unsafeAtomicAdd(&p.colsum[col0 + pc * 8 + q], cs[q]);
      }
    }
  }
  return tk;
}

// reciprocals of the tile walk's divisors: q = n / d = (n * mg) >> 32 with mg = floor(2^32 / d) + 1, exact while n * d < 2^32 (n, d < 2^16 here)
struct PkWalk { uint32_t mg_tn, mg_per, mg_w, mg_wl, per, nfull, wl; };
__device__ __forceinline__ uint32_t pk_div(uint32_t n, uint32_t d, uint32_t mg) { return d == 1 ? n : __umulhi(n, mg); }

// EPK: 0 = bias | 1 = bias, GELU (+ GELU') by table | 2 = bias, + residual | 3 = * saved derivative (+ column sums)
//      4 = 2 + row statistics out | 5 = 0 with the LayerNorm fold | 6 = 1 with the LayerNorm fold | 7 = 3 with the rows scaled by rstd (see pk_epi_ext, pk_epi_plain)
template <int EPK>
__global__ __launch_bounds__(512) void gemm_8pp_kernel(GemmParams p, PkWalk wk_arg, int* __restrict__ sched) {
  constexpr int BK = 64, HALF = PK_HALF;
  constexpr bool GELU = epk_gelu(EPK), FOLD = epk_fold(EPK), AUX = epk_aux(EPK);
  constexpr int TABB = GELU ? GELU_TAB_BYTES : 0;
  constexpr bool HAS_OP = (EPK == 2 || EPK == 4 || AUX);
  extern __shared__ __attribute__((aligned(16))) char smem8[];
  char* const lds = smem8 + TABB;                       // the ring: 8 half-tile slots (kind x K-tile parity), as in gemm_8p_kernel
  char* const ext = lds + 8 * HALF;                     // behind the ring: 32 KB (8 KB next to the table), idle during the K loop
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wn = wave & 3;
  if constexpr (GELU) {       // the table: once per workgroup
    __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)g_gelu_tab, 0, GELU_TAB_BYTES, 0x00020000);
#pragma unroll
    for (int i = 0; i < GELU_TAB_BYTES / (8 * 1024); ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, AVT_LDS_PTR(smem8 + (wave * (GELU_TAB_BYTES / 8192) + i) * 1024), 16,
                                               (uint32_t)((wave * (GELU_TAB_BYTES / 8192) + i) * 1024 + lane * 16), 0, 0, 0);
  }
  // the wave's epilogue space: 8 KB in a parity-1 slot (two waves per slot), and its share of the space behind the ring
  char* const P1 = lds + (2 * wn + 1) * HALF + grp * 8192;
  char* const P2 = ext + wave * (GELU ? 1024 : 4096);

  const int ntile = p.tiles_m * p.tiles_n;
  const int nk = p.K / BK;                               // even, >= 4 (host-checked)
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.b_bytes, 0x00020000);

  // per-lane source offsets of the wave's DMA instructions: LDS row rr = wave * 8 + lane / 8 (+ 64 j) of a half-tile takes the 16-byte chunk
  // (lane & 7) ^ swizzle(rr) of a source row; the row's place in the tile (half h, instruction j) and the tile / K-tile origin are scalar
  // terms added per instruction, so two registers hold the per-lane state for the whole kernel.  The row terms stay in the VECTOR
  // offset (base + scalar sum, one v_add per instruction): the descriptor's bounds check ignores the scalar offset, and rows past M in
  // the last row tile have to come back as zeros, not as reads past the end of the operand.
  // (every epilogue but the plain one needs all registers: there the two are re-derived at the start of each tile instead of kept)
  uint32_t base_a, base_b;
  auto lane_bases = [&](int l) __attribute__((always_inline)) {
    const int rr = wave * 8 + (l >> 3);
    const int c = (l & 7) ^ kmajor_swz<BK>(rr);
    base_a = (uint32_t)(((size_t)rr * (size_t)p.lda + (size_t)(c * 8)) * 2);
    base_b = (uint32_t)(((size_t)((rr >> 5) * 64 + (rr & 31)) * (size_t)p.ldb + (size_t)(c * 8)) * 2);
  };
  lane_bases(lane);
  const uint32_t lda2 = (uint32_t)p.lda * 2u, ldb2 = (uint32_t)p.ldb * 2u;
  int lane_k = lane;                                       // the fragment addresses hang on this copy, re-derived per tile (pk_lane_id)
  char* const dstw = lds + wave * 8 * (BK * 2);
  constexpr int JSTEP = 8192;

  // tile walk (32-bit byte offsets: every operand is below 4 GiB)
  struct Tile { int tm0, tn0; uint32_t a, b; };
  // position in the XCD-ordered walk (gemm.hip: xcd_remap) -> tile; the divisions by tiles_n / tiles per strip / strip width go through
  // host-computed reciprocals (scalar multiply-high): the walk is a dozen scalar instructions between two tiles
  auto tile_at = [&](int t_) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)     // (re-read from the kernel-argument segment, as the epilogue's arguments are: not kept across the K loop)
    const __attribute__((address_space(4))) char* kp = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    const PkWalk wk = *(const __attribute__((address_space(4))) PkWalk*)(kp + ((sizeof(GemmParams) + 3) & ~(size_t)3));
#else
    const PkWalk wk = wk_arg;
#endif
    int tm_i, tn_i;
    if (p.strip_w > 0) {
      const int strip = (int)pk_div((uint32_t)t_, wk.per, wk.mg_per), r_ = t_ - strip * (int)wk.per;
      const bool lastw = strip >= (int)wk.nfull;
      const int w_ = lastw ? (int)wk.wl : p.strip_w;
      tm_i = (int)pk_div((uint32_t)r_, (uint32_t)w_, lastw ? wk.mg_wl : wk.mg_w); tn_i = strip * p.strip_w + (r_ - tm_i * w_);
    } else { tm_i = (int)pk_div((uint32_t)t_, (uint32_t)p.tiles_n, wk.mg_tn); tn_i = t_ - tm_i * p.tiles_n; }
    Tile t;
    t.tm0 = tm_i * 256; t.tn0 = tn_i * 256;
    t.a = (uint32_t)t.tm0 * (uint32_t)p.lda * 2u; t.b = (uint32_t)t.tn0 * (uint32_t)p.ldb * 2u;
    return t;
  };
  // this XCD's range of the walk and its ticket counter (one 128-byte line per XCD; sched[8 * 32] counts the workgroups that have left)
  uint32_t xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7u;
  const int xq = ntile >> 3, xr = ntile & 7;
  // range r of the walk (the tiles XCD r would own): first position, number of tiles, ticket counter (one 128-byte line each;
  // sched[8 * 32] counts the workgroups that have left)
  auto rng_start = [&](int r) __attribute__((always_inline)) { return (r < xr) ? r * (xq + 1) : xr * (xq + 1) + (r - xr) * xq; };
  auto rng_cnt = [&](int r) __attribute__((always_inline)) { return xq + ((r < xr) ? 1 : 0); };
  // `vic` = how many ranges this workgroup has moved past (0 = it still draws from its own XCD's range); uniform, changes only at barriers
  int vic = 0;
  // thread 0 only: the next position of the walk, drawn synchronously, moving on to the next range when one is exhausted
  auto draw_sync = [&](int& v) __attribute__((always_inline)) {
#pragma nounroll
    for (; v < 8; ++v) {
      const int r = (int)((xcc + (uint32_t)v) & 7u);
      const int t_ = pk_draw_sync(sched + r * 32);
      if (t_ < rng_cnt(r)) return rng_start(r) + t_;
    }
    return 0x7fffffff;
  };
  int* const mailbox = (int*)(ext + 512);                 // free at both times it is used (start of the kernel, end of an epilogue)
  // the first two positions
  if (tid == 0) {
    const int own = (int)xcc, oc = rng_cnt(own);
    const int t0_ = __hip_atomic_fetch_add(sched + own * 32, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int v = 0, g0, g1;
    if (t0_ + 1 < oc) { g0 = rng_start(own) + t0_; g1 = g0 + 1; }
    else if (t0_ < oc) { g0 = rng_start(own) + t0_; v = 1; g1 = draw_sync(v); }
    else { v = 1; g0 = draw_sync(v); g1 = (g0 == 0x7fffffff) ? g0 : draw_sync(v); }
    mailbox[0] = g0; mailbox[1] = g1; mailbox[2] = v;
  }
  __syncthreads();
  const int gp0 = __builtin_amdgcn_readfirstlane(mailbox[0]), gp1 = __builtin_amdgcn_readfirstlane(mailbox[1]);
  vic = __builtin_amdgcn_readfirstlane(mailbox[2]);
  __syncthreads();
  bool has_cur = gp0 != 0x7fffffff, has_next = gp1 != 0x7fffffff;
  Tile cur = tile_at(has_cur ? gp0 : 0), nxt = cur;
  if (has_next) nxt = tile_at(gp1);
  int tkn = 0;
  bool walk_pending = false;

  // slot index = kind * 2 + (K tile & 1); kinds 0 = A0h, 1 = B0h, 2 = B1h, 3 = A1h.  kt counts the current tile's K tiles; kt == nk is
  // K tile 0 of the next output tile (without one, `nxt` = `cur`: the fetch is repeated and never read)
  auto stage_a = [&](int h, int kt) __attribute__((always_inline)) {
    const bool in = kt < nk;
    const uint32_t adv = in ? cur.a + (uint32_t)kt * (BK * 2) : nxt.a + (uint32_t)(kt - nk) * (BK * 2);
    char* d = dstw + ((h ? 3 : 0) * 2 + (kt & 1)) * HALF;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, AVT_LDS_PTR(d + j * JSTEP), 16, base_a + (adv + (uint32_t)(j * 128 + h * 64) * lda2), 0, 0, 0);
  };
  auto stage_b = [&](int h, int kt) __attribute__((always_inline)) {
    const bool in = kt < nk;
    const uint32_t adv = in ? cur.b + (uint32_t)kt * (BK * 2) : nxt.b + (uint32_t)(kt - nk) * (BK * 2);
    char* d = dstw + ((h ? 2 : 1) * 2 + (kt & 1)) * HALF;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, AVT_LDS_PTR(d + j * JSTEP), 16, base_b + (adv + (uint32_t)(j * 128 + h * 32) * ldb2), 0, 0, 0);
  };
  bf16x8_t fa[2][4], fb0[4], fb1[4], fb0n[4];
  auto read_a = [&](bf16x8_t (&f)[2][4], int h, int par) __attribute__((always_inline)) {
    const char* slot = lds + ((h ? 3 : 0) * 2 + par) * HALF;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) f[i][ks] = frag_kmajor<BK>(slot + grp * 64 * (BK * 2), i, ks, lane_k);
  };
  auto read_b = [&](bf16x8_t (&f)[4], int h, int par) __attribute__((always_inline)) {
    const char* slot = lds + ((h ? 2 : 1) * 2 + par) * HALF;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) f[ks] = frag_kmajor<BK>(slot + wn * 32 * (BK * 2), 0, ks, lane_k);
  };

  const bool two_outputs = p.C2 != nullptr;
  PkOperand op;
  if (HAS_OP) { op.init(AUX ? p.aux : p.res, AUX ? p.ldaux : p.ldres); op.nq = p.N >> 6; }

#define P8_BARRIER() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define P8_PIN() __builtin_amdgcn_sched_barrier(0)
#define P8_MFMA(FA, FB, I0, J)                                                                          \
  do {                                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    __builtin_amdgcn_s_setprio(1);                                                                       \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                     \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                      \
        acc[(I0) + i][J] = mma<0>(FA[i][ks], FB[ks], acc[(I0) + i][J]);                                  \
    __builtin_amdgcn_s_setprio(0);                                                                       \
  } while (0)
// the first touch of an accumulator quadrant in a tile (phases 0-3 of its first iteration) starts from the MFMA's zero operand
// instead of 128 v_mov_b32 per wave and tile
#define P8_MFMA0(FA, FB, I0, J)                                                                         \
  do {                                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    __builtin_amdgcn_s_setprio(1);                                                                       \
    if (t == 0) {                                                                                        \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) acc[(I0) + i][J] = mma<0>(FA[i][0], FB[0], zero16);  \
    } else {                                                                                             \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) acc[(I0) + i][J] = mma<0>(FA[i][0], FB[0], acc[(I0) + i][J]); \
    }                                                                                                    \
    _Pragma("unroll") for (int ks = 1; ks < 4; ++ks)                                                     \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                      \
        acc[(I0) + i][J] = mma<0>(FA[i][ks], FB[ks], acc[(I0) + i][J]);                                  \
    __builtin_amdgcn_s_setprio(0);                                                                       \
  } while (0)

  // first tile: its K tile 0 (the later tiles find theirs in the ring when they start)
  if (has_cur) {
  stage_a(0, 0); stage_b(0, 0); stage_b(1, 0); stage_a(1, 0);
  wait_vmcnt<0>();
  P8_BARRIER();

  for (;;) {
    // here: the four parity-0 slots hold K tile 0 of `cur`, every other slot is free,
    // nothing but stores is in flight
    stage_a(0, 1); stage_b(0, 1);
    if (grp == 1) P8_BARRIER();
    f32x16_t acc[4][2];                                     // (first written by the zero-operand MFMAs of iteration 0)
    const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float bias_v = 0.f, c_v = 0.f;
    f32x2_t rst[4] = {};
    const int m0_e = cur.tm0 + grp * 128, n0_e = cur.tn0 + wn * 64;
    const int mrem = p.M - m0_e;                           // valid rows of this wave's strip (>= 128 everywhere but in the last row tile)
    read_b(fb0, 0, 0);
#pragma nounroll
    for (int t = 0; t < nk; t += 2) {
      const bool last = t + 2 >= nk;
      // ---- even K tile t (slot parity 0).  t == 0: B1h / A1h of K tile 0 landed before the epilogue's barrier -- no counted wait
      //      (a vmcnt(8) there would wait for the previous tile's stores) ----
      read_a(fa, 0, 0); P8_PIN(); stage_b(1, t + 1); if (t) wait_vmcnt<8>(); P8_BARRIER();
      P8_MFMA0(fa, fb0, 0, 0); P8_BARRIER();
      read_b(fb1, 1, 0); P8_PIN(); stage_a(1, t + 1); if (t) wait_vmcnt<8>(); P8_BARRIER();
      P8_MFMA0(fa, fb1, 0, 1); P8_BARRIER();
      read_a(fa, 1, 0); P8_PIN(); stage_a(0, t + 2); wait_vmcnt<6>(); P8_BARRIER();
      P8_MFMA0(fa, fb1, 2, 1); P8_BARRIER();
      read_b(fb0n, 0, 1); P8_PIN(); stage_b(0, t + 2); wait_vmcnt<8>(); P8_BARRIER();
      P8_MFMA0(fa, fb0, 2, 0); P8_BARRIER();
      // ---- odd K tile t+1 (slot parity 1).  In the last iteration the stages of "K tile nk" fetch the next output tile's K tile 0;
      //      those of K tile nk + 1 are left out (their slots become the epilogue's) ----
      if (walk_pending) {                                   // (first iteration of every tile but the workgroup's first)
        walk_pending = false;
        has_next = tkn != 0x7fffffff;
        if (has_next) nxt = tile_at(tkn);
      }
      if (last && !AUX && p.bias) bias_v = p.bias[n0_e + pk_lane_id()];   // used two phases later at the earliest
      if constexpr (FOLD) {
        if (last) {                                         // the c strip and the lane's row statistics of the four blocks: requested here, ahead of the next
          const int le = pk_lane_id();                      // tile's operand stages, so that waiting for them in the epilogue does not wait for those
          c_v = p.ln_c[n0_e + le];
#pragma unroll
          for (int i = 0; i < 4; ++i) {                     // (a clamped row instead of a branch: rows past M are never stored)
            const int mrow = m0_e + i * 32 + (le & 31);
            rst[i] = *(const f32x2_t*)(p.ln_stat + 2 * (size_t)(mrow < p.M ? mrow : p.M - 1));
          }
        }
      }
      read_a(fa, 0, 1); P8_PIN(); stage_b(1, t + 2); wait_vmcnt<8>(); P8_BARRIER();
      P8_MFMA(fa, fb0n, 0, 0); P8_BARRIER();
      read_b(fb1, 1, 1); P8_PIN(); stage_a(1, t + 2); wait_vmcnt<8>(); P8_BARRIER();
      P8_MFMA(fa, fb1, 0, 1); P8_BARRIER();
      read_a(fa, 1, 1); P8_PIN();
      if (!last) { stage_a(0, t + 3); wait_vmcnt<6>(); }
      else {
        if (HAS_OP) op.template dma_block<epk_fragr(EPK)>(P2, pk_lane_id(), m0_e, n0_e, 0, mrem);          // the epilogue's second operand, block 0 -> behind the ring
      }
      P8_BARRIER();
      P8_MFMA(fa, fb1, 2, 1); P8_BARRIER();
      if (!last) { read_b(fb0, 0, 0); P8_PIN(); stage_b(0, t + 3); wait_vmcnt<8>(); }
      P8_BARRIER();
      P8_MFMA(fa, fb0n, 2, 0); P8_BARRIER();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (grp == 0) P8_BARRIER();
    P8_BARRIER();                                          // every fragment read retired: the parity-1 slots are free
    int tk;
    {
      // the epilogue's pointers and strides are re-read from the kernel-argument segment for every tile (scalar loads through a
      // laundered pointer) instead of occupying two dozen scalar registers across the K loop
#if defined(__HIP_DEVICE_COMPILE__)
      const __attribute__((address_space(4))) GemmParams* pp = (const __attribute__((address_space(4))) GemmParams*)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(pp));
      const GemmParams pe = *pp;
#else
      const GemmParams pe = p;
#endif
      const int lane_e = pk_lane_id();
      int* const tkt = (wave == 0 && has_next && vic < 8) ? sched + ((xcc + (uint32_t)vic) & 7u) * 32 : nullptr;   // wave 0 draws the ticket for the tile after next
      // (fold: the strip and the statistics are consumed "by" this empty statement, next to the bias, while the compiler still knows how many
      // younger operations are in flight: at their first real use, behind the ticket's branch, it would wait for ALL of them)
      if constexpr (FOLD) asm volatile("" : "+v"(c_v), "+v"(rst[0]), "+v"(rst[1]), "+v"(rst[2]), "+v"(rst[3]), "+v"(bias_v));
      if constexpr (EPK == 0 || EPK == 5) {
        tk = pk_ticket(tkt, lane_e);
        ((float*)P2)[lane_e] = bias_v;
        if (FOLD) ((float*)P2)[64 + lane_e] = c_v;
        pk_epi_plain<FOLD>(pe, acc, P1, (const float*)P2, lane_e, m0_e, n0_e, mrem, rst);
      } else if constexpr (GELU) {
        tk = pk_ticket(tkt, lane_e);
        ((float*)P2)[lane_e] = bias_v;
        if (FOLD) ((float*)P2)[64 + lane_e] = c_v;
        pk_epi_gelu<FOLD, epk_fragw(EPK)>(pe, acc, P1, (const float*)P2, lane_e, m0_e, n0_e, smem8, mrem, rst);
      } else {
        tk = pk_epi_ext<EPK>(pe, acc, P1, P2, P1 + 4096, op, bias_v, lane_e, m0_e, n0_e, tkt, mrem);
      }
    }
    if (GELU && two_outputs) wait_vmcnt<32>();        // (GELU + GELU': 32 stores per wave and tile)
    else if (EPK == 4) wait_vmcnt<20>();                   // (16 output stores + 4 of the row statistics)
    else wait_vmcnt<16>();                                 // everything older than the tile's last 16 stores: the next tile's K tile 0, the ticket
    asm volatile("" :: "v"(tk));                           // (every wave "uses" the ticket here: the compiler's own bookkeeping of the atomic ends at this
                                                           //  point on every path, not only inside wave 0's branch below)
    if (wave == 0 && pk_lane_id() == 0) {
      // ticket -> position of the walk; a ticket past the end of the range it was drawn from means that range is exhausted: go on with the
      // next ranges, synchronously (this happens once per range and workgroup, at the launch's tail)
      int gpos = 0x7fffffff, v = vic;
      if (has_next && v < 8) {
        const int r = (int)((xcc + (uint32_t)v) & 7u);
        if (tk < rng_cnt(r)) gpos = rng_start(r) + tk;
        else { ++v; gpos = draw_sync(v); }
      }
      mailbox[0] = gpos; mailbox[1] = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    P8_BARRIER();                                          // ... for every wave's share of it; and every patch has been read back
    if (!has_next) break;
    vic = __builtin_amdgcn_readfirstlane(mailbox[1]);
    tkn = __builtin_amdgcn_readfirstlane(mailbox[0]);      // the position of the tile after the one that starts now: turned into a tile under
    lane_k = pk_lane_id();                                 // the first iteration of the K loop (needed in its last one)
    lane_bases(lane_k);                                    // (re-derived per tile in every variant: kept across the epilogue they cost the plain one two spilled registers)
    cur = nxt;
    walk_pending = true;
  }
  }
  // leave: the last workgroup of the launch zeroes the counters (the next launch on this stream finds them clean).  Relaxed, device-scope
  // atomics are enough: a workgroup has consumed the results of all its ticket draws before it gets here, so when the leave count says
  // "everybody else has left" no draw is outstanding (acquire / release here would write back and invalidate the L2 once per workgroup)
  if (tid == 0) {
    const int gone = __hip_atomic_fetch_add(sched + 8 * 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gone == (int)gridDim.x - 1) {
#pragma unroll
      for (int x = 0; x <= 8; ++x) __hip_atomic_store(sched + x * 32, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#undef P8_PIN
#undef P8_MFMA
#undef P8_MFMA0
#undef P8_BARRIER
}

// ticket counters: one block (8 counters + the leave count, a 128-byte line each) per (device, stream), zeroed once; every launch leaves
// its block zeroed.  Launches on one stream are ordered, so they can share a block whatever the depth of the launch queue; launches on
// different streams (the forward thread and the autograd thread call in here concurrently) never share one.  The table is guarded by a
// mutex and grows in chunks of 64 blocks (round-5 advisor: it used to end at 64 streams per device, and a 65th stream's fragment-major
// GEMMs -- which only this kernel can run -- failed); a stream gets no block only when the allocation itself fails, and its GEMMs then take
// the one-tile-per-workgroup kernel.  (A new chunk allocates and clears device memory: warm the library up before capturing a stream into a graph.)
constexpr int PK_SCHED_INTS = 9 * 32, PK_SCHED_BLOCKS = 64;
int* sched_block(hipStream_t s) {
  struct Dev { std::vector<int*> chunks; std::unordered_map<hipStream_t, int*> of; size_t used = 0; };
  static std::mutex mu;
  static Dev devs[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  Dev& d = devs[dev];
  auto it = d.of.find(s);
  if (it != d.of.end()) return it->second;
  if (d.used == d.chunks.size() * PK_SCHED_BLOCKS) {
    int* ptr = nullptr;
    if (hipMalloc(&ptr, (size_t)PK_SCHED_BLOCKS * PK_SCHED_INTS * 4) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipMemset(ptr, 0, (size_t)PK_SCHED_BLOCKS * PK_SCHED_INTS * 4) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(ptr); return nullptr; }
    d.chunks.push_back(ptr);
  }
  int* blk = d.chunks.back() + (d.used % PK_SCHED_BLOCKS) * PK_SCHED_INTS;
  ++d.used;
  d.of.emplace(s, blk);
  return blk;
}

template <int EPK>
int launch_8pp(const GemmParams& p, int grid, hipStream_t s) {
  constexpr int smem = 8 * PK_HALF + (epk_gelu(EPK) ? GELU_TAB_BYTES + 8192 : 32768);
  static_assert(smem <= 160 * 1024, "persistent 8-phase kernel: LDS");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_8pp_kernel<EPK>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  int* sched = sched_block(s);
  if (!sched) return 0;                                     // no counter block: the caller's non-persistent kernel does the job
  auto magic = [](uint32_t d) { return d <= 1 ? 0u : (uint32_t)((1ull << 32) / d + 1); };
  PkWalk wk{};
  wk.mg_tn = magic((uint32_t)p.tiles_n);
  if (p.strip_w > 0) {
    wk.per = (uint32_t)p.tiles_m * (uint32_t)p.strip_w;
    wk.nfull = (uint32_t)(p.tiles_n / p.strip_w);           // strips of full width; a narrower last one when tiles_n % strip_w != 0
    wk.wl = (uint32_t)(p.tiles_n % p.strip_w);
    if (wk.wl == 0) wk.wl = (uint32_t)p.strip_w;
    wk.mg_per = magic(wk.per); wk.mg_w = magic((uint32_t)p.strip_w); wk.mg_wl = magic(wk.wl);
  }
  hipLaunchKernelGGL((gemm_8pp_kernel<EPK>), dim3(grid), dim3(512), smem, s, p, wk, sched);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { avt_set_error("avt_gemm: launch failed: %s", hipGetErrorString(e)); return (int)e; }
  return 1;
}

}  // namespace

int avt_gemm_persist(GemmParams& p, int kinds, bool force, hipStream_t s) {
  // covered: N a multiple of 256 (any M), an even number (>= 4) of 64-wide K tiles, bf16 output through 16-byte stores, one of the four epilogues
  if (p.splitk != 1 || p.out_f32 || !p.wide_ok || p.drop_thresh || p.res_period) return 0;
  if (p.N % 256 || p.K % 128 || p.K < 256) return 0;
  int kmax = PK_KMAX;
  if (p.K > kmax && !force) return 0;
  const int ntile = p.tiles_m * p.tiles_n;
  if (ntile < 512 || ntile >= 65536) return 0;              // fewer than two tiles per CU: nothing to overlap (measured at 288 tiles: 122-133 us against 64-104 us for the one-tile kernel's GELU epilogues -- table load and set-up per workgroup for one tile each); (the walk's reciprocals: n, d < 2^16)
  if ((uint64_t)p.a_bytes + 256ull * p.lda * 2 >= (1ull << 32) || (uint64_t)p.b_bytes + 256ull * p.ldb * 2 >= (1ull << 32)) return 0;
  const int grid = 256;                                     // one workgroup per CU (a multiple of the 8 XCDs)
  const bool fold = p.ln_c != nullptr, scale = !fold && p.ln_stat != nullptr, stats = p.stat_part != nullptr;
  if (fold) {                                                // LayerNorm fold: bias | GELU epilogues only
    if (scale || stats || p.res || p.colsum) return 0;
    if (p.aux_frag) return 0;
    if ((kinds & 1) && p.act == 0 && !p.C2) return launch_8pp<5>(p, grid, s);
    if ((kinds & 2) && p.act == 1) return p.c2_frag ? launch_8pp<9>(p, grid, s) : launch_8pp<6>(p, grid, s);
    return 0;
  }
  if (scale) {                                               // rows scaled by rstd: saved-derivative epilogue only
    if ((kinds & 8) && !stats && p.act == 3 && p.aux && !p.res && !p.bias && !p.C2) return p.aux_frag ? launch_8pp<11>(p, grid, s) : launch_8pp<7>(p, grid, s);
    return 0;
  }
  if (stats) {                                               // row statistics out: bias + residual epilogue only
    if (p.c2_frag || p.aux_frag) return 0;
    if ((kinds & 4) && p.act == 0 && p.res && !p.colsum && !p.C2 && p.N % 64 == 0) return launch_8pp<4>(p, grid, s);
    return 0;
  }
  if ((kinds & 1) && p.act == 0 && !p.res && !p.colsum && !p.C2 && !p.aux_frag) return launch_8pp<0>(p, grid, s);
  if ((kinds & 2) && p.act == 1 && !p.res && !p.colsum && !p.aux_frag) return p.c2_frag ? launch_8pp<8>(p, grid, s) : launch_8pp<1>(p, grid, s);
  if (p.c2_frag) return 0;
  if ((kinds & 4) && p.act == 0 && p.res && !p.colsum && !p.C2) return launch_8pp<2>(p, grid, s);
  if ((kinds & 8) && p.act == 3 && p.aux && !p.res && !p.bias && !p.C2) return p.aux_frag ? launch_8pp<10>(p, grid, s) : launch_8pp<3>(p, grid, s);
  return 0;
}

// The fragment-major private layout (file comment): bytes of such a tensor, and whether a call of this shape is taken by the persistent kernel
// (the shape part of avt_gemm_persist's conditions and of the automatic tile choice in gemm.hip, for contiguous operands: lda = ldb = K).
extern "C" size_t avt_gemm_frag_bytes(int M, int N) {
  if (M <= 0 || N <= 0) return 0;
  return (size_t)((M + 127) / 128) * 128 * (size_t)((N + 63) / 64) * 64 * 2;
}
extern "C" int avt_gemm_frag_ok(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if (N % 256 || K % 128 || K < 256 || K > PK_KMAX) return 0;
  const long tm = (M + 255) / 256, tn = N / 256, ntile = tm * tn;
  if (ntile < 512 || ntile >= 65536) return 0;               // (>= 200 tiles: the automatic tile choice is the 8-phase schedule)
  if (((uint64_t)M + 256) * (uint64_t)K * 2 >= (1ull << 32) || ((uint64_t)N + 256) * (uint64_t)K * 2 >= (1ull << 32)) return 0;
  if (avt_gemm_frag_bytes(M, N) >= 0xFFFFFFF0ull || (uint64_t)M * (uint64_t)N * 2 >= 0xFFFFFFF0ull) return 0;
  return 1;
}
