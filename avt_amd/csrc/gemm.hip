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
#include <cstdlib>
#include <type_traits>
#include "common.hpp"
#include "../../include/avt_hip.h"

namespace {

struct GemmParams {
  const bf16_t* A; const bf16_t* B; void* C; bf16_t* C2;
  const float* bias; const bf16_t* res; const bf16_t* aux; float* colsum;
  float* colsum_part;                      // [row slots][N] partial column sums (one row per wave row of the grid) instead of atomics
  int M, N, K;
  int lda, ldb, ldc, ldc2, ldres, ldaux;
  int res_period;
  int act;          // 0 none | 1 gelu_erf | 2 gelu_tanh (C2 = derivative) | 3 *= aux
  int out_f32;
  int splitk;
  int tiles_m, tiles_n;
  int strip_w;                   // 8-phase kernel, activation epilogue: walk the tiles in column strips of this many tiles (0 = row-major)
  size_t ws_bytes;
  float* ws;                     // EPI 2: split-K partial slabs, [splitk * tiles][BM * BN] fp32 in accumulator order
  uint32_t a_bytes, b_bytes;     // buffer-descriptor bounds
  uint32_t drop_thresh; float drop_scale; uint64_t drop_seed;
#ifdef AVT_LAB
  int stagger;                   // lab only: cycles over which the first wave of workgroups spreads its start (0 = off)
#endif
  int wide_ok;                   // all epilogue leading dims are multiples of 8 -> 16-byte accesses allowed
#ifdef AVT_LAB
  long long* dbg;                // lab only: per-block phase timestamps (s_memtime)
#endif
};
// Instrumentation and experiment switches exist only in the lab build (make lab -> libavt_hip_lab.so, used by tools/):
// the product library reads no environment variable and takes no pointer from anywhere but its arguments.
#ifdef AVT_LAB
#define AVT_DBG(p) ((p).dbg)
#define AVT_STAGGER(p) ((p).stagger)
#else
#define AVT_DBG(p) ((long long*)nullptr)
#define AVT_STAGGER(p) 0
#endif

constexpr int BK64 = 64;
// cache policy of the 8-phase kernel's operand streams (aux of buffer_load ... lds: 0 = default, 2 = nt)
#ifndef AVT_LDA_AUX
#define AVT_LDA_AUX 0
#endif
#ifndef AVT_LDB_AUX
#define AVT_LDB_AUX 0
#endif
#ifndef AVT_LDP_AUX          // the epilogue's second operand (saved derivative / residual), read exactly once
#define AVT_LDP_AUX 0
#endif

// XCD-aware bijective remap of the linear block id: XCD x (= id % 8 by dispatch order) owns a contiguous
// range of logical blocks, ordered (split, tile row, tile column), so tiles sharing an A row-panel sit behind the same L2
// and -- for split-K weight gradients -- an XCD works on one or two K ranges only, instead of pulling every K range of
// the shared B operand through each of the eight L2s (fc1 wgrad: 2.65 GB fetched for 0.97 GB of operands before).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  int q = nblk >> 3, r = nblk & 7;
  int xcd = bid & 7, idx = bid >> 3;
  int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + idx;
}

// One MFMA step; the operand order decides whether a lane ends up holding a column or a row of the output block:
// EPI 0 (activation epilogue): B first, so block (i,j) comes out TRANSPOSED -- lane l holds output row i*32 + (l&31) and its
//   16 registers are four groups q of 4 CONSECUTIVE columns j*32 + 8q + 4(l>>5) + (r&3): elementwise work, bf16 packing and
//   the LDS hand-off all work on 8-/16-byte units instead of single floats.
// EPI 1 (atomic accumulate): A first, lane l holds column j*32 + (l&31), so a wave's atomics hit 128 consecutive bytes.
template <int EPI>
__device__ __forceinline__ f32x16_t mma(bf16x8_t a, bf16x8_t b, f32x16_t c) {
  return EPI == 0 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ---- GELU by table (fc1 forward: GELU + GELU' outputs) --------------------------------------------------------------------------
// The erf polynomial + exponential of gelu_erf_both4 is 36 packed / conversion instructions + 4 v_exp per 4 elements, and the
// epilogue of these launches is bound by the vector-ALU issue rate (profiles/r03x_pmc_sq.txt: 353 M non-MFMA vector instructions per
// launch = 0.26 of the SIMD-cycles next to 0.43 of MFMA).  The table form rounds the biased pre-activation to bf16 (one
// v_cvt_pk_bf16_f32 per pair -- the same rounding the stored activation would get one step later) and reads {GELU, GELU'} as a packed
// bf16 pair from a 24-KB LDS table indexed by the bf16 bits: 24 exponents (2^-16 <= |x| < 2^8) x 128 mantissas x sign, generated by
// tools/gen/gelu_table.py from the exact erf form in double precision.  11 vector instructions + 2 ds_read_b32 per PAIR of elements.
// Outside the table: |x| < 2^-16 takes the entry of 2^-16 (|error| <= 7.7e-6 on GELU, GELU' = 0.5 is exact to bf16); |x| >= 2^8 (or NaN)
// is handled exactly by a fix-up pass that a block only enters when one of its values is that large (tracked with one v_pk_max_u16 per pair).
constexpr int GELU_TAB_ELO = 111, GELU_TAB_NEXP = 24, GELU_TAB_NT = GELU_TAB_NEXP * 128;
constexpr int GELU_TAB_BYTES = 2 * GELU_TAB_NT * 4;                       // 24576
constexpr float GELU_TAB_TOP = 256.f;                                     // = 2^(GELU_TAB_ELO + GELU_TAB_NEXP - 127): first magnitude above the table
static_assert(GELU_TAB_ELO + GELU_TAB_NEXP - 127 == 8, "GELU_TAB_TOP");
__device__ const uint32_t g_gelu_tab[2 * GELU_TAB_NT] = {
#include "gelu_table.inc"
};
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2_t;
// the packed bf16 pair w = {x0, x1} -> the byte offsets of its two table entries, packed as 16-bit halves; mx tracks max |bits|
__device__ __forceinline__ uint32_t gelu_tab_offsets(uint32_t w, u16x2_t& mx) {
  constexpr unsigned short LO = GELU_TAB_ELO << 7, HI = ((GELU_TAB_ELO + GELU_TAB_NEXP) << 7) - 1;
  const u16x2_t m = __builtin_bit_cast(u16x2_t, w & 0x7fff7fffu);
  mx = __builtin_elementwise_max(mx, m);
  const u16x2_t mc = __builtin_elementwise_min(__builtin_elementwise_max(m, (u16x2_t){LO, LO}), (u16x2_t){HI, HI});
  // byte offset ((mc - LO) * 2 + sign) * 4 for both halves at once (modulo 2^16): shift, add, and the sign bit moved to bit 2
  const u16x2_t i8 = (mc << (u16x2_t){3, 3}) + (u16x2_t){(unsigned short)(0u - 8u * LO), (unsigned short)(0u - 8u * LO)};
  const u16x2_t sg = __builtin_bit_cast(u16x2_t, w) >> (u16x2_t){13, 13};
  return (__builtin_bit_cast(uint32_t, sg) & 0x00040004u) | __builtin_bit_cast(uint32_t, i8);
}
// exact values for a pre-activation the table does not cover from above (|x| >= 2^8, inf, NaN): GELU = x | -0 (|GELU(x)| < 1e-300 there),
// GELU' = 1 | 0
__device__ __forceinline__ void gelu_big(float x, bf16_t& h, bf16_t& d) {
  if (x != x) { h = 0x7fc0; d = 0x7fc0; return; }
  h = x > 0.f ? f2bf(x) : (bf16_t)0x8000; d = x > 0.f ? (bf16_t)0x3f80 : (bf16_t)0;
}
// The same look-up from the table in GLOBAL memory, element by element, for every other place an erf GELU is evaluated (small-tile
// kernels, epilogues with further terms): all erf-GELU epilogues agree bit for bit, whatever tile a shape is routed to.
__device__ __forceinline__ void gelu_tab_scalar(float x, float& y, float& dy) {
  u16x2_t mx = {0, 0};
  const uint32_t off = gelu_tab_offsets(pack2bf(x, x), mx) & 0xffffu;
  const uint32_t e = *(const uint32_t*)((const char*)g_gelu_tab + off);
  y = bflo(e); dy = bfhi(e);
  const float xr = bf2f(f2bf(x));                                            // the bf16-rounded pre-activation, as in the look-up
  if (!(fabsf(xr) < GELU_TAB_TOP)) { bf16_t hb, db; gelu_big(xr, hb, db); y = bf2f(hb); dy = bf2f(db); }
}
__device__ __forceinline__ void gelu_tab_both4(f32x2_t& x0, f32x2_t& x1, f32x2_t& d0, f32x2_t& d1) {
  float y, d;
  gelu_tab_scalar(x0[0], y, d); x0[0] = y; d0[0] = d;
  gelu_tab_scalar(x0[1], y, d); x0[1] = y; d0[1] = d;
  gelu_tab_scalar(x1[0], y, d); x1[0] = y; d1[0] = d;
  gelu_tab_scalar(x1[1], y, d); x1[1] = y; d1[1] = d;
}

// ---- operand tile loaders (LDS-DMA, swizzle on the source address) ------------------------------------
// k-major operand: LDS tile [BR][BK] bf16 (BK*2-byte rows).  The 16-B chunk c of row r lives at chunk
// c ^ ((r>>1)&7) for BK=64 (128-B rows) and c ^ ((r>>2)&3) for BK=32 (64-B rows): the 16 rows a ds_read_b128 lane
// group touches then land on 16 distinct 16-B slots of the 256-B bank row.
template <int BR>
__device__ __forceinline__ int kstrided_swz_fwd(int r) { return BR >= 128 ? ((r & 3) << 2) : (((r >> 1) & 1) << 2); }
template <int BK>
__device__ __forceinline__ int kmajor_swz(int r) { return BK == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3); }
template <int BR, int NW, int BK>
__device__ __forceinline__ void stage_kmajor_part(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int row0, int k0, int ld,
                                                  int K, int wave, int lane, int j) {
  constexpr int CPR = BK / 8;
  constexpr int RPI = 64 / CPR;
  int r = j * (NW * RPI) + wave * RPI + lane / CPR;
  int c = (lane % CPR) ^ kmajor_swz<BK>(r);
  int kcol = k0 + c * 8;
  uint32_t off = (uint32_t)(((size_t)(row0 + r) * (size_t)ld + (size_t)kcol) * 2);
  if (kcol >= K) off = 0xFFFFFFF0u;
  char* dst = lds_tile + (j * (NW * RPI) + wave * RPI) * (BK * 2);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, AVT_LDS_PTR(dst), 16, off, 0, 0, 0);
}
template <int BR, int NW, int BK>
__device__ __forceinline__ void stage_kstrided_part(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int col0, int k0, int ld,
                                                    int ncols, int wave, int lane, int j) {
  constexpr int CPR = BR / 8;
  constexpr int RPI = 64 / CPR;
  int r = j * NW * RPI + wave * RPI + lane / CPR;
  int c = (lane % CPR) ^ kstrided_swz_fwd<BR>(r);
  int col = col0 + c * 8;
  uint32_t off = (uint32_t)(((size_t)(k0 + r) * (size_t)ld + (size_t)col) * 2);
  if (col >= ncols) off = 0xFFFFFFF0u;
  char* dst = lds_tile + (j * NW * RPI + wave * RPI) * (BR * 2);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, AVT_LDS_PTR(dst), 16, off, 0, 0, 0);
}
template <int BR, int NW, int BK>
__device__ __forceinline__ void stage_kmajor(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int row0, int k0, int ld,
                                             int K, int wave, int lane) {
  constexpr int CPR = BK / 8;            // 16-B chunks per row
  constexpr int RPI = 64 / CPR;          // rows per wave instruction
#pragma unroll
  for (int j = 0; j < BR / (NW * RPI); ++j) {
    int r = j * (NW * RPI) + wave * RPI + lane / CPR;
    int c = (lane % CPR) ^ kmajor_swz<BK>(r);
    int kcol = k0 + c * 8;
    uint32_t off = (uint32_t)(((size_t)(row0 + r) * (size_t)ld + (size_t)kcol) * 2);
    if (kcol >= K) off = 0xFFFFFFF0u;                     // forces the bounds check -> zeros
    char* dst = lds_tile + (j * (NW * RPI) + wave * RPI) * (BK * 2);   // wave-uniform base; lane l lands at +16*l
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, AVT_LDS_PTR(dst), 16, off, 0, 0, 0);
  }
}
// reduction-index-as-row operand: LDS tile [BK][BR] bf16; chunk swizzle keeps the 4 rows of a tr-read on
// distinct 64-B bank segments (BR>=128: c ^ ((r&3)<<2); BR=64: c ^ (((r>>1)&1)<<2)).
template <int BR>
__device__ __forceinline__ int kstrided_swz(int r) { return BR >= 128 ? ((r & 3) << 2) : (((r >> 1) & 1) << 2); }
template <int BR, int NW, int BK>
__device__ __forceinline__ void stage_kstrided(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int col0, int k0, int ld,
                                               int ncols, int wave, int lane) {
  constexpr int CPR = BR / 8;            // 16-B chunks per row
  constexpr int RPI = 64 / CPR;          // rows per wave instruction
#pragma unroll
  for (int j = 0; j < BK / (NW * RPI); ++j) {
    int r = j * NW * RPI + wave * RPI + lane / CPR;
    int c = (lane % CPR) ^ kstrided_swz<BR>(r);
    int col = col0 + c * 8;
    uint32_t off = (uint32_t)(((size_t)(k0 + r) * (size_t)ld + (size_t)col) * 2);
    if (col >= ncols) off = 0xFFFFFFF0u;
    char* dst = lds_tile + (j * NW * RPI + wave * RPI) * (BR * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, AVT_LDS_PTR(dst), 16, off, 0, 0, 0);
  }
}

// ---- fragment reads --------------------------------------------------------------------------------------
// Both forms give lane l the 8 values k = ks*16 + (l>>5)*8 + e, e = 0..7, of operand row (tile*32 + (l&31)).
template <int BK>
__device__ __forceinline__ bf16x8_t frag_kmajor(const char* lds_tile, int tile, int ks, int lane) {
  int r = tile * 32 + (lane & 31);
  int c = (ks * 2 + (lane >> 5)) ^ kmajor_swz<BK>(r);
  return *(const bf16x8_t*)(lds_tile + r * (BK * 2) + c * 16);
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

// frag_kstrided through inline assembly (see ds_read_tr_na in common.hpp): no compiler-placed s_waitcnt vmcnt(0); the caller
// waits (frag_wait) before the first MFMA that consumes the fragment.  Rows r and r + 4 share the swizzle: one address, two immediates.
template <int BR>
__device__ __forceinline__ bf16x8_t frag_kstrided_na(const char* lds_tile, int tile, int ks, int lane) {
  const int g = lane >> 4, i16 = lane & 15;
  const int col = tile * 32 + (g & 1) * 16 + (i16 & 3) * 4;
  const int r = ks * 16 + (g >> 1) * 8 + (i16 >> 2);
  const int c = (col >> 3) ^ kstrided_swz<BR>(r);
  const uint32_t a = lds_addr32(lds_tile + r * (BR * 2) + c * 16 + (col & 7) * 2);
  const u32x2_t lo = ds_read_tr_na<0>(a), hi = ds_read_tr_na<4 * BR * 2>(a);
  return tr_join(lo, hi);
}
// wait until at most N of this wave's LDS operations are outstanding, then pass the fragments through an empty statement so
// that no MFMA reading them can be scheduled above the wait
template <int N, int TM, int TN>
__device__ __forceinline__ void frag_wait(bf16x8_t (&a)[TM], bf16x8_t (&b)[TN]) {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
#pragma unroll
  for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(a[i]));
#pragma unroll
  for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(b[j]));
}
// The same transposing read issued through inline assembly, for kernels that place their own s_waitcnt.  hipcc cannot tell that
// a __builtin_amdgcn_ds_read_tr16_b64 does not alias the LDS-DMA (buffer_load ... lds) transfers still in flight and puts an
// s_waitcnt vmcnt(0) in front of every group of them: the whole ring drains before each fragment read and the 1.5-K-tile
// prefetch of the 8-phase kernel degenerates to none (found in round 3: every k-strided operand -- all weight gradients, the
// proj data gradient -- had been running like that; plain ds_read_b128 loads are not affected).  `addr` = the lane's LDS byte
// address, OFF = compile-time offset (slot, k-step, half).  The result may only be used after an explicit s_waitcnt lgkmcnt.
// the four k-steps of one 32-column block of a [64][128] k-strided half-tile (256-B rows): f[ks] = rows ks*16 .. +15
template <int OFF>
__device__ __forceinline__ void frag4_tr_na(bf16x8_t (&f)[4], uint32_t addr) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    u32x2_t lo, hi;
    switch (ks) {             // compile-time immediates
      case 0: lo = ds_read_tr_na<OFF>(addr); hi = ds_read_tr_na<OFF + 4 * 256>(addr); break;
      case 1: lo = ds_read_tr_na<OFF + 16 * 256>(addr); hi = ds_read_tr_na<OFF + 20 * 256>(addr); break;
      case 2: lo = ds_read_tr_na<OFF + 32 * 256>(addr); hi = ds_read_tr_na<OFF + 36 * 256>(addr); break;
      default: lo = ds_read_tr_na<OFF + 48 * 256>(addr); hi = ds_read_tr_na<OFF + 52 * 256>(addr); break;
    }
    f[ks] = tr_join(lo, hi);
  }
}
// De-synchronise the chip: all CUs start together and would otherwise hit their output-store tails together (a burst at
// the HBM write rate while the MFMA pipes idle).  The workgroups of the FIRST dispatch wave start spread over
// `cycles`; every CU keeps its offset afterwards because it picks up its next tile when it finishes the previous one.
__device__ __forceinline__ void stagger_start(int cycles, int bid) {
  if (cycles <= 0 || bid >= 256) return;
  const long long target = (long long)cycles * ((bid >> 3) & 31) / 32;
  const long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < target) __builtin_amdgcn_s_sleep(16);
}

#ifdef AVT_LAB
// lab: two workgroups share a CU; the one in the odd threadgroup slot of the FIRST dispatch wave starts `cycles` late, so that
// afterwards one workgroup's epilogue (vector ALU, stores) runs under the other's K loop (matrix pipe) instead of both doing
// the same thing at the same time.  HW_REG_HW_ID (id 4): TG_ID = bits 19:16.
__device__ __forceinline__ void stagger_slot(int cycles, int bid, int nfirst) {
  if (cycles <= 0 || bid >= nfirst) return;
  uint32_t hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if (((hw >> 16) & 1u) == 0) return;
  const long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < cycles) __builtin_amdgcn_s_sleep(32);
}
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- activation epilogue, factored so that it can run in one go or be dribbled out under the next tile's K loop ------
// Per-lane state: the lane owns 8 consecutive output columns (n .. n+7) of every row it touches.
struct EpiLane {
  int n, cl; bool ncol_ok, wide;
  float bias8[8]; float csum[8];
};
// reload the bias strip (used by the dribbled epilogue, which cannot afford to keep it in registers across a K loop)
__device__ __forceinline__ void epi_load_bias(EpiLane& e, const GemmParams& p) {
#pragma unroll
  for (int q = 0; q < 8; ++q) e.bias8[q] = 0.f;
  if (p.bias && e.ncol_ok) {
    f32x4_t b = *(const f32x4_t*)(p.bias + e.n);
    e.bias8[0] = b[0]; e.bias8[1] = b[1]; e.bias8[2] = b[2]; e.bias8[3] = b[3];
    if (e.n + 4 < p.N) { f32x4_t c = *(const f32x4_t*)(p.bias + e.n + 4); e.bias8[4] = c[0]; e.bias8[5] = c[1]; e.bias8[6] = c[2]; e.bias8[7] = c[3]; }
  }
}
template <int WN>
__device__ __forceinline__ void epi_setup(EpiLane& e, const GemmParams& p, int lane, int col0) {
  constexpr int LPR = WN / 8;
  e.cl = (lane % LPR) * 8;
  e.n = col0 + e.cl;
  e.ncol_ok = e.n < p.N;                 // N % 4 == 0 is a host-checked precondition
  e.wide = p.wide_ok && (e.n + 8 <= p.N);
#pragma unroll
  for (int q = 0; q < 8; ++q) { e.bias8[q] = 0.f; e.csum[q] = 0.f; }
  if (p.bias && e.ncol_ok) {
    f32x4_t b = *(const f32x4_t*)(p.bias + e.n);
    e.bias8[0] = b[0]; e.bias8[1] = b[1]; e.bias8[2] = b[2]; e.bias8[3] = b[3];
    if (e.n + 4 < p.N) { f32x4_t c = *(const f32x4_t*)(p.bias + e.n + 4); e.bias8[4] = c[0]; e.bias8[5] = c[1]; e.bias8[6] = c[2]; e.bias8[7] = c[3]; }
  }
}
// W (8 or 4) consecutive columns starting at column offset `co` of the lane's strip; `src` = the strip's 8 values (fp32), row m
struct EpiStrip { u32x4_t w; };        // 8 bf16 of a second operand (saved derivative / residual) for one row strip
__device__ __forceinline__ EpiStrip epi_load_strip(const bf16_t* base, int ld, int m, const EpiLane& e, const GemmParams& p) {
  EpiStrip s; s.w = (u32x4_t){0u, 0u, 0u, 0u};
  const bf16_t* ptr = base + (size_t)m * ld + e.n;
  if (e.wide) s.w = *(const u32x4_t*)ptr;
  else {
    u32x2_t a = *(const u32x2_t*)ptr; s.w[0] = a[0]; s.w[1] = a[1];
    if (e.n + 4 < p.N) { u32x2_t b = *(const u32x2_t*)(ptr + 4); s.w[2] = b[0]; s.w[3] = b[1]; }
  }
  return s;
}
template <int W>
__device__ __forceinline__ void epi_cols(EpiLane& e, const GemmParams& p, const float (&src)[8], int co, int m,
                                         const EpiStrip& aux_s, const EpiStrip& res_s) {
  const int nn = e.n + co;
  float v[W];
#pragma unroll
  for (int k = 0; k < W; ++k) v[k] = src[co + k] + e.bias8[co + k];
  auto store_bf = [&](bf16_t* ptr, const float* x) {
    if (W == 8) {
      u32x4_t o; o[0] = pack2bf(x[0], x[1]); o[1] = pack2bf(x[2], x[3]); o[2] = pack2bf(x[4], x[5]); o[3] = pack2bf(x[6], x[7]);
      *(u32x4_t*)ptr = o;
    } else {
      u32x2_t o; o[0] = pack2bf(x[0], x[1]); o[1] = pack2bf(x[2], x[3]);
      *(u32x2_t*)ptr = o;
    }
  };
  if (p.act == 3) {                       // backward of an activation: multiply by the saved derivative
#pragma unroll
    for (int k = 0; k < W; ++k) { uint32_t w = aux_s.w[(co + k) >> 1]; v[k] *= ((co + k) & 1) ? bfhi(w) : bflo(w); }
  }
  if (p.act == 1 || p.act == 2) {         // GELU; the optional second output is GELU'(pre-activation) for backward
    float d[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
      if (p.act == 1) gelu_tab_scalar(v[k], v[k], d[k]); else gelu_tanh_both(v[k], v[k], d[k]);
    }
    if (p.C2) store_bf(p.C2 + (size_t)m * p.ldc2 + nn, d);
  } else if (p.C2) {
    store_bf(p.C2 + (size_t)m * p.ldc2 + nn, v);
  }
  if (p.drop_thresh) {
#pragma unroll
    for (int k = 0; k < W; ++k)
      v[k] = drop_keep(p.drop_seed, (uint64_t)m * (uint64_t)p.N + (uint64_t)(nn + k), p.drop_thresh) ? v[k] * p.drop_scale : 0.f;
  }
  if (p.res) {
#pragma unroll
    for (int k = 0; k < W; ++k) { uint32_t w = res_s.w[(co + k) >> 1]; v[k] += ((co + k) & 1) ? bfhi(w) : bflo(w); }
  }
  if (p.colsum) {
#pragma unroll
    for (int k = 0; k < W; ++k) e.csum[co + k] += v[k];
  }
  if (p.out_f32) {
    float* crow = (float*)p.C + (size_t)m * p.ldc + nn;
#pragma unroll
    for (int q = 0; q < W / 4; ++q) *(f32x4_t*)(crow + 4 * q) = (f32x4_t){v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
  } else {
    store_bf((bf16_t*)p.C + (size_t)m * p.ldc + nn, v);
  }
}
// one output row (global row m): the lane's 8 values of it
// `prim` = the row's strip of the PRIMARY second operand (saved derivative when act == 3, else the residual): staged through
// LDS by LDS-DMA when `staged`, else (narrow leading dimensions) loaded here; a residual next to act == 3 is always loaded here
__device__ __forceinline__ void epi_row(EpiLane& e, const GemmParams& p, const float (&src)[8], int m, EpiStrip prim, bool staged) {
  if (m < p.M && e.ncol_ok) {
    EpiStrip res_s = prim;
    if (p.res && (p.act == 3 || !staged)) res_s = epi_load_strip(p.res, p.ldres, p.res_period ? (m % p.res_period) : m, e, p);
    if (p.act == 3 && !staged) prim = epi_load_strip(p.aux, p.ldaux, m, e, p);
    if (e.wide) epi_cols<8>(e, p, src, 0, m, prim, res_s);
    else {
      epi_cols<4>(e, p, src, 0, m, prim, res_s);
      if (e.n + 4 < p.N) epi_cols<4>(e, p, src, 4, m, prim, res_s);
    }
  }
}
template <int WN>
__device__ __forceinline__ void epi_flush_colsum(EpiLane& e, const GemmParams& p, int lane, int slot) {
  if (!p.colsum) return;
  constexpr int LPR = WN / 8;
  // lanes sharing (lane % LPR) own the same 8 columns: fold the row groups, one atomic per column
#pragma unroll
  for (int q = 0; q < 8; ++q) {
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) e.csum[q] += __shfl_xor(e.csum[q], o, 64);
  }
  if (lane < LPR && e.ncol_ok) {
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (e.n + q < p.N) {
        if (p.colsum_part) p.colsum_part[(size_t)slot * p.N + e.n + q] = e.csum[q];
        else unsafeAtomicAdd(&p.colsum[e.n + q], e.csum[q]);
      }
  }
}
// per-wave LDS of the activation epilogue: general path = fp32 patch [32][WN+4] + 2 second-operand buffers [32][WN] bf16,
// fast path = bias strip + 2 bf16 patches [32][WN*2+8 bytes] + 2 second-operand buffers [32][WN] bf16
template <int WN> constexpr int epi_wave_lds() {
  constexpr int general = 32 * (WN + 4) * 4 + 2 * 32 * WN * 2;
  constexpr int fast = WN * 4 + 2 * 32 * (WN * 2 + 8) + 2 * 32 * WN * 2;     // + 2 second-operand buffers
  return ((general > fast ? general : fast) + 15) & ~15;
}
// ---- general path: 32-row block of the wave tile -> wave-private fp32 LDS patch (4 consecutive columns = one ds_write_b128;
// rows 272 B apart, so the 8 lanes of a store group and the 2 rows of a load group sit on disjoint banks)
template <int TN> struct EpiBlk { f32x16_t t[TN]; };     // one 32-row block of the wave tile, passed by value (keeps the accumulators in registers)
template <int TM, int TN, int I>
__device__ __forceinline__ EpiBlk<TN> epi_take(const f32x16_t (&acc)[TM][TN]) {
  EpiBlk<TN> b;
#pragma unroll
  for (int j = 0; j < TN; ++j) b.t[j] = acc[I < TM ? I : 0][j];
  return b;
}
template <int TN, int WN>
__device__ __forceinline__ void epi_write_block(float* patch, const EpiBlk<TN> blk_, int lane) {
  const f32x16_t* blk = blk_.t;
  constexpr int LDP = WN + 4;
  float* dst = patch + (lane & 31) * LDP + 4 * (lane >> 5);
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *(f32x4_t*)(dst + j * 32 + 8 * q) = (f32x4_t){blk[j][4 * q], blk[j][4 * q + 1], blk[j][4 * q + 2], blk[j][4 * q + 3]};
}
template <int TM, int TN, int WN>
__device__ __forceinline__ void epi_write_block_i(float* patch, const f32x16_t (&acc)[TM][TN], int i, int lane) {
  switch (i) {          // accumulator registers need compile-time indices
    case 0: epi_write_block<TN, WN>(patch, epi_take<TM, TN, 0>(acc), lane); break;
    case 1: if (TM > 1) epi_write_block<TN, WN>(patch, epi_take<TM, TN, 1>(acc), lane); break;
    case 2: if (TM > 2) epi_write_block<TN, WN>(patch, epi_take<TM, TN, 2>(acc), lane); break;
    case 3: if (TM > 3) epi_write_block<TN, WN>(patch, epi_take<TM, TN, 3>(acc), lane); break;
    default: break;
  }
}

// 16-byte global stores of the epilogue outputs with a selectable L2 policy.  AVT_ST_AUX = 0: plain stores (the line stays in
// the XCD's L2: an M x 3072 activation written by one GEMM is consumed hundreds of microseconds later by another kernel, so all
// it does there is push the B operand out); 2 = nt (the line stays but is the first to go); 16 = sc1 (write-through, the line is
// dropped from L2 -- MI355X_MICROARCH.md, price list "stores of each flavour").  Measured on the whole step (256 clips, same box,
// profiles/r04_cache_policy.txt): plain 878.7 / 880.3 clips/s, sc1 883.6 / 882.7, nt 889.2 / 887.4 (+1.0 %; fc1 forward 3094 ->
// 2990 us with sc1) -> nt is the product's policy.  Rows are addressed relative to the wave tile's origin through a buffer descriptor.
#ifndef AVT_ST_AUX
#define AVT_ST_AUX 2
#endif
struct TileStore {
  __amdgpu_buffer_rsrc_t r; bf16_t* base; int ld;
  __device__ __forceinline__ void init(bf16_t* origin, int ld_) {
    base = origin; ld = ld_;
    if (AVT_ST_AUX != 0) r = __builtin_amdgcn_make_buffer_rsrc((void*)origin, 0, 0xFFFFFFF0u, 0x00020000);
  }
  __device__ __forceinline__ void st(int drow, int dcol, u32x4_t v) const {
    if (AVT_ST_AUX == 0) *(u32x4_t*)(base + (size_t)drow * ld + dcol) = v;
    else __builtin_amdgcn_raw_buffer_store_b128(v, r, (uint32_t)((drow * ld + dcol) * 2), 0, AVT_ST_AUX);
  }
};

// ---- fast path (bias / GELU (+ GELU') only, bf16 output): all arithmetic in the accumulator layout with packed fp32 ops,
// bf16 pairs through a [32][WN] bf16 patch (ds_write_b64 in, 16 B per lane out), one 16-byte global store per lane and row
template <int TN, int WN, int ACT, bool TAB = false>
__device__ __forceinline__ void epi_fast_block(const GemmParams& p, const EpiBlk<TN> blk_, char* patch_c, char* patch_d,
                                               const float* bias_l, int lane, int m0, int col0, const TileStore& sc, const TileStore& sd, int i32,
                                               const char* tab = nullptr) {
  const f32x16_t* blk = blk_.t;
  constexpr int LDB = WN * 2 + 8;                                  // patch row pitch (bytes): 16 store lanes -> 32 distinct banks
  const int ml = lane & 31, h = lane >> 5;
  if constexpr (TAB && ACT == 1) {
    // (the derivative patch is written whether or not the caller wants the second output: no per-element branch on it)
    u16x2_t mx = {0, 0};
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      // all 16 look-ups of a 32-column slice are in flight together (one LDS round trip per slice, not per pair)
      uint32_t off[8], e[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nl = j * 32 + 8 * q + 4 * h;
        const f32x4_t b = *(const f32x4_t*)(bias_l + nl);
        const f32x2_t v0 = (f32x2_t){blk[j][4 * q], blk[j][4 * q + 1]} + (f32x2_t){b[0], b[1]};
        const f32x2_t v1 = (f32x2_t){blk[j][4 * q + 2], blk[j][4 * q + 3]} + (f32x2_t){b[2], b[3]};
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
        const int nl = j * 32 + 8 * q + 4 * h;
        const uint32_t h0 = __builtin_amdgcn_perm(e[4 * q + 1], e[4 * q], 0x05040100u), d0 = __builtin_amdgcn_perm(e[4 * q + 1], e[4 * q], 0x07060302u);
        const uint32_t h1 = __builtin_amdgcn_perm(e[4 * q + 3], e[4 * q + 2], 0x05040100u), d1 = __builtin_amdgcn_perm(e[4 * q + 3], e[4 * q + 2], 0x07060302u);
        *(u32x2_t*)(patch_d + ml * LDB + nl * 2) = (u32x2_t){d0, d1};
        *(u32x2_t*)(patch_c + ml * LDB + nl * 2) = (u32x2_t){h0, h1};
      }
    }
    constexpr unsigned short HI = ((GELU_TAB_ELO + GELU_TAB_NEXP) << 7) - 1;
    if (__builtin_expect(__any((mx[0] > HI) | (mx[1] > HI)), 0)) {          // some value of this block lies above the table: patch those elements
#pragma unroll 1
      for (int j = 0; j < TN; ++j)
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
          const int nl = j * 32 + 8 * q + 4 * h;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x = bf2f(f2bf(blk[j][4 * q + e] + bias_l[nl + e]));       // the bf16-rounded pre-activation, as in the look-up
            if (!(fabsf(x) < GELU_TAB_TOP)) {
              bf16_t hb, db; gelu_big(x, hb, db);
              *(bf16_t*)(patch_c + ml * LDB + (nl + e) * 2) = hb;
              if (p.C2) *(bf16_t*)(patch_d + ml * LDB + (nl + e) * 2) = db;
            }
          }
        }
    }
  } else
  {
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int nl = j * 32 + 8 * q + 4 * h;
      const f32x4_t b = *(const f32x4_t*)(bias_l + nl);
      f32x2_t v0 = (f32x2_t){blk[j][4 * q], blk[j][4 * q + 1]} + (f32x2_t){b[0], b[1]};
      f32x2_t v1 = (f32x2_t){blk[j][4 * q + 2], blk[j][4 * q + 3]} + (f32x2_t){b[2], b[3]};
      if (ACT != 0) {
        f32x2_t d0, d1;
        if (ACT == 1) gelu_tab_both4(v0, v1, d0, d1);
        else {
          float y4[4], d4[4];
          gelu_tanh_both(v0[0], y4[0], d4[0]); gelu_tanh_both(v0[1], y4[1], d4[1]);
          gelu_tanh_both(v1[0], y4[2], d4[2]); gelu_tanh_both(v1[1], y4[3], d4[3]);
          v0 = (f32x2_t){y4[0], y4[1]}; v1 = (f32x2_t){y4[2], y4[3]}; d0 = (f32x2_t){d4[0], d4[1]}; d1 = (f32x2_t){d4[2], d4[3]};
        }
        if (p.C2) *(u32x2_t*)(patch_d + ml * LDB + nl * 2) = (u32x2_t){pack2bf(d0[0], d0[1]), pack2bf(d1[0], d1[1])};
      }
      *(u32x2_t*)(patch_c + ml * LDB + nl * 2) = (u32x2_t){pack2bf(v0[0], v0[1]), pack2bf(v1[0], v1[1])};
    }
  }
  constexpr int LPR = WN / 8, RPI = 64 / LPR, IT = 32 / RPI;
  const int rl = lane / LPR, cl = (lane % LPR) * 8;
  const int n = col0 + cl;
  // with no activation the second output is the same tensor as the first (C2 = value before dropout, and the fast path has none)
  const char* patch_2 = (ACT != 0) ? patch_d : patch_c;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int row = it * RPI + rl, m = m0 + row;
    const char* src = patch_c + row * LDB + cl * 2;
    u32x2_t lo = *(const u32x2_t*)src, hi = *(const u32x2_t*)(src + 8);
    u32x2_t lo2 = lo, hi2 = hi;
    if (p.C2) { const char* s2 = patch_2 + row * LDB + cl * 2; lo2 = *(const u32x2_t*)s2; hi2 = *(const u32x2_t*)(s2 + 8); }
    if (m < p.M && n < p.N) {
      sc.st(i32 + row, cl, (u32x4_t){lo[0], lo[1], hi[0], hi[1]});
      if (p.C2) sd.st(i32 + row, cl, (u32x4_t){lo2[0], lo2[1], hi2[0], hi2[1]});
    }
  }
}
template <int TM, int TN, int WN, int ACT, bool TAB = false>
__device__ __forceinline__ void epi_fast(const GemmParams& p, const f32x16_t (&acc)[TM][TN], char* wave_lds, int lane, int row0, int col0,
                                         const char* tab = nullptr) {
  constexpr int LDB = WN * 2 + 8;
  float* bias_l = (float*)wave_lds;
  char* patch_c = wave_lds + WN * 4;
  char* patch_d = patch_c + 32 * LDB;
  for (int c_ = lane; c_ < WN; c_ += 64) bias_l[c_] = (p.bias && col0 + c_ < p.N) ? p.bias[col0 + c_] : 0.f;
  TileStore sc, sd;
  sc.init((bf16_t*)p.C + (size_t)row0 * p.ldc + col0, p.ldc);
  sd.init(p.C2 ? p.C2 + (size_t)row0 * p.ldc2 + col0 : (bf16_t*)p.C, p.ldc2);
  if (ACT == 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      EpiBlk<TN> b;
#pragma unroll
      for (int j = 0; j < TN; ++j) b.t[j] = acc[i][j];
      epi_fast_block<TN, WN, ACT>(p, b, patch_c, patch_d, bias_l, lane, row0 + i * 32, col0, sc, sd, i * 32);
    }
  } else {
    // the GELU arithmetic of one block is ~1.5 k instructions: keep ONE copy of it (instruction cache) and move the block
    // into place instead (32 register copies per block)
#pragma unroll 1
    for (int i = 0; i < TM; ++i) {
      EpiBlk<TN> b;
      switch (i) {
        case 0: b = epi_take<TM, TN, 0>(acc); break;
        case 1: b = epi_take<TM, TN, 1>(acc); break;
        case 2: b = epi_take<TM, TN, 2>(acc); break;
        default: b = epi_take<TM, TN, 3>(acc); break;
      }
      epi_fast_block<TN, WN, ACT, TAB>(p, b, patch_c, patch_d, bias_l, lane, row0 + i * 32, col0, sc, sd, i * 32, tab);
    }
  }
}

// ---- extended fast path: the same register-layout arithmetic plus everything that needs a second operand or an index:
//   v = acc + bias;  act 3: v *= aux;  act 1|2: GELU (+ GELU' -> C2);  act 0 with C2: C2 = v;  dropout;  v += res;
//   column sums;  C = bf16(v)
// The second operand (aux when act == 3, else res) of block i is brought in by LDS-DMA two blocks ahead.  LDS-DMA lands
// lane L's 16 bytes at buffer + 16 L, so the layout is chosen through the SOURCE address: position (row r, chunk pc) holds
// chunk pc ^ ((r >> 1) & (LPR-1)) of row r, which makes the 8-byte reads of the accumulator layout (32 lanes = 32 rows, same
// column) at most 2-way bank conflicted.  Column sums (bias gradients) are taken over the bf16-rounded outputs on their way
// out (row-strip layout: 8 running sums per lane, folded across the lanes that share columns once per tile).
template <int TM, int TN, int WN, int ACT>
__device__ __forceinline__ void epi_fast_ext(const GemmParams& p, const f32x16_t (&acc)[TM][TN], char* wave_lds, int lane, int row0, int col0) {
  constexpr int LDB = WN * 2 + 8;
  constexpr int LPR = WN / 8, RPI = 64 / LPR, IT = 32 / RPI;
  constexpr int OPB = 32 * WN * 2;
  float* bias_l = (float*)wave_lds;
  char* patch_c = wave_lds + WN * 4;
  char* patch_d = patch_c + 32 * LDB;
  char* opbuf = patch_d + 32 * LDB;
  for (int c_ = lane; c_ < WN; c_ += 64) bias_l[c_] = (p.bias && col0 + c_ < p.N) ? p.bias[col0 + c_] : 0.f;
  const int ml = lane & 31, h = lane >> 5;
  const int rl = lane / LPR, pc = lane % LPR, cl = pc * 8;
  const bf16_t* prim_ptr = (ACT == 3) ? p.aux : p.res;
  const int prim_ld = (ACT == 3) ? p.ldaux : p.ldres;
  const int prim_period = (ACT == 3) ? 0 : p.res_period;
  const bool has_prim = prim_ptr != nullptr;
  __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)prim_ptr, 0, 0xFFFFFFF0u, 0x00020000);
  auto dma_block = [&](int i) __attribute__((always_inline)) {
#pragma unroll
    for (int itr = 0; itr < IT; ++itr) {
      const int r = itr * RPI + rl;
      const int m = row0 + i * 32 + r;
      const int n = col0 + ((pc ^ ((r >> 1) & (LPR - 1))) * 8);
      const int mr = prim_period ? (m % prim_period) : m;
      uint32_t off = (uint32_t)(((size_t)mr * (size_t)prim_ld + (size_t)n) * 2);
      if (m >= p.M || n >= p.N) off = 0xFFFFFFF0u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, AVT_LDS_PTR(opbuf + (i & 1) * OPB + itr * 1024), 16, off, 0, 0, AVT_LDP_AUX);
    }
  };
  if (has_prim) { dma_block(0); if (TM > 1) dma_block(1); }
  TileStore sc, sd;
  sc.init((bf16_t*)p.C + (size_t)row0 * p.ldc + col0, p.ldc);
  sd.init(p.C2 ? p.C2 + (size_t)row0 * p.ldc2 + col0 : (bf16_t*)p.C, p.ldc2);
  float cs[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) cs[k] = 0.f;
#pragma unroll (ACT == 0 || ACT == 3 ? 4 : 1)
  for (int i = 0; i < TM; ++i) {
    EpiBlk<TN> b;
    switch (i) {
      case 0: b = epi_take<TM, TN, 0>(acc); break;
      case 1: b = epi_take<TM, TN, 1>(acc); break;
      case 2: b = epi_take<TM, TN, 2>(acc); break;
      default: b = epi_take<TM, TN, 3>(acc); break;
    }
    if (has_prim) {          // see the general path for the counts
      switch (i) {
        case 0: if (TM > 1) wait_vmcnt<IT>(); else wait_vmcnt<0>(); break;
        case 1: if (TM > 2) wait_vmcnt<2 * IT>(); else wait_vmcnt<IT>(); break;
        case 2: if (TM > 3) wait_vmcnt<3 * IT>(); else wait_vmcnt<2 * IT>(); break;
        default: wait_vmcnt<2 * IT>(); break;
      }
    }
    const int m = row0 + i * 32 + ml;
    u32x2_t opv[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        opv[j][q] = (u32x2_t){0u, 0u};
        if (has_prim) opv[j][q] = *(const u32x2_t*)(opbuf + (i & 1) * OPB + ml * (WN * 2) + (((j * 4 + q) ^ ((ml >> 1) & (LPR - 1))) * 16) + h * 8);
      }
    if (has_prim && i + 2 < TM) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the buffer's previous contents are in registers
      dma_block(i + 2);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nl = j * 32 + 8 * q + 4 * h;
        const f32x4_t bb = *(const f32x4_t*)(bias_l + nl);
        f32x2_t v0 = (f32x2_t){b.t[j][4 * q], b.t[j][4 * q + 1]} + (f32x2_t){bb[0], bb[1]};
        f32x2_t v1 = (f32x2_t){b.t[j][4 * q + 2], b.t[j][4 * q + 3]} + (f32x2_t){bb[2], bb[3]};
        const f32x2_t o0 = (f32x2_t){bflo(opv[j][q][0]), bfhi(opv[j][q][0])}, o1 = (f32x2_t){bflo(opv[j][q][1]), bfhi(opv[j][q][1])};
        if (ACT == 3) { v0 *= o0; v1 *= o1; }
        if (ACT == 1 || ACT == 2) {
          f32x2_t d0, d1;
          if (ACT == 1) gelu_tab_both4(v0, v1, d0, d1);
          else {
            float y4[4], d4[4];
            gelu_tanh_both(v0[0], y4[0], d4[0]); gelu_tanh_both(v0[1], y4[1], d4[1]);
            gelu_tanh_both(v1[0], y4[2], d4[2]); gelu_tanh_both(v1[1], y4[3], d4[3]);
            v0 = (f32x2_t){y4[0], y4[1]}; v1 = (f32x2_t){y4[2], y4[3]}; d0 = (f32x2_t){d4[0], d4[1]}; d1 = (f32x2_t){d4[2], d4[3]};
          }
          if (p.C2) *(u32x2_t*)(patch_d + ml * LDB + nl * 2) = (u32x2_t){pack2bf(d0[0], d0[1]), pack2bf(d1[0], d1[1])};
        } else if (p.C2) {
          *(u32x2_t*)(patch_d + ml * LDB + nl * 2) = (u32x2_t){pack2bf(v0[0], v0[1]), pack2bf(v1[0], v1[1])};
        }
        if (p.drop_thresh) {
          const uint64_t idx = (uint64_t)m * (uint64_t)p.N + (uint64_t)(col0 + nl);
          v0[0] = drop_keep(p.drop_seed, idx, p.drop_thresh) ? v0[0] * p.drop_scale : 0.f;
          v0[1] = drop_keep(p.drop_seed, idx + 1, p.drop_thresh) ? v0[1] * p.drop_scale : 0.f;
          v1[0] = drop_keep(p.drop_seed, idx + 2, p.drop_thresh) ? v1[0] * p.drop_scale : 0.f;
          v1[1] = drop_keep(p.drop_seed, idx + 3, p.drop_thresh) ? v1[1] * p.drop_scale : 0.f;
        }
        if (ACT != 3 && has_prim) { v0 += o0; v1 += o1; }
        *(u32x2_t*)(patch_c + ml * LDB + nl * 2) = (u32x2_t){pack2bf(v0[0], v0[1]), pack2bf(v1[0], v1[1])};
      }
    const int n = col0 + cl;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int row = it * RPI + rl, mm = row0 + i * 32 + row;
      const char* src = patch_c + row * LDB + cl * 2;
      u32x2_t lo = *(const u32x2_t*)src, hi = *(const u32x2_t*)(src + 8);
      u32x2_t lo2 = lo, hi2 = hi;
      if (p.C2) { const char* s2 = patch_d + row * LDB + cl * 2; lo2 = *(const u32x2_t*)s2; hi2 = *(const u32x2_t*)(s2 + 8); }
      if (mm < p.M && n < p.N) {
        sc.st(i * 32 + row, cl, (u32x4_t){lo[0], lo[1], hi[0], hi[1]});
        if (p.C2) sd.st(i * 32 + row, cl, (u32x4_t){lo2[0], lo2[1], hi2[0], hi2[1]});
        if (p.colsum) {
          cs[0] += bflo(lo[0]); cs[1] += bfhi(lo[0]); cs[2] += bflo(lo[1]); cs[3] += bfhi(lo[1]);
          cs[4] += bflo(hi[0]); cs[5] += bfhi(hi[0]); cs[6] += bflo(hi[1]); cs[7] += bfhi(hi[1]);
        }
      }
    }
  }
  if (p.colsum) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) cs[q] += __shfl_xor(cs[q], o, 64);
    }
    if (lane < LPR && col0 + cl < p.N) {
      if (p.colsum_part) {
        float* dst = p.colsum_part + (size_t)(row0 / (TM * 32)) * p.N + col0 + cl;
        *(f32x4_t*)dst = (f32x4_t){cs[0], cs[1], cs[2], cs[3]};
        *(f32x4_t*)(dst + 4) = (f32x4_t){cs[4], cs[5], cs[6], cs[7]};
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) unsafeAtomicAdd(&p.colsum[col0 + cl + q], cs[q]);
      }
    }
  }
}

template <int TM, int TN, int WM, int WN, int EPI, int PR = 0, bool TAB = false>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16_t (&acc)[TM][TN], char* lds, int wave, int lane,
                                              int row0, int col0, const char* tab = nullptr) {
  // row0/col0: global coordinates of this wave's tile origin
  static_assert(WM == TM * 32 && WN == TN * 32 && TM <= 4, "wave tile geometry");
  if (EPI == 2) {
    // deterministic weight-gradient epilogue: this block's partial tile goes to its own slab of the caller's workspace in
    // accumulator order (full 1-KB wave stores); splitk_reduce_kernel adds the slabs in split order into C
    const int NWV = (int)(blockDim.x >> 6);
    const int slab = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n * p.splitk);      // = split * tiles + tile
    float* dst = p.ws + (size_t)slab * (size_t)(NWV * TM * TN * 1024) + (size_t)wave * (TM * TN * 1024) + lane * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *(f32x4_t*)(dst + ((i * TN + j) * 4 + q) * 256) = (f32x4_t){acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
    return;
  } else if (EPI == 1) {
    // weight-gradient epilogue: fp32 accumulate into C (atomics; C is pre-zeroed or holds the running sum)
    float* C = (float*)p.C;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        int n = col0 + j * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int m = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (m < p.M && n < p.N) unsafeAtomicAdd(&C[(size_t)m * p.ldc + n], acc[i][j][r]);
        }
      }
    return;
  } else {
    // The LDS pipe executes one wave's operations in order, so a wave-private patch needs no wait between writing it and
    // reading it back, and a patch can be rewritten as soon as the reads of its previous contents have been ISSUED.
    constexpr int LDP = WN + 4;
    char* wave_lds = lds + wave * epi_wave_lds<WN>();
    const bool fast_ok = !p.out_f32 && p.wide_ok && (p.N % 8 == 0) && !(p.act == 3 && p.res);
    if (fast_ok) {
      const bool extra = p.res || p.act == 3 || p.colsum || p.drop_thresh;
      if (!extra) {
        if (p.act == 0) epi_fast<TM, TN, WN, 0>(p, acc, wave_lds, lane, row0, col0);
        else if (p.act == 1) epi_fast<TM, TN, WN, 1, TAB>(p, acc, wave_lds, lane, row0, col0, tab);
        else epi_fast<TM, TN, WN, 2>(p, acc, wave_lds, lane, row0, col0);
      } else {
        if (p.act == 3) epi_fast_ext<TM, TN, WN, 3>(p, acc, wave_lds, lane, row0, col0);
        else if (p.act == 0) epi_fast_ext<TM, TN, WN, 0>(p, acc, wave_lds, lane, row0, col0);
        else if (p.act == 1) epi_fast_ext<TM, TN, WN, 1>(p, acc, wave_lds, lane, row0, col0);
        else epi_fast_ext<TM, TN, WN, 2>(p, acc, wave_lds, lane, row0, col0);
      }
      return;
    }
    // general path, per 32-row block i: wait for the second operand of block i (LDS-DMA issued two blocks earlier: global ->
    // LDS, no registers, lane l's 16 bytes land at buffer + 16 l, exactly its row strip), request it and the rows of patch i
    // from LDS, start the DMA of block i+2 into the buffer just read, queue the patch writes of block i+1, then do the
    // arithmetic and the global stores of block i.
    constexpr int LPR = WN / 8;          // lanes per row
    constexpr int RPI = 64 / LPR;        // rows per iteration = rows per DMA instruction
    constexpr int IT = 32 / RPI;
    float* patch = (float*)wave_lds;
    char* opbuf = wave_lds + 32 * LDP * 4;
    constexpr int OPB = 32 * WN * 2;     // one buffer: 32 rows of the operand
    EpiLane e;
    epi_setup<WN>(e, p, lane, col0);
    const bf16_t* prim_ptr = (p.act == 3) ? p.aux : p.res;
    const int prim_ld = (p.act == 3) ? p.ldaux : p.ldres;
    const int prim_period = (p.act == 3) ? 0 : p.res_period;
    const bool staged = prim_ptr != nullptr && p.wide_ok && (p.N % 8 == 0);
    __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)prim_ptr, 0, 0xFFFFFFF0u, 0x00020000);
    const int rl = lane / LPR;
    auto dma_block = [&](int i) __attribute__((always_inline)) {
#pragma unroll
      for (int itr = 0; itr < IT; ++itr) {
        const int m = row0 + i * 32 + itr * RPI + rl;
        const int mr = prim_period ? (m % prim_period) : m;
        uint32_t off = (uint32_t)(((size_t)mr * (size_t)prim_ld + (size_t)e.n) * 2);
        if (m >= p.M || !e.ncol_ok) off = 0xFFFFFFF0u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, AVT_LDS_PTR(opbuf + (i & 1) * OPB + itr * 1024), 16, off, 0, 0, AVT_LDP_AUX);
      }
    };
    if (staged) { dma_block(0); if (TM > 1) dma_block(1); }
    epi_write_block<TN, WN>(patch, epi_take<TM, TN, 0>(acc), lane);
#pragma unroll 1
    for (int i = 0; i < TM; ++i) {
      // outstanding VMEM operations issued after DMA(i), counting one store per row strip (more stores only make the wait
      // stricter than needed): i = 0: DMA(1);  i = 1: DMA(2), stores(0);  i >= 2: stores(i-2), DMA(i+1), stores(i-1)
      if (staged) {
        switch (i) {
          case 0: if (TM > 1) wait_vmcnt<IT>(); else wait_vmcnt<0>(); break;
          case 1: if (TM > 2) wait_vmcnt<2 * IT>(); else wait_vmcnt<IT>(); break;
          case 2: if (TM > 3) wait_vmcnt<3 * IT>(); else wait_vmcnt<2 * IT>(); break;
          default: wait_vmcnt<2 * IT>(); break;
        }
      }
      float rows[IT][8];
      EpiStrip prim[IT];
#pragma unroll
      for (int itr = 0; itr < IT; ++itr) {
        const float* src = patch + (itr * RPI + rl) * LDP + e.cl;
        f32x4_t lo = *(const f32x4_t*)src, hi = *(const f32x4_t*)(src + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { rows[itr][k] = lo[k]; rows[itr][4 + k] = hi[k]; }
        prim[itr].w = (u32x4_t){0u, 0u, 0u, 0u};
        if (staged) prim[itr].w = *(const u32x4_t*)(opbuf + (i & 1) * OPB + itr * 1024 + lane * 16);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (staged && i + 2 < TM) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the buffer's previous contents are in registers
        dma_block(i + 2);
      }
      if (i + 1 < TM) epi_write_block_i<TM, TN, WN>(patch, acc, i + 1, lane);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int itr = 0; itr < IT; ++itr)
        epi_row(e, p, rows[itr], row0 + i * 32 + itr * RPI + rl, prim[itr], staged);
    }
    epi_flush_colsum<WN>(e, p, lane, row0 / WM);
  }
}

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
#ifdef AVT_LAB
  if (MINW >= 2) stagger_slot(p.stagger, bid, 512); else
#endif
  stagger_start(AVT_STAGGER(p), bid);

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

  long long t_start = 0, t_loop = 0;
  if (AVT_DBG(p)) t_start = __builtin_readcyclecounter();
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
  if (AVT_DBG(p)) t_loop = __builtin_readcyclecounter();

  gemm_epilogue<TM, TN, WM, WN, EPI>(p, acc, lds, wave, lane, tm0 + wm * WM, tn0 + wn * WN);
#ifdef AVT_LAB
  if (p.dbg && tid == 0) {
    const long long t_math = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t_end = __builtin_readcyclecounter();
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long* d_ = p.dbg + (size_t)bid * 16;            // same record stride as the 8-phase kernel (two groups x 8)
    d_[0] = t_start; d_[1] = t_loop; d_[2] = t_math; d_[3] = t_end; d_[4] = hw; d_[5] = xcc; d_[6] = nk; d_[7] = bid;
  }
#endif
}

// Second pass of the deterministic split-K accumulate: one wave per 1-KB chunk (producing wave w, block (i, j), register
// quad q) mirrors the producer's register layout, sums the chunk over the splits IN ORDER and adds the result to C (every C
// element has exactly one owner, so plain read-modify-write; C keeps the running sum of earlier GEMMs into the same gradient).
template <int TM, int TN, int WGM, int WGN>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, int ldc, int M, int N,
                                                            int tiles_n, int ntile, int splitk) {
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
      if (m0 + k < M) C[(size_t)(m0 + k) * ldc + n] += v[k];
  }
}
template <int TM, int TN, int WGM, int WGN>
int launch_reduce(const GemmParams& p, hipStream_t s) {
  const int ntile = p.tiles_m * p.tiles_n;
  constexpr int CHUNKS = WGM * WGN * TM * TN * 4;
  hipLaunchKernelGGL((splitk_reduce_kernel<TM, TN, WGM, WGN>), dim3((ntile * CHUNKS + 3) / 4), dim3(256), 0, s, (const float*)p.ws, (float*)p.C, p.ldc,
                     p.M, p.N, p.tiles_n, ntile, p.splitk);
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

#ifdef AVT_LAB   // lab-only kernel variants (negative results kept for A/B in tools/): not part of libavt_hip.so
#include "../../tools/lab/gemm_lab_variants.inc"
#endif

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
#ifdef AVT_LAB
  long long t8_start = 0, t8_loop = 0;
  if (p.dbg) t8_start = __builtin_readcyclecounter();
#endif

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
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, AVT_LDS_PTR(d + j * JSTEP), 16, offA[h][j] + adv, 0, 0, AVT_LDA_AUX);
  };
  auto stage_b = [&](int h, int tile) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t r = (tile < nk) ? rb : rb_null;
    char* d = dstB + ((h ? 2 : 1) * 2 + (tile & 1)) * HALF;
    const uint32_t adv = (uint32_t)tile * kstepB;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, AVT_LDS_PTR(d + j * JSTEP), 16, offB[h][j] + adv, 0, 0, AVT_LDB_AUX);
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
#ifdef AVT_LAB
  if (p.dbg) t8_loop = __builtin_readcyclecounter();
#endif
  gemm_epilogue<TM, TN, WM, WN, EPI, 0, GTAB>(p, acc, lds, wave, lane_e, m0_e, n0_e, smem8);
#ifdef AVT_LAB
  if (p.dbg && (tid == 0 || tid == 256)) {            // first wave of each group: start, end of K loop, arithmetic done, stores drained, placement
    const long long t_math = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t_end = __builtin_readcyclecounter();
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long* d_ = p.dbg + ((size_t)bid * 2 + grp) * 8;
    d_[0] = t8_start; d_[1] = t8_loop; d_[2] = t_math; d_[3] = t_end; d_[4] = hw; d_[5] = xcc; d_[6] = nk; d_[7] = bid;
  }
#endif
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

int dispatch_8p(GemmParams& p, int epi, int a_kmajor, int b_kmajor, int splitk, hipStream_t s) {
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
#ifdef AVT_LAB
  { static const char* e = getenv("AVT_GEMM_STRIP"); if (e) p.strip_w = (epi == 0 && atoi(e) < p.tiles_n) ? atoi(e) : 0; }
#endif
  if (epi == 2) {
    if (a_kmajor || b_kmajor) { avt_set_error("avt_gemm_accum_bf16: operands must both be stored reduction-index-major"); return -1; }
    if ((size_t)p.tiles_m * p.tiles_n * splitk * 65536 * 4 > p.ws_bytes) { avt_set_error("avt_gemm_accum_bf16: workspace too small (%zu bytes needed)", (size_t)p.tiles_m * p.tiles_n * splitk * 65536 * 4); return -2; }
    int rc = launch_8p<false, false, 2>(p, s);
    return rc ? rc : launch_reduce<4, 2, 2, 4>(p, s);
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

// ---- 4-wave kernel: 256x128x32 tile, TWO workgroups per CU, so that one's epilogue runs under the other's K loop ------------
// Measured on gfx950 (tools/lab/coissue_lab.hip, coissue2_lab.hip, valu_rate_lab.hip; profiles/r03_issue_rules.txt):
//   * a wave's vector-ALU instructions never overlap its OWN MFMAs (8 MFMA + k packed FMAs = 8 x 32 + 4.7 k cycles), but the
//     vector ALU of ANOTHER wave on the same SIMD does run under them (the MFMA wave keeps 32.1 cycles per MFMA; the other
//     wave's packed FMAs slow from one per 9.5 to one per 17.5 cycles);
//   * a wave's own ds_read_b128 and LDS-DMA issue DO overlap its MFMAs (8 MFMA + 6 reads = 257 cycles, + 2 LDS-DMA = 281).
// In the 8-phase kernel all eight waves of the CU reach the epilogue together: for a GELU (+ GELU') tile that is 22 k cycles of
// vector-ALU work next to a 32 k-cycle K loop with the matrix pipe idle (tools/gemm_timeline.py).  Here a workgroup is four
// waves (one per SIMD, wave tile 128x64 as in the 8-phase kernel, so the epilogue code is shared) on a 256x128 tile with a
// 72-KB ring (3 stages of 32 k), and two workgroups share the CU: while one converts and stores its tile, the other owns the
// matrix pipe.  The K loop therefore has to keep the pipe busy from ONE wave per SIMD: it contains no vector-ALU instruction at
// all (per-lane offsets are computed once, the K advance goes through the scalar offset of the buffer instruction, stages past
// the end read through a zero-length descriptor), fragments of the next k-step are requested before the MFMAs of the current
// one, and the six LDS-DMA instructions of a stage are spread between the MFMAs.
//   stage s (slot s % 3):  wait own DMA of stage s (vmcnt 6), lgkmcnt(0), s_barrier
//                          read fragments (s, k-step 0) | 8 MFMA of (s-1, k-step 1) with 3 DMA of stage s+2 between them
//                          read fragments (s, k-step 1) | 8 MFMA of (s,   k-step 0) with 3 DMA of stage s+2 between them
// The barrier of stage s also tells that every wave has its (s-1, k-step 1) fragments in registers, so slot (s+2) % 3 = (s-1) % 3 is free.
template <bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(256, 2) void gemm_4w_kernel(GemmParams p) {
  static_assert(A_KMAJOR && B_KMAJOR, "4-wave kernel: k-major operands only");
  constexpr int BM = 256, BN = 128, BK = 32, WM = 128, WN = 64, TM = 4, TN = 2;
  constexpr int A_ST = BM * BK * 2, STAGE = (BM + BN) * BK * 2;      // 16 KB + 8 KB
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntile = p.tiles_m * p.tiles_n;
  const int t_ = xcd_remap(blockIdx.x, ntile);
  const int tm0 = (t_ / p.tiles_n) * BM;
  const int tn0 = (t_ % p.tiles_n) * BN;
  const int nk = p.K / BK;                                           // host-checked: K % 32 == 0
#ifdef AVT_LAB
  long long t4_start = 0, t4_loop = 0;
  if (p.dbg) t4_start = __builtin_readcyclecounter();
#endif

  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.b_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t ra_null = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0, 0x00020000);
  __amdgpu_buffer_rsrc_t rb_null = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, 0, 0x00020000);

  // LDS-DMA: one instruction = 16 rows x 64 B; wave w stages rows j*64 + w*16 .. +15 of A (j = 0..3) and of B (j = 0, 1).
  // Rows past the matrix edge lie past the descriptor's range and arrive as zeros.
  uint32_t voffA[4], voffB[2];
  {
    const int rl = wave * 16 + (lane >> 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = j * 64 + rl;
      const int c = (lane & 3) ^ kmajor_swz<BK>(r);
      voffA[j] = (uint32_t)(((size_t)(tm0 + r) * (size_t)p.lda + (size_t)c * 8) * 2);
      if (j < 2) voffB[j] = (uint32_t)(((size_t)(tn0 + r) * (size_t)p.ldb + (size_t)c * 8) * 2);
    }
  }
  char* const dst = lds + wave * 16 * (BK * 2);                       // wave-uniform part of the destination
  auto dma = [&](int q, int st) __attribute__((always_inline)) {      // q = 0..3: A rows, 4..5: B rows; st = stage to fetch
    const uint32_t kadv = (uint32_t)st * (BK * 2);                    // scalar: k advance in bytes
    char* d = dst + (st % 3) * STAGE;
    if (q < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(st < nk ? ra : ra_null, AVT_LDS_PTR(d + q * 64 * (BK * 2)), 16, voffA[q], kadv, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(st < nk ? rb : rb_null, AVT_LDS_PTR(d + A_ST + (q - 4) * 64 * (BK * 2)), 16, voffB[q - 4], kadv, 0, 0);
  };
  // fragment addresses: lane (i = l & 31, h = l >> 5) reads row (block*32 + i), 16-B chunk (ks*2 + h) ^ ((i >> 2) & 3); the
  // row-block, the k-step (chunk ^ 2) and the slot are immediates of the ds_read
  const int ch0 = (lane >> 5) ^ (((lane & 31) >> 2) & 3);
  const char* const fa0 = lds + (wm * 128 + (lane & 31)) * (BK * 2) + ch0 * 16;
  const char* const fa1 = lds + (wm * 128 + (lane & 31)) * (BK * 2) + (ch0 ^ 2) * 16;
  const char* const fb0 = lds + A_ST + (wn * 64 + (lane & 31)) * (BK * 2) + ch0 * 16;
  const char* const fb1 = lds + A_ST + (wn * 64 + (lane & 31)) * (BK * 2) + (ch0 ^ 2) * 16;
  bf16x8_t af[2][TM], bfr[2][TN];
  auto rdf = [&](int ks, int slot) __attribute__((always_inline)) {
    const char* a = (ks ? fa1 : fa0) + slot * STAGE;
    const char* b = (ks ? fb1 : fb0) + slot * STAGE;
#pragma unroll
    for (int i = 0; i < TM; ++i) af[ks][i] = *(const bf16x8_t*)(a + i * 32 * (BK * 2));
#pragma unroll
    for (int j = 0; j < TN; ++j) bfr[ks][j] = *(const bf16x8_t*)(b + j * 32 * (BK * 2));
  };

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define W4_PIN() __builtin_amdgcn_sched_barrier(0)
  // 8 MFMAs of k-step KS with the DMA instructions Q0 .. Q0+2 of stage ST after the 1st, 3rd and 5th of them
#define W4_MFMA(KS, Q0, ST)                                                                              \
  do {                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                       \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                   \
        acc[i][j] = mma<0>(af[KS][i], bfr[KS][j], acc[i][j]);                                            \
        if (j == 0 && i < 3) { W4_PIN(); dma((Q0) + i, (ST)); W4_PIN(); }                               \
      }                                                                                                  \
  } while (0)
  // one stage: S = stage index (runtime), SLOT = S % 3 (compile time)
#define W4_STAGE(S, SLOT, FIRST)                                                                         \
  do {                                                                                                   \
    wait_vmcnt<6>();                                                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
    W4_PIN(); asm volatile("s_barrier" ::: "memory"); W4_PIN();                                          \
    rdf(0, SLOT); W4_PIN();                                                                              \
    if (!(FIRST)) { W4_MFMA(1, 0, (S) + 2); } else { dma(0, (S) + 2); dma(1, (S) + 2); dma(2, (S) + 2); } \
    W4_PIN(); rdf(1, SLOT); W4_PIN();                                                                    \
    asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");       /* the six k-step-0 fragments are older than the six just requested */ \
    W4_PIN(); W4_MFMA(0, 3, (S) + 2); W4_PIN();                                                          \
  } while (0)

  // prologue: stages 0 and 1
#pragma unroll
  for (int q = 0; q < 6; ++q) dma(q, 0);
#pragma unroll
  for (int q = 0; q < 6; ++q) dma(q, 1);
  W4_STAGE(0, 0, true);
  int s = 1;
  for (; s + 2 < nk; s += 3) {            // slots 1, 2, 0
    W4_STAGE(s, 1, false);
    W4_STAGE(s + 1, 2, false);
    W4_STAGE(s + 2, 0, false);
  }
  if (s < nk) { W4_STAGE(s, 1, false); ++s; }
  if (s < nk) { W4_STAGE(s, 2, false); ++s; }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  W4_PIN();
#pragma unroll
  for (int i = 0; i < TM; ++i)            // k-step 1 of the last stage
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = mma<0>(af[1][i], bfr[1][j], acc[i][j]);
  wait_vmcnt<0>();                        // the zero-length fetches past the end
  W4_PIN(); asm volatile("s_barrier" ::: "memory"); W4_PIN();       // every wave is done with the ring: the epilogue reuses it
#undef W4_STAGE
#undef W4_MFMA
#undef W4_PIN
  int lane_e = lane, m0_e = tm0 + wm * WM, n0_e = tn0 + wn * WN;
  asm volatile("" : "+v"(lane_e), "+s"(m0_e), "+s"(n0_e));
#ifdef AVT_LAB
  if (p.dbg) t4_loop = __builtin_readcyclecounter();
#endif
  gemm_epilogue<TM, TN, WM, WN, 0>(p, acc, lds, wave, lane_e, m0_e, n0_e);
#ifdef AVT_LAB
  if (p.dbg && tid == 0) {
    const long long t_math = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t_end = __builtin_readcyclecounter();
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long* d_ = p.dbg + (size_t)blockIdx.x * 16;
    d_[0] = t4_start; d_[1] = t4_loop; d_[2] = t_math; d_[3] = t_end; d_[4] = hw; d_[5] = xcc; d_[6] = nk; d_[7] = blockIdx.x;
  }
#endif
}

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
template <int EPI>
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
  const int lb = xcd_remap(blockIdx.x, ntile * p.splitk);
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
    const uint32_t a_ = ((F) == 0) ? adA[(SLOT) >> 1][0] : ((F) <= 4 ? adB[(SLOT) >> 1][(F) - 1] : adA[(SLOT) >> 1][(F) - 4]);  \
    const u32x2_t lo_ = ds_read_tr_na<imm_>(a_), hi_ = ds_read_tr_na<imm_ + 4 * ROWB>(a_);   /* inline asm: no compiler-placed vmcnt(0) */ \
    const bf16x8_t v_ = tr_join(lo_, hi_);                                                                            \
    if ((F) == 0) af[NB][0] = v_; else if ((F) <= 4) bfr[NB][(F) - 1] = v_; else af[NB][(F) - 4] = v_;                \
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
  gemm_epilogue<TM, TN, WM, WN, EPI>(p, acc, lds, wave, lane, tm0 + wm * WM, tn0 + wn * WN);
}

int dispatch_w4(GemmParams& p, int a_kmajor, int b_kmajor, int splitk, hipStream_t s) {
  if (a_kmajor || b_kmajor) { avt_set_error("avt_gemm_accum_bf16: operands must both be stored reduction-index-major"); return -1; }
  p.tiles_m = (p.M + 255) / 256; p.tiles_n = (p.N + 255) / 256;
  const int nk64 = (p.K + 63) / 64;
  if (splitk <= 0) splitk = pick_splitk((long)p.tiles_m * p.tiles_n, nk64, 1, 8);      // same choice as the 8-phase kernel: the workspace query mirrors it
  if (splitk > nk64) splitk = nk64;
  p.splitk = splitk;
  if ((size_t)p.tiles_m * p.tiles_n * splitk * 65536 * 4 > p.ws_bytes) { avt_set_error("avt_gemm_accum_bf16: workspace too small (%zu bytes needed)", (size_t)p.tiles_m * p.tiles_n * splitk * 65536 * 4); return -2; }
  constexpr int smem = 5 * 2 * 32 * 512;               // 160 KB
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)gemm_w4_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr_set = true; }
  hipLaunchKernelGGL((gemm_w4_kernel<2>), dim3(p.tiles_m * p.tiles_n * splitk), dim3(256), smem, s, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { avt_set_error("avt_gemm: launch failed: %s", hipGetErrorString(e)); return (int)e; }
  return launch_reduce<4, 4, 2, 2>(p, s);
}

int dispatch_4w(GemmParams& p, int epi, int a_kmajor, int b_kmajor, hipStream_t s) {
  if (epi != 0 || !a_kmajor || !b_kmajor || p.K % 32 != 0) { avt_set_error("avt_gemm_bf16: tile 2564 (4-wave, two workgroups per CU) needs the activation epilogue, k-major operands and K %% 32 == 0"); return -1; }
  p.tiles_m = (p.M + 255) / 256; p.tiles_n = (p.N + 127) / 128; p.splitk = 1;
  constexpr int ring = 3 * (256 + 128) * 32 * 2, patch = 4 * epi_wave_lds<64>();
  constexpr int smem = ring > patch ? ring : patch;
  static_assert(2 * smem <= 160 * 1024, "two workgroups must fit one CU's LDS");
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)gemm_4w_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr_set = true; }
  hipLaunchKernelGGL((gemm_4w_kernel<true, true>), dim3(p.tiles_m * p.tiles_n), dim3(256), smem, s, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { avt_set_error("avt_gemm: launch failed: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

}  // namespace

static int gemm_dispatch(GemmParams& p, int bm, int epi, int a_kmajor, int b_kmajor, int splitk, int K, hipStream_t s);
static int gemm_impl(const void* A, int a_kmajor, int lda, const void* B, int b_kmajor, int ldb,
                     void* C, int ldc, int M, int N, int K,
                     const float* bias, int act, const void* aux, int ldaux,
                     void* C2, int ldc2, const void* res, int ldres, int res_period,
                     float drop_p, uint64_t drop_seed, float* colsum,
                     int out_mode, int splitk, int tile, void* ws, size_t ws_bytes, float* part, size_t part_bytes, void* stream) {
  AVT_CHECK(A && B && C, "avt_gemm_bf16: null operand");
  AVT_CHECK(M > 0 && N > 0 && K > 0, "avt_gemm_bf16: bad dims M=%d N=%d K=%d", M, N, K);
  AVT_CHECK(aligned16(A) && aligned16(B) && aligned16(C), "avt_gemm_bf16: operands must be 16-byte aligned");
  AVT_CHECK(lda % 8 == 0 && ldb % 8 == 0, "avt_gemm_bf16: lda/ldb must be multiples of 8 (got %d, %d)", lda, ldb);
  AVT_CHECK(K % 8 == 0 || (!a_kmajor && !b_kmajor), "avt_gemm_bf16: K must be a multiple of 8 for k-major operands (K=%d)", K);
  AVT_CHECK(out_mode >= 0 && out_mode <= 3, "avt_gemm_bf16: out_mode must be 0 (bf16), 1 (fp32) or 2 (fp32 atomic accumulate)");
  AVT_CHECK(out_mode != 3 || (ws && aligned16(ws)), "avt_gemm_accum_bf16: needs a 16-byte aligned workspace");
  AVT_CHECK(act >= 0 && act <= 3, "avt_gemm_bf16: bad act %d", act);
  AVT_CHECK(act < 3 || aux, "avt_gemm_bf16: act %d needs aux", act);
  AVT_CHECK(drop_p >= 0.f && drop_p < 1.f, "avt_gemm_bf16: bad dropout p");
  GemmParams p{};
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.C2 = (bf16_t*)C2; p.ws = (float*)ws; p.ws_bytes = ws_bytes;
  p.bias = bias; p.res = (const bf16_t*)res; p.aux = (const bf16_t*)aux; p.colsum = colsum;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldc2 = ldc2; p.ldres = ldres; p.ldaux = ldaux;
  p.res_period = res_period; p.act = act; p.out_f32 = (out_mode == 1);
  p.drop_thresh = drop_threshold(drop_p); p.drop_scale = 1.0f / (1.0f - drop_p); p.drop_seed = drop_seed;
#ifdef AVT_LAB
  { static const char* e2 = getenv("AVT_GEMM_STAGGER"); p.stagger = e2 ? atoi(e2) : 0; }
  { static const char* e = getenv("AVT_GEMM_DBG_PTR"); p.dbg = e ? (long long*)strtoull(e, nullptr, 0) : nullptr; }
#endif
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
    if (epi == 0) bm = (t256 >= 200) ? 256 : (t128 >= 192 ? 128 : 64);   // epi 1 | 2: below
    else {
      long sk = ((K + 63) / 64) / 4; if (sk < 1) sk = 1; if (sk > 64) sk = 64;
      bm = (t256 * sk >= 256 && t256 < 4096) ? 256 : 128;
    }
  }
  if (tile == 0 && bm == 256 && (K % 64 == 0 || (!a_kmajor && !b_kmajor))) bm = 808;       // default big-tile kernel: the 8-phase schedule
  if (tile == 0 && bm == 808 && epi == 2) bm = 2565;                                          // weight gradients: 4 waves of 128x128 (+2-3 % over the 8-phase kernel)
  if (tile == 0 && bm == 64 && epi == 0 && a_kmajor && b_kmajor) bm = 643;                   // all-k-major small outputs: 3-deep ring (+15-25 % on the head's data gradients)
  int nslots = 0;
  if (colsum && part) {
    // one partial row per wave row of the grid: every tile shape here has two wave rows per tile
    const int BM = (bm == 64 || bm == 643) ? 64 : (bm == 128 ? 128 : 256);
    nslots = ((M + BM - 1) / BM) * 2;
    AVT_CHECK(aligned16(part) && part_bytes >= (size_t)nslots * N * 4, "avt_gemm_bf16: partials workspace too small or misaligned (%zu bytes needed)", (size_t)nslots * N * 4);
    p.colsum_part = part;
  }
  int rc = gemm_dispatch(p, bm, epi, a_kmajor, b_kmajor, splitk, K, s);
  if (rc == 0 && nslots) { float* outs[1] = {colsum}; rc = avt_reduce_partials(part, nslots, N, outs, 1, s); }
  return rc;
}

extern "C" size_t avt_gemm_colsum_workspace_bytes(int M, int N, int tile) {
  // mirrors gemm_impl's automatic tile choice for the activation epilogue; two wave rows per tile
  int BM = (tile == 64 || tile == 643) ? 64 : (tile == 128 ? 128 : 256);
  if (tile == 0) {
    const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256), t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    BM = (t256 >= 200) ? 256 : (t128 >= 192 ? 128 : 64);
  }
  return (size_t)(((M + BM - 1) / BM) * 2) * (size_t)N * 4;
}

static int gemm_dispatch(GemmParams& p, int bm, int epi, int a_kmajor, int b_kmajor, int splitk, int K, hipStream_t s) {
  switch (bm) {
    case 64:  return dispatch_epi<64, 64, 2, 2, 64, 2>(p, epi, a_kmajor, b_kmajor, splitk, s);
    case 128: return dispatch_epi<128, 128, 2, 2, 64, 2>(p, epi, a_kmajor, b_kmajor, splitk, s);
    case 643: return dispatch_epi<64, 64, 2, 2, 64, 3>(p, epi, a_kmajor, b_kmajor, splitk, s);     // 3-deep ring
    case 256:                                                                                     // one barrier per K tile, dribbled LDS-DMA issued by 4 loader waves
      if (K % 64 == 0 || (!a_kmajor && !b_kmajor)) return dispatch_epi<256, 256, 2, 4, 64, 2, true, 0, 1, 4>(p, epi, a_kmajor, b_kmajor, splitk, s);
      return dispatch_epi<256, 256, 2, 4, 64, 2, true>(p, epi, a_kmajor, b_kmajor, splitk, s);
    case 2568: return dispatch_epi<256, 256, 2, 4, 64, 2, true>(p, epi, a_kmajor, b_kmajor, splitk, s);       // all 8 waves issue LDS-DMA
    case 2564: return dispatch_4w(p, epi, a_kmajor, b_kmajor, s);
    case 2565:                                                                                    // 4-wave weight-gradient kernel (128x128 wave tiles, AGPR accumulators)
      if (epi != 2) { avt_set_error("avt_gemm: tile 2565 is the deterministic weight-gradient kernel (avt_gemm_accum_bf16 only)"); return -1; }
      return dispatch_w4(p, a_kmajor, b_kmajor, splitk, s);                                  // 4 waves, 256x128x32, two workgroups per CU
    case 808:                                                                                     // 8-phase schedule (needs K % 64 == 0 for k-major operands)
      if (K % 64 == 0 || (!a_kmajor && !b_kmajor)) return dispatch_8p(p, epi, a_kmajor, b_kmajor, splitk, s);
      return dispatch_epi<256, 256, 2, 4, 64, 2, true>(p, epi, a_kmajor, b_kmajor, splitk, s);
#ifdef AVT_LAB
    // two independent 4-wave workgroups per CU (<= 80 KB LDS, <= 256 registers each), optionally started half a tile apart
    // (AVT_GEMM_STAGGER cycles): while one is in its epilogue the other owns the matrix pipe
    case 2563: return dispatch_epi<256, 128, 2, 2, 32, 3, true, 0, 2>(p, epi, a_kmajor, b_kmajor, splitk, s);
    case 2562: return dispatch_epi<256, 128, 2, 2, 32, 3, false, 0, 2>(p, epi, a_kmajor, b_kmajor, splitk, s);
    case 1283: return dispatch_epi<128, 256, 1, 4, 32, 3, true, 0, 2>(p, epi, a_kmajor, b_kmajor, splitk, s);
    case 258: return dispatch_deepa(p, epi, a_kmajor, b_kmajor, splitk, s);                          // A ring 3 deep, B ring 2 deep
    case 512: return dispatch_pp(p, epi, a_kmajor, b_kmajor, splitk, s);                            // ping-pong 256x256
#endif
    default: break;
  }
  avt_set_error("avt_gemm_bf16: tile must be 0 (choose), 64, 128, 643, 256 / 2568 (one barrier per K tile) or 808 (8-phase) (got %d)", bm);
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

extern "C" int avt_gemm_accum_bf16(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K,
                                   int splitk, int tile, void* workspace, size_t workspace_bytes, void* stream) {
  return gemm_impl(A, 0, lda, B, 0, ldb, C, ldc, M, N, K, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, 0.f, 0, nullptr,
                   3, splitk, tile, workspace, workspace_bytes, nullptr, 0, stream);
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
