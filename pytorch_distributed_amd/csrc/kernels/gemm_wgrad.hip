// Weight-gradient GEMM for gfx950: C[I, J] = A[M, I]^T . B[M, J] with the LONG dimension M
// (tokens) as the contraction, plus the column sums of A (the bias gradient) for free.
//
// Why a dedicated kernel. In dW = dY^T . X both operands are stored with the contraction
// index as their ROW index, and the output (I x J = a weight matrix) has far fewer
// 256 x 256 tiles than the chip has CUs (25-100 for GPT-2 Large), so the shape needs
// split-K and transposed operand access at once. Here:
//   * a workgroup owns one 256 x 256 output tile and one slice of M, so tiles x slices ~ 2 x 256
//     workgroups fill the 256 CUs twice over. Two kernels share everything below: the default
//     splits the tile over 4 waves of 128 x 128 (accumulators in the AGPR file, LDS reads and
//     global loads issued in the shadow of the MFMAs by an explicit schedule), the older one over
//     8 waves of 128 x 64 (PDA_WGRAD_WAVES=8; 11-13 % slower on GPT-2 shapes);
//   * both operand tiles are staged as they lie in memory — 64 rows of M x 256 columns,
//     fully coalesced 512-byte rows — and the MFMA fragments are read TRANSPOSED out of LDS
//     with ds_read_b64_tr_b16, so no transpose pass and no strided global access exists;
//     one XOR swizzle keeps those reads bank-conflict free;
//   * tiles travel global -> LDS directly (global_load_lds, 1 KiB per wave-instruction, no
//     VGPR staging and no ds_write pass — LDS store bandwidth is a third of its load
//     bandwidth and was the limiter of a register-staged version); the swizzle is applied to
//     the per-lane SOURCE address because the LDS destination of such a load is linear;
//     a 4-stage LDS ring keeps three stages (96 KiB per CU) in flight behind the one being
//     read — with one stage in flight the kernel ran at the loaded memory latency
//     (~2 us per 64 KiB) instead of the MFMA rate; waits are counted (vmcnt(8)), barriers raw;
//   * workgroups whose tile is in the first column block also add up their A tile's columns
//     (one 16-byte LDS read per thread and row): that is the bias gradient, which otherwise
//     costs a separate full read of dY;
//   * consecutive task ids share an M slice and workgroup ids are dealt to XCDs in
//     contiguous task ranges, so the CUs of one XCD stream the same rows of A and B
//     through their shared L2 (affinity only, never correctness).
// Partial tiles go to an fp32 workspace [slices][I][J]; wgrad_reduce folds the slices,
// converts, and optionally accumulates into an existing gradient.
#include "attention_common.h"
#include "pda_kernels.h"
#include <stdlib.h>

#include <type_traits>

namespace pda {
namespace {

using namespace attn;

constexpr int kThreads = 512;
constexpr int kTile = 256;     // output tile edge
constexpr int kBM = 32;        // contraction rows per pipeline stage
constexpr int kStages = 4;     // LDS ring: 3 stages (96 KiB per CU) in flight behind the one being read
constexpr int kRowBytes = kTile * 2;
constexpr int kTileBytes = kBM * kRowBytes;  // 16 KiB per operand tile
constexpr int kStageBytes = 2 * kTileBytes;
constexpr int kLoadsPerStage = 4;            // global_load_lds instructions per wave and stage

// Byte offset of 16-byte chunk `chunk` (0..31) of row `row` in a [64][256] bf16 LDS tile.
// A transposed read touches rows 4a..4a+3 x one 64-byte segment per 32 lanes; XOR-ing the
// segment index with row & 3 spreads the four rows over the four segments of a 256-byte
// bank row.
__device__ __forceinline__ int wg_tile_off(int row, int chunk) {
  return row * kRowBytes + ((chunk ^ ((row & 3) << 2)) << 4);
}

// LDS reads are issued through inline asm on purpose. While global_load_lds operations are
// in flight the compiler cannot prove that an ordinary LDS read does not alias their
// destination and drains the whole queue (s_waitcnt vmcnt(0)) in front of it, which would
// serialise the ring. The asm reads are invisible to that bookkeeping; their completion is
// awaited explicitly (counted s_waitcnt lgkmcnt + a scheduling fence) before the MFMAs.
__device__ __forceinline__ unsigned lds_addr(const char* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

template <int kOff>
__device__ __forceinline__ s16x4 tr_read(unsigned addr) {
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(kOff));
  return v;
}

template <int kOff>
__device__ __forceinline__ u32x4 b128_read(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(kOff));
  return v;
}

// Per-lane LDS byte offset (inside one operand tile) of the fragment rows for k-step 0 and
// 32-column block `cblock`: lane (r, h) will receive
// tile[16*kstep + 8*(j>>2) + 4*h + (j&3)][32*cblock + r], j = 0..7, from two transposed reads
// at this offset + 8192*kstep and + 4096 (rows +8; the swizzle only looks at row & 3).
__device__ __forceinline__ unsigned frag_base(int cblock, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const int q = i >> 2, p = i & 3;
  const int h = g >> 1;
  const int col = cblock * 32 + 16 * (g & 1) + 4 * p;
  return (unsigned)(wg_tile_off(4 * h + q, col >> 3) + (col & 7) * 2);
}

__device__ __forceinline__ bf16x8 join(s16x4 lo, s16x4 hi) {
  s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// Direct global -> LDS copy of 16 rows (one MFMA k-step) of a [32][256] bf16 tile by the 8
// waves of the workgroup: wave w moves rows 16*half + 2w and + 2w + 1 (1 KiB) in one
// instruction. The LDS destination is "wave base + lane * 16" (hardware rule), so lane p lands
// on (row + p/32, slot p % 32); it therefore FETCHES the chunk that belongs in that slot:
// slot ^ swizzle(row).
// Columns past `cols` (edge tiles of a dimension that is not a multiple of 256) are clamped to the
// last 16-byte chunk of the row: the duplicate data lands in tile columns whose results are
// never stored.
// The per-lane source address of stage 0 is computed once (stage_src); a later stage only adds
// a wave-uniform byte offset, so the steady-state loop carries no 64-bit multiplies.
__device__ __forceinline__ const bf16_raw* stage_src(const bf16_raw* base, int64_t ld, int row0, int col0, int cols,
                                                     int half, int wave, int lane) {
  const int row = 16 * half + 2 * wave + (lane >> 5);
  const int chunk = (lane & 31) ^ ((row & 3) << 2);
  const int col = min(col0 + chunk * 8, cols - 8);
  return base + int64_t(row0 + row) * ld + col;
}

// The same address in two parts, for the four-wave kernel: a wave-uniform element offset (rows of this
// wave's role) and a per-lane BYTE offset that fits 32 bits. A load whose address is "scalar base +
// zero-extended 32-bit lane offset" is issued in the scalar-base form of global_load_lds: its per-stage
// address update is scalar arithmetic only, no vector instruction next to the MFMAs.
__device__ __forceinline__ int64_t stage_src_uniform(int64_t ld, int row0, int wave) {
  return int64_t(row0 + 2 * wave) * ld;
}
__device__ __forceinline__ unsigned stage_src_lane(int64_t ld, int col0, int cols, int wave, int lane) {
  const int row = 2 * wave + (lane >> 5);
  const int chunk = (lane & 31) ^ ((row & 3) << 2);
  const int col = min(col0 + chunk * 8, cols - 8);
  return unsigned((int64_t(lane >> 5) * ld + col) * int64_t(sizeof(bf16_raw)));
}

__device__ __forceinline__ void stage_half(const bf16_raw* src, char* tile, int half, int wave) {
  typedef __attribute__((address_space(1))) const void gptr_t;
  typedef __attribute__((address_space(3))) void lptr_t;
  char* dst = tile + (16 * half + 2 * wave) * kRowBytes;  // wave-uniform
  __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)dst, 16, 0, 0);
}

template <int kMode>
__global__ __launch_bounds__(kThreads) void wgrad_kernel(const bf16_raw* __restrict__ A, int64_t lda,
                                                         const bf16_raw* __restrict__ B, int64_t ldb,
                                                         float* __restrict__ c_partial,
                                                         float* __restrict__ bias_partial, int M, int I,
                                                         int J, int S) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A tile | B tile]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 2, wj = wave & 3;  // 2 x 4 waves: 128 (i) x 64 (j) per wave
  const int h = lane >> 5, r = lane & 31;

  // ---- task mapping -------------------------------------------------------------------
  const int tiles_j = (J + kTile - 1) / kTile, tiles_i = (I + kTile - 1) / kTile;  // edge tiles are partial
  const int n_tiles = tiles_i * tiles_j;
  const int n_tasks = n_tiles * S;
  // Workgroup ids are dealt round-robin to the 8 XCDs; give each XCD a contiguous task range.
  const int q8 = n_tasks / 8, r8 = n_tasks % 8;
  const int xcd = blockIdx.x % 8, within = blockIdx.x / 8;
  const int task = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
  const int slice = task / n_tiles;
  const int tile = task % n_tiles;
  const int ti = tile / tiles_j, tj = tile % tiles_j;

  const int steps_total = M / kBM;
  const int step_lo = int(int64_t(steps_total) * slice / S);
  const int step_hi = int(int64_t(steps_total) * (slice + 1) / S);
  const int n_steps = step_hi - step_lo;

  f32x16 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  const bool do_bias = bias_partial != nullptr && tj == 0;  // workgroup-uniform
  float bsum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bsum[e] = 0.f;

  const unsigned smem_base = lds_addr(smem);
  unsigned a_base[4], b_base[2];
#pragma unroll
  for (int a = 0; a < 4; ++a) a_base[a] = frag_base(wi * 4 + a, lane);
#pragma unroll
  for (int b = 0; b < 2; ++b) b_base[b] = frag_base(wj * 2 + b, lane);
  const unsigned bias_off = (unsigned)wg_tile_off(tid >> 5, tid & 31);  // rows +16: +8192 bytes

  // Loads of one stage, oldest first: A rows 0-15, B rows 0-15 (k-step 0), then A and B rows
  // 16-31 (k-step 1) — so "all but the last two" of a stage means its k-step 0 has landed.
  const bf16_raw* a_src[2];
  const bf16_raw* b_src[2];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    a_src[half] = stage_src(A, lda, step_lo * kBM, ti * kTile, I, half, wave, lane);
    b_src[half] = stage_src(B, ldb, step_lo * kBM, tj * kTile, J, half, wave, lane);
  }
  const int64_t a_stride = int64_t(kBM) * lda, b_stride = int64_t(kBM) * ldb;  // elements per stage
  auto issue = [&](int t, int slot_idx) {
    char* slot = smem + slot_idx * kStageBytes;
    const int64_t ao = int64_t(t) * a_stride, bo = int64_t(t) * b_stride;  // wave-uniform
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      stage_half(a_src[half] + ao, slot, half, wave);
      stage_half(b_src[half] + bo, slot + kTileBytes, half, wave);
    }
  };
#pragma unroll
  for (int t = 0; t < kStages - 1; ++t)
    if (t < n_steps) issue(t, t);

  // Fragments of k-step 0 of the stage about to be multiplied; always fetched one stage ahead
  // (before the barrier that opens that stage), so the matrix pipe has work the moment the
  // barrier releases instead of idling for an LDS round trip.
  s16x4 cl[4], ch[4], dl[2], dh[2];
  auto read_k0 = [&](int slot_idx) {
    const unsigned a_tile = smem_base + slot_idx * kStageBytes;
    const unsigned b_tile = a_tile + kTileBytes;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      cl[a] = tr_read<0>(a_tile + a_base[a]);
      ch[a] = tr_read<4096>(a_tile + a_base[a]);
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      dl[b] = tr_read<0>(b_tile + b_base[b]);
      dh[b] = tr_read<4096>(b_tile + b_base[b]);
    }
  };
  {
    // k-step 0 of stage 0: this wave's first two loads, then everyone's.
    if (n_steps >= 3) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (n_steps == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_k0(0);
  }

  for (int t0 = 0; t0 < n_steps; t0 += kStages) {
#pragma unroll
    for (int u = 0; u < kStages; ++u) {
      const int t = t0 + u;
      if (t >= n_steps) break;  // workgroup-uniform
      // Retire this wave's loads of stage t and of k-step 0 of stage t+1, leaving everything
      // younger in flight, then meet the other waves: after the barrier every wave's part of
      // those rows has landed AND every wave is done reading stage t-1, whose slot the next
      // issue overwrites.
      const int younger = n_steps - 1 - t;
      if (kMode == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (younger >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (kMode != 1 && t + kStages - 1 < n_steps) issue(t + kStages - 1, (u + kStages - 1) % kStages);
      if (kMode == 2) continue;

      const unsigned a_tile = smem_base + u * kStageBytes;
      const unsigned b_tile = a_tile + kTileBytes;
      s16x4 el[4], eh[4], fl[2], fh[2];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        el[a] = tr_read<8192>(a_tile + a_base[a]);
        eh[a] = tr_read<12288>(a_tile + a_base[a]);
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        fl[b] = tr_read<8192>(b_tile + b_base[b]);
        fh[b] = tr_read<12288>(b_tile + b_base[b]);
      }
      u32x4 bias_rows[2];
      if (do_bias) {  // column sums of the A tile: thread owns chunk (tid & 31) of rows (tid >> 5) + 16 i
        bias_rows[0] = b128_read<0>(a_tile + bias_off);
        bias_rows[1] = b128_read<8192>(a_tile + bias_off);
        asm volatile("s_waitcnt lgkmcnt(14)" ::: "memory");  // k-step 0 fragments (read a stage ago)
      } else {
        asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 af[4], bfr[2];
#pragma unroll
      for (int a = 0; a < 4; ++a) af[a] = join(cl[a], ch[a]);
#pragma unroll
      for (int b = 0; b < 2; ++b) bfr[b] = join(dl[b], dh[b]);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = mfma32(af[a], bfr[b], acc[a][b]);
      __builtin_amdgcn_sched_barrier(0);
      if (younger >= 1) {  // k-step 0 of stage t+1 (landed before this stage's barrier)
        read_k0((u + 1) % kStages);
        asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = mfma32(join(el[a], eh[a]), join(fl[b], fh[b]), acc[a][b]);
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            bsum[2 * w] += bf16_lo(bias_rows[i][w]);
            bsum[2 * w + 1] += bf16_hi(bias_rows[i][w]);
          }
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();  // the ring is reused below

  // ---- epilogue: fp32 partial tile ------------------------------------------------------
  // acc[a][b] register e of lane (r, h): row i = 32*(4*wi + a) + acc_row(e, h), column j = 32*(2*wj + b) + r.
  float* cp = c_partial + (int64_t(slice) * I + int64_t(ti) * kTile) * J + int64_t(tj) * kTile;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = 32 * (4 * wi + a) + acc_row(e, h);
        const int j = 32 * (2 * wj + b) + r;
        if (ti * kTile + i < I && tj * kTile + j < J) cp[int64_t(i) * J + j] = acc[a][b][e];
      }

  if (do_bias) {  // fold the 16 threads that share a chunk (same tid & 31) through LDS
    float* red = reinterpret_cast<float*>(smem);  // [16][256]
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(tid >> 5) * kTile + (tid & 31) * 8 + e] = bsum[e];
    __syncthreads();
    if (tid < kTile) {
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) s += red[g * kTile + tid];
      if (ti * kTile + tid < I) bias_partial[int64_t(slice) * I + ti * kTile + tid] = s;
    }
  }
}

// ---- four-wave variant ---------------------------------------------------------------------
// Same tile, ring, swizzle and task mapping, but the 256 x 256 tile is split over 2 x 2 waves of
// 128 x 128 each (16 accumulator tiles = 256 accumulator registers per lane, the whole AGPR
// file): a k-step then needs 8 fragments (16 transposed reads) for 16 MFMAs instead of 6 for 8,
// i.e. a third less LDS traffic per MFMA, and half as many waves meet at each barrier. One wave
// per SIMD means nothing hides latency but the schedule itself: fragments of the next k-step are
// always requested before the MFMAs of the current one are issued.
constexpr int kThreads4 = 256;
#ifndef PDA_WGRAD_READS_PER_SLOT
#define PDA_WGRAD_READS_PER_SLOT 2
#endif
#ifndef PDA_WGRAD_LOADS_PER_SLOT
#define PDA_WGRAD_LOADS_PER_SLOT 1
#endif
constexpr int kReadsPerSlot = PDA_WGRAD_READS_PER_SLOT;  // fragment reads issued behind one MFMA
constexpr int kLoadsPerSlot = PDA_WGRAD_LOADS_PER_SLOT;  // LDS-DMA loads issued behind one MFMA

// Phase timing for tools/micro/wgrad_stamp.hip (which includes this file with PDA_WGRAD_STAMPS
// defined): s_memtime ticks per wave summed over the stages of a task, in four phases. The value of a
// stamp is only touched behind a wait the kernel performs anyway, so the schedule is the one that ships.
#ifdef PDA_WGRAD_STAMPS
__device__ unsigned long long g_wgrad_stamps[1024 * 4 * 8];  // [workgroup][wave][wait+barrier, k-step 0, fragment wait, k-step 1,
                                                             //  start -> loop, loop, loop -> end, start (absolute)]
#define PDA_STAMP(var) var = __builtin_amdgcn_s_memtime()
#else
#define PDA_STAMP(var)
#endif

// Segmented operands (kSegmented = 1): the M rows are the concatenation of up to kMaxSegments equally
// long pieces that live at unrelated addresses — the (dY, X) pairs of the micro-batches of a gradient
// accumulation window, whose weight gradient is then ONE long-K product instead of one short product
// plus one slice reduction per micro-batch. Slices may span pieces: the piece and the row of a stage
// follow from its index by a shift and a mask (a piece holds a power-of-two number of stages), and
// the per-lane source pointers are taken relative to piece 0 plus the piece's wave-uniform address
// difference. No state is carried from stage to stage: the stage body stays branch-free.
constexpr int kMaxSegments = 4;
struct WgradSegments {  // plain scalars: an array member would be indexed through scratch memory
  const bf16_raw *a1, *a2, *a3, *b1, *b2, *b3;  // pieces 1..3 (piece 0 is the kernel's A / B)
  int n;          // pieces in use
  int log2_steps;  // log2 of the pipeline stages (kBM rows each) per piece
};

template <int kSegmented>
__global__ __launch_bounds__(kThreads4) void wgrad_kernel4(const bf16_raw* __restrict__ A, int64_t lda,
                                                           const bf16_raw* __restrict__ B, int64_t ldb,
                                                           float* __restrict__ c_partial,
                                                           float* __restrict__ bias_partial, int M, int I,
                                                           int J, int S, int i_fast, WgradSegments segs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef PDA_WGRAD_STAMPS
  const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 1, wj = wave & 1;  // 2 x 2 waves: 128 (i) x 128 (j) per wave
  const int h = lane >> 5, r = lane & 31;

  const int tiles_j = (J + kTile - 1) / kTile, tiles_i = (I + kTile - 1) / kTile;
  const int n_tiles = tiles_i * tiles_j;
  const int n_tasks = n_tiles * S;
  const int q8 = n_tasks / 8, r8 = n_tasks % 8;
  const int xcd = blockIdx.x % 8, within = blockIdx.x / 8;
  const int task = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
  const int slice = task / n_tiles;
  const int tile = task % n_tiles;
  // Consecutive tasks run together on one XCD and share its L2: walking the shorter side of the tile
  // grid fastest keeps the set of distinct A and B blocks they stream per stage small (32 concurrent
  // tasks of the 5 x 20 grid of a 1280 x 5120 gradient touch 12 blocks instead of 22: 392.6 -> 384.5 us
  // at M = 32768; the 15 x 5 and 20 x 5 grids are already on their shorter side).
  const int ti = i_fast ? tile % tiles_i : tile / tiles_j, tj = i_fast ? tile / tiles_i : tile % tiles_j;

  const int steps_total = M / kBM;
  const int step_lo = int(int64_t(steps_total) * slice / S);
  const int step_hi = int(int64_t(steps_total) * (slice + 1) / S);
  const int n_steps = step_hi - step_lo;
  if (n_steps <= 0) return;  // never true (S <= steps); tells the compiler the loop runs, so no path merges
                             // 256 zero accumulators with the loop's results in ordinary registers

  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  const bool do_bias = bias_partial != nullptr && tj == 0;  // workgroup-uniform
  float bsum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bsum[e] = 0.f;

  const unsigned smem_base = lds_addr(smem);
  unsigned a_base[4], b_base[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) a_base[a] = frag_base(wi * 4 + a, lane);
#pragma unroll
  for (int b = 0; b < 4; ++b) b_base[b] = frag_base(wj * 4 + b, lane);
  const unsigned bias_off = (unsigned)wg_tile_off(tid >> 5, tid & 31);  // rows 0..7; +8 rows: +4096 bytes

  // Each wave plays two of the eight "row pair" roles of the eight-wave kernel (wave, wave + 4).
  // Loads of one stage per wave, oldest first: A rows 0-15 (2), B rows 0-15 (2), A rows 16-31 (2),
  // B rows 16-31 (2): the first four are k-step 0.
  // The four roles of a wave differ by 8 q + 16 half rows, a multiple of 4: the swizzle (row & 3) and
  // with it the per-lane column are the same for all of them, so ONE per-lane pointer per operand
  // serves every load and the rest of the address is wave-uniform.
  const bf16_raw* a_uni = A + stage_src_uniform(lda, kSegmented ? 0 : step_lo * kBM, wave);  // wave-uniform
  const bf16_raw* b_uni = B + stage_src_uniform(ldb, kSegmented ? 0 : step_lo * kBM, wave);
  const unsigned a_lane = stage_src_lane(lda, ti * kTile, I, wave, lane);  // bytes, per lane, loop-invariant
  const unsigned b_lane = stage_src_lane(ldb, tj * kTile, J, wave, lane);
  // The loads are written in the scalar-base form by hand (left to the compiler, "uniform + lane" is folded
  // back into one 64-bit vector add per load, next to the MFMAs). M0, the LDS destination, is written and
  // read inside the one statement, one wait state apart as the LDS-DMA read of M0 requires.
  // Stamps of the c_fc gradient at 65,536 rows: k-step 0 block 640 -> 627 ticks, kernel 873-878 -> 809-857 us.
  auto load_to_lds = [&](const bf16_raw* uniform, unsigned lane_bytes, char* tile, int half, int role) {
    const unsigned dst = lds_addr(tile + (16 * half + 2 * role) * kRowBytes);  // wave-uniform
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(lane_bytes), "s"(uniform), "s"(dst) : "memory");  // M0 is reserved: the compiler rewrites it before every use of its own
  };
  const int64_t a_stride = int64_t(kBM) * lda, b_stride = int64_t(kBM) * ldb;
  // Segmented operands: wave-uniform element differences of pieces 1..3 against piece 0.
  // (Locals, so that no lambda below captures the argument struct itself: a captured struct has its
  // address taken, is copied to scratch memory, and drags the loop's scalars into vector registers.)
  int64_t a_d1 = 0, a_d2 = 0, a_d3 = 0, b_d1 = 0, b_d2 = 0, b_d3 = 0;
  const int seg_log2 = kSegmented ? segs.log2_steps : 0;
  if (kSegmented) {
    a_d1 = int64_t(segs.a1 - A), a_d2 = int64_t(segs.a2 - A), a_d3 = int64_t(segs.a3 - A);
    b_d1 = int64_t(segs.b1 - B), b_d2 = int64_t(segs.b2 - B), b_d3 = int64_t(segs.b3 - B);
  }
  // Load `i` (0..7) of stage t into ring slot `slot_idx`; issue order = age order:
  // A rows 0-15 (2 loads), B rows 0-15 (2), A rows 16-31 (2), B rows 16-31 (2).
  auto issue_one = [&](int t, int slot_idx, int i) {
    char* slot = smem + slot_idx * kStageBytes;
    const int half = i >> 2, is_b = (i >> 1) & 1, q = i & 1;
    if (kSegmented) {
      const int g = step_lo + (t < n_steps ? t : n_steps - 1);  // stage index over all pieces, wave-uniform
      const int piece = g >> seg_log2, local = g & ((1 << seg_log2) - 1);
      const int64_t rows = int64_t(local) * kBM + 16 * half + 8 * q;
      // Selects between VALUES (copied first): `c ? b_d3 : b_d2` on the by-reference captures themselves
      // is a select between two addresses followed by a load, which pins every captured local in
      // scratch memory and drags the loop's scalars into vector registers.
      const bool lo = (piece & 1) != 0, hi = (piece & 2) != 0;
      if (is_b) {
        const int64_t d1 = b_d1, d2 = b_d2, d3 = b_d3;
        const int64_t d_hi = lo ? d3 : d2, d_lo = lo ? d1 : int64_t(0);
        const int64_t d = hi ? d_hi : d_lo;
        load_to_lds(b_uni + d + rows * ldb, b_lane, slot + kTileBytes, half, wave + 4 * q);
      } else {
        const int64_t d1 = a_d1, d2 = a_d2, d3 = a_d3;
        const int64_t d_hi = lo ? d3 : d2, d_lo = lo ? d1 : int64_t(0);
        const int64_t d = hi ? d_hi : d_lo;
        load_to_lds(a_uni + d + rows * lda, a_lane, slot, half, wave + 4 * q);
      }
      return;
    }
    const int64_t rows = int64_t(t < n_steps ? t : n_steps - 1) * kBM + 16 * half + 8 * q;  // wave-uniform
    if (is_b) load_to_lds(b_uni + rows * ldb, b_lane, slot + kTileBytes, half, wave + 4 * q);
    else load_to_lds(a_uni + rows * lda, a_lane, slot, half, wave + 4 * q);
  };
  // Every stage issues exactly 8 loads per wave — past the end of the slice they re-fetch its last
  // rows into a slot nobody reads any more — so the counted waits are the same in every iteration
  // and the loop body has no branch that would cut its MFMA sequences into separate blocks.
#pragma unroll
  for (int t = 0; t < kStages - 1; ++t)
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_one(t, t, i);

  // Fragment registers: set 0 = k-step 0 of the stage being multiplied (requested a stage ahead),
  // set 1 = its k-step 1 (requested under the MFMAs of k-step 0).
  s16x4 al[2][4], ah[2][4], bl[2][4], bh[2][4];
  // Transposed read number `i` (0..15) of k-step `ks` from the stage in ring slot `slot_idx`.
  auto frag_read = [&](int set, int ks, int slot_idx, int i) {
    const unsigned a_tile = smem_base + slot_idx * kStageBytes;
    const unsigned b_tile = a_tile + kTileBytes;
    const int x = (i >> 1) & 3;
    if (ks == 0) {
      if (i < 8) { if (i & 1) ah[set][x] = tr_read<4096>(a_tile + a_base[x]); else al[set][x] = tr_read<0>(a_tile + a_base[x]); }
      else       { if (i & 1) bh[set][x] = tr_read<4096>(b_tile + b_base[x]); else bl[set][x] = tr_read<0>(b_tile + b_base[x]); }
    } else {
      if (i < 8) { if (i & 1) ah[set][x] = tr_read<12288>(a_tile + a_base[x]); else al[set][x] = tr_read<8192>(a_tile + a_base[x]); }
      else       { if (i & 1) bh[set][x] = tr_read<12288>(b_tile + b_base[x]); else bl[set][x] = tr_read<8192>(b_tile + b_base[x]); }
    }
  };
  {
    // k-step 0 of stage 0: this wave's first four loads, then everyone's.
    asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) frag_read(0, 0, 0, i);
  }

#ifdef PDA_WGRAD_STAMPS
  const unsigned long long t_loop = __builtin_amdgcn_s_memtime();
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts2p = 0, ts3 = 0, sum_wait = 0, sum_k0 = 0, sum_frag = 0, sum_k1 = 0;
#endif
  auto main_loop = [&](auto with_bias) {
    constexpr bool kBias = decltype(with_bias)::value;
    for (int t0 = 0; t0 < n_steps; t0 += kStages) {
#pragma unroll
      for (int u = 0; u < kStages; ++u) {
        const int t = t0 + u;
        if (t >= n_steps) break;  // workgroup-uniform
        // Stage t complete and k-step 0 of stage t+1 landed; the 12 younger loads stay in flight.
        // This wave's k-step 0 fragments (requested a stage ago) are awaited here as well, so that
        // the matrix pipe starts the moment the barrier opens.
        PDA_STAMP(ts0);
        asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        PDA_STAMP(ts1);
        const unsigned a_tile = smem_base + u * kStageBytes;
        // ---- k-step 0: 16 MFMAs; in their shadow the 16 fragment reads of k-step 1 (two after
        // each of the first eight), then the 8 global loads that refill the slot the barrier freed.
        bf16x8 af[4], bfr[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          af[x] = join(al[0][x], ah[0][x]);
          bfr[x] = join(bl[0][x], bh[0][x]);
        }
        u32x4 bias_rows[4];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          acc[i >> 2][i & 3] = mfma32(af[i >> 2], bfr[i & 3], acc[i >> 2][i & 3]);
          if (i < 16 / kReadsPerSlot) {
#pragma unroll
            for (int q = 0; q < kReadsPerSlot; ++q) frag_read(1, 1, u, kReadsPerSlot * i + q);
          }
          if (i >= 16 - 8 / kLoadsPerSlot) {
#pragma unroll
            for (int q = 0; q < kLoadsPerSlot; ++q)
              issue_one(t + kStages - 1, (u + kStages - 1) % kStages, kLoadsPerSlot * (i - (16 - 8 / kLoadsPerSlot)) + q);
          }
          if (kBias) {  // column sums of the A tile: chunk (tid & 31) of rows (tid >> 5) + 8 j
            if (i == 8) bias_rows[0] = b128_read<0>(a_tile + bias_off);
            if (i == 9) bias_rows[1] = b128_read<4096>(a_tile + bias_off);
            if (i == 10) bias_rows[2] = b128_read<8192>(a_tile + bias_off);
            if (i == 11) bias_rows[3] = b128_read<12288>(a_tile + bias_off);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        PDA_STAMP(ts2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // k-step 1 fragments (and the bias rows)
        __builtin_amdgcn_sched_barrier(0);
#ifdef PDA_WGRAD_STAMPS
        if (ts3 != 0) {  // ts3, ts2p: the previous stage's
          sum_k1 += ts0 - ts3;
          sum_frag += ts3 - ts2p;
        }
        ts2p = ts2;
        sum_wait += ts1 - ts0;
        sum_k0 += ts2 - ts1;
        PDA_STAMP(ts3);
        __builtin_amdgcn_sched_barrier(0);
#endif

        // ---- k-step 1: 16 MFMAs; in their shadow the fragment reads of k-step 0 of stage t+1
        // (landed before this stage's barrier; after the last stage they fetch stale bytes that
        // nothing uses).
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          af[x] = join(al[1][x], ah[1][x]);
          bfr[x] = join(bl[1][x], bh[1][x]);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          acc[i >> 2][i & 3] = mfma32(af[i >> 2], bfr[i & 3], acc[i >> 2][i & 3]);
          if (i < 16 / kReadsPerSlot) {
#pragma unroll
            for (int q = 0; q < kReadsPerSlot; ++q) frag_read(0, 0, (u + 1) % kStages, kReadsPerSlot * i + q);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (kBias) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              bsum[2 * w] += bf16_lo(bias_rows[i][w]);
              bsum[2 * w + 1] += bf16_hi(bias_rows[i][w]);
            }
        }
      }
    }
  };
  if (do_bias) main_loop(std::true_type{});
  else main_loop(std::false_type{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail's dummy loads
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();  // the ring is reused below
#ifdef PDA_WGRAD_STAMPS
  const unsigned long long t_loop_end = __builtin_amdgcn_s_memtime();
#endif

  // One accumulator tile at a time leaves the accumulator file (the fences keep the compiler from
  // moving all 256 values into ordinary registers at once, which it can only do by spilling).
  float* cp = c_partial + (int64_t(slice) * I + int64_t(ti) * kTile) * J + int64_t(tj) * kTile;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      f32x16 tile_vals = acc[a][b];
      asm volatile("" : "+v"(tile_vals));
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = 32 * (4 * wi + a) + acc_row(e, h);
        const int j = 32 * (4 * wj + b) + r;
        if (ti * kTile + i < I && tj * kTile + j < J) cp[int64_t(i) * J + j] = tile_vals[e];
      }
      __builtin_amdgcn_sched_barrier(0);
    }

  if (do_bias) {  // fold the 8 threads that share a chunk (same tid & 31) through LDS
    float* red = reinterpret_cast<float*>(smem);  // [8][256]
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(tid >> 5) * kTile + (tid & 31) * 8 + e] = bsum[e];
    __syncthreads();
    {
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) s += red[g * kTile + tid];
      if (ti * kTile + tid < I) bias_partial[int64_t(slice) * I + ti * kTile + tid] = s;
    }
  }
#ifdef PDA_WGRAD_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the partial stores have left the wave
  const unsigned long long t_end = __builtin_amdgcn_s_memtime();
  if (lane == 0 && blockIdx.x < 1024) {
    unsigned long long* g = g_wgrad_stamps + (blockIdx.x * 4 + wave) * 8;
    g[0] = sum_wait, g[1] = sum_k0, g[2] = sum_frag, g[3] = sum_k1;
    g[4] = t_loop - t_start, g[5] = t_loop_end - t_loop, g[6] = t_end - t_loop_end, g[7] = t_start;
  }
#endif
}

// ---- four-wave variant on 16 x 16 x 32 MFMAs ----------------------------------------------------
// Same 256 x 256 tile over 2 x 2 waves of 128 x 128, same LDS-DMA ring, but the products are
// v_mfma_f32_16x16x32_bf16: 64 accumulator tiles of 16 x 16 per wave (still 256 accumulator
// registers) and ONE MFMA k-step (32 rows of M) per pipeline stage. Why: the 32 x 32 x 16 kernel above
// runs power-throttled (effective clock 2.00 GHz in the GPT-2 large step, against 2.13-2.20 GHz for
// the library's 16 x 16 kernels next to it), and at equal cycles per FLOP the chip holds a higher
// clock on the 16 x 16 x 32 shape. Measured: 1-2 % faster than the 32 x 32 x 16 kernel back to back in
// isolation (c_fc 404 vs 411 us incl. the slice reduction), 1 % slower inside the training step
// (306.6 vs 303.0 us average): per stage the 16 x 16 form leaves the instruction issue port 8 of every
// 16 cycles instead of 24 of every 32, and the 32 fragment reads plus 8 LDS-DMA loads (tens of issue
// cycles each) no longer fit the MFMAs' shadow. Opt-in (PDA_WGRAD_MFMA=16).
// Where the 32 x 32 x 16 kernel's time goes (tools/micro/wgrad_stamp.hip, s_memtime per phase; c_fc
// gradient, back to back): a stage takes ~1,330 clock ticks against 1,024 of matrix work — the
// k-step 0 block (16 MFMAs, 16 fragment reads, 8 LDS-DMA loads) 656, k-step 1 (16 MFMAs, 16 reads)
// 579, the two waits ~100 after subtracting the stamps' own round trip — at a clock of 1.71 GHz
// (2.4 nominal: power). The loop is 93 % of a task, the partial-tile stores 5 %. The L1 has 62 lines
// in flight on average (it sustains 143 in tools/micro/feed_probe.hip) and the data of a stage lands
// two stages ahead: neither the L2 -> LDS feed nor the counted waits hold the loop, the stalls sit
// inside the MFMA blocks (fragment reads and DMA writes keep the LDS at ~75 B/cycle of its 128).
//   * k order: lane group g (lane / 16) of an MFMA operand holds contraction slots 8g..8g+7; they are
//     fed tile rows {4g..4g+3, 16+4g..16+4g+3} — any assignment works as long as both operands use the
//     same one — so that the first transposed read of every fragment touches rows 0-15 and the second
//     rows 16-31, and the two groups of a half-wave read rows that differ in bit 2;
//   * swizzle: 16-byte chunk index XOR ((row & 3) << 2 | ((row >> 2) & 1) << 1): the 32 lanes of a
//     half-wave (rows 4g + q for two values of g, four 8-byte pieces per row) then cover all 64 banks
//     once. Rows that differ by a multiple of 8 share the swizzle, so one per-lane source pointer per
//     operand still serves every LDS-DMA load of a wave;
//   * schedule: all 32 fragment reads of stage t+1 are issued under the first 32 MFMAs of stage t
//     into the other half of a double-buffered fragment file, the 8 LDS-DMA loads of stage t+4 under
//     the last 32 into the slot of stage t itself (its fragments are in registers by then): four
//     stages, three of them (96 KiB per CU) in flight behind the one being read, one barrier and one
//     counted wait (vmcnt(16)) per stage;
//   * the product is formed with the B-operand fragment on the MFMA's row side, so a lane ends up
//     with four CONSECUTIVE output columns per accumulator tile: the fp32 partial tile leaves in
//     16-byte stores.
// Compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}).
// The 64-MFMA stage body below must be straight-line code with constant register indices whatever the
// loop unroller's size limits say.
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ int wg_tile_off16(int row, int chunk) {
  return row * kRowBytes + ((chunk ^ (((row & 3) << 2) | (((row >> 2) & 1) << 1))) << 4);
}

// Per-lane LDS byte offset (inside one operand tile) of the fragment of 16-column block `cblock`:
// lane (g, i) receives tile[4g + j][16*cblock + i], j = 0..3, from the transposed read at this
// offset and tile[16 + 4g + j][...] from the one at + 8192.
__device__ __forceinline__ unsigned frag_base16(int cblock, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const int q = i >> 2, p = i & 3;
  const int col = cblock * 16 + 4 * p;
  return (unsigned)(wg_tile_off16(4 * g + q, col >> 3) + (col & 7) * 2);
}

__device__ __forceinline__ const bf16_raw* stage_src16(const bf16_raw* base, int64_t ld, int row0, int col0, int cols,
                                                       int wave, int lane) {
  const int row = 2 * wave + (lane >> 5);  // + 8 q + 16 half for the other pieces: same swizzle
  const int chunk = (lane & 31) ^ (((row & 3) << 2) | (((row >> 2) & 1) << 1));
  const int col = min(col0 + chunk * 8, cols - 8);
  return base + int64_t(row0 + row) * ld + col;
}

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <int kUnused>
__global__ __launch_bounds__(kThreads4) void wgrad_kernel16(const bf16_raw* __restrict__ A, int64_t lda,
                                                            const bf16_raw* __restrict__ B, int64_t ldb,
                                                            float* __restrict__ c_partial,
                                                            float* __restrict__ bias_partial, int M, int I,
                                                            int J, int S) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 1, wj = wave & 1;  // 2 x 2 waves: 128 (i) x 128 (j) per wave

  const int tiles_j = (J + kTile - 1) / kTile, tiles_i = (I + kTile - 1) / kTile;
  const int n_tiles = tiles_i * tiles_j;
  const int n_tasks = n_tiles * S;
  const int q8 = n_tasks / 8, r8 = n_tasks % 8;
  const int xcd = blockIdx.x % 8, within = blockIdx.x / 8;
  const int task = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
  const int slice = task / n_tiles;
  const int tile = task % n_tiles;
  const int ti = tile / tiles_j, tj = tile % tiles_j;

  const int steps_total = M / kBM;
  const int step_lo = int(int64_t(steps_total) * slice / S);
  const int step_hi = int(int64_t(steps_total) * (slice + 1) / S);
  const int n_steps = step_hi - step_lo;
  if (n_steps <= 0) return;  // never true (S <= steps); removes the path that merges zero accumulators with the loop's

  f32x4 acc[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[a][b][e] = 0.f;
  const bool do_bias = bias_partial != nullptr && tj == 0;  // workgroup-uniform
  float bsum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bsum[e] = 0.f;

  // Fragment offsets are the same for both operands up to a wave-uniform 256 bytes per 128 columns,
  // which goes into the (scalar) tile base: one set of eight per-lane offsets serves A and B.
  const unsigned smem_base = lds_addr(smem);
  unsigned f_base[8];
#pragma unroll
  for (int x = 0; x < 8; ++x) f_base[x] = frag_base16(x, lane);
  const unsigned a_wave = smem_base + 256u * wi, b_wave = smem_base + kTileBytes + 256u * wj;
  const unsigned bias_off = (unsigned)wg_tile_off16(tid >> 5, tid & 31);  // rows 0..7; + 8 rows: + 4096 bytes

  const bf16_raw* a_src = stage_src16(A, lda, step_lo * kBM, ti * kTile, I, wave, lane);
  const bf16_raw* b_src = stage_src16(B, ldb, step_lo * kBM, tj * kTile, J, wave, lane);
  // Load `i` (0..7) of stage t into ring slot `slot_idx`: pieces (half, q) of A (i < 4) or B; a piece is rows
  // 16 half + 8 q + 2 wave, + 1 (1 KiB per wave instruction). Past the end of the slice the last stage
  // is fetched again into a slot nobody reads, so every stage issues exactly 8 loads per wave and the
  // counted wait is the same in every iteration.
  auto issue_one = [&](int t, int slot_idx, int i) {
    char* slot = smem + slot_idx * kStageBytes;
    const int is_b = i >> 2, half = (i >> 1) & 1, q = i & 1;
    const int64_t rows = int64_t(t < n_steps ? t : n_steps - 1) * kBM + 16 * half + 8 * q;  // wave-uniform
    if (is_b) stage_half(b_src + rows * ldb, slot + kTileBytes, half, wave + 4 * q);
    else stage_half(a_src + rows * lda, slot, half, wave + 4 * q);
  };
#pragma unroll
  for (int t = 0; t < kStages; ++t)
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_one(t, t, i);

  // Fragment file (lo = contraction slots from tile rows 0-15, hi = rows 16-31). The MFMAs of a stage
  // run row by row over the A fragments (a = 0..7, all eight B fragments each), so the B fragments and
  // the last two A fragments are double buffered by stage parity, while A fragments 0-5 are single
  // buffered: fragment a of the NEXT stage is requested as soon as row a of this stage has been issued.
  s16x4 alo[2][8], ahi[2][8], blo[2][8], bhi[2][8];
  auto read_a = [&](int set, int slot_idx, int x, int hi) {
    const unsigned base = a_wave + slot_idx * kStageBytes + f_base[x];
    const int as = x >= 6 ? set : 0;
    if (hi) ahi[as][x] = tr_read<8192>(base);
    else alo[as][x] = tr_read<0>(base);
  };
  auto read_b = [&](int set, int slot_idx, int x, int hi) {
    const unsigned base = b_wave + slot_idx * kStageBytes + f_base[x];
    if (hi) bhi[set][x] = tr_read<8192>(base);
    else blo[set][x] = tr_read<0>(base);
  };
  u32x4 bias_rows[4];
  auto bias_read = [&](int slot_idx, int j) {
    const unsigned a_tile = smem_base + slot_idx * kStageBytes;
    if (j == 0) bias_rows[0] = b128_read<0>(a_tile + bias_off);
    if (j == 1) bias_rows[1] = b128_read<4096>(a_tile + bias_off);
    if (j == 2) bias_rows[2] = b128_read<8192>(a_tile + bias_off);
    if (j == 3) bias_rows[3] = b128_read<12288>(a_tile + bias_off);
  };
  {
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");  // stage 0: this wave's part, then everyone's
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      read_b(0, 0, x, 0);
      read_b(0, 0, x, 1);
      read_a(0, 0, x, 0);
      read_a(0, 0, x, 1);
    }
    if (do_bias) {
#pragma unroll
      for (int j = 0; j < 4; ++j) bias_read(0, j);
    }
  }

  auto main_loop = [&](auto with_bias) {
    constexpr bool kBias = decltype(with_bias)::value;
    for (int t0 = 0; t0 < n_steps; t0 += kStages) {
      static_for<kStages>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        const int t = t0 + u;
        if (t >= n_steps) return;  // workgroup-uniform
        constexpr int set = u & 1, nslot = (u + 1) % kStages;
        // Stage t+1 landed (stages t+2 and t+3 stay in flight) and this wave's fragments of stage t, read
        // during stage t-1, are in registers; after the barrier that holds for every wave, so the slot
        // of stage t may be refilled and the slot of stage t+1 may be read.
        asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (kBias) {  // column sums of the A tile of THIS stage (rows read during the previous one)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              bsum[2 * w] += bf16_lo(bias_rows[j][w]);
              bsum[2 * w + 1] += bf16_hi(bias_rows[j][w]);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
        static_for<64>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          constexpr int a = i >> 3, b = i & 7;
          constexpr int as = a >= 6 ? set : 0;
          acc[a][b] = mfma16(join(blo[set][b], bhi[set][b]), join(alo[as][a], ahi[as][a]), acc[a][b]);
          // In the MFMA's shadow: next stage's B fragments and A fragments 6, 7 (other buffer) first, then A
          // fragment x once row x is behind us, and the eight loads that refill this stage's slot.
          if constexpr (i < 16) read_b(set ^ 1, nslot, i >> 1, i & 1);
          else if constexpr (i < 20) read_a(set ^ 1, nslot, 6 + ((i - 16) >> 1), i & 1);
          else if constexpr (i < 24) read_a(0, nslot, (i - 20) >> 1, i & 1);                    // A0, A1 (rows 0, 1 done at i = 15)
          else if constexpr (i < 56 && (i & 7) < 2) read_a(0, nslot, (i >> 3) - 1, i & 1);     // A2 at 24, A3 at 32, A4 at 40, A5 at 48
          else if constexpr (i >= 26 && (i & 3) == 2 && i < 58) issue_one(t + kStages, u, (i - 26) >> 2);
          if constexpr (kBias && (i == 28 || i == 36)) bias_read(nslot, (i - 28) >> 3);
          if constexpr (kBias && (i == 59 || i == 61)) bias_read(nslot, 2 + ((i - 59) >> 1));
          __builtin_amdgcn_sched_barrier(0);
        });
      });
    }
  };
  if (do_bias) main_loop(std::true_type{});
  else main_loop(std::false_type{});
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the tail's dummy loads and reads
  __syncthreads();  // the ring is reused below

  // acc[a][b] register e of lane (g, c): row i = 16*(8*wi + a) + c, column j = 16*(8*wj + b) + 4*g + e.
  // One accumulator tile at a time leaves the accumulator file (see the kernel above).
  const int g = lane >> 4, c = lane & 15;
  float* cp = c_partial + (int64_t(slice) * I + int64_t(ti) * kTile) * J + int64_t(tj) * kTile;
#pragma unroll
  for (int a = 0; a < 8; ++a) {
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      f32x4 v = acc[a][b];
      asm volatile("" : "+v"(v));
      const int i = 16 * (8 * wi + a) + c;
      const int j = 16 * (8 * wj + b) + 4 * g;
      if (ti * kTile + i < I && tj * kTile + j < J) *reinterpret_cast<f32x4*>(cp + int64_t(i) * J + j) = v;
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  if (do_bias) {  // fold the 8 threads that share a chunk (same tid & 31) through LDS
    float* red = reinterpret_cast<float*>(smem);  // [8][256]
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(tid >> 5) * kTile + (tid & 31) * 8 + e] = bsum[e];
    __syncthreads();
    {
      float s = 0.f;
#pragma unroll
      for (int gi = 0; gi < 8; ++gi) s += red[gi * kTile + tid];
      if (ti * kTile + tid < I) bias_partial[int64_t(slice) * I + ti * kTile + tid] = s;
    }
  }
}

// out[n] (+)= sum_s partial[s][n], n over `numel` elements, 4 per thread. The last
// `bias_blocks` workgroups of the grid do the same for the (much smaller) bias partials, so a
// linear layer's weight and bias gradients are folded by one launch (each written or added
// to its destination, which may be a slice of a reducer's flat gradient buffer).
template <typename OutT, typename BiasT>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int S,
                                                           int64_t numel, int64_t slice_stride, OutT* __restrict__ out,
                                                           bool accumulate, const float* __restrict__ bias_partial,
                                                           int64_t bias_numel, BiasT* __restrict__ bias_out,
                                                           bool bias_accumulate, int bias_blocks) {
  const int main_blocks = int(gridDim.x) - bias_blocks;
  if (int(blockIdx.x) >= main_blocks) {
    const int64_t stride = int64_t(bias_blocks) * 256 * 4;
    for (int64_t n = (int64_t(int(blockIdx.x) - main_blocks) * 256 + threadIdx.x) * 4; n < bias_numel; n += stride) {
      f32x4 s = *reinterpret_cast<const f32x4*>(bias_partial + n);
      for (int k = 1; k < S; ++k) s += *reinterpret_cast<const f32x4*>(bias_partial + int64_t(k) * bias_numel + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = s[e];
        if (bias_accumulate) v += VecIO<BiasT>::load1(bias_out + n + e);
        VecIO<BiasT>::store1(bias_out + n + e, v);
      }
    }
    return;
  }
  const int64_t stride = int64_t(main_blocks) * 256 * 4;
  for (int64_t n = (int64_t(blockIdx.x) * 256 + threadIdx.x) * 4; n < numel; n += stride) {
    f32x4 s = *reinterpret_cast<const f32x4*>(partial + n);
    for (int k = 1; k < S; ++k) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(partial + int64_t(k) * slice_stride + n);
      s += v;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = s[e];
      if (accumulate) v += VecIO<OutT>::load1(out + n + e);
      VecIO<OutT>::store1(out + n + e, v);
    }
  }
}

}  // namespace

int wgrad_pick_slices(int M, int I, int J) {
  const int n_tiles = ((I + kTile - 1) / kTile) * ((J + kTile - 1) / kTile);
  const int steps = M / kBM;
  static const int forced = getenv("PDA_WGRAD_SLICES") ? atoi(getenv("PDA_WGRAD_SLICES")) : 0;  // A/B measurements
  if (forced > 0) return forced < steps ? forced : steps;
  int best = 1;
  double best_score = -1.0;
  for (int s = 1; s <= 16 && s <= steps; ++s) {
    const int tasks = n_tiles * s;
    const int rounds = (tasks + 255) / 256;
    const double score = double(tasks) / (rounds * 256.0) - 0.015 * s;
    if (score > best_score) {
      best_score = score;
      best = s;
    }
  }
  return best;
}

bool wgrad_supported(int M, int I, int J) {
  // 16-byte chunks must not straddle the end of a row; partial edge tiles are handled in-kernel.
  return M >= kBM && M % kBM == 0 && I % 8 == 0 && J % 8 == 0 && I >= 8 && J >= 8;
}

void wgrad_gemm(const void* A, int64_t lda, const void* B, int64_t ldb, float* c_partial,
                float* bias_partial, int M, int I, int J, int S, hipStream_t stream) {
  const int n_tasks = ((I + kTile - 1) / kTile) * ((J + kTile - 1) / kTile) * S;
  static int mode = -1, waves = 4, mfma = 32, order = -1;
  if (mode < 0) {
    (void)hipFuncSetAttribute((const void*)wgrad_kernel4<0>, hipFuncAttributeMaxDynamicSharedMemorySize, kStages * kStageBytes);
    (void)hipFuncSetAttribute((const void*)wgrad_kernel16<0>, hipFuncAttributeMaxDynamicSharedMemorySize, kStages * kStageBytes);
    const char* mf = getenv("PDA_WGRAD_MFMA");  // 16 selects the 16 x 16 x 32 four-wave kernel (A/B measurements)
    mfma = mf ? atoi(mf) : 32;
    const char* w = getenv("PDA_WGRAD_WAVES");  // 8 selects the eight-wave kernel (A/B measurements)
    waves = w ? atoi(w) : 4;
    (void)hipFuncSetAttribute((const void*)wgrad_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, kStages * kStageBytes);
    (void)hipFuncSetAttribute((const void*)wgrad_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kStages * kStageBytes);
    (void)hipFuncSetAttribute((const void*)wgrad_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, kStages * kStageBytes);
    const char* o = getenv("PDA_WGRAD_ORDER");  // 0 / 1 force the j-fastest / i-fastest tile order (A/B measurements)
    order = o ? atoi(o) : -1;
    const char* e = getenv("PDA_WGRAD_MODE");
    mode = e ? atoi(e) : 0;
  }
  if (waves == 4 && mode == 0 && mfma == 16) {
    hipLaunchKernelGGL(wgrad_kernel16<0>, dim3((unsigned)n_tasks), dim3(kThreads4), kStages * kStageBytes, stream,
                       (const bf16_raw*)A, lda, (const bf16_raw*)B, ldb, c_partial, bias_partial, M, I, J, S);
    return;
  }
  if (waves == 4 && mode == 0) {
    const int tiles_i = (I + kTile - 1) / kTile, tiles_j = (J + kTile - 1) / kTile;
    const int i_fast = order < 0 ? (tiles_j > tiles_i) : order;
    hipLaunchKernelGGL(wgrad_kernel4<0>, dim3((unsigned)n_tasks), dim3(kThreads4), kStages * kStageBytes, stream,
                       (const bf16_raw*)A, lda, (const bf16_raw*)B, ldb, c_partial, bias_partial, M, I, J, S, i_fast,
                       WgradSegments{});
    return;
  }
  auto kern = mode == 1 ? wgrad_kernel<1> : mode == 2 ? wgrad_kernel<2> : wgrad_kernel<0>;
  hipLaunchKernelGGL(kern, dim3((unsigned)n_tasks), dim3(kThreads), kStages * kStageBytes, stream,
                     (const bf16_raw*)A, lda, (const bf16_raw*)B, ldb, c_partial, bias_partial, M, I, J, S);
}

int wgrad_max_segments() { return kMaxSegments; }

void wgrad_gemm_segments(const void* const* A, int64_t lda, const void* const* B, int64_t ldb, int n_segments,
                         float* c_partial, float* bias_partial, int rows_per_segment, int I, int J, int S,
                         hipStream_t stream) {
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)wgrad_kernel4<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kStages * kStageBytes);
    configured = true;
  }
  auto piece = [&](const void* const* p, int k) { return (const bf16_raw*)p[k < n_segments ? k : 0]; };
  int log2_steps = 0;
  while ((1 << log2_steps) < rows_per_segment / kBM) ++log2_steps;
  WgradSegments segs{piece(A, 1), piece(A, 2), piece(A, 3), piece(B, 1), piece(B, 2), piece(B, 3), n_segments,
                     log2_steps};
  const int tiles_i = (I + kTile - 1) / kTile, tiles_j = (J + kTile - 1) / kTile;
  const int n_tasks = tiles_i * tiles_j * S;
  hipLaunchKernelGGL(wgrad_kernel4<1>, dim3((unsigned)n_tasks), dim3(kThreads4), kStages * kStageBytes, stream,
                     piece(A, 0), lda, piece(B, 0), ldb, c_partial, bias_partial, rows_per_segment * n_segments, I, J, S,
                     int(tiles_j > tiles_i), segs);
}

void wgrad_reduce(const float* partial, int S, int64_t numel, int64_t slice_stride, DType out_dtype, void* out, bool accumulate,
                  const float* bias_partial, int64_t bias_numel, DType bias_dtype, void* bias_out,
                  bool bias_accumulate, hipStream_t stream) {
  int64_t blocks = ceil_div(numel, 1024);
  if (blocks > 2048) blocks = 2048;
  const int bias_blocks = bias_partial != nullptr ? (int)(ceil_div(bias_numel, 1024) < 8 ? ceil_div(bias_numel, 1024) : 8) : 0;
  const dim3 grid((unsigned)(blocks + bias_blocks));
#define PDA_REDUCE(OUT_T, BIAS_T)                                                                             \
  hipLaunchKernelGGL((wgrad_reduce_kernel<OUT_T, BIAS_T>), grid, dim3(256), 0, stream, partial, S, numel,       \
                     slice_stride, (OUT_T*)out, accumulate, bias_partial, bias_numel, (BIAS_T*)bias_out, bias_accumulate, \
                     bias_blocks)
  if (out_dtype == kBF16) {
    if (bias_dtype == kBF16) PDA_REDUCE(bf16_raw, bf16_raw); else PDA_REDUCE(bf16_raw, float);
  } else {
    if (bias_dtype == kBF16) PDA_REDUCE(float, bf16_raw); else PDA_REDUCE(float, float);
  }
#undef PDA_REDUCE
}

}  // namespace pda
