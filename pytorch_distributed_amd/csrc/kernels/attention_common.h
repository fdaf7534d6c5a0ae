// Shared pieces of the causal attention kernels (forward, dQ, dK/dV) for gfx950.
//
// Data layout. The kernels read the packed QKV activation the c_attn GEMM
// produces, [B, T, 3, H, D] bf16, directly (token stride 3*H*D), and write
// [B, T, H, D] outputs, so no split / transpose / reshape copy exists on
// either side (the reference materialises three of them plus the T x T score
// matrix: assignments/assignment1/model/my_gpt2.py:43-48,63-76).
//
// MFMA tiling. Everything is built from v_mfma_f32_32x32x16_bf16 with the
// "reduction index on the registers, free index on the lane" orientation: a
// 32x32 fp32 result tile keeps its column on the lane (lane & 31) and its 32
// rows spread over 16 registers x 2 lane halves. A following product that
// sums over the rows of such a tile takes it straight from the accumulator
// registers as its B operand (converted to bf16), provided the other operand
// is fed in the same permuted k order:
//     element j of lane-half h  <->  row 16*s + 8*(j>>2) + 4*h + (j&3)
// for k-step s. The transposed LDS read ds_read_b64_tr_b16 delivers exactly
// that order from a row-major tile, so P (forward) and dS (backward) never
// travel through LDS.
//
// LDS tiles are row-major [rows][D] bf16 with the 16-byte chunk index XOR
// swizzled (tile_off) so that both access shapes are bank-conflict free:
//   * "row reads"  (ds_read_b128, one row per lane, same chunk in all lanes)
//   * "transposed reads" (ds_read_b64_tr_b16, 4 rows x 64 bytes per 32 lanes)
#pragma once

#include "common.h"
#include "philox.h"

namespace pda {
namespace attn {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

typedef __attribute__((ext_vector_type(8))) short s16x8;

// Byte offset of 16-byte chunk `chunk` of row `row` in an LDS tile. The XOR
// term serves both access shapes at once:
//   D = 64 (128-byte rows, two rows per 256-byte bank row): a row read touches
//   16 rows distinct mod 16 at one chunk -> (row&1, bits 1..3 of row) must map
//   to 16 different 16-byte slots; a transposed read touches rows 4a..4a+3 x
//   4 chunks of one 64-byte segment -> bit 1 of the row must flip the segment.
//   D = 128 (256-byte rows): rows mod 16 must permute all 16 slots, and
//   row & 3 must select the segment.
template <int D>
__device__ __forceinline__ int tile_off(int row, int chunk) {
  if constexpr (D == 64) {
    const int x = (((row >> 1) & 1) << 2) | ((row >> 2) & 1) | (((row >> 3) & 1) << 1);
    return row * 128 + ((chunk ^ x) << 4);
  } else {
    const int x = ((row & 3) << 2) | ((row >> 2) & 3);
    return row * 256 + ((chunk ^ x) << 4);
  }
}

__device__ __forceinline__ bf16x8 lds_read_b128(const char* lds, int byte_off) {
  return *reinterpret_cast<const bf16x8*>(lds + byte_off);
}

// One ds_read_b64_tr_b16: within each group of 16 lanes, lane i receives
// column i of a 4-row x 16-column block; lane 4q+p supplies the address of
// row q, columns 4p..4p+3. Must execute with all 64 lanes active.
__device__ __forceinline__ s16x4 lds_read_tr(const char* lds, int byte_off) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (lds_s16x4*)(__attribute__((address_space(3))) char*)(lds + byte_off));
}

// A-operand fragment (32 rows of the transposed tile x 16 k) for k-step
// `kstep` and 32-column block `cblock` of a transposed-read tile holding
// [k rows][D columns]: returns, for lane (r = lane&31, h = lane>>5), the 8
// values tile[16*kstep + 8*(j>>2) + 4*h + (j&3)][32*cblock + r], j = 0..7.
template <int D>
__device__ __forceinline__ bf16x8 tr_fragment(const char* tile, int kstep, int cblock, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const int q = i >> 2, p = i & 3;
  const int h = g >> 1;
  const int col = cblock * 32 + 16 * (g & 1) + 4 * p;  // first of 4 columns this lane addresses
  const int row0 = 16 * kstep + 4 * h + q;
  const int chunk = col >> 3, within = (col & 7) * 2;
  s16x4 lo = lds_read_tr(tile, tile_off<D>(row0, chunk) + within);
  s16x4 hi = lds_read_tr(tile, tile_off<D>(row0 + 8, chunk) + within);
  s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// Eight consecutive accumulator registers -> one bf16 B-operand fragment.
__device__ __forceinline__ bf16x8 pack_acc8(float a0, float a1, float a2, float a3, float a4, float a5,
                                            float a6, float a7) {
  u32x4 w = {pack_bf16x2(a0, a1), pack_bf16x2(a2, a3), pack_bf16x2(a4, a5), pack_bf16x2(a6, a7)};
  return __builtin_bit_cast(bf16x8, w);
}

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// Row index, inside a 32-row accumulator tile, of register `reg` for lane-half `h`.
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// max(a, b, c) as one v_max3_f32. The operands must be results of ordinary VALU instructions,
// NOT of an MFMA: the compiler does not insert the MFMA -> VALU wait states for inline assembly.
__device__ __forceinline__ float max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// Exchange between lane l and lane l ^ 32 with v_permlane32_swap (one VALU instruction; given the
// same value twice it returns the lower half's values in both halves and the upper half's values in
// both halves) instead of a ds_bpermute round trip through the LDS crossbar, which sits on the
// critical path between the row maximum and the exponentials of every tile.
__device__ __forceinline__ float lane_pair_max(float v) {
  const unsigned u = __float_as_uint(v);
  const auto both = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(both[0]), __uint_as_float(both[1]));
}
__device__ __forceinline__ float lane_pair_sum(float v) {
  const unsigned u = __float_as_uint(v);
  const auto both = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(both[0]) + __uint_as_float(both[1]);
}

// ------------------------------------------------------------------ attention dropout --
// Counter-based dropout mask shared by the forward and both backward kernels. The T x T
// probability matrix of one (batch, head) pair is cut into 4 x 4 blocks; one Philox-4x32
// call keyed by the launch seed with counter (query >> 2, key >> 2, batch*H + head, 0) yields
// 128 bits = one byte per element of the block: element (q, k) reads byte (k & 3) of word
// (q & 3) and is KEPT when that byte >= threshold (threshold = round(256 p), so the drop
// probability is quantised to 1/256 and the survivors are scaled by 256 / (256 - threshold)).
// A lane owns four consecutive keys of one query in the forward / dQ kernels and four
// consecutive queries of one key in the dK/dV kernel, so either orientation needs exactly
// one call per four elements. Seven rounds (the "Crush-resistant" count of the Philox paper).
// The four random words of the 4 x 4 block holding (query, key).
__device__ __forceinline__ u32x4 dropout_block(const DropoutParams& dp, int bh, int query, int key) {
  return philox4x32_7(uint32_t(query) >> 2, uint32_t(key) >> 2, uint32_t(bh), 0u, dp.seed_lo, dp.seed_hi);
}

// Keep decision of element (query, key) given its block's words.
__device__ __forceinline__ bool dropout_keep(const DropoutParams& dp, const u32x4& block, int query, int key) {
  const int qw = query & 3;
  const uint32_t word = qw == 0 ? block[0] : qw == 1 ? block[1] : qw == 2 ? block[2] : block[3];
  return ((word >> (8 * (key & 3))) & 0xffu) >= dp.threshold;
}

// ---- sharing Philox calls across a quad -------------------------------------------------
// In every kernel the four lanes of a quad (lanes 4m .. 4m+3) own four consecutive queries (or
// keys), i.e. one side of a 4 x 4 dropout block, and walk over groups of four consecutive keys
// (queries) on the other side. All four lanes therefore need the same blocks. Instead of every
// lane evaluating every block (7 Philox rounds, ~60 instructions, for one useful word), lane j
// evaluates the block of group j and the quad exchanges words with DPP moves.
// These helpers must execute with all 64 lanes active.
template <int kCtrl>
__device__ __forceinline__ uint32_t quad_perm(uint32_t v) {
  return uint32_t(__builtin_amdgcn_mov_dpp(int(v), kCtrl, 0xf, 0xf, true));
}

// 4 x 4 transpose across a quad: on entry lane j holds w[0..3]; on exit lane r holds, in w[g],
// what lane g held in w[r].
__device__ __forceinline__ void quad_transpose(uint32_t (&w)[4], int lane) {
  const bool odd = lane & 1, hi = lane & 2;
  const uint32_t a = quad_perm<0xB1>(odd ? w[0] : w[1]);  // exchange with lane ^ 1
  const uint32_t b = quad_perm<0xB1>(odd ? w[2] : w[3]);
  if (odd) {
    w[0] = a;
    w[2] = b;
  } else {
    w[1] = a;
    w[3] = b;
  }
  const uint32_t c = quad_perm<0x4E>(hi ? w[0] : w[2]);  // exchange with lane ^ 2
  const uint32_t d = quad_perm<0x4E>(hi ? w[1] : w[3]);
  if (hi) {
    w[0] = c;
    w[1] = d;
  } else {
    w[2] = c;
    w[3] = d;
  }
}

// Query-per-lane kernels (forward, dQ): this lane's query is `query` (query & 3 == lane & 3) and
// its accumulator registers 4g .. 4g+3 hold keys kfirst + 8g + 4h .. +3 for g = 0..3. Returns in
// words[g] the random word of (query, that key group): byte e decides key kfirst + 8g + 4h + e.
__device__ __forceinline__ void dropout_words_for_query(const DropoutParams& dp, int bh, int query, int kfirst,
                                                        int h, int lane, uint32_t (&words)[4]) {
  const u32x4 blk = dropout_block(dp, bh, query, kfirst + 8 * (lane & 3) + 4 * h);
  words[0] = blk[0];
  words[1] = blk[1];
  words[2] = blk[2];
  words[3] = blk[3];
  quad_transpose(words, lane);
}

// Key-per-lane kernel (dK/dV): this lane's key is `key` (key & 3 == lane & 3) and its accumulator
// registers 4g .. 4g+3 hold queries qfirst + 8g + 4h .. +3. Returns a 16-bit mask: bit 4g + e is
// set when element (query qfirst + 8g + 4h + e, key) is kept.
__device__ __forceinline__ uint32_t dropout_keep_mask_for_key(const DropoutParams& dp, int bh, int qfirst, int h,
                                                              int key, int lane) {
  const u32x4 blk = dropout_block(dp, bh, qfirst + 8 * (lane & 3) + 4 * h, key);
  const uint32_t shift = 8u * (uint32_t(key) & 3u);
  uint32_t mask = 0;
  auto take = [&](uint32_t word, int bit) {
    if (((word >> shift) & 0xffu) >= dp.threshold) mask |= 1u << bit;
  };
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    take(quad_perm<0x00>(blk[e]), e);       // group 0: lane 0's block
    take(quad_perm<0x55>(blk[e]), 4 + e);   // group 1: lane 1's block
    take(quad_perm<0xAA>(blk[e]), 8 + e);   // group 2
    take(quad_perm<0xFF>(blk[e]), 12 + e);  // group 3
  }
  return mask;
}

// Cooperative copy of a [kRows][D] bf16 tile from global memory (row stride
// `row_stride` elements, rows clamped to `last_row`) into registers and from
// registers into an LDS tile. Split in two so the global latency hides under
// compute (issue early, write late).
template <int D, int kRows, int kThreads>
struct TileStager {
  static constexpr int kChunksPerRow = D / 8;
  static constexpr int kChunks = kRows * kChunksPerRow;
  static constexpr int kPerThread = (kChunks + kThreads - 1) / kThreads;
  u32x4 regs[kPerThread];

  __device__ __forceinline__ void load(const bf16_raw* base, int64_t row_stride, int row0, int last_row,
                                       int tid) {
    if (row0 + kRows - 1 <= last_row) {
      // Whole tile in range (every tile but possibly the last): address = wave-uniform tile base +
      // a per-lane byte offset that does not depend on the tile, so the loop carries no 64-bit
      // address arithmetic on the vector unit (the clamped form below costs ~8 vector
      // instructions per load, a sixth of the forward kernel's instruction count per tile).
      const char* tile_base = reinterpret_cast<const char*>(base) + int64_t(row0) * row_stride * 2;
      const unsigned pitch = unsigned(row_stride) * 2u;
#pragma unroll
      for (int i = 0; i < kPerThread; ++i) {
        const int c = tid + i * kThreads;
        if (kChunks % kThreads == 0 || c < kChunks) {
          const unsigned off = unsigned(c / kChunksPerRow) * pitch + unsigned(c % kChunksPerRow) * 16u;
          regs[i] = *reinterpret_cast<const u32x4*>(tile_base + off);
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < kPerThread; ++i) {
      const int c = tid + i * kThreads;
      if (kChunks % kThreads == 0 || c < kChunks) {
        int row = row0 + c / kChunksPerRow;
        row = row < last_row ? row : last_row;
        regs[i] = *reinterpret_cast<const u32x4*>(base + int64_t(row) * row_stride + (c % kChunksPerRow) * 8);
      }
    }
  }
  __device__ __forceinline__ void store(char* tile, int tid) const {
#pragma unroll
    for (int i = 0; i < kPerThread; ++i) {
      const int c = tid + i * kThreads;
      if (kChunks % kThreads == 0 || c < kChunks) {
        const int row = c / kChunksPerRow, chunk = c % kChunksPerRow;
        *reinterpret_cast<u32x4*>(tile + tile_off<D>(row, chunk)) = regs[i];
      }
    }
  }
};

}  // namespace attn
}  // namespace pda
