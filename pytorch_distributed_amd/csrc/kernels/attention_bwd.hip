// Causal attention backward for gfx950: two kernels, no atomics, no T x T tensor.
//
//   attn_bwd_dq_kernel    query-stationary. One workgroup = 128 queries (32 per wave),
//                         walks the key tiles up to the diagonal like the forward:
//                           S^T  = K . Q^T            (recompute, keys on registers)
//                           P^T  = exp2(S'^T)         (S' = c S - LSE log2 e straight from the MFMA chain:
//                                                      Q is pre-multiplied by c = scale log2 e and the row
//                                                      constant is the accumulator's initial value)
//                           dP^T = V . dO^T
//                           dS^T = P^T (dP^T - delta)
//                           dQ^T += K^T . dS^T        (K read transposed from LDS,
//                                                      dS^T straight from registers)
//                         Also produces delta = rowsum(dO * O) for the second kernel.
//   attn_bwd_dkv_kernel   key-stationary. One workgroup = 128 keys (32 per wave, K and V
//                         rows live in registers as MFMA B operands), walks the query
//                         tiles from the diagonal to the end:
//                           S' = Q . (c K)^T - LSE log2 e   (row constant preloaded in the accumulator,
//                                                             K pre-multiplied by c: P = exp2(S'))
//                           dP = dO . V^T - delta  (idem)
//                           dV^T += dO^T . P ,  dK^T += Q^T . dS
//                         Q and dO tiles are read by rows for S / dP and transposed for
//                         dV / dK from ONE swizzled LDS image each.
//
// Both kernels keep the "free index on the lane" orientation of attention_common.h, so P
// and dS feed the next MFMA from the accumulator registers and never pass through LDS.
// dK and dV of a key are summed inside one wave and dQ of a query inside one wave, so the
// result is deterministic. The price is recomputing S and dP in both kernels (7 matrix
// products instead of 5); the alternative single-kernel form needs fp32 atomics for dQ.
//
// What bounds them (tools/micro/attn_stamp.hip, s_memtime per phase of the dQ kernel, B 64 / H 20 / T 1024):
// a 64-key tile costs a wave ~3,300 cycles with a second wave beside it on the SIMD — score chains 1,270 (16
// MFMAs = 512 cycles of matrix pipe), softmax gradient 600, dQ accumulation 320, waiting for and storing the
// next tile 400-500, barrier 290-560 — and a workgroup pays 7-8,000 cycles of prologue (fragment loads, delta)
// and 2,600 of epilogue around on average nine tiles: the matrix pipe is busy 25-29 % of the time.
// The structural limit is the LDS port, not the vector ALU: with 32 queries (or keys) per wave every MFMA takes a
// fresh 1-KiB A fragment from LDS, 32 FLOPs per LDS byte, which is exactly what the CU's 128 B/cycle port
// delivers at the four matrix pipes' full rate; 24 KiB of fragment reads per wave and tile keep the port ~55 %
// busy at today's speed. Halving the bytes per FLOP takes 64 rows per wave, i.e. one wave per SIMD with the
// whole register file and a hand-placed instruction stream. A first such dK/dV kernel (two 32-key blocks per
// wave, score chains as VGPR-form inline-asm MFMAs, K / V operands in AGPRs) was correct and 15 % slower than
// the kernel below in its straightforward form; its interleaved form did not survive the compiler's register
// allocation (16-register accumulator tuples stay live whole, 28 spills). Not shipped; the analysis stands.
// The kernels keep the vector part small: both row constants ride in the accumulators' initial values and the
// softmax scale is folded into an operand once per kernel, so a probability costs one v_exp_f32, half a packed
// multiply and one or two halves of a v_cvt_pk_bf16_f32.
#include "attention_common.h"
#include "pda_kernels.h"

#include <type_traits>

namespace pda {
using attn::DropoutParams;
using attn::make_dropout_params;
namespace {

using namespace attn;

constexpr int kThreads = 256;
constexpr int kBM = 128;  // rows owned by a workgroup (queries in dq, keys in dkv)
constexpr int kBN = 64;   // rows per streamed tile (keys in dq, queries in dkv)

// Phase stamps of the dQ kernel (tools/micro/attn_stamp.hip): s_memtime around the phases of every tile,
// summed per wave. Compiled out of the shipped kernel.
#ifdef PDA_ATTN_STAMPS
__device__ unsigned long long g_attn_stamps[8192 * 4 * 10];
#define PDA_ASTAMP(x) x = __builtin_amdgcn_s_memtime()
#else
#define PDA_ASTAMP(x)
#endif

// ============================================================================ dQ
// amdgpu_waves_per_eu: with a register budget of at most 256 per wave the compiler keeps MFMA
// accumulators in ordinary VGPRs. Left at the launch-bounds default (one wave per SIMD, 512
// registers) it parks them in AGPRs and moves every S / dP / dS element across with
// v_accvgpr_read/write — 130-160 extra VALU instructions per 16 MFMAs, which made these
// kernels VALU-bound at 20 % matrix-pipe utilisation.
template <int D, bool kDrop>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(D == 64 ? 2 : 1))) void attn_bwd_dq_kernel(
    const bf16_raw* __restrict__ qkv, const bf16_raw* __restrict__ out,
    const bf16_raw* __restrict__ dout, const float* __restrict__ lse, float* __restrict__ delta_out,
    bf16_raw* __restrict__ dqkv, int B, int T, int H, float scale, DropoutParams drop) {
  constexpr int kKSteps = D / 16;
  constexpr int kDBlocks = D / 32;
  constexpr int kTileBytes = kBN * D * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][K tile | V tile]
#ifdef PDA_ATTN_STAMPS
  unsigned long long ta = 0, tb = 0, tc = 0, td = 0, te = 0, tf = 0, s_p1 = 0, s_valu = 0, s_p2 = 0, s_store = 0, s_bar = 0, n_done = 0;
  const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  const int BH = B * H;
  const int n_qt = (T + kBM - 1) / kBM;
  const int bh = blockIdx.x % BH;
  const int qt = n_qt - 1 - blockIdx.x / BH;
  const int b = bh / H, head = bh % H;
  const int64_t tok_stride = int64_t(3) * H * D;
  const int64_t o_stride = int64_t(H) * D;
  const bf16_raw* q_base = qkv + int64_t(b) * T * tok_stride + int64_t(head) * D;
  const bf16_raw* k_base = q_base + int64_t(H) * D;
  const bf16_raw* v_base = q_base + int64_t(2) * H * D;

  const int q0 = qt * kBM + wave * 32;
  const int q_row = q0 + r;
  const int q_ld = q_row < T ? q_row : T - 1;

  // The first key tile's loads go out before anything that waits for data, so that their latency runs beside
  // that of the fragment loads below instead of behind it.
  TileStager<D, kBN, kThreads> k_stage, v_stage;
  k_stage.load(k_base, tok_stride, 0, T - 1, tid);
  v_stage.load(v_base, tok_stride, 0, T - 1, tid);

  // Q and dO fragments (B operands), and delta = sum_d dO * O for this lane's query.
  bf16x8 qf[kKSteps], dof[kKSteps];
  float dpart = 0.f;
  {
    const bf16_raw* o_row = out + (int64_t(b) * T + q_ld) * o_stride + int64_t(head) * D;
    const bf16_raw* do_row = dout + (int64_t(b) * T + q_ld) * o_stride + int64_t(head) * D;
#pragma unroll
    for (int ks = 0; ks < kKSteps; ++ks) {
      qf[ks] = *reinterpret_cast<const bf16x8*>(q_base + int64_t(q_ld) * tok_stride + 16 * ks + 8 * h);
      dof[ks] = *reinterpret_cast<const bf16x8*>(do_row + 16 * ks + 8 * h);
      const bf16x8 of = *reinterpret_cast<const bf16x8*>(o_row + 16 * ks + 8 * h);
#pragma unroll
      for (int j = 0; j < 8; ++j) dpart += (float)dof[ks][j] * (float)of[j];
    }
  }
  const float delta = lane_pair_sum(dpart);
  const float lse2 = lse[(int64_t(b) * H + head) * T + q_ld] * kLog2e;
  if (h == 0 && q_row < T) delta_out[(int64_t(b) * H + head) * T + q_row] = delta;
  // The softmax works in exp2 units: P = exp2(c S - LSE log2 e), c = scale * log2 e. Both constants are
  // moved out of the tile loop: this lane's Q fragments are multiplied by c once (rounded to bf16 like
  // Q itself) and -LSE log2 e is the initial value of the S accumulator, so the MFMA chain ends as the
  // exponent and a probability costs one v_exp_f32 (a packed multiply-add per pair less per tile).
  const float c = scale * kLog2e;
#pragma unroll
  for (int ks = 0; ks < kKSteps; ++ks)
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[ks][e] = (__bf16)((float)qf[ks][e] * c);

  f32x16 dq_acc[kDBlocks];
#pragma unroll
  for (int i = 0; i < kDBlocks; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) dq_acc[i][j] = 0.f;

  const int q_tile_end = min(qt * kBM + kBM, T);
  const int n_tiles = (q_tile_end + kBN - 1) / kBN;
  const int wave_last_tile = min(q0 + 31, T - 1) / kBN;
  const bool wave_active = q0 < T;

  k_stage.store(smem, tid);
  v_stage.store(smem + kTileBytes, tid);
  __syncthreads();
#ifdef PDA_ATTN_STAMPS
  const unsigned long long t_loop = __builtin_amdgcn_s_memtime();
#endif

  for (int j = 0; j < n_tiles; ++j) {
    const char* k_tile = smem + (j & 1) * 2 * kTileBytes;
    const char* v_tile = k_tile + kTileBytes;
    const bool has_next = j + 1 < n_tiles;
    PDA_ASTAMP(ta);
    if (has_next) {
      k_stage.load(k_base, tok_stride, (j + 1) * kBN, T - 1, tid);
      v_stage.load(v_base, tok_stride, (j + 1) * kBN, T - 1, tid);
    }

    if (wave_active && j <= wave_last_tile) {  // wave-uniform
      f32x16 s_acc[2], dp_acc[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          s_acc[kb][i] = -lse2;    // row constants preloaded: the chains end as c S - LSE log2 e
          dp_acc[kb][i] = -delta;  // and dP - delta
        }
      // The 2 x 2 x kKSteps operand fragments are requested a whole half-tile ahead of the MFMA that takes them
      // (the first half's before any MFMA, the second half's one behind each MFMA of the first), so the matrix
      // pipe runs its 4 kKSteps products back to back instead of waiting out an LDS round trip before each one.
      bf16x8 frag[2][2 * kKSteps];
      auto request = [&](int kb, int i) {
        const char* tile = i < kKSteps ? k_tile : v_tile;
        frag[kb][i] = lds_read_b128(tile, tile_off<D>(kb * 32 + r, 2 * (i % kKSteps) + h));
      };
      auto multiply = [&](int kb, int i) {
        if (i < kKSteps) s_acc[kb] = mfma32(frag[kb][i], qf[i], s_acc[kb]);
        else dp_acc[kb] = mfma32(frag[kb][i], dof[i - kKSteps], dp_acc[kb]);
      };
#pragma unroll
      for (int i = 0; i < 2 * kKSteps; ++i) request(0, i);
      __builtin_amdgcn_sched_barrier(0);
#ifdef PDA_ATTN_PRIO
      __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int i = 0; i < 2 * kKSteps; ++i) {
        multiply(0, i);
        request(1, i);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < 2 * kKSteps; ++i) multiply(1, i);
#ifdef PDA_ATTN_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      __builtin_amdgcn_sched_barrier(0);
#ifdef PDA_ATTN_STAMPS
      asm volatile("" : "+v"(s_acc[0]), "+v"(s_acc[1]), "+v"(dp_acc[0]), "+v"(dp_acc[1]));  // the chains' results exist
      PDA_ASTAMP(tb);
#endif
      const int key0 = j * kBN;
      const bool diag = key0 + kBN - 1 > q0;  // wave-uniform
      // dS^T = P^T (dP^T - delta), without the softmax scale. Two elements per packed fp32
      // instruction; the causal mask is only evaluated on tiles that straddle the diagonal.
      auto softmax_grad = [&](auto masked) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            f32x2 p = {__builtin_amdgcn_exp2f(s_acc[kb][i]), __builtin_amdgcn_exp2f(s_acc[kb][i + 1])};
            if (decltype(masked)::value) {
              if (key0 + kb * 32 + acc_row(i, h) > q_row) p[0] = 0.f;
              if (key0 + kb * 32 + acc_row(i + 1, h) > q_row) p[1] = 0.f;
            }
            const f32x2 ds = p * f32x2{dp_acc[kb][i], dp_acc[kb][i + 1]};
            s_acc[kb][i] = ds[0];
            s_acc[kb][i + 1] = ds[1];
          }
      };
      if (kDrop) {
        // dP of a dropped probability is zero and of a kept one is scaled: rebuild dP from the
        // chain's (dP - delta) before the softmax gradient. The mask is the forward's.
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          uint32_t words[4];
          dropout_words_for_query(drop, bh, q_row, key0 + kb * 32, h, lane, words);
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const bool keep = ((words[g] >> (8 * e)) & 0xffu) >= drop.threshold;
              const float dp_true = dp_acc[kb][4 * g + e] + delta;
              dp_acc[kb][4 * g + e] = (keep ? dp_true * drop.keep_scale : 0.f) - delta;
            }
        }
      }
      if (__builtin_expect(diag, 0)) softmax_grad(std::true_type{});
      else softmax_grad(std::false_type{});
#ifdef PDA_ATTN_STAMPS
      asm volatile("" : "+v"(s_acc[0]), "+v"(s_acc[1]));
      PDA_ASTAMP(tc);
#endif
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const f32x16& a = s_acc[s >> 1];
        const int o = 8 * (s & 1);
        const bf16x8 dsf = pack_acc8(a[o], a[o + 1], a[o + 2], a[o + 3], a[o + 4], a[o + 5], a[o + 6], a[o + 7]);
#pragma unroll
        for (int db = 0; db < kDBlocks; ++db) {
          const bf16x8 ktf = tr_fragment<D>(k_tile, s, db, lane);
          dq_acc[db] = mfma32(ktf, dsf, dq_acc[db]);
        }
      }
#ifdef PDA_ATTN_STAMPS
      PDA_ASTAMP(td);
      s_p1 += tb - ta, s_valu += tc - tb, s_p2 += td - tc, ++n_done;
#endif
    }
#ifdef PDA_ATTN_STAMPS
    else { PDA_ASTAMP(td); }
#endif

    if (has_next) {
      char* nk = smem + ((j + 1) & 1) * 2 * kTileBytes;
      k_stage.store(nk, tid);
      v_stage.store(nk + kTileBytes, tid);
    }
    PDA_ASTAMP(te);
    __syncthreads();
#ifdef PDA_ATTN_STAMPS
    PDA_ASTAMP(tf);
    s_store += te - td, s_bar += tf - te;
#endif
  }
#ifdef PDA_ATTN_STAMPS
  const unsigned long long t_loop_end = __builtin_amdgcn_s_memtime();
#endif

  if (q_row < T) {
    bf16_raw* dq_row = dqkv + (int64_t(b) * T + q_row) * tok_stride + int64_t(head) * D;
#pragma unroll
    for (int db = 0; db < kDBlocks; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 w = {pack_bf16x2(dq_acc[db][4 * g] * scale, dq_acc[db][4 * g + 1] * scale),
                   pack_bf16x2(dq_acc[db][4 * g + 2] * scale, dq_acc[db][4 * g + 3] * scale)};
        *reinterpret_cast<u32x2*>(dq_row + db * 32 + 8 * g + 4 * h) = w;
      }
  }
#ifdef PDA_ATTN_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t_end = __builtin_amdgcn_s_memtime();
  if (lane == 0 && blockIdx.x < 8192) {
    unsigned long long* g = g_attn_stamps + (blockIdx.x * 4 + wave) * 10;
    g[0] = t_loop - t_start, g[1] = s_p1, g[2] = s_valu, g[3] = s_p2, g[4] = s_store, g[5] = s_bar;
    g[6] = t_end - t_loop_end, g[7] = n_done, g[8] = t_loop_end - t_loop, g[9] = (unsigned long long)qt;
  }
#endif
}

// ========================================================================= dK / dV
template <int D, bool kDrop>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(D == 64 ? 2 : 1))) void attn_bwd_dkv_kernel(
    const bf16_raw* __restrict__ qkv, const bf16_raw* __restrict__ dout, const float* __restrict__ lse,
    const float* __restrict__ delta, bf16_raw* __restrict__ dqkv, int B, int T, int H, float scale,
    DropoutParams drop) {
  constexpr int kKSteps = D / 16;
  constexpr int kDBlocks = D / 32;
  constexpr int kTileBytes = kBN * D * 2;
  // [2][Q tile | dO tile], then [2][64 x (-LSE log2 e) | 64 x (-delta)] fp32 row constants
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* rowc = reinterpret_cast<float*>(smem + 4 * kTileBytes);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  const int BH = B * H;
  const int bh = blockIdx.x % BH;
  const int kt = blockIdx.x / BH;  // key block 0 sees every query: heaviest first
  const int b = bh / H, head = bh % H;
  const int64_t tok_stride = int64_t(3) * H * D;
  const int64_t o_stride = int64_t(H) * D;
  const bf16_raw* q_base = qkv + int64_t(b) * T * tok_stride + int64_t(head) * D;
  const bf16_raw* k_base = q_base + int64_t(H) * D;
  const bf16_raw* v_base = q_base + int64_t(2) * H * D;
  const bf16_raw* do_base = dout + int64_t(b) * T * o_stride + int64_t(head) * D;
  const float* lse_row = lse + (int64_t(b) * H + head) * T;
  const float* delta_row = delta + (int64_t(b) * H + head) * T;

  const int key0 = kt * kBM + wave * 32;  // first key of this wave
  const int key = key0 + r;               // this lane's key
  const int key_ld = key < T ? key : T - 1;
  const bool wave_active = key0 < T;

  // K and V rows as B operands: lane (key r, h) holds row[16*ks + 8*h .. +7]. The K fragments only feed
  // the score product, so they carry the softmax's exp2 factor c = scale * log2 e (applied once here,
  // rounded to bf16 like K itself): with -LSE log2 e preloaded in the accumulator the S chain ends as
  // the exponent itself.
  const float c = scale * kLog2e;
  bf16x8 kf[kKSteps], vf[kKSteps];
#pragma unroll
  for (int ks = 0; ks < kKSteps; ++ks) {
    kf[ks] = *reinterpret_cast<const bf16x8*>(k_base + int64_t(key_ld) * tok_stride + 16 * ks + 8 * h);
    vf[ks] = *reinterpret_cast<const bf16x8*>(v_base + int64_t(key_ld) * tok_stride + 16 * ks + 8 * h);
  }
  // The first query tile's loads go out before the scaling below waits for K: the two latencies overlap.
  const int first_tile = (kt * kBM) / kBN;
  const int n_tiles = (T + kBN - 1) / kBN;
  const float* rowc_src = tid < 64 ? lse_row : delta_row;
  const float rowc_mul = tid < 64 ? -kLog2e : -1.f;
  auto load_rowc = [&](int tile) -> float {
    if (tid >= 128) return 0.f;
    int q = tile * kBN + (tid & 63);
    q = q < T ? q : T - 1;
    return rowc_src[q];
  };
  TileStager<D, kBN, kThreads> q_stage, do_stage;
  q_stage.load(q_base, tok_stride, first_tile * kBN, T - 1, tid);
  do_stage.load(do_base, o_stride, first_tile * kBN, T - 1, tid);
  float rc_stage = load_rowc(first_tile);
#pragma unroll
  for (int ks = 0; ks < kKSteps; ++ks)
#pragma unroll
    for (int e = 0; e < 8; ++e) kf[ks][e] = (__bf16)((float)kf[ks][e] * c);
  // Pin the arrival of the K / V fragments before the tile loop: waited for at their first use
  // inside it, the wait lands at the loop header, where the compiler cannot tell them from the
  // next tile's prefetch and waits for every outstanding load before the MFMAs of each tile.
#pragma unroll
  for (int ks = 0; ks < kKSteps; ++ks) {
    asm volatile("" ::"v"(__builtin_bit_cast(u32x4, kf[ks])));
    asm volatile("" ::"v"(__builtin_bit_cast(u32x4, vf[ks])));
  }
  f32x16 dk_acc[kDBlocks], dv_acc[kDBlocks];
#pragma unroll
  for (int i = 0; i < kDBlocks; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      dk_acc[i][j] = 0.f;
      dv_acc[i][j] = 0.f;
    }

  q_stage.store(smem, tid);
  do_stage.store(smem + kTileBytes, tid);
  if (tid < 128) rowc[tid] = rc_stage * rowc_mul;
  __syncthreads();

  for (int j = first_tile; j < n_tiles; ++j) {
    const int buf = (j - first_tile) & 1;
    const char* q_tile = smem + buf * 2 * kTileBytes;
    const char* do_tile = q_tile + kTileBytes;
    const bool has_next = j + 1 < n_tiles;
    const float* rc = rowc + buf * 128;
    if (has_next) {
      q_stage.load(q_base, tok_stride, (j + 1) * kBN, T - 1, tid);
      do_stage.load(do_base, o_stride, (j + 1) * kBN, T - 1, tid);
      rc_stage = load_rowc(j + 1);
    }

    if (wave_active) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {      // two 32-query sub-tiles
        const int qs = j * kBN + sub * 32;     // first query of the sub-tile
        if (qs + 31 < key0 || qs >= T) continue;  // wave-uniform: entirely above the diagonal / past the end
        // Row constants of the 16 accumulator rows this lane holds (queries qs + acc_row(i, h)).
        f32x16 s_acc, dp_acc;
        float neg_delta[kDrop ? 16 : 1];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          // chains end as c S - LSE log2 e (so exp2(acc) = P) and dP - delta
          const f32x4 a = *reinterpret_cast<const f32x4*>(rc + sub * 32 + 8 * g + 4 * h);
          const f32x4 d = *reinterpret_cast<const f32x4*>(rc + 64 + sub * 32 + 8 * g + 4 * h);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s_acc[4 * g + e] = a[e];
            dp_acc[4 * g + e] = d[e];
            if (kDrop) neg_delta[4 * g + e] = d[e];
          }
        }
#pragma unroll
        for (int ks = 0; ks < kKSteps; ++ks) {
          const bf16x8 qa = lds_read_b128(q_tile, tile_off<D>(sub * 32 + r, 2 * ks + h));
          s_acc = mfma32(qa, kf[ks], s_acc);
        }
#pragma unroll
        for (int ks = 0; ks < kKSteps; ++ks) {
          const bf16x8 da = lds_read_b128(do_tile, tile_off<D>(sub * 32 + r, 2 * ks + h));
          dp_acc = mfma32(da, vf[ks], dp_acc);
        }
        const bool diag = qs < key0 + 31;          // wave-uniform: some (q, key) pairs are masked
        const bool tail = qs + 31 >= T;            // wave-uniform: padding queries in this sub-tile
        // P and dS = P (dP - delta), two elements per packed fp32 instruction. Only tiles on the
        // diagonal or past the end of the sequence evaluate the mask (separate code paths:
        // folded into one, the per-element compares ran on every tile).
        // Dropout: the forward multiplied V by keep * P / (1 - p). dV takes that dropped P; dS is
        // the undropped P times (keep * dP / (1 - p) - delta). keep_mask bit i = element i kept.
        uint32_t keep_mask = 0xffffu;
        if (kDrop) {
          keep_mask = dropout_keep_mask_for_key(drop, bh, qs, h, key, lane);
        }
        auto softmax_grad = [&](auto masked) {
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            f32x2 p = {__builtin_amdgcn_exp2f(s_acc[i]), __builtin_amdgcn_exp2f(s_acc[i + 1])};
            if (decltype(masked)::value) {
              const int q = qs + acc_row(i, h);
              if (q < key || q >= T) p[0] = 0.f;
              if (q + 1 < key || q + 1 >= T) p[1] = 0.f;
            }
            if (kDrop) {
              const bool k0 = (keep_mask >> i) & 1, k1 = (keep_mask >> (i + 1)) & 1;
              const float dp0 = (k0 ? (dp_acc[i] - neg_delta[i]) * drop.keep_scale : 0.f) + neg_delta[i];
              const float dp1 = (k1 ? (dp_acc[i + 1] - neg_delta[i + 1]) * drop.keep_scale : 0.f) + neg_delta[i + 1];
              dp_acc[i] = p[0] * dp0;
              dp_acc[i + 1] = p[1] * dp1;
              s_acc[i] = k0 ? p[0] * drop.keep_scale : 0.f;
              s_acc[i + 1] = k1 ? p[1] * drop.keep_scale : 0.f;
            } else {
              const f32x2 ds = p * f32x2{dp_acc[i], dp_acc[i + 1]};
              s_acc[i] = p[0];
              s_acc[i + 1] = p[1];
              dp_acc[i] = ds[0];
              dp_acc[i + 1] = ds[1];
            }
          }
        };
        if (__builtin_expect(diag || tail, 0)) softmax_grad(std::true_type{});
        else softmax_grad(std::false_type{});
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int o = 8 * s;
          const bf16x8 pf = pack_acc8(s_acc[o], s_acc[o + 1], s_acc[o + 2], s_acc[o + 3], s_acc[o + 4],
                                      s_acc[o + 5], s_acc[o + 6], s_acc[o + 7]);
          const bf16x8 dsf = pack_acc8(dp_acc[o], dp_acc[o + 1], dp_acc[o + 2], dp_acc[o + 3], dp_acc[o + 4],
                                       dp_acc[o + 5], dp_acc[o + 6], dp_acc[o + 7]);
#pragma unroll
          for (int db = 0; db < kDBlocks; ++db) {
            const bf16x8 dot = tr_fragment<D>(do_tile, 2 * sub + s, db, lane);
            dv_acc[db] = mfma32(dot, pf, dv_acc[db]);
            const bf16x8 qt = tr_fragment<D>(q_tile, 2 * sub + s, db, lane);
            dk_acc[db] = mfma32(qt, dsf, dk_acc[db]);
          }
        }
      }
    }

    if (has_next) {
      char* nq = smem + (buf ^ 1) * 2 * kTileBytes;
      q_stage.store(nq, tid);
      do_stage.store(nq + kTileBytes, tid);
      if (tid < 128) rowc[(buf ^ 1) * 128 + tid] = rc_stage * rowc_mul;
    }
    __syncthreads();
  }

  if (key < T) {
    bf16_raw* dk_row = dqkv + (int64_t(b) * T + key) * tok_stride + int64_t(H) * D + int64_t(head) * D;
    bf16_raw* dv_row = dk_row + int64_t(H) * D;
#pragma unroll
    for (int db = 0; db < kDBlocks; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 wk = {pack_bf16x2(dk_acc[db][4 * g] * scale, dk_acc[db][4 * g + 1] * scale),
                    pack_bf16x2(dk_acc[db][4 * g + 2] * scale, dk_acc[db][4 * g + 3] * scale)};
        u32x2 wv = {pack_bf16x2(dv_acc[db][4 * g], dv_acc[db][4 * g + 1]),
                    pack_bf16x2(dv_acc[db][4 * g + 2], dv_acc[db][4 * g + 3])};
        *reinterpret_cast<u32x2*>(dk_row + db * 32 + 8 * g + 4 * h) = wk;
        *reinterpret_cast<u32x2*>(dv_row + db * 32 + 8 * g + 4 * h) = wv;
      }
  }
}

}  // namespace

void attention_backward(const void* qkv, const void* out, const void* dout, const float* lse,
                        float* delta, void* dqkv, int B, int T, int H, int D, float softmax_scale,
                        float dropout_p, uint64_t seed, hipStream_t stream) {
  const int n_bt = (T + kBM - 1) / kBM;
  dim3 grid((unsigned)(n_bt * B * H)), block(kThreads);
  const DropoutParams drop = make_dropout_params(dropout_p, seed);
  const size_t lds = size_t(2) * 2 * kBN * D * 2;
  auto launch = [&](auto dq_kern, auto dkv_kern) {
    hipLaunchKernelGGL(dq_kern, grid, block, lds, stream, (const bf16_raw*)qkv, (const bf16_raw*)out,
                       (const bf16_raw*)dout, lse, delta, (bf16_raw*)dqkv, B, T, H, softmax_scale, drop);
    hipLaunchKernelGGL(dkv_kern, grid, block, lds + 1024, stream, (const bf16_raw*)qkv, (const bf16_raw*)dout, lse,
                       delta, (bf16_raw*)dqkv, B, T, H, softmax_scale, drop);
  };
  if (D == 64) {
    if (drop.threshold) launch(attn_bwd_dq_kernel<64, true>, attn_bwd_dkv_kernel<64, true>);
    else launch(attn_bwd_dq_kernel<64, false>, attn_bwd_dkv_kernel<64, false>);
  } else {
    if (drop.threshold) launch(attn_bwd_dq_kernel<128, true>, attn_bwd_dkv_kernel<128, true>);
    else launch(attn_bwd_dq_kernel<128, false>, attn_bwd_dkv_kernel<128, false>);
  }
}

// keep[b][h][q][k] = 1 where the dropout mask keeps probability (q, k): the mask the kernels
// apply, materialised for tests.
namespace {
__global__ void dropout_mask_kernel(uint8_t* __restrict__ keep, int BH, int T, DropoutParams drop) {
  const int64_t n = int64_t(BH) * T * T;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const int k = int(i % T), q = int((i / T) % T), bh = int(i / (int64_t(T) * T));
    const u32x4 blk = attn::dropout_block(drop, bh, q, k);
    keep[i] = attn::dropout_keep(drop, blk, q, k) ? 1 : 0;
  }
}
}  // namespace

void attention_dropout_mask(uint8_t* keep, int B, int T, int H, float dropout_p, uint64_t seed, hipStream_t stream) {
  const DropoutParams drop = make_dropout_params(dropout_p, seed);
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(1024), dim3(256), 0, stream, keep, B * H, T, drop);
}

}  // namespace pda
