// bf16 GEMM with fused element-wise epilogues for gfx950: out[M, N] = epilogue(X[M, K] . W[N, K]^T).
//
// Why a dedicated kernel. The MLP of a GPT block is c_fc -> GELU -> c_proj. With the GEMMs in a
// library the activation is a kernel of its own on either side of the MLP: forward reads u and
// writes gelu(u) (2 passes over the [tokens, 4d] activation), backward reads dgelu's two inputs and
// writes du (3 passes) — 5 x 335 MB per layer at GPT-2 Large / 32 K tokens, a third as much time
// as the c_fc GEMM itself. The library's own GELU epilogues are not usable on this stack (no bf16
// GELU_AUX algorithm; DGELU 4x slower than GEMM + separate kernel). So the two GEMMs that touch the
// activation are done here and apply it on the accumulators:
//   kEpiBiasGelu      u = acc + bias -> out (bf16),  gelu(u) -> aux       (c_fc forward)
//   kEpiMulGeluGrad   out = acc * gelu'(u), u read from aux              (c_proj input gradient,
//                                                                          on the transposed weight)
//   kEpiBias / kEpiNone  plain linear layers (used by the tests and as a planner candidate).
//
// Structure (shared with the weight-gradient kernel, gemm_wgrad.hip):
//   * a workgroup of 8 waves (2 x 4) owns a 256 x 256 output tile, each wave 128 (tokens) x 64
//     (features) as 4 x 2 accumulators of v_mfma_f32_32x32x16_bf16;
//   * both operands are contraction-contiguous, so a pipeline stage is a [256 rows][32 k] slab of
//     each (64-byte rows, 16 KiB), moved global -> LDS by global_load_lds (1 KiB per wave
//     instruction, no VGPR staging) into a 4-stage ring (128 KiB): three stages stay in flight
//     behind the one being multiplied, waits are counted (vmcnt), barriers raw;
//   * the 16-byte chunk index is XOR-swizzled with bits 2-3 of the row — applied to the per-lane
//     SOURCE address, the LDS destination of such a load being linear — which makes the
//     ds_read_b128 fragment reads (32 rows x one chunk per half-wave) bank-conflict free;
//   * the product is formed TRANSPOSED (features on the accumulator registers, tokens on the
//     lanes), so every group of four accumulator registers is four consecutive output features:
//     the epilogue reads / writes 8-byte pieces straight from registers, no LDS round trip;
//   * workgroup ids are dealt to the 8 XCDs in contiguous task ranges, and consecutive tasks walk
//     4 token tiles x all feature tiles, so the CUs of an XCD share operand panels in their L2.
//
// Needs K % 32 == 0 and N % 8 == 0; M is arbitrary (edge rows are clamped on load, masked on store).
//
// Measured (MI355X, 32768 x 5120 x 1280, one box; library = hipBLASLt's tuned algorithm):
//                          library (+ GELU kernel)   eight-wave kernel   four-wave persistent kernel
//   GEMM + bias                  306 us                   583 us                405 us
//   GEMM + bias, u and gelu(u)   439 us                   941 us                499 us
//   GEMM * gelu'(u)              508 us                     -                   489 us
// The four-wave kernel's main loop runs at ~1.15 PFLOP/s (library: 1.4). What keeps the fused forms
// from paying: the activation is ~9 VALU operations and two transcendentals per element — about
// 77 us of the whole chip's vector throughput for this tensor — and in an epilogue that work runs
// while the matrix pipe of the same CU waits, whereas the stand-alone kernel hides it under its own
// memory traffic. Overlapping it needs producer / consumer wave specialisation (MFMA waves and
// epilogue waves on one CU), which this kernel does not have. A staggered start of the persistent
// workgroups (to spread the store bursts) was measured and changed nothing. The Python side
// therefore keeps these kernels opt-in (ops/mlp.py).
#include "attention_common.h"
#include "gelu_math.h"
#include "pda_kernels.h"
#include <stdlib.h>

namespace pda {
namespace {

using namespace attn;

constexpr int kThreads = 512;
constexpr int kTile = 256;
constexpr int kBK = 32;
constexpr int kStages = 4;
constexpr int kRowBytes = kBK * 2;                 // 64
constexpr int kTileBytes = kTile * kRowBytes;      // 16 KiB per operand slab
constexpr int kStageBytes = 2 * kTileBytes;        // X slab | W slab
constexpr int kLoadsPerStage = 4;                  // global_load_lds per wave and stage (2 X + 2 W)
constexpr int kGroupM = 4;                         // token tiles walked together by consecutive tasks

__device__ __forceinline__ unsigned lds_addr(const char* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

// LDS reads go through inline asm: with global_load_lds in flight the compiler cannot prove an
// ordinary LDS read does not alias their destination and would drain the whole queue in front of
// it. Completion is awaited explicitly (counted lgkmcnt + a scheduling fence) before the MFMAs.
template <int kOff>
__device__ __forceinline__ bf16x8 frag_read(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(kOff));
  return __builtin_bit_cast(bf16x8, v);
}

// Source address of this lane's 16 bytes for the piece (wave, q) of a slab whose first row is row0:
// the wave instruction fills rows 32*wave + 16*q .. +15 (lane p -> row + p/4, slot p%4), and the
// lane fetches the chunk that belongs in its slot: slot ^ ((row >> 2) & 3).
__device__ __forceinline__ const bf16_raw* piece_src(const bf16_raw* base, int64_t ld, int row0, int rows_total, int wave,
                                                     int q, int lane) {
  const int row = 32 * wave + 16 * q + (lane >> 2);
  const int chunk = (lane & 3) ^ ((row >> 2) & 3);
  const int grow = min(row0 + row, rows_total - 1);  // edge tiles: duplicate the last row, never stored
  return base + int64_t(grow) * ld + chunk * 8;
}

__device__ __forceinline__ void load_piece(const bf16_raw* src, char* slab, int wave, int q) {
  typedef __attribute__((address_space(1))) const void gptr_t;
  typedef __attribute__((address_space(3))) void lptr_t;
  char* dst = slab + (32 * wave + 16 * q) * kRowBytes;  // wave-uniform
  __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)dst, 16, 0, 0);
}

__device__ __forceinline__ u32x2 pack4(float a, float b, float c, float d) {
  u32x2 w;
  w[0] = pack_bf16x2(a, b);
  w[1] = pack_bf16x2(c, d);
  return w;
}

template <int kEpi>
__global__ __launch_bounds__(kThreads) void gemm_fused_kernel(const bf16_raw* __restrict__ X, int64_t ldx,
                                                              const bf16_raw* __restrict__ W, int64_t ldw,
                                                              const bf16_raw* __restrict__ bias, bf16_raw* __restrict__ out,
                                                              int64_t ldo, bf16_raw* __restrict__ aux, int64_t ldaux, int M,
                                                              int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 2, wj = wave & 3;  // 2 (tokens) x 4 (features) waves: 128 x 64 per wave
  const int r = lane & 31, kg = lane >> 5;

  // ---- task mapping ---------------------------------------------------------------------
  const int tiles_m = (M + kTile - 1) / kTile, tiles_n = (N + kTile - 1) / kTile;
  const int n_tasks = tiles_m * tiles_n;
  const int q8 = n_tasks / 8, r8 = n_tasks % 8;
  const int xcd = blockIdx.x % 8, within = blockIdx.x / 8;
  const int task = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
  const int group = task / (kGroupM * tiles_n);
  const int in_group = task - group * (kGroupM * tiles_n);
  const int group_rows = min(kGroupM, tiles_m - group * kGroupM);  // the last group may be short
  const int tn = in_group / group_rows;
  const int tm = group * kGroupM + in_group % group_rows;
  const int m0 = tm * kTile, n0 = tn * kTile;

  f32x16 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  // ---- staging ----------------------------------------------------------------------------
  const bf16_raw* x_src[2];
  const bf16_raw* w_src[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    x_src[q] = piece_src(X, ldx, m0, M, wave, q, lane);
    w_src[q] = piece_src(W, ldw, n0, N, wave, q, lane);
  }
  const int n_steps = K / kBK;
  auto issue = [&](int t, int slot_idx) {
    char* slot = smem + slot_idx * kStageBytes;
    const int ko = t * kBK;  // elements along the contraction, wave-uniform
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      load_piece(x_src[q] + ko, slot, wave, q);
      load_piece(w_src[q] + ko, slot + kTileBytes, wave, q);
    }
  };
#pragma unroll
  for (int t = 0; t < kStages - 1; ++t)
    if (t < n_steps) issue(t, t);

  // ---- fragment addresses -----------------------------------------------------------------
  // Lane (r, kg) reads row r of each 32-row block, chunk 2*kstep + kg, at slot chunk ^ ((r >> 2) & 3).
  const unsigned smem_base = lds_addr(smem);
  const int f = (r >> 2) & 3;
  const unsigned x_row = (unsigned)((128 * wi + r) * kRowBytes);
  const unsigned w_row = (unsigned)(kTileBytes + (64 * wj + r) * kRowBytes);
  const unsigned s0 = (unsigned)(((kg ^ f) & 3) << 4), s1 = (unsigned)((((2 + kg) ^ f) & 3) << 4);
  const unsigned xa0 = x_row + s0, xa1 = x_row + s1, wa0 = w_row + s0, wa1 = w_row + s1;

  for (int t0 = 0; t0 < n_steps; t0 += kStages) {
#pragma unroll
    for (int u = 0; u < kStages; ++u) {
      const int t = t0 + u;
      if (t >= n_steps) break;  // workgroup-uniform
      // Retire this wave's loads of stage t, leave the younger stages in flight, then meet the other
      // waves: after the barrier every wave's part of stage t has landed AND every wave is done
      // reading stage t-1, whose slot the next issue overwrites.
      const int younger = n_steps - 1 - t;
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (t + kStages - 1 < n_steps) issue(t + kStages - 1, (u + kStages - 1) % kStages);

      const unsigned base = smem_base + u * kStageBytes;
      bf16x8 x0[4], w0[2], x1[4], w1[2];
      w0[0] = frag_read<0>(base + wa0);
      w0[1] = frag_read<2048>(base + wa0);
      x0[0] = frag_read<0>(base + xa0);
      x0[1] = frag_read<2048>(base + xa0);
      x0[2] = frag_read<4096>(base + xa0);
      x0[3] = frag_read<6144>(base + xa0);
      w1[0] = frag_read<0>(base + wa1);
      w1[1] = frag_read<2048>(base + wa1);
      x1[0] = frag_read<0>(base + xa1);
      x1[1] = frag_read<2048>(base + xa1);
      x1[2] = frag_read<4096>(base + xa1);
      x1[3] = frag_read<6144>(base + xa1);
      asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");  // k-step 0 fragments
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = mfma32(w0[b], x0[a], acc[a][b]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = mfma32(w1[b], x1[a], acc[a][b]);
      __builtin_amdgcn_s_setprio(0);
    }
  }

  // ---- epilogue ---------------------------------------------------------------------------
  // acc[a][b] register e of lane (r, h = kg): token m0 + 128*wi + 32*a + r, feature
  // n0 + 64*wj + 32*b + 8*(e >> 2) + 4*h + (e & 3): each register quad is 4 consecutive features.
  const int h = kg;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = n0 + 64 * wj + 32 * b + 8 * g + 4 * h;
      if (n >= N) continue;
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (kEpi == kEpiBias || kEpi == kEpiBiasGelu) {
        const u32x2 braw = *reinterpret_cast<const u32x2*>(bias + n);
        bv[0] = bf16_lo(braw[0]);
        bv[1] = bf16_hi(braw[0]);
        bv[2] = bf16_lo(braw[1]);
        bv[3] = bf16_hi(braw[1]);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int m = m0 + 128 * wi + 32 * a + r;
        if (m >= M) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[a][b][4 * g + e] + bv[e];
        if constexpr (kEpi == kEpiMulGeluGrad) {
          const u32x2 uraw = *reinterpret_cast<const u32x2*>(aux + int64_t(m) * ldaux + n);
          v[0] *= gelu_grad_f(bf16_lo(uraw[0]));
          v[1] *= gelu_grad_f(bf16_hi(uraw[0]));
          v[2] *= gelu_grad_f(bf16_lo(uraw[1]));
          v[3] *= gelu_grad_f(bf16_hi(uraw[1]));
        }
        const u32x2 packed = pack4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<u32x2*>(out + int64_t(m) * ldo + n) = packed;
        if constexpr (kEpi == kEpiBiasGelu) {
          // the activation of the ROUNDED pre-activation: what the backward recomputes from `out`
          const u32x2 act = pack4(gelu_f(bf16_lo(packed[0])), gelu_f(bf16_hi(packed[0])), gelu_f(bf16_lo(packed[1])),
                                  gelu_f(bf16_hi(packed[1])));
          *reinterpret_cast<u32x2*>(aux + int64_t(m) * ldaux + n) = act;
        }
      }
    }
  }
}


// ---- four-wave kernel ------------------------------------------------------------------------
// Same ring, slabs and swizzle, restructured like the four-wave weight-gradient kernel: 2 x 2 waves
// of 128 x 128 each (16 accumulator tiles in the AGPR file; a k-step needs 8 fragment reads for 16
// MFMAs instead of 6 for 8), a branch-free stage body in which the fragment reads of the next
// k-step and the global loads that refill the ring are issued in the shadow of the MFMAs, and an
// epilogue that goes through LDS: the finished tile is written as bf16 into the (now idle) ring,
// XOR-swizzled, and leaves in full 512-byte rows — 16 bytes per lane — where the bias, the
// activation and the second output are applied. Needs M % 256 == 0 and N % 256 == 0 (one per-lane
// source pointer per operand, no edge handling).
constexpr int kThreads4 = 256;
constexpr int kOutRowBytes = kTile * 2;  // a staged output row: 256 bf16

__device__ __forceinline__ void load_piece_at(const bf16_raw* src, char* slab, int row) {
  typedef __attribute__((address_space(1))) const void gptr_t;
  typedef __attribute__((address_space(3))) void lptr_t;
  __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(slab + row * kRowBytes), 16, 0, 0);
}

template <int kEpi>
__global__ __launch_bounds__(kThreads4) void gemm_fused_kernel4(const bf16_raw* __restrict__ X, int64_t ldx,
                                                                const bf16_raw* __restrict__ W, int64_t ldw,
                                                                const bf16_raw* __restrict__ bias,
                                                                bf16_raw* __restrict__ out, int64_t ldo,
                                                                bf16_raw* __restrict__ aux, int64_t ldaux, int M, int N,
                                                                int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 1, wj = wave & 1;  // 2 (tokens) x 2 (features) waves: 128 x 128 per wave
  const int r = lane & 31, kg = lane >> 5;

  // Persistent workgroups: the grid is at most one workgroup per CU and each walks its share of the
  // tiles. The global stores of a finished tile are only ISSUED by its epilogue; they drain while
  // the same workgroup is already loading and multiplying its next tile, so the output traffic
  // overlaps the matrix pipe instead of stopping it once per tile on all CUs at the same time.
  // Workgroup ids are dealt round-robin to the 8 XCDs: XCD x owns the contiguous task range
  // [lo, hi) and its workgroups stride through it together.
  const int tiles_m = M / kTile, tiles_n = N / kTile;
  const int n_tasks = tiles_m * tiles_n;
  const int q8 = n_tasks / 8, r8 = n_tasks % 8;
  const int xcd = blockIdx.x % 8, within = blockIdx.x / 8;
  const int per_xcd = (int(gridDim.x) + 7 - xcd) / 8;  // workgroups of this launch that sit on this XCD
  const int lo = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int hi = lo + (xcd < r8 ? q8 + 1 : q8);
  const int n_steps = K / kBK;
  if (n_steps <= 0) return;  // never true; removes the path that would merge 256 zero accumulators
  const unsigned smem_base = lds_addr(smem);
  const int f = (r >> 2) & 3;
  const unsigned x_row = (unsigned)((128 * wi + r) * kRowBytes);
  const unsigned w_row = (unsigned)(kTileBytes + (128 * wj + r) * kRowBytes);
  const unsigned s0 = (unsigned)(((kg ^ f) & 3) << 4), s1 = (unsigned)((((2 + kg) ^ f) & 3) << 4);
  const unsigned xa[2] = {x_row + s0, x_row + s1}, wa[2] = {w_row + s0, w_row + s1};
  const int lrow = 64 * wave + (lane >> 2);
  const int lchunk = (lane & 3) ^ ((lrow >> 2) & 3);

  for (int task = lo + within; task < hi; task += per_xcd) {
  const int group = task / (kGroupM * tiles_n);
  const int in_group = task - group * (kGroupM * tiles_n);
  const int group_rows = min(kGroupM, tiles_m - group * kGroupM);
  const int tn = in_group / group_rows;
  const int tm = group * kGroupM + in_group % group_rows;
  const int m0 = tm * kTile, n0 = tn * kTile;

  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  // ---- staging: piece i (0..7) of a stage = rows 64*wave + 16*(i & 3) .. +15 of the X slab (i < 4)
  // or of the W slab. The swizzle only looks at bits 2-3 of the row, which those offsets leave
  // alone, so one per-lane pointer per operand serves all pieces.
  const bf16_raw* x_src = X + int64_t(m0 + lrow) * ldx + lchunk * 8;
  const bf16_raw* w_src = W + int64_t(n0 + lrow) * ldw + lchunk * 8;
  auto issue_one = [&](int t, int slot_idx, int i) {
    char* slot = smem + slot_idx * kStageBytes;
    const int ko = (t < n_steps ? t : n_steps - 1) * kBK;  // past the end: re-fetch the last slab (never read)
    const int rows = 16 * (i & 3);
    if (i < 4) load_piece_at(x_src + (int64_t(rows) * ldx + ko), slot, 64 * wave + rows);
    else load_piece_at(w_src + (int64_t(rows) * ldw + ko), slot + kTileBytes, 64 * wave + rows);
  };
#pragma unroll
  for (int t = 0; t < kStages - 1; ++t)
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_one(t, t, i);

  // (fragment addresses: lane (r, kg) reads row r of a 32-row block, chunk 2*kstep + kg)
  bf16x8 xf[2][4], wf[2][4];  // [k-step][32-row block]
  // Fragment read i (0..7) of k-step ks: order x0 w0 w1 w2 w3 x1 x2 x3, the order the MFMAs need them.
  auto frag = [&](int ks, unsigned base, int i) {
    if (i == 0) xf[ks][0] = frag_read<0>(base + xa[ks]);
    if (i == 1) wf[ks][0] = frag_read<0>(base + wa[ks]);
    if (i == 2) wf[ks][1] = frag_read<2048>(base + wa[ks]);
    if (i == 3) wf[ks][2] = frag_read<4096>(base + wa[ks]);
    if (i == 4) wf[ks][3] = frag_read<6144>(base + wa[ks]);
    if (i == 5) xf[ks][1] = frag_read<2048>(base + xa[ks]);
    if (i == 6) xf[ks][2] = frag_read<4096>(base + xa[ks]);
    if (i == 7) xf[ks][3] = frag_read<6144>(base + xa[ks]);
  };

  for (int t0 = 0; t0 < n_steps; t0 += kStages) {
#pragma unroll
    for (int u = 0; u < kStages; ++u) {
      const int t = t0 + u;
      if (t >= n_steps) break;  // workgroup-uniform
      // Stage t landed (the 16 younger loads stay in flight); after the barrier everyone's part has,
      // and everyone is done reading stage t-1, whose slot is refilled below.
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const unsigned base = smem_base + u * kStageBytes;
#pragma unroll
      for (int i = 0; i < 8; ++i) frag(0, base, i);
      __builtin_amdgcn_sched_barrier(0);
      // k-step 0: MFMA (a, b) needs x[a] and w[b]; the reads return in order, so the waits count
      // down as the first row of products proceeds. In the shadow: the 8 reads of k-step 1, then
      // the 8 global loads of stage t+3.
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i == 0) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        if (i == 1) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");   // w1 (one k-step-1 read queued behind)
        if (i == 2) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");   // w2
        if (i == 3) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");   // w3
        if (i == 4) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");   // x1
        if (i == 8) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");   // x2
        if (i == 12) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");  // x3
        __builtin_amdgcn_sched_barrier(0);
        acc[i >> 2][i & 3] = mfma32(wf[0][i & 3], xf[0][i >> 2], acc[i >> 2][i & 3]);
        if (i < 8) frag(1, base, i);
        else issue_one(t + kStages - 1, (u + kStages - 1) % kStages, i - 8);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[i >> 2][i & 3] = mfma32(wf[1][i & 3], xf[1][i >> 2], acc[i >> 2][i & 3]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the tail's dummy loads
  __syncthreads();  // the ring becomes the staging tile

  // ---- epilogue, part 1: accumulators -> bf16 tile in LDS ---------------------------------------
  // acc[a][b] register e of lane (r, h = kg): token 128*wi + 32*a + r, feature
  // 128*wj + 32*b + 8*(e >> 2) + 4*h + (e & 3): a register quad is 4 consecutive features = 8 bytes.
  // Staged row = token (512 bytes), 16-byte chunk index XOR (token & 31): the 32 lanes of a half-wave
  // write 32 different rows at one feature offset -> 32 different chunk slots.
  const int h = kg;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      f32x16 v = acc[a][b];
      asm volatile("" : "+v"(v));  // one tile at a time leaves the accumulator file
      const int tok = 128 * wi + 32 * a + r;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int feat = 128 * wj + 32 * b + 8 * g + 4 * h;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (kEpi == kEpiBias || kEpi == kEpiBiasGelu) {
          const u32x2 braw = *reinterpret_cast<const u32x2*>(bias + n0 + feat);
          bv[0] = bf16_lo(braw[0]);
          bv[1] = bf16_hi(braw[0]);
          bv[2] = bf16_lo(braw[1]);
          bv[3] = bf16_hi(braw[1]);
        }
        const u32x2 packed = pack4(v[4 * g] + bv[0], v[4 * g + 1] + bv[1], v[4 * g + 2] + bv[2], v[4 * g + 3] + bv[3]);
        const int chunk = (feat >> 3) ^ (tok & 31);
        *reinterpret_cast<u32x2*>(smem + tok * kOutRowBytes + chunk * 16 + (feat & 7) * 2) = packed;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  __syncthreads();

  // ---- epilogue, part 2: full rows out; element-wise work on 8 features per lane -----------------
  // Thread -> chunk (tid & 31) of rows (tid >> 5) + 8 i: a wave instruction moves two whole rows.
  const int ch = tid & 31;
#pragma unroll 4
  for (int i = 0; i < 32; ++i) {
    const int tok = (tid >> 5) + 8 * i;
    const u32x4 raw = *reinterpret_cast<const u32x4*>(smem + tok * kOutRowBytes + ((ch ^ (tok & 31)) << 4));
    const int64_t m = m0 + tok;
    const int n = n0 + ch * 8;
    if constexpr (kEpi == kEpiMulGeluGrad) {
      const u32x4 uraw = *reinterpret_cast<const u32x4*>(aux + m * ldaux + n);
      u32x4 res;
#pragma unroll
      for (int w = 0; w < 4; ++w)
        res[w] = pack_bf16x2(bf16_lo(raw[w]) * gelu_grad_f(bf16_lo(uraw[w])), bf16_hi(raw[w]) * gelu_grad_f(bf16_hi(uraw[w])));
      *reinterpret_cast<u32x4*>(out + m * ldo + n) = res;
    } else {
      *reinterpret_cast<u32x4*>(out + m * ldo + n) = raw;
      if constexpr (kEpi == kEpiBiasGelu) {
        // the activation of the ROUNDED pre-activation: what the backward recomputes from `out`
        u32x4 act;
#pragma unroll
        for (int w = 0; w < 4; ++w) act[w] = pack_bf16x2(gelu_f(bf16_lo(raw[w])), gelu_f(bf16_hi(raw[w])));
        *reinterpret_cast<u32x4*>(aux + m * ldaux + n) = act;
      }
    }
  }
  __syncthreads();  // every wave has read its rows out of LDS: the ring may be refilled
  }  // next tile of this workgroup
}

}  // namespace

bool gemm_fused_supported(int M, int N, int K) { return M >= 1 && K >= kBK && K % kBK == 0 && N >= 8 && N % 8 == 0; }

void gemm_fused(const void* X, int64_t ldx, const void* W, int64_t ldw, const void* bias, void* out, int64_t ldo, void* aux,
                int64_t ldaux, int M, int N, int K, int epilogue, hipStream_t stream) {
  const int n_tasks = ((M + kTile - 1) / kTile) * ((N + kTile - 1) / kTile);
  static bool configured = false;
  static int waves = 4;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)gemm_fused_kernel4<kEpiNone>, hipFuncAttributeMaxDynamicSharedMemorySize, kStages * kStageBytes);
    (void)hipFuncSetAttribute((const void*)gemm_fused_kernel4<kEpiBias>, hipFuncAttributeMaxDynamicSharedMemorySize, kStages * kStageBytes);
    (void)hipFuncSetAttribute((const void*)gemm_fused_kernel4<kEpiBiasGelu>, hipFuncAttributeMaxDynamicSharedMemorySize, kStages * kStageBytes);
    (void)hipFuncSetAttribute((const void*)gemm_fused_kernel4<kEpiMulGeluGrad>, hipFuncAttributeMaxDynamicSharedMemorySize, kStages * kStageBytes);
    const char* w = getenv("PDA_FUSED_GEMM_WAVES");  // 8 selects the eight-wave kernel (A/B measurements)
    waves = w ? atoi(w) : 4;
    (void)hipFuncSetAttribute((const void*)gemm_fused_kernel<kEpiNone>, hipFuncAttributeMaxDynamicSharedMemorySize, kStages * kStageBytes);
    (void)hipFuncSetAttribute((const void*)gemm_fused_kernel<kEpiBias>, hipFuncAttributeMaxDynamicSharedMemorySize, kStages * kStageBytes);
    (void)hipFuncSetAttribute((const void*)gemm_fused_kernel<kEpiBiasGelu>, hipFuncAttributeMaxDynamicSharedMemorySize, kStages * kStageBytes);
    (void)hipFuncSetAttribute((const void*)gemm_fused_kernel<kEpiMulGeluGrad>, hipFuncAttributeMaxDynamicSharedMemorySize, kStages * kStageBytes);
    configured = true;
  }
  if (waves == 4 && M % kTile == 0 && N % kTile == 0) {
    static int cus = 0;
    if (cus == 0) {
      int dev = 0;
      (void)hipGetDevice(&dev);
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      if (cus <= 0) cus = 256;
    }
    const int grid4 = n_tasks < cus ? n_tasks : cus;  // one persistent workgroup per CU (128 KiB of LDS each)
#define PDA_LAUNCH4(EPI)                                                                                             \
  hipLaunchKernelGGL(gemm_fused_kernel4<EPI>, dim3((unsigned)grid4), dim3(kThreads4), kStages * kStageBytes, stream, \
                     (const bf16_raw*)X, ldx, (const bf16_raw*)W, ldw, (const bf16_raw*)bias, (bf16_raw*)out, ldo,     \
                     (bf16_raw*)aux, ldaux, M, N, K)
    switch (epilogue) {
      case kEpiNone: PDA_LAUNCH4(kEpiNone); break;
      case kEpiBias: PDA_LAUNCH4(kEpiBias); break;
      case kEpiBiasGelu: PDA_LAUNCH4(kEpiBiasGelu); break;
      default: PDA_LAUNCH4(kEpiMulGeluGrad); break;
    }
#undef PDA_LAUNCH4
    return;
  }
#define PDA_LAUNCH(EPI)                                                                                              \
  hipLaunchKernelGGL(gemm_fused_kernel<EPI>, dim3((unsigned)n_tasks), dim3(kThreads), kStages * kStageBytes, stream,   \
                     (const bf16_raw*)X, ldx, (const bf16_raw*)W, ldw, (const bf16_raw*)bias, (bf16_raw*)out, ldo,     \
                     (bf16_raw*)aux, ldaux, M, N, K)
  switch (epilogue) {
    case kEpiNone: PDA_LAUNCH(kEpiNone); break;
    case kEpiBias: PDA_LAUNCH(kEpiBias); break;
    case kEpiBiasGelu: PDA_LAUNCH(kEpiBiasGelu); break;
    default: PDA_LAUNCH(kEpiMulGeluGrad); break;
  }
#undef PDA_LAUNCH
}

}  // namespace pda
