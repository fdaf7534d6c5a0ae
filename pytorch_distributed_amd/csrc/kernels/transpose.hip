// bf16 matrix transpose for gfx950: dst[c][r] = src[r][c].
//
// Why it exists: the input gradient of a linear layer, dx = dy . W with W stored [out, in],
// contracts over W's ROW index. The library's kernels for that layout run 5-15 % below the ones
// for two contraction-contiguous operands (measured 881-1249 vs 977-1386 TFLOP/s on GPT-2 Large
// shapes), while transposing a weight costs a few microseconds: it is a few MB against a GEMM
// that streams hundreds. So backward transposes the weight and calls the fast layout.
//
// One 256-thread workgroup moves a 64 x 64 tile: 16-byte global loads along the source rows,
// element-wise stores into a transposed LDS tile (row pitch 66 elements = 33 words, so the
// eight column groups of a wave land in different banks), 4-byte LDS reads and 16-byte global
// stores along the destination rows. Rows of dst may be padded (`ld_dst` >= `rows`, the
// padding is written as zeros up to `rows_padded`) so an odd contraction length such as
// V = 50257 becomes an aligned one.
#include "common.h"
#include "pda_kernels.h"

namespace pda {
namespace {

constexpr int kThreads = 256;
constexpr int kTile = 64;
constexpr int kPitch = kTile + 2;  // elements

__global__ __launch_bounds__(kThreads) void transpose_bf16_kernel(const bf16_raw* __restrict__ src, int64_t ld_src,
                                                                  bf16_raw* __restrict__ dst, int64_t ld_dst, int rows,
                                                                  int cols, int rows_padded) {
  __shared__ __attribute__((aligned(16))) bf16_raw tile[kTile * kPitch];  // [source column][source row]
  const int tid = threadIdx.x;
  const int r0 = blockIdx.y * kTile, c0 = blockIdx.x * kTile;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + i * kThreads;
    const int row = id >> 3, cc = (id & 7) * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r0 + row < rows && c0 + cc < cols)  // cols % 8 == 0: a chunk is entirely inside or outside
      v = *reinterpret_cast<const u32x4*>(src + int64_t(r0 + row) * ld_src + c0 + cc);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      tile[(cc + 2 * j) * kPitch + row] = bf16_raw(v[j] & 0xffffu);
      tile[(cc + 2 * j + 1) * kPitch + row] = bf16_raw(v[j] >> 16);
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + i * kThreads;
    const int oc = id >> 3, rc = (id & 7) * 8;  // destination row (source column), first of 8 source rows
    if (c0 + oc < cols && r0 + rc < rows_padded) {
      const uint32_t* p = reinterpret_cast<const uint32_t*>(tile + oc * kPitch + rc);
      u32x4 v = {p[0], p[1], p[2], p[3]};
      *reinterpret_cast<u32x4*>(dst + int64_t(c0 + oc) * ld_dst + r0 + rc) = v;
    }
  }
}

}  // namespace

void transpose_bf16(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int rows, int cols, int rows_padded,
                    hipStream_t stream) {
  if (rows <= 0 || cols <= 0) return;
  dim3 grid((unsigned)ceil_div(cols, kTile), (unsigned)ceil_div(rows_padded, kTile));
  hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(kThreads), 0, stream, (const bf16_raw*)src, ld_src,
                     (bf16_raw*)dst, ld_dst, rows, cols, rows_padded);
}

}  // namespace pda
