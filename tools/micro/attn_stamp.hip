// Where the attention dQ kernel spends its time: compiles the shipped kernel source with PDA_ATTN_STAMPS
// (s_memtime around the phases of every key tile, summed per wave) and prints mean ticks per tile and
// wave in each phase, per query-block index (a causal query block qt visits 2 (qt + 1) key tiles).
// Build: hipcc --offload-arch=gfx950 -O3 -I pytorch_distributed_amd/csrc/kernels -o tools/micro/bin/attn_stamp tools/micro/attn_stamp.hip
// Run:   attn_stamp [B H T]   (default 64 20 1024, D = 64: the GPT-2 large step)
#define PDA_ATTN_STAMPS 1
#include "../../pytorch_distributed_amd/csrc/kernels/attention_bwd.hip"

#include <stdio.h>
#include <vector>

#define CHECK(x)                                                                \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

__global__ void fill(unsigned short* p, int64_t n, unsigned seed) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15;
    p[i] = (unsigned short)((h & 0x8000) | 0x3c00 | (h & 0x3ff));  // bf16 of magnitude ~0.008-0.016, random sign
  }
}
__global__ void fill_f(float* p, int64_t n, float v) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) p[i] = v;
}

int main(int argc, char** argv) {
  const int B = argc > 3 ? atoi(argv[1]) : 64, H = argc > 3 ? atoi(argv[2]) : 20, T = argc > 3 ? atoi(argv[3]) : 1024, D = 64;
  const int64_t n_qkv = int64_t(B) * T * 3 * H * D, n_o = int64_t(B) * T * H * D, n_row = int64_t(B) * H * T;
  unsigned short *qkv, *out, *dout, *dqkv;
  float *lse, *delta;
  CHECK(hipMalloc(&qkv, n_qkv * 2)); CHECK(hipMalloc(&dqkv, n_qkv * 2));
  CHECK(hipMalloc(&out, n_o * 2)); CHECK(hipMalloc(&dout, n_o * 2));
  CHECK(hipMalloc(&lse, n_row * 4)); CHECK(hipMalloc(&delta, n_row * 4));
  hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, qkv, n_qkv, 1u);
  hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, out, n_o, 2u);
  hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, dout, n_o, 3u);
  hipLaunchKernelGGL(fill_f, dim3(2048), dim3(256), 0, 0, lse, n_row, 6.0f);  // ~ln(T): probabilities stay finite
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float ms_dq = 0.f;
  for (int it = 0; it < 3; ++it) {
    CHECK(hipEventRecord(e0));
    pda::attention_backward(qkv, out, dout, lse, delta, dqkv, B, T, H, D, 0.125f, 0.f, 0, 0);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms_dq, e0, e1));
  }
  printf("B %d H %d T %d D %d: dQ + dK/dV kernels %.1f us (stamped build: slower than the shipped one)\n", B, H, T, D, ms_dq * 1e3);
  const int n_wg = (T + 127) / 128 * B * H, shown = n_wg < 8192 ? n_wg : 8192;
  std::vector<unsigned long long> st(size_t(shown) * 4 * 10);
  CHECK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(pda::g_attn_stamps), st.size() * 8));
  // per query-block index: means over its waves
  const int n_qt = (T + 127) / 128;
  printf("%4s %6s | %9s | per tile: %8s %8s %8s %8s %8s | %9s | %9s\n", "qt", "tiles", "prologue", "S,dP", "softmax", "dQ",
         "stage", "barrier", "epilogue", "loop/tile");
  for (int qt = n_qt - 1; qt >= 0; --qt) {
    double sum[10] = {0}; long waves = 0;
    for (int w = 0; w < shown * 4; ++w) {
      const unsigned long long* g = &st[size_t(w) * 10];
      if ((int)g[9] != qt || g[7] == 0) continue;
      for (int k = 0; k < 9; ++k) sum[k] += (double)g[k];
      ++waves;
    }
    if (!waves) continue;
    const double tiles = sum[7] / waves, n_loop = 2.0 * (qt + 1);
    printf("%4d %6.1f | %9.0f | %18.0f %8.0f %8.0f %8.0f %8.0f | %9.0f | %9.0f\n", qt, tiles, sum[0] / waves, sum[1] / sum[7],
           sum[2] / sum[7], sum[3] / sum[7], sum[4] / waves / n_loop, sum[5] / waves / n_loop, sum[6] / waves,
           sum[8] / waves / n_loop);
  }
  return 0;
}
