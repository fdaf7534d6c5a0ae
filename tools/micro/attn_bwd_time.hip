// Times the shipped attention backward kernels (dQ, then dK/dV) stand-alone: no stamps, random-sign data.
// Build: hipcc --offload-arch=gfx950 -O3 [-D<experiment switch>] -I pytorch_distributed_amd/csrc/kernels -I pytorch_distributed_amd/csrc -o tools/micro/bin/attn_bwd_time tools/micro/attn_bwd_time.hip
// Run:   attn_bwd_time [B H T]   (default 64 20 1024, D = 64)
#include "../../pytorch_distributed_amd/csrc/kernels/attention_bwd.hip"

#include <stdio.h>
#include <algorithm>
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
    p[i] = (unsigned short)((h & 0x8000) | 0x3c00 | (h & 0x3ff));
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
  hipLaunchKernelGGL(fill_f, dim3(2048), dim3(256), 0, 0, lse, n_row, 6.0f);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::vector<float> t;
  for (int it = 0; it < 12; ++it) {
    CHECK(hipEventRecord(e0));
    pda::attention_backward(qkv, out, dout, lse, delta, dqkv, B, T, H, D, 0.125f, 0.f, 0, 0);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 2) t.push_back(ms * 1e3f);
  }
  // Bits of the result, so that two builds of the kernels can be compared (both kernels are deterministic).
  std::vector<unsigned short> hd(n_qkv);
  CHECK(hipMemcpy(hd.data(), dqkv, n_qkv * 2, hipMemcpyDeviceToHost));
  unsigned long long sum = 0, mix = 0;
  for (int64_t i = 0; i < n_qkv; ++i) { sum += hd[i]; mix = mix * 1099511628211ull + hd[i]; }
  printf("result checksum: sum %llu mix %016llx\n", sum, mix);
  std::sort(t.begin(), t.end());
  printf("B %d H %d T %d D %d: backward (dQ + dK/dV) median %.1f us, min %.1f us\n", B, H, T, D, t[t.size() / 2], t[0]);
  return 0;
}
