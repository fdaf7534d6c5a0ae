// Checks three forms of the 4 x 4 transpose across a quad of lanes (builtin mov_dpp, builtin update_dpp, inline
// asm) on a 32 x 32 MFMA accumulator image: lane (r, h) register e holds element (row acc_row(e, h), column r);
// after the transpose each lane stores four consecutive columns of one row with one 16-byte store.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pytorch_distributed_amd/csrc/kernels -o tools/micro/bin/quad_transpose_check tools/micro/quad_transpose_check.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "attention_common.h"
using namespace pda;
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int kCtrl> __device__ __forceinline__ uint32_t qp_update(uint32_t v) {
  return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), kCtrl, 0xf, 0xf, true));
}
__device__ __forceinline__ uint32_t asm_xor1(uint32_t v) {
  uint32_t o;
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(o) : "v"(v));
  return o;
}
__device__ __forceinline__ uint32_t asm_xor2(uint32_t v) {
  uint32_t o;
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=&v"(o) : "v"(v));
  return o;
}
template <int kForm> __device__ __forceinline__ void transpose(uint32_t (&w)[4], int lane) {
  if (kForm == 0) { attn::quad_transpose(w, lane); return; }
  const bool odd = lane & 1, hi = lane & 2;
  const uint32_t s0 = odd ? w[0] : w[1], s1 = odd ? w[2] : w[3];
  const uint32_t a = kForm == 1 ? qp_update<0xB1>(s0) : asm_xor1(s0);
  const uint32_t b = kForm == 1 ? qp_update<0xB1>(s1) : asm_xor1(s1);
  if (odd) { w[0] = a; w[2] = b; } else { w[1] = a; w[3] = b; }
  const uint32_t s2 = hi ? w[0] : w[2], s3 = hi ? w[1] : w[3];
  const uint32_t c = kForm == 1 ? qp_update<0x4E>(s2) : asm_xor2(s2);
  const uint32_t d = kForm == 1 ? qp_update<0x4E>(s3) : asm_xor2(s3);
  if (hi) { w[0] = c; w[1] = d; } else { w[2] = c; w[3] = d; }
}

template <int kForm, bool kDirect> __global__ void k(const float* in, float* out, int J) {
  f32x16 t;
  for (int e = 0; e < 16; ++e) t[e] = in[threadIdx.x * 16 + e];
  asm volatile("" : "+v"(t));
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      w[e] = kDirect ? __builtin_bit_cast(uint32_t, in[threadIdx.x * 16 + 4 * g + e]) : __float_as_uint(float(t[4 * g + e]));
    transpose<kForm>(w, lane);
    const int i = 8 * g + 4 * h + (r & 3), j = r & ~3;
    *reinterpret_cast<u32x4*>(out + int64_t(i) * J + j) = u32x4{w[0], w[1], w[2], w[3]};
  }
}

int main() {
  const int J = 32;
  std::vector<float> in(64 * 16), out(32 * J);
  for (int lane = 0; lane < 64; ++lane)
    for (int e = 0; e < 16; ++e) {
      const int r = lane & 31, h = lane >> 5, row = (e & 3) + 8 * (e >> 2) + 4 * h;
      in[lane * 16 + e] = float(row * 100 + r);  // the value names its place
    }
  float *din, *dout;
  hipMalloc(&din, in.size() * 4); hipMalloc(&dout, out.size() * 4);
  hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
  int bad_total = 0;
  for (int form = 0; form < 6; ++form) {
    hipMemset(dout, 0, out.size() * 4);
    if (form == 0) hipLaunchKernelGGL((k<0, false>), dim3(1), dim3(64), 0, 0, din, dout, J);
    if (form == 1) hipLaunchKernelGGL((k<1, false>), dim3(1), dim3(64), 0, 0, din, dout, J);
    if (form == 2) hipLaunchKernelGGL((k<2, false>), dim3(1), dim3(64), 0, 0, din, dout, J);
    if (form == 3) hipLaunchKernelGGL((k<0, true>), dim3(1), dim3(64), 0, 0, din, dout, J);
    if (form == 4) hipLaunchKernelGGL((k<1, true>), dim3(1), dim3(64), 0, 0, din, dout, J);
    if (form == 5) hipLaunchKernelGGL((k<2, true>), dim3(1), dim3(64), 0, 0, din, dout, J);
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) bad += out[i * J + j] != float(i * 100 + j);
    if (form == 2) for (int i = 0; i < 6; ++i) { for (int j = 0; j < 8; ++j) printf("%6.0f", out[i * J + j]); printf("\n"); }
    printf("form %d (%s): %d of 1024 elements wrong\n", form, form % 3 == 0 ? "mov_dpp builtin" : form % 3 == 1 ? "update_dpp builtin" : "inline asm", bad);
    bad_total += bad;
  }
  return 0;
}
