// Token + position embedding for gfx950: fused lookup forward, deterministic scatter backward.
//
// Forward   out[n, :] = wte[ids[n], :] (+ wpe[n mod T, :]) in one pass (16-byte accesses); the
//           reference does two gathers and an add (assignments/assignment1/model/my_gpt2.py:166-172).
// Backward  dwte[v, :] (+)= sum over the tokens n with ids[n] == v of dout[n, :]
//           dwpe[t, :] (+)= sum over the batch of dout[b*T + t, :]
//
// The token-embedding gradient is a scatter-add with collisions. It is computed without float
// atomics, so it is bit-reproducible, and with launches whose shape depends on the token COUNT
// only, never on the token values, so it can be recorded into a HIP graph and replayed with
// different tokens:
//   1. keys = ids (as int32), values = 0..N-1; stable radix sort by key (rocPRIM; the key is
//      sorted over ceil(log2 V) bits only). Equal tokens end up adjacent, in position order.
//   2. one wave per sorted slot; the wave at the head of a run of equal keys sums the run's
//      rows of dout in fp32, in order, and writes (or adds into) row `key` of dwte. Every row
//      has exactly one writer.
// The gradient of the tied LM head is already in the buffer when this runs (the LM head is the
// first thing backward visits, the embedding the last), which is what `accumulate` is for.
#include "common.h"
#include "pda_kernels.h"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

namespace pda {
namespace {

constexpr int kThreads = 256;

template <typename T>
__global__ __launch_bounds__(kThreads) void embed_fwd_kernel(const int64_t* __restrict__ ids, const T* __restrict__ wte,
                                                             const T* __restrict__ wpe, T* __restrict__ out,
                                                             int64_t n_tokens, int T_len, int d, int vocab) {
  constexpr int kVec = VecIO<T>::kVec;
  const int dv = d / kVec;
  const int64_t total = n_tokens * dv;
  const int64_t stride = int64_t(gridDim.x) * kThreads;
  for (int64_t v = int64_t(blockIdx.x) * kThreads + threadIdx.x; v < total; v += stride) {
    const int64_t n = v / dv;
    const int c = (int)(v - n * dv) * kVec;
    int64_t id = ids[n];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // out-of-range ids are clamped: checking them on the host would cost a device synchronisation
    float a[kVec];
    VecIO<T>::load(wte + id * d + c, a);
    if (wpe != nullptr) {
      float p[kVec];
      VecIO<T>::load(wpe + int64_t(n % T_len) * d + c, p);
#pragma unroll
      for (int k = 0; k < kVec; ++k) a[k] += p[k];
    }
    VecIO<T>::store(out + n * d + c, a);
  }
}

__global__ __launch_bounds__(kThreads) void make_keys_kernel(const int64_t* __restrict__ ids, int* __restrict__ keys,
                                                             int* __restrict__ vals, int n, int vocab) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  int64_t id = ids[i];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  keys[i] = (int)id;
  vals[i] = i;
}

// One wave per sorted slot; the wave whose slot starts a run of equal keys reduces the run.
template <typename T>
__global__ __launch_bounds__(kThreads) void embed_bwd_wte_kernel(const int* __restrict__ keys, const int* __restrict__ vals,
                                                                 const T* __restrict__ dout, T* __restrict__ dwte, int n,
                                                                 int d, bool accumulate) {
  constexpr int kVec = VecIO<T>::kVec;
  const int slot = blockIdx.x * (kThreads / kWave) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (slot >= n) return;
  const int key = keys[slot];
  if (slot > 0 && keys[slot - 1] == key) return;  // not the head of its run (wave-uniform)
  int end = slot + 1;
  while (end < n && keys[end] == key) ++end;
  T* row = dwte + int64_t(key) * d;
  for (int c = lane * kVec; c < d; c += kWave * kVec) {
    float acc[kVec];
#pragma unroll
    for (int k = 0; k < kVec; ++k) acc[k] = 0.f;
    for (int j = slot; j < end; ++j) {
      float g[kVec];
      VecIO<T>::load(dout + int64_t(vals[j]) * d + c, g);
#pragma unroll
      for (int k = 0; k < kVec; ++k) acc[k] += g[k];
    }
    if (accumulate) {
      float old[kVec];
      VecIO<T>::load(row + c, old);
#pragma unroll
      for (int k = 0; k < kVec; ++k) acc[k] += old[k];
    }
    VecIO<T>::store(row + c, acc);
  }
}

// dwpe[t, c..c+kVec) = sum_b dout[b, t, c..]; one thread per (t, chunk), batch summed in order.
template <typename T>
__global__ __launch_bounds__(kThreads) void embed_bwd_wpe_kernel(const T* __restrict__ dout, T* __restrict__ dwpe,
                                                                 int batch, int T_len, int d, bool accumulate) {
  constexpr int kVec = VecIO<T>::kVec;
  const int dv = d / kVec;
  const int64_t v = int64_t(blockIdx.x) * kThreads + threadIdx.x;
  if (v >= int64_t(T_len) * dv) return;
  const int t = (int)(v / dv);
  const int c = (int)(v - int64_t(t) * dv) * kVec;
  float acc[kVec];
#pragma unroll
  for (int k = 0; k < kVec; ++k) acc[k] = 0.f;
  for (int b = 0; b < batch; ++b) {
    float g[kVec];
    VecIO<T>::load(dout + (int64_t(b) * T_len + t) * d + c, g);
#pragma unroll
    for (int k = 0; k < kVec; ++k) acc[k] += g[k];
  }
  if (accumulate) {
    float old[kVec];
    VecIO<T>::load(dwpe + int64_t(t) * d + c, old);
#pragma unroll
    for (int k = 0; k < kVec; ++k) acc[k] += old[k];
  }
  VecIO<T>::store(dwpe + int64_t(t) * d + c, acc);
}

unsigned key_bits(int vocab) {
  unsigned b = 1;
  while ((1ll << b) < vocab) ++b;
  return b;
}

size_t align256(size_t v) { return (v + 255) / 256 * 256; }

size_t sort_temp_bytes(int n, int vocab) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const int*)nullptr, (int*)nullptr, (const int*)nullptr, (int*)nullptr,
                                  (unsigned)n, 0u, key_bits(vocab), (hipStream_t) nullptr, false);
  return bytes;
}

}  // namespace

void embedding_forward(DType dtype, const int64_t* ids, const void* wte, const void* wpe, void* out, int64_t n_tokens,
                       int T_len, int d, int vocab, hipStream_t stream) {
  if (n_tokens <= 0) return;
  const int vec = dtype == kBF16 ? 8 : 4;
  int64_t blocks = ceil_div(n_tokens * (d / vec), kThreads);
  if (blocks > 8192) blocks = 8192;
  if (dtype == kBF16)
    hipLaunchKernelGGL(embed_fwd_kernel<bf16_raw>, dim3((unsigned)blocks), dim3(kThreads), 0, stream, ids,
                       (const bf16_raw*)wte, (const bf16_raw*)wpe, (bf16_raw*)out, n_tokens, T_len, d, vocab);
  else
    hipLaunchKernelGGL(embed_fwd_kernel<float>, dim3((unsigned)blocks), dim3(kThreads), 0, stream, ids, (const float*)wte,
                       (const float*)wpe, (float*)out, n_tokens, T_len, d, vocab);
}

size_t embedding_backward_workspace_bytes(int n_tokens, int vocab) {
  return 4 * align256(sizeof(int) * (size_t)n_tokens) + align256(sort_temp_bytes(n_tokens, vocab));
}

void embedding_backward_wte(DType dtype, const int64_t* ids, const void* dout, void* dwte, bool accumulate, int n_tokens,
                            int d, int vocab, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (n_tokens <= 0) return;
  const size_t seg = align256(sizeof(int) * (size_t)n_tokens);
  char* ws = static_cast<char*>(workspace);
  int* keys_in = reinterpret_cast<int*>(ws);
  int* keys_out = reinterpret_cast<int*>(ws + seg);
  int* vals_in = reinterpret_cast<int*>(ws + 2 * seg);
  int* vals_out = reinterpret_cast<int*>(ws + 3 * seg);
  void* temp = ws + 4 * seg;
  size_t temp_bytes = workspace_bytes - 4 * seg;
  hipLaunchKernelGGL(make_keys_kernel, dim3((unsigned)ceil_div(n_tokens, kThreads)), dim3(kThreads), 0, stream, ids,
                     keys_in, vals_in, n_tokens, vocab);
  (void)rocprim::radix_sort_pairs(temp, temp_bytes, (const int*)keys_in, keys_out, (const int*)vals_in, vals_out,
                                  (unsigned)n_tokens, 0u, key_bits(vocab), stream, false);
  const unsigned blocks = (unsigned)ceil_div(n_tokens, kThreads / kWave);
  if (dtype == kBF16)
    hipLaunchKernelGGL(embed_bwd_wte_kernel<bf16_raw>, dim3(blocks), dim3(kThreads), 0, stream, keys_out, vals_out,
                       (const bf16_raw*)dout, (bf16_raw*)dwte, n_tokens, d, accumulate);
  else
    hipLaunchKernelGGL(embed_bwd_wte_kernel<float>, dim3(blocks), dim3(kThreads), 0, stream, keys_out, vals_out,
                       (const float*)dout, (float*)dwte, n_tokens, d, accumulate);
}

void embedding_backward_wpe(DType dtype, const void* dout, void* dwpe, bool accumulate, int batch, int T_len, int d,
                            hipStream_t stream) {
  if (batch <= 0 || T_len <= 0) return;
  const int vec = dtype == kBF16 ? 8 : 4;
  const unsigned blocks = (unsigned)ceil_div(int64_t(T_len) * (d / vec), kThreads);
  if (dtype == kBF16)
    hipLaunchKernelGGL(embed_bwd_wpe_kernel<bf16_raw>, dim3(blocks), dim3(kThreads), 0, stream, (const bf16_raw*)dout,
                       (bf16_raw*)dwpe, batch, T_len, d, accumulate);
  else
    hipLaunchKernelGGL(embed_bwd_wpe_kernel<float>, dim3(blocks), dim3(kThreads), 0, stream, (const float*)dout,
                       (float*)dwpe, batch, T_len, d, accumulate);
}

}  // namespace pda
