// Shared device helpers for the gfx950 (CDNA4) kernels: 16-byte vector I/O for
// bf16 / fp32 rows, bf16 <-> fp32 conversion, and wave64 / workgroup reductions.
//
// Everything here assumes a 64-lane wavefront; nothing is written for 32-lane
// hardware.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pda {

constexpr int kWave = 64;

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

typedef uint16_t bf16_raw;  // storage type used in kernel signatures

__device__ __forceinline__ float bf16_to_f32(bf16_raw v) {
  return __uint_as_float(((uint32_t)v) << 16);
}

// Round-to-nearest-even through the hardware convert (keeps NaN a NaN).
__device__ __forceinline__ bf16_raw f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_raw, b);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  f32x2 f = {lo, hi};
  bf16x2 b = __builtin_convertvector(f, bf16x2);
  return __builtin_bit_cast(uint32_t, b);
}

__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// ---------------------------------------------------------------------------
// VecIO<T>: one 16-byte access = kVec elements of T, widened to fp32 registers.
// ---------------------------------------------------------------------------
template <typename T>
struct VecIO;

template <>
struct VecIO<bf16_raw> {
  static constexpr int kVec = 8;
  typedef u32x4 Raw;  // the 16 bytes as loaded, for values kept packed across passes
  __device__ static __forceinline__ Raw load_raw(const bf16_raw* p) {
    return *reinterpret_cast<const u32x4*>(p);
  }
  // Non-temporal forms for data a streaming kernel touches once (no reuse to keep in the caches).
  __device__ static __forceinline__ Raw load_raw_nt(const bf16_raw* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  }
  __device__ static __forceinline__ void store_raw_nt(bf16_raw* p, const Raw& w) {
    __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(p));
  }
  __device__ static __forceinline__ void unpack(const Raw& w, float (&out)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      out[2 * i] = bf16_lo(w[i]);
      out[2 * i + 1] = bf16_hi(w[i]);
    }
  }
  __device__ static __forceinline__ void load(const bf16_raw* p, float (&out)[8]) {
    unpack(load_raw(p), out);
  }
  __device__ static __forceinline__ Raw pack(const float (&in)[8]) {
    u32x4 w;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(in[2 * i], in[2 * i + 1]);
    return w;
  }
  __device__ static __forceinline__ void store(bf16_raw* p, const float (&in)[8]) {
    *reinterpret_cast<u32x4*>(p) = pack(in);
  }
  __device__ static __forceinline__ float load1(const bf16_raw* p) { return bf16_to_f32(*p); }
  __device__ static __forceinline__ void store1(bf16_raw* p, float f) { *p = f32_to_bf16(f); }
};

template <>
struct VecIO<float> {
  static constexpr int kVec = 4;
  typedef f32x4 Raw;
  __device__ static __forceinline__ Raw load_raw(const float* p) {
    return *reinterpret_cast<const f32x4*>(p);
  }
  __device__ static __forceinline__ Raw load_raw_nt(const float* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  }
  __device__ static __forceinline__ void store_raw_nt(float* p, const Raw& w) {
    __builtin_nontemporal_store(w, reinterpret_cast<f32x4*>(p));
  }
  __device__ static __forceinline__ void unpack(const Raw& w, float (&out)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = w[i];
  }
  __device__ static __forceinline__ void load(const float* p, float (&out)[4]) {
    unpack(load_raw(p), out);
  }
  __device__ static __forceinline__ Raw pack(const float (&in)[4]) {
    f32x4 w;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = in[i];
    return w;
  }
  __device__ static __forceinline__ void store(float* p, const float (&in)[4]) {
    *reinterpret_cast<f32x4*>(p) = pack(in);
  }
  __device__ static __forceinline__ float load1(const float* p) { return *p; }
  __device__ static __forceinline__ void store1(float* p, float f) { *p = f; }
};

// ---------------------------------------------------------------------------
// Reductions. wave_* combine all 64 lanes; every lane receives the result.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// Workgroup reductions through a caller-provided LDS scratch of >= 16 floats.
// All threads of the block must call; every thread receives the result.
template <int kThreads>
__device__ __forceinline__ float block_sum(float v, float* scratch) {
  constexpr int kWaves = kThreads / kWave;
  v = wave_sum(v);
  if constexpr (kWaves == 1) return v;
  const int wid = threadIdx.x >> 6;
  __syncthreads();  // protect scratch reuse across consecutive calls
  if ((threadIdx.x & 63) == 0) scratch[wid] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < kWaves; ++i) r += scratch[i];
  return r;
}

template <int kThreads>
__device__ __forceinline__ float block_max(float v, float* scratch) {
  constexpr int kWaves = kThreads / kWave;
  v = wave_max(v);
  if constexpr (kWaves == 1) return v;
  const int wid = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[wid] = v;
  __syncthreads();
  float r = scratch[0];
#pragma unroll
  for (int i = 1; i < kWaves; ++i) r = fmaxf(r, scratch[i]);
  return r;
}

__host__ __device__ __forceinline__ int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace pda
