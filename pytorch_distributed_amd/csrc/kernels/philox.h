// Counter-based random bits for the dropout masks of gfx950 kernels: Philox-4x32 with seven
// rounds (the "Crush-resistant" count of the Philox paper). A mask is a pure function of
// (launch seed, element coordinates), so backward kernels regenerate it instead of reading a
// stored one. Each call yields 128 bits = sixteen 8-bit uniform numbers; an element is KEPT when
// its byte >= threshold, with threshold = round(256 p): the drop probability is quantised to
// 1/256 and survivors are scaled by 256 / (256 - threshold).
#pragma once
#include "common.h"

namespace pda {
namespace attn {

struct DropoutParams {
  uint32_t seed_lo, seed_hi;
  uint32_t threshold;  // 0 disables dropout
  float keep_scale;    // 256 / (256 - threshold)
};

inline DropoutParams make_dropout_params(float p, uint64_t seed) {
  DropoutParams d;
  d.seed_lo = uint32_t(seed);
  d.seed_hi = uint32_t(seed >> 32);
  int thr = int(p * 256.f + 0.5f);
  thr = thr < 0 ? 0 : (thr > 255 ? 255 : thr);
  d.threshold = uint32_t(thr);
  d.keep_scale = 256.f / float(256 - thr);
  return d;
}

__device__ __forceinline__ u32x4 philox4x32_7(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1) {
  constexpr uint32_t kM0 = 0xD2511F53u, kM1 = 0xCD9E8D57u, kW0 = 0x9E3779B9u, kW1 = 0xBB67AE85u;
#pragma unroll
  for (int round = 0; round < 7; ++round) {
    const uint32_t hi0 = __umulhi(kM0, c0), lo0 = kM0 * c0;
    const uint32_t hi1 = __umulhi(kM1, c2), lo1 = kM1 * c2;
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += kW0;
    k1 += kW1;
  }
  return u32x4{c0, c1, c2, c3};
}

}  // namespace attn
}  // namespace pda
