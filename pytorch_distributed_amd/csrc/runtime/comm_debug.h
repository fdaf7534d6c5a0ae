// Debug aids for the bulk collectives issued by the native runtime (gradient reducer, sharding
// units). Both are off unless asked for through the environment:
//
//   PDA_COMM_DEBUG=1      before a collective is launched, every rank publishes
//                         (sequence number, kind, element count) and checks that all ranks are
//                         about to issue the same one — a rank that took a different branch
//                         (different bucket order, a skipped unit) raises an error naming the
//                         mismatch instead of hanging or silently reducing different tensors.
//   PDA_COMM_SERIALIZE=1  every collective is waited for (stream and host) right after its
//                         launch, which removes all overlap with compute: if a wrong result
//                         disappears under this flag the bug is an ordering / overlap bug.
#pragma once

#include <ATen/ATen.h>
#include <c10/util/Exception.h>

#include "comm/communicator.h"

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace pda {
namespace comm_debug {

inline bool env_flag(const char* name) {
  const char* v = std::getenv(name);
  return v != nullptr && v[0] == '1';
}

inline bool check_enabled() {
  static const bool on = env_flag("PDA_COMM_DEBUG");
  return on;
}

inline bool serialize_enabled() {
  static const bool on = env_flag("PDA_COMM_SERIALIZE");
  return on;
}

// Cross-rank agreement check (host side, through the communicator's exchange primitive).
inline void before(const std::shared_ptr<comm::Communicator>& c, comm::Kind kind, int64_t numel, const char* who) {
  static int64_t process_sequence = 0;  // one numbering for every collective of the process
  const int64_t sequence = ++process_sequence;
  if (!check_enabled() || !c || c->world() <= 1) return;
  int64_t mine[3] = {sequence, (int64_t)kind, numel};
  const std::string bytes(reinterpret_cast<const char*>(mine), sizeof(mine));
  const std::vector<std::string> all = c->exchange(bytes);
  for (int r = 0; r < (int)all.size(); ++r) {
    int64_t other[3] = {0, 0, 0};
    if (all[r].size() == sizeof(other)) std::memcpy(other, all[r].data(), sizeof(other));
    TORCH_CHECK(all[r] == bytes, who, ": collective mismatch between rank ", c->rank(), " and rank ", r,
                " — this rank is about to issue (sequence ", sequence, ", kind ", (int64_t)kind, ", elements ", numel,
                "), the other (sequence ", other[0], ", kind ", other[1], ", elements ", other[2], ")");
  }
}

inline void after(const std::shared_ptr<comm::Work>& work) {
  if (!serialize_enabled() || !work) return;
  work->wait();
  work->synchronize();
}

}  // namespace comm_debug
}  // namespace pda
