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
#include <hip/hip_runtime_api.h>
#include <torch/csrc/distributed/c10d/ProcessGroup.hpp>

#include <cstdlib>
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

enum Kind : int64_t { kAllReduce = 1, kAllGather = 2, kReduceScatter = 3 };

// Cross-rank agreement check; `like` supplies the device the communicator works on.
inline void before(const c10::intrusive_ptr<c10d::ProcessGroup>& pg, Kind kind, int64_t numel, int64_t& sequence,
                   const at::Tensor& like, const char* who) {
  static int64_t process_sequence = 0;  // one numbering for every collective of the process
  sequence = ++process_sequence;
  if (!check_enabled() || !pg || pg->getSize() <= 1) return;
  at::Tensor mine = at::tensor({sequence, (int64_t)kind, numel}, at::TensorOptions().dtype(at::kLong)).to(like.device());
  std::vector<at::Tensor> outs;
  for (int r = 0; r < pg->getSize(); ++r) outs.push_back(at::empty_like(mine));
  std::vector<std::vector<at::Tensor>> out_lists{outs};
  std::vector<at::Tensor> ins{mine};
  pg->allgather(out_lists, ins)->wait();
  const at::Tensor mine_cpu = mine.cpu();
  for (int r = 0; r < pg->getSize(); ++r) {
    const at::Tensor other = out_lists[0][r].cpu();
    TORCH_CHECK(at::equal(other, mine_cpu), who, ": collective mismatch between rank ", pg->getRank(), " and rank ", r,
                " — this rank is about to issue (sequence ", sequence, ", kind ", (int64_t)kind, ", elements ", numel,
                "), the other (sequence ", other[0].item<int64_t>(), ", kind ", other[1].item<int64_t>(), ", elements ",
                other[2].item<int64_t>(), ")");
  }
}

inline void after(const c10::intrusive_ptr<c10d::Work>& work, const at::Tensor& like) {
  if (!serialize_enabled() || !work) return;
  work->wait();
  if (like.is_cuda()) (void)hipDeviceSynchronize();
}

}  // namespace comm_debug
}  // namespace pda
