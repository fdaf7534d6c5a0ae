// The framework's collective layer. Everything that communicates (gradient reducer,
// sharding units, trainers, checkpoints) talks to a pda::comm::Communicator; PyTorch is
// only the tensor / autograd layer underneath.
//
// Implementations:
//   RcclCommunicator    GPUs. One ncclComm per process created from a unique id that rank 0
//                       publishes in the rendezvous store; collectives run on two HIP streams
//                       the communicator owns (queue kGather for all-gathers / broadcasts,
//                       queue kReduce for all-reduces / reduce-scatters), never on the compute
//                       stream. RCCL executes the collectives of ONE communicator in the order
//                       they were issued whatever stream they are on, so the two queues do not
//                       make a gather overtake an earlier reduction; what they separate is the
//                       bookkeeping (fences, waits, trace rows) of the two traffic classes, and
//                       they are the seam where a second library communicator would go if
//                       gathers and reductions are ever to run concurrently. hipEvent fences
//                       order the queues against the caller's stream in both directions; a
//                       watchdog aborts the communicator (ncclCommAbort) when a collective
//                       exceeds the timeout.
//   SocketCommunicator  CPU tensors over a TCP mesh between the ranks: the device-free
//                       configuration and every multi-process test.
//   C10dCommunicator    adapter around an existing torch.distributed process group, for
//                       callers that bring their own.
//   RecordingCommunicator  decorator that logs (kind, elements, queue) of every collective.
//
// Ordering contract of an asynchronous collective on queue q issued from stream s:
//   1. everything enqueued on s before the call happens-before the collective;
//   2. Work::wait() makes the *current* stream wait for the collective (no host blocking on
//      GPUs; on CPUs the collective has completed when the call returns);
//   3. buffers passed in must stay allocated until wait() was called and may only be freed
//      or reused from a stream that has waited.
//
// Reference counterpart: the NCCL process group behind dist.init_process_group /
// dist.all_reduce / dist.destroy_process_group (assignments/assignment1/train_ddp.py:23-36,146;
// assignments/assignment1/train/distributed_trainer.py:131-154) and the collectives torch
// DDP / FSDP issue internally.
#pragma once

#include <ATen/ATen.h>

#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace pda {
namespace comm {

enum class RedOp : int { kSum = 0, kAvg = 1, kMax = 2 };
enum class Queue : int { kGather = 0, kReduce = 1 };
enum class Kind : int { kAllReduce = 1, kAllGather = 2, kReduceScatter = 3, kBroadcast = 4 };

class Work {
 public:
  virtual ~Work() = default;
  virtual void wait() = 0;         // current stream (GPU) / host (CPU) waits for the collective
  virtual void synchronize() = 0;  // host blocks until the collective has completed
  virtual bool completed() = 0;
};

struct Counters {
  int64_t all_reduce = 0, all_gather = 0, reduce_scatter = 0, broadcast = 0;
  int64_t bytes = 0;
};

class Communicator {
 public:
  virtual ~Communicator() = default;
  virtual int rank() const = 0;
  virtual int world() const = 0;
  virtual std::string backend() const = 0;
  virtual bool on_gpu() const = 0;

  virtual std::shared_ptr<Work> all_reduce(const at::Tensor& t, RedOp op, Queue q) = 0;
  // `out` holds world() consecutive copies of `in`'s shape; rank r's contribution lands in slot r.
  virtual std::shared_ptr<Work> all_gather(const at::Tensor& out, const at::Tensor& in, Queue q) = 0;
  // `in` holds world() consecutive chunks of `out`'s size; `out` receives the reduction of chunk rank().
  virtual std::shared_ptr<Work> reduce_scatter(const at::Tensor& out, const at::Tensor& in, RedOp op, Queue q) = 0;
  virtual std::shared_ptr<Work> broadcast(const at::Tensor& t, int root, Queue q) = 0;
  virtual void barrier() = 0;  // host-side rendezvous of all ranks (does not drain GPU streams)
  // Orderly shutdown: waits for outstanding work, releases the library communicator and the
  // streams; later collectives raise. Called by parallel.shutdown() so that nothing is torn down
  // from a destructor at interpreter exit, after the HIP runtime may already be gone.
  virtual void close() {}
  // Tears the communicator down without waiting for outstanding work (after a timeout / fault).
  virtual void abort() = 0;
  virtual bool aborted() const { return false; }

  // Small host-side agreement primitive used by the debug checks: every rank contributes a
  // byte string and receives all of them, in rank order.
  virtual std::vector<std::string> exchange(const std::string& mine) = 0;

  const Counters& counters() const { return counters_; }

 protected:
  void count(Kind k, const at::Tensor& payload) {
    std::lock_guard<std::mutex> lock(counter_mu_);
    switch (k) {
      case Kind::kAllReduce: ++counters_.all_reduce; break;
      case Kind::kAllGather: ++counters_.all_gather; break;
      case Kind::kReduceScatter: ++counters_.reduce_scatter; break;
      case Kind::kBroadcast: ++counters_.broadcast; break;
    }
    counters_.bytes += (int64_t)payload.numel() * payload.element_size();
  }
  Counters counters_;
  std::mutex counter_mu_;
};

// A collective that had completed by the time the call returned.
class CompletedWork final : public Work {
 public:
  void wait() override {}
  void synchronize() override {}
  bool completed() override { return true; }
};

void check_collective_args(const char* what, const at::Tensor& out, const at::Tensor& in, int world, bool on_gpu);

}  // namespace comm
}  // namespace pda
