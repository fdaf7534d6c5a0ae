// RCCL communicator: the GPU implementation of pda::comm::Communicator (see communicator.h).
//
// Streams. The communicator owns two HIP streams, one per Queue. Together with the caller's
// compute stream that is three of the four hardware queues a process gets, so collectives never
// share a hardware queue with compute kernels.
//
// Library communicators. One ncclComm serves both queues by default: RCCL then runs the
// collectives in issue order whatever stream they are on, so a backward all-gather prefetch
// queues behind the reduce-scatter issued before it. With `dual` a SECOND ncclComm (its own
// unique id through the store) serves the gather queue and the two traffic classes really run
// concurrently; whether gathers use it is a run-time switch (set_dual_route) so the start-up
// autotuner can time an all-gather behind a large reduce-scatter both ways and keep the faster
// (parallel/autotune.py). Every rank issues the same collectives in the same order on each
// communicator, which is what makes two communicators on two streams safe.
//
// Fences. issue(): an event recorded on the caller's stream is awaited by the queue's stream
// (the collective sees every kernel enqueued before it), the RCCL call is enqueued, and a
// completion event is recorded behind it. Work::wait() makes the then-current stream wait on
// that completion event. No host thread ever blocks in the steady state.
//
// Failure handling. A watchdog thread (watchdog.h) polls the completion events of outstanding
// collectives; one that stays incomplete for longer than the timeout (a peer died, a rank
// took a different branch) gets the communicator aborted with ncclCommAbort, which ends the
// stuck kernels (and releases the library communicator's resources: an aborted communicator is
// not destroyed a second time), and every later call on it raises.
#include "communicator.h"
#include "store.h"
#include "watchdog.h"

#include <c10/hip/HIPStream.h>
#include <c10/util/Exception.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace pda {
namespace comm {

namespace {

#define PDA_HIP(expr)                                                                                    \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    TORCH_CHECK(e_ == hipSuccess, "HIP error in communicator: ", hipGetErrorString(e_), " at ", #expr);  \
  } while (0)

#define PDA_NCCL(expr)                                                                                    \
  do {                                                                                                    \
    ncclResult_t r_ = (expr);                                                                             \
    TORCH_CHECK(r_ == ncclSuccess, "RCCL error: ", ncclGetErrorString(r_), " at ", #expr);                \
  } while (0)

ncclDataType_t nccl_dtype(at::ScalarType t) {
  switch (t) {
    case at::kFloat: return ncclFloat32;
    case at::kBFloat16: return ncclBfloat16;
    case at::kHalf: return ncclFloat16;
    case at::kDouble: return ncclFloat64;
    case at::kLong: return ncclInt64;
    case at::kInt: return ncclInt32;
    case at::kByte: return ncclUint8;
    case at::kChar: return ncclInt8;
    default: TORCH_CHECK(false, "communicator: unsupported dtype ", t);
  }
}

ncclRedOp_t nccl_op(RedOp op) {
  switch (op) {
    case RedOp::kSum: return ncclSum;
    case RedOp::kAvg: return ncclAvg;
    case RedOp::kMax: return ncclMax;
  }
  return ncclSum;
}

double now_s() {
  using namespace std::chrono;
  return duration<double>(steady_clock::now().time_since_epoch()).count();
}

// Completion events are recycled: creating / destroying a hipEvent per collective costs more
// host time than the launch itself.
class EventPool {
 public:
  hipEvent_t get() {
    std::lock_guard<std::mutex> lock(mu_);
    if (!free_.empty()) {
      hipEvent_t e = free_.back();
      free_.pop_back();
      return e;
    }
    hipEvent_t e;
    PDA_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return e;
  }
  void put(hipEvent_t e) {
    std::lock_guard<std::mutex> lock(mu_);
    free_.push_back(e);
  }
  ~EventPool() {
    for (hipEvent_t e : free_) (void)hipEventDestroy(e);
  }

 private:
  std::mutex mu_;
  std::vector<hipEvent_t> free_;
};

struct PooledEvent {
  PooledEvent(std::shared_ptr<EventPool> p) : pool(std::move(p)), event(pool->get()) {}
  ~PooledEvent() { pool->put(event); }
  std::shared_ptr<EventPool> pool;
  hipEvent_t event;
};

class RcclCommunicator;

class RcclWork final : public Work {
 public:
  RcclWork(std::shared_ptr<PooledEvent> done, int device, std::vector<at::Tensor> keep,
           std::shared_ptr<std::atomic<bool>> aborted)
      : done_(std::move(done)), device_(device), keep_(std::move(keep)), aborted_(std::move(aborted)) {}
  // A handle dropped without wait() (a logged scalar nobody read) would hand its buffers back to the
  // caching allocator's pool of the COMPUTE stream while the queue's stream may still be writing
  // them: hold them until the collective has completed. Rare, so blocking the host is fine; after
  // an abort or at interpreter exit the event call just fails and nothing is waited for.
  ~RcclWork() override {
    if (!waited_ && !aborted_->load()) (void)hipEventSynchronize(done_->event);
  }
  void wait() override {
    hipStream_t cur = c10::hip::getCurrentHIPStream((c10::DeviceIndex)device_).stream();
    PDA_HIP(hipStreamWaitEvent(cur, done_->event, 0));
    waited_ = true;
    check();
  }
  void synchronize() override {
    // Polls rather than blocking in hipEventSynchronize: an aborted communicator's stuck kernel may
    // never signal the event, and the caller must get the error instead of a hang.
    while (hipEventQuery(done_->event) != hipSuccess) {
      check();
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    waited_ = true;
    check();
  }
  bool completed() override { return hipEventQuery(done_->event) == hipSuccess; }

 private:
  void check() const {
    TORCH_CHECK(!aborted_->load(), "communicator was aborted (a collective timed out or a peer failed)");
  }
  std::shared_ptr<PooledEvent> done_;
  int device_;
  std::vector<at::Tensor> keep_;  // buffers stay allocated at least as long as the handle
  std::shared_ptr<std::atomic<bool>> aborted_;
  bool waited_ = false;
};

class RcclCommunicator final : public Communicator {
 public:
  RcclCommunicator(std::shared_ptr<Rendezvous> rv, int device, double timeout_s, bool dual)
      : rv_(std::move(rv)), device_(device), timeout_s_(timeout_s), pool_(std::make_shared<EventPool>()),
        aborted_(std::make_shared<std::atomic<bool>>(false)) {
    rank_ = rv_->rank;
    world_ = rv_->world;
    PDA_HIP(hipSetDevice(device_));
    // The streams are created BEFORE the library initialises, at normal priority. Measured on this
    // stack (ROCm 7.0 runtime, RCCL 2.26): the first high-priority stream created after
    // ncclCommInitRank stops draining as soon as it exchanges more than a couple of event fences with
    // the default stream (tools/micro/comm_chain.py reproduces it; streams created before the
    // initialisation, or at normal priority, do not show it). PDA_COMM_STREAM_PRIORITY=high opts into
    // high-priority streams (still created first).
    {
      const char* pr = std::getenv("PDA_COMM_STREAM_PRIORITY");
      const bool high = pr != nullptr && pr[0] == 'h';
      int least = 0, greatest = 0;
      PDA_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
      for (int q = 0; q < 2; ++q) {
        if (high) PDA_HIP(hipStreamCreateWithPriority(&streams_[q], hipStreamNonBlocking, greatest));
        else PDA_HIP(hipStreamCreateWithFlags(&streams_[q], hipStreamNonBlocking));
      }
    }
    comms_[0].store(init_library_communicator());
    if (dual) {
      comms_[1].store(init_library_communicator());
      dual_route_.store(true);
    }
    int version = 0;
    (void)ncclGetVersion(&version);
    version_ = version;
    const char* wd = std::getenv("PDA_COMM_WATCHDOG");
    if (!(wd != nullptr && wd[0] == '0'))
      watchdog_ = std::make_unique<Watchdog>(
          timeout_s_,
          [this](const std::string& what, double age) {
            fprintf(stderr,
                    "[pda.comm] rank %d: %s has not completed after %.0f s; aborting the communicator "
                    "(a peer died or the ranks issued different collectives)\n",
                    rank_, what.c_str(), age);
            abort();
          },
          [this] { (void)hipSetDevice(device_); });
  }

  ~RcclCommunicator() override { close(); }

  void close() override {
    if (closed_.exchange(true)) return;
    if (watchdog_) watchdog_->stop();
    std::lock_guard<std::mutex> lock(mu_);
    const bool was_aborted = aborted_->load();
    if (!was_aborted) {
      (void)hipSetDevice(device_);
      for (auto s : streams_)
        if (s) (void)hipStreamSynchronize(s);
    }
    for (auto& c : comms_) {
      ncclComm_t lib = c.exchange(nullptr);
      if (lib != nullptr && !was_aborted) (void)ncclCommDestroy(lib);  // abort() already released it
    }
    for (auto& s : streams_) {
      if (s) (void)hipStreamDestroy(s);
      s = nullptr;
    }
  }

  int rank() const override { return rank_; }
  int world() const override { return world_; }
  std::string backend() const override { return "rccl"; }
  bool on_gpu() const override { return true; }
  bool aborted() const override { return aborted_->load(); }
  int version() const { return version_; }
  int64_t stream_handle(int q) const { return (int64_t)(uintptr_t)streams_[q]; }
  bool has_dual() const { return comms_[1].load() != nullptr; }
  bool dual_route() const { return dual_route_.load(); }
  void set_dual_route(bool on) {
    TORCH_CHECK(!on || has_dual(), "communicator was created without a second library communicator");
    dual_route_.store(on);
  }

  std::shared_ptr<Work> all_reduce(const at::Tensor& t, RedOp op, Queue q) override {
    check_collective_args("all_reduce", t, t, 1, true);
    count(Kind::kAllReduce, t);
    return issue(q, {t}, "all_reduce", [&](ncclComm_t comm, hipStream_t s) {
      PDA_NCCL(ncclAllReduce(t.data_ptr(), t.data_ptr(), (size_t)t.numel(), nccl_dtype(t.scalar_type()), nccl_op(op),
                             comm, s));
    });
  }

  std::shared_ptr<Work> all_gather(const at::Tensor& out, const at::Tensor& in, Queue q) override {
    check_collective_args("all_gather", out, in, world_, true);
    count(Kind::kAllGather, out);
    return issue(q, {out, in}, "all_gather", [&](ncclComm_t comm, hipStream_t s) {
      PDA_NCCL(ncclAllGather(in.data_ptr(), out.data_ptr(), (size_t)in.numel(), nccl_dtype(in.scalar_type()), comm, s));
    });
  }

  std::shared_ptr<Work> reduce_scatter(const at::Tensor& out, const at::Tensor& in, RedOp op, Queue q) override {
    check_collective_args("reduce_scatter", in, out, world_, true);
    count(Kind::kReduceScatter, in);
    return issue(q, {out, in}, "reduce_scatter", [&](ncclComm_t comm, hipStream_t s) {
      PDA_NCCL(ncclReduceScatter(in.data_ptr(), out.data_ptr(), (size_t)out.numel(), nccl_dtype(in.scalar_type()),
                                 nccl_op(op), comm, s));
    });
  }

  std::shared_ptr<Work> broadcast(const at::Tensor& t, int root, Queue q) override {
    check_collective_args("broadcast", t, t, 1, true);
    TORCH_CHECK(root >= 0 && root < world_, "broadcast: bad root ", root);
    count(Kind::kBroadcast, t);
    return issue(q, {t}, "broadcast", [&](ncclComm_t comm, hipStream_t s) {
      PDA_NCCL(ncclBroadcast(t.data_ptr(), t.data_ptr(), (size_t)t.numel(), nccl_dtype(t.scalar_type()), root, comm, s));
    });
  }

  void barrier() override { rv_->barrier("rccl"); }

  std::vector<std::string> exchange(const std::string& mine) override { return rv_->exchange("rccl", mine); }

  // Callable from any thread, also while another thread is inside an RCCL call on this
  // communicator (that is what ncclCommAbort is for), hence no lock: the pointers are atomics and
  // the first caller wins.
  void abort() override {
    bool expected = false;
    if (!aborted_->compare_exchange_strong(expected, true)) return;
    for (auto& c : comms_) {
      ncclComm_t lib = c.load();
      if (lib != nullptr) (void)ncclCommAbort(lib);
    }
  }

 private:
  ncclComm_t init_library_communicator() {
    ncclUniqueId id;
    std::memset(&id, 0, sizeof(id));
    static std::atomic<int> generation{0};  // several communicators per process get distinct ids
    const std::string key = "rccl/unique_id/" + std::to_string(generation++);
    if (rank_ == 0) {
      PDA_NCCL(ncclGetUniqueId(&id));
      if (rv_->client) rv_->client->set(key, std::string(reinterpret_cast<const char*>(&id), sizeof(id)));
    } else {
      const std::string bytes = rv_->client->get(key);
      TORCH_CHECK(bytes.size() == sizeof(id), "communicator: malformed RCCL unique id in the store");
      std::memcpy(&id, bytes.data(), sizeof(id));
    }
    ncclComm_t lib = nullptr;
    PDA_NCCL(ncclCommInitRank(&lib, world_, id, rank_));
    return lib;
  }

  template <typename F>
  std::shared_ptr<Work> issue(Queue q, std::vector<at::Tensor> keep, const char* what, F&& launch) {
    TORCH_CHECK(!aborted_->load(), "communicator was aborted (a collective timed out or a peer failed)");
    std::lock_guard<std::mutex> lock(mu_);  // RCCL calls of one process are serialised
    ncclComm_t lib = comms_[q == Queue::kGather && dual_route_.load() ? 1 : 0].load();
    TORCH_CHECK(lib != nullptr, "communicator is closed");
    int prev = 0;
    PDA_HIP(hipGetDevice(&prev));
    if (prev != device_) PDA_HIP(hipSetDevice(device_));
    hipStream_t s = streams_[(int)q];
    hipStream_t cur = c10::hip::getCurrentHIPStream((c10::DeviceIndex)device_).stream();
    {
      PooledEvent ready(pool_);
      PDA_HIP(hipEventRecord(ready.event, cur));
      PDA_HIP(hipStreamWaitEvent(s, ready.event, 0));
    }
    launch(lib, s);
    auto done = std::make_shared<PooledEvent>(pool_);
    PDA_HIP(hipEventRecord(done->event, s));
    if (prev != device_) PDA_HIP(hipSetDevice(prev));
    if (watchdog_) watchdog_->watch([done] { return hipEventQuery(done->event) == hipSuccess; }, what);
    return std::make_shared<RcclWork>(std::move(done), device_, std::move(keep), aborted_);
  }

  std::shared_ptr<Rendezvous> rv_;
  int device_, rank_ = 0, world_ = 1, version_ = 0;
  double timeout_s_;
  std::atomic<ncclComm_t> comms_[2] = {nullptr, nullptr};  // [1]: the gather queue's own, when dual
  std::atomic<bool> dual_route_{false};
  hipStream_t streams_[2] = {nullptr, nullptr};
  std::shared_ptr<EventPool> pool_;
  std::shared_ptr<std::atomic<bool>> aborted_;
  std::mutex mu_;
  std::atomic<bool> closed_{false};
  std::unique_ptr<Watchdog> watchdog_;
};

}  // namespace

std::shared_ptr<Communicator> make_rccl_communicator(std::shared_ptr<Rendezvous> rv, int device, double timeout_s,
                                                     bool dual) {
  return std::make_shared<RcclCommunicator>(std::move(rv), device, timeout_s, dual);
}

// dual_route: -1 = query only, 0 / 1 = route gathers to the shared / the second library communicator.
// Returns {has second communicator, gathers currently use it}.
std::pair<bool, bool> rccl_dual_route(const std::shared_ptr<Communicator>& c, int set) {
  auto* r = dynamic_cast<RcclCommunicator*>(c.get());
  if (r == nullptr) return {false, false};
  if (set >= 0) r->set_dual_route(set != 0);
  return {r->has_dual(), r->dual_route()};
}

int64_t rccl_stream_handle(const std::shared_ptr<Communicator>& c, int q) {
  auto* r = dynamic_cast<RcclCommunicator*>(c.get());
  TORCH_CHECK(r != nullptr, "not an RCCL communicator");
  return r->stream_handle(q);
}

int rccl_version(const std::shared_ptr<Communicator>& c) {
  auto* r = dynamic_cast<RcclCommunicator*>(c.get());
  return r ? r->version() : 0;
}

}  // namespace comm
}  // namespace pda
