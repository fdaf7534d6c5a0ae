// RCCL communicator: the GPU implementation of pda::comm::Communicator (see communicator.h).
//
// Streams. The communicator owns two HIP streams, one per Queue. Together with the caller's
// compute stream that is three of the four hardware queues a process gets, so collectives never
// share a hardware queue with compute kernels and an all-gather for the next unit is never
// stuck behind a reduce-scatter of the previous one.
//
// Fences. issue(): an event recorded on the caller's stream is awaited by the queue's stream
// (the collective sees every kernel enqueued before it), the RCCL call is enqueued, and a
// completion event is recorded behind it. Work::wait() makes the then-current stream wait on
// that completion event. No host thread ever blocks in the steady state.
//
// Failure handling. A watchdog thread polls the completion events of outstanding
// collectives; one that stays incomplete for longer than the timeout (a peer died, a rank
// took a different branch) gets the communicator aborted with ncclCommAbort, which ends the
// stuck kernels, and every later call on it raises.
#include "communicator.h"
#include "store.h"

#include <c10/hip/HIPStream.h>
#include <c10/util/Exception.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
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
  void wait() override {
    hipStream_t cur = c10::hip::getCurrentHIPStream((c10::DeviceIndex)device_).stream();
    PDA_HIP(hipStreamWaitEvent(cur, done_->event, 0));
    check();
  }
  void synchronize() override {
    PDA_HIP(hipEventSynchronize(done_->event));
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
};

class RcclCommunicator final : public Communicator {
 public:
  RcclCommunicator(std::shared_ptr<Rendezvous> rv, int device, double timeout_s)
      : rv_(std::move(rv)), device_(device), timeout_s_(timeout_s), pool_(std::make_shared<EventPool>()),
        aborted_(std::make_shared<std::atomic<bool>>(false)) {
    rank_ = rv_->rank;
    world_ = rv_->world;
    PDA_HIP(hipSetDevice(device_));
    ncclUniqueId id;
    std::memset(&id, 0, sizeof(id));
    static int generation = 0;  // several communicators per process get distinct ids
    const std::string key = "rccl/unique_id/" + std::to_string(generation++);
    if (rank_ == 0) {
      PDA_NCCL(ncclGetUniqueId(&id));
      if (rv_->client) rv_->client->set(key, std::string(reinterpret_cast<const char*>(&id), sizeof(id)));
    } else {
      const std::string bytes = rv_->client->get(key);
      TORCH_CHECK(bytes.size() == sizeof(id), "communicator: malformed RCCL unique id in the store");
      std::memcpy(&id, bytes.data(), sizeof(id));
    }
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
    PDA_NCCL(ncclCommInitRank(&comm_, world_, id, rank_));
    int version = 0;
    (void)ncclGetVersion(&version);
    version_ = version;
    const char* wd = std::getenv("PDA_COMM_WATCHDOG");
    use_watchdog_ = !(wd != nullptr && wd[0] == '0');
    if (use_watchdog_) watchdog_ = std::thread([this] { watch(); });
  }

  ~RcclCommunicator() override { close(); }

  void close() override {
    {
      std::lock_guard<std::mutex> lock(watch_mu_);
      if (closed_) return;
      closed_ = true;
      stop_ = true;
    }
    watch_cv_.notify_all();
    if (watchdog_.joinable()) watchdog_.join();
    std::lock_guard<std::mutex> lock(mu_);
    if (comm_ != nullptr) {
      if (!aborted_->load()) {  // abort() already tore the library communicator down
        (void)hipSetDevice(device_);
        for (auto s : streams_)
          if (s) (void)hipStreamSynchronize(s);
        (void)ncclCommDestroy(comm_);
      }
      comm_ = nullptr;
    }
    for (auto& s : streams_) {
      if (s) (void)hipStreamDestroy(s);
      s = nullptr;
    }
    watched_.clear();
  }

  int rank() const override { return rank_; }
  int world() const override { return world_; }
  std::string backend() const override { return "rccl"; }
  bool on_gpu() const override { return true; }
  bool aborted() const override { return aborted_->load(); }
  int version() const { return version_; }
  int64_t stream_handle(int q) const { return (int64_t)(uintptr_t)streams_[q]; }

  std::shared_ptr<Work> all_reduce(const at::Tensor& t, RedOp op, Queue q) override {
    check_collective_args("all_reduce", t, t, 1, true);
    count(Kind::kAllReduce, t);
    return issue(q, {t}, "all_reduce", [&](hipStream_t s) {
      PDA_NCCL(ncclAllReduce(t.data_ptr(), t.data_ptr(), (size_t)t.numel(), nccl_dtype(t.scalar_type()), nccl_op(op),
                             comm_, s));
    });
  }

  std::shared_ptr<Work> all_gather(const at::Tensor& out, const at::Tensor& in, Queue q) override {
    check_collective_args("all_gather", out, in, world_, true);
    count(Kind::kAllGather, out);
    return issue(q, {out, in}, "all_gather", [&](hipStream_t s) {
      PDA_NCCL(ncclAllGather(in.data_ptr(), out.data_ptr(), (size_t)in.numel(), nccl_dtype(in.scalar_type()), comm_, s));
    });
  }

  std::shared_ptr<Work> reduce_scatter(const at::Tensor& out, const at::Tensor& in, RedOp op, Queue q) override {
    check_collective_args("reduce_scatter", in, out, world_, true);
    count(Kind::kReduceScatter, in);
    return issue(q, {out, in}, "reduce_scatter", [&](hipStream_t s) {
      PDA_NCCL(ncclReduceScatter(in.data_ptr(), out.data_ptr(), (size_t)out.numel(), nccl_dtype(in.scalar_type()),
                                 nccl_op(op), comm_, s));
    });
  }

  std::shared_ptr<Work> broadcast(const at::Tensor& t, int root, Queue q) override {
    check_collective_args("broadcast", t, t, 1, true);
    TORCH_CHECK(root >= 0 && root < world_, "broadcast: bad root ", root);
    count(Kind::kBroadcast, t);
    return issue(q, {t}, "broadcast", [&](hipStream_t s) {
      PDA_NCCL(ncclBroadcast(t.data_ptr(), t.data_ptr(), (size_t)t.numel(), nccl_dtype(t.scalar_type()), root, comm_, s));
    });
  }

  void barrier() override { rv_->barrier("rccl"); }

  std::vector<std::string> exchange(const std::string& mine) override {
    if (world_ <= 1 || !rv_->client) return {mine};
    const std::string base = "xchg/rccl/" + std::to_string(exchange_seq_++) + "/";
    rv_->client->set(base + std::to_string(rank_), mine);
    std::vector<std::string> out;
    for (int r = 0; r < world_; ++r) out.push_back(r == rank_ ? mine : rv_->client->get(base + std::to_string(r)));
    return out;
  }

  void abort() override {
    bool expected = false;
    if (!aborted_->compare_exchange_strong(expected, true)) return;
    if (comm_ != nullptr) (void)ncclCommAbort(comm_);
  }

 private:
  template <typename F>
  std::shared_ptr<Work> issue(Queue q, std::vector<at::Tensor> keep, const char* what, F&& launch) {
    TORCH_CHECK(!aborted_->load(), "communicator was aborted (a collective timed out or a peer failed)");
    std::lock_guard<std::mutex> lock(mu_);  // RCCL calls on one communicator are serialised
    TORCH_CHECK(comm_ != nullptr, "communicator is closed");
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
    launch(s);
    auto done = std::make_shared<PooledEvent>(pool_);
    PDA_HIP(hipEventRecord(done->event, s));
    if (prev != device_) PDA_HIP(hipSetDevice(prev));
    if (use_watchdog_) {
      std::lock_guard<std::mutex> wl(watch_mu_);
      watched_.push_back({done, now_s(), what});
    }
    return std::make_shared<RcclWork>(std::move(done), device_, std::move(keep), aborted_);
  }

  struct Watched {
    std::shared_ptr<PooledEvent> done;
    double issued;
    const char* what;
  };

  void watch() {
    (void)hipSetDevice(device_);
    std::unique_lock<std::mutex> lock(watch_mu_);
    while (!stop_) {
      watch_cv_.wait_for(lock, std::chrono::milliseconds(watched_.empty() ? 1000 : 250));
      if (stop_) break;
      const double now = now_s();
      while (!watched_.empty()) {
        Watched& w = watched_.front();
        if (hipEventQuery(w.done->event) == hipSuccess) {
          watched_.pop_front();
          continue;
        }
        if (timeout_s_ > 0 && now - w.issued > timeout_s_) {
          fprintf(stderr,
                  "[pda.comm] rank %d: %s has not completed after %.0f s; aborting the communicator "
                  "(a peer died or the ranks issued different collectives)\n",
                  rank_, w.what, now - w.issued);
          lock.unlock();
          abort();
          lock.lock();
          watched_.clear();
        }
        break;  // collectives of one queue complete in order; younger ones are re-checked next round
      }
    }
  }

  std::shared_ptr<Rendezvous> rv_;
  int device_, rank_ = 0, world_ = 1, version_ = 0;
  double timeout_s_;
  ncclComm_t comm_ = nullptr;
  hipStream_t streams_[2] = {nullptr, nullptr};
  std::shared_ptr<EventPool> pool_;
  std::shared_ptr<std::atomic<bool>> aborted_;
  std::mutex mu_;
  int64_t exchange_seq_ = 0;

  std::thread watchdog_;
  std::mutex watch_mu_;
  std::condition_variable watch_cv_;
  std::deque<Watched> watched_;
  bool stop_ = false;
  bool closed_ = false;
  bool use_watchdog_ = true;
};

}  // namespace

std::shared_ptr<Communicator> make_rccl_communicator(std::shared_ptr<Rendezvous> rv, int device, double timeout_s) {
  return std::make_shared<RcclCommunicator>(std::move(rv), device, timeout_s);
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
