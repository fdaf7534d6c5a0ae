// Two thin Communicator implementations that wrap something else:
//
//   C10dCommunicator       adapter around a torch.distributed process group a caller already
//                          owns (`DDP(model, process_group=pg)`), so code written against
//                          torch's groups keeps working; the framework's own bootstrap never
//                          creates one.
//   RecordingCommunicator  decorator: forwards to an inner communicator and keeps a log of
//                          (kind, elements, queue) per collective, in issue order. Tests assert
//                          exact collective sequences on it (prefetch order, "no_sync issues
//                          nothing"), and the trace tools use the log to label collectives.
//   HangingCommunicator    fault injection: forwards to an inner communicator until a chosen
//                          number of collectives has been issued; the next one never completes (a
//                          peer that died mid-collective). It owns a Watchdog like the RCCL
//                          communicator's, so the whole failure path — poll, time-out, abort, the
//                          blocked caller raising, every later collective raising — runs on a
//                          CPU without a GPU or a second process.
#include "communicator.h"
#include "watchdog.h"

#include <c10/util/Exception.h>
#include <torch/csrc/distributed/c10d/ProcessGroup.hpp>

#include <atomic>
#include <chrono>
#include <thread>

namespace pda {
namespace comm {

namespace {

class C10dWork final : public Work {
 public:
  explicit C10dWork(c10::intrusive_ptr<c10d::Work> w, std::vector<at::Tensor> keep)
      : work_(std::move(w)), keep_(std::move(keep)) {}
  void wait() override {
    if (work_) work_->wait();
  }
  void synchronize() override {
    if (!work_) return;
    work_->wait();
    work_->synchronize();
  }
  bool completed() override { return !work_ || work_->isCompleted(); }

 private:
  c10::intrusive_ptr<c10d::Work> work_;
  std::vector<at::Tensor> keep_;
};

class C10dCommunicator final : public Communicator {
 public:
  // `device_tensors`: the group is given GPU tensors although its backend is a host one (gloo stages
  // them through host memory): several ranks rehearsing on one GPU, which the GPU library refuses.
  explicit C10dCommunicator(c10::intrusive_ptr<c10d::ProcessGroup> pg, bool device_tensors = false)
      : pg_(std::move(pg)), device_tensors_(device_tensors) {
    TORCH_CHECK(pg_, "C10dCommunicator: null process group");
    name_ = pg_->getBackendName();
  }
  int rank() const override { return pg_->getRank(); }
  int world() const override { return pg_->getSize(); }
  std::string backend() const override { return "c10d:" + name_; }
  bool on_gpu() const override { return name_ == "nccl" || device_tensors_; }

  std::shared_ptr<Work> all_reduce(const at::Tensor& t, RedOp op, Queue) override {
    check_collective_args("all_reduce", t, t, 1, t.is_cuda());
    if (staged(t)) {  // rehearsal on a shared GPU: through host memory, complete on return (counted by the inner call)
      at::NoGradGuard ng;
      at::Tensor host = t.cpu();
      all_reduce(host, op, Queue::kReduce)->wait();
      at::Tensor(t).copy_(host);
      return std::make_shared<CompletedWork>();
    }
    count(Kind::kAllReduce, t);
    std::vector<at::Tensor> ts{t};
    c10d::AllreduceOptions opts;
    const bool scale = op == RedOp::kAvg && name_ != "nccl";  // gloo has no AVG
    opts.reduceOp = op == RedOp::kMax ? c10d::ReduceOp::MAX
                                      : (op == RedOp::kAvg && !scale ? c10d::ReduceOp::AVG : c10d::ReduceOp::SUM);
    auto w = pg_->allreduce(ts, opts);
    if (scale) {
      w->wait();
      at::NoGradGuard ng;
      at::Tensor(t).div_((double)world());
      return std::make_shared<CompletedWork>();
    }
    return std::make_shared<C10dWork>(std::move(w), std::vector<at::Tensor>{t});
  }

  std::shared_ptr<Work> all_gather(const at::Tensor& out, const at::Tensor& in, Queue) override {
    check_collective_args("all_gather", out, in, world(), out.is_cuda());
    count(Kind::kAllGather, out);
    if (staged(out)) {
      at::NoGradGuard ng;
      at::Tensor host_in = in.cpu(), host_out = at::empty(out.sizes(), host_in.options());
      pg_->_allgather_base(host_out, host_in)->wait();
      at::Tensor(out).copy_(host_out);
      return std::make_shared<CompletedWork>();
    }
    at::Tensor o = out, i = in;
    return std::make_shared<C10dWork>(pg_->_allgather_base(o, i), std::vector<at::Tensor>{out, in});
  }

  std::shared_ptr<Work> reduce_scatter(const at::Tensor& out, const at::Tensor& in, RedOp op, Queue) override {
    check_collective_args("reduce_scatter", in, out, world(), out.is_cuda());
    if (staged(in)) {
      at::NoGradGuard ng;
      at::Tensor host_in = in.cpu(), host_out = at::empty(out.sizes(), host_in.options());
      reduce_scatter(host_out, host_in, op, Queue::kReduce)->wait();
      at::Tensor(out).copy_(host_out);
      return std::make_shared<CompletedWork>();
    }
    count(Kind::kReduceScatter, in);
    at::Tensor o = out, i = in;
    c10d::ReduceScatterOptions opts;
    const bool scale = op == RedOp::kAvg && name_ != "nccl";
    opts.reduceOp = op == RedOp::kMax ? c10d::ReduceOp::MAX
                                      : (op == RedOp::kAvg && !scale ? c10d::ReduceOp::AVG : c10d::ReduceOp::SUM);
    auto w = pg_->_reduce_scatter_base(o, i, opts);
    if (scale) {
      w->wait();
      at::NoGradGuard ng;
      o.div_((double)world());
      return std::make_shared<CompletedWork>();
    }
    return std::make_shared<C10dWork>(std::move(w), std::vector<at::Tensor>{out, in});
  }

  std::shared_ptr<Work> broadcast(const at::Tensor& t, int root, Queue) override {
    if (staged(t)) {
      at::NoGradGuard ng;
      at::Tensor host = t.cpu();
      broadcast(host, root, Queue::kGather)->wait();
      at::Tensor(t).copy_(host);
      return std::make_shared<CompletedWork>();
    }
    count(Kind::kBroadcast, t);
    std::vector<at::Tensor> ts{t};
    c10d::BroadcastOptions opts;
    opts.rootRank = root;
    return std::make_shared<C10dWork>(pg_->broadcast(ts, opts), std::vector<at::Tensor>{t});
  }

  void barrier() override { pg_->barrier()->wait(); }
  void abort() override { pg_->abort(); }

  std::vector<std::string> exchange(const std::string& mine) override {
    if (world() <= 1) return {mine};
    // Two rounds over the group: lengths, then padded payloads.
    const auto dev = name_ == "nccl" ? at::Device(at::kCUDA) : at::Device(at::kCPU);
    auto gather = [&](const at::Tensor& local) {
      std::vector<at::Tensor> outs;
      for (int r = 0; r < world(); ++r) outs.push_back(at::empty_like(local));
      std::vector<std::vector<at::Tensor>> lists{outs};
      std::vector<at::Tensor> ins{local};
      pg_->allgather(lists, ins)->wait();
      return lists[0];
    };
    at::Tensor len = at::tensor({(int64_t)mine.size()}, at::TensorOptions().dtype(at::kLong)).to(dev);
    auto lens = gather(len);
    int64_t longest = 0;
    for (auto& l : lens) longest = std::max(longest, l.cpu().item<int64_t>());
    at::Tensor payload = at::zeros({longest}, at::TensorOptions().dtype(at::kByte));
    std::memcpy(payload.data_ptr(), mine.data(), mine.size());
    auto all = gather(payload.to(dev));
    std::vector<std::string> out;
    for (int r = 0; r < world(); ++r) {
      at::Tensor c = all[r].cpu();
      out.emplace_back(static_cast<const char*>(c.data_ptr()), (size_t)lens[r].cpu().item<int64_t>());
    }
    return out;
  }

 private:
  // GPU tensor on a host backend: staged through host memory here, synchronously, instead of through
  // the backend's own device path (with several processes on one GPU its copy streams and pinned
  // buffers were seen to leave two ranks waiting for each other).
  bool staged(const at::Tensor& t) const { return device_tensors_ && t.is_cuda() && name_ != "nccl"; }

  c10::intrusive_ptr<c10d::ProcessGroup> pg_;
  std::string name_;
  bool device_tensors_ = false;
};

}  // namespace

struct RecordedCall {
  int kind;
  int64_t numel;
  int queue;
};

class RecordingCommunicator final : public Communicator {
 public:
  explicit RecordingCommunicator(std::shared_ptr<Communicator> inner) : inner_(std::move(inner)) {
    TORCH_CHECK(inner_, "RecordingCommunicator: null inner communicator");
  }
  int rank() const override { return inner_->rank(); }
  int world() const override { return inner_->world(); }
  std::string backend() const override { return "recording:" + inner_->backend(); }
  bool on_gpu() const override { return inner_->on_gpu(); }
  bool aborted() const override { return inner_->aborted(); }

  std::shared_ptr<Work> all_reduce(const at::Tensor& t, RedOp op, Queue q) override {
    note(Kind::kAllReduce, t.numel(), q, t);
    return inner_->all_reduce(t, op, q);
  }
  std::shared_ptr<Work> all_gather(const at::Tensor& out, const at::Tensor& in, Queue q) override {
    note(Kind::kAllGather, out.numel(), q, out);
    return inner_->all_gather(out, in, q);
  }
  std::shared_ptr<Work> reduce_scatter(const at::Tensor& out, const at::Tensor& in, RedOp op, Queue q) override {
    note(Kind::kReduceScatter, in.numel(), q, in);
    return inner_->reduce_scatter(out, in, op, q);
  }
  std::shared_ptr<Work> broadcast(const at::Tensor& t, int root, Queue q) override {
    note(Kind::kBroadcast, t.numel(), q, t);
    return inner_->broadcast(t, root, q);
  }
  void barrier() override { inner_->barrier(); }
  void close() override { inner_->close(); }
  void abort() override { inner_->abort(); }
  std::vector<std::string> exchange(const std::string& mine) override { return inner_->exchange(mine); }

  std::vector<std::tuple<std::string, int64_t, int>> log() {
    std::lock_guard<std::mutex> lock(log_mu_);
    static const char* names[] = {"?", "all_reduce", "all_gather", "reduce_scatter", "broadcast"};
    std::vector<std::tuple<std::string, int64_t, int>> out;
    for (auto& c : log_) out.emplace_back(names[c.kind], c.numel, c.queue);
    return out;
  }
  void clear() {
    std::lock_guard<std::mutex> lock(log_mu_);
    log_.clear();
  }

 private:
  void note(Kind k, int64_t numel, Queue q, const at::Tensor& payload) {
    count(k, payload);
    std::lock_guard<std::mutex> lock(log_mu_);
    log_.push_back({(int)k, numel, (int)q});
  }
  std::shared_ptr<Communicator> inner_;
  std::mutex log_mu_;
  std::vector<RecordedCall> log_;
};

namespace {

class HungWork final : public Work {
 public:
  explicit HungWork(std::shared_ptr<std::atomic<bool>> aborted) : aborted_(std::move(aborted)) {}
  void wait() override { synchronize(); }  // a CPU collective's wait() blocks the host
  void synchronize() override {
    while (!aborted_->load()) std::this_thread::sleep_for(std::chrono::milliseconds(5));
    TORCH_CHECK(false, "communicator was aborted (a collective timed out or a peer failed)");
  }
  bool completed() override { return false; }

 private:
  std::shared_ptr<std::atomic<bool>> aborted_;
};

class HangingCommunicator final : public Communicator {
 public:
  HangingCommunicator(std::shared_ptr<Communicator> inner, int64_t healthy_calls, double timeout_s)
      : inner_(std::move(inner)), healthy_left_(healthy_calls), aborted_(std::make_shared<std::atomic<bool>>(false)) {
    TORCH_CHECK(inner_, "HangingCommunicator: null inner communicator");
    watchdog_ = std::make_unique<Watchdog>(
        timeout_s,
        [this](const std::string& what, double age) {
          fprintf(stderr, "[pda.comm] rank %d: %s has not completed after %.1f s; aborting the communicator\n",
                  inner_->rank(), what.c_str(), age);
          abort();
        },
        nullptr, /*poll_ms=*/20);
  }
  ~HangingCommunicator() override { watchdog_->stop(); }
  int rank() const override { return inner_->rank(); }
  int world() const override { return inner_->world(); }
  std::string backend() const override { return "hanging:" + inner_->backend(); }
  bool on_gpu() const override { return inner_->on_gpu(); }
  bool aborted() const override { return aborted_->load(); }

  std::shared_ptr<Work> all_reduce(const at::Tensor& t, RedOp op, Queue q) override {
    if (auto w = maybe_hang("all_reduce")) return w;
    return inner_->all_reduce(t, op, q);
  }
  std::shared_ptr<Work> all_gather(const at::Tensor& out, const at::Tensor& in, Queue q) override {
    if (auto w = maybe_hang("all_gather")) return w;
    return inner_->all_gather(out, in, q);
  }
  std::shared_ptr<Work> reduce_scatter(const at::Tensor& out, const at::Tensor& in, RedOp op, Queue q) override {
    if (auto w = maybe_hang("reduce_scatter")) return w;
    return inner_->reduce_scatter(out, in, op, q);
  }
  std::shared_ptr<Work> broadcast(const at::Tensor& t, int root, Queue q) override {
    if (auto w = maybe_hang("broadcast")) return w;
    return inner_->broadcast(t, root, q);
  }
  void barrier() override {
    check();
    inner_->barrier();
  }
  void close() override {
    watchdog_->stop();
    inner_->close();
  }
  void abort() override {
    bool expected = false;
    if (!aborted_->compare_exchange_strong(expected, true)) return;
    inner_->abort();
  }
  std::vector<std::string> exchange(const std::string& mine) override { return inner_->exchange(mine); }

 private:
  void check() const {
    TORCH_CHECK(!aborted_->load(), "communicator was aborted (a collective timed out or a peer failed)");
  }
  std::shared_ptr<Work> maybe_hang(const char* what) {
    check();
    if (healthy_left_.fetch_sub(1) > 0) return nullptr;
    auto w = std::make_shared<HungWork>(aborted_);
    watchdog_->watch([] { return false; }, what);
    return w;
  }
  std::shared_ptr<Communicator> inner_;
  std::atomic<int64_t> healthy_left_;
  std::shared_ptr<std::atomic<bool>> aborted_;
  std::unique_ptr<Watchdog> watchdog_;
};

}  // namespace

std::shared_ptr<Communicator> make_hanging_communicator(std::shared_ptr<Communicator> inner, int64_t healthy_calls,
                                                        double timeout_s) {
  return std::make_shared<HangingCommunicator>(std::move(inner), healthy_calls, timeout_s);
}

std::shared_ptr<Communicator> make_c10d_communicator(c10::intrusive_ptr<c10d::ProcessGroup> pg, bool device_tensors) {
  return std::make_shared<C10dCommunicator>(std::move(pg), device_tensors);
}

std::shared_ptr<Communicator> make_recording_communicator(std::shared_ptr<Communicator> inner) {
  return std::make_shared<RecordingCommunicator>(std::move(inner));
}

std::vector<std::tuple<std::string, int64_t, int>> recording_log(const std::shared_ptr<Communicator>& c, bool clear) {
  auto* r = dynamic_cast<RecordingCommunicator*>(c.get());
  TORCH_CHECK(r != nullptr, "not a recording communicator");
  auto out = r->log();
  if (clear) r->clear();
  return out;
}

}  // namespace comm
}  // namespace pda
