// Collective watchdog: a thread that polls the completion of outstanding collectives and calls
// a handler when the oldest one has been outstanding for longer than the timeout (a peer died,
// the ranks issued different collectives). The handler of a communicator aborts it, which ends
// the stuck work and makes every later collective raise instead of hanging the job.
//
// The poll loop knows nothing about GPUs: a watched item is a `done()` probe plus a label, so
// the RCCL communicator feeds it hipEventQuery probes and a test feeds it a probe that never
// turns true (tests/test_comm_cpu.py drives abort -> raise through HangingCommunicator).
//
// Items complete in the order they were registered (collectives of one queue do), so only the
// front of the list is probed; younger items are looked at once it has completed.
//
// Reference counterpart: the watchdog thread of ProcessGroupNCCL behind
// dist.init_process_group('nccl') (assignments/assignment1/train_ddp.py:23-24).
#pragma once

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>

namespace pda {
namespace comm {

class Watchdog {
 public:
  using Probe = std::function<bool()>;                                  // true once the item has completed
  using Handler = std::function<void(const std::string& what, double age_s)>;  // called at most once per expiry
  using ThreadInit = std::function<void()>;                             // runs first on the watchdog thread

  Watchdog(double timeout_s, Handler on_timeout, ThreadInit init = nullptr, int poll_ms = 250)
      : timeout_s_(timeout_s), on_timeout_(std::move(on_timeout)), poll_ms_(poll_ms) {
    thread_ = std::thread([this, init] {
      if (init) init();
      loop();
    });
  }

  ~Watchdog() { stop(); }

  void stop() {
    {
      std::lock_guard<std::mutex> lock(mu_);
      if (stop_) return;
      stop_ = true;
    }
    cv_.notify_all();
    if (thread_.joinable() && thread_.get_id() != std::this_thread::get_id()) thread_.join();
  }

  void watch(Probe done, std::string what) {
    std::lock_guard<std::mutex> lock(mu_);
    if (stop_) return;
    items_.push_back({std::move(done), now_s(), std::move(what)});
  }

  size_t outstanding() {
    std::lock_guard<std::mutex> lock(mu_);
    return items_.size();
  }

  int64_t expirations() {
    std::lock_guard<std::mutex> lock(mu_);
    return expirations_;
  }

 private:
  struct Item {
    Probe done;
    double issued;
    std::string what;
  };

  static double now_s() {
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
  }

  void loop() {
    std::unique_lock<std::mutex> lock(mu_);
    while (!stop_) {
      cv_.wait_for(lock, std::chrono::milliseconds(items_.empty() ? 4 * poll_ms_ : poll_ms_));
      if (stop_) break;
      const double now = now_s();
      while (!items_.empty()) {
        Item& it = items_.front();
        if (it.done()) {
          items_.pop_front();
          continue;
        }
        if (timeout_s_ > 0 && now - it.issued > timeout_s_) {
          const std::string what = it.what;
          const double age = now - it.issued;
          items_.clear();  // everything behind a stuck collective is stuck with it
          ++expirations_;
          lock.unlock();
          on_timeout_(what, age);
          lock.lock();
        }
        break;
      }
    }
  }

  const double timeout_s_;
  Handler on_timeout_;
  const int poll_ms_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<Item> items_;
  std::thread thread_;
  bool stop_ = false;
  int64_t expirations_ = 0;
};

}  // namespace comm
}  // namespace pda
