// In-process GPU activity collection: one record per kernel dispatch / memory copy with the
// device timestamps of its begin and end, collected through roctracer's activity-buffer API
// while a window is open. The range tracer opens and closes the window with the profiler's
// wait / warm-up / active schedule and writes the records into the same Chrome trace as its
// ranges, so a plain `python train_ddp.py` produces the kernel-level per-rank trace the
// reference gets from torch.profiler (assignments/assignment1/train_baseline.py:79-87,
// train_ddp.py:128-139) without an external profiler process.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace pda {
namespace ktrace {

struct Activity {
  enum Kind : int { kKernel = 0, kCopy = 1, kBarrier = 2, kApi = 3 };
  int kind = kKernel;
  std::string name;       // demangled kernel name, or the copy's description
  uint64_t begin_ns = 0;  // roctracer clock
  uint64_t end_ns = 0;
  int device = 0;
  uint64_t queue = 0;  // HIP stream / hardware queue id as the runtime reports it (kApi: the host thread id)
  uint64_t correlation = 0;
  uint64_t bytes = 0;  // copies only
};

// True when libroctracer64 could be loaded and its activity API resolved. Never throws.
bool available();
// Why available() is false ("" when it is true).
std::string unavailable_reason();

// Opens a collection window: every kernel dispatch and copy that COMPLETES on the GPU from now
// until stop() is recorded. Returns false (and records nothing) when the API is unavailable or
// a window is already open. `keep_api_calls`: also keep one record per HIP API call (hipLaunchKernel,
// hipMemcpyAsync, ...) with the correlation id it shares with the device-side record it caused.
bool start(bool keep_api_calls = false);
// Closes the window, flushes the library's buffers and returns what was collected, in completion
// order. Safe to call without a matching start (returns an empty vector).
std::vector<Activity> stop();
bool collecting();

// Nanoseconds on the clock the records use (0 when unavailable): sampled next to the host clock
// to place the records on a host timeline.
uint64_t now_ns();

// Diagnosis: {buffers, records, records of other domains} the library's callback has delivered so far.
std::vector<uint64_t> debug_counts();

}  // namespace ktrace
}  // namespace pda
