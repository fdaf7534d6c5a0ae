// Range tracer: named host + GPU time ranges -> Chrome-trace JSON, mirrored to roctx.
//
// Every push/pop pair records the host wall-clock interval and, when a GPU is in use, a pair
// of HIP events on the calling thread's current stream, so the exported trace shows both
// where the host spent its time launching a phase (forward, backward, optimizer, all-gather
// wait, ...) and when the GPU actually executed it. The same ranges are forwarded to
// roctxRangePush / roctxRangePop when libroctx64 is loadable, so a rocprofv3 marker trace
// of the same run carries identical annotations next to the kernel rows.
//
// Ranges come from two places: Python (`StepProfiler.range`, the trainers' forward / backward /
// optimizer phases) and the C++ runtime itself (all-gather wait, reduce-scatter launch,
// end-of-backward fence in reducer.cpp / fsdp_unit.cpp) through pda::trace::Scope, which
// records into whichever tracer is currently ACTIVE (RangeTracer.activate()) and costs one
// atomic load when none is. Ranges may be opened on any thread (the autograd engine runs the
// runtime's hooks on its own thread); every thread gets its own row.
//
// This is the framework's own tracer (the reference wraps training in torch.profiler:
// assignments/assignment1/train_baseline.py:79-87). Kernel rows: while the tracer is enabled
// (the profiler schedule's active window) kernel_tracer.cpp collects one record per kernel
// dispatch and memory copy in this process; they are written into the same file on rows of
// their own — one per HIP queue, plus ONE row for collectives whatever queue they ran on — on
// the host clock of the ranges. Under an external rocprofv3 the in-process collection stays
// off; the roctx mirror then puts the ranges on rocprofv3's clock and utils/rocprof_trace.py
// merges its kernel trace with them.
#include "range_tracer.h"

#include "kernel_tracer.h"

#include <c10/hip/HIPStream.h>
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <torch/extension.h>

#include <atomic>
#include <cctype>
#include <chrono>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

namespace py = pybind11;

namespace {

using Clock = std::chrono::steady_clock;

double now_us() {
  return std::chrono::duration<double, std::micro>(Clock::now().time_since_epoch()).count();
}

struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    // rocprofv3 listens to the rocprofiler-sdk build of ROCTx; the roctracer build serves older tools.
    void* lib = nullptr;
    for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1",
                             "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "libroctx64.so", "libroctx64.so.4"}) {
      lib = dlopen(name, RTLD_LAZY | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) return;
    push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
    pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
  }
};

Roctx& roctx() {
  static Roctx r;
  return r;
}

struct Range {
  std::string name, cat;
  double host_begin = 0, host_end = 0;
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;
  int depth = 0;
  int thread = 0;  // 0 = the thread that created the tracer, 1.. = others in order of appearance
};

std::string json_escape(const std::string& s) {
  std::string o;
  for (char c : s) {
    if (c == '"' || c == '\\') o += '\\';
    if (c == '\n') { o += "\\n"; continue; }
    o += c;
  }
  return o;
}

constexpr int kCollectiveRow = 9000;  // tid of the collectives row (utils/rocprof_trace.py uses the same)

bool is_collective_name(const std::string& name) {
  std::string l(name.size(), ' ');
  for (size_t i = 0; i < name.size(); ++i) l[i] = (char)std::tolower((unsigned char)name[i]);
  for (const char* key : {"nccl", "rccl", "allreduce", "all_reduce", "allgather", "all_gather", "reducescatter",
                          "reduce_scatter", "broadcast", "sendrecv", "onerankreduce"})
    if (l.find(key) != std::string::npos) return true;
  return false;
}

class RangeTracer {
 public:
  explicit RangeTracer(bool use_gpu, bool kernels = false)
      : use_gpu_(use_gpu), kernels_(kernels && use_gpu && pda::ktrace::available()) {
    if (use_gpu_) {
      TORCH_CHECK(hipEventCreate(&base_event_) == hipSuccess, "hipEventCreate failed");
      hipStream_t s = c10::hip::getCurrentHIPStream().stream();
      TORCH_CHECK(hipEventRecord(base_event_, s) == hipSuccess, "hipEventRecord failed");
      hipEventSynchronize(base_event_);
    }
    base_host_ = now_us();
  }

  ~RangeTracer() {
    deactivate();
    if (window_open_) pda::ktrace::stop();
    for (auto& r : done_) destroy(r);
    for (auto& kv : open_)
      for (auto& r : kv.second) destroy(r);
    if (base_event_) hipEventDestroy(base_event_);
  }

  // Makes this tracer the one the C++ runtime's pda::trace::Scope ranges go to.
  void activate();
  void deactivate();

  // Enabling opens the kernel-collection window (when this tracer collects kernels), disabling
  // closes it: the device is drained first, so every kernel launched inside the window has
  // completed and has its record.
  void set_enabled(bool e) {
    if (kernels_ && e && !enabled_ && !pda::ktrace::collecting()) {
      const uint64_t k0 = pda::ktrace::now_ns();
      const double h0 = now_us();
      if (pda::ktrace::start(/*keep_api_calls=*/true)) {
        window_k0_ns_ = k0;
        window_h0_us_ = h0;
        window_open_ = true;
      }
    } else if (!e && window_open_) {
      hipDeviceSynchronize();
      std::vector<pda::ktrace::Activity> got = pda::ktrace::stop();
      std::lock_guard<std::mutex> lk(mu_);
      for (auto& a : got) {
        Kernel k;
        k.ts_us = window_h0_us_ + (double(a.begin_ns) - double(window_k0_ns_)) * 1e-3;
        k.dur_us = double(a.end_ns - a.begin_ns) * 1e-3;
        k.copy = a.kind == pda::ktrace::Activity::kCopy;
        k.api = a.kind == pda::ktrace::Activity::kApi;
        k.correlation = a.correlation;
        k.queue = a.queue;
        k.bytes = a.bytes;
        k.name = std::move(a.name);
        kernels_done_.push_back(std::move(k));
      }
      window_open_ = false;
    }
    enabled_ = e;
  }
  bool enabled() const { return enabled_; }
  bool collects_kernels() const { return kernels_; }
  int64_t num_kernels() const {
    int64_t n = 0;
    for (auto& k : kernels_done_) n += k.api ? 0 : 1;
    return n;
  }

  void push(const std::string& name, const std::string& cat) {
    if (roctx().push) roctx().push(name.c_str());
    std::lock_guard<std::mutex> lk(mu_);
    auto& stack = open_[std::this_thread::get_id()];
    Range r;
    r.name = name;
    r.cat = cat;
    r.depth = (int)stack.size();
    r.thread = thread_row();
    r.host_begin = now_us();
    if (enabled_ && use_gpu_) {
      hipEventCreate(&r.ev_begin);
      hipEventRecord(r.ev_begin, c10::hip::getCurrentHIPStream().stream());
    }
    stack.push_back(std::move(r));
  }

  void pop() {
    if (roctx().pop) roctx().pop();
    std::lock_guard<std::mutex> lk(mu_);
    auto& stack = open_[std::this_thread::get_id()];
    TORCH_CHECK(!stack.empty(), "RangeTracer.pop without a matching push on this thread");
    Range r = std::move(stack.back());
    stack.pop_back();
    r.host_end = now_us();
    if (r.ev_begin) {
      hipEventCreate(&r.ev_end);
      hipEventRecord(r.ev_end, c10::hip::getCurrentHIPStream().stream());
    }
    if (enabled_) done_.push_back(std::move(r));
    else destroy(r);
  }

  // For pda::trace::Scope: closes the innermost range of this thread if there is one (a Scope that
  // began while the tracer was disabled pushed nothing).
  void pop_if_open() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      auto it = open_.find(std::this_thread::get_id());
      if (it == open_.end() || it->second.empty()) return;
    }
    pop();
  }

  void instant(const std::string& name) {
    std::lock_guard<std::mutex> lk(mu_);
    if (enabled_) instants_.emplace_back(name, now_us());
  }

  int64_t num_ranges() const { return (int64_t)done_.size(); }

  // Chrome trace: pid = rank; tid 0 = host ranges of the main thread, tid 1 = all ranges on the GPU
  // clock (stream order), tid 2.. = host ranges of other threads (the autograd engine's).
  std::string to_json(int64_t rank) {
    std::lock_guard<std::mutex> lk(mu_);
    std::ostringstream o;
    o.setf(std::ios::fixed);
    o.precision(3);
    o << "{\"schemaVersion\":1,\"distributedInfo\":{\"rank\":" << rank << "},\"traceEvents\":[\n";
    o << "{\"ph\":\"M\",\"pid\":" << rank << ",\"tid\":0,\"name\":\"thread_name\",\"args\":{\"name\":\"host ranges\"}},\n";
    o << "{\"ph\":\"M\",\"pid\":" << rank << ",\"tid\":1,\"name\":\"thread_name\",\"args\":{\"name\":\"gpu ranges (stream order)\"}}";
    for (int t = 1; t < next_thread_row_; ++t)
      o << ",\n{\"ph\":\"M\",\"pid\":" << rank << ",\"tid\":" << (t + 1)
        << ",\"name\":\"thread_name\",\"args\":{\"name\":\"host ranges (runtime thread " << t << ")\"}}";
    for (auto& r : done_) {
      o << ",\n{\"ph\":\"X\",\"cat\":\"" << json_escape(r.cat) << "\",\"name\":\"" << json_escape(r.name)
        << "\",\"pid\":" << rank << ",\"tid\":" << (r.thread == 0 ? 0 : r.thread + 1) << ",\"ts\":"
        << (r.host_begin - base_host_) << ",\"dur\":" << (r.host_end - r.host_begin) << "}";
      if (r.ev_begin && r.ev_end) {
        hipEventSynchronize(r.ev_end);
        float b_ms = 0.f, d_ms = 0.f;
        if (hipEventElapsedTime(&b_ms, base_event_, r.ev_begin) == hipSuccess &&
            hipEventElapsedTime(&d_ms, r.ev_begin, r.ev_end) == hipSuccess) {
          o << ",\n{\"ph\":\"X\",\"cat\":\"gpu_" << json_escape(r.cat) << "\",\"name\":\"" << json_escape(r.name)
            << "\",\"pid\":" << rank << ",\"tid\":1,\"ts\":" << (double)b_ms * 1e3 << ",\"dur\":" << (double)d_ms * 1e3
            << "}";
        }
      }
    }
    // Kernel rows: one per HIP queue in order of first appearance, collectives on a row of their own.
    std::map<uint64_t, int> queue_row, api_row;
    bool any_collective = false;
    for (auto& k : kernels_done_) {
      if (k.api) {
        if (!api_row.count(k.queue)) api_row.emplace(k.queue, 50 + (int)api_row.size());
      } else if (!k.copy && is_collective_name(k.name)) {
        any_collective = true;
      } else if (!queue_row.count(k.queue)) {
        queue_row.emplace(k.queue, 100 + (int)queue_row.size());
      }
    }
    for (auto& [t, row] : api_row)
      o << ",\n{\"ph\":\"M\",\"pid\":" << rank << ",\"tid\":" << row
        << ",\"name\":\"thread_name\",\"args\":{\"name\":\"HIP API calls (host thread " << t << ")\"}}";
    for (auto& [q, row] : queue_row)
      o << ",\n{\"ph\":\"M\",\"pid\":" << rank << ",\"tid\":" << row
        << ",\"name\":\"thread_name\",\"args\":{\"name\":\"HIP queue " << (row - 100) << " (id " << q << ")\"}}";
    if (any_collective)
      o << ",\n{\"ph\":\"M\",\"pid\":" << rank << ",\"tid\":" << kCollectiveRow
        << ",\"name\":\"thread_name\",\"args\":{\"name\":\"collectives (RCCL)\"}}";
    for (auto& k : kernels_done_) {
      if (k.api) {  // the host side of a launch / copy: same correlation id as the device-side record
        o << ",\n{\"ph\":\"X\",\"cat\":\"cuda_runtime\",\"name\":\"" << json_escape(k.name) << "\",\"pid\":" << rank
          << ",\"tid\":" << api_row[k.queue] << ",\"ts\":" << (k.ts_us - base_host_) << ",\"dur\":" << k.dur_us
          << ",\"args\":{\"correlation\":" << k.correlation << "}}";
        continue;
      }
      const bool coll = !k.copy && is_collective_name(k.name);
      o << ",\n{\"ph\":\"X\",\"cat\":\"" << (k.copy ? "gpu_memcpy" : "kernel") << "\",\"name\":\"" << json_escape(k.name)
        << "\",\"pid\":" << rank << ",\"tid\":" << (coll ? kCollectiveRow : queue_row[k.queue]) << ",\"ts\":"
        << (k.ts_us - base_host_) << ",\"dur\":" << k.dur_us << ",\"args\":{\"queue\":" << k.queue
        << ",\"correlation\":" << k.correlation;
      if (k.copy) o << ",\"bytes\":" << k.bytes;
      o << "}}";
    }
    for (auto& [name, ts] : instants_)
      o << ",\n{\"ph\":\"i\",\"s\":\"p\",\"name\":\"" << json_escape(name) << "\",\"pid\":" << rank
        << ",\"tid\":0,\"ts\":" << (ts - base_host_) << "}";
    o << "\n]}\n";
    return o.str();
  }

  void save(const std::string& path, int64_t rank) {
    std::string s = to_json(rank);
    std::ofstream f(path);
    TORCH_CHECK(f.good(), "cannot write trace to ", path);
    f << s;
  }

  void clear() {
    std::lock_guard<std::mutex> lk(mu_);
    for (auto& r : done_) destroy(r);
    done_.clear();
    instants_.clear();
    kernels_done_.clear();
  }

 private:
  static void destroy(Range& r) {
    if (r.ev_begin) hipEventDestroy(r.ev_begin);
    if (r.ev_end) hipEventDestroy(r.ev_end);
    r.ev_begin = r.ev_end = nullptr;
  }

  int thread_row() {  // called under mu_
    const auto id = std::this_thread::get_id();
    auto it = thread_rows_.find(id);
    if (it == thread_rows_.end()) it = thread_rows_.emplace(id, next_thread_row_++).first;
    return it->second;
  }

  struct Kernel {
    std::string name;
    double ts_us = 0, dur_us = 0;  // host clock
    uint64_t queue = 0, bytes = 0, correlation = 0;
    bool copy = false, api = false;  // api: a host-side HIP call (queue = host thread id)
  };

  std::map<std::thread::id, int> thread_rows_;
  int next_thread_row_ = 0;
  bool use_gpu_;
  bool kernels_;
  bool window_open_ = false;
  uint64_t window_k0_ns_ = 0;
  double window_h0_us_ = 0;
  std::vector<Kernel> kernels_done_;
  bool enabled_ = true;
  hipEvent_t base_event_ = nullptr;
  double base_host_ = 0;
  std::map<std::thread::id, std::vector<Range>> open_;
  std::vector<Range> done_;
  std::vector<std::pair<std::string, double>> instants_;
  std::mutex mu_;
};

std::atomic<RangeTracer*> g_active{nullptr};

void RangeTracer::activate() { g_active.store(this); }
void RangeTracer::deactivate() {
  RangeTracer* me = this;
  g_active.compare_exchange_strong(me, nullptr);
}

}  // namespace

namespace pda {
namespace trace {

bool active() { return g_active.load(std::memory_order_relaxed) != nullptr; }

bool push(const char* name, const char* cat) {
  RangeTracer* t = g_active.load();
  if (t == nullptr || !t->enabled()) return false;
  t->push(name, cat);
  return true;
}

void pop() {
  RangeTracer* t = g_active.load();
  if (t != nullptr) t->pop_if_open();
}

}  // namespace trace
}  // namespace pda

void bind_tracer(py::module_& m) {
  py::class_<RangeTracer>(m, "RangeTracer")
      .def(py::init<bool, bool>(), py::arg("use_gpu"), py::arg("kernels") = false)
      .def("activate", &RangeTracer::activate)
      .def("deactivate", &RangeTracer::deactivate)
      .def("push", &RangeTracer::push, py::arg("name"), py::arg("cat") = "range")
      .def("pop", &RangeTracer::pop)
      .def("instant", &RangeTracer::instant)
      .def("set_enabled", &RangeTracer::set_enabled)
      .def("enabled", &RangeTracer::enabled)
      .def("to_json", &RangeTracer::to_json, py::arg("rank") = 0)
      .def("save", &RangeTracer::save, py::arg("path"), py::arg("rank") = 0)
      .def("clear", &RangeTracer::clear)
      .def_property_readonly("num_ranges", &RangeTracer::num_ranges)
      .def_property_readonly("num_kernels", &RangeTracer::num_kernels)
      .def_property_readonly("collects_kernels", &RangeTracer::collects_kernels);
  m.def("kernel_trace_available", &pda::ktrace::available);
  m.def("kernel_trace_debug_counts", &pda::ktrace::debug_counts);
  m.def("kernel_trace_unavailable_reason", &pda::ktrace::unavailable_reason);
}
