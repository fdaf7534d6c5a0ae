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
// assignments/assignment1/train_baseline.py:79-87). Kernel rows come from a rocprofv3
// kernel trace of the same run; the roctx mirror puts both on one clock and
// utils/rocprof_trace.py merges them into the per-rank Chrome trace.
#include "range_tracer.h"

#include <c10/hip/HIPStream.h>
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <torch/extension.h>

#include <atomic>
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

class RangeTracer {
 public:
  explicit RangeTracer(bool use_gpu) : use_gpu_(use_gpu) {
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
    for (auto& r : done_) destroy(r);
    for (auto& kv : open_)
      for (auto& r : kv.second) destroy(r);
    if (base_event_) hipEventDestroy(base_event_);
  }

  // Makes this tracer the one the C++ runtime's pda::trace::Scope ranges go to.
  void activate();
  void deactivate();

  void set_enabled(bool e) { enabled_ = e; }
  bool enabled() const { return enabled_; }

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

  std::map<std::thread::id, int> thread_rows_;
  int next_thread_row_ = 0;
  bool use_gpu_;
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
      .def(py::init<bool>(), py::arg("use_gpu"))
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
      .def_property_readonly("num_ranges", &RangeTracer::num_ranges);
}
