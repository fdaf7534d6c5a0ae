// GPU activity records through roctracer's buffer API, resolved at run time from the
// libroctracer64 that is already in the process (PyTorch ships and loads one), so nothing links
// against a second copy of the HIP runtime and a box without the library only loses kernel rows.
//
// Delivery is asynchronous: the HIP runtime files a dispatch's device-side record some time after
// the dispatch has completed (when it retires the command), not when the host observes completion,
// so a record can arrive after hipDeviceSynchronize() has returned. stop() therefore keeps the
// domains enabled and flushes until the device-side record of the LAST kernel launch of the window
// (matched by correlation id with the launch call's own record) has arrived, bounded by a time-out;
// disabling right after the synchronisation loses everything since the last retirement.
//
// Life cycle: the first start() opens a memory pool of our own (never roctracer's default pool,
// which another tool in the process may own). A window enables the HIP_OPS activity domain on
// that pool; stop() disables it and flushes, and the pool's buffer callback — run on the
// library's thread during the flush and whenever a buffer fills — copies every record into a
// vector behind a mutex. Kernel names arrive mangled and are demangled once per distinct name.
#include "kernel_tracer.h"

#include <cxxabi.h>
#include <dlfcn.h>
#include <roctracer/roctracer.h>
#include <roctracer/roctracer_hip.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <unordered_map>

namespace pda {
namespace ktrace {
namespace {

struct Api {
  using open_pool_expl_t = roctracer_status_t (*)(const roctracer_properties_t*, roctracer_pool_t**);
  using close_pool_expl_t = roctracer_status_t (*)(roctracer_pool_t*);
  using enable_domain_expl_t = roctracer_status_t (*)(activity_domain_t, roctracer_pool_t*);
  using disable_domain_t = roctracer_status_t (*)(activity_domain_t);
  using flush_expl_t = roctracer_status_t (*)(roctracer_pool_t*);
  using next_record_t = roctracer_status_t (*)(const activity_record_t*, const activity_record_t**);
  using get_timestamp_t = roctracer_status_t (*)(roctracer_timestamp_t*);
  using error_string_t = const char* (*)();
  using op_string_t = const char* (*)(uint32_t, uint32_t, uint32_t);

  open_pool_expl_t open_pool = nullptr;
  close_pool_expl_t close_pool = nullptr;
  enable_domain_expl_t enable_domain = nullptr;
  disable_domain_t disable_domain = nullptr;
  flush_expl_t flush = nullptr;
  next_record_t next_record = nullptr;
  get_timestamp_t get_timestamp = nullptr;
  error_string_t error_string = nullptr;
  op_string_t op_string = nullptr;
  std::string why;
  bool ok = false;

  Api() {
    if (const char* off = std::getenv("PDA_KERNEL_TRACE"); off != nullptr && std::string(off) == "0") {
      why = "disabled by PDA_KERNEL_TRACE=0";
      return;
    }
    // Under rocprofv3 the kernel rows come from the external profiler; two collectors on one
    // process double the per-dispatch bookkeeping for nothing.
    if (std::getenv("ROCP_TOOL_LIBRARIES") != nullptr) {
      why = "an external rocprofiler tool is attached (ROCP_TOOL_LIBRARIES is set)";
      return;
    }
    void* lib = nullptr;
    for (const char* name : {"libroctracer64.so", "libroctracer64.so.4", "/opt/rocm/lib/libroctracer64.so"}) {
      lib = dlopen(name, RTLD_LAZY | RTLD_GLOBAL);
      if (lib != nullptr) break;
    }
    if (lib == nullptr) {
      why = "libroctracer64.so is not loadable";
      return;
    }
    auto sym = [&](const char* n) {
      void* p = dlsym(lib, n);
      if (p == nullptr && why.empty()) why = std::string("libroctracer64 lacks ") + n;
      return p;
    };
    open_pool = reinterpret_cast<open_pool_expl_t>(sym("roctracer_open_pool_expl"));
    close_pool = reinterpret_cast<close_pool_expl_t>(sym("roctracer_close_pool_expl"));
    enable_domain = reinterpret_cast<enable_domain_expl_t>(sym("roctracer_enable_domain_activity_expl"));
    disable_domain = reinterpret_cast<disable_domain_t>(sym("roctracer_disable_domain_activity"));
    flush = reinterpret_cast<flush_expl_t>(sym("roctracer_flush_activity_expl"));
    next_record = reinterpret_cast<next_record_t>(sym("roctracer_next_record"));
    get_timestamp = reinterpret_cast<get_timestamp_t>(sym("roctracer_get_timestamp"));
    error_string = reinterpret_cast<error_string_t>(sym("roctracer_error_string"));
    op_string = reinterpret_cast<op_string_t>(sym("roctracer_op_string"));
    ok = why.empty();
  }
};

Api& api() {
  static Api a;
  return a;
}

struct State {
  std::mutex mu;
  roctracer_pool_t* pool = nullptr;
  bool collecting = false;
  std::vector<Activity> records;
  std::unordered_map<std::string, std::string> demangled;
  uint64_t seen_records = 0, seen_buffers = 0, seen_other = 0;  // diagnosis: everything the callback was handed
  bool keep_api = false;  // keep the host-side API call records too (a row of launches in the trace)
  uint64_t max_launch_corr = 0;  // highest correlation id of a kernel-launch API call seen in this window
  uint64_t max_op_corr = 0;      // highest correlation id of a device-side record seen in this window
};

State& state() {
  static State* s = new State();  // leaked on purpose: the library's thread may outlive static destruction
  return *s;
}

const std::string& pretty_name(State& st, const char* mangled) {
  auto it = st.demangled.find(mangled);
  if (it != st.demangled.end()) return it->second;
  int status = 0;
  char* d = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
  std::string out = (status == 0 && d != nullptr) ? std::string(d) : std::string(mangled);
  std::free(d);
  // hipcc appends ".kd" to the kernel-descriptor symbol; tools show the bare function name.
  if (out.size() > 3 && out.compare(out.size() - 3, 3, ".kd") == 0) out.resize(out.size() - 3);
  return st.demangled.emplace(mangled, std::move(out)).first->second;
}

// API calls that say something about the timeline (launches, copies, fences); the hundreds of
// hipGetDevice / attribute queries libraries make per call are dropped.
bool worth_a_row(const char* name) {
  if (name == nullptr || std::strstr(name, "CallConfiguration") != nullptr) return false;
  for (const char* key : {"Launch", "Memcpy", "Memset", "Synchronize", "EventRecord", "StreamWaitEvent", "Malloc", "Free"})
    if (std::strstr(name, key) != nullptr) return true;
  return false;
}

void on_buffer(const char* begin, const char* end, void* /*arg*/) {
  State& st = state();
  std::lock_guard<std::mutex> lk(st.mu);
  const activity_record_t* rec = reinterpret_cast<const activity_record_t*>(begin);
  const activity_record_t* last = reinterpret_cast<const activity_record_t*>(end);
  ++st.seen_buffers;
  while (rec < last) {
    ++st.seen_records;
    if (rec->domain != ACTIVITY_DOMAIN_HIP_OPS) ++st.seen_other;
    if (rec->domain == ACTIVITY_DOMAIN_HIP_API) {
      const char* n = api().op_string ? api().op_string(ACTIVITY_DOMAIN_HIP_API, rec->op, rec->kind) : nullptr;
      if (n != nullptr && std::strstr(n, "Launch") != nullptr && std::strstr(n, "CallConfiguration") == nullptr &&
          rec->correlation_id > st.max_launch_corr)
        st.max_launch_corr = rec->correlation_id;
    }
    if (rec->domain == ACTIVITY_DOMAIN_HIP_OPS && rec->correlation_id > st.max_op_corr) st.max_op_corr = rec->correlation_id;
    if (rec->domain == ACTIVITY_DOMAIN_HIP_API && rec->end_ns >= rec->begin_ns && st.keep_api &&
        worth_a_row(api().op_string ? api().op_string(ACTIVITY_DOMAIN_HIP_API, rec->op, rec->kind) : nullptr)) {
      Activity a;
      a.kind = Activity::kApi;
      a.begin_ns = rec->begin_ns;
      a.end_ns = rec->end_ns;
      a.correlation = rec->correlation_id;
      a.queue = rec->thread_id;  // host thread of the call
      const char* n = api().op_string ? api().op_string(ACTIVITY_DOMAIN_HIP_API, rec->op, rec->kind) : nullptr;
      a.name = n != nullptr ? n : "hip api";
      st.records.push_back(std::move(a));
    }
    if (rec->domain == ACTIVITY_DOMAIN_HIP_OPS && rec->end_ns >= rec->begin_ns) {
      Activity a;
      a.begin_ns = rec->begin_ns;
      a.end_ns = rec->end_ns;
      a.device = rec->device_id;
      a.queue = rec->queue_id;
      a.correlation = rec->correlation_id;
      if (rec->op == HIP_OP_ID_DISPATCH) {
        a.kind = Activity::kKernel;
        a.name = rec->kernel_name != nullptr ? pretty_name(st, rec->kernel_name) : std::string("kernel");
        st.records.push_back(std::move(a));
      } else if (rec->op == HIP_OP_ID_COPY) {
        a.kind = Activity::kCopy;
        a.bytes = rec->bytes;
        a.name = "memcpy";
        st.records.push_back(std::move(a));
      }  // barriers (stream fences, event waits) carry no work
    }
    const activity_record_t* next = nullptr;
    if (api().next_record(rec, &next) != ROCTRACER_STATUS_SUCCESS || next == nullptr || next <= rec) break;
    rec = next;
  }
}

}  // namespace

bool available() { return api().ok; }

std::string unavailable_reason() { return api().ok ? std::string() : api().why; }

bool collecting() {
  State& st = state();
  std::lock_guard<std::mutex> lk(st.mu);
  return st.collecting;
}

bool start(bool keep_api_calls) {
  Api& a = api();
  if (!a.ok) return false;
  State& st = state();
  {
    std::lock_guard<std::mutex> lk(st.mu);
    if (st.collecting) return false;
    if (st.pool == nullptr) {
      roctracer_properties_t props{};
      props.buffer_size = size_t(8) << 20;
      props.buffer_callback_fun = on_buffer;
      props.buffer_callback_arg = nullptr;
      if (a.open_pool(&props, &st.pool) != ROCTRACER_STATUS_SUCCESS || st.pool == nullptr) {
        a.why = std::string("roctracer_open_pool_expl failed: ") + (a.error_string ? a.error_string() : "?");
        a.ok = false;
        st.pool = nullptr;
        return false;
      }
    }
    st.records.clear();
    st.keep_api = keep_api_calls;
    st.max_launch_corr = st.max_op_corr = 0;
  }
  // The HIP runtime only reports a kernel's device-side record when the API call that launched it
  // is being traced as well (the call is what carries the correlation id the record is filed under):
  // with the OPS domain alone, most dispatches of a single-stream program produce no record at all.
  if (a.enable_domain(ACTIVITY_DOMAIN_HIP_API, st.pool) != ROCTRACER_STATUS_SUCCESS ||
      a.enable_domain(ACTIVITY_DOMAIN_HIP_OPS, st.pool) != ROCTRACER_STATUS_SUCCESS) {
    a.why = std::string("roctracer_enable_domain_activity_expl failed: ") + (a.error_string ? a.error_string() : "?");
    a.ok = false;
    return false;
  }
  std::lock_guard<std::mutex> lk(st.mu);
  st.collecting = true;
  return true;
}

std::vector<Activity> stop() {
  Api& a = api();
  State& st = state();
  {
    std::lock_guard<std::mutex> lk(st.mu);
    if (!st.collecting) return {};
    st.collecting = false;
  }
  // Not under the mutex: these calls can run the buffer callback, which takes it.
  // Wait (bounded) for the device-side record of the window's last kernel launch: see the header comment.
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(3000);
  for (;;) {
    a.flush(st.pool);
    {
      std::lock_guard<std::mutex> lk(st.mu);
      if (st.max_op_corr >= st.max_launch_corr) break;
    }
    if (std::chrono::steady_clock::now() > deadline) break;
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
  a.disable_domain(ACTIVITY_DOMAIN_HIP_OPS);
  a.disable_domain(ACTIVITY_DOMAIN_HIP_API);
  a.flush(st.pool);
  std::lock_guard<std::mutex> lk(st.mu);
  std::vector<Activity> out;
  out.swap(st.records);
  return out;
}

std::vector<uint64_t> debug_counts() {
  State& st = state();
  std::lock_guard<std::mutex> lk(st.mu);
  return {st.seen_buffers, st.seen_records, st.seen_other, st.max_launch_corr, st.max_op_corr};
}

uint64_t now_ns() {
  Api& a = api();
  if (!a.ok) return 0;
  roctracer_timestamp_t t = 0;
  return a.get_timestamp(&t) == ROCTRACER_STATUS_SUCCESS ? uint64_t(t) : 0;
}

}  // namespace ktrace
}  // namespace pda
