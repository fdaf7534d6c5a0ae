// Range annotations from inside the C++ runtime. A Scope records a named host + GPU range into
// the currently active RangeTracer (range_tracer.cpp) and mirrors it to roctx; with no active
// tracer it costs one atomic load.
#pragma once

namespace pda {
namespace trace {

bool active();
bool push(const char* name, const char* cat);  // false: no tracer is recording, nothing was opened
void pop();

class Scope {
 public:
  explicit Scope(const char* name, const char* cat = "runtime") : on_(active() && push(name, cat)) {}
  ~Scope() {
    if (on_) pop();
  }
  Scope(const Scope&) = delete;
  Scope& operator=(const Scope&) = delete;

 private:
  bool on_;
};

}  // namespace trace
}  // namespace pda
