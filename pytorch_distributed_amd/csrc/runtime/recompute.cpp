#include "recompute.h"

#include <torch/extension.h>

#include <mutex>
#include <vector>

namespace py = pybind11;

namespace pda {
namespace recompute {

namespace {
std::mutex g_mu;
int g_depth = 0;
std::vector<std::function<void()>> g_deferred;
}  // namespace

int depth() {
  std::lock_guard<std::mutex> lock(g_mu);
  return g_depth;
}

void enter() {
  std::lock_guard<std::mutex> lock(g_mu);
  ++g_depth;
}

void exit() {
  std::vector<std::function<void()>> run;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    TORCH_CHECK(g_depth > 0, "recompute::exit without a matching enter");
    if (--g_depth == 0) run.swap(g_deferred);
  }
  for (auto& fn : run) fn();
}

void defer(std::function<void()> fn) {
  {
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_depth > 0) {
      g_deferred.push_back(std::move(fn));
      return;
    }
  }
  fn();
}

}  // namespace recompute
}  // namespace pda

void bind_recompute(py::module_& m) {
  m.def("recompute_enter", &pda::recompute::enter);
  m.def("recompute_exit", &pda::recompute::exit);
  m.def("recompute_depth", &pda::recompute::depth);
}
