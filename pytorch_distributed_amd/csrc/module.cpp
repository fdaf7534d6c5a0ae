// Python module `pytorch_distributed_amd._C`: one shared object holding the
// gfx950 kernels' bindings and the native runtime (gradient reducer, flat
// parameter sharding units, communicators, token-shard loader, range tracer, GEMM planner).
#include <torch/extension.h>

namespace py = pybind11;

void bind_ops(py::module_& m);
void bind_comm(py::module_& m);
void bind_reducer(py::module_& m);
void bind_fsdp(py::module_& m);
void bind_loader(py::module_& m);
void bind_tracer(py::module_& m);
void bind_recompute(py::module_& m);
namespace pda {
void bind_gemm(py::module_& m);
}

PYBIND11_MODULE(_C, m) {
  m.doc() = "MI355X-native kernels and runtime for pytorch_distributed_amd";
  m.attr("arch") = "gfx950";
  bind_ops(m);
  bind_comm(m);
  bind_reducer(m);
  bind_fsdp(m);
  bind_loader(m);
  bind_tracer(m);
  bind_recompute(m);
  pda::bind_gemm(m);
}
