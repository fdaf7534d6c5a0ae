// Python surface of the communication layer (`_C.Communicator`, `_C.CommWork`, factories).
#include "communicator.h"
#include "store.h"

#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>
#include <torch/csrc/distributed/c10d/ProcessGroup.hpp>
#include <torch/extension.h>

#include <thread>

namespace py = pybind11;

namespace pda {
namespace comm {
std::shared_ptr<Communicator> make_rccl_communicator(std::shared_ptr<Rendezvous> rv, int device, double timeout_s,
                                                     bool dual);
std::pair<bool, bool> rccl_dual_route(const std::shared_ptr<Communicator>& c, int set);
std::shared_ptr<Communicator> make_hanging_communicator(std::shared_ptr<Communicator> inner, int64_t healthy_calls,
                                                        double timeout_s);
std::shared_ptr<Communicator> make_socket_communicator(std::shared_ptr<Rendezvous> rv, double timeout_s);
std::shared_ptr<Communicator> make_c10d_communicator(c10::intrusive_ptr<c10d::ProcessGroup> pg, bool device_tensors);
std::shared_ptr<Communicator> make_recording_communicator(std::shared_ptr<Communicator> inner);
std::vector<std::tuple<std::string, int64_t, int>> recording_log(const std::shared_ptr<Communicator>& c, bool clear);
int64_t rccl_stream_handle(const std::shared_ptr<Communicator>& c, int q);
int rccl_version(const std::shared_ptr<Communicator>& c);
}  // namespace comm
}  // namespace pda

namespace {

using namespace pda::comm;

RedOp parse_op(const std::string& s) {
  if (s == "sum") return RedOp::kSum;
  if (s == "avg" || s == "mean") return RedOp::kAvg;
  if (s == "max") return RedOp::kMax;
  TORCH_CHECK(false, "unknown reduction '", s, "' (sum, avg, max)");
}

Queue parse_queue(int q) {
  TORCH_CHECK(q == 0 || q == 1, "queue must be 0 (gather) or 1 (reduce)");
  return q == 0 ? Queue::kGather : Queue::kReduce;
}

}  // namespace

void bind_comm(py::module_& m) {
  py::class_<Work, std::shared_ptr<Work>>(m, "CommWork")
      .def("wait", &Work::wait, py::call_guard<py::gil_scoped_release>())
      .def("synchronize", &Work::synchronize, py::call_guard<py::gil_scoped_release>())
      .def("completed", &Work::completed);

  py::class_<Communicator, std::shared_ptr<Communicator>>(m, "Communicator")
      .def_property_readonly("rank", &Communicator::rank)
      .def_property_readonly("world_size", &Communicator::world)
      .def_property_readonly("backend", &Communicator::backend)
      .def_property_readonly("on_gpu", &Communicator::on_gpu)
      .def_property_readonly("aborted", &Communicator::aborted)
      .def(
          "all_reduce",
          [](Communicator& c, const at::Tensor& t, const std::string& op, int queue) {
            return c.all_reduce(t, parse_op(op), parse_queue(queue));
          },
          py::arg("tensor"), py::arg("op") = "sum", py::arg("queue") = 1, py::call_guard<py::gil_scoped_release>())
      .def(
          "all_gather",
          [](Communicator& c, const at::Tensor& out, const at::Tensor& in, int queue) {
            return c.all_gather(out, in, parse_queue(queue));
          },
          py::arg("out"), py::arg("input"), py::arg("queue") = 0, py::call_guard<py::gil_scoped_release>())
      .def(
          "reduce_scatter",
          [](Communicator& c, const at::Tensor& out, const at::Tensor& in, const std::string& op, int queue) {
            return c.reduce_scatter(out, in, parse_op(op), parse_queue(queue));
          },
          py::arg("out"), py::arg("input"), py::arg("op") = "sum", py::arg("queue") = 1,
          py::call_guard<py::gil_scoped_release>())
      .def(
          "broadcast",
          [](Communicator& c, const at::Tensor& t, int root, int queue) { return c.broadcast(t, root, parse_queue(queue)); },
          py::arg("tensor"), py::arg("root") = 0, py::arg("queue") = 0, py::call_guard<py::gil_scoped_release>())
      .def("barrier", &Communicator::barrier, py::call_guard<py::gil_scoped_release>())
      .def("abort", &Communicator::abort, py::call_guard<py::gil_scoped_release>())
      .def("close", &Communicator::close, py::call_guard<py::gil_scoped_release>())
      .def(
          "exchange",
          [](Communicator& c, const py::bytes& mine) {
            std::vector<std::string> got;
            const std::string s = mine;
            {
              py::gil_scoped_release nogil;
              got = c.exchange(s);
            }
            py::list out;
            for (auto& g : got) out.append(py::bytes(g));
            return out;
          },
          py::arg("mine"))
      .def("counters",
           [](Communicator& c) {
             const Counters& k = c.counters();
             py::dict d;
             d["all_reduce"] = k.all_reduce;
             d["all_gather"] = k.all_gather;
             d["reduce_scatter"] = k.reduce_scatter;
             d["broadcast"] = k.broadcast;
             d["bytes"] = k.bytes;
             return d;
           })
      .def("stream_handle", [](std::shared_ptr<Communicator> c, int q) { return rccl_stream_handle(c, q); },
           py::arg("queue"), "hipStream_t of an RCCL communicator's queue as an integer (for traces)")
      .def("rccl_version", [](std::shared_ptr<Communicator> c) { return rccl_version(c); })
      .def("has_dual", [](std::shared_ptr<Communicator> c) { return rccl_dual_route(c, -1).first; },
           "an RCCL communicator that owns a second library communicator for the gather queue")
      .def("dual_route", [](std::shared_ptr<Communicator> c) { return rccl_dual_route(c, -1).second; })
      .def("set_dual_route", [](std::shared_ptr<Communicator> c, bool on) { return rccl_dual_route(c, on ? 1 : 0).second; },
           py::arg("on"), "route all-gathers / broadcasts to the second library communicator (true) or the shared one")
      .def("log", [](std::shared_ptr<Communicator> c, bool clear) { return recording_log(c, clear); },
           py::arg("clear") = false, "(kind, elements, queue) per collective of a recording communicator");

  m.def(
      "create_communicator",
      [](const std::string& kind, int rank, int world, const std::string& host, int port, int device, double timeout_s,
         const std::string& job_id, bool dual) -> std::shared_ptr<Communicator> {
        TORCH_CHECK(world >= 1 && rank >= 0 && rank < world, "create_communicator: bad rank / world ", rank, " / ", world);
        auto rv = rendezvous_from_env(rank, world, host, port, timeout_s, job_id);
        if (kind == "rccl") return make_rccl_communicator(std::move(rv), device, timeout_s, dual);
        if (kind == "socket") return make_socket_communicator(std::move(rv), timeout_s);
        TORCH_CHECK(false, "create_communicator: unknown kind '", kind, "' (rccl, socket)");
      },
      py::arg("kind"), py::arg("rank"), py::arg("world_size"), py::arg("host"), py::arg("port"), py::arg("device") = 0,
      py::arg("timeout_s") = 600.0, py::arg("job_id") = "", py::arg("dual") = false,
      py::call_guard<py::gil_scoped_release>());
  // Diagnosis: the communicator's fence pattern (current stream -> side stream -> current stream)
  // n times without a host synchronisation, with a chosen event life cycle:
  //   0 = fresh events, all kept until the end      1 = fresh events, destroyed right after use
  //   2 = ONE event re-recorded for every fence      3 = two events (ready / done) re-recorded
  m.def(
      "debug_fence_chain",
      [](int64_t side_handle, int n, int mode) {
        hipStream_t side = reinterpret_cast<hipStream_t>((uintptr_t)side_handle);
        hipStream_t cur = c10::hip::getCurrentHIPStream().stream();
        std::vector<hipEvent_t> keep;
        hipEvent_t fixed[2] = {nullptr, nullptr};
        auto make = [&](int which) {
          hipEvent_t e = nullptr;
          if (mode == 2 || mode == 3) {
            const int slot = mode == 2 ? 0 : which;
            if (!fixed[slot]) TORCH_CHECK(hipEventCreateWithFlags(&fixed[slot], hipEventDisableTiming) == hipSuccess);
            return fixed[slot];
          }
          TORCH_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess);
          if (mode == 0) keep.push_back(e);
          return e;
        };
        auto drop = [&](hipEvent_t e) {
          if (mode == 1) (void)hipEventDestroy(e);
        };
        for (int i = 0; i < n; ++i) {
          hipEvent_t ready = make(0);
          TORCH_CHECK(hipEventRecord(ready, cur) == hipSuccess);
          TORCH_CHECK(hipStreamWaitEvent(side, ready, 0) == hipSuccess);
          drop(ready);
          hipEvent_t done = make(1);
          TORCH_CHECK(hipEventRecord(done, side) == hipSuccess);
          TORCH_CHECK(hipStreamWaitEvent(cur, done, 0) == hipSuccess);
          drop(done);
        }
        TORCH_CHECK(hipStreamSynchronize(cur) == hipSuccess);
        for (hipEvent_t e : keep) (void)hipEventDestroy(e);
        for (hipEvent_t e : fixed)
          if (e) (void)hipEventDestroy(e);
        return (int64_t)(uintptr_t)cur;
      },
      py::arg("side_stream"), py::arg("n"), py::arg("mode"), py::call_guard<py::gil_scoped_release>());
  // Diagnosis / tests: `world` ranks (threads of this process, each with its own client connection)
  // run `rounds` barriers and exchanges through one store server; returns the number of keys the
  // server holds afterwards. Bounded by a small multiple of `world`, whatever `rounds` is.
  m.def(
      "debug_store_key_growth",
      [](int world, int rounds, int port) {
        std::vector<std::shared_ptr<Rendezvous>> rvs((size_t)world);
        std::vector<std::thread> threads;
        std::vector<std::string> errors((size_t)world);
        rvs[0] = rendezvous_from_env(0, world, "127.0.0.1", port, 30.0, "store-growth");
        const int bound = rvs[0]->server ? rvs[0]->server->port() : port;
        for (int r = 1; r < world; ++r) rvs[(size_t)r] = rendezvous_from_env(r, world, "127.0.0.1", bound, 30.0, "store-growth");
        for (int r = 0; r < world; ++r)
          threads.emplace_back([&, r] {
            try {
              for (int i = 0; i < rounds; ++i) {
                rvs[(size_t)r]->barrier("growth");
                auto got = rvs[(size_t)r]->exchange("growth", std::to_string(r * 1000 + i));
                for (int q = 0; q < world; ++q)
                  if (got[(size_t)q] != std::to_string(q * 1000 + i)) throw std::runtime_error("exchange returned a wrong entry");
              }
            } catch (const std::exception& e) {
              errors[(size_t)r] = e.what();
            }
          });
        for (auto& t : threads) t.join();
        for (auto& e : errors) TORCH_CHECK(e.empty(), "store round failed: ", e);
        const int64_t keys = rvs[0]->server ? (int64_t)rvs[0]->server->num_keys() : 0;
        for (int r = world - 1; r >= 0; --r) rvs[(size_t)r].reset();
        return keys;
      },
      py::arg("world"), py::arg("rounds"), py::arg("port"), py::call_guard<py::gil_scoped_release>());
  m.def("wrap_process_group", &make_c10d_communicator, py::arg("process_group"), py::arg("device_tensors") = false);
  m.def("recording_communicator", &make_recording_communicator, py::arg("inner"));
  m.def("hanging_communicator", &make_hanging_communicator, py::arg("inner"), py::arg("healthy_calls"),
        py::arg("timeout_s"),
        "fault injection: after `healthy_calls` collectives the next one never completes; a watchdog aborts after `timeout_s`");
}
