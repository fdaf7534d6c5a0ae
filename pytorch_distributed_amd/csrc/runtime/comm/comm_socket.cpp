// CPU implementation of pda::comm::Communicator: host tensors over a full TCP mesh.
//
// Purpose: the device-free configuration (small GPT / MLP data parallel on CPUs, world size 2+)
// and every multi-process test of the reducer, the sharding units, the trainers and the
// checkpoints — the same C++ runtime as on GPUs, only the wire differs.
//
// Mesh: every rank listens on an ephemeral port and publishes "host:port" in the rendezvous
// store; rank r connects to every lower rank and accepts from every higher one. Collectives
// are built from one primitive, a full-duplex pairwise exchange (poll-driven, so two peers
// that both send first cannot deadlock on full socket buffers), visiting peers in rank order:
// every rank walks its pairs in the same global (lexicographic) order, so there is no cycle.
//
//   all_gather      exchange own chunk with every peer
//   reduce_scatter  send chunk p to peer p, receive own chunk from every peer, reduce in RANK
//                   ORDER in fp32 / fp64 accumulators (the same summation order on every rank
//                   and every run: results are bitwise reproducible)
//   all_reduce      reduce_scatter + all_gather on a padded scratch copy
//   broadcast       root sends to every peer
//
// Collectives are synchronous: they have completed when the call returns.
#include <cstdlib>
#include "communicator.h"
#include "store.h"

#include <c10/util/BFloat16.h>
#include <c10/util/Exception.h>
#include <c10/util/Half.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <cstring>

namespace pda {
namespace comm {

void check_collective_args(const char* what, const at::Tensor& big, const at::Tensor& small, int world, bool on_gpu) {
  TORCH_CHECK(big.defined() && small.defined(), what, ": undefined tensor");
  TORCH_CHECK(big.is_contiguous() && small.is_contiguous(), what, ": tensors must be contiguous");
  TORCH_CHECK(big.scalar_type() == small.scalar_type(), what, ": dtype mismatch");
  TORCH_CHECK(big.is_cuda() == on_gpu && small.is_cuda() == on_gpu, what, ": this communicator works on ",
              on_gpu ? "GPU" : "CPU", " tensors");
  TORCH_CHECK(big.numel() == small.numel() * world, what, ": sizes ", big.numel(), " vs ", small.numel(), " x world ",
              world);
  TORCH_CHECK(big.storage().nbytes() > 0 || big.numel() == 0, what, ": buffer has no storage (released?)");
}

namespace {

double now_s() {
  using namespace std::chrono;
  return duration<double>(steady_clock::now().time_since_epoch()).count();
}

template <typename T, typename Acc>
void reduce_typed(void* dst, const std::vector<const void*>& srcs, int64_t n, RedOp op) {
  T* d = static_cast<T*>(dst);
  const double inv = 1.0 / (double)srcs.size();
  for (int64_t i = 0; i < n; ++i) {
    Acc acc = (Acc) static_cast<const T*>(srcs[0])[i];
    for (size_t s = 1; s < srcs.size(); ++s) {
      const Acc v = (Acc) static_cast<const T*>(srcs[s])[i];
      if (op == RedOp::kMax) acc = v > acc ? v : acc;
      else acc += v;
    }
    if (op == RedOp::kAvg) acc = (Acc)(acc * (Acc)inv);
    d[i] = (T)acc;
  }
}

void reduce_into(at::ScalarType t, void* dst, const std::vector<const void*>& srcs, int64_t n, RedOp op) {
  switch (t) {
    case at::kFloat: reduce_typed<float, float>(dst, srcs, n, op); break;
    case at::kDouble: reduce_typed<double, double>(dst, srcs, n, op); break;
    case at::kBFloat16: reduce_typed<c10::BFloat16, float>(dst, srcs, n, op); break;
    case at::kHalf: reduce_typed<c10::Half, float>(dst, srcs, n, op); break;
    case at::kLong: reduce_typed<int64_t, int64_t>(dst, srcs, n, op); break;
    case at::kInt: reduce_typed<int32_t, int64_t>(dst, srcs, n, op); break;
    default: TORCH_CHECK(false, "socket communicator: unsupported dtype ", t);
  }
}

constexpr int32_t kMeshMagic = 0x4D414450;  // "PDAM": first word of a peer's greeting on the mesh

class SocketCommunicator final : public Communicator {
 public:
  SocketCommunicator(std::shared_ptr<Rendezvous> rv, double timeout_s) : rv_(std::move(rv)), timeout_s_(timeout_s) {
    rank_ = rv_->rank;
    world_ = rv_->world;
    peers_.assign(world_, -1);
    if (world_ <= 1) return;
    static int generation = 0;
    const std::string base = "sock/" + std::to_string(generation++) + "/";
    int lfd = net::listen_on("", 0, world_);
    TORCH_CHECK(lfd >= 0, "socket communicator: cannot listen");
    // The address peers reach this rank by: the local end of its connection to the store — except
    // when that is a loopback address although rank 0 was named otherwise (rank 0 talking to its own
    // server that listens on every interface): peers on other nodes then need the master's name.
    std::string my_addr = rv_->client->local_address();
    const bool loop = my_addr.rfind("127.", 0) == 0 || my_addr == "::1";
    const std::string& mh = rv_->master_host;
    if (rank_ == 0 && loop && !mh.empty() && mh.rfind("127.", 0) != 0 && mh != "localhost" && mh != "::1") my_addr = mh;
    rv_->client->set(base + std::to_string(rank_), my_addr + ":" + std::to_string(net::bound_port(lfd)));
    for (int p = 0; p < rank_; ++p) {  // connect downwards
      const std::string addr = rv_->client->get(base + std::to_string(p));
      const size_t colon = addr.rfind(':');
      if (const char* inject = std::getenv("PDA_SOCKET_MESH_STRANGER")) {  // fault injection for the test of the
        if (inject[0] == '1') {                                             // accept loop: a foreign client first
          int stray = net::connect_to(addr.substr(0, colon), std::stoi(addr.substr(colon + 1)), timeout_s_);
          if (stray >= 0) {
            const char foreign[12] = {'P', 'D', 'A', 'S', 'T', 'O', 'R', '1', 3, 0, 0, 0};
            net::send_all(stray, foreign, sizeof(foreign));
            ::close(stray);
          }
        }
      }
      int fd = net::connect_to(addr.substr(0, colon), std::stoi(addr.substr(colon + 1)), timeout_s_);
      TORCH_CHECK(fd >= 0, "socket communicator: cannot connect to rank ", p, " at ", addr);
      const int32_t hello[2] = {kMeshMagic, rank_};
      net::send_all(fd, hello, sizeof(hello));
      peers_[p] = fd;
    }
    for (int k = rank_ + 1; k < world_;) {  // accept upwards (any order; the peer names itself)
      pollfd pf{lfd, POLLIN, 0};
      TORCH_CHECK(::poll(&pf, 1, (int)(timeout_s_ * 1000)) > 0, "socket communicator: timed out waiting for higher ranks");
      int fd = ::accept(lfd, nullptr, nullptr);
      TORCH_CHECK(fd >= 0, "socket communicator: accept failed");
      net::set_nodelay(fd);
      // The listening port is an ephemeral one, and a rendezvous-store client of this or another job that
      // scans a port range for its server may land on it: whoever does not open with the mesh's own
      // greeting is turned away and the wait goes on.
      int32_t hello[2] = {0, -1};
      bool greeted = false;
      try {
        net::recv_all(fd, hello, sizeof(hello), 5.0);
        greeted = hello[0] == kMeshMagic;
      } catch (const std::exception&) {
      }
      if (!greeted) {
        ::close(fd);
        continue;
      }
      const int32_t who = hello[1];
      TORCH_CHECK(who > rank_ && who < world_ && peers_[who] < 0, "socket communicator: unexpected peer ", who);
      peers_[who] = fd;
      ++k;
    }
    ::close(lfd);
    rv_->barrier("sock-mesh");
  }

  ~SocketCommunicator() override {
    for (int fd : peers_)
      if (fd >= 0) ::close(fd);
  }

  int rank() const override { return rank_; }
  int world() const override { return world_; }
  std::string backend() const override { return "socket"; }
  bool on_gpu() const override { return false; }
  bool aborted() const override { return aborted_; }

  std::shared_ptr<Work> all_gather(const at::Tensor& out, const at::Tensor& in, Queue) override {
    check_collective_args("all_gather", out, in, world_, false);
    count(Kind::kAllGather, out);
    std::lock_guard<std::mutex> lock(mu_);
    guard();
    const size_t chunk = (size_t)in.numel() * in.element_size();
    char* o = static_cast<char*>(out.data_ptr());
    if (o + chunk * rank_ != in.data_ptr()) std::memcpy(o + chunk * rank_, in.data_ptr(), chunk);
    for (int p = 0; p < world_; ++p) {
      if (p == rank_) continue;
      exchange_with(p, o + chunk * rank_, chunk, o + chunk * p, chunk);
    }
    return done();
  }

  std::shared_ptr<Work> reduce_scatter(const at::Tensor& out, const at::Tensor& in, RedOp op, Queue) override {
    check_collective_args("reduce_scatter", in, out, world_, false);
    count(Kind::kReduceScatter, in);
    std::lock_guard<std::mutex> lock(mu_);
    guard();
    reduce_scatter_raw(out.data_ptr(), in.data_ptr(), out.numel(), out.scalar_type(), op);
    return done();
  }

  std::shared_ptr<Work> all_reduce(const at::Tensor& t, RedOp op, Queue) override {
    check_collective_args("all_reduce", t, t, 1, false);
    count(Kind::kAllReduce, t);
    std::lock_guard<std::mutex> lock(mu_);
    guard();
    if (world_ == 1) return done();
    const int64_t n = t.numel();
    const int64_t per = (n + world_ - 1) / world_;
    const size_t es = t.element_size();
    std::vector<char> padded((size_t)per * world_ * es, 0), mine((size_t)per * es);
    std::memcpy(padded.data(), t.data_ptr(), (size_t)n * es);
    reduce_scatter_raw(mine.data(), padded.data(), per, t.scalar_type(), op);
    std::memcpy(padded.data() + (size_t)per * rank_ * es, mine.data(), mine.size());
    for (int p = 0; p < world_; ++p) {
      if (p == rank_) continue;
      exchange_with(p, mine.data(), mine.size(), padded.data() + (size_t)per * p * es, mine.size());
    }
    std::memcpy(t.data_ptr(), padded.data(), (size_t)n * es);
    return done();
  }

  std::shared_ptr<Work> broadcast(const at::Tensor& t, int root, Queue) override {
    check_collective_args("broadcast", t, t, 1, false);
    TORCH_CHECK(root >= 0 && root < world_, "broadcast: bad root ", root);
    count(Kind::kBroadcast, t);
    std::lock_guard<std::mutex> lock(mu_);
    guard();
    const size_t bytes = (size_t)t.numel() * t.element_size();
    if (rank_ == root) {
      for (int p = 0; p < world_; ++p)
        if (p != rank_) exchange_with(p, t.data_ptr(), bytes, nullptr, 0);
    } else {
      exchange_with(root, nullptr, 0, t.data_ptr(), bytes);
    }
    return done();
  }

  void barrier() override { rv_->barrier("sock"); }

  std::vector<std::string> exchange(const std::string& mine) override { return rv_->exchange("sock", mine); }

  void abort() override {
    aborted_ = true;
    for (int& fd : peers_) {
      if (fd >= 0) {
        ::shutdown(fd, SHUT_RDWR);
      }
    }
  }

 private:
  void guard() const { TORCH_CHECK(!aborted_, "communicator was aborted"); }
  static std::shared_ptr<Work> done() { return std::make_shared<CompletedWork>(); }

  void reduce_scatter_raw(void* out, const void* in, int64_t n_out, at::ScalarType dt, RedOp op) {
    const size_t es = c10::elementSize(dt), chunk = (size_t)n_out * es;
    const char* src = static_cast<const char*>(in);
    std::vector<std::vector<char>> inbox(world_);
    for (int p = 0; p < world_; ++p) {
      if (p == rank_) continue;
      inbox[p].resize(chunk);
      exchange_with(p, src + chunk * p, chunk, inbox[p].data(), chunk);
    }
    std::vector<const void*> srcs;
    for (int p = 0; p < world_; ++p) srcs.push_back(p == rank_ ? (const void*)(src + chunk * rank_) : inbox[p].data());
    reduce_into(dt, out, srcs, n_out, op);
  }

  // Sends `ns` bytes to and receives `nr` bytes from peer `p`, both directions progressing together.
  void exchange_with(int p, const void* send, size_t ns, void* recv, size_t nr) {
    const int fd = peers_[p];
    const char* s = static_cast<const char*>(send);
    char* r = static_cast<char*>(recv);
    const double deadline = now_s() + timeout_s_;
    while (ns > 0 || nr > 0) {
      pollfd pf{fd, (short)((ns > 0 ? POLLOUT : 0) | (nr > 0 ? POLLIN : 0)), 0};
      int pr = ::poll(&pf, 1, 1000);
      if (pr < 0 && errno != EINTR) TORCH_CHECK(false, "socket communicator: poll failed");
      TORCH_CHECK(!aborted_, "communicator was aborted");
      if (pr <= 0) {
        if (now_s() > deadline) {
          abort();
          TORCH_CHECK(false, "socket communicator: collective with rank ", p, " timed out after ", timeout_s_,
                      " s (a peer died or the ranks issued different collectives)");
        }
        continue;
      }
      if (nr > 0 && (pf.revents & (POLLIN | POLLHUP | POLLERR))) {
        ssize_t got = ::recv(fd, r, nr, MSG_DONTWAIT);
        if (got == 0) TORCH_CHECK(false, "socket communicator: rank ", p, " closed the connection");
        if (got < 0 && errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR)
          TORCH_CHECK(false, "socket communicator: recv from rank ", p, " failed: ", std::strerror(errno));
        if (got > 0) {
          r += got;
          nr -= (size_t)got;
        }
      }
      if (ns > 0 && (pf.revents & POLLOUT)) {
        ssize_t put = ::send(fd, s, ns, MSG_DONTWAIT | MSG_NOSIGNAL);
        if (put < 0 && errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR)
          TORCH_CHECK(false, "socket communicator: send to rank ", p, " failed: ", std::strerror(errno));
        if (put > 0) {
          s += put;
          ns -= (size_t)put;
        }
      }
    }
  }

  std::shared_ptr<Rendezvous> rv_;
  double timeout_s_;
  int rank_ = 0, world_ = 1;
  std::vector<int> peers_;
  std::mutex mu_;
  bool aborted_ = false;
  int64_t exchange_seq_ = 0;
};

}  // namespace

std::shared_ptr<Communicator> make_socket_communicator(std::shared_ptr<Rendezvous> rv, double timeout_s) {
  return std::make_shared<SocketCommunicator>(std::move(rv), timeout_s);
}

}  // namespace comm
}  // namespace pda
