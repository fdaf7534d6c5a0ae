// Native token-shard reader with a prefetch thread.
//
// Reads the fineweb ".bin" shard format (256 x int32 header: magic 20240520, version 1,
// token count; then uint16 tokens — assignments/assignment1/data/data_loader.py:70-75) through
// mmap, and produces the (rank-partitioned) batch windows of
// assignments/assignment1/data/distributed_data_loader.py:69-105 ahead of the training loop:
// a background thread widens uint16 -> int64 into a ring of caller-provided (pinned) host
// tensors, so the training thread only issues one non-blocking host-to-device copy per batch.
// With world_size = 1 the windows are exactly the single-process loader's.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <torch/extension.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace py = pybind11;

namespace {

constexpr int32_t kMagic = 20240520;
constexpr int32_t kVersion = 1;
constexpr size_t kHeaderBytes = 256 * 4;

struct MappedShard {
  void* base = nullptr;
  size_t bytes = 0;
  const uint16_t* tokens = nullptr;
  int64_t count = 0;

  void open(const std::string& path) {
    close();
    int fd = ::open(path.c_str(), O_RDONLY);
    TORCH_CHECK(fd >= 0, "cannot open shard ", path);
    struct stat st;
    TORCH_CHECK(::fstat(fd, &st) == 0, "cannot stat shard ", path);
    bytes = (size_t)st.st_size;
    TORCH_CHECK(bytes >= kHeaderBytes, path, ": truncated header");
    base = ::mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    TORCH_CHECK(base != MAP_FAILED, "mmap failed for ", path);
    const int32_t* header = reinterpret_cast<const int32_t*>(base);
    TORCH_CHECK(header[0] == kMagic, path, ": invalid magic number ", header[0]);
    TORCH_CHECK(header[1] == kVersion, path, ": unsupported version ", header[1]);
    count = header[2];
    TORCH_CHECK((size_t)count * 2 + kHeaderBytes <= bytes, path, ": token count mismatch");
    tokens = reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(base) + kHeaderBytes);
    ::madvise(base, bytes, MADV_SEQUENTIAL);
  }
  void close() {
    if (base && base != MAP_FAILED) ::munmap(base, bytes);
    base = nullptr;
    tokens = nullptr;
    count = 0;
  }
  ~MappedShard() { close(); }
};

class TokenPrefetcher {
 public:
  // `slots`: ring of 1-D int64 CPU tensors with at least local_batch * seq_len + 1 elements.
  TokenPrefetcher(std::vector<std::string> files, int64_t local_batch, int64_t seq_len, int64_t rank,
                  int64_t world, std::vector<at::Tensor> slots)
      : files_(std::move(files)), local_(local_batch * seq_len), rank_(rank), world_(world),
        slots_(std::move(slots)) {
    TORCH_CHECK(!files_.empty(), "Empty file list provided");
    TORCH_CHECK(rank_ >= 0 && rank_ < world_, "rank outside world");
    TORCH_CHECK(slots_.size() >= 2, "need at least two ring slots");
    for (auto& s : slots_)
      TORCH_CHECK(s.device().is_cpu() && s.scalar_type() == at::kLong && s.is_contiguous() &&
                      s.numel() >= local_ + 1,
                  "ring slots must be contiguous int64 CPU tensors of local_batch*seq_len+1 elements");
    state_.assign(slots_.size(), kFree);
    worker_ = std::thread([this] { this->run(); });
  }

  ~TokenPrefetcher() { stop(); }

  void stop() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    if (worker_.joinable()) worker_.join();
  }

  // Index of the next filled slot, or -1 when the data is exhausted. Blocks (GIL released).
  int64_t next() {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return state_[read_pos_] == kFull || (done_ && state_[read_pos_] != kFull) || !error_.empty(); });
    if (!error_.empty()) throw std::runtime_error(error_);
    if (state_[read_pos_] != kFull) return -1;
    const int64_t idx = (int64_t)read_pos_;
    state_[read_pos_] = kInUse;
    read_pos_ = (read_pos_ + 1) % slots_.size();
    return idx;
  }

  // The consumer is finished with slot `idx` (its host-to-device copy has completed).
  void release(int64_t idx) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      state_[(size_t)idx] = kFree;
    }
    cv_.notify_all();
  }

  int64_t shards_opened() const { return shards_opened_.load(); }

 private:
  enum SlotState { kFree, kFull, kInUse };

  void run() {
    try {
      MappedShard shard;
      size_t next_file = 0;
      int64_t pos = 0;
      size_t write_pos = 0;
      while (true) {
        // Same shard-advance rule on every rank: leave when the whole step does not fit.
        while (shard.tokens == nullptr || pos + world_ * local_ >= shard.count) {
          if (next_file >= files_.size()) {
            std::lock_guard<std::mutex> lk(mu_);
            done_ = true;
            cv_.notify_all();
            return;
          }
          shard.open(files_[next_file++]);
          shards_opened_.fetch_add(1);
          pos = 0;
        }
        {
          std::unique_lock<std::mutex> lk(mu_);
          cv_.wait(lk, [&] { return stop_ || state_[write_pos] == kFree; });
          if (stop_) return;
        }
        const uint16_t* src = shard.tokens + pos + rank_ * local_;
        int64_t* dst = slots_[write_pos].data_ptr<int64_t>();
        for (int64_t i = 0; i <= local_; ++i) dst[i] = (int64_t)src[i];
        pos += world_ * local_;
        {
          std::lock_guard<std::mutex> lk(mu_);
          state_[write_pos] = kFull;
        }
        cv_.notify_all();
        write_pos = (write_pos + 1) % slots_.size();
      }
    } catch (const std::exception& e) {
      std::lock_guard<std::mutex> lk(mu_);
      error_ = e.what();
      done_ = true;
      cv_.notify_all();
    }
  }

  std::vector<std::string> files_;
  int64_t local_, rank_, world_;
  std::vector<at::Tensor> slots_;
  std::vector<SlotState> state_;
  size_t read_pos_ = 0;
  bool stop_ = false, done_ = false;
  std::string error_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::thread worker_;
  std::atomic<int64_t> shards_opened_{0};
};

}  // namespace

void bind_loader(py::module_& m) {
  py::class_<TokenPrefetcher>(m, "TokenPrefetcher")
      .def(py::init<std::vector<std::string>, int64_t, int64_t, int64_t, int64_t, std::vector<at::Tensor>>(),
           py::arg("files"), py::arg("local_batch"), py::arg("seq_len"), py::arg("rank"), py::arg("world"),
           py::arg("slots"))
      .def("next", &TokenPrefetcher::next, py::call_guard<py::gil_scoped_release>())
      .def("release", &TokenPrefetcher::release)
      .def("stop", &TokenPrefetcher::stop, py::call_guard<py::gil_scoped_release>())
      .def_property_readonly("shards_opened", &TokenPrefetcher::shards_opened);
}
