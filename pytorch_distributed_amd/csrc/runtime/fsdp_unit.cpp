// Flat-parameter sharding runtime: the engine underneath pytorch_distributed_amd.parallel.FSDP.
//
// ShardedUnit — one communication unit (a transformer block, a group of blocks, or the
// root with embeddings / final norm / LM head):
//   * all parameters of the unit live at 128-byte-aligned offsets of ONE flat buffer,
//     padded to world_size equal shards; this rank keeps shard `rank` as fp32 master
//     weights (plus a bf16 mirror when the model is bf16, written by the optimizer kernel);
//   * the unsharded flat parameter is a tensor with RESIZABLE storage of which the module's
//     parameters are permanent views: "unshard" allocates the storage and all-gathers into
//     it, "reshard" releases the storage, nothing is re-pointed. With 288 GB of HBM per GPU
//     the policy knob is simply how many units stay allocated;
//   * in backward, C++ post-accumulate hooks move each parameter's fresh gradient into the
//     unit's flat gradient buffer; when the last one has arrived the buffer is
//     reduce-scattered (FULL_SHARD / SHARD_GRAD_OP) or all-reduced (NO_SHARD) asynchronously
//     and, under FULL_SHARD, the parameters are resharded at once.
//
// ShardedRuntime — the units of one model in execution order and the whole schedule:
//   * pre_forward(i): all-gather unit i (if it is not already in flight), start the
//     all-gathers of the next `forward_prefetch` units on the gather queue, then make the
//     compute stream wait for unit i only;
//   * post_forward(i, outputs): reshard (FULL_SHARD) and plant C++ gradient hooks on the
//     unit's outputs: the first gradient that reaches them in backward is the unit's
//     "pre-backward" moment — unit i is all-gathered again and the units backward will
//     visit next are prefetched. No Python runs in backward;
//   * one end-of-backward callback waits for all outstanding reductions and publishes the
//     reduced shard as `.grad` of each unit's flat shard parameter;
//   * resident budget (set_resident_budget): FULL_SHARD units, counted from the LAST one (the
//     first that backward needs), keep their unsharded parameters from forward to their own
//     backward while their bytes fit the budget. Those units are not re-gathered in backward —
//     the all-gathers that sit exposed at the start of backward disappear — and their memory is
//     still released unit by unit as backward passes them. With 288 GB per GPU the whole bf16
//     model of the GPT-2 / 7B-class configurations fits a small fraction of free memory. How
//     many unsharded units exist at any time is otherwise bounded by the prefetch depths
//     (depth + 1 in either direction), which is the rate limit on all-gathers in flight.
//
// All-gathers run on the communicator's gather queue, reductions on its reduce queue; neither
// ever sits on the compute stream. (The collective library runs one communicator's collectives in
// issue order; per unit that is a reduce-scatter plus the next all-gather under a whole block's
// backward, see communicator.h.)
//
// Safety rule for every buffer: storage is only released after the collective that reads
// or writes it has been waited on from the stream that will reuse the memory.
//
// Replaces the torch FSDP wrapping at assignments/assignment1/train_fsdp.py:64-81.
#include <c10/core/StorageImpl.h>
#include <torch/csrc/autograd/engine.h>
#include <torch/csrc/autograd/graph_task.h>
#include <torch/csrc/autograd/utils/lambda_post_hook.h>
#include <torch/extension.h>

#include "comm/communicator.h"
#include "comm_debug.h"
#include "range_tracer.h"
#include "recompute.h"

#include <memory>
#include <mutex>
#include <vector>

namespace py = pybind11;

namespace {

using pda::comm::Communicator;
using pda::comm::Kind;
using pda::comm::Queue;
using pda::comm::RedOp;
using pda::comm::Work;

constexpr int64_t kAlignElems = 64;
enum Strategy : int { kFullShard = 0, kShardGradOp = 1, kNoShard = 2 };

int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

// A 1-D tensor whose storage can be released and re-allocated while views of it stay alive.
at::Tensor make_resizable(int64_t numel, const at::TensorOptions& opts) {
  at::Tensor t = at::empty({numel}, opts);
  t.storage().unsafeGetStorageImpl()->set_resizable(true);
  return t;
}

void release_storage(const at::Tensor& t) {
  t.storage().unsafeGetStorageImpl()->reset();
}

void allocate_storage(const at::Tensor& t, int64_t numel) {
  auto* impl = t.storage().unsafeGetStorageImpl();
  if (impl->nbytes() > 0) return;
  const size_t nbytes = (size_t)numel * t.element_size();
  c10::DeviceGuard guard(t.device());
  impl->set_data_ptr_noswap(impl->allocator()->allocate(nbytes));
  impl->set_nbytes(nbytes);
}

class ShardedRuntime;

class ShardedUnit {
 public:
  // accum_fp32: the unit's unsharded gradient buffer is fp32 although the parameters are not, so
  // the micro-batches of a no_sync window (and the slices of a weight-gradient kernel) are summed
  // in fp32 (the reference accumulates in fp32: assignments/assignment1/train/trainer.py:57-61);
  // at the synchronising backward the sum is narrowed to the parameter dtype for the collective,
  // or reduced as it is with reduce_fp32.
  ShardedUnit(std::vector<at::Tensor> params, std::shared_ptr<Communicator> comm, int strategy,
              bool force_collectives, bool reduce_fp32, bool accum_fp32)
      : params_(std::move(params)), comm_(std::move(comm)), strategy_(strategy),
        force_collectives_(force_collectives), reduce_fp32_(reduce_fp32), accum_fp32_(accum_fp32) {
    TORCH_CHECK(!params_.empty(), "ShardedUnit: no parameters");
    TORCH_CHECK(strategy_ >= 0 && strategy_ <= 2, "ShardedUnit: bad strategy");
    world_ = comm_ ? comm_->world() : 1;
    rank_ = comm_ ? comm_->rank() : 0;
    const at::Tensor& p0 = params_[0];
    if (comm_) {
      TORCH_CHECK(comm_->on_gpu() == p0.is_cuda(), "ShardedUnit: the '", comm_->backend(),
                  "' communicator does not work on ", p0.is_cuda() ? "GPU" : "CPU", " tensors");
    }
    int64_t off = 0;
    for (auto& p : params_) {
      TORCH_CHECK(p.is_leaf() && p.requires_grad(), "ShardedUnit: parameters must be trainable leaves");
      TORCH_CHECK(p.scalar_type() == p0.scalar_type() && p.device() == p0.device(),
                  "ShardedUnit: one dtype and one device per unit");
      offsets_.push_back(off);
      off += align_up(p.numel(), kAlignElems);
    }
    total_numel_ = off;
    const int64_t parts = strategy_ == kNoShard ? 1 : world_;
    shard_numel_ = align_up((total_numel_ + parts - 1) / parts, kAlignElems);
    padded_numel_ = shard_numel_ * parts;
    shard_index_ = strategy_ == kNoShard ? 0 : rank_;

    at::NoGradGuard ng;
    const auto opts = p0.options().requires_grad(false);
    full_param_ = make_resizable(padded_numel_, opts);
    full_param_.zero_();
    for (size_t i = 0; i < params_.size(); ++i) {
      full_param_.narrow(0, offsets_[i], params_[i].numel()).copy_(params_[i].reshape({-1}));
      params_[i].set_data(full_param_.narrow(0, offsets_[i], params_[i].numel()).view(params_[i].sizes()));
    }
    at::Tensor shard = full_param_.narrow(0, shard_index_ * shard_numel_, shard_numel_);
    mixed_ = p0.scalar_type() != at::kFloat;
    if (strategy_ == kNoShard) {
      if (mixed_) {
        shard_master_ = shard.to(at::kFloat);
        shard_lowp_ = full_param_;  // the optimizer's bf16 mirror IS the live parameter buffer
      } else {
        shard_master_ = full_param_;
      }
    } else {
      shard_master_ = mixed_ ? shard.to(at::kFloat) : shard.clone();
      if (mixed_) shard_lowp_ = shard.clone();
    }
    unsharded_ = true;
    if (strategy_ != kNoShard) reshard();
    accum_fp32_ = accum_fp32_ && mixed_;
    full_grad_ = make_resizable(padded_numel_, accum_fp32_ ? opts.dtype(at::kFloat) : opts);
    release_storage(full_grad_);
    if (accum_fp32_) {
      if (!reduce_fp32_) {
        narrow_grad_ = make_resizable(padded_numel_, opts);
        release_storage(narrow_grad_);
      }
    } else if (reduce_fp32_ && mixed_) {
      wide_grad_ = make_resizable(padded_numel_, opts.dtype(at::kFloat));
      release_storage(wide_grad_);
    }
    grad_written_.assign(params_.size(), false);
    sunk_.assign(params_.size(), false);
    counted_.assign(params_.size(), false);
    install_hooks();
  }

  ~ShardedUnit() { remove_hooks(); }

  void remove_hooks() {
    for (size_t i = 0; i < accumulators_.size(); ++i)
      if (accumulators_[i]) accumulators_[i]->del_post_hook(hook_keys_[i]);
    accumulators_.clear();
    hook_keys_.clear();
  }

  bool communicates() const { return comm_ && (world_ > 1 || force_collectives_); }

  // ---- parameters ---------------------------------------------------------------
  // Starts (or joins) the all-gather of this unit's parameters; returns at once.
  void unshard() {
    if (strategy_ == kNoShard || unsharded_ || ag_work_ || ag_issued_) return;
    allocate_storage(full_param_, padded_numel_);
    const at::Tensor& src = mixed_ ? shard_lowp_ : shard_master_;
    if (!communicates()) {
      at::NoGradGuard ng;
      full_param_.copy_(src);
    } else {
      pda::comm_debug::before(comm_, Kind::kAllGather, full_param_.numel(), "ShardedUnit");
      ag_work_ = comm_->all_gather(full_param_, src, Queue::kGather);
      pda::comm_debug::after(ag_work_);
      ++num_allgathers_;
    }
    ag_issued_ = true;
  }

  // Makes the current stream (GPU) / the host (CPU) wait until the parameters are usable.
  void wait_unshard() {
    if (strategy_ == kNoShard || unsharded_) return;
    if (!ag_issued_) unshard();
    if (ag_work_) {
      pda::trace::Scope range("all_gather wait", "comm");  // on the GPU clock: the exposed part of the all-gather
      ag_work_->wait();
      ag_work_.reset();
    }
    ag_issued_ = false;
    unsharded_ = true;
  }

  void reshard() {
    if (strategy_ == kNoShard) return;
    if (ag_issued_) wait_unshard();  // never release a buffer with a collective in flight
    if (!unsharded_) return;
    release_storage(full_param_);
    unsharded_ = false;
  }

  bool is_unsharded() const { return unsharded_ || strategy_ == kNoShard; }
  bool unshard_in_flight() const { return ag_issued_; }

  // After the module parameters were overwritten in place (load_state_dict on an unsharded
  // unit): refresh this rank's shard (master + mirror) from the flat buffer.
  void refresh_shard_from_full() {
    TORCH_CHECK(is_unsharded(), "refresh_shard_from_full: unit is sharded");
    at::NoGradGuard ng;
    if (strategy_ == kNoShard) {
      if (mixed_) shard_master_.copy_(full_param_);
      return;
    }
    at::Tensor shard = full_param_.narrow(0, shard_index_ * shard_numel_, shard_numel_);
    shard_master_.copy_(shard);
    if (mixed_) shard_lowp_.copy_(shard);
  }

  // All ranks' shards of a shard-shaped tensor (fp32 master weights, optimizer moments),
  // concatenated into a fresh flat tensor of padded_numel elements. Collective unless the
  // unit is replicated. This is what makes full checkpoints fp32-faithful: the master
  // weights are gathered themselves instead of being re-derived from the bf16 parameters.
  at::Tensor gather_shards(const at::Tensor& shard) {
    TORCH_CHECK(shard.numel() == shard_numel_, "gather_shards: expected ", shard_numel_, " elements, got ", shard.numel());
    at::NoGradGuard ng;
    at::Tensor flat = shard.detach().reshape({-1}).contiguous();
    if (padded_numel_ == shard_numel_ || world_ == 1 || !comm_) return flat.clone();
    at::Tensor full = at::empty({padded_numel_}, flat.options());
    auto w = comm_->all_gather(full, flat, Queue::kGather);
    w->wait();
    return full;
  }

  // Inverse for loading: overwrites the fp32 master shard (and the low-precision mirror) with
  // this rank's slice of a full flat tensor laid out like the unit.
  void load_master_from_full(const at::Tensor& full) {
    TORCH_CHECK(full.numel() == padded_numel_ || full.numel() == total_numel_, "load_master_from_full: size mismatch");
    at::NoGradGuard ng;
    at::Tensor padded = full.reshape({-1});
    if (padded.numel() != padded_numel_) {
      at::Tensor tmp = at::zeros({padded_numel_}, padded.options());
      tmp.narrow(0, 0, padded.numel()).copy_(padded);
      padded = tmp;
    }
    at::Tensor mine = strategy_ == kNoShard ? padded : padded.narrow(0, shard_index_ * shard_numel_, shard_numel_);
    shard_master_.copy_(mine);
    if (mixed_) shard_lowp_.copy_(strategy_ == kNoShard ? padded : mine);
    invalidate_unsharded();
  }

  // The shard changed behind an unsharded copy's back (a checkpoint was loaded after a forward
  // that left the unit gathered: the root, SHARD_GRAD_OP units and resident units stay unsharded
  // until a backward ends): drop the copy so the next forward gathers the new values.
  void invalidate_unsharded() {
    if (strategy_ != kNoShard && (unsharded_ || ag_issued_)) reshard();
  }

  // ---- accessors for the Python wrapper -------------------------------------------
  at::Tensor shard_master() const { return shard_master_; }
  at::Tensor shard_lowp() const { return shard_lowp_; }
  at::Tensor full_param() const { return full_param_; }
  void set_flat_param(at::Tensor p) { flat_param_ = std::move(p); }
  std::vector<at::Tensor> params() const { return params_; }
  std::vector<int64_t> offsets() const { return offsets_; }
  int64_t shard_numel() const { return shard_numel_; }
  int64_t padded_numel() const { return padded_numel_; }
  int64_t total_numel() const { return total_numel_; }
  int64_t shard_index() const { return shard_index_; }
  int strategy() const { return strategy_; }
  bool mixed() const { return mixed_; }
  bool accumulates_fp32() const { return accum_fp32_; }
  int64_t num_allgathers() const { return num_allgathers_; }
  int64_t num_reductions() const { return num_reductions_; }
  bool reshard_after_forward = true;  // FULL_SHARD non-root units; the wrapper clears it for the root
  bool window_resident_ = false;      // set with the runtime's resident budget

  // Gradient sink: hands an op the slice of the flat gradient buffer that belongs to parameter
  // `i`, so the op's weight-gradient kernel writes (or, second element true, accumulates) there
  // directly and the per-parameter copy of the hook below disappears. The op then returns an
  // uninitialised placeholder gradient to autograd purely to make the accumulator hook fire.
  std::pair<at::Tensor, bool> grad_slot(size_t i);

 private:
  friend class ShardedRuntime;

  void install_hooks() {
    for (size_t i = 0; i < params_.size(); ++i) {
      auto acc = torch::autograd::impl::grad_accumulator(params_[i]);
      auto key = acc->add_post_hook(std::make_unique<torch::autograd::utils::LambdaPostHook>(
          [this, i](const torch::autograd::variable_list& outputs, const torch::autograd::variable_list&) {
            this->on_grad_ready(i);
            return outputs;
          }));
      accumulators_.push_back(std::move(acc));
      hook_keys_.push_back(key);
    }
  }

  void on_grad_ready(size_t i);  // defined after ShardedRuntime

  void reset_pass() {
    pending_ = params_.size();
    std::fill(counted_.begin(), counted_.end(), false);
    reduce_launched_ = false;
    unit_done_ = false;
  }

  // Everything the backward did not write must reduce as zeros: parameters that received no
  // gradient, the alignment gap after a parameter and the tail padding. Zeroing those ranges
  // here replaces a memset of the whole flat gradient buffer at the start of every backward
  // (a full extra pass over the unit's gradients per step: sinks and copies overwrite the rest).
  void zero_unwritten() {
    at::NoGradGuard ng;
    int64_t run_begin = -1, run_end = -1;
    auto add = [&](int64_t b, int64_t e) {
      if (e <= b) return;
      if (run_begin >= 0 && b == run_end) {
        run_end = e;
        return;
      }
      if (run_begin >= 0) full_grad_.narrow(0, run_begin, run_end - run_begin).zero_();
      run_begin = b;
      run_end = e;
    };
    for (size_t i = 0; i < params_.size(); ++i) {
      const int64_t begin = offsets_[i], used_end = begin + params_[i].numel();
      const int64_t next = i + 1 < params_.size() ? offsets_[i + 1] : padded_numel_;
      if (!grad_written_[i]) add(begin, next);
      else add(used_end, next);
    }
    if (run_begin >= 0) full_grad_.narrow(0, run_begin, run_end - run_begin).zero_();
  }

  void launch_reduce() {
    reduce_launched_ = true;
    if (!grad_allocated_) return;
    pda::trace::Scope range(strategy_ == kNoShard ? "all_reduce launch" : "reduce_scatter launch", "comm");
    zero_unwritten();
    // fp32 reduction of low-precision gradients: the collective runs on a widened copy (twice
    // the bytes on the wire, no rounding between the ranks' contributions); with an fp32 master
    // shard the reduced fp32 result is then the optimizer's gradient as it is.
    const bool comms = communicates();
    at::Tensor in = full_grad_;
    if (narrow_grad_.defined() && comms) {
      at::NoGradGuard ng;
      allocate_storage(narrow_grad_, padded_numel_);
      narrow_grad_.copy_(full_grad_);  // the fp32 sum of the window, rounded once
      in = narrow_grad_;
    } else if (wide_grad_.defined() && comms) {
      at::NoGradGuard ng;
      allocate_storage(wide_grad_, padded_numel_);
      wide_grad_.copy_(full_grad_);
      in = wide_grad_;
    }
    if (strategy_ == kNoShard) {
      reduced_ = in;
      if (comms) {
        pda::comm_debug::before(comm_, Kind::kAllReduce, in.numel(), "ShardedUnit");
        rs_work_ = comm_->all_reduce(in, RedOp::kAvg, Queue::kReduce);
        pda::comm_debug::after(rs_work_);
        ++num_reductions_;
      }
    } else if (!comms) {
      reduced_ = full_grad_.narrow(0, 0, shard_numel_);
    } else {
      reduced_ = at::empty({shard_numel_}, in.options());
      pda::comm_debug::before(comm_, Kind::kReduceScatter, in.numel(), "ShardedUnit");
      rs_work_ = comm_->reduce_scatter(reduced_, in, RedOp::kAvg, Queue::kReduce);
      pda::comm_debug::after(rs_work_);
      ++num_reductions_;
    }
  }

  // Waits for the reduction and publishes it as flat_param.grad (adds if one exists).
  void finish_reduce() {
    if (!reduced_.defined()) return;
    if (rs_work_) {
      rs_work_->wait();
      rs_work_.reset();
    }
    at::NoGradGuard ng;
    TORCH_CHECK(flat_param_.defined(), "ShardedUnit: flat parameter was never registered");
    at::Tensor& g = flat_param_.mutable_grad();
    const bool aliases_buffer = reduced_.storage().is_alias_of(full_grad_.storage()) ||
                                (wide_grad_.defined() && reduced_.storage().is_alias_of(wide_grad_.storage())) ||
                                (narrow_grad_.defined() && reduced_.storage().is_alias_of(narrow_grad_.storage()));
    if (!g.defined()) {
      if (reduced_.scalar_type() == flat_param_.scalar_type() && !aliases_buffer) {
        g = reduced_;
      } else {
        g = reduced_.to(flat_param_.scalar_type(), /*non_blocking=*/false, /*copy=*/true);
      }
    } else {
      g.add_(reduced_);
    }
    reduced_.reset();
    if (wide_grad_.defined()) release_storage(wide_grad_);
    if (narrow_grad_.defined()) release_storage(narrow_grad_);
    release_storage(full_grad_);
    grad_allocated_ = false;
    std::fill(grad_written_.begin(), grad_written_.end(), false);
  }

  // Forget an interrupted pass (a backward that raised): nothing may stay in flight.
  void abandon_pass() {
    if (rs_work_) {
      rs_work_->wait();
      rs_work_.reset();
    }
    reduced_.reset();
    if (wide_grad_.defined()) release_storage(wide_grad_);
    if (narrow_grad_.defined()) release_storage(narrow_grad_);
    if (grad_allocated_) {
      release_storage(full_grad_);
      grad_allocated_ = false;
    }
    std::fill(grad_written_.begin(), grad_written_.end(), false);
    std::fill(sunk_.begin(), sunk_.end(), false);
    reset_pass();
  }

  std::vector<at::Tensor> params_;
  std::shared_ptr<Communicator> comm_;
  int strategy_;
  bool force_collectives_;
  bool reduce_fp32_ = false;  // reduce low-precision gradients through an fp32 copy
  bool accum_fp32_ = false;   // keep the unsharded gradients of low-precision parameters in fp32
  int world_ = 1, rank_ = 0;
  bool mixed_ = false;
  std::vector<int64_t> offsets_;
  int64_t total_numel_ = 0, shard_numel_ = 0, padded_numel_ = 0, shard_index_ = 0;

  at::Tensor shard_master_, shard_lowp_, full_param_, full_grad_, wide_grad_, narrow_grad_, flat_param_, reduced_;
  bool unsharded_ = false, ag_issued_ = false;
  std::shared_ptr<Work> ag_work_, rs_work_;
  bool grad_allocated_ = false, reduce_launched_ = false, unit_done_ = false;
  std::vector<bool> grad_written_, counted_, sunk_;
  size_t pending_ = 0;
  ShardedRuntime* runtime_ = nullptr;

  std::vector<std::shared_ptr<torch::autograd::Node>> accumulators_;
  std::vector<uintptr_t> hook_keys_;
  int64_t num_allgathers_ = 0, num_reductions_ = 0;
};

class ShardedRuntime {
 public:
  ShardedRuntime(std::vector<std::shared_ptr<ShardedUnit>> units, int forward_prefetch, int backward_prefetch)
      : units_(std::move(units)), forward_prefetch_(forward_prefetch), backward_prefetch_(backward_prefetch),
        alive_(std::make_shared<bool>(true)) {
    for (auto& u : units_) {
      TORCH_CHECK(u->runtime_ == nullptr, "ShardedRuntime: a unit already belongs to a runtime");
      u->runtime_ = this;
    }
  }
  ~ShardedRuntime() {
    *alive_ = false;
    for (auto& u : units_) u->runtime_ = nullptr;
  }

  void set_sync(bool s) { require_sync_ = s; }
  bool sync_enabled() const { return require_sync_; }
  bool in_backward() const { return in_pass_ || pre_backward_seen_; }

  // ---- schedule -------------------------------------------------------------------
  void pre_forward(int64_t index) {
    TORCH_CHECK(index >= 0 && index < (int64_t)units_.size(), "ShardedRuntime: bad unit index");
    std::lock_guard<std::recursive_mutex> lock(mu_);
    if (index == 0 && in_backward() && torch::autograd::get_current_graph_task_id() == -1) {
      // A new top-level forward (no autograd graph task is executing on this thread, so this is
      // not a recomputation) while a pass is still marked open: the previous backward raised
      // before its end-of-backward callback could run. Close it.
      for (auto& u : units_) u->abandon_pass();
      in_pass_ = false;
      pre_backward_seen_ = false;
      ++num_abandoned_passes_;
    }
    ShardedUnit& u = *units_[index];
    u.unshard();
    if (!in_backward()) {
      for (int k = 1; k <= forward_prefetch_; ++k)  // next units' all-gathers overlap this unit's math
        if (index + k < (int64_t)units_.size()) units_[index + k]->unshard();
    }
    u.wait_unshard();
  }

  void post_forward(int64_t index, const std::vector<at::Tensor>& outputs) {
    TORCH_CHECK(index >= 0 && index < (int64_t)units_.size(), "ShardedRuntime: bad unit index");
    if (torch::autograd::GradMode::is_enabled()) {
      // The first gradient reaching any output of the unit marks the start of its backward.
      auto fired = std::make_shared<bool>(false);
      auto alive = alive_;
      for (const at::Tensor& o : outputs) {
        if (!o.defined() || !o.requires_grad()) continue;
        o.register_hook([this, index, fired, alive](const at::Tensor& grad) -> at::Tensor {
          if (*alive && !*fired) {
            *fired = true;
            this->pre_backward(index);
          }
          return grad;
        });
      }
    }
    std::lock_guard<std::recursive_mutex> lock(mu_);
    ShardedUnit& u = *units_[index];
    // A recomputed forward inside backward must not reshard: the backward still needs the parameters.
    const bool resident = index < (int64_t)resident_.size() && resident_[index];
    if (u.strategy_ == kFullShard && u.reshard_after_forward && !resident && !in_backward()) u.reshard();
    // End of a forward that no backward will follow (evaluation under no_grad): nothing would ever
    // reshard the root, SHARD_GRAD_OP units and resident units, and an optimizer step or a loaded
    // checkpoint would then go unnoticed by the next forward. Release them now.
    if (index == 0 && !torch::autograd::GradMode::is_enabled() && !in_backward())
      for (auto& other : units_)
        if (other->strategy_ != kNoShard) other->reshard();
  }

  // Chooses the units that stay unsharded between forward and backward: the longest suffix of
  // FULL_SHARD units (the root excluded: it never reshards inside a pass) whose unsharded
  // parameters fit `bytes`. Must be called with the same value on every rank (the choice
  // changes which all-gathers are issued). Returns the number of resident units.
  int64_t set_resident_budget(int64_t bytes) {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    resident_.assign(units_.size(), false);
    resident_bytes_ = 0;
    for (auto& u : units_) u->window_resident_ = false;
    int64_t count = 0;
    for (int64_t i = (int64_t)units_.size() - 1; i >= 1; --i) {
      ShardedUnit& u = *units_[i];
      if (u.strategy_ != kFullShard || !u.reshard_after_forward) continue;
      const int64_t unit_bytes = u.padded_numel_ * (int64_t)u.full_param_.element_size();
      if (resident_bytes_ + unit_bytes > bytes) break;
      resident_[i] = true;
      u.window_resident_ = true;
      resident_bytes_ += unit_bytes;
      ++count;
    }
    return count;
  }
  int64_t resident_bytes() const { return resident_bytes_; }

  void pre_backward(int64_t index) {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    pre_backward_seen_ = true;
    ShardedUnit& u = *units_[index];
    u.unshard();
    // Backward visits the units as root (LM head), last block, ..., first block, root
    // (embeddings); the root never reshards in between, so the chain to prefetch is
    // root -> n-1 -> n-2 -> ... -> 1.
    int64_t j = index;
    const int64_t n = (int64_t)units_.size();
    for (int k = 0; k < backward_prefetch_; ++k) {
      j = j == 0 ? n - 1 : j - 1;
      if (j < 1) break;
      units_[j]->unshard();
    }
    u.wait_unshard();
  }

  // Called (under the mutex) by the first gradient hook of a backward pass.
  void begin_pass() {
    in_pass_ = true;
    for (auto& u : units_) u->reset_pass();
    // (through recompute::defer: inside a recomputed block the current graph task is the block's)
    pda::recompute::defer([this, alive = alive_] {
      if (!*alive) return;
      torch::autograd::Engine::get_default_engine().queue_callback([this, alive] {
        if (*alive) this->finish_pass();
      });
    });
  }

  void finish_pass() {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    if (!in_pass_) return;
    pda::trace::Scope range("gradient reduction wait (end of backward)", "comm");
    if (require_sync_) {
      // Units whose parameters did not all receive a gradient: reduce what there is, in
      // unit order (identical on every rank).
      for (auto& u : units_)
        if (!u->reduce_launched_ && u->grad_allocated_) u->launch_reduce();
      for (auto& u : units_) u->finish_reduce();
      for (auto& u : units_)
        if (u->strategy_ != kNoShard) u->reshard();  // parameters change at the optimizer step
    }
    in_pass_ = false;
    pre_backward_seen_ = false;
  }

  int64_t num_abandoned_passes() const { return num_abandoned_passes_; }

  std::recursive_mutex mu_;
  bool in_pass_ = false;

 private:
  std::vector<std::shared_ptr<ShardedUnit>> units_;
  int forward_prefetch_ = 1, backward_prefetch_ = 1;
  bool require_sync_ = true;
  bool pre_backward_seen_ = false;
  int64_t num_abandoned_passes_ = 0;
  std::vector<bool> resident_;  // units that keep their unsharded parameters from forward to backward
  int64_t resident_bytes_ = 0;
  std::shared_ptr<bool> alive_;
};

std::pair<at::Tensor, bool> ShardedUnit::grad_slot(size_t i) {
  TORCH_CHECK(runtime_ != nullptr, "ShardedUnit: grad_slot before the unit joined a runtime");
  TORCH_CHECK(i < params_.size(), "ShardedUnit: bad parameter index");
  std::lock_guard<std::recursive_mutex> lock(runtime_->mu_);
  at::NoGradGuard ng;
  if (!grad_allocated_) {
    allocate_storage(full_grad_, padded_numel_);  // unwritten ranges are zeroed by launch_reduce
    grad_allocated_ = true;
  }
  at::Tensor slot = full_grad_.narrow(0, offsets_[i], params_[i].numel()).view(params_[i].sizes());
  const bool accumulate = grad_written_[i];
  grad_written_[i] = true;
  sunk_[i] = true;
  return {slot, accumulate};
}

void ShardedUnit::on_grad_ready(size_t i) {
  TORCH_CHECK(runtime_ != nullptr, "ShardedUnit: gradient arrived before the unit joined a runtime");
  std::lock_guard<std::recursive_mutex> lock(runtime_->mu_);
  if (!runtime_->in_pass_) runtime_->begin_pass();
  at::Tensor& g = params_[i].mutable_grad();
  if (sunk_[i]) {
    sunk_[i] = false;  // the op already wrote this gradient into the flat buffer
    if (g.defined()) g.reset();
  } else if (g.defined()) {
    at::NoGradGuard ng;
    if (!grad_allocated_) {
      allocate_storage(full_grad_, padded_numel_);  // unwritten ranges are zeroed by launch_reduce
      grad_allocated_ = true;
    }
    at::Tensor slot = full_grad_.narrow(0, offsets_[i], params_[i].numel()).view(params_[i].sizes());
    if (grad_written_[i]) slot.add_(g);
    else slot.copy_(g);
    grad_written_[i] = true;
    g.reset();  // the flat buffer is now the only holder of this gradient
  }
  if (!counted_[i]) {
    counted_[i] = true;
    if (pending_ > 0) --pending_;
  }
  if (pending_ == 0 && !unit_done_) {
    unit_done_ = true;
    if (runtime_->sync_enabled() && !reduce_launched_) launch_reduce();
    // FULL_SHARD: every kernel of this unit's backward is already enqueued; release the
    // unsharded parameters now rather than at the end of backward (also under no_sync,
    // where only the gradients stay unsharded).
    // A unit inside the resident budget stays gathered through a no_sync window as well: no optimizer
    // step can change its parameters before the window's last backward, so the next micro-batch's
    // forward finds them in place instead of gathering them again.
    const bool stays = window_resident_ && !runtime_->sync_enabled();
    if (strategy_ == kFullShard && reshard_after_forward && !stays) reshard();
  }
}

}  // namespace

void bind_fsdp(py::module_& m) {
  py::class_<ShardedUnit, std::shared_ptr<ShardedUnit>>(m, "ShardedUnit")
      .def(py::init<std::vector<at::Tensor>, std::shared_ptr<Communicator>, int, bool, bool, bool>(), py::arg("params"),
           py::arg("communicator"), py::arg("strategy"), py::arg("force_collectives") = false,
           py::arg("reduce_fp32") = false, py::arg("accum_fp32") = false)
      .def("accumulates_fp32", &ShardedUnit::accumulates_fp32)
      .def("unshard", &ShardedUnit::unshard)
      .def("wait_unshard", &ShardedUnit::wait_unshard)
      .def("reshard", &ShardedUnit::reshard)
      .def("is_unsharded", &ShardedUnit::is_unsharded)
      .def("unshard_in_flight", &ShardedUnit::unshard_in_flight)
      .def("refresh_shard_from_full", &ShardedUnit::refresh_shard_from_full)
      .def("gather_shards", &ShardedUnit::gather_shards, py::arg("shard"), py::call_guard<py::gil_scoped_release>())
      .def("load_master_from_full", &ShardedUnit::load_master_from_full, py::arg("full"))
      .def("invalidate_unsharded", &ShardedUnit::invalidate_unsharded)
      .def("remove_hooks", &ShardedUnit::remove_hooks)
      .def("shard_master", &ShardedUnit::shard_master)
      .def("shard_lowp", &ShardedUnit::shard_lowp)
      .def("full_param", &ShardedUnit::full_param)
      .def("set_flat_param", &ShardedUnit::set_flat_param)
      .def("grad_slot", &ShardedUnit::grad_slot, py::arg("index"))
      .def("params", &ShardedUnit::params)
      .def("offsets", &ShardedUnit::offsets)
      .def_property_readonly("shard_numel", &ShardedUnit::shard_numel)
      .def_property_readonly("padded_numel", &ShardedUnit::padded_numel)
      .def_property_readonly("total_numel", &ShardedUnit::total_numel)
      .def_property_readonly("shard_index", &ShardedUnit::shard_index)
      .def_property_readonly("strategy", &ShardedUnit::strategy)
      .def_property_readonly("mixed", &ShardedUnit::mixed)
      .def_property_readonly("num_allgathers", &ShardedUnit::num_allgathers)
      .def_property_readonly("num_reductions", &ShardedUnit::num_reductions)
      .def_readwrite("reshard_after_forward", &ShardedUnit::reshard_after_forward);
  py::class_<ShardedRuntime, std::shared_ptr<ShardedRuntime>>(m, "ShardedRuntime")
      .def(py::init<std::vector<std::shared_ptr<ShardedUnit>>, int, int>(), py::arg("units"),
           py::arg("forward_prefetch") = 1, py::arg("backward_prefetch") = 1)
      .def("set_sync", &ShardedRuntime::set_sync)
      .def("sync_enabled", &ShardedRuntime::sync_enabled)
      .def("in_backward", &ShardedRuntime::in_backward)
      .def("pre_forward", &ShardedRuntime::pre_forward, py::arg("index"), py::call_guard<py::gil_scoped_release>())
      .def("post_forward", &ShardedRuntime::post_forward, py::arg("index"), py::arg("outputs"))
      .def("pre_backward", &ShardedRuntime::pre_backward, py::arg("index"), py::call_guard<py::gil_scoped_release>())
      .def("set_resident_budget", &ShardedRuntime::set_resident_budget, py::arg("bytes"))
      .def_property_readonly("resident_bytes", &ShardedRuntime::resident_bytes)
      .def_property_readonly("num_abandoned_passes", &ShardedRuntime::num_abandoned_passes);
}
