// Bookkeeping for block recomputation (utils/recompute.py).
//
// The recompute engine re-runs a block's forward inside the backward of an outer autograd
// function and then differentiates it with a nested engine call. PyTorch attaches
// end-of-backward callbacks (Engine::queue_callback) to the graph task that is current when
// they are queued; queued from inside the nested call they would fire when the *block's*
// backward ends, long before the model's. The gradient reducer and the sharding runtime
// therefore queue theirs through defer(): immediately when no recomputation is running, and
// otherwise right after the nested call has returned, when the current graph task is the
// outer one again.
#pragma once

#include <functional>

namespace pda {
namespace recompute {

int depth();
void enter();
void exit();  // runs the deferred closures once the depth is back to zero
void defer(std::function<void()> fn);

}  // namespace recompute
}  // namespace pda
