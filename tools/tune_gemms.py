#!/usr/bin/env python3
"""Exhaustive GEMM kernel search for the shapes of given models, written to a plan file.

For every (model, micro-batch) pair one forward + backward of the bare model runs with the planner
in exhaustive mode (all hipBLASLt kernels that support each problem are timed; input-gradient shapes
additionally choose between the direct and the transposed-weight route), then the decisions are
merged into the plan file that ``pytorch_distributed_amd.tuning.gemm_plans`` loads at start-up.

  python tools/tune_gemms.py --spec gpt2-large:32 gpt2-large:8 gpt2-medium:32 --out gpurun_out/plans.json
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spec", nargs="+", default=["gpt2-large:32"], help="model:micro_batch pairs")
    ap.add_argument("--seq-len", type=int, default=1024)
    ap.add_argument("--out", default=None, help="plan file to write (default: the package's own)")
    ap.add_argument("--layers", type=int, default=2, help="blocks to instantiate (shapes do not depend on depth)")
    ap.add_argument("--heuristic-only", action="store_true", help="time only the heuristic's candidates")
    args = ap.parse_args()

    import torch

    from pytorch_distributed_amd import _ext
    from pytorch_distributed_amd.models import LlamaForCausalLM, MyGPT2LMHeadModel, get_config
    from pytorch_distributed_amd.parallel import plan_sync
    from pytorch_distributed_amd.tuning import gemm_plans

    os.environ["PDA_NO_GEMM_PLAN_FILE"] = "1"  # search from scratch
    ext = _ext.require()
    ext.gemm_planner_configure(True, 48, 3, not args.heuristic_only)
    dev = torch.device("cuda", 0)
    out = args.out or gemm_plans.PLAN_FILE
    for spec in args.spec:
        name, _, mb = spec.partition(":")
        mb = int(mb or 32)
        cfg = get_config(name).without_dropout()
        cfg.n_layer = min(cfg.n_layer, args.layers)
        cls = LlamaForCausalLM if hasattr(cfg, "intermediate_size") else MyGPT2LMHeadModel
        torch.manual_seed(0)
        model = cls(cfg).to(device=dev, dtype=torch.bfloat16)
        t0 = time.time()
        n = plan_sync.plan_gemms(model, mb, args.seq_len, cfg.vocab_size)
        print(f"[tune] {spec}: {n} problems planned so far, {time.time() - t0:.1f} s", flush=True)
        del model
        gemm_plans.save(out)  # after every model: a later failure keeps what was found
    st = ext.gemm_planner_stats()
    for row in sorted(st["table"]):
        m, n, k, ta, tb, ep, cands, chosen, first, best, sol = row
        gain = 100.0 * (first / best - 1.0) if best > 0 and first > 0 else 0.0
        print(f"[tune] m={m} n={n} k={k} ta={int(ta)} tb={int(tb)} ep={ep}: {cands} candidates, heuristic first "
              f"{first:.1f} us, best {best:.1f} us ({gain:+.1f} %), solution {sol}, "
              f"{2.0 * m * n * k / max(best, 1e-3) / 1e6:.0f} TFLOP/s")
    print(f"[tune] wrote {out} ({ext.gemm_planner_library()})")


if __name__ == "__main__":
    main()
