"""Fully sharded data parallelism on the native flat-parameter runtime.

API shape follows the reference's use of torch FSDP (assignments/assignment1/train_fsdp.py:64-81):
``FSDP(block, sharding_strategy=ShardingStrategy.X)`` around every ``transformer.h[i]`` and
then ``FSDP(model, sharding_strategy=...)`` as the root, each instance becoming one
communication unit; ``no_sync()``, ``parameters()`` (the local flat shards — so
``AdamW(model.parameters())`` keeps optimizer state sharded), ``state_dict()`` (full,
original names, collective) and ``load_state_dict()`` are provided.
``fsdp_wrap_gpt(model, strategy, unit_blocks=k)`` does the same wrapping with ``k`` blocks
per unit: on 288 GB GPUs bigger units mean fewer, larger collectives.

Strategies: FULL_SHARD (parameters, gradients, optimizer state sharded; all-gather before a
unit's forward and backward, reduce-scatter after its backward), SHARD_GRAD_OP (parameters
stay unsharded from forward to the end of backward), NO_SHARD (replicated; all-reduce).

The mechanics AND the schedule live in C++ (``csrc/runtime/fsdp_unit.cpp``). The module's
forward hooks make one native call each (``ShardedRuntime.pre_forward`` unshards the unit and
prefetches the next ones; ``post_forward`` reshards and plants C++ gradient hooks on the unit's
outputs), and the whole backward — re-gathering parameters in reverse order with prefetch,
gradient gathering, reduce-scatter launch, resharding, end-of-backward fence — runs from native
autograd hooks without entering Python. Collectives go through the native communicator
(``csrc/runtime/comm``): all-gathers on its gather stream, reductions on its reduce stream.
"""
from __future__ import annotations

import functools

import contextlib
import enum
from typing import Dict, Iterator, List, Optional, Tuple

import torch
from torch import nn

from .. import _ext
from . import comm


class ShardingStrategy(enum.Enum):
    FULL_SHARD = 0
    SHARD_GRAD_OP = 1
    NO_SHARD = 2


def _as_strategy(s) -> ShardingStrategy:
    if isinstance(s, ShardingStrategy):
        return s
    if isinstance(s, str):
        return ShardingStrategy[s]
    name = getattr(s, "name", None)  # e.g. torch.distributed.fsdp.ShardingStrategy
    if name in ShardingStrategy.__members__:
        return ShardingStrategy[name]
    raise ValueError(f"unknown sharding strategy {s!r}")


class FSDP(nn.Module):
    """One sharded communication unit around ``module`` (nestable; the outermost is the root)."""

    def __init__(self, module: nn.Module, sharding_strategy=ShardingStrategy.FULL_SHARD,
                 process_group=None, forward_prefetch: int = 1, backward_prefetch: int = 1,
                 force_collectives: bool = False, reduce_fp32: bool = False, communicator=None,
                 resident_gb=0.0, grad_accum_dtype: Optional[str] = None):
        """``reduce_fp32``: reduce bf16 gradients through an fp32 copy (twice the bytes on the wire,
        no rounding between the ranks' contributions); the default reduces in the gradient dtype.
        ``process_group``: a torch.distributed group to run on instead of the native communicator.
        ``grad_accum_dtype="fp32"``: the unit's unsharded gradient buffer is fp32 for bf16 parameters,
        so the micro-batches of a ``no_sync`` window are summed in fp32 (the reference accumulates in
        fp32: assignments/assignment1/train/trainer.py:57-61) and rounded once, for the collective
        (or not at all with ``reduce_fp32``); the default keeps the parameter dtype.
        ``resident_gb`` (read from the root instance; FULL_SHARD only): GiB of unsharded parameters
        that may stay allocated between forward and backward. The last units of the model — the
        first ones backward visits — keep their parameters while they fit and are then not
        all-gathered again in backward; 0 is textbook FULL_SHARD, a budget covering the model gives
        SHARD_GRAD_OP's traffic with FULL_SHARD's unit-by-unit release in backward. ``"auto"``
        takes a tenth of the device memory that is free at the first forward (the minimum over the
        ranks, so that every rank issues the same collectives)."""
        super().__init__()
        self.module = module
        self.sharding_strategy = _as_strategy(sharding_strategy)
        self.communicator = comm.resolve(communicator, process_group)
        self.world_size = self.communicator.world_size
        self.rank = self.communicator.rank
        self.forward_prefetch = forward_prefetch
        self.backward_prefetch = backward_prefetch
        self.resident_gb = resident_gb
        self.resident_units = 0
        first = next((p for p in module.parameters() if p.requires_grad), None)
        if first is not None and first.is_cuda != self.communicator.on_gpu:
            raise RuntimeError(f"FSDP: the '{self.communicator.backend}' communicator cannot move "
                               f"{'GPU' if first.is_cuda else 'CPU'} tensors")

        # Parameters already owned by nested FSDP instances stay with them.
        nested = [m for m in module.modules() if isinstance(m, FSDP)]
        owned = {id(p) for m in nested for p in m._unit_params}
        seen, mine, names = set(), [], []
        for name, p in module.named_parameters():
            if id(p) in owned or id(p) in seen or not p.requires_grad:
                continue
            seen.add(id(p))
            mine.append(p)
            names.append(name)
        if not mine:
            raise ValueError("FSDP: nothing to shard in this module (all parameters belong to nested units)")
        self._unit_params: List[nn.Parameter] = mine
        self._unit_param_names = names
        comm.assert_same_across_ranks([p.numel() for p in mine], "FSDP unit parameter sizes", self.communicator)

        ext = _ext.require()
        if grad_accum_dtype not in (None, "param", "fp32"):
            raise ValueError("grad_accum_dtype must be None / 'param' or 'fp32'")
        self.unit = ext.ShardedUnit(mine, self.communicator, self.sharding_strategy.value, force_collectives,
                                    reduce_fp32, grad_accum_dtype == "fp32")
        # Weight matrices hand ops.linear a slice of this unit's flat gradient buffer, so the
        # weight-gradient kernel writes there directly instead of producing a tensor that the
        # gradient hook then copies (one copy kernel per parameter and micro-batch).
        for i, p in enumerate(mine):
            p._pda_grad_sink = functools.partial(self.unit.grad_slot, i)
            if self.unit.accumulates_fp32():
                p._pda_grad_sink_dtype = torch.float32
        # The flat shard is deliberately NOT registered as a module parameter (plain attribute,
        # bypassing nn.Module.__setattr__): state_dict() keeps the original names, .to() never
        # touches it, and parameters() is overridden below to return it.
        flat = nn.Parameter(self.unit.shard_master(), requires_grad=True)
        lowp = self.unit.shard_lowp()
        if lowp is not None and lowp.numel() > 0:
            flat._pda_lowp = lowp            # bf16 mirror the optimizer kernel keeps in sync
        flat._pda_free_grad_after_step = True  # the reduced shard gradient is re-created every step
        self.unit.set_flat_param(flat)
        object.__setattr__(self, "flat_param", flat)

        self._is_root: Optional[bool] = None
        self._runtime = None
        self._order: List["FSDP"] = []
        self._index = 0
        object.__setattr__(self, "_root", None)  # plain attribute: a Module assigned normally becomes a child
        module.register_forward_pre_hook(self._pre_forward, with_kwargs=True)
        module.register_forward_hook(self._post_forward)

    # ------------------------------------------------------------------ root set-up
    def _lazy_init_root(self) -> None:
        """First forward of the outermost instance: fix the unit order and build the runtime."""
        if self._is_root is not None:
            return
        self._is_root = True
        units = [m for m in self.modules() if isinstance(m, FSDP)]  # pre-order: root first
        for i, u in enumerate(units):
            u._index = i
            object.__setattr__(u, "_root", self)
            if u is not self:
                u._is_root = False
        # The root's own parameters (embeddings, final norm, LM head) are needed at both ends
        # of forward and of backward: it stays unsharded from its first use to the end.
        self.unit.reshard_after_forward = False
        self._order = units
        self._runtime = _ext.require().ShardedRuntime([u.unit for u in units], self.forward_prefetch,
                                                      self.backward_prefetch)
        budget = self._resident_budget_bytes()
        if budget > 0:
            self.resident_units = self._runtime.set_resident_budget(budget)

    def _resident_budget_bytes(self) -> int:
        if self.sharding_strategy is not ShardingStrategy.FULL_SHARD:
            return 0
        if self.resident_gb != "auto":
            return int(float(self.resident_gb or 0.0) * 2**30)
        dev = self.flat_param.device
        free = torch.cuda.mem_get_info(dev)[0] if dev.type == "cuda" else 0
        budget = torch.tensor([-0.10 * free], dtype=torch.float64, device=dev)
        if self.world_size > 1:  # the smallest budget of any rank, agreed before the first collective of the pass
            comm.all_reduce(budget, "max", self.communicator)
        return int(-budget.item())

    # ------------------------------------------------------------------ hooks
    def _units(self) -> List["FSDP"]:
        return (self._root or self)._order

    def _pre_forward(self, module, args, kwargs):
        if self._is_root is None:
            self._lazy_init_root()
        (self._root or self)._runtime.pre_forward(self._index)
        return None

    def _post_forward(self, module, args, output):
        if torch.is_tensor(output):
            outs = [output]
        else:
            outs = []
            _collect_tensors(output, outs)
        (self._root or self)._runtime.post_forward(self._index, outs)
        return None

    # ------------------------------------------------------------------ module surface
    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    @property
    def device(self) -> torch.device:
        return self.flat_param.device

    def _all_units(self) -> List["FSDP"]:
        return [m for m in self.modules() if isinstance(m, FSDP)]

    def parameters(self, recurse: bool = True) -> Iterator[nn.Parameter]:
        """The local flat shards of this unit and of every nested unit."""
        for u in (self._all_units() if recurse else [self]):
            yield u.flat_param

    def named_parameters(self, prefix: str = "", recurse: bool = True, remove_duplicate: bool = True):
        for i, u in enumerate(self._all_units() if recurse else [self]):
            yield f"{prefix}{'.' if prefix else ''}_flat_shard_{i}", u.flat_param

    @contextlib.contextmanager
    def no_sync(self):
        """Backward passes inside accumulate UNSHARDED gradients and do not reduce-scatter."""
        if self._is_root is None:
            self._lazy_init_root()
        rt = (self._root or self)._runtime
        previous = rt.sync_enabled()
        rt.set_sync(False)
        try:
            yield
        finally:
            rt.set_sync(previous)

    @contextlib.contextmanager
    def summon_full_params(self, writeback: bool = False):
        """All units unsharded inside the context (collective: every rank must enter)."""
        units = self._all_units()
        for u in units:
            u.unit.unshard()
        for u in units:
            u.unit.wait_unshard()
        try:
            yield
        finally:
            for u in units:
                if writeback:
                    u.unit.refresh_shard_from_full()
                u.unit.reshard()

    def state_dict(self, *args, rank0_only: bool = False, cpu_offload: bool = True, **kwargs):
        """FULL state dict under the original parameter names, values = the fp32 master weights
        (gathered from the shards; the bf16 compute copy is never the source, so a checkpoint
        round trip loses nothing). Collective: every rank must call it (a rank-0-only call
        deadlocks, which is what the reference's ``if rank == 0: save_checkpoint`` would do
        under sharding)."""
        if kwargs.get("destination") is not None or (args and args[0] is not None):
            # Called as a child while an enclosing module collects its state dict: behave like a
            # plain container (the outermost call owns the gathering and the key clean-up).
            return nn.Module.state_dict(self, *args, **kwargs)
        from .checkpoint import full_model_state_dict

        return full_model_state_dict(self, cpu_offload=cpu_offload, rank0_only=rank0_only)

    def load_state_dict(self, state_dict, strict: bool = True, **kwargs):
        """Loads a FULL state dict (every rank passes the same one): each unit's fp32 master
        shard is cut straight out of it."""
        from .checkpoint import load_full_model_state_dict

        missing, unexpected = load_full_model_state_dict(self, state_dict, strict=strict)
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def full_optim_state_dict(self, optimizer, cpu_offload: bool = True):
        """Optimizer state in torch.optim.AdamW layout over the bare model's parameters (collective)."""
        from .checkpoint import full_optim_state_dict

        return full_optim_state_dict(self, optimizer, cpu_offload)

    def load_full_optim_state_dict(self, optimizer, state) -> None:
        from .checkpoint import load_full_optim_state_dict

        load_full_optim_state_dict(self, optimizer, state)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(super().__getattr__("module"), name)


def _collect_tensors(o, out: list) -> None:
    if torch.is_tensor(o):
        out.append(o)
    elif isinstance(o, (list, tuple)):
        for x in o:
            _collect_tensors(x, out)
    elif isinstance(o, dict):
        for x in o.values():
            _collect_tensors(x, out)


def fsdp_wrap_gpt(model: nn.Module, sharding_strategy=ShardingStrategy.FULL_SHARD, unit_blocks: int = 1,
                  **kwargs) -> FSDP:
    """Per-block wrapping of a GPT-style model (``model.transformer.h``) plus the root unit —
    the layout of the reference's ``setup_fsdp_model`` — with ``unit_blocks`` consecutive
    blocks per communication unit."""
    blocks = model.transformer.h
    strategy = _as_strategy(sharding_strategy)
    if unit_blocks <= 1:
        for i in range(len(blocks)):
            blocks[i] = FSDP(blocks[i], strategy, **kwargs)
    else:
        groups = [list(range(s, min(s + unit_blocks, len(blocks)))) for s in range(0, len(blocks), unit_blocks)]
        new_blocks = []
        for idxs in groups:
            new_blocks.append(FSDP(_BlockGroup([blocks[i] for i in idxs]), strategy, **kwargs))
        model.transformer.h = _GroupedBlockList(new_blocks, [len(g) for g in groups])
    return FSDP(model, strategy, **kwargs)


class _BlockGroup(nn.Module):
    """Several consecutive transformer blocks executed as one unit (keeps their state_dict
    names ``<i>.…`` relative to the group; ``_GroupedBlockList`` restores the global index)."""

    def __init__(self, blocks):
        super().__init__()
        self.blocks = nn.ModuleList(blocks)

    def forward(self, hidden, residual=None, *, fused: bool = False, delta_pdrop: float = 0.0, regen: bool = False):
        if not fused:
            for b in self.blocks:
                hidden = b(hidden)
            return hidden
        for b in self.blocks:
            if hasattr(b, "out_pdrop"):  # blocks whose pending output dropout the next LayerNorm applies
                hidden, residual = b(hidden, residual, fused=True, delta_pdrop=delta_pdrop, regen=regen)
                delta_pdrop = b.out_pdrop
            else:
                hidden, residual = b(hidden, residual, fused=True, regen=regen)
        return hidden, residual

    @property
    def out_pdrop(self) -> float:
        return getattr(self.blocks[-1], "out_pdrop", 0.0)


class _GroupedBlockList(nn.ModuleList):
    """ModuleList of FSDP(_BlockGroup) whose state_dict uses the flat ``h.<global index>.``
    names of the ungrouped model."""

    def __init__(self, groups, sizes):
        super().__init__(groups)
        self._sizes = list(sizes)

    def _name_map(self) -> Dict[str, str]:
        m, g = {}, 0
        for gi, n in enumerate(self._sizes):
            for j in range(n):
                m[f"{gi}.module.blocks.{j}."] = f"{g}."
                m[f"{gi}.blocks.{j}."] = f"{g}."
                g += 1
        return m

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        raw = super().state_dict(*args, destination=None, prefix="", keep_vars=keep_vars)
        out = destination if destination is not None else {}
        nm = self._name_map()
        for k, v in raw.items():
            for old, new in nm.items():
                if k.startswith(old):
                    k = new + k[len(old):]
                    break
            out[prefix + k] = v
        return out

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        inv = {}
        g = 0
        for gi, n in enumerate(self._sizes):
            for j in range(n):
                inv[f"{prefix}{g}."] = f"{prefix}{gi}.module.blocks.{j}."
                g += 1
        for k in list(state_dict.keys()):
            for old, new in inv.items():
                if k.startswith(old):
                    state_dict[new + k[len(old):]] = state_dict.pop(k)
                    break
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
