"""Checkpoints of sharded models: full (reference-compatible) and sharded (per rank).

Full format — exactly the reference trainer's dictionary
(assignments/assignment1/train/trainer.py:117-128): ``model_state_dict`` under the bare model's
parameter names, ``optimizer_state_dict`` in ``torch.optim.AdamW`` layout indexed by the bare
model's ``parameters()`` order, ``step`` and ``lr_scheduler_state_dict``. Under sharding both
dictionaries are gathered collectively (``FSDP.state_dict`` and ``full_optim_state_dict``
below) and only rank 0 writes, so the file loads into the reference's ``Trainer`` and a
reference checkpoint loads into a sharded model at any world size. The gathered model values
are the fp32 MASTER weights (not the bf16 compute copies), so save -> resume of a bf16 run
continues bit-exactly.

Sharded format — one file per rank, ``rank{r}_of_{W}.pt``, holding this rank's fp32 flat
shards and optimizer moments per communication unit plus the layout needed to interpret
them (parameter names, shapes, offsets, shard size), and a ``meta.json`` written by rank 0.
Saving is local (no collective, no full copy on any rank): the form for the 7B-class models.
Loading at the saved world size is local too; at a DIFFERENT world size every rank re-slices
its new shard ranges out of the saved files that overlap them (``load_sharded_checkpoint``
reads only those). ``consolidate_sharded_checkpoint`` turns a sharded checkpoint directory
into the full format offline.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional

import torch

from . import comm

STATE_KEYS = ("exp_avg", "exp_avg_sq")


def _settle_optimizer(optimizer, replacing_state: bool = False) -> None:
    """FusedAdamW keeps its step counts as one integer per group between steps and caches views of its
    state: bring the per-parameter ``state[p]["step"]`` tensors up to date before they are read, and
    drop the cached views before state entries are replaced."""
    if optimizer is None:
        return
    if replacing_state and hasattr(optimizer, "_drop_fast"):
        optimizer._drop_fast()
    elif hasattr(optimizer, "_catch_up_host_steps"):
        optimizer._catch_up_host_steps()


def _units(model) -> list:
    from .fsdp import FSDP

    return [m for m in model.modules() if isinstance(m, FSDP)]


def _param_names(model) -> Dict[int, str]:
    """id(parameter) -> clean fully-qualified name (first name wins for tied parameters)."""
    names: Dict[int, str] = {}
    sd = model.module.state_dict(keep_vars=True)
    for k, v in sd.items():
        if isinstance(v, torch.nn.Parameter) and id(v) not in names:
            names[id(v)] = k.replace("module.", "")
    return names


def _ordered_params(model) -> List[torch.nn.Parameter]:
    """The bare model's ``parameters()`` order (tied parameters once)."""
    seen, out = set(), []
    for v in model.module.state_dict(keep_vars=True).values():
        if isinstance(v, torch.nn.Parameter) and id(v) not in seen:
            seen.add(id(v))
            out.append(v)
    return out


def unit_layouts(model) -> List[dict]:
    names = _param_names(model)
    layouts = []
    for u in _units(model):
        params = u.unit.params()
        layouts.append({
            "names": [names[id(p)] for p in u._unit_params],
            "shapes": [list(p.shape) for p in params],
            "offsets": list(u.unit.offsets()),
            "numels": [p.numel() for p in params],
            "shard_numel": u.unit.shard_numel,
            "padded_numel": u.unit.padded_numel,
            "strategy": u.sharding_strategy.name,
        })
    return layouts


# ------------------------------------------------------------------ full model state
def full_model_state_dict(model, cpu_offload: bool = True, rank0_only: bool = False) -> dict:
    """The bare model's ``state_dict()`` (same keys, same order, tied aliases included) with every
    parameter taken from the gathered fp32 master shards. Collective: every rank must call it."""
    owner = {}
    for u in _units(model):
        for i, p in enumerate(u._unit_params):
            owner[id(p)] = (u, i)
    gathered = {}
    out = {}
    made = {}  # tied parameters appear under several keys and share one tensor, as in the bare model
    keep = not (rank0_only and model.rank != 0)
    for k, v in model.module.state_dict(keep_vars=True).items():
        k = k.replace("module.", "")
        if id(v) in owner:
            u, i = owner[id(v)]
            if id(u) not in gathered:  # one collective per unit, in unit order on every rank
                gathered[id(u)] = u.unit.gather_shards(u.unit.shard_master())
            if not keep:
                continue
            if id(v) not in made:
                off = u.unit.offsets()[i]
                t = gathered[id(u)][off: off + v.numel()].view(v.shape)
                made[id(v)] = t.cpu() if cpu_offload else t.clone()
            out[k] = made[id(v)]
        elif keep:
            t = v.detach()
            out[k] = t.cpu() if cpu_offload else t.clone()
    return out


def load_full_model_state_dict(model, state_dict: dict, strict: bool = True):
    """Inverse: every rank passes the same full state dict; each unit's fp32 master shard (and
    its low-precision mirror) is cut straight out of it — nothing passes through bf16."""
    state_dict = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state_dict.items()}
    names = _param_names(model)
    used = set()
    missing = []
    with torch.no_grad():
        for u in _units(model):
            flat = torch.zeros(u.unit.padded_numel, dtype=torch.float32, device=u.flat_param.device)
            current = None
            for p, off in zip(u._unit_params, u.unit.offsets()):
                name = names[id(p)]
                if name in state_dict:
                    src = state_dict[name]
                    if tuple(src.shape) != tuple(p.shape):
                        raise RuntimeError(f"size mismatch for {name}: checkpoint {tuple(src.shape)}, "
                                           f"model {tuple(p.shape)}")
                    flat[off: off + p.numel()].copy_(src.reshape(-1))
                    used.add(name)
                else:
                    missing.append(name)
                    if current is None:  # keep the present values of parameters the file lacks
                        current = u.unit.gather_shards(u.unit.shard_master())
                    flat[off: off + p.numel()].copy_(current[off: off + p.numel()])
            u.unit.load_master_from_full(flat)
        # tied aliases and buffers
        own = model.module.state_dict(keep_vars=True)
        param_ids = {id(p) for u in _units(model) for p in u._unit_params}
        unexpected = []
        for k, v in state_dict.items():
            if k in used:
                continue
            kk = next((o for o in own if o.replace("module.", "") == k), None)
            if kk is None:
                unexpected.append(k)
            elif id(own[kk]) not in param_ids:
                own[kk].copy_(v)
    if strict and (missing or unexpected):
        raise RuntimeError(f"load_state_dict: missing keys {missing}, unexpected keys {unexpected}")
    return missing, unexpected


# ------------------------------------------------------------------ full optimizer state
def _gather_flat(shard: torch.Tensor, unit) -> torch.Tensor:
    """All ranks' shards of one flat tensor, concatenated (collective unless replicated)."""
    return unit.unit.gather_shards(shard.reshape(-1))


def full_optim_state_dict(model, optimizer, cpu_offload: bool = True) -> dict:
    """Optimizer state of a sharded model in ``torch.optim.AdamW`` layout over the bare model's
    parameters. Collective: every rank must call it."""
    _settle_optimizer(optimizer)
    order = {id(p): i for i, p in enumerate(_ordered_params(model))}
    state = {}
    for u in _units(model):
        st = optimizer.state.get(u.flat_param, {})
        if "exp_avg" not in st:
            continue
        fulls = {k: _gather_flat(st[k], u) for k in STATE_KEYS}
        step = st["step"]
        for p, off in zip(u._unit_params, u.unit.offsets()):
            entry = {"step": torch.as_tensor(float(step), dtype=torch.float32)}
            for k in STATE_KEYS:
                t = fulls[k][off: off + p.numel()].view(p.shape).clone()
                entry[k] = t.cpu() if cpu_offload else t
            state[order[id(p)]] = entry
    groups = []
    for g in optimizer.param_groups:
        gg = {k: v for k, v in g.items() if k != "params"}
        gg["params"] = sorted(order.values())
        groups.append(gg)
    return {"state": dict(sorted(state.items())), "param_groups": groups[:1]}


def load_full_optim_state_dict(model, optimizer, full_state: dict) -> None:
    """Inverse of :func:`full_optim_state_dict`: slices a full AdamW state into the local shards."""
    _settle_optimizer(optimizer, replacing_state=True)
    order = {id(p): i for i, p in enumerate(_ordered_params(model))}
    saved = full_state["state"]
    for u in _units(model):
        flat = u.flat_param
        padded, shard_n, idx = u.unit.padded_numel, u.unit.shard_numel, u.unit.shard_index
        bufs = {k: torch.zeros(padded, dtype=torch.float32, device=flat.device) for k in STATE_KEYS}
        step, found = 0.0, False
        for p, off in zip(u._unit_params, u.unit.offsets()):
            entry = saved.get(order[id(p)], saved.get(str(order[id(p)])))
            if entry is None:
                continue
            found = True
            step = float(entry["step"])
            for k in STATE_KEYS:
                bufs[k][off: off + p.numel()].copy_(entry[k].reshape(-1).to(torch.float32))
        if not found:
            continue
        st = optimizer.state[flat]
        st["step"] = torch.tensor(step, dtype=torch.float32)
        for k in STATE_KEYS:
            st[k] = bufs[k][idx * shard_n: (idx + 1) * shard_n].clone().view(flat.shape)
    for g, sg in zip(optimizer.param_groups, full_state["param_groups"]):
        for k, v in sg.items():
            if k != "params":
                g[k] = v


# ------------------------------------------------------------------ sharded checkpoints
def save_sharded_checkpoint(directory: str, model, optimizer=None, step: int = 0, lr_scheduler=None) -> str:
    """Writes this rank's shards to ``directory/rank{r}_of_{W}.pt`` (no collective)."""
    os.makedirs(directory, exist_ok=True)
    _settle_optimizer(optimizer)
    rank, world = comm.get_rank(), comm.get_world_size()
    layouts = unit_layouts(model)
    units = []
    for u in _units(model):
        entry = {"shard": u.flat_param.detach().cpu(), "shard_index": u.unit.shard_index}
        if optimizer is not None:
            st = optimizer.state.get(u.flat_param, {})
            if "exp_avg" in st:
                entry["optim"] = {"step": float(st["step"]), **{k: st[k].detach().cpu() for k in STATE_KEYS}}
        units.append(entry)
    payload = {"format": "pda-sharded-v1", "rank": rank, "world_size": world, "step": step,
               "layouts": layouts, "units": units}
    if optimizer is not None:
        payload["param_groups"] = [{k: v for k, v in g.items() if k != "params"} for g in optimizer.param_groups]
    if lr_scheduler is not None:
        payload["lr_scheduler_state_dict"] = lr_scheduler.state_dict()
    path = os.path.join(directory, f"rank{rank}_of_{world}.pt")
    tmp = f"{path}.tmp{os.getpid()}"
    torch.save(payload, tmp)
    os.replace(tmp, path)
    if rank == 0:
        with open(os.path.join(directory, "meta.json"), "w") as f:
            json.dump({"format": "pda-sharded-v1", "world_size": world, "step": step, "layouts": layouts}, f)
    comm.barrier()
    return path


def _load_rank_file(directory: str, r: int, world: int) -> dict:
    path = os.path.join(directory, f"rank{r}_of_{world}.pt")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} not found (incomplete sharded checkpoint?)")
    return torch.load(path, map_location="cpu", weights_only=True, mmap=True)


def _restore_unit(u, shard: torch.Tensor, optim: Optional[dict], optimizer) -> None:
    with torch.no_grad():
        u.flat_param.copy_(shard.view(u.flat_param.shape))
        lowp = getattr(u.flat_param, "_pda_lowp", None)
        if lowp is not None:
            if lowp.numel() == u.flat_param.numel():
                lowp.copy_(u.flat_param)
            else:  # NO_SHARD: the mirror is the whole live buffer
                lowp.view(-1)[: u.flat_param.numel()].copy_(u.flat_param)
    u.unit.invalidate_unsharded()  # a gathered copy from an earlier forward holds the old values
    if optimizer is not None and optim is not None:
        st = optimizer.state[u.flat_param]
        st["step"] = torch.tensor(optim["step"], dtype=torch.float32)
        for k in STATE_KEYS:
            st[k] = optim[k].to(u.flat_param.device).view(u.flat_param.shape).clone()


def load_sharded_checkpoint(directory: str, model, optimizer=None, lr_scheduler=None) -> int:
    """Loads a checkpoint written by :func:`save_sharded_checkpoint`. At the saved world size this
    is one local file read per rank; at any other world size (or sharding strategy) every rank
    cuts its new shard ranges out of the saved rank files that overlap them. The unit layout
    (which parameters form a unit, in which order) must be the one that was saved. Returns the
    saved step."""
    _settle_optimizer(optimizer, replacing_state=True)
    rank, world = comm.get_rank(), comm.get_world_size()
    with open(os.path.join(directory, "meta.json")) as f:
        meta = json.load(f)
    saved_world = int(meta["world_size"])
    units = _units(model)
    layouts = meta["layouts"]
    if len(units) != len(layouts):
        raise ValueError("sharded checkpoint was saved with a different unit layout "
                         f"({len(layouts)} units, the model has {len(units)})")
    mine = unit_layouts(model)
    for now, then in zip(mine, layouts):
        if now["names"] != then["names"] or now["offsets"] != then["offsets"]:
            raise ValueError("sharded checkpoint was saved with a different unit layout; "
                             "consolidate_sharded_checkpoint() converts it to the full format")
    same = saved_world == world and all(a["shard_numel"] == b["shard_numel"] and a["padded_numel"] == b["padded_numel"]
                                        for a, b in zip(mine, layouts))
    cache: Dict[int, dict] = {}

    def rank_file(r: int) -> dict:
        if r not in cache:
            cache[r] = _load_rank_file(directory, r, saved_world)
        return cache[r]

    if same:
        payload = rank_file(rank)
        for u, saved in zip(units, payload["units"]):
            _restore_unit(u, saved["shard"], saved.get("optim") if optimizer is not None else None, optimizer)
    else:
        for ui, (u, then) in enumerate(zip(units, layouts)):
            n_new, idx = u.unit.shard_numel, u.unit.shard_index
            lo, hi = idx * n_new, (idx + 1) * n_new           # my range of the unit's flat layout
            n_old = then["shard_numel"]
            replicated_then = then["padded_numel"] == then["shard_numel"]
            bufs = {"shard": torch.zeros(n_new, dtype=torch.float32)}
            have_opt, step = optimizer is not None, 0.0
            first = 0 if replicated_then else lo // n_old
            last = 0 if replicated_then else min((hi - 1) // n_old, saved_world - 1)
            for r in range(first, last + 1):
                saved = rank_file(r)["units"][ui]
                o_lo = 0 if replicated_then else r * n_old
                a, b = max(lo, o_lo), min(hi, o_lo + n_old)
                if b <= a:
                    continue
                bufs["shard"][a - lo: b - lo].copy_(saved["shard"].view(-1)[a - o_lo: b - o_lo])
                if have_opt and "optim" in saved:
                    step = saved["optim"]["step"]
                    for k in STATE_KEYS:
                        bufs.setdefault(k, torch.zeros(n_new, dtype=torch.float32))
                        bufs[k][a - lo: b - lo].copy_(saved["optim"][k].view(-1)[a - o_lo: b - o_lo])
                else:
                    have_opt = False
            optim = {"step": step, **{k: bufs[k] for k in STATE_KEYS}} if have_opt and "exp_avg" in bufs else None
            _restore_unit(u, bufs["shard"], optim, optimizer)
    payload = rank_file(rank if same else 0)
    if optimizer is not None and "param_groups" in payload:
        for g, sg in zip(optimizer.param_groups, payload["param_groups"]):
            g.update(sg)
    if lr_scheduler is not None and "lr_scheduler_state_dict" in payload:
        lr_scheduler.load_state_dict(payload["lr_scheduler_state_dict"])
    return int(payload["step"])


def consolidate_sharded_checkpoint(directory: str, dtype: Optional[torch.dtype] = None) -> dict:
    """Reads every rank file of a sharded checkpoint and returns the FULL checkpoint dictionary
    (reference format). Runs in any single process; no process group needed."""
    with open(os.path.join(directory, "meta.json")) as f:
        meta = json.load(f)
    world = meta["world_size"]
    payloads = [torch.load(os.path.join(directory, f"rank{r}_of_{world}.pt"), map_location="cpu",
                           weights_only=True) for r in range(world)]
    layouts = payloads[0]["layouts"]
    model_sd: Dict[str, torch.Tensor] = {}
    optim_by_name: Dict[str, dict] = {}
    for ui, layout in enumerate(layouts):
        replicated = layout["padded_numel"] == layout["shard_numel"]
        srcs = payloads[:1] if replicated else payloads
        full = torch.cat([p["units"][ui]["shard"].view(-1) for p in srcs])
        has_opt = all("optim" in p["units"][ui] for p in srcs)
        opt_full = {k: torch.cat([p["units"][ui]["optim"][k].view(-1) for p in srcs]) for k in STATE_KEYS} \
            if has_opt else None
        for name, shape, off, n in zip(layout["names"], layout["shapes"], layout["offsets"], layout["numels"]):
            t = full[off: off + n].view(shape).clone()
            model_sd[name] = t.to(dtype) if dtype is not None else t
            if opt_full is not None:
                optim_by_name[name] = {"step": torch.tensor(payloads[0]["units"][ui]["optim"]["step"]),
                                       **{k: opt_full[k][off: off + n].view(shape).clone() for k in STATE_KEYS}}
    out = {"model_state_dict": model_sd, "step": payloads[0]["step"], "param_names": list(model_sd.keys())}
    if optim_by_name:
        names = list(model_sd.keys())
        out["optimizer_state_dict"] = {
            "state": {i: optim_by_name[n] for i, n in enumerate(names) if n in optim_by_name},
            "param_groups": [{**payloads[0].get("param_groups", [{}])[0], "params": list(range(len(names)))}],
        }
    if "lr_scheduler_state_dict" in payloads[0]:
        out["lr_scheduler_state_dict"] = payloads[0]["lr_scheduler_state_dict"]
    return out


def order_like_model(consolidated: dict, model) -> dict:
    """Re-indexes a consolidated optimizer state by ``model.parameters()`` order and adds the
    alias keys of tied parameters (``lm_head.weight``) to the model state dict."""
    names = consolidated["param_names"]
    sd = dict(consolidated["model_state_dict"])
    target = model.state_dict()
    by_ptr: Dict[int, str] = {}
    for k, v in model.state_dict(keep_vars=True).items():
        if v.data_ptr() in by_ptr and k not in sd and by_ptr[v.data_ptr()] in sd:
            sd[k] = sd[by_ptr[v.data_ptr()]]
        by_ptr.setdefault(v.data_ptr(), k)
    out = dict(consolidated)
    out["model_state_dict"] = {k: sd[k] for k in target if k in sd}
    if "optimizer_state_dict" in consolidated:
        name_to_idx = {}
        seen = set()
        for k, v in model.state_dict(keep_vars=True).items():
            if isinstance(v, torch.nn.Parameter) and id(v) not in seen:
                seen.add(id(v))
                name_to_idx[k] = len(name_to_idx)
        old = consolidated["optimizer_state_dict"]
        new_state = {name_to_idx[names[i]]: st for i, st in old["state"].items() if names[i] in name_to_idx}
        out["optimizer_state_dict"] = {"state": dict(sorted(new_state.items())),
                                       "param_groups": [{**old["param_groups"][0],
                                                         "params": list(range(len(name_to_idx)))}]}
    return out
