"""Checkpoint / resume in the reference's file format (src/train.py:145-159, 225-248; src/utils/utility.py:22-50), so that
result folders are interchangeable: ``model_ckpt.h5`` = {start_epoch_nb, model_state_dict, optim_state_dict,
confidence_thresh, rng_state, best_log, train_remaining_file}, ``model_best.h5`` = {epoch_nb, model_state_dict,
optim_state_dict, confidence_thresh}.  ``model_state_dict`` has the reference's 305 keys (WrapperModel.state_dict()),
``optim_state_dict`` is torch.optim.Adam's layout with parameter indices in ``model.parameters()`` order.

Unlike the reference the RNG helpers do not touch ``torch.cuda`` when the device is the CPU.
"""
import os
import random

import numpy as np
import torch


def get_rng_state(device, model=None):
    """utility.py:32-39 plus, when ``model`` is given, the encoders' counter-based dropout streams (``rng.DropoutStream``:
    they replace the draws the reference takes from torch's generator)."""
    dev = torch.device(device)
    state = {"rand_state": random.getstate(), "numpy_state": np.random.get_state(),
             "torch_state": torch.random.get_rng_state(), "os_hash_state": str(os.environ.get("PYTHONHASHSEED", "100"))}
    if model is not None:
        from . import rng
        state["dropout_streams"] = rng.collect(model)
    state["cuda_state"] = torch.cuda.get_rng_state(device=dev) if dev.type == "cuda" and torch.cuda.is_available() else None
    return state


def seed_resume(rng_state, device, model=None):
    dev = torch.device(device)
    if model is not None and rng_state.get("dropout_streams"):
        from . import rng
        rng.restore(model, rng_state["dropout_streams"])
    random.setstate(rng_state["rand_state"])
    np.random.set_state(rng_state["numpy_state"])
    torch.random.set_rng_state(rng_state["torch_state"])
    if rng_state.get("cuda_state") is not None and dev.type == "cuda" and torch.cuda.is_available():
        torch.cuda.set_rng_state(rng_state["cuda_state"], device=dev)
    os.environ["PYTHONHASHSEED"] = str(rng_state["os_hash_state"])


def optimizer_state_dict(optimizer, model):
    """FusedAdam state in torch.optim.Adam's layout, indices following ``model.parameters()`` (the order the reference's
    ``optimizer.state_dict()`` uses), independent of the flat buffer's internal (reversed) order."""
    flat = optimizer.flat
    where = {id(p): k for k, p in enumerate(flat.params)}
    params = [p for p in model.parameters() if p.requires_grad] if model is not None else flat.module_params
    state = {}
    for i, p in enumerate(params):
        off, n = flat.offsets[where[id(p)]]
        if optimizer.step_count > 0:
            state[i] = {"step": torch.tensor(float(optimizer.step_count)),
                        "exp_avg": optimizer.exp_avg[off:off + n].view(p.shape).detach().cpu().clone(),
                        "exp_avg_sq": optimizer.exp_avg_sq[off:off + n].view(p.shape).detach().cpu().clone()}
    group = {"lr": optimizer.lr, "betas": tuple(optimizer.betas), "eps": optimizer.eps,
             "weight_decay": optimizer.weight_decay, "amsgrad": False, "maximize": False, "foreach": None,
             "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(params)))}
    return {"state": state, "param_groups": [group]}


def load_optimizer_state_dict(optimizer, model, sd):
    flat = optimizer.flat
    where = {id(p): k for k, p in enumerate(flat.params)}
    params = [p for p in model.parameters() if p.requires_grad] if model is not None else flat.module_params
    g = sd["param_groups"][0]
    if len(g["params"]) != len(params):
        raise ValueError("optimizer state has %d parameters, the model %d" % (len(g["params"]), len(params)))
    optimizer.lr, optimizer.betas, optimizer.eps = g["lr"], tuple(g["betas"]), g["eps"]
    optimizer.weight_decay = g.get("weight_decay", 0.0)
    if g.get("amsgrad", False):
        raise NotImplementedError("amsgrad Adam state is not supported by the fused gfx950 Adam")
    steps = set()
    for i, p in enumerate(params):
        st = sd["state"].get(g["params"][i], sd["state"].get(i))
        off, n = flat.offsets[where[id(p)]]
        if st is None:
            optimizer.exp_avg[off:off + n].zero_()
            optimizer.exp_avg_sq[off:off + n].zero_()
            continue
        if tuple(st["exp_avg"].shape) != tuple(p.shape):
            raise ValueError("optimizer state %d has shape %s, parameter %s" % (i, tuple(st["exp_avg"].shape), tuple(p.shape)))
        optimizer.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
        optimizer.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
        steps.add(int(st["step"]))
    if len(steps) > 1:
        raise ValueError("per-parameter step counts differ (%s): not a plain Adam run" % sorted(steps))
    optimizer.step_count = steps.pop() if steps else 0


def _state_dict_for_save(model, group=None):
    """-> (this process writes?, the state_dict to write).

    Data parallelism: BatchNorm running statistics are per-rank (each rank sees its own shard).  The file holds their AVERAGE
    over the ranks, formed in temporary copies: the live buffers are not touched (round 4, ADVICE: averaging them in place
    made a run that saves follow another trajectory than one that does not).  THE ALL-RANKS CONTRACT: with a process group of
    more than one rank, ``save_checkpoint`` / ``save_best`` are COLLECTIVES -- every rank must call them (an
    ``if rank == 0: save_checkpoint(...)`` would leave rank 0 waiting in all_reduce for ever); only rank 0 writes the file.
    ``group``: the trainer's process group (``TrainStep.reducer.group``; None = the default group)."""
    import torch.distributed as dist
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    writes = True
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        world = dist.get_world_size(group)
        for name, buf in model.named_buffers():
            if name in sd and buf.is_floating_point():          # running_mean / running_var (num_batches_tracked is equal)
                t = buf.detach().clone()
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                sd[name] = t / world
        writes = dist.get_rank(group) == 0
    return writes, {k: v.cpu() for k, v in sd.items()}


def save_checkpoint(path, model, optimizer, start_epoch_nb, conf_thresh, best_log, train_remaining_file, device, group=None):
    """``model_ckpt.h5`` in the reference's format (train.py:241-248).  A collective under data parallelism: see
    ``_state_dict_for_save``."""
    writes, sd = _state_dict_for_save(model, group)
    if not writes:
        return
    torch.save({"start_epoch_nb": start_epoch_nb,
                "model_state_dict": sd,
                "optim_state_dict": optimizer_state_dict(optimizer, model),
                "confidence_thresh": float(conf_thresh), "rng_state": get_rng_state(device, model), "best_log": best_log,
                "train_remaining_file": train_remaining_file}, path)


def save_best(path, model, optimizer, epoch_nb, conf_thresh, group=None):
    """``model_best.h5`` (train.py:234-238).  A collective under data parallelism: see ``_state_dict_for_save``."""
    writes, sd = _state_dict_for_save(model, group)
    if not writes:
        return
    torch.save({"epoch_nb": epoch_nb, "model_state_dict": sd,
                "optim_state_dict": optimizer_state_dict(optimizer, model), "confidence_thresh": float(conf_thresh)}, path)


def load_checkpoint(path, model, optimizer=None, device="cpu", restore_rng=True):
    """-> the checkpoint dictionary; model (strict) and optimizer are restored in place.  Parameters stay views of the
    flat buffer: ``load_state_dict`` copies into them."""
    ck = torch.load(path, map_location="cpu", weights_only=False)
    model.load_state_dict(ck["model_state_dict"], strict=True)
    if optimizer is not None and "optim_state_dict" in ck:
        load_optimizer_state_dict(optimizer, model, ck["optim_state_dict"])
    if restore_rng and ck.get("rng_state") is not None:
        seed_resume(ck["rng_state"], device, model)
    return ck
