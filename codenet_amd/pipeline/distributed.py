"""Multi-GPU plumbing: start-up broadcast, image shards, detections all_gather, the
global-range parity mode's switch.

Part of codenet_amd.pipeline (split by concern in round 6; `from codenet_amd import pipeline` exposes every name as
before)."""
import os

import torch
import torch.nn as nn



def broadcast_parameters(net, src=0):
    """Start-up broadcast of every parameter and buffer from `src` as ONE flat tensor
    (RCCL over xGMI when the process group backend is nccl; gloo in the CPU tests)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    tensors = list(net.parameters()) + [b for b in net.buffers() if b.dtype.is_floating_point]
    if not tensors:
        return 0
    with torch.no_grad():
        flat = torch.cat([t.detach().reshape(-1).float() for t in tensors])
        dist.broadcast(flat, src=src)
        off = 0
        for t in tensors:
            # copy into the parameter / buffer itself (not .data): the in-place write bumps the tensor's
            # version counter, which is what the derived-weight caches (fake-quantised, folded, int8 forms,
            # BN affines) are keyed on
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
    return flat.numel() * 4


def exchange_shard_sizes(n_local, device=None):
    """ONE start-up exchange of the ranks' per-batch image counts (shard sizes are static for a run): the list
    `counts` that gather_detections() then takes, so that no batch pays a count all_gather + host sync."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [int(n_local)]
    world = dist.get_world_size()
    nb = torch.tensor([int(n_local)], device=device, dtype=torch.int64)
    parts = [torch.zeros_like(nb) for _ in range(world)]
    dist.all_gather(parts, nb)
    return [int(c.item()) for c in parts]


def gather_detections(dets, dst=None, counts=None):
    """Per-batch collection of every rank's detections [B, K, 6] (collective 2; the
    reference's only mechanism is DataParallel's gather, lib/models/data_parallel.py:64-84,120-129).
    all_gather over RCCL (gloo in the CPU tests).  `counts` = the ranks' shard sizes from exchange_shard_sizes()
    (static per run): with it a batch is ONE collective and no host synchronisation -- equal shards gather
    straight into one [world*B, K, 6] tensor; shards that differ by an image (shard_range) are padded to the
    largest.  Without `counts` the sizes are exchanged first (one extra small all_gather + a host read per call;
    kept for one-off calls).  Returns [sum B_r, K, 6] in rank order on every rank (dst=None) or only on rank dst
    (other ranks: None)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dets
    world = dist.get_world_size()
    if counts is None:
        counts = exchange_shard_sizes(dets.shape[0], dets.device)
    if len(counts) != world or counts[dist.get_rank()] != dets.shape[0]:
        raise ValueError("gather_detections: counts %r do not describe this rank's shard of %d images"
                         % (counts, dets.shape[0]))
    bmax = max(counts)
    pad = dets if dets.shape[0] == bmax else torch.cat(
        [dets, dets.new_zeros((bmax - dets.shape[0],) + tuple(dets.shape[1:]))])
    pad = pad.contiguous()
    if dst is None and min(counts) == bmax:
        out = pad.new_empty((world * bmax,) + tuple(pad.shape[1:]))
        dist.all_gather_into_tensor(out, pad)
        return out
    if dst is None:
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
    else:
        parts = [torch.empty_like(pad) for _ in range(world)] if dist.get_rank() == dst else None
        dist.gather(pad, parts, dst=dst)
        if parts is None:
            return None
    return torch.cat([p[:c] for p, c in zip(parts, counts)])


def set_global_range(model, flag=True):
    """Multi-process parity mode of every QuantAct of `model` (collective 3): the batch extremes
    are all-reduced (MIN / MAX, two 4-byte collectives per QuantAct call, RCCL over xGMI / gloo) before the range
    update, so R ranks x B images track the ranges of one R*B-image run.  The three deform stages stay on the fused
    schedule (round 5: the stage call is split at its QuantActs, FusedHotPath._global_commit); backbone and heads keep
    the module-by-module path in this mode (their fused schedules update ranges inside the producing kernels).  Returns
    the number of QuantActs set."""
    from ..portable_quantizer.quant_modules import QuantAct
    n = 0
    for m in model.modules():
        if isinstance(m, QuantAct):
            m.global_range = bool(flag)
            n += 1
    return n


def shard_range(total, rank, world):
    """Contiguous images [lo, hi) of a `total`-image batch owned by `rank` (sizes differ by <= 1)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
