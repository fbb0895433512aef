"""Data-parallel training over the GPUs of one node.

The path shards by queries: every rank holds a full replica of the parameter arena, draws
the SAME formula per batch (shared seed), trains on its own slice of that formula's queries
and scales its loss weights by 1/world, so that after ONE sum all-reduce of the contiguous
gradient arena every rank holds the gradient of the mean loss over the global batch and
applies the identical fused Adam step.  The reference has no counterpart (single process);
the definition of correctness is: W ranks x batch b == 1 rank x batch W*b on the
concatenated queries (tests/test_parallel_gloo.py, 2 gloo ranks on CPU).

Two exchange forms (both leave bit-identical replicas):
  dense  : fold the per-row gradient lists into the dense arena, all-reduce all P floats
           (what the north star names).
  sparse : ONE all-gather of per-rank slabs = the contribution entries the fused kernel wrote
           (d floats + a row id per (query, role)) followed by the small dense relation / Pre /
           Post gradients; the other ranks' entries are linked into the local lists and the dense
           parts summed in rank order (gqe_import_entries).  At the Bio d=128 full mix a slab is
           ~10 MB per rank instead of the 50 MB arena.

The collectives are ``torch.distributed`` ones — backend "nccl" is RCCL over xGMI on ROCm,
"gloo" in the CPU tests.
"""
from __future__ import annotations

import os


def init_from_env(backend=None):
    """(rank, world, local_rank, dist-or-None) from the torchrun environment."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        return rank, world, local_rank, None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kwargs = {}
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        kwargs["device_id"] = torch.device("cuda", local_rank)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
    dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local_rank, dist


def rank_slice(n_queries, batch_size, step, rank, world):
    """Slice [start, end) of a formula's query list that ``rank`` trains on at ``step``:
    the reference's wrap-around rule (train_helpers.py:102-104) with the iteration counter
    replaced by step*world + rank, so the W ranks of a step cover W consecutive slices."""
    it = step * world + rank
    start = (it * batch_size) % n_queries
    end = min(((it + 1) * batch_size) % n_queries, n_queries)
    end = n_queries if end <= start else end
    return start, end


def dp_weight(loss_weight, world):
    """Per-rank loss weight: the W per-rank mean losses average to the global mean."""
    return loss_weight / float(world)


def exchange_sparse(engine, dist):
    """Sparse form (module docstring): ONE in-place all-gather of the ranks' slabs.  Call between the margin
    launch and the optimiser step; every rank must use the same slab size (same formulas and batch sizes, or
    Engine.exchange_reserve)."""
    S, slabs = engine.export_entries()
    r = engine.rank
    dist.all_gather_into_tensor(slabs, slabs[r * S:(r + 1) * S])
    engine.import_entries(S)


def exchange_gradients(flat_grads, dist, engine=None):
    """Sum the dense gradient arena over the ranks (in place).  With an Engine, the per-row
    gradient lists are folded into the dense arena first (gqe_materialize_grads)."""
    if engine is not None:
        engine.materialize()
    if dist is not None:
        dist.all_reduce(flat_grads)
    return flat_grads
