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
           (what the north star names; the only form for bag modes).
  sparse : all-gather the contribution entries the fused kernel wrote (d floats + a row id per
           (query, role): ~A_q bytes per query instead of 4 P per step), link the other ranks'
           entries into the local lists (gqe_import_entries) and all-reduce only the small dense
           relation / Pre / Post gradients.  At the Bio d=128 full mix this moves 9.4 MB per rank
           instead of 50 MB through the xGMI links.

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


def dense_spans(layout, is_table):
    """Contiguous [begin, end) float spans of the arena covering the tensors that are NOT embedding tables."""
    spans = []
    for key, (off, shape) in layout.entries.items():
        if is_table(key, shape):
            continue
        n = 1
        for x in shape:
            n *= int(x)
        end = off + (n + layout.ALIGN - 1) // layout.ALIGN * layout.ALIGN
        if spans and spans[-1][1] == off:
            spans[-1][1] = end
        else:
            spans.append([off, end])
    return [tuple(x) for x in spans]


def exchange_sparse(engine, dist, spans):
    """Sparse form (module docstring).  ``spans``: dense_spans() of the layout.  Call between the margin
    launch and the optimiser step; every rank must have produced the same number of entries per slab
    (same formulas and batch sizes, or Engine.exchange_reserve)."""
    n, contrib, rows = engine.exchange_buffers()
    r = engine.rank
    if n > 0:
        dist.all_gather_into_tensor(contrib, contrib[r * n:(r + 1) * n])
        dist.all_gather_into_tensor(rows, rows[r * n:(r + 1) * n])
        engine.import_entries(n)
    for b, e in spans:
        dist.all_reduce(engine.grads[b:e])


def exchange_gradients(flat_grads, dist, engine=None):
    """Sum the dense gradient arena over the ranks (in place).  With an Engine, the per-row
    gradient lists are folded into the dense arena first (gqe_materialize_grads)."""
    if engine is not None:
        engine.materialize()
    if dist is not None:
        dist.all_reduce(flat_grads)
    return flat_grads
