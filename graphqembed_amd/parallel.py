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

  sharded: "owner computes" (include/gqe.h, gqe_set_shard): the tables are NOT replicated — rank k owns the rows
           r % W == k of every table together with their Adam moments.  Per step a rank fetches the rows its batch
           reads from their owners (all-to-all), sends each row's gradient contribution back to its owner
           (all-to-all), all-reduces the small relation / Pre / Post gradients, and runs the fused Adam pass over its
           own shards only.  Optimiser bytes per rank fall as 1/W and the inbound traffic is two batches' worth of
           rows, independent of W.

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


def _mark(events, name):
    """events: None, or a dict the caller collects per-phase torch.cuda.Event pairs in (bench.py: exchange_parts_ms)."""
    if events is None:
        return None
    import torch
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    events.setdefault(name, []).append(e)
    return e


def exchange_sparse(engine, dist, events=None):
    """Sparse form (module docstring): ONE in-place all-gather of the ranks' slabs.  Call between the margin
    launch and the optimiser step; every rank must use the same slab size (same formulas and batch sizes, or
    Engine.exchange_reserve).  ``events``: a dict that receives an event at every phase boundary (start, exported,
    gathered, imported)."""
    _mark(events, "start")
    S, slabs = engine.export_entries()
    _mark(events, "exported")
    r = engine.rank
    dist.all_gather_into_tensor(slabs, slabs[r * S:(r + 1) * S])
    _mark(events, "gathered")
    engine.import_entries(S)
    _mark(events, "imported")


def exchange_gradients(flat_grads, dist, engine=None, events=None):
    """Sum the dense gradient arena over the ranks (in place).  With an Engine, the per-row
    gradient lists are folded into the dense arena first (gqe_materialize_grads).  ``events``: as exchange_sparse
    (start, materialized, reduced)."""
    _mark(events, "start")
    if engine is not None:
        engine.materialize()
    _mark(events, "materialized")
    if dist is not None:
        dist.all_reduce(flat_grads)
    _mark(events, "reduced")
    return flat_grads


class RcclComm(object):
    """An RCCL communicator of the library's own (for gqe_allreduce_grads): ncclGetUniqueId on rank 0, the id handed to
    the other ranks through ``torch.distributed`` (or nothing for a single rank), ncclCommInitRank on every rank."""

    def __init__(self, rank=0, world=1, dist=None, device=None):
        import ctypes as C
        import torch
        self._C = C
        self.lib = C.CDLL("librccl.so")

        class _Id(C.Structure):                     # ncclUniqueId: 128 opaque bytes, passed BY VALUE to ncclCommInitRank
            _fields_ = [("internal", C.c_ubyte * 128)]
        uid = _Id()
        if rank == 0:
            self.lib.ncclGetUniqueId.argtypes = [C.POINTER(_Id)]
            rc = self.lib.ncclGetUniqueId(C.byref(uid))
            if rc != 0:
                raise RuntimeError("ncclGetUniqueId failed: %d" % rc)
        if world > 1:
            if dist is None:
                raise ValueError("RcclComm(world > 1) needs torch.distributed to hand out the unique id")
            t = torch.tensor(list(uid.internal), dtype=torch.uint8, device=device if dist.get_backend() == "nccl" else "cpu")
            dist.broadcast(t, 0)
            for i, v in enumerate(t.cpu().tolist()):
                uid.internal[i] = v
        self.comm = C.c_void_p()
        self.lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _Id, C.c_int]
        self.lib.ncclCommInitRank.restype = C.c_int
        rc = self.lib.ncclCommInitRank(C.byref(self.comm), world, uid, rank)
        if rc != 0:
            raise RuntimeError("ncclCommInitRank failed: %d" % rc)

    @property
    def handle(self):
        return self.comm.value

    def close(self):
        if self.comm:
            self.lib.ncclCommDestroy.argtypes = [self._C.c_void_p]
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None


# ---- row-sharded data parallelism ("owner computes") ---------------------------------------------------------
def shard_rows(n_rows, world):
    """Local rows of a table of ``n_rows`` global rows: the same on every rank (the last ranks' tails are padding)."""
    return -(-int(n_rows) // int(world))


def shard_of(full, rank, world):
    """Rows rank, rank + world, ... of a [rows, d] array / tensor, zero-padded to shard_rows(rows, world) rows."""
    part = full[rank::world]
    need = shard_rows(full.shape[0], world) - part.shape[0]
    if need == 0:
        return part
    if hasattr(part, "new_zeros"):
        import torch
        return torch.cat([part, part.new_zeros((need,) + tuple(part.shape[1:]))])
    import numpy as np
    return np.concatenate([part, np.zeros((need,) + part.shape[1:], dtype=part.dtype)])


def shard_plan_numpy(idx, table_of_idx, head_base, world, bag_tables=()):
    """What gqe_shard_plan computes, in numpy (host logic of the row-sharded protocol; CPU tests, cross-check of the
    library): idx[n] global rows, table_of_idx[n] the table each index names, head_base[t] the first list-head index of
    local table t.  Returns (positions[n], requests grouped by owner, send_counts[world]): a stable counting sort of
    the feed by owner = row % world; a request is head_base[table] + row // world.  Indices into ``bag_tables``
    (replicated EmbeddingBag tables) are bag ids: they pass through as their own position and are not requested."""
    import numpy as np
    idx = np.asarray(idx, dtype=np.int64)
    tid = np.asarray(table_of_idx)
    direct = ~np.isin(tid, list(bag_tables)) if len(bag_tables) else np.ones(len(idx), dtype=bool)
    owner = idx[direct] % world
    order = np.argsort(owner, kind="stable")
    positions = idx.astype(np.int32).copy()
    pos_direct = np.empty(int(direct.sum()), dtype=np.int32)
    pos_direct[order] = np.arange(len(order), dtype=np.int32)
    positions[direct] = pos_direct
    req = (np.asarray(head_base, dtype=np.int64)[tid[direct]] + idx[direct] // world)[order].astype(np.int32)
    return positions, req, np.bincount(owner, minlength=world).astype(np.int64)


def _all_to_all(dist, out, inp, out_splits, in_splits):
    """all_to_all_single along dim 0; gloo moves host memory only, so device tensors are staged (test boxes where
    several ranks share one GPU — RCCL takes the device buffers directly)."""
    if inp.is_cuda and dist.get_backend() == "gloo":
        o = out.cpu()
        dist.all_to_all_single(o, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits)


class TorchTransport(object):
    """gqe_transport over ``torch.distributed`` (include/gqe.h): the two collectives gqe_shard_step needs, as callbacks.
    With the gloo backend device buffers are staged through the host (several ranks sharing one GPU: the tests); a nccl
    process group moves them in place — but there the library's own RCCL path (Engine.shard_open(nccl_comm=...)) is the
    one to use.  The callbacks run on the calling thread inside gqe_shard_step; the stream is torch's current stream."""

    def __init__(self, engine, dist, skip_own=True):
        """``skip_own``: the all-to-all leaves this rank's own block alone (gqe_transport.skips_own_block): the library keeps
        it in place, as on its RCCL path."""
        import torch
        from .engine import A2A_FN, ALLREDUCE_FN, gqe_transport
        self.engine, self.dist, self.error = engine, dist, None
        world, me = dist.get_world_size(), dist.get_rank()

        def a2a(user, send, scounts, recv, rcounts, elem_bytes, stream):
            try:
                sc = [int(scounts[i]) for i in range(world)]
                rc = [int(rcounts[i]) for i in range(world)]
                cols = int(elem_bytes) // 4
                src = engine.view_bytes(send, sum(sc) * elem_bytes).view(torch.float32).view(sum(sc), cols)
                dst = engine.view_bytes(recv, sum(rc) * elem_bytes).view(torch.float32).view(sum(rc), cols)
                if not skip_own:
                    _all_to_all(dist, dst, src, rc, sc)
                    return 0
                # the own block stays where it is: exchange the other blocks, compacted on both sides
                so, ro = sum(sc[:me]), sum(rc[:me])
                inp = torch.cat([src[:so], src[so + sc[me]:]])
                out = torch.empty((sum(rc) - rc[me], cols), dtype=torch.float32, device=dst.device)
                sc[me] = rc[me] = 0
                _all_to_all(dist, out, inp, rc, sc)
                dst[:ro].copy_(out[:ro])
                dst[ro + int(rcounts[me]):].copy_(out[ro:])
                return 0
            except Exception as e:                      # noqa: an exception must not unwind through the C frames
                self.error = e
                return 1

        def allreduce(user, buf, n, stream):
            try:
                t = engine.view_bytes(buf, int(n) * 4).view(torch.float32)
                if t.is_cuda and dist.get_backend() == "gloo":
                    h = t.cpu()
                    dist.all_reduce(h)
                    t.copy_(h)
                else:
                    dist.all_reduce(t)
                return 0
            except Exception as e:                      # noqa
                self.error = e
                return 1
        self._a2a, self._ar = A2A_FN(a2a), ALLREDUCE_FN(allreduce)     # keep the thunks alive
        self.struct = gqe_transport(None, self._a2a, self._ar, 1 if skip_own else 0)


def shard_session(engine, dist, rank=0, world=1, skip_own=True):
    """Open the row-sharded session of ``engine`` (gqe_shard_open): agree on a plan-board name, pick the transport — the
    library's own RCCL calls on a nccl process group (a communicator of its own: RcclComm), callbacks over
    torch.distributed otherwise (gloo), nothing for a single rank.  Returns the object that keeps the transport alive."""
    if dist is None or world == 1:
        engine.shard_open(None)
        return None
    import uuid
    name = [uuid.uuid4().hex[:16] if rank == 0 else None]
    dist.broadcast_object_list(name, src=0)
    if dist.get_backend() == "nccl":
        comm = RcclComm(rank, world, dist, device=engine.device)
        engine.shard_open(name[0], nccl_comm=comm.handle)
        return comm
    tr = TorchTransport(engine, dist, skip_own=skip_own)
    engine.shard_open(name[0], transport=tr.struct)
    return tr


def shard_prepare(engine, dist, descs, idx, with_negatives=True):
    """Plan one call in row-sharded mode: sort its index feed (numpy int32, GLOBAL rows) by owner, tell every owner
    which of its rows this rank will request (one all-to-all of counts, one of row ids) and freeze the launch
    descriptors.  The result is reusable for as long as the batch is (bench.py replays pre-sampled iterations; a trainer
    prepares the next iteration while the current one runs)."""
    import torch
    pos, req, send = engine.shard_plan(descs, idx, with_negatives)
    dev = engine.device
    send_counts = [int(c) for c in send]
    cs = torch.tensor(send_counts, dtype=torch.int64, device=dev)
    cr = torch.empty_like(cs)
    _all_to_all(dist, cr, cs, None, None)
    recv_counts = [int(c) for c in cr.tolist()]
    n_send, n_recv = int(sum(send_counts)), int(sum(recv_counts))
    v = engine.shard_views()
    if n_recv > v["cap_recv"]:
        raise RuntimeError("row-sharded mode: %d rows requested from rank %d exceed its receive capacity %d"
                           % (n_recv, engine.shard_rank, v["cap_recv"]))
    req_send = torch.from_numpy(req[:n_send]).to(dev)
    req_recv = torch.empty(n_recv, dtype=torch.int32, device=dev)
    _all_to_all(dist, req_recv, req_send, recv_counts, send_counts)
    total = sum(dsc["n"] for dsc in descs)
    pos_dev = torch.from_numpy(pos).to(dev)
    ps = {"arr": engine.make_batches(descs), "n": len(descs), "idx": pos_dev, "idx_ptr": __import__("ctypes").c_void_p(pos_dev.data_ptr()),
          "n_idx": int(pos_dev.numel()), "queries": total,
          "losses": torch.zeros(len(descs) + 1, dtype=torch.float32, device=dev),
          "send_counts": send_counts, "recv_counts": recv_counts, "n_send": n_send, "n_recv": n_recv, "req_recv": req_recv,
          # the views the per-step collectives move (sliced once: the step itself should cost no Python beyond the calls)
          "workspace": engine.workspace, "rows_send": v["rows_send"][:n_recv], "fetched": v["fetched"][:n_send],
          "contrib_send": v["contrib_send"][:n_send], "contrib_recv": v["contrib_recv"][:n_recv],
          "dense": [engine.grads[off:off + n] for off, n in engine.dense_spans()],
          # bag (EmbeddingBag) tables are replicated: their gradient is folded into the dense arena and all-reduced
          "bag_keys": list(engine.bag_keys),
          "bag_grads": [engine.layout.view(engine.grads, k).view(-1) for k in engine.bag_keys]}
    return ps


def shard_fetch(engine, dist, ps):
    """Serve the rows the other ranks asked this one for, and receive the rows this rank's batch reads."""
    if ps["workspace"] is not engine.workspace:
        raise RuntimeError("the engine's workspace was re-bound after shard_prepare: prepare the batch again")
    engine.shard_serve(ps["req_recv"], ps["n_recv"], ps["rows_send"])
    _all_to_all(dist, ps["fetched"], ps["rows_send"], ps["send_counts"], ps["recv_counts"])


def shard_exchange(engine, dist, ps):
    """After the margin launch: every row's gradient contribution goes to the row's owner, which links it onto its
    lists; the relation / Pre / Post gradients (replicated tensors) are summed over the ranks."""
    _all_to_all(dist, ps["contrib_recv"], ps["contrib_send"], ps["recv_counts"], ps["send_counts"])
    engine.shard_link(ps["req_recv"], ps["n_recv"])
    if ps["bag_keys"]:
        engine.materialize_tables(ps["bag_keys"])
        for g in ps["bag_grads"]:
            dist.all_reduce(g)
    _ordered_sum(dist, ps["dense"])


def _ordered_sum(dist, spans):
    """Sum of the ranks' copies of ``spans`` (the relation / Pre / Post gradients) IN RANK ORDER, written back in place: what
    gqe_shard_step does with the blocks that ride in its contribution exchange (csrc/gqe_shard_step.h, gqe_launch_dense_sum) —
    an all-reduce sums in an order of the transport's choosing, and from three ranks on that is a different float."""
    import torch
    world = dist.get_world_size()
    if world == 1 or not spans:
        return
    flat = torch.cat([s.reshape(-1) for s in spans])
    if flat.is_cuda and dist.get_backend() == "gloo":
        host = flat.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host)
        parts = [x.to(flat.device) for x in parts]
    else:
        parts = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(parts, flat)
    acc = torch.zeros_like(flat)
    for x in parts:
        acc += x
    at = 0
    for s in spans:
        n = s.numel()
        s.copy_(acc[at:at + n].view_as(s))
        at += n


def shard_margin_step(engine, dist, ps, adam=None, lr=0.01, betas=(0.9, 0.999), eps=1e-8):
    """One training iteration in row-sharded mode on a prepared batch: fetch rows, fused forward / backward on the
    fetched rows, route the contributions to their owners, step the own shards."""
    shard_fetch(engine, dist, ps)
    engine.run_margin(ps)
    shard_exchange(engine, dist, ps)
    if adam is not None:
        engine.run_adam(adam, lr, betas, eps)
    return ps["losses"]


def shard_forward(engine, dist, ps, n_scores, out=None):
    """gqe_forward in row-sharded mode (``ps`` from shard_prepare(..., with_negatives=False))."""
    import torch
    shard_fetch(engine, dist, ps)
    scores = out if out is not None else torch.empty(n_scores, dtype=torch.float32, device=engine.device)
    engine._check(engine.lib.gqe_forward(engine.ctx, ps["arr"], ps["n"], ps["idx_ptr"], ps["n_idx"], 1, scores.data_ptr(), engine._stream()))
    return scores
