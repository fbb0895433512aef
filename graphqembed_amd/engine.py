"""ctypes binding of libgqe.so (include/gqe.h) + the flat parameter arena.

PyTorch is used only as the owner of device memory and streams: every pointer handed
to the library is ``tensor.data_ptr()`` of a tensor this module keeps alive, and every
call is enqueued on ``torch.cuda.current_stream()``.

There is NO CPU fallback: importing the package works without the library (so host-side
code and CPU tests can run), but anything that computes raises ``GqeLibraryError`` if
``libgqe.so`` is missing or no HIP device is present.
"""
from __future__ import annotations

import ctypes as C
import os
from collections import OrderedDict

import numpy as np

ABI_VERSION = 1
MAX_BRANCH, MAX_HOPS, MAX_BATCHES, MAX_DIM = 3, 3, 64, 256
TQ = 16

DECODERS = {"bilinear-diag": 0, "transe": 1, "bilinear": 2}           # utils.py:128-137
INTER_DECODERS = {"min": 0, "mean": 1, "min-simple": 2, "mean-simple": 3}  # utils.py:139-150
QTYPES = {"1-chain": 0, "2-chain": 1, "3-chain": 2, "2-inter": 3, "3-inter": 4,
          "3-inter_chain": 5, "3-chain_inter": 6}

LIB_PATH = os.environ.get("GQE_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgqe.so")   # GQE_LIB: a probe build (tools/spill_probe.py)


class GqeLibraryError(RuntimeError):
    pass


class GqeError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "libgqe error %d: %s" % (code, msg))
        self.code = code


class gqe_config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("dim", C.c_int32),
                ("decoder", C.c_int32), ("inter", C.c_int32), ("reserved", C.c_int32 * 3)]


class gqe_batch(C.Structure):
    _fields_ = [("qtype", C.c_int32), ("n_queries", C.c_int32), ("n_anchors", C.c_int32),
                ("idx_offset", C.c_int32),
                ("target_table", C.c_int64), ("anchor_table", C.c_int64 * MAX_BRANCH),
                ("n_hops", C.c_int32 * MAX_BRANCH), ("n_final", C.c_int32),
                ("hop_param", (C.c_int64 * MAX_HOPS) * MAX_BRANCH),
                ("final_param", C.c_int64), ("pre_param", C.c_int64), ("post_param", C.c_int64),
                ("margin", C.c_float), ("loss_weight", C.c_float),
                ("out_offset", C.c_int32), ("n_candidates", C.c_int32)]


class gqe_shard_buffers(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("req_send", "req_recv", "rows_send", "fetched", "contrib_send", "contrib_recv",
                                         "cap_send", "cap_recv")]


class gqe_segment(C.Structure):
    _fields_ = [("offset", C.c_int64), ("numel", C.c_int64), ("step", C.c_int32), ("reserved", C.c_int32)]


# gqe_transport (include/gqe.h): the two collectives of the row-sharded step as callbacks
A2A_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64), C.c_int64, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


class gqe_transport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_to_all", A2A_FN), ("all_reduce_sum_f32", ALLREDUCE_FN), ("skips_own_block", C.c_int32)]


# every symbol include/gqe.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = OrderedDict([
    ("gqe_abi_version", (C.c_int, [])),
    ("gqe_dim_supported", (C.c_int, [C.c_int32, C.c_int32, C.c_int32])),
    ("gqe_last_error", (C.c_char_p, [_P])),
    ("gqe_create", (C.c_int, [C.POINTER(gqe_config), C.POINTER(_P)])),
    ("gqe_destroy", (C.c_int, [_P])),
    ("gqe_bind_arena", (C.c_int, [_P, _P, _P, _P, _P, C.c_int64])),
    ("gqe_params_changed", (C.c_int, [_P])),
    ("gqe_set_deferred_gemm", (C.c_int, [_P, C.c_int32])),
    ("gqe_deferred_gemm_rides", (C.c_int64, [_P])),
    ("gqe_set_tables", (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32])),
    ("gqe_set_bag", (C.c_int, [_P, C.c_int64, _P, _P, C.c_int64, C.c_int32])),
    ("gqe_set_limits", (C.c_int, [_P, C.c_int32, C.c_int32])),
    ("gqe_workspace_bytes", (C.c_int64, [_P, C.c_int64, C.c_int32])),
    ("gqe_bind_workspace", (C.c_int, [_P, _P, C.c_int64, _P])),
    ("gqe_materialize_grads", (C.c_int, [_P, _P])),
    ("gqe_materialize_tables", (C.c_int, [_P, C.POINTER(C.c_int64), C.c_int32, _P])),
    ("gqe_set_lazy_adam", (C.c_int, [_P, C.c_int32])),
    ("gqe_optimizer_sync", (C.c_int, [_P, _P])),
    ("gqe_lazy_prefetch", (C.c_int, [_P, C.POINTER(gqe_batch), C.c_int32, _P, C.c_int64, C.c_int32])),
    ("gqe_set_exchange", (C.c_int, [_P, C.c_int32, C.c_int32])),
    ("gqe_exchange_reserve", (C.c_int, [_P, C.c_int64])),
    ("gqe_export_entries", (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _P])),
    ("gqe_import_entries", (C.c_int, [_P, C.c_int64, _P])),
    ("gqe_set_shard", (C.c_int, [_P, C.c_int32, C.c_int32])),
    ("gqe_set_ordered_sums", (C.c_int, [_P, C.c_int32])),
    ("gqe_hot_rows", (C.c_int, [_P, C.POINTER(C.c_int32)])),
    ("gqe_hot_sub_lists", (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)])),
    ("gqe_shard_layout", (C.c_int, [_P, C.POINTER(gqe_shard_buffers)])),
    ("gqe_shard_plan", (C.c_int, [_P, C.POINTER(gqe_batch), C.c_int32, _P, C.c_int64, C.c_int32, _P, _P, _P])),
    ("gqe_shard_serve", (C.c_int, [_P, _P, C.c_int64, _P, _P])),
    ("gqe_shard_link", (C.c_int, [_P, _P, C.c_int64, _P])),
    ("gqe_shard_open", (C.c_int, [_P, C.c_char_p, _P, C.POINTER(gqe_transport)])),
    ("gqe_shard_close", (C.c_int, [_P])),
    ("gqe_shard_profile", (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64)])),
    ("gqe_shard_post", (C.c_int, [_P, C.POINTER(gqe_batch), C.c_int32, _P, C.c_int64, C.c_int32, C.POINTER(gqe_segment), C.c_int32])),
    ("gqe_shard_step", (C.c_int, [_P, C.c_float, C.c_float, C.c_float, C.c_float, _P, _P, _P, _P])),
    ("gqe_shard_forward", (C.c_int, [_P, _P, _P])),
    ("gqe_rank_candidates", (C.c_int, [_P, _P, _P, C.c_int32, _P, _P])),
    ("gqe_auc_pair_counts", (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, _P, _P])),
    ("gqe_forward", (C.c_int, [_P, C.POINTER(gqe_batch), C.c_int32, _P, C.c_int64, C.c_int32, _P, _P])),
    ("gqe_margin_fwd_bwd", (C.c_int, [_P, C.POINTER(gqe_batch), C.c_int32, _P, C.c_int64, C.c_int32, _P, _P, _P, _P])),
    ("gqe_allreduce_grads", (C.c_int, [_P, _P, _P])),
    ("gqe_adam_step", (C.c_int, [_P, C.POINTER(gqe_segment), C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, _P])),
    ("gqe_encode_rows", (C.c_int, [_P, C.c_int64, _P, C.c_int32, _P, _P])),
    ("gqe_decoder_project", (C.c_int, [_P, C.c_int64, _P, C.c_int32, _P, _P])),
    ("gqe_decoder_forward", (C.c_int, [_P, C.POINTER(C.c_int64), C.c_int32, _P, _P, C.c_int32, _P, _P])),
    ("gqe_set_intersection", (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P, _P, C.c_int32, _P, _P])),
    ("gqe_train_step", (C.c_int, [_P, C.POINTER(gqe_batch), C.c_int32, _P, C.c_int64, C.c_int32, C.POINTER(gqe_segment), C.c_int32,
                                  C.c_float, C.c_float, C.c_float, C.c_float, _P, _P])),
    ("gqe_split_steps", (C.c_int64, [_P])),
    ("gqe_sgd_step", (C.c_int, [_P, C.POINTER(gqe_segment), C.c_int32, C.c_float, _P])),
    ("gqe_zero_grads", (C.c_int, [_P, C.POINTER(gqe_segment), C.c_int32, _P])),
    ("gqe_feeder_create", (C.c_int, [_P, C.c_uint64, C.c_int32, C.c_float, C.c_float, C.POINTER(_P)])),
    ("gqe_feeder_destroy", (C.c_int, [_P])),
    ("gqe_feeder_add_pool", (C.c_int, [_P, C.POINTER(gqe_batch), C.c_int64, _P, _P, _P, _P])),
    ("gqe_feeder_set_mode_rows", (C.c_int, [_P, C.c_int64, _P, C.c_int64])),
    ("gqe_feeder_set_feed", (C.c_int, [_P, C.c_int32])),
    ("gqe_feeder_add_pool_lists", (C.c_int, [_P, C.POINTER(gqe_batch), C.c_int64, _P, _P, _P, _P, _P, _P])),
    ("gqe_feeder_set_reference_streams", (C.c_int, [_P, _P, _P])),
    ("gqe_feeder_set_pvals", (C.c_int, [_P, C.c_int32, _P, C.c_int32])),
    ("gqe_feeder_set_type_order", (C.c_int, [_P, _P, C.c_int32])),
    ("gqe_feeder_set_loss_stride", (C.c_int, [_P, C.c_int64])),
    ("gqe_feeder_set_sgd", (C.c_int, [_P, C.c_int32])),
    ("gqe_adam_step_count", (C.c_int, [_P, C.c_int64, C.POINTER(C.c_int32)])),
    ("gqe_set_adam_step_count", (C.c_int, [_P, C.c_int64, C.c_int32])),
    ("gqe_feeder_queries", (C.c_int64, [_P])),
    ("gqe_feeder_host_seconds", (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double)])),
    ("gqe_feeder_debug_feed", (C.c_int, [_P, C.c_int64, C.POINTER(gqe_batch), C.c_int32, C.POINTER(C.c_int32), _P, C.c_int64,
                                         C.POINTER(C.c_int64)])),
    ("gqe_feeder_run", (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, _P, _P])),
    ("gqe_timing_enable", (C.c_int, [_P, C.c_int32])),
    ("gqe_debug_profile", (C.c_int, [_P, _P])),
    ("gqe_debug_fused_variant", (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)])),
    ("gqe_timing_read", (C.c_int, [_P, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_int32)])),
])

_lib = None


def load_library(path=None):
    """dlopen libgqe.so and type every entry point; raises GqeLibraryError if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise GqeLibraryError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). There is no CPU fallback." % p)
    # torch first: it brings its OWN libamdhip64 / libhsa-runtime64 (torch/lib).  Loaded after libgqe.so had bound the
    # system ROCm's copies, the process would hold two HIP runtimes and the second one to initialise sees no device
    # ("no ROCm-capable device is detected" from gqe_create, although torch.cuda.is_available())
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        lib = C.CDLL(p)
    except OSError as e:
        raise GqeLibraryError("cannot load %s: %s" % (p, e))
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.gqe_abi_version() != ABI_VERSION:
        raise GqeLibraryError("libgqe ABI %d != binding ABI %d" % (lib.gqe_abi_version(), ABI_VERSION))
    if path is None:
        _lib = lib
    return lib


def _align(x, a):
    return (x + a - 1) // a * a


class ArenaLayout(object):
    """Where each parameter tensor (state_dict key of the reference) lives in the flat arena."""

    ALIGN = 64  # floats

    def __init__(self):
        self.entries = OrderedDict()   # key -> (offset, shape)
        self.total = 0

    def add(self, key, shape):
        if key in self.entries:
            raise ValueError("duplicate parameter %r" % key)
        numel = int(np.prod(shape))
        self.entries[key] = (self.total, tuple(int(s) for s in shape))
        self.total = _align(self.total + numel, self.ALIGN)
        return self.entries[key][0]

    def offset(self, key):
        return self.entries[key][0]

    def numel(self, key):
        return int(np.prod(self.entries[key][1]))

    def view(self, flat, key):
        off, shape = self.entries[key]
        return flat[off:off + int(np.prod(shape))].view(*shape)


class Engine(object):
    """One gqe_ctx + the tensors it borrows."""

    def __init__(self, dim, decoder, inter_decoder, layout, device=None, max_queries=8192, max_batches=16, bags=None,
                 rank=0, world=1, lazy_adam=False, max_formulas=0, shard=None, ordered_sums=False):
        """``bags``: {table key: (ptr int32[n+1], ids int32[nnz])} for modes whose feature is an
        nn.EmbeddingBag (mean over table rows) — an index into such a mode is a bag index.
        ``world`` > 1 (and no bags): size the gradient-entry space for the data-parallel exchange
        (include/gqe.h, gqe_set_exchange) and make the optimiser sum lists in a replica-independent order.
        ``lazy_adam``: deferred bit-exact Adam (include/gqe.h, gqe_set_lazy_adam); ``params`` / ``exp_avg`` /
        ``exp_avg_sq`` then synchronise on access.
        ``shard`` = (rank, world): row-sharded data parallelism (include/gqe.h, gqe_set_shard): ``layout`` describes this
        rank's SHARDS (tables with ceil(rows / world) rows: local row i = global row i * world + rank); index feeds name
        global rows and go through ``shard_plan`` (graphqembed_amd/parallel.py drives the protocol)."""
        import torch
        if not torch.cuda.is_available():
            raise GqeLibraryError("no HIP device visible to torch; the query path only runs on an MI355X "
                                  "(there is no CPU fallback)")
        self.lib = load_library()
        self.torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.dim = int(dim)
        self.decoder = decoder
        self.inter_decoder = inter_decoder
        self.layout = layout
        cfg = gqe_config(ABI_VERSION, self.device.index or 0, self.dim, DECODERS[decoder], INTER_DECODERS[inter_decoder])
        handle = _P()
        rc = self.lib.gqe_create(C.byref(cfg), C.byref(handle))
        if rc != 0:
            raise GqeError(rc, (self.lib.gqe_last_error(None) or b"").decode())
        self.ctx = handle
        n = max(layout.total, 64)
        z = lambda: torch.zeros(n, dtype=torch.float32, device=self.device)
        self.lazy_adam = False
        self._params, self.grads, self._exp_avg, self._exp_avg_sq = z(), z(), z(), z()
        self._check(self.lib.gqe_bind_arena(self.ctx, self._params.data_ptr(), self.grads.data_ptr(),
                                            self._exp_avg.data_ptr(), self._exp_avg_sq.data_ptr(), n))
        tables = [(off, shape[0]) for k, (off, shape) in layout.entries.items()
                  if k.startswith("enc.") and len(shape) == 2 and shape[1] == self.dim]
        if tables:
            offs = (C.c_int64 * len(tables))(*[t[0] for t in tables])
            rows = (C.c_int64 * len(tables))(*[t[1] for t in tables])
            self._check(self.lib.gqe_set_tables(self.ctx, offs, rows, len(tables)))
        self._bags = {}
        self.bag_keys = list((bags or {}).keys())    # tables whose indices are bags (replicated in row-sharded mode)
        for key, (ptr, ids) in (bags or {}).items():
            ptr = np.ascontiguousarray(ptr, dtype=np.int32)
            ids = np.ascontiguousarray(ids, dtype=np.int32)
            dp, di = torch.from_numpy(ptr).to(self.device), torch.from_numpy(ids).to(self.device)
            self._bags[key] = (dp, di)      # keep the borrowed device buffers alive
            self._check(self.lib.gqe_set_bag(self.ctx, layout.offset(key), dp.data_ptr(), di.data_ptr(), len(ptr) - 1,
                                             int(np.diff(ptr).max())))
        # every tensor of the layout may be stepped; the formula-descriptor cache keeps its default unless asked
        self._check(self.lib.gqe_set_limits(self.ctx, max(len(layout.entries), 1), int(max_formulas)))
        self.rank, self.world = int(rank), int(world)
        self.sparse_exchange = self.world > 1
        if self.sparse_exchange:
            self._check(self.lib.gqe_set_exchange(self.ctx, self.rank, self.world))
        self.shard_rank, self.shard_world = (int(shard[0]), int(shard[1])) if shard else (0, 1)
        self.sharded = shard is not None             # world = 1 is allowed: the degenerate case, every row owned by this rank
        if self.sharded:
            self._check(self.lib.gqe_set_shard(self.ctx, self.shard_rank, self.shard_world))
        self._shard_views = None
        if ordered_sums:             # bit-reproducible list sums (include/gqe.h, gqe_set_ordered_sums)
            self._check(self.lib.gqe_set_ordered_sums(self.ctx, 1))
        if lazy_adam:
            self._check(self.lib.gqe_set_lazy_adam(self.ctx, 1))
            self.lazy_adam = True
        self.workspace = None
        self.max_queries = self.max_batches = 0
        self.reserve(max_queries, max_batches)
        self.steps = {k: 0 for k in layout.entries}   # per-tensor Adam step counters

    # -- plumbing ---------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise GqeError(rc, (self.lib.gqe_last_error(self.ctx) or b"").decode())

    # the arenas as the caller sees them: in lazy-Adam mode rows may owe deferred steps until synchronised
    def sync(self):
        """gqe_optimizer_sync: lazy Adam's deferred row steps, and the matrix step a split train_step left for the next call."""
        if self.workspace is not None and getattr(self, "ctx", None):
            self._check(self.lib.gqe_optimizer_sync(self.ctx, self._stream()))

    @property
    def params(self):
        """The parameter arena.  Handing it out may be followed by a write the library cannot see (initialisation, a checkpoint
        load): the operand-ordered copies of the d x d matrices are rebuilt before the next forward / backward launch
        (gqe_params_changed).  Code that keeps the tensor and writes to it LATER calls ``params_changed()`` itself."""
        self.sync()
        self.params_changed()
        return self._params

    def set_deferred_gemm(self, enable=True):
        """gqe_set_deferred_gemm (include/gqe.h): the matrix-gradient launch of margin_fwd_bwd / run_margin rides in the next
        Adam step's launch; ``losses`` of a margin call are then only defined BEHIND that step."""
        self._check(self.lib.gqe_set_deferred_gemm(self.ctx, 1 if enable else 0))

    def gemm_rides(self):
        """How many Adam passes carried a deferred pair GEMM so far (gqe_deferred_gemm_rides)."""
        return int(self.lib.gqe_deferred_gemm_rides(self.ctx))

    def params_changed(self):
        if getattr(self, "ctx", None):
            self._check(self.lib.gqe_params_changed(self.ctx))

    @property
    def exp_avg(self):
        self.sync()
        return self._exp_avg

    @property
    def exp_avg_sq(self):
        self.sync()
        return self._exp_avg_sq

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def reserve(self, max_queries, max_batches):
        if max_queries <= self.max_queries and max_batches <= self.max_batches:
            return
        if getattr(self, "_shard_session_open", False):
            # the session's plan board, pinned feeds and planning thread are sized for the bound capacities, and the peers do
            # not grow with this rank: growing here would leave the posted plans pointing into a workspace that is gone
            raise GqeError(-3, "a step of %d queries / %d batches exceeds what this Engine was sized for (%d / %d) while a row-sharded "
                           "session is open: build every rank's Engine with max_queries / max_batches for the LARGEST step (the "
                           "cross-rank maximum), or shard_close(), reserve() the same capacities on every rank and re-open"
                           % (max_queries, max_batches, self.max_queries, self.max_batches))
        self.max_queries = max(max_queries, self.max_queries)
        self.max_batches = max(max_batches, self.max_batches)
        if self.workspace is not None:
            self.materialize()            # pending gradient lists live in the old workspace
            self.sync()                   # ... and so do the per-row step counts of lazy Adam
        nbytes = self.lib.gqe_workspace_bytes(self.ctx, self.max_queries, min(self.max_batches, MAX_BATCHES))
        if nbytes < 0:
            raise GqeError(int(nbytes), "gqe_workspace_bytes")
        self.torch.cuda.synchronize(self.device)
        self.workspace = self.torch.empty(int(nbytes) + 256, dtype=self.torch.uint8, device=self.device)
        ptr = _align(self.workspace.data_ptr(), 256)
        self._check(self.lib.gqe_bind_workspace(self.ctx, ptr, int(nbytes), self._stream()))
        self._shard_views = None

    # -- row-sharded data parallelism (gqe_set_shard) ---------------------------------------------
    def shard_views(self):
        """The transport buffers of the row-sharded protocol as views of the workspace (include/gqe.h,
        gqe_shard_buffers): int32 requests sent / received, served rows, fetched rows, contributions sent / received."""
        if self._shard_views is None:
            b = gqe_shard_buffers()
            self._check(self.lib.gqe_shard_layout(self.ctx, C.byref(b)))
            base = _align(self.workspace.data_ptr(), 256) - self.workspace.data_ptr()
            t, d = self.torch, self.dim

            def view(off, n, dtype, cols):
                raw = self.workspace[base + off: base + off + 4 * n * cols].view(dtype)
                return raw if cols == 1 else raw.view(n, cols)
            self._shard_views = {
                "req_send": view(b.req_send, b.cap_send, t.int32, 1), "req_recv": view(b.req_recv, b.cap_recv, t.int32, 1),
                "rows_send": view(b.rows_send, b.cap_recv, t.float32, d), "fetched": view(b.fetched, b.cap_send, t.float32, d),
                "contrib_send": view(b.contrib_send, b.cap_send, t.float32, d),
                "contrib_recv": view(b.contrib_recv, b.cap_recv, t.float32, d),
                "cap_send": int(b.cap_send), "cap_recv": int(b.cap_recv)}
        return self._shard_views

    def shard_plan(self, descs, idx, with_negatives=True):
        """Sort a step's index feed (numpy int32, GLOBAL rows) by owner: returns (positions[n] — the feed for the
        kernels, requests[n] grouped by owner, send_counts[world])."""
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        total = sum(dsc["n"] for dsc in descs)
        self.reserve(total, len(descs))
        arr = self.make_batches(descs)
        pos = np.empty(idx.size, dtype=np.int32)
        req = np.empty(idx.size, dtype=np.int32)
        counts = np.zeros(self.shard_world, dtype=np.int64)
        self._check(self.lib.gqe_shard_plan(self.ctx, arr, len(descs), C.c_void_p(idx.ctypes.data), int(idx.size),
                                            1 if with_negatives else 0, C.c_void_p(pos.ctypes.data), C.c_void_p(req.ctypes.data),
                                            C.c_void_p(counts.ctypes.data)))
        return pos, req, counts

    # -- the row-sharded step as one call (include/gqe.h: gqe_shard_open / post / step / forward) ----------------
    def shard_open(self, session=None, nccl_comm=None, transport=None):
        """Join the node's plan board "/gqe_<session>" (the same name on every rank; None for world = 1) and name the
        transport: an ncclComm_t of RCCL (``nccl_comm``: integer handle, e.g. parallel.RcclComm.handle) or a
        ``gqe_transport`` of callbacks (parallel.TorchTransport)."""
        self._shard_transport = transport          # keep the callbacks alive
        self._check(self.lib.gqe_shard_open(self.ctx, None if session is None else session.encode(),
                                            None if nccl_comm is None else C.c_void_p(int(nccl_comm)),
                                            None if transport is None else C.byref(transport)))
        self._shard_session_open = True

    def shard_close(self):
        if getattr(self, "ctx", None):
            self._check(self.lib.gqe_shard_close(self.ctx))
        self._shard_session_open = False

    def shard_profile(self):
        """(planning-thread us per step, caller-thread us per step, steps) of the open session — include/gqe.h, gqe_shard_profile
        (zeros unless GQE_SHARD_PROFILE is set)."""
        us, n = (C.c_double * 10)(), C.c_int64(0)
        self._check(self.lib.gqe_shard_profile(self.ctx, us, C.byref(n)))
        k = max(int(n.value), 1)
        return sum(us[0:2]) / k, sum(us[2:10]) / k, int(n.value)

    def prepare_shard(self, descs, idx, keys=None, with_negatives=True):
        """Freeze one step for gqe_shard_post: ctypes descriptors, the HOST index feed of GLOBAL rows (numpy int32) and,
        for a margin step, the parameter tensors its batches touch."""
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        total = sum(dsc["n"] for dsc in descs)
        self.reserve(max(total, (int(idx.size) + 4) // 5), len(descs))   # candidate lists: every index is a fetched row
        t = self.torch
        ps = {"arr": self.make_batches(descs), "n": len(descs), "idx": idx, "idx_ptr": C.c_void_p(idx.ctypes.data), "n_idx": int(idx.size),
              "queries": total, "with_negatives": 1 if with_negatives else 0,
              "losses": t.zeros(len(descs) + 1, dtype=t.float32, device=self.device), "segs": None, "n_segs": 0}
        if with_negatives:
            keys = [k for k in self.layout.entries if k in set(keys)]
            ps["keys"], ps["segs"], ps["n_segs"] = keys, self._segments(keys, False), len(keys)
        return ps

    def shard_post(self, ps):
        """gqe_shard_post: plan the step on the host (owner sort) and publish it to the other ranks; never blocks on a GPU."""
        self._check(self.lib.gqe_shard_post(self.ctx, ps["arr"], ps["n"], ps["idx_ptr"], ps["n_idx"], ps["with_negatives"],
                                            ps["segs"], ps["n_segs"]))

    def shard_step(self, ps, lr=0.01, betas=(0.9, 0.999), eps=1e-8, pos=None, neg=None):
        """gqe_shard_step: run the oldest posted plan (must be ``ps``'s): rows in, fused forward / backward, contributions
        out, Adam on the own shards — one library call; returns ps["losses"]."""
        self._check(self.lib.gqe_shard_step(self.ctx, lr, betas[0], betas[1], eps, ps["losses"].data_ptr(),
                                            None if pos is None else pos.data_ptr(), None if neg is None else neg.data_ptr(), self._stream()))
        return ps["losses"]

    def shard_forward(self, n_scores, out=None):
        scores = out if out is not None else self.torch.empty(n_scores, dtype=self.torch.float32, device=self.device)
        self._check(self.lib.gqe_shard_forward(self.ctx, scores.data_ptr(), self._stream()))
        return scores

    def view_bytes(self, ptr, nbytes):
        """uint8 view of ``nbytes`` at device address ``ptr`` inside one of the tensors the ctx borrows (transport callbacks)."""
        t = self.torch
        for base in (self.workspace, self.grads, self._params):
            off = int(ptr) - base.data_ptr()
            if 0 <= off and off + nbytes <= base.numel() * base.element_size():
                return base.view(-1).view(t.uint8)[off:off + nbytes]
        raise ValueError("address %#x (+%d) is outside the workspace and the arenas" % (int(ptr), nbytes))

    def shard_serve(self, requests, n, rows_out):
        self._check(self.lib.gqe_shard_serve(self.ctx, C.c_void_p(requests.data_ptr()), int(n), C.c_void_p(rows_out.data_ptr()), self._stream()))

    def shard_link(self, requests, n):
        self._check(self.lib.gqe_shard_link(self.ctx, C.c_void_p(requests.data_ptr()), int(n), self._stream()))

    def dense_spans(self):
        """(offset, length) runs of the arena that are NOT embedding tables (relation vectors / matrices, Pre / Post):
        what replicas all-reduce in row-sharded mode."""
        spans = []
        for k, (off, shape) in self.layout.entries.items():
            if k.startswith("enc.") and len(shape) == 2 and shape[1] == self.dim:
                continue
            n = _align(int(np.prod(shape)), self.layout.ALIGN)
            if spans and spans[-1][0] + spans[-1][1] == off:
                spans[-1] = (spans[-1][0], spans[-1][1] + n)
            else:
                spans.append((off, n))
        return [(o, min(n, self.layout.total - o)) for o, n in spans]

    # -- data-parallel exchange (gqe_set_exchange) ---------------------------------------------
    def exchange_reserve(self, slab_entries):
        """Contribution entries per rank slab for the following margin calls (0: each call's own count, which
        then has to be equal on all ranks)."""
        self._check(self.lib.gqe_exchange_reserve(self.ctx, int(slab_entries)))

    def export_entries(self):
        """Pack this rank's slab of the pending margin call and return (S, slabs float32[world*S, dim]) — a view of
        the workspace's entry space; this rank's slab is rows [rank*S, (rank+1)*S) (include/gqe.h)."""
        S, c_off = C.c_int64(), C.c_int64()
        self._check(self.lib.gqe_export_entries(self.ctx, C.byref(S), C.byref(c_off), self._stream()))
        S = int(S.value)
        base = _align(self.workspace.data_ptr(), 256) - self.workspace.data_ptr()
        cb = self.workspace[base + c_off.value: base + c_off.value + 4 * self.world * S * self.dim]
        return S, cb.view(self.torch.float32).view(self.world * S, self.dim)

    def import_entries(self, n):
        self._check(self.lib.gqe_import_entries(self.ctx, int(n), self._stream()))

    def close(self):
        if getattr(self, "ctx", None):
            self.torch.cuda.synchronize(self.device)
            self.lib.gqe_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- batches ----------------------------------------------------------------
    @staticmethod
    def make_batches(descs):
        arr = (gqe_batch * len(descs))()
        for i, dsc in enumerate(descs):
            b = arr[i]
            b.qtype, b.n_queries, b.n_anchors, b.idx_offset = dsc["qtype"], dsc["n"], dsc["n_anchors"], dsc["idx_offset"]
            b.target_table = dsc["target_table"]
            for k in range(MAX_BRANCH):
                b.anchor_table[k] = dsc["anchor_table"][k] if k < len(dsc["anchor_table"]) else -1
                hops = dsc["hops"][k] if k < len(dsc["hops"]) else ()
                b.n_hops[k] = len(hops)
                for h in range(MAX_HOPS):
                    b.hop_param[k][h] = hops[h] if h < len(hops) else -1
            b.n_final = 1 if dsc.get("final", -1) >= 0 else 0
            b.final_param = dsc.get("final", -1)
            b.pre_param = dsc.get("pre", -1)
            b.post_param = dsc.get("post", -1)
            b.margin = dsc.get("margin", 1.0)
            b.loss_weight = dsc.get("weight", 1.0)
            b.out_offset = dsc["out_offset"]
            b.n_candidates = dsc.get("n_candidates", 0)
        return arr

    def _idx_arg(self, idx):
        """idx: torch int32 CUDA tensor (already resident) or a numpy int32 array (host feed)."""
        if isinstance(idx, np.ndarray):
            if idx.dtype != np.int32 or not idx.flags["C_CONTIGUOUS"]:
                idx = np.ascontiguousarray(idx, dtype=np.int32)
            return idx, C.c_void_p(idx.ctypes.data), int(idx.size), 0
        assert idx.dtype == self.torch.int32 and idx.is_cuda and idx.is_contiguous()
        return idx, C.c_void_p(idx.data_ptr()), int(idx.numel()), 1

    def forward(self, descs, idx, n_scores, out=None):
        """gqe_forward: scores[n_scores] for the listed batches."""
        total = sum(dsc["n"] for dsc in descs)
        n_all = int(idx.size if isinstance(idx, np.ndarray) else idx.numel())
        self.reserve(max(total, (n_all + 4) // 5), len(descs))      # candidate lists can dwarf the query count
        arr = self.make_batches(descs)
        keep, ptr, n_idx, on_dev = self._idx_arg(idx)
        scores = out if out is not None else self.torch.empty(n_scores, dtype=self.torch.float32, device=self.device)
        self._check(self.lib.gqe_forward(self.ctx, arr, len(descs), ptr, n_idx, on_dev, scores.data_ptr(), self._stream()))
        return scores

    # -- ranking statistics on the device (only query-level numbers are read back) ---------------
    def rank_candidates(self, scores, ptr):
        """Percentile (scipy percentileofscore, kind 'rank') of each list's FIRST score among the list's others;
        ``scores`` device float32 (flat, candidate order), ``ptr`` int32[n+1] (numpy or device).  Returns a device tensor[n]."""
        t = self.torch
        if isinstance(ptr, np.ndarray):
            ptr = t.from_numpy(np.ascontiguousarray(ptr, dtype=np.int32)).to(self.device)
        n = int(ptr.numel()) - 1
        out = t.empty(n, dtype=t.float64, device=self.device)
        self._check(self.lib.gqe_rank_candidates(self.ctx, scores.data_ptr(), ptr.data_ptr(), n, out.data_ptr(), self._stream()))
        return out

    def auc(self, pos, neg):
        """ROC AUC of positive vs negative scores (device tensors), ties count 1/2, NaN reads as 0 — sklearn's
        roc_auc_score on np.nan_to_num(scores) (utils.py:63,66).  One 8-byte read-back."""
        t = self.torch
        if pos.numel() == 0 or neg.numel() == 0:
            raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
        pos, neg = pos.contiguous(), neg.contiguous()
        count = t.zeros(1, dtype=t.int64, device=self.device)
        self._check(self.lib.gqe_auc_pair_counts(self.ctx, pos.data_ptr(), int(pos.numel()), neg.data_ptr(), int(neg.numel()),
                                                 count.data_ptr(), self._stream()))
        return float(count.item()) / (2.0 * float(pos.numel()) * float(neg.numel()))

    def auc_pair_count(self, pos, neg, counts, slot):
        """The pair count behind ``auc`` written to counts[slot] (device int64) — no read-back: an evaluation over hundreds of
        formulas reads all of its AUCs back at once (utils.eval_auc_queries).  AUC = count / (2 |pos| |neg|)."""
        if pos.numel() == 0 or neg.numel() == 0:
            raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
        pos, neg = pos.contiguous(), neg.contiguous()
        self._check(self.lib.gqe_auc_pair_counts(self.ctx, pos.data_ptr(), int(pos.numel()), neg.data_ptr(), int(neg.numel()),
                                                 counts.data_ptr() + 8 * int(slot), self._stream()))
        return 2.0 * float(pos.numel()) * float(neg.numel())

    def margin_fwd_bwd(self, descs, idx, n_scores=0, want_scores=False, losses=None):
        """gqe_margin_fwd_bwd: grads += d(sum_i w_i loss_i); returns (losses[n+1], pos, neg)."""
        total = sum(dsc["n"] for dsc in descs)
        self.reserve(total, len(descs))
        arr = self.make_batches(descs)
        keep, ptr, n_idx, on_dev = self._idx_arg(idx)
        t = self.torch
        if losses is None:
            losses = t.empty(len(descs) + 1, dtype=t.float32, device=self.device)
        pos = neg = None
        pp = pn = None
        if want_scores:
            pos = t.empty(n_scores, dtype=t.float32, device=self.device)
            neg = t.empty(n_scores, dtype=t.float32, device=self.device)
            pp, pn = pos.data_ptr(), neg.data_ptr()
        self._check(self.lib.gqe_margin_fwd_bwd(self.ctx, arr, len(descs), ptr, n_idx, on_dev, losses.data_ptr(),
                                                pp, pn, self._stream()))
        # gqe_set_deferred_gemm: the library writes ``losses`` only when the next call is enqueued — the buffer has to outlive a
        # caller that drops it (torch's caching allocator would hand the block to somebody else)
        self._held_losses = losses
        return losses, pos, neg

    # -- pre-packed steps (lowest host overhead: bench / steady-state trainer) ------
    def prepare_margin(self, descs, idx_dev):
        """Freeze one iteration's batches: ctypes descriptors + HBM-resident index feed."""
        total = sum(dsc["n"] for dsc in descs)
        self.reserve(total, len(descs))
        t = self.torch
        assert idx_dev.dtype == t.int32 and idx_dev.is_cuda and idx_dev.is_contiguous()
        return {"arr": self.make_batches(descs), "n": len(descs), "idx": idx_dev,
                "idx_ptr": C.c_void_p(idx_dev.data_ptr()), "n_idx": int(idx_dev.numel()),
                "losses": t.zeros(len(descs) + 1, dtype=t.float32, device=self.device), "queries": total}

    def run_margin(self, ps, stream=None):
        self._check(self.lib.gqe_margin_fwd_bwd(self.ctx, ps["arr"], ps["n"], ps["idx_ptr"], ps["n_idx"], 1,
                                                ps["losses"].data_ptr(), None, None,
                                                stream if stream is not None else self._stream()))

    def lazy_prefetch(self, ps, with_negatives=True):
        """Lazy Adam: name the prepared batch the NEXT margin / forward call will run (include/gqe.h,
        gqe_lazy_prefetch) — call between run_margin and run_adam; the step's row launch then also catches up that
        batch's rows and the next call skips its own catch-up launch.  No-op in eager mode."""
        if self.lazy_adam:
            self._check(self.lib.gqe_lazy_prefetch(self.ctx, ps["arr"], ps["n"], ps["idx_ptr"], ps["n_idx"], 1 if with_negatives else 0))

    def prepare_adam(self, keys):
        keys = [k for k in self.layout.entries if k in set(keys)]
        arr = self._segments(keys, False)
        for i in range(len(keys)):
            arr[i].step = 0
        return {"keys": keys, "arr": arr, "n": len(keys)}

    def run_adam(self, pa, lr=0.01, betas=(0.9, 0.999), eps=1e-8, stream=None):
        """step <= 0 in the prepared segments: libgqe keeps the per-tensor Adam step counters."""
        self._check(self.lib.gqe_adam_step(self.ctx, pa["arr"], pa["n"], lr, betas[0], betas[1], eps,
                                           stream if stream is not None else self._stream()))

    def run_train_step(self, ps, pa, lr=0.01, betas=(0.9, 0.999), eps=1e-8, stream=None):
        """gqe_train_step on a prepared margin batch + prepared Adam segments (library-kept step counters): one call per
        iteration — run_margin + run_adam with the step's work ordered by the library (include/gqe.h)."""
        self._check(self.lib.gqe_train_step(self.ctx, ps["arr"], ps["n"], ps["idx_ptr"], ps["n_idx"], 1, pa["arr"], pa["n"],
                                            lr, betas[0], betas[1], eps, ps["losses"].data_ptr(),
                                            stream if stream is not None else self._stream()))

    def train_step(self, descs, idx, keys, lr=0.01, betas=(0.9, 0.999), eps=1e-8, losses=None):
        """gqe_train_step: margin_fwd_bwd(descs, idx) + adam_step(keys) as one library call; returns losses[n + 1]."""
        total = sum(dsc["n"] for dsc in descs)
        self.reserve(total, len(descs))
        arr = self.make_batches(descs)
        keep, ptr, n_idx, on_dev = self._idx_arg(idx)
        t = self.torch
        if losses is None:
            losses = t.empty(len(descs) + 1, dtype=t.float32, device=self.device)
        keys = [k for k in self.layout.entries if k in set(keys)]   # arena order
        segs = self._segments(keys, True)
        try:
            self._check(self.lib.gqe_train_step(self.ctx, arr, len(descs), ptr, n_idx, on_dev, segs, len(keys), lr, betas[0], betas[1], eps,
                                                losses.data_ptr(), self._stream()))
        except Exception:
            for k in keys:                      # (a refused call stepped nothing: the counters stay with the library's)
                self.steps[k] -= 1
            raise
        self._held_losses = losses
        return losses

    # -- the reference's decoder / encoder extension points on [d, B] tensors (include/gqe.h) --------------------------
    def _embeds(self, x, what):
        t = self.torch
        if not (isinstance(x, t.Tensor) and x.dim() == 2 and x.shape[0] == self.dim):
            raise Exception("%s: expected a [%d, B] tensor" % (what, self.dim))
        return x.to(device=self.device, dtype=t.float32).contiguous()

    def encode_rows(self, table_key, rows):
        """DirectEncoder.forward: the L2-normalised table rows as columns, [d, B] (gqe_encode_rows)."""
        t = self.torch
        rows = t.as_tensor(np.ascontiguousarray(rows, dtype=np.int32)).to(self.device)
        out = t.empty((self.dim, len(rows)), dtype=t.float32, device=self.device)
        self.params_changed()
        self._check(self.lib.gqe_encode_rows(self.ctx, self.layout.offset(table_key), rows.data_ptr(), len(rows), out.data_ptr(), self._stream()))
        return out

    def decoder_project(self, rel_key, embeds):
        e = self._embeds(embeds, "project")
        out = self.torch.empty_like(e)
        self.params_changed()
        self._check(self.lib.gqe_decoder_project(self.ctx, self.layout.offset(rel_key), e.data_ptr(), e.shape[1], out.data_ptr(), self._stream()))
        return out

    def decoder_forward(self, rel_keys, embeds1, embeds2):
        e1, e2 = self._embeds(embeds1, "forward"), self._embeds(embeds2, "forward")
        if e1.shape != e2.shape:
            raise Exception("forward: embeds1 and embeds2 differ in shape")
        offs = (C.c_int64 * max(len(rel_keys), 1))(*[self.layout.offset(k) for k in rel_keys])
        out = self.torch.empty(e1.shape[1], dtype=self.torch.float32, device=self.device)
        self.params_changed()
        self._check(self.lib.gqe_decoder_forward(self.ctx, offs, len(rel_keys), e1.data_ptr(), e2.data_ptr(), e1.shape[1], out.data_ptr(), self._stream()))
        return out

    def set_intersection(self, pre_key, post_key, embeds1, embeds2, embeds3=None):
        es = [self._embeds(x, "intersection") for x in (embeds1, embeds2) + ((embeds3,) if embeds3 is not None else ())]
        if any(e.shape != es[0].shape for e in es):
            raise Exception("intersection: the embedding batches differ in shape")
        out = self.torch.empty_like(es[0])
        self.params_changed()
        self._check(self.lib.gqe_set_intersection(self.ctx, -1 if pre_key is None else self.layout.offset(pre_key),
                                                  -1 if post_key is None else self.layout.offset(post_key), es[0].data_ptr(), es[1].data_ptr(),
                                                  es[2].data_ptr() if len(es) == 3 else None, es[0].shape[1], out.data_ptr(), self._stream()))
        return out

    def split_steps(self):
        """How many train_step calls ran as split steps so far (gqe_split_steps)."""
        return int(self.lib.gqe_split_steps(self.ctx))

    def materialize_tables(self, keys):
        """gqe_materialize_tables: fold the pending gradient lists of the listed tables only into the dense arena."""
        offs = (C.c_int64 * len(keys))(*[self.layout.offset(k) for k in keys])
        self._check(self.lib.gqe_materialize_tables(self.ctx, offs, len(keys), self._stream()))

    def allreduce_grads(self, nccl_comm):
        """gqe_allreduce_grads: lists -> dense gradient arena, then ncclAllReduce (RCCL) over ``nccl_comm`` (an
        ncclComm_t as an integer / c_void_p, e.g. from parallel.RcclComm)."""
        self._check(self.lib.gqe_allreduce_grads(self.ctx, C.c_void_p(int(nccl_comm)), self._stream()))

    def hot_rows(self):
        """Rows promoted to dense gradient accumulators so far (include/gqe.h, gqe_hot_rows); synchronises."""
        n = C.c_int32(0)
        self._check(self.lib.gqe_hot_rows(self.ctx, C.byref(n)))
        return int(n.value)

    def hot_sub_lists(self):
        """(sub-list heads handed out to hot word rows so far, whether the fused launches link onto them yet) — include/gqe.h,
        gqe_hot_sub_lists; synchronises."""
        n, on = C.c_int32(0), C.c_int32(0)
        self._check(self.lib.gqe_hot_sub_lists(self.ctx, C.byref(n), C.byref(on)))
        return int(n.value), bool(on.value)

    def materialize(self):
        """Fold pending per-row gradient lists into the dense gradient arena (gqe_materialize_grads)."""
        self._check(self.lib.gqe_materialize_grads(self.ctx, self._stream()))

    # -- optimiser --------------------------------------------------------------
    def _segments(self, keys, bump):
        arr = (gqe_segment * len(keys))()
        for i, k in enumerate(keys):
            if bump:
                self.steps[k] += 1
            arr[i].offset = self.layout.offset(k)
            arr[i].numel = self.layout.numel(k)
            arr[i].step = max(self.steps[k], 1)
        return arr

    def adam_step(self, keys, lr=0.01, betas=(0.9, 0.999), eps=1e-8):
        keys = [k for k in self.layout.entries if k in set(keys)]   # arena order
        arr = self._segments(keys, True)
        try:
            self._check(self.lib.gqe_adam_step(self.ctx, arr, len(keys), lr, betas[0], betas[1], eps, self._stream()))
        except Exception:
            for k in keys:
                self.steps[k] -= 1
            raise

    def sgd_step(self, keys, lr=0.01):
        keys = [k for k in self.layout.entries if k in set(keys)]
        arr = self._segments(keys, False)
        self._check(self.lib.gqe_sgd_step(self.ctx, arr, len(keys), lr, self._stream()))

    def zero_grads(self, keys):
        keys = [k for k in self.layout.entries if k in set(keys)]
        if not keys:
            return
        arr = self._segments(keys, False)
        self._check(self.lib.gqe_zero_grads(self.ctx, arr, len(keys), self._stream()))

    # -- native training feed ---------------------------------------------------------
    def make_feeder(self, pools_by_plan, mode_rows, batch_size=512, path_weight=0.01, inter_weight=0.005, seed=0, feed="zero-copy"):
        """pools_by_plan: [(FormulaPlan, pool)] with pool.target[n], pool.anchors[k,n], pool.neg[n] or None,
        pool.hard[n] or None (int32 rows); mode_rows: {table key: int32 rows} for 1-chain negatives."""
        h = _P()
        self._check(self.lib.gqe_feeder_create(self.ctx, seed, batch_size, path_weight, inter_weight, C.byref(h)))
        # "zero-copy": the kernels read each iteration's feed from pinned host memory; "copy": pinned staging + hipMemcpyAsync
        self._check(self.lib.gqe_feeder_set_feed(h, {"copy": 0, "zero-copy": 1}[feed]))
        for plan, pool in pools_by_plan:
            arr = self.make_batches([plan.batch(1, 0, 0)])
            c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32)
            t, a, ng, hd = c(pool.target), c(pool.anchors), c(getattr(pool, "neg", None)), c(getattr(pool, "hard", None))
            p = lambda x: None if x is None else C.c_void_p(x.ctypes.data)
            self._check(self.lib.gqe_feeder_add_pool(h, arr, len(t), p(t), p(a), p(ng), p(hd)))
        for key, rows in mode_rows.items():
            rows = np.ascontiguousarray(rows, dtype=np.int32)
            self._check(self.lib.gqe_feeder_set_mode_rows(h, self.layout.offset(key), C.c_void_p(rows.ctypes.data), len(rows)))
        return h

    def feeder_run(self, feeder, first_iteration, n_iterations, burn_in=0, lr=0.01, betas=(0.9, 0.999), eps=1e-8, losses=None):
        if losses is None:
            losses = self.torch.zeros(MAX_BATCHES + 1, dtype=self.torch.float32, device=self.device)
        self._check(self.lib.gqe_feeder_run(feeder, first_iteration, n_iterations, burn_in, lr, betas[0], betas[1], eps,
                                            losses.data_ptr(), self._stream()))
        return losses

    def feeder_destroy(self, feeder):
        self.lib.gqe_feeder_destroy(feeder)

    LOSS_STRIDE = 32    # floats per iteration of a reference feeder's loss history (>= GQE_LAUNCH_BATCHES + 1)

    def make_reference_feeder(self, pools_by_type, mode_rows, batch_size, path_weight, inter_weight, feed="copy", sgd=False):
        """The reference's own training loop, natively (include/gqe.h "reference streams"; train_helpers.run_train):
        pools_by_type = {query type name: [(FormulaPlan, target[n], anchors[k, n], (neg_ptr, neg_rows) | None,
        (hard_ptr, hard_rows) | None)]} in the order of the training dictionary (1-chain pools carry no lists),
        mode_rows = {table key: rows of graph.full_lists[mode]}.  The probability vector of a type is computed here exactly as
        train_helpers.py:96-98 does."""
        h = _P()
        self._check(self.lib.gqe_feeder_create(self.ctx, 0, batch_size, path_weight, inter_weight, C.byref(h)))
        try:
            self._check(self.lib.gqe_feeder_set_feed(h, {"copy": 0, "zero-copy": 1}[feed]))
            self._check(self.lib.gqe_feeder_set_loss_stride(h, self.LOSS_STRIDE))
            self._check(self.lib.gqe_feeder_set_sgd(h, 1 if sgd else 0))       # (--opt sgd: the iterations close with gqe_sgd_step)
            i32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32)
            i64 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int64)
            ptr = lambda x: None if x is None else C.c_void_p(x.ctypes.data)
            order = []
            for qt, pools in pools_by_type.items():
                if qt != "1-chain":
                    order.append(QTYPES[qt])
                for plan, target, anchors, neg, hard in pools:
                    arr = self.make_batches([plan.batch(1, 0, 0)])
                    t, a = i32(target), i32(anchors)
                    np_, nr = (i64(neg[0]), i32(neg[1])) if neg is not None else (None, None)
                    hp, hr = (i64(hard[0]), i32(hard[1])) if hard is not None else (None, None)
                    self._check(self.lib.gqe_feeder_add_pool_lists(h, arr, len(t), ptr(t), ptr(a), ptr(np_), ptr(nr), ptr(hp), ptr(hr)))
                sizes = [float(len(p[1])) for p in pools]
                pv = np.ascontiguousarray(np.array(sizes) / float(sum(sizes)), dtype=np.float64)
                self._check(self.lib.gqe_feeder_set_pvals(h, QTYPES[qt], C.c_void_p(pv.ctypes.data), len(pv)))
            od = np.ascontiguousarray(order, dtype=np.int32)
            self._check(self.lib.gqe_feeder_set_type_order(h, C.c_void_p(od.ctypes.data), len(od)))
            for key, rows in mode_rows.items():
                rows = np.ascontiguousarray(rows, dtype=np.int32)
                self._check(self.lib.gqe_feeder_set_mode_rows(h, self.layout.offset(key), C.c_void_p(rows.ctypes.data), len(rows)))
        except Exception:
            self.lib.gqe_feeder_destroy(h)
            raise
        return h

    def reference_feeder_run(self, feeder, np_state, py_state, first_iteration, n_iterations, edges_only, lr, betas, eps):
        """n_iterations of the reference's loop on the two generator states (625 uint32 words each, updated in place); returns the
        loss history [n_iterations, LOSS_STRIDE] (device tensor; row i, column <number of batches of iteration i> = its total
        loss).  The per-tensor Adam step counters of this engine follow the library's afterwards."""
        t = self.torch
        hist = t.zeros((n_iterations, self.LOSS_STRIDE), dtype=t.float32, device=self.device)
        self._check(self.lib.gqe_feeder_set_reference_streams(feeder, C.c_void_p(np_state.ctypes.data), C.c_void_p(py_state.ctypes.data)))
        try:
            burn_in = 0x7fffffff if edges_only else 0      # (a constant: the feeder starts from a clean ring when it changes)
            self._check(self.lib.gqe_feeder_run(feeder, first_iteration, n_iterations, burn_in, lr, betas[0], betas[1], eps,
                                                hist.data_ptr(), self._stream()))
        finally:
            self.lib.gqe_feeder_set_reference_streams(feeder, None, None)
            self.sync_step_counts()
        return hist

    def feeder_queries(self, feeder):
        return int(self.lib.gqe_feeder_queries(feeder))

    def feeder_host_seconds(self, feeder):
        """(seconds spent sampling + packing, seconds inside gqe_feeder_run) of this feeder so far (gqe_feeder_host_seconds)."""
        a, b = C.c_double(0), C.c_double(0)
        self._check(self.lib.gqe_feeder_host_seconds(feeder, C.byref(a), C.byref(b)))
        return a.value, b.value

    def feeder_debug_feed(self, feeder, iteration):
        """(batches, idx) of one of the feeder's last prepared iterations: [(qtype, n_queries, n_anchors, idx_offset, loss_weight)]
        and the packed int32 feed (tests)."""
        nb, ni = C.c_int32(0), C.c_int64(0)
        self._check(self.lib.gqe_feeder_debug_feed(feeder, iteration, None, 0, C.byref(nb), None, 0, C.byref(ni)))
        arr = (gqe_batch * nb.value)()
        idx = np.empty(ni.value, dtype=np.int32)
        self._check(self.lib.gqe_feeder_debug_feed(feeder, iteration, arr, nb.value, C.byref(nb), C.c_void_p(idx.ctypes.data), ni.value, C.byref(ni)))
        return [(b.qtype, b.n_queries, b.n_anchors, b.idx_offset, b.loss_weight) for b in arr], idx

    def sync_step_counts(self):
        """This engine's per-tensor Adam step counters := the library's (gqe_adam_step_count), behind a native run."""
        c = C.c_int32(0)
        for k in self.layout.entries:
            self._check(self.lib.gqe_adam_step_count(self.ctx, self.layout.offset(k), C.byref(c)))
            if c.value > self.steps[k]:
                self.steps[k] = int(c.value)

    def push_step_counts(self):
        """The library's per-tensor Adam step counters := this engine's (gqe_set_adam_step_count): behind a restored optimiser
        checkpoint, so that native runs / library-counted steps continue the bias correction where the checkpoint was."""
        self.sync()
        for k in self.layout.entries:
            self._check(self.lib.gqe_set_adam_step_count(self.ctx, self.layout.offset(k), int(self.steps[k])))

    # -- timing (bench.py roofline) ------------------------------------------------
    def timing_enable(self, stride):
        """Record hipEvents around every ``stride``-th launch of each kernel (0 / False = off)."""
        self._check(self.lib.gqe_timing_enable(self.ctx, int(stride)))

    def timing_read(self, kernel):
        ms, n = C.c_float(0), C.c_int32(0)
        self._check(self.lib.gqe_timing_read(self.ctx, kernel, C.byref(ms), C.byref(n)))
        return ms.value, n.value
