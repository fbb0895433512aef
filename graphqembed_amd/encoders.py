"""Node encoders.  Only the depth-0 ``DirectEncoder`` is on the accelerated path
(netquery/encoders.py:11-45): an embedding-table lookup followed by
a per-vector L2 normalisation.  Here the class is the *owner of the tables* and of the
node-id -> table-row mapping; the gather + normalise themselves happen inside the fused
HIP kernel (graphqembed_amd/csrc/gqe_kernels.hip: ``gather_norm``).
"""
from __future__ import annotations

import numpy as np
import torch.nn as nn


class DirectEncoder(nn.Module):
    """``DirectEncoder(features, feature_modules)`` as in the reference; tables are
    registered as ``feat-<mode>`` so state_dict keys stay ``enc.feat-<mode>.weight``.

    ``node_maps`` ({mode: {node_id: index}}, table row = index + 1 as in
    netquery/bio/data_utils.py:20-21) replaces what the reference hides inside the
    ``features`` closure; without it rows are ``node_id + 1`` (utils.py:19-20).
    """

    def __init__(self, features, feature_modules, node_maps=None, bags=None):
        """``bags``: {mode: {node_id: sequence of row ids}} for modes whose feature module is an
        ``nn.EmbeddingBag`` (Reddit posts = mean of word rows, reddit/data_utils_new.py:155,162-169);
        what the reference keeps in its ``post_words`` dict."""
        super(DirectEncoder, self).__init__()
        for name, module in feature_modules.items():
            self.add_module("feat-" + name, module)
        self.features = features
        self.modes = list(feature_modules.keys())
        self.node_maps = node_maps
        self._lut = {}
        self._flat_lut = {}
        self.bag_csr = {}       # mode -> (ptr int32[n+1], ids int32[nnz]); a node's "row" is its bag index
        self._bag_index = {}
        for mode, module in feature_modules.items():
            if isinstance(module, nn.EmbeddingBag):
                if bags is None or mode not in bags:
                    raise Exception("mode %r is an EmbeddingBag: pass bags={%r: {node: row ids}}" % (mode, mode))
                if module.mode != "mean":
                    raise Exception("only EmbeddingBag(mode='mean') is supported (the reference's default)")
                nodes = list(bags[mode].keys())
                lens = [len(bags[mode][n]) for n in nodes]
                if min(lens) < 1:
                    raise Exception("empty bag in mode %r" % mode)
                ptr = np.zeros(len(nodes) + 1, dtype=np.int32)
                ptr[1:] = np.cumsum(lens)
                ids = np.concatenate([np.asarray(bags[mode][n], dtype=np.int32) for n in nodes])
                self.bag_csr[mode] = (ptr, ids)
                self._bag_index[mode] = {n: i for i, n in enumerate(nodes)}

    def table(self, mode):
        return getattr(self, "feat-" + mode)

    def _lookup(self, mode):
        lut = self._lut.get(mode)
        if lut is None:
            nm = self.node_maps[mode]
            ids = np.fromiter((k for k in nm.keys() if k >= 0), dtype=np.int64)
            lo, hi = (int(ids.min()), int(ids.max())) if len(ids) else (0, 0)
            arr = np.full(hi - lo + 1, -1, dtype=np.int32)
            for k, v in nm.items():
                if k >= 0:
                    arr[k - lo] = v + 1
            lut = self._lut[mode] = (lo, arr)
        return lut

    def rows(self, nodes, mode):
        """Vectorised node ids -> int32 table rows (node -1 -> the dummy row 0)."""
        if mode in self._bag_index:
            idx = self._bag_index[mode]
            return np.fromiter((idx[n] for n in nodes), dtype=np.int32, count=len(nodes))
        nodes = np.asarray(nodes, dtype=np.int64)
        if self.node_maps is None:
            return (nodes + 1).astype(np.int32)
        lo, arr = self._lookup(mode)
        out = np.zeros(len(nodes), dtype=np.int32)
        real = nodes >= 0
        if real.any():
            rel = nodes[real] - lo
            if (rel < 0).any() or (rel >= len(arr)).any():
                raise KeyError("node id outside mode %r" % mode)
            got = arr[rel]
            if (got < 0).any():
                raise KeyError("node id not in mode %r" % mode)
            out[real] = got
        return out

    def flat_rows(self, rows, mode):
        """Rows of the FLAT data convention (sampler / converted files: node_maps index + 1 in every mode) -> this encoder's table
        rows: the same for ordinary tables; for an EmbeddingBag mode the bag index of the node (``rows`` above: the order of the
        ``bags`` dictionary)."""
        rows = np.asarray(rows, dtype=np.int32)
        if mode not in self._bag_index:
            return rows
        lut = self._flat_lut.get(mode)
        if lut is None:
            if self.node_maps is None:
                raise Exception("flat query lists on the EmbeddingBag mode %r need node_maps" % mode)
            nm, idx = self.node_maps[mode], self._bag_index[mode]
            lut = np.full(max(nm.values()) + 2, -1, dtype=np.int32)
            for n, i in nm.items():
                if n >= 0 and n in idx:
                    lut[i + 1] = idx[n]
            self._flat_lut[mode] = lut
        out = lut[rows]
        if (out < 0).any():
            raise KeyError("a node of mode %r has no bag" % mode)
        return out

    _engine = None            # set by QueryEncoderDecoder; (key prefix, engine)

    def forward(self, nodes, mode, offset=None, **kwargs):
        """encoders.py:40-43: the L2-normalised feature rows of ``nodes`` as columns, [d, B] — a small forward-only HIP launch
        (gqe_encode_rows) once the encoder belongs to a model; inside queries the fused kernel does the gather itself."""
        if self._engine is None:
            raise NotImplementedError("DirectEncoder is evaluated on the model's HIP engine; build a QueryEncoderDecoder first")
        prefix, eng = self._engine
        return eng.encode_rows(prefix + "feat-%s.weight" % mode, self.rows(nodes, mode))
