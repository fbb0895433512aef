"""Relation ("metapath") decoders and set-intersection operators.

These classes own the parameters — same constructor arguments, parameter names,
shapes and initialisers as the reference, so ``state_dict`` interchanges with it:
  BilinearMetapathDecoder      netquery/decoders.py:123-150   M_r [d,d], xavier-uniform
  TransEMetapathDecoder        netquery/decoders.py:181-208   w_r [d],  U(+-6/sqrt(d))
  BilinearDiagMetapathDecoder  netquery/decoders.py:211-236   w_r [d],  U(+-6/sqrt(d))
  SetIntersection              netquery/decoders.py:270-300   Pre_m/Post_m [d,d] per mode
  SimpleSetIntersection        netquery/decoders.py:302-319   no parameters
The arithmetic (projection chains, relu/min/mean, Pre/Post contractions) runs inside the
fused HIP kernel; ``kind`` tells the engine which specialisation to launch.  The reference's
per-decoder entry points (``forward`` / ``project`` / the intersection's ``forward`` on [d, B]
tensors) are served by small forward-only HIP launches once the decoder belongs to a model.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
from torch.nn import init


def directed_relations(relations):
    """``{mode: [(to_mode, name), ...]}`` -> (mode, name, to_mode) triples, reference order."""
    for m1 in relations:
        for (m2, name) in relations[m1]:
            yield (m1, name, m2)


class _MetapathDecoder(nn.Module):
    kind = None

    def __init__(self, relations, dims):
        super(_MetapathDecoder, self).__init__()
        self.relations = relations
        self.rels = []
        for rel in directed_relations(relations):
            self.register_parameter("_".join(rel), nn.Parameter(self._new(dims, rel)))
            self.rels.append(rel)

    @staticmethod
    def param_name(rel):
        return "_".join(rel)

    # The reference's extension point (decoders.py:142-150, 200-208, 228-236) on [d, B] tensors.  QueryEncoderDecoder never calls
    # it — queries run through the fused kernel — but a caller that scores one hop on its own gets the same arithmetic from small
    # forward-only HIP launches (gqe_decoder_forward / gqe_decoder_project, include/gqe.h) once the decoder belongs to a model
    # (the model's engine owns the parameters).  No gradient flows through these calls.
    _engine = None            # set by QueryEncoderDecoder; (key prefix, engine)

    def _attached(self):
        if self._engine is None:
            raise NotImplementedError("this decoder is not part of a QueryEncoderDecoder yet: its arithmetic runs on the model's "
                                      "HIP engine (there is no torch fallback)")
        return self._engine

    def forward(self, embeds1, embeds2, rels):
        prefix, eng = self._attached()
        return eng.decoder_forward([prefix + self.param_name(r) for r in rels], embeds1, embeds2)

    def project(self, embeds, rel):
        prefix, eng = self._attached()
        return eng.decoder_project(prefix + self.param_name(rel), embeds)


class BilinearMetapathDecoder(_MetapathDecoder):
    kind = "bilinear"

    @staticmethod
    def _new(dims, rel):
        t = torch.empty(dims[rel[0]], dims[rel[2]])
        init.xavier_uniform_(t)
        return t


class _VecDecoder(_MetapathDecoder):
    @staticmethod
    def _new(dims, rel):
        d = dims[rel[0]]
        t = torch.empty(d)
        init.uniform_(t, a=-6.0 / np.sqrt(d), b=6.0 / np.sqrt(d))
        return t


class TransEMetapathDecoder(_VecDecoder):
    kind = "transe"


class BilinearDiagMetapathDecoder(_VecDecoder):
    kind = "bilinear-diag"


class SetIntersection(nn.Module):
    """``Post_m . agg_i relu(Pre_m . e_i)`` with agg = torch.min / torch.mean."""

    def __init__(self, mode_dims, expand_dims, agg_func=torch.min):
        super(SetIntersection, self).__init__()
        if agg_func not in (torch.min, torch.mean):
            raise Exception("Intersection decoder not recognized.")
        self.agg_func = agg_func
        self.kind = "min" if agg_func is torch.min else "mean"
        for mode in mode_dims:
            if expand_dims[mode] != mode_dims[mode]:
                raise Exception("the fused path needs expand_dims == mode_dims "
                                "(what utils.get_intersection_decoder passes)")
            pre = torch.empty(expand_dims[mode], mode_dims[mode])
            init.xavier_uniform_(pre)
            self.register_parameter(mode + "_premat", nn.Parameter(pre))
            post = torch.empty(mode_dims[mode], expand_dims[mode])
            init.xavier_uniform_(post)
            self.register_parameter(mode + "_postmat", nn.Parameter(post))

    _engine = None            # set by QueryEncoderDecoder; (key prefix, engine)

    def forward(self, embeds1, embeds2, mode, embeds3=[]):
        """decoders.py:288-300 on [d, B] tensors, as a small forward-only HIP launch (gqe_set_intersection)."""
        if self._engine is None:
            raise NotImplementedError("this intersection decoder is not part of a QueryEncoderDecoder yet (no torch fallback)")
        prefix, eng = self._engine
        return eng.set_intersection(prefix + mode + "_premat", prefix + mode + "_postmat", embeds1, embeds2, embeds3 if len(embeds3) > 0 else None)


class SimpleSetIntersection(nn.Module):
    """Element-wise min / mean over the branches."""

    def __init__(self, agg_func=torch.min):
        super(SimpleSetIntersection, self).__init__()
        if agg_func not in (torch.min, torch.mean):
            raise Exception("Intersection decoder not recognized.")
        self.agg_func = agg_func
        self.kind = "min-simple" if agg_func is torch.min else "mean-simple"

    _engine = None

    def forward(self, embeds1, embeds2, mode, embeds3=[]):
        """decoders.py:311-319 on [d, B] tensors (gqe_set_intersection without Pre / Post)."""
        if self._engine is None:
            raise NotImplementedError("this intersection decoder is not part of a QueryEncoderDecoder yet (no torch fallback)")
        return self._engine[1].set_intersection(None, None, embeds1, embeds2, embeds3 if len(embeds3) > 0 else None)
