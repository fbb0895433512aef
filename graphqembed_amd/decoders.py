"""Relation ("metapath") decoders and set-intersection operators.

These classes own the parameters — same constructor arguments, parameter names,
shapes and initialisers as the reference, so ``state_dict`` interchanges with it:
  BilinearMetapathDecoder      netquery/decoders.py:123-150   M_r [d,d], xavier-uniform
  TransEMetapathDecoder        netquery/decoders.py:181-208   w_r [d],  U(+-6/sqrt(d))
  BilinearDiagMetapathDecoder  netquery/decoders.py:211-236   w_r [d],  U(+-6/sqrt(d))
  SetIntersection              netquery/decoders.py:270-300   Pre_m/Post_m [d,d] per mode
  SimpleSetIntersection        netquery/decoders.py:302-319   no parameters
The arithmetic (projection chains, relu/min/mean, Pre/Post contractions) runs inside the
fused HIP kernel; ``kind`` tells the engine which specialisation to launch.
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

    def forward(self, embeds1, embeds2, rels):
        raise NotImplementedError("relation decoders are evaluated inside the fused HIP kernel")

    def project(self, embeds, rel):
        raise NotImplementedError("relation decoders are evaluated inside the fused HIP kernel")


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

    def forward(self, embeds1, embeds2, mode, embeds3=[]):
        raise NotImplementedError("set intersection is evaluated inside the fused HIP kernel")


class SimpleSetIntersection(nn.Module):
    """Element-wise min / mean over the branches."""

    def __init__(self, agg_func=torch.min):
        super(SimpleSetIntersection, self).__init__()
        if agg_func not in (torch.min, torch.mean):
            raise Exception("Intersection decoder not recognized.")
        self.agg_func = agg_func
        self.kind = "min-simple" if agg_func is torch.min else "mean-simple"

    def forward(self, embeds1, embeds2, mode, embeds3=[]):
        raise NotImplementedError("set intersection is evaluated inside the fused HIP kernel")
