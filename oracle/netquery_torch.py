"""ORACLE / CPU BASELINE (test infrastructure, not product code) — torch-CPU restatement of
the reference's per-iteration work, op for op: what the reference's own CPU path executes.

Used by ``bench.py``'s ``cpu_baseline`` leg (kind "port") and by tests; never by the
product path.  Pinned against the same golden vectors as the numpy oracle
(tests/test_oracle_golden.py::test_torch_port_*).

Mirrors (file:line of the reference):
  encode        encoders.py:40-43    features(nodes).t() ; / column L2 norm
  decoders      decoders.py:142-150 (bilinear: act.mm(M) / M.mm(e)), 200-208 (transe),
                228-236 (bilinear-diag)
  intersections decoders.py:288-300, 311-319 (relu(Pre.mm(e)) stacked, torch.min/mean, Post.mm)
  forward       model.py:70-109 ;  margin loss model.py:122-126 (TWO forwards, anchors recomputed)
  iteration     train_helpers.py:50-79: zero_grad, weighted sum of batch losses, ONE backward
                (dense [N,d] embedding grads), ONE dense torch.optim.Adam step
Vectors are [d, B] columns as in the reference.  Index inputs are table rows (the Python
dict lookup per node of bio/data_utils.py:20-21 is NOT reproduced, which favours this baseline).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .netquery_numpy import BAGS_KEY, make_plan, pre_key, post_key, rel_key, table_key, CHAIN_TYPES


class TorchPort(object):
    def __init__(self, params, dec, inter, lr=0.01):
        self.dec, self.inter = dec, inter
        self.bags = params.get(BAGS_KEY) or {}
        self.p = {k: torch.nn.Parameter(torch.from_numpy(np.array(v, dtype=np.float32))) for k, v in params.items() if k != BAGS_KEY}
        self.cos = torch.nn.CosineSimilarity(dim=0)
        self.opt = torch.optim.Adam(list(self.p.values()), lr=lr)

    def enc(self, mode, rows):
        if mode in self.bags:   # nn.EmbeddingBag(mode='mean') over the bag's word ids (reddit/data_utils_new.py:155,167-169)
            ptr, ids = self.bags[mode]
            parts = [ids[ptr[r]:ptr[r + 1]] for r in np.asarray(rows)]
            offsets = np.concatenate(([0], np.cumsum([len(x) for x in parts[:-1]])))
            e = F.embedding_bag(torch.as_tensor(np.concatenate(parts), dtype=torch.long), self.p[table_key(mode)],
                                torch.as_tensor(offsets, dtype=torch.long), mode="mean").t()
        else:
            e = F.embedding(torch.as_tensor(np.asarray(rows), dtype=torch.long), self.p[table_key(mode)]).t()
        return e.div(e.norm(p=2, dim=0, keepdim=True).expand_as(e))

    def project(self, e, rel):
        w = self.p[rel_key(rel)]
        if self.dec == "bilinear":
            return w.mm(e)
        if self.dec == "transe":
            return e + w.unsqueeze(1).expand(w.size(0), e.size(1))
        return e * w.unsqueeze(1).expand(w.size(0), e.size(1))

    def chain_score(self, t, a, rels):
        if self.dec == "bilinear":
            act = t.t()
            for r in rels:
                act = act.mm(self.p[rel_key(r)])
            return self.cos(act.t(), a)
        if self.dec == "transe":
            u = t
            for r in rels:
                w = self.p[rel_key(r)]
                u = u + w.unsqueeze(1).expand(w.size(0), t.size(1))
            return self.cos(a, u)
        acts = t
        for r in rels:
            w = self.p[rel_key(r)]
            acts = acts * w.unsqueeze(1).expand(w.size(0), t.size(1))
        return (acts * a).sum(0)

    def intersect(self, es, mode):
        agg = torch.min if self.inter.startswith("min") else torch.mean
        if not self.inter.endswith("simple"):
            pre, post = self.p[pre_key(mode)], self.p[post_key(mode)]
            es = [F.relu(pre.mm(e)) for e in es]
        comb = agg(torch.stack(es), dim=0)
        if isinstance(comb, tuple):
            comb = comb[0]
        return comb if self.inter.endswith("simple") else post.mm(comb)

    def forward(self, plan, target_rows, anchor_rows):
        t = self.enc(plan["target_mode"], target_rows)
        if plan["type"] in CHAIN_TYPES:
            return self.chain_score(t, self.enc(plan["anchor_modes"][0], anchor_rows[0]), plan["chain"])
        es = []
        for i, br in enumerate(plan["branches"]):
            e = self.enc(plan["anchor_modes"][i], anchor_rows[i])
            for r in br:
                e = self.project(e, r)
            es.append(e)
        q = self.intersect(es, plan["inter_mode"])
        for r in plan["final"]:
            q = self.project(q, r)
        return self.cos(t, q)

    def margin_loss(self, plan, target_rows, neg_rows, anchor_rows, margin=1.0):
        affs = self.forward(plan, target_rows, anchor_rows)
        neg_affs = self.forward(plan, neg_rows, anchor_rows)
        return torch.clamp(margin - (affs - neg_affs), min=0).mean()

    def train_iteration(self, items):
        """items: [(plan, target, neg, anchors, weight, margin)] -> iteration loss (float)."""
        self.opt.zero_grad()
        loss = None
        for plan, t, g, a, w, m in items:
            l = w * self.margin_loss(plan, t, g, a, m)
            loss = l if loss is None else loss + l
        value = loss.item()
        loss.backward()
        self.opt.step()
        return value

    def grads(self):
        return {k: (None if v.grad is None else v.grad.detach().numpy()) for k, v in self.p.items()}

    def state(self):
        return {k: v.detach().numpy() for k, v in self.p.items()}
