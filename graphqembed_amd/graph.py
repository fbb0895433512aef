"""Host-side data model: relations, ``Formula``, ``Query`` and the heterogeneous ``Graph``.

These are the API objects of the reference that *stay* (SURVEY.md §8 row a10):
same class names, same public fields, same (de)serialisation tuple, so that
pickled query files and user code written against ``netquery.graph`` keep
working.  Everything here runs on the host cores; the only thing the GPU path
ever sees of these objects is their *tensorised* form (``tensorize.py``).

Reference behaviour followed (never copied):
  * ``_reverse_relation``      netquery/graph.py:4-5
  * ``Formula``                netquery/graph.py:11-36
  * ``Query``                  netquery/graph.py:38-100
  * ``Graph``                  netquery/graph.py:104-443 (container, negative
                               sets, query-subgraph sampler) and the sampler
                               invariants of 447-534.
"""
from __future__ import annotations

import random
from collections import defaultdict

CHAIN_TYPES = ("1-chain", "2-chain", "3-chain")
INTER_TYPES = ("2-inter", "3-inter", "3-inter_chain", "3-chain_inter")
QUERY_TYPES = CHAIN_TYPES + INTER_TYPES


def _reverse_relation(relation):
    """(m1, name, m2) -> (m2, name, m1)   [netquery/graph.py:4-5]"""
    return (relation[2], relation[1], relation[0])


def _reverse_edge(edge):
    """(u, rel, v) -> (v, rev(rel), u)   [netquery/graph.py:7-8]"""
    return (edge[2], _reverse_relation(edge[1]), edge[0])


class Formula(object):
    """A query *shape*: query type + (possibly nested) relation tuple.

    Fields (as in netquery/graph.py:13-24): ``query_type``, ``rels``,
    ``target_mode`` (= ``rels[0][0]``) and ``anchor_modes``.
    Hash/equality are on ``(query_type, rels)`` (graph.py:26-33).
    """

    __slots__ = ("query_type", "rels", "target_mode", "anchor_modes")

    def __init__(self, query_type, rels):
        if query_type not in QUERY_TYPES:
            raise ValueError("unknown query type %r" % (query_type,))
        self.query_type = query_type
        self.rels = rels
        self.target_mode = rels[0][0]
        if query_type in CHAIN_TYPES:
            modes = (rels[-1][-1],)
        elif query_type in ("2-inter", "3-inter"):
            modes = tuple(r[-1] for r in rels)
        elif query_type == "3-inter_chain":
            modes = (rels[0][-1], rels[1][-1][-1])
        else:  # 3-chain_inter
            modes = (rels[1][0][-1], rels[1][1][-1])
        self.anchor_modes = modes

    def _key(self):
        return (self.query_type, self.rels)

    def __hash__(self):
        return hash(self._key())

    def __eq__(self, other):
        return isinstance(other, Formula) and self._key() == other._key()

    def __ne__(self, other):
        return not self.__eq__(other)

    def __str__(self):
        return "%s: %s" % (self.query_type, self.rels)

    __repr__ = __str__


def _split_query_graph(query_graph):
    """query_graph -> (rels, anchor_nodes, target_node), shapes per graph.py:42-54."""
    qt = query_graph[0]
    if qt in CHAIN_TYPES:
        rels = tuple(query_graph[i][1] for i in range(1, len(query_graph)))
        anchors = (query_graph[-1][-1],)
    elif qt in ("2-inter", "3-inter"):
        rels = tuple(query_graph[i][1] for i in range(1, len(query_graph)))
        anchors = tuple(query_graph[i][-1] for i in range(1, len(query_graph)))
    elif qt == "3-inter_chain":
        rels = (query_graph[1][1], (query_graph[2][0][1], query_graph[2][1][1]))
        anchors = (query_graph[1][-1], query_graph[2][-1][-1])
    elif qt == "3-chain_inter":
        rels = (query_graph[1][1], (query_graph[2][0][1], query_graph[2][1][1]))
        anchors = (query_graph[2][0][-1], query_graph[2][1][-1])
    else:
        raise ValueError("unknown query type %r" % (qt,))
    return rels, anchors, query_graph[1][0]


class Query(object):
    """One training / evaluation example (netquery/graph.py:38-100).

    ``query_graph`` shapes: ``("2-chain",(t,r1,v),(v,r2,a))``,
    ``("2-inter",(t,r1,a1),(t,r2,a2))``,
    ``("3-inter_chain",(t,r1,a1),((t,r2,v),(v,r3,a2)))``,
    ``("3-chain_inter",(t,r1,v),((v,r2,a1),(v,r3,a2)))`` ...

    Fields: ``formula``, ``anchor_nodes``, ``target_node``, ``neg_samples``,
    ``hard_neg_samples``, ``query_graph`` (None unless ``keep_graph``).
    Negative lists longer than ``neg_sample_max`` are subsampled
    (``<`` for negatives, ``<=`` for hard negatives, as the reference does).
    """

    __slots__ = ("formula", "anchor_nodes", "target_node", "query_graph",
                 "neg_samples", "hard_neg_samples")

    def __init__(self, query_graph, neg_samples, hard_neg_samples,
                 neg_sample_max=100, keep_graph=False):
        rels, anchors, target = _split_query_graph(query_graph)
        self.formula = Formula(query_graph[0], rels)
        self.anchor_nodes = anchors
        self.target_node = target
        self.query_graph = query_graph if keep_graph else None
        if neg_samples is None:
            self.neg_samples = None
        elif len(neg_samples) < neg_sample_max:
            self.neg_samples = list(neg_samples)
        else:
            self.neg_samples = random.sample(list(neg_samples), neg_sample_max)
        if hard_neg_samples is None:
            self.hard_neg_samples = None
        elif len(hard_neg_samples) <= neg_sample_max:
            self.hard_neg_samples = list(hard_neg_samples)
        else:
            self.hard_neg_samples = random.sample(list(hard_neg_samples), neg_sample_max)

    # -- edge helpers (need keep_graph=True) ---------------------------------
    def _edges(self):
        if self.query_graph is None:
            raise Exception("Can only test edge contain if graph is kept. Reinit with keep_graph=True")
        edges = self.query_graph[1:]
        if "inter_chain" in self.query_graph[0] or "chain_inter" in self.query_graph[0]:
            edges = (edges[0], edges[1][0], edges[1][1])
        return edges

    def contains_edge(self, edge):
        edges = self._edges()
        return edge in edges or (edge[1], _reverse_relation(edge[1]), edge[0]) in edges

    def get_edges(self):
        edges = self._edges()
        return set(edges) | set(_reverse_edge(e) for e in edges)

    def _key(self):
        return (self.formula, self.target_node, self.anchor_nodes)

    def __hash__(self):
        return hash(self._key())

    def __eq__(self, other):
        return isinstance(other, Query) and self._key() == other._key()

    def __ne__(self, other):
        return not self.__eq__(other)

    def serialize(self):
        if self.query_graph is None:
            raise Exception("Cannot serialize query loaded with query graph!")
        return (self.query_graph, self.neg_samples, self.hard_neg_samples)

    @staticmethod
    def deserialize(serial_info, keep_graph=False):
        negs = serial_info[1]
        return Query(serial_info[0], negs, serial_info[2],
                     None if negs is None else len(negs), keep_graph=keep_graph)


class Graph(object):
    """Heterogeneous graph container + host-side query sampler.

    ``relations``: ``{mode: [(to_mode, rel_name), ...]}``;
    ``adj_lists``: ``{(mode, rel_name, to_mode): {node: set(neighbours)}}`` with
    both directions populated; ``features(nodes, mode)`` is whatever the encoder
    wants to call (kept for API compatibility, netquery/graph.py:108-122).

    Only ``full_lists`` is touched during training (negative choice for 1-chain
    queries, netquery/model.py:118); the sampler below is used offline to make
    query sets (netquery/graph.py:222-443) and stays on the host.
    """

    def __init__(self, features, feature_dims, relations, adj_lists):
        self.features = features
        self.feature_dims = feature_dims
        self.relations = relations
        self.adj_lists = adj_lists
        self.full_sets = defaultdict(set)
        self.full_lists = {}
        self.meta_neighs = defaultdict(dict)
        # built with the same set operations as the reference (graph.py:116-120): the ORDER of full_lists
        # decides which node ``random.choice`` returns for 1-chain negatives under a given seed
        for rel, adjs in self.adj_lists.items():
            self.full_sets[rel[0]] = self.full_sets[rel[0]].union(set(adjs.keys()))
        for mode, nodes in self.full_sets.items():
            self.full_lists[mode] = list(nodes)
        self._refresh_caches()

    # -- caches ----------------------------------------------------------------
    def _refresh_caches(self):
        self.flat_adj_lists = defaultdict(lambda: defaultdict(list))
        self.rel_edges = {}
        self.edges = 0.0
        for rel, adjs in self.adj_lists.items():
            cnt = 0.0
            for node, neighs in adjs.items():
                self.flat_adj_lists[rel[0]][node].extend((rel, n) for n in neighs)
                cnt += len(neighs)
                self.edges += 1.0
            self.rel_edges[rel] = cnt
        self._rel_keys = list(self.adj_lists.keys())

    def remove_edges(self, edge_list):
        """Drop (u, rel, v) and its reverse; silently skip missing ones (graph.py:148-162)."""
        for u, rel, v in edge_list:
            try:
                self.adj_lists[rel][u].remove(v)
                self.adj_lists[_reverse_relation(rel)][v].remove(u)
            except KeyError:
                continue
        self.meta_neighs = defaultdict(dict)
        self._refresh_caches()

    def get_all_edges(self, seed=0, exclude_rels=frozenset()):
        rng = random.Random(seed)
        edges = []
        for rel, adjs in self.adj_lists.items():
            if rel in exclude_rels:
                continue
            for node, neighs in adjs.items():
                edges.extend((node, rel, n) for n in neighs if n != -1)
        rng.shuffle(edges)
        return edges

    # -- negatives --------------------------------------------------------------
    def get_negative_edge_samples(self, edge, num, rejection_sample=True):
        """Nodes of the edge's source mode NOT linked to edge[2] (graph.py:188-202)."""
        linked = self.adj_lists[_reverse_relation(edge[1])][edge[2]]
        mode = edge[1][0]
        if rejection_sample:
            out, tries = set(), 0
            while len(out) < num and tries <= 100 * num:
                cand = random.choice(self.full_lists[mode])
                if cand not in linked:
                    out.add(cand)
                tries += 1
            if len(out) < num:
                return self.get_negative_edge_samples(edge, num, rejection_sample=False)
        else:
            out = self.full_sets[mode] - linked
        out = list(out)
        return out if len(out) <= num else random.sample(out, num)

    def get_metapath_neighs(self, node, rels):
        cache = self.meta_neighs[rels]
        if node not in cache:
            frontier = {node}
            for rel in rels:
                adj = self.adj_lists[rel]
                frontier = set(n for u in frontier for n in adj.get(u, ()))
            cache[node] = frontier
        return cache[node]

    def _branch_answer_set(self, branch):
        """Nodes t that satisfy one branch ``(t, r, a)`` or ``((t,r,v),(v,r',a))``."""
        if isinstance(branch[0], tuple):     # a 2-hop chain branch
            rels = tuple(_reverse_relation(e[1]) for e in branch[::-1])
            return self.get_metapath_neighs(branch[-1][-1], rels)
        return self.adj_lists[_reverse_relation(branch[1])].get(branch[-1], set())

    def get_negative_samples(self, query):
        """(negatives, hard_negatives) node sets for a query graph (graph.py:240-291).

        negatives      = target-mode nodes that do not satisfy the query;
        hard negatives = nodes satisfying at least one but not all branches.
        Returns (None, None) if either required set is empty.
        """
        qt = query[0]
        target_mode = query[1][1][0]
        if qt in ("2-chain", "3-chain"):
            rels = tuple(_reverse_relation(e[1]) for e in query[1:][::-1])
            negs = self.full_sets[target_mode] - self.get_metapath_neighs(query[-1][-1], rels)
            return (negs, None) if negs else (None, None)
        if qt in ("2-inter", "3-inter", "3-inter_chain"):
            sets = [self._branch_answer_set(b) for b in query[1:]]
            inter = set.intersection(*[set(s) for s in sets])
            union = set.union(*[set(s) for s in sets])
            pos, union_pos = inter, union
        elif qt == "3-chain_inter":
            s1 = self._branch_answer_set(query[-1][0])
            s2 = self._branch_answer_set(query[-1][1])
            adj = self.adj_lists[_reverse_relation(query[1][1])]
            pos = set(n for v in (set(s1) & set(s2)) for n in adj.get(v, ()))
            union_pos = set(n for v in (set(s1) | set(s2)) for n in adj.get(v, ()))
        else:
            raise ValueError("no negatives defined for %r" % (qt,))
        negs = self.full_sets[target_mode] - pos
        hard = union_pos - pos
        if not negs or not hard:
            return None, None
        return negs, hard

    # -- query-subgraph sampling -------------------------------------------------
    def sample_edge(self, node, mode):
        rel, neigh = random.choice(self.flat_adj_lists[mode][node])
        return (node, rel, neigh)

    def _random_start(self):
        rel = random.choice(self._rel_keys)
        node = random.choice(list(self.adj_lists[rel].keys()))
        return node, rel[0]

    def _distinct_out_edges(self, node, mode, k):
        """k pairwise-distinct (rel, neigh) out-edges of node, drawn as the
        reference draws them (first free, the others re-drawn until distinct)."""
        flat = self.flat_adj_lists[mode][node]
        picked = [random.choice(flat)]
        while len(picked) < k:
            cand = random.choice(flat)
            while cand in picked:
                cand = random.choice(flat)
            picked.append(cand)
        return [(node, rel, neigh) for rel, neigh in picked]

    def _sample(self, shape_of, arity, start_node):
        node, mode = self._random_start() if start_node is None else start_node
        num_edges = shape_of(arity)
        if num_edges > len(self.flat_adj_lists[mode][node]):
            return None
        if arity == 3:
            if num_edges == 1:
                edge = self.sample_edge(node, mode)
                sub = self._sample(shape_of.sub(), 2, (edge[2], edge[1][0]))
                if sub is None:
                    return None
                if sub[0] == "2-chain":
                    return ("3-chain", edge, sub[1], sub[2])
                return ("3-chain_inter", edge, (sub[1], sub[2]))
            if num_edges == 2:
                e1, e2 = self._distinct_out_edges(node, mode, 2)
                return ("3-inter_chain", e1, (e2, self.sample_edge(e2[2], e2[1][-1])))
            e1, e2, e3 = self._distinct_out_edges(node, mode, 3)
            return ("3-inter", e1, e2, e3)
        if num_edges == 1:
            edge = self.sample_edge(node, mode)
            return ("2-chain", edge, self.sample_edge(edge[2], edge[1][-1]))
        e1, e2 = self._distinct_out_edges(node, mode, 2)
        return ("2-inter", e1, e2)

    def sample_query_subgraph(self, arity, start_node=None):
        """Random query graph of the given arity (graph.py:364-434):
        arity 3 -> 1/2 one out-edge (3-chain / 3-chain_inter), 1/4 two, 1/4 three;
        arity 2 -> 1/2 2-chain, 1/2 2-inter.  None when the start node is too small."""
        if arity not in (2, 3):
            raise Exception("Only arity of at most 3 is supported for queries")
        return self._sample(_RandomShape(), arity, start_node)

    def sample_query_subgraph_bytype(self, q_type, start_node=None):
        """Random query graph of one named type (graph.py:298-361)."""
        return self._sample(_FixedShape(q_type), int(q_type[0]), start_node)

    def sample_queries(self, arity, num_samples, neg_sample_max, verbose=False):
        out = []
        while len(out) < num_samples:
            q = self.sample_query_subgraph(arity)
            if q is None:
                continue
            negs, hard = self.get_negative_samples(q)
            if negs is None or ("inter" in q[0] and hard is None):
                continue
            out.append(Query(q, negs, hard, neg_sample_max=neg_sample_max, keep_graph=True))
        return out

    def sample_test_queries(self, train_graph, q_types, samples_per_type, neg_sample_max, verbose=False):
        out = []
        for q_type in q_types:
            got = 0
            while got < samples_per_type:
                q = self.sample_query_subgraph_bytype(q_type)
                if q is None or not train_graph._is_negative(q, q[1][0], False):
                    continue
                negs, hard = self.get_negative_samples(q)
                if negs is None or ("inter" in q[0] and hard is None):
                    continue
                out.append(Query(q, negs, hard, neg_sample_max=neg_sample_max, keep_graph=True))
                got += 1
        return out

    # -- structural checks (sampler invariants, graph.py:447-534) -----------------
    def _has_edge(self, edge):
        return edge[2] in self.adj_lists.get(edge[1], {}).get(edge[0], ())

    def _is_subgraph(self, query):
        """Every edge of the query graph exists and chains hook up."""
        qt = query[0]
        if qt in CHAIN_TYPES:
            edges = query[1:]
            ok = all(self._has_edge(e) for e in edges)
            return ok and all(edges[i][2] == edges[i + 1][0] for i in range(len(edges) - 1))
        if qt in ("2-inter", "3-inter"):
            return all(self._has_edge(e) and e[0] == query[1][0] for e in query[1:])
        if qt == "3-inter_chain":
            e1, (e2, e3) = query[1], query[2]
            return (self._has_edge(e1) and self._has_edge(e2) and self._has_edge(e3)
                    and e1[0] == e2[0] and e2[2] == e3[0])
        if qt == "3-chain_inter":
            e1, (e2, e3) = query[1], query[2]
            return (self._has_edge(e1) and self._has_edge(e2) and self._has_edge(e3)
                    and e1[2] == e2[0] and e2[0] == e3[0])
        return False

    def _satisfies(self, query, node):
        """Does ``node`` in the target slot satisfy the (conjunctive) query?"""
        qt = query[0]
        if qt in CHAIN_TYPES:
            rels = tuple(_reverse_relation(e[1]) for e in query[1:][::-1])
            return node in self.get_metapath_neighs(query[-1][-1], rels)
        if qt in ("2-inter", "3-inter", "3-inter_chain"):
            return all(node in self._branch_answer_set(b) for b in query[1:])
        s1 = self._branch_answer_set(query[-1][0])
        s2 = self._branch_answer_set(query[-1][1])
        adj = self.adj_lists[query[1][1]]
        return any(v in s1 and v in s2 for v in adj.get(node, ()))

    def _is_negative(self, query, neg_node, is_hard):
        """``neg_node`` does not satisfy ``query``; a *hard* negative additionally
        satisfies at least one branch (graph.py:487-534)."""
        if self._satisfies(query, neg_node):
            return False
        if not is_hard:
            return True
        qt = query[0]
        if qt in ("2-inter", "3-inter", "3-inter_chain"):
            return any(neg_node in self._branch_answer_set(b) for b in query[1:])
        if qt == "3-chain_inter":
            s1 = self._branch_answer_set(query[-1][0])
            s2 = self._branch_answer_set(query[-1][1])
            adj = self.adj_lists[query[1][1]]
            return any((v in s1) or (v in s2) for v in adj.get(neg_node, ()))
        return False

    def _run_test(self, num_samples=1000):
        """Self-check of the sampler, as graph.py:538-562 does."""
        for arity in (2, 3):
            for q in self.sample_queries(arity, num_samples, 1):
                assert self._is_subgraph(q.query_graph)
                assert self._is_negative(q.query_graph, q.neg_samples[0], False)
                if q.hard_neg_samples is not None:
                    assert self._is_negative(q.query_graph, q.hard_neg_samples[0], True)
        return True


class _RandomShape(object):
    """Number of out-edges at the root for arity-driven sampling."""

    def __call__(self, arity):
        return random.choice([1, 1, 2, 3]) if arity == 3 else random.choice([1, 2])

    def sub(self):
        return self


class _FixedShape(object):
    _ROOT = {"3-chain": 1, "3-chain_inter": 1, "3-inter_chain": 2, "3-inter": 3,
             "2-chain": 1, "2-inter": 2}

    def __init__(self, q_type):
        self.q_type = q_type

    def __call__(self, arity):
        return self._ROOT[self.q_type]

    def sub(self):
        return _FixedShape("2-chain" if self.q_type == "3-chain" else "2-inter")
