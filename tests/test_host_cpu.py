"""CPU tests of the host side: data model, sampler invariants, batch selection, eval statistics,
descriptor building, and that libgqe.so loads and exports every symbol include/gqe.h declares."""
import os
import pickle
import random
import re

import numpy as np
import pytest

from golden_utils import GOLDEN, to_rels
from graphqembed_amd import data_utils, graph as G
from oracle import netquery_numpy as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tiny():
    rel, adj, ids = data_utils.make_synthetic_graph(data_utils.BIO_TINY_SIZES, edges_per_kind=data_utils.BIO_TINY_EDGES_PER_KIND, seed=0)
    return G.Graph(None, {m: 8 for m in rel}, rel, adj), ids


def test_library_exports_every_declared_symbol():
    from graphqembed_amd import engine
    header = open(os.path.join(ROOT, "include", "gqe.h")).read()
    declared = set(re.findall(r"\b(gqe_[a-z_]+)\s*\(", header))
    assert declared == set(engine.SYMBOLS), declared ^ set(engine.SYMBOLS)
    lib = engine.load_library()                     # build() must have run (the driver runs it first)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.gqe_abi_version() == engine.ABI_VERSION
    # the host-side sampler's header (include/gqe_sampler.h) is served by the same library
    from graphqembed_amd import sampler
    header = open(os.path.join(ROOT, "include", "gqe_sampler.h")).read()
    declared = set(re.findall(r"\b(gqe_[a-z_]+)\s*\(", header))
    assert declared == set(sampler.SAMPLER_SYMBOLS), declared ^ set(sampler.SAMPLER_SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_no_gpu_means_loud_failure():
    import torch
    from graphqembed_amd.engine import ArenaLayout, Engine, GqeLibraryError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lay = ArenaLayout()
    lay.add("enc.feat-a.weight", (4, 16))
    with pytest.raises(GqeLibraryError):
        Engine(16, "bilinear-diag", "min", lay)


def test_formula_and_query_follow_the_reference_shapes():
    with open(os.path.join(GOLDEN, "queries_tiny.pkl"), "rb") as f:
        data = pickle.load(f)
    seen = set()
    for qtype, infos in data["train"].items():
        for info in infos[:50]:
            q = G.Query.deserialize(info, keep_graph=True)
            assert q.formula.query_type == qtype
            plan = O.make_plan(qtype, q.formula.rels)            # oracle's independent reading of the formula
            assert q.formula.target_mode == plan["target_mode"]
            assert list(q.formula.anchor_modes) == plan["anchor_modes"]
            assert len(q.anchor_nodes) == len(q.formula.anchor_modes)
            assert q.serialize()[0] == info[0]
            assert hash(G.Formula(qtype, q.formula.rels)) == hash(q.formula) and G.Formula(qtype, q.formula.rels) == q.formula
            seen.add(qtype)
    assert seen == set(G.QUERY_TYPES)
    q = G.Query(("1-chain", (1, ("a", "r", "b"), 2)), [5, 6, 7, 8], None, neg_sample_max=2)
    assert len(q.neg_samples) == 2 and q.hard_neg_samples is None and q.query_graph is None
    with pytest.raises(Exception):
        q.serialize()
    assert G._reverse_relation(("a", "r", "b")) == ("b", "r", "a")


def test_sampler_invariants(tiny):
    g, _ = tiny
    random.seed(5)
    assert g._run_test(num_samples=60)
    counts = {}
    for arity in (2, 3):
        for q in g.sample_queries(arity, 150, 3):
            counts[q.formula.query_type] = counts.get(q.formula.query_type, 0) + 1
            assert g._is_subgraph(q.query_graph)
            assert not g._is_negative(q.query_graph, q.target_node, False)      # the target satisfies its query
            for n in q.neg_samples:
                assert g._is_negative(q.query_graph, n, False)
            if "inter" in q.formula.query_type:
                for n in q.hard_neg_samples:
                    assert g._is_negative(q.query_graph, n, True)
    assert set(counts) == {"2-chain", "2-inter", "3-chain", "3-inter", "3-inter_chain", "3-chain_inter"}
    for t in ("2-chain", "3-inter_chain", "3-chain_inter"):
        q = None
        while q is None:
            q = g.sample_query_subgraph_bytype(t)
        assert q[0] == t and g._is_subgraph(q)
    e = g.get_all_edges(seed=1)[0]
    negs = g.get_negative_edge_samples(e, 5)
    assert len(negs) == 5 and all(n not in g.adj_lists[G._reverse_relation(e[1])][e[2]] for n in negs)
    n_before = sum(len(v) for a in g.adj_lists.values() for v in a.values())
    g.remove_edges([e])
    assert sum(len(v) for a in g.adj_lists.values() for v in a.values()) == n_before - 2


def test_draw_batch_is_the_reference_slicing_rule():
    from graphqembed_amd.train_helpers import BatchSpec, check_conv, draw_batch, iteration_plan, update_loss
    fa, fb = G.Formula("1-chain", (("a", "r", "b"),)), G.Formula("1-chain", (("b", "r", "a"),))
    tq = {fa: list(range(700)), fb: list(range(100))}
    for it in range(40):
        np.random.seed(it)
        f, queries = draw_batch(tq, it, 512)
        after_mine = np.random.get_state()[1][:4].tolist()
        np.random.seed(it)                                   # the reference's rule, train_helpers.py:96-105
        num = np.array([700.0, 100.0])
        want_f = [fa, fb][int(np.argmax(np.random.multinomial(1, num / num.sum())))]
        n = len(tq[want_f])
        start = (it * 512) % n
        end = min(((it + 1) * 512) % n, n)
        end = n if end <= start else end
        assert f == want_f and queries == tq[want_f][start:end] and 1 <= len(queries) <= 512
        assert np.random.get_state()[1][:4].tolist() == after_mine           # the same amount of np.random consumed
    # the schedule: nothing beyond the 1-chain batch during burn-in; then chains once, intersections twice (regular, hard)
    types = ["1-chain", "2-chain", "2-inter", "3-chain", "3-inter_chain"]
    assert list(iteration_plan(types, False, 0.01, 0.005)) == []
    assert list(iteration_plan(types, True, 0.01, 0.005)) == [
        BatchSpec("2-chain", 0.01, False), BatchSpec("2-inter", 0.005, False), BatchSpec("2-inter", 0.005, True),
        BatchSpec("3-chain", 0.01, False), BatchSpec("3-inter_chain", 0.005, False), BatchSpec("3-inter_chain", 0.005, True)]
    assert not check_conv([1.0, 2.0, 3.0]) and check_conv([1.0, 2.0, 1.0, 2.0]) and not check_conv([1.0, 1.0, 2.0, 2.0])
    losses, ema = update_loss(2.0, [], None)
    losses, ema = update_loss(4.0, losses, ema)
    assert losses == [2.0, 4.0] and abs(ema - 2.02) < 1e-12


def test_auc_and_percentile_match_sklearn_and_scipy():
    from scipy import stats
    from sklearn.metrics import roc_auc_score
    from graphqembed_amd.utils import _auc, _percentile_of_score
    rng = np.random.RandomState(0)
    for n in (2, 10, 500):
        labels = np.r_[np.ones(n), np.zeros(n)].astype(int)
        scores = np.round(rng.randn(2 * n), 1 if n > 10 else 3)            # ties on purpose
        assert abs(_auc(labels, scores) - roc_auc_score(labels, scores)) < 1e-12
    for _ in range(50):
        a = np.round(rng.randn(rng.randint(1, 30)), 1)
        s = float(np.round(rng.randn(), 1))
        assert abs(_percentile_of_score(a, s) - stats.percentileofscore(a, s)) < 1e-12


def test_formula_plan_descriptors_agree_with_the_oracle_plan():
    from graphqembed_amd.engine import ArenaLayout
    from graphqembed_amd.tensorize import FormulaPlan, pack_margin_batches, post_key, pre_key, rel_key, table_key
    from gpu_utils import TOY_FORMULAS, TOY_KINDS, TOY_SIZES, random_params
    params = random_params(np.random.RandomState(0), 16, "bilinear-diag", "min", TOY_SIZES, TOY_KINDS)
    lay = ArenaLayout()
    for k, v in params.items():
        lay.add(k, v.shape)
    assert all(off % 64 == 0 for off, _ in lay.entries.values())
    for qtype, rels in TOY_FORMULAS.items():
        fp = FormulaPlan(G.Formula(qtype, rels), lay, "min")
        op = O.make_plan(qtype, rels)
        assert fp.static["target_table"] == lay.offset(table_key(op["target_mode"]))
        assert fp.static["anchor_table"] == [lay.offset(table_key(m)) for m in op["anchor_modes"]]
        branches = [op["chain"]] if "chain" in op else op["branches"]
        assert fp.static["hops"] == [[lay.offset(rel_key(r)) for r in br] for br in branches]
        if op.get("final"):
            assert fp.static["final"] == lay.offset(rel_key(op["final"][0]))
        if "inter_mode" in op:
            assert fp.static["pre"] == lay.offset(pre_key(op["inter_mode"])) and fp.static["post"] == lay.offset(post_key(op["inter_mode"]))
        assert fp.touched == O.touched_keys(op, "bilinear-diag", "min")
    fp = FormulaPlan(G.Formula("2-inter", TOY_FORMULAS["2-inter"]), lay, "min")
    descs, idx, n = pack_margin_batches([(fp, np.arange(3), np.arange(3) + 10, np.arange(6).reshape(2, 3) + 20, 0.5, 1.0)] * 2)
    assert n == 6 and descs[1]["idx_offset"] == 12 and descs[1]["out_offset"] == 3 and idx.dtype == np.int32
    assert idx.tolist()[:12] == [0, 1, 2, 10, 11, 12, 20, 21, 22, 23, 24, 25]


def test_direct_encoder_row_lookup():
    import torch
    from graphqembed_amd.encoders import DirectEncoder
    ids = {"a": [100, 101, 105], "b": [7, 9]}
    nm = data_utils.make_node_maps(ids)
    mods = {m: torch.nn.Embedding(len(nm[m]) + 1, 4) for m in ids}
    enc = DirectEncoder(None, mods, node_maps=nm)
    assert enc.rows([105, 100, -1, 101], "a").tolist() == [3, 1, 0, 2]
    with pytest.raises(KeyError):
        enc.rows([102], "a")
    assert DirectEncoder(None, mods).rows([4, 0], "b").tolist() == [5, 1]
    assert sorted(k for k, _ in enc.named_parameters()) == ["feat-a.weight", "feat-b.weight"]


def test_decoder_parameter_names_shapes_and_errors():
    import torch
    from graphqembed_amd import utils
    rel, adj, ids = data_utils.make_synthetic_graph({"a": 5, "b": 4}, kinds=(("a", "r", "b"), ("a", "s", "a")), edges_per_kind=6, seed=0)
    g = G.Graph(None, {"a": 16, "b": 16}, rel, adj)
    dims = {"a": 16, "b": 16}
    for name, shape in (("bilinear", (16, 16)), ("transe", (16,)), ("bilinear-diag", (16,))):
        dec = utils.get_metapath_decoder(g, dims, name)
        assert dec.kind == name
        assert sorted(n for n, _ in dec.named_parameters()) == ["a_r_b", "a_s_a", "b_r_a"]
        assert all(tuple(p.shape) == shape for p in dec.parameters())
    inter = utils.get_intersection_decoder(g, dims, "min")
    assert sorted(n for n, _ in inter.named_parameters()) == ["a_postmat", "a_premat", "b_postmat", "b_premat"]
    assert [utils.get_intersection_decoder(g, dims, k).kind for k in ("mean", "min-simple", "mean-simple")] == ["mean", "min-simple", "mean-simple"]
    for bad in (lambda: utils.get_metapath_decoder(g, dims, "nope"), lambda: utils.get_intersection_decoder(g, dims, "nope"),
                lambda: utils.get_encoder(4, g, dims, {}, False)):
        with pytest.raises(Exception):
            bad()


def test_synthetic_query_pools_have_the_right_shape():
    from graphqembed_amd import synth
    g = synth.CsrGraph({"drug": 300, "disease": 200, "protein": 500, "sideeffect": 150, "function": 250}, edges_per_kind=2000, seed=1)
    assert len(g.rels) == 14
    pools = synth.make_pools(g, list(G.QUERY_TYPES), formulas_per_type=2, pool_size=256, seed=0)
    for qt, plist in pools.items():
        for p in plist:
            assert p.formula.query_type == qt and p.n == 256
            assert p.anchors.shape == (len(p.formula.anchor_modes), 256)
            assert p.target.min() >= 1 and p.target.max() <= g.mode_sizes[p.formula.target_mode]
            for i, m in enumerate(p.formula.anchor_modes):
                assert p.anchors[i].min() >= 1 and p.anchors[i].max() <= g.mode_sizes[m]
            assert (p.hard is not None) == ("inter" in qt)
    items = synth.mix_iteration(pools, synth.FULL_MIX, 3, 64, rank=1, world=2)
    assert len(items) == 9 and all(len(it[1]) == 64 for it in items) and abs(items[0][4] - 0.5) < 1e-12


def test_sharded_trainer_refuses_an_optimiser_without_adam_hyperparameters():
    """Row-sharded training steps its shards with the library's Adam; the optimiser object only supplies lr / betas / eps.
    An SGD, or an object naming none of them, must raise instead of training with defaults."""
    import torch
    from graphqembed_amd.trainer import TensorizedTrainer

    class Eng(object):
        sharded = True

    class Pool(object):
        def __init__(self):
            self.formula, self.n = G.Formula("1-chain", (("a", "r", "b"),)), 4
            self.target = np.arange(4, dtype=np.int32)
            self.anchors = np.arange(4, dtype=np.int32)[None]
            self.neg = self.hard = None

    class Bare(object):
        def step(self):
            pass

    class Named(Bare):
        lr, betas, eps = 0.02, (0.8, 0.9), 1e-6

    w = torch.nn.Parameter(torch.zeros(2))
    mk = lambda opt: TensorizedTrainer(None, opt, {"1-chain": [Pool()]}, {}, engine=Eng(), plan_of=lambda f: None)
    for bad in (Bare(), torch.optim.SGD([w], lr=0.1)):
        with pytest.raises(Exception, match="must be an Adam"):
            mk(bad)
    assert mk(Named())._adam_hyper() == (0.02, (0.8, 0.9), 1e-6)
    assert mk(torch.optim.Adam([w], lr=0.03))._adam_hyper() == (0.03, (0.9, 0.999), 1e-8)


def test_zipf_graphs_are_heavy_tailed():
    """synth's Zipf option (hub nodes, Zipfian words): the largest degree / word frequency dwarfs the uniform graph's."""
    from graphqembed_amd import synth
    sizes = {"drug": 300, "disease": 200, "protein": 500, "sideeffect": 150, "function": 250}
    gu = synth.bio_synth(seed=1, sizes=sizes, edges_per_kind=2000)
    gz = synth.bio_synth(seed=1, sizes=sizes, edges_per_kind=2000, zipf=1.0)
    rel = ("drug", "targets", "protein")
    du, dz = np.diff(gu.csr[rel][0]), np.diff(gz.csr[rel][0])
    assert dz.max() > 4 * du.max() and len(gz.csr[rel][1]) > 500
    pools = synth.make_pools(gz, ["1-chain", "2-inter", "3-inter_chain"], formulas_per_type=2, pool_size=300, seed=0)
    for plist in pools.values():
        for p in plist:
            assert p.target.min() >= 1 and p.anchors.min() >= 1
    r = synth.reddit_synth(seed=0, sizes={"user": 400, "post": 300, "community": 20}, edges_per_kind=3000, n_words=500, zipf=1.0)
    cnt = np.bincount(r.bags["post"][1], minlength=500)
    assert cnt.max() > 8 * np.median(cnt[cnt > 0])


def test_operand_ordered_matrix_layout_is_a_permutation_the_loader_reads_linearly():
    """GQE_TILE_INDEX (graphqembed_amd/csrc/gqe_dev.h) restated: the copy of a d x d matrix is a permutation of its elements, and
    what load_a_slab (gqe_fused.h) reads for output row block i0, k-block kb, lane l = lq + 16 lk — sixteen bytes at byte offset
    ((i0 / 16) (d / 16) + kb) 1024 + 16 l — are exactly M[i0 + lq][16 kb + 4 lk .. + 3], the lane's four MFMA A operands."""
    def tile_index(i, k, d):
        return (((i >> 4) * (d >> 4) + (k >> 4)) * 64 + (i & 15) + 16 * ((k & 15) >> 2)) * 4 + (k & 3)

    for d in (16, 48, 64, 80, 128, 256):
        i, k = np.meshgrid(np.arange(d), np.arange(d), indexing="ij")
        idx = tile_index(i, k, d)
        assert sorted(idx.ravel().tolist()) == list(range(d * d))
        m = np.arange(d * d, dtype=np.float32).reshape(d, d)
        copy = np.empty(d * d, dtype=np.float32)
        copy[idx] = m
        for i0 in range(0, d, 16):
            for kb in range(d // 16):
                for lane in (0, 5, 17, 42, 63):
                    lq, lk = lane & 15, lane >> 4
                    off = (((i0 // 16) * (d // 16) + kb) * 1024 + 16 * lane) // 4
                    np.testing.assert_array_equal(copy[off:off + 4], m[i0 + lq, 16 * kb + 4 * lk: 16 * kb + 4 * lk + 4])


def test_native_run_length_never_crosses_an_event_of_the_schedule():
    """train_helpers.native_run_length against the reference's loop walked iteration by iteration (train_helpers.py:48-79): a run
    handed to the native loop ends AT the next iteration behind which validation runs, before the iteration that finds max_burn_in
    losses in the edges-only phase, at max_iter, or at the run cap — whichever comes first — and is never empty."""
    from graphqembed_amd.train_helpers import native_run_length
    rng = np.random.RandomState(0)
    for _ in range(3000):
        max_iter = int(rng.randint(1, 400))
        val_every = int(rng.choice([1, 2, 3, 7, 50, 100, 1000]))
        max_burn_in = int(rng.randint(1, 300))
        max_run = int(rng.choice([1, 5, 64, 4096]))
        all_types = bool(rng.randint(2))
        first = int(rng.randint(0, max_iter))
        seen = int(rng.randint(0, max_burn_in)) if not all_types else int(rng.randint(0, 1000))
        n = native_run_length(first, max_iter, all_types, seen, max_burn_in, val_every, max_run)
        # walk the loop: iteration i runs if it is inside max_iter and (edges-only) fewer than max_burn_in losses were seen at its
        # start; it is the last of the run if validation follows it, or the cap is reached
        want, count = 0, seen
        for i in range(first, max_iter):
            if not all_types and count >= max_burn_in:
                break
            want += 1
            count += 1
            if (i >= val_every and i % val_every == 0) or want == max_run:
                break
        assert n == want and n >= 1, (first, max_iter, all_types, seen, max_burn_in, val_every, max_run, n, want)
