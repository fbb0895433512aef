#!/usr/bin/env python3
"""bench.py — queries/sec of the fused MI355X training step on the BASELINE workload.

    python bench.py [--gpus N --steps K --warmup W]
        N = 1 runs in-process; N > 1 without a launcher environment re-executes itself under
        ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`` (one rank per
        GPU, backend nccl = RCCL); under torchrun (WORLD_SIZE set) it is one of the N ranks.

step      = one training iteration of the reference's post-burn-in schedule over the "Bio full conjunctive mix"
            (train_helpers.py:50-79): 9 batches x B=512 = 4 608 (query, negative) pairs -> fused forward/backward
            (one grouped launch) -> deferred matrix gradients -> [N>1: gradient exchange] -> one fused dense Adam
            step (+ gradient reset).
workload  = "bio-synth" (SURVEY.md §8d C3): 5 modes / 97 000 nodes / 14 directed relations, d=128, bilinear-diag
            decoder + SetIntersection(min), P = 12 582 912 parameters; index feeds of 32 distinct pre-sampled
            iterations are resident in HBM (``host_fed`` reports the rate with sampling + packing + pinned upload
            by the native feeder, gqe_feeder_run, inside the timed region).
timing    = W warm-up steps, then blocks of EXACTLY K steps, each bracketed by barrier + torch.cuda.synchronize()
            on both sides (max over ranks); blocks repeat until >= 0.5 s have been timed and the MEDIAN block is
            reported (K=20 at 0.1 ms per step is a 2 ms region: one noisy step would move it by 5 %).
value     = whole-job queries/s = K * 4608 * N / median block time (weak scaling: every rank trains its own 4 608
            queries per step; gradients are averaged).
roofline  = the dominant kernel, the fused Adam pass.  ``achieved`` = the bytes the pass has to move / its mean
            launch duration (hipEvents recorded by the library on the launch stream, every 16th launch of the timed
            region).  Bytes: 24 B per table parameter (p, m, v read + written; the dense table gradient does not
            exist: row gradients are per-row lists) + 32 B per relation / Pre / Post parameter + 4 B list head per
            table row + (4 d + 8) B per gradient contribution.  ``frac`` is against the 8 TB/s spec,
            ``frac_of_measured_copy_peak`` against the 6.29 TB/s a float4 copy reaches (MI355X_MICROARCH.md).
            SURVEY.md §8d's 32 B/param figure is kept as ``survey_bytes_per_launch`` for reference only — 8 B/param
            of it are never moved.  ``traffic`` = PMC HBM bytes of the last profiled run (profiles/).
            One GPU, eager Adam: the step runs with gqe_set_deferred_gemm (include/gqe.h) — the pair-GEMM units and the
            loss finalize ride in front of the pass's chunks, in ITS launch (gqe_opt_gemm_kernel: ``roofline.kernel`` says so;
            the bytes then include the units' operand rows, ``optimiser_pass_bytes_per_launch`` is the pass alone), and the
            d x d matrices are stepped by a small launch behind it (``kernels.pair_gemm.matrix_step_launch``).  The losses are
            read behind the timed loop.  GQE_BENCH_NO_DEFERRED_GEMM=1 measures the three-launch step of rounds 1-3.
kernels   = fused forward/backward and pair GEMM: mean launch time, algorithmic bytes, and the fp32 MFMA rate of
            their d x d contractions against the 157.3 TF/s exact-fp32 MFMA peak.
configs   = (N=1) the other measurement configurations of SURVEY.md §8d, each with ms/step and kernel times:
            C1 (1-chain only), C2 (2-chain + 2-inter, with and without the 1-chain batch), C4 (full Bilinear decoder,
            the MFMA path), the scaled-batch variant (B=8192 per formula) and the 11-batch mix with 3-chain_inter.
reddit_synth = BASELINE config 5's workload (3 modes, 12 directed relations, EmbeddingBag post features, d=256;
            tables far beyond the 256 MB Infinity Cache) on this many GPUs; ``--workload reddit-synth --dim 256``
            makes it the main measurement.
lazy_exact_adam (N=1, extra key, NOT the headline) = the same loop with gqe_set_lazy_adam.
cpu_baseline = oracle/netquery_torch.py (torch-CPU port of the reference's iteration) on the same parameters and
            batches, timed on this host (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0          # ... 6.29 TB/s measured with a float4 copy
MFMA_F32_TFS = 157.3           # ... exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) peak
ROWS_TOUCHED = {"1-chain": 3, "2-chain": 3, "3-chain": 3, "2-inter": 4, "3-inter_chain": 4, "3-chain_inter": 4, "3-inter": 5}
N_BRANCH = {"2-inter": 2, "3-inter": 3, "3-inter_chain": 2, "3-chain_inter": 2}
CHAIN_HOPS = {"1-chain": 1, "2-chain": 2, "3-chain": 3}


def pmc_traffic(kernel, tag):
    """HBM bytes per launch from the newest committed PMC summary (profiles/r*_pmc_traffic.json, key ``tag`` =
    workload); counters cannot be collected from inside this process, so this is the last profiled value."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        with open(path) as f:
            data = json.load(f)
        entry = data.get(tag, data if tag == "bio-synth" else {}).get(kernel)
        if entry:
            return entry.get("hbm_bytes_per_launch")
    return None


def rocprof_ms(kernel, tag):
    """Average duration (ms) of ``kernel`` in the newest committed rocprofv3 kernel trace of this workload
    (profiles/r*_pmc_traffic.json carries it as ``rocprof_avg_us``): the figure to compare with ``avg_launch_ms``, whose hipEvent
    bracket also contains the launch boundary in front of the kernel."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        with open(path) as f:
            data = json.load(f)
        entry = data.get(tag, {}).get(kernel)
        if entry and entry.get("rocprof_avg_us") is not None:
            return round(entry["rocprof_avg_us"] * 1e-3, 5)
    return None


def build_layout(g, d, decoder, inter, shard_world=1):
    """``shard_world`` > 1: the layout of ONE rank's shards in row-sharded mode (ceil(rows / world) rows per table)."""
    from graphqembed_amd.engine import ArenaLayout
    from graphqembed_amd.parallel import shard_rows
    from graphqembed_amd.tensorize import post_key, pre_key, rel_key, table_key
    layout = ArenaLayout()
    for m in g.modes:                                             # (bag tables — the posts' word table — stay replicated)
        rows = g.table_rows[m] if m in g.bags else shard_rows(g.table_rows[m], shard_world)
        layout.add(table_key(m), (rows, d))                       # bio/data_utils.py:14-17, reddit/data_utils_new.py:154-158
    for m in g.relations:                                         # decoders.py:136-140 order
        for (to, name) in g.relations[m]:
            layout.add(rel_key((m, name, to)), (d, d) if decoder == "bilinear" else (d,))
    if not inter.endswith("simple"):
        for m in g.modes:
            layout.add(pre_key(m), (d, d))
            layout.add(post_key(m), (d, d))
    return layout


def init_params(eng, d, seed):
    """The reference's initial distributions (bio/data_utils.py:19, decoders.py:139,225,282-285)."""
    import torch
    gen = torch.Generator(device=eng.device)
    gen.manual_seed(seed)
    for k, (off, shape) in eng.layout.entries.items():
        v = eng.layout.view(eng.params, k)
        if k.startswith("enc."):
            v.normal_(0, 1.0 / d, generator=gen)
        elif len(shape) == 1:
            v.uniform_(-6.0 / np.sqrt(d), 6.0 / np.sqrt(d), generator=gen)
        else:
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            v.uniform_(-lim, lim, generator=gen)


def mfma_flops(qtype, decoder, inter, d, n):
    """fp32 MFMA flops of (fused kernel, pair-GEMM kernel) for one batch of n queries: every d x d contraction is
    2 d^2 flops per query.  SetIntersection: Pre per branch + Post forward, their transposes backward, one
    deferred rank-n update per use.  Full Bilinear: a hop is a contraction (both score sides on chains)."""
    unit = 2.0 * d * d * n
    fused = gemm = 0.0
    nb = N_BRANCH.get(qtype, 0)
    if nb and not inter.endswith("simple"):
        fused += (2 * nb + 2) * unit
        gemm += (nb + 1) * unit
    if decoder == "bilinear":
        if nb:
            hops = nb + (1 if qtype == "3-inter_chain" else 0) + (1 if qtype == "3-chain_inter" else 0)
            fused += 2 * hops * unit
            gemm += hops * unit
        else:
            k = CHAIN_HOPS[qtype]
            fused += 4 * k * unit
            gemm += 2 * k * unit
    return fused, gemm


class Workload(object):
    """A synthetic graph, its parameter layout, query pools and ``n_distinct`` pre-sampled iterations."""

    def __init__(self, name, d, decoder, inter, mix, B, rank=0, world=1, n_distinct=32, formulas_per_type=6, zipf=None):
        """``zipf`` = exponent: node degrees (and reddit-synth's word frequencies) follow 1 / rank^zipf instead of the uniform
        draw — hub rows collect hundreds of gradient contributions per step (synth.CsrGraph)."""
        from graphqembed_amd import synth
        self.name, self.d, self.decoder, self.inter, self.mix, self.B, self.zipf = name, d, decoder, inter, mix, B, zipf
        if name == "reddit-synth":
            # GQE_BENCH_REDDIT_SCALE (--reddit-scale): the config-5 world shrunk by that factor — functional runs of many ranks on one GPU
            scale = float(os.environ.get("GQE_BENCH_REDDIT_SCALE", "1"))
            kw = {}
            if scale != 1.0:
                kw = dict(sizes={m: max(64, int(n * scale)) for m, n in synth.REDDIT_SYNTH_SIZES.items()},
                          edges_per_kind=max(2000, int(synth.REDDIT_SYNTH_EDGES_PER_KIND * scale)), n_words=max(256, int(synth.REDDIT_SYNTH_WORDS * scale)))
            self.g = synth.reddit_synth(seed=0, zipf=zipf, **kw)
        else:
            self.g = synth.bio_synth(seed=0, zipf=zipf)
        self.layout = build_layout(self.g, d, decoder, inter)
        self.types = sorted(set(m[0] for m in mix))
        self.pools = synth.make_pools(self.g, self.types, formulas_per_type=formulas_per_type, pool_size=max(16 * B, 8192), seed=0)
        self.qpi = B * len(mix)
        self.n_distinct = n_distinct
        self.item_sets = [synth.mix_iteration(self.pools, mix, s, B, rank=rank, world=world) for s in range(n_distinct)]
        from graphqembed_amd.tensorize import table_key
        self.bags = {table_key(m): csr for m, csr in self.g.bags.items()}

    def describe(self):
        g = self.g
        n_rel = sum(len(v) for v in g.relations.values())
        return "%d modes, %d nodes, %d directed relations, seed 0%s" % (len(g.modes), sum(g.mode_sizes.values()), n_rel,
                                                                       (", Zipf(%.2g) degrees%s" % (self.zipf, " and word frequencies" if g.bags else "")) if self.zipf else "")

    def engine(self, rank=0, world=1, lazy=False, shard=None):
        """``shard`` = (rank, world): row-sharded mode — the engine holds this rank's shards of the tables."""
        from graphqembed_amd.engine import Engine
        layout = build_layout(self.g, self.d, self.decoder, self.inter, shard[1]) if shard else self.layout
        eng = Engine(self.d, self.decoder, self.inter, layout, max_queries=self.qpi, max_batches=len(self.mix),
                     rank=rank, world=world, lazy_adam=lazy, bags=self.bags, shard=shard)
        init_params(eng, self.d, seed=0 if not shard else 1000 + shard[0])   # replicas start equal; shards are what they are
        if shard:                                                  # ... but what is replicated starts equal: relation / Pre / Post
            import torch                                           # tensors and the bag (word) tables
            gen = torch.Generator(device=eng.device)
            gen.manual_seed(0)
            for off, n in eng.dense_spans():
                eng.params[off:off + n].uniform_(-0.1, 0.1, generator=gen)
            for k in eng.bag_keys:
                eng.layout.view(eng.params, k).normal_(0, 1.0 / self.d, generator=gen)
        return eng

    def prepare(self, eng, dist=None):
        import torch
        from graphqembed_amd import parallel
        from graphqembed_amd.tensorize import FormulaPlan, pack_margin_batches, table_key
        bag_len = {m: np.diff(ptr) for m, (ptr, ids) in self.g.bags.items()}
        prepared = []
        plans = {}
        for items in self.item_sets:
            packed = []
            for (f, t, ng, a, w, m) in items:
                if f not in plans:
                    plans[f] = FormulaPlan(f, eng.layout, self.inter)
                packed.append((plans[f], t, ng, a, w, m))
            descs, idx, _ = pack_margin_batches(packed)
            if eng.sharded:                                        # host feed of GLOBAL rows: planned (owner sort + publication) per step, inside the timed loop
                ps = eng.prepare_shard(descs, idx, set().union(*[p[0].touched for p in packed]))
            else:
                ps = eng.prepare_margin(descs, torch.from_numpy(idx).to(eng.device))
            ps["adam"] = eng.prepare_adam(set().union(*[p[0].touched for p in packed]))
            # ---- algorithmic bytes / flops of this iteration ----
            d = self.d
            direct = bagged = links = 0                            # gradient contributions: on table rows / on bags; word links
            aq = 0.0
            ff = gf = 0.0
            per_row = {}                                           # contributions per table row of the step (the longest gradient list)
            for (f, t, ng, a, w, m) in items:
                n = len(t)
                roles = [(f.target_mode, t), (f.target_mode, ng)] + [(am, a[i]) for i, am in enumerate(f.anchor_modes)]
                for mode, rows in roles:
                    cnt = per_row.setdefault(mode, np.zeros(self.g.table_rows[mode], dtype=np.int64))
                    if mode in bag_len:
                        ptr, ids = self.g.bags[mode]
                        lens = bag_len[mode][rows]
                        flat = np.repeat(ptr[rows].astype(np.int64), lens) + (np.arange(int(lens.sum())) - np.repeat(np.cumsum(lens) - lens, lens))
                        cnt += np.bincount(ids[flat], minlength=len(cnt))
                    else:
                        cnt += np.bincount(rows, minlength=len(cnt))
                    if mode in bag_len:                            # a post = the mean of its word rows
                        words = int(bag_len[mode][rows].sum())
                        bagged += n
                        links += words
                        aq += words * 4 * d + n * (4 * d + 12)     # word rows read, one contribution written, index + ptr pair
                    else:
                        direct += n
                        aq += n * (8 * d + 4)                      # SURVEY.md §8d A_q: row read + gradient write + index
                x, y = mfma_flops(f.query_type, self.decoder, self.inter, d, n)
                ff += x
                gf += y
            keys = ps["adam"]["keys"]
            p_tab = sum(eng.layout.numel(k) for k in keys if k.startswith("enc."))
            p_den = sum(eng.layout.numel(k) for k in keys if not k.startswith("enc."))
            rows_tab = sum(eng.layout.entries[k][1][0] for k in keys if k.startswith("enc."))
            ps["n_entries"], ps["aq_bytes"], ps["fused_flops"], ps["gemm_flops"] = direct + bagged, aq, ff, gf
            ps["longest_list"] = int(max(int(c.max()) for c in per_row.values()))
            ps["rows_over_32"] = int(sum(int((c > 32).sum()) for c in per_row.values()))
            ps["p_touched"] = p_tab + p_den
            ps["p_tab"], ps["rows_tab"] = p_tab, rows_tab
            ps["p_mat"] = sum(eng.layout.numel(k) for k in keys if not k.startswith("enc.") and eng.layout.numel(k) == d * d)
            ps["p_vec"] = p_den - ps["p_mat"]
            # rows of the stepped plain tables the iteration's feed names at least once (the split step: stepped by its second launch)
            ps["named_rows"] = int(sum(int((c > 0).sum()) for mode, c in per_row.items() if mode not in bag_len))
            ps["direct"], ps["links"] = direct, links
            # the fused Adam pass: p, m, v of every table parameter in and out; p, g, m, v in / p, m, v, g := 0 out for the
            # relation / Pre / Post tensors; one list head per table row; per contribution on a row its vector, its link
            # and the head reset; per word link of a bag contribution the vector again, the link and the entry id
            ps["opt_bytes"] = 24.0 * p_tab + 32.0 * p_den + 4.0 * rows_tab + direct * (4.0 * d + 8.0) + links * (4.0 * d + 12.0)
            prepared.append(ps)
        return prepared


EVENT_NOTE = ("avg_launch_ms = mean time between the two hipEvents the library records around a launch on the launch stream: "
              "it contains the launch boundary in front of the kernel (1.5-3 us here, MI355X_MICROARCH.md 'boundary'), so the "
              "three brackets of a step add up to MORE than ms_per_step and every GB/s / TF/s derived from them is a lower bound; "
              "rocprof_avg_launch_ms = the kernel's own duration in the committed rocprofv3 trace of the same command "
              "(profiles/, null for configurations that were not profiled)")


def split_block(eng, used, ms_per_step, d, tag, ms_a, n_a, ms_m, n_m, ms_b, n_b):
    """roofline / kernels / step_roofline of the split step (gqe_train_step, csrc/gqe_split.h).  Its three launches:
      M  gqe_prestep_kernel     Adam on the previous step's d x d matrices + the stamps of the rows this step's feed names
      A  gqe_fused_kernel       the forward / backward tiles | rider workgroups: Adam over every row the feed does not name
      B  gqe_split_rows_kernel  loss finalize + matrix-gradient units | Adam over the named rows (gradient lists) | the vectors
    Bytes each launch has to move (per iteration, averaged over the iterations used):
      A  A_q of the tiles + 24 B per parameter of an unnamed row + 4 B stamp per table row
      B  24 B per parameter of a named row + per contribution its vector, link, head and index + the stamp exchanges +
         32 B per vector parameter + the units' operand rows
      M  32 B per matrix parameter + its two operand-ordered copies (8 B) + index and stamp per feed entry"""
    mean = lambda k: float(np.mean([p[k] for p in used]))
    a_q, ff, gf = mean("aq_bytes"), mean("fused_flops"), mean("gemm_flops")
    rows_tab, named, direct = mean("rows_tab"), mean("named_rows"), mean("direct")
    riders = 24.0 * d * (rows_tab - named) + 4.0 * rows_tab
    a_bytes = a_q + riders
    b_bytes = 24.0 * d * named + direct * (4.0 * d + 8.0 + 4.0 + 4.0 + 8.0) + 32.0 * mean("p_vec") + gf * 4.0 / d
    m_bytes = 40.0 * mean("p_mat") + 8.0 * direct
    a_pass = mean("opt_bytes")                                     # the dense Adam step's bytes, as every other mode prices them
    rp = (lambda k: rocprof_ms(k, tag)) if tag else (lambda k: None)
    rp_a, rp_b, rp_m = rp("gqe_fused_kernel"), rp("gqe_split_rows_kernel"), rp("gqe_prestep_kernel")
    gbs = lambda b, ms: round(b / (ms * 1e-3) / 1e9, 1) if ms and ms > 0 else None
    achieved = gbs(a_bytes, ms_a) or 0.0
    return {
        "roofline": {"bound": "hbm",
                     "kernel": ("gqe_fused_kernel with rider workgroups (gqe_train_step's split step): the forward / backward tiles of the "
                                "iteration + Adam over every table row its batches do not name, in ONE launch; bytes = the tiles' A_q + "
                                "24 B per parameter of the unnamed rows + their stamps"),
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "frac_of_measured_copy_peak": round(achieved / HBM_COPY_GBS, 4), "traffic": None,
                     "algorithmic_bytes_per_launch": a_bytes, "riders_bytes_per_launch": riders, "tiles_bytes_per_launch": a_q,
                     "avg_launch_ms": round(ms_a, 5), "launches": n_a, "rocprof_avg_launch_ms": rp_a,
                     "frac_at_rocprof_duration": round(a_bytes / (rp_a * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if rp_a else None,
                     "regime": ("p / m / v of the stepped tables (%.0f MB) sit inside the 256 MB Infinity Cache: a plain float4 read-modify-write of "
                                "the same footprint reaches 7.7-8.2 TB/s here (tools/probes/rider_probe.hip), so a fraction near 0.8 is the ceiling "
                                "of this regime, not of HBM; beyond the cache (reddit-synth) the optimiser pass is the HBM-bound kernel"
                                % (12.0 * mean("p_tab") / 1e6)),
                     "survey_bytes_per_launch": None},
        "kernels": {"timing_note": EVENT_NOTE,
                    "sum_of_event_brackets_ms": round(ms_a + ms_b + ms_m, 5),
                    "sum_of_rocprof_durations_ms": round(rp_a + rp_b + rp_m, 5) if (rp_a and rp_b and rp_m) else None,
                    "fused_fwd_bwd": {"avg_launch_ms": round(ms_a, 5), "rocprof_avg_launch_ms": rp_a, "launches": n_a, "algorithmic_bytes_per_launch": a_bytes,
                                      "achieved_GBs": achieved, "carries": "rider workgroups: Adam over the rows the iteration does not name",
                                      "mfma_flop_per_launch": ff, "mfma_TFs": round(ff / (ms_a * 1e-3) / 1e12, 2) if ms_a > 0 and ff else None,
                                      "mfma_frac_of_f32_peak": round(ff / (ms_a * 1e-3) / 1e12 / MFMA_F32_TFS, 4) if ms_a > 0 and ff else None},
                    "named_rows_and_pair_gemm": {"kernel": "gqe_split_rows_kernel: loss finalize + matrix-gradient units | Adam over the named rows | vectors",
                                                 "avg_launch_ms": round(ms_b, 5), "rocprof_avg_launch_ms": rp_b, "launches": n_b,
                                                 "algorithmic_bytes_per_launch": b_bytes, "achieved_GBs": gbs(b_bytes, ms_b),
                                                 "mfma_flop_per_launch": gf, "mfma_TFs": round(gf / (ms_b * 1e-3) / 1e12, 2) if ms_b > 0 and gf else None},
                    "pair_gemm": {"rides_in": "named_rows_and_pair_gemm (the split step's second launch): no launch of its own", "mfma_flop_per_launch": gf,
                                  "matrix_step_launch": {"kernel": "gqe_prestep_kernel: Adam on the previous step's d x d matrices + the stamps of this step's named rows",
                                                         "avg_launch_ms": round(ms_m, 5), "rocprof_avg_launch_ms": rp_m, "launches": n_m,
                                                         "algorithmic_bytes_per_launch": m_bytes}}},
        "step_roofline": {"algorithmic_bytes_per_step": a_pass + a_q,
                          "achieved_GBs": round((a_pass + a_q) / (ms_per_step * 1e-3) / 1e9, 1),
                          "frac": round((a_pass + a_q) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
    }


def kernel_block(eng, prepared, used, ms_per_step, lazy=False, world=1, d=128, tag=None, deferred=False, train_step=False):
    """roofline / kernels / step_roofline from the library's hipEvent timings of the timed region.  ``tag``: the workload
    whose committed rocprofv3 trace this configuration corresponds to (None: not profiled).  ``deferred``:
    gqe_set_deferred_gemm is on — where the library let the pair-GEMM units ride in the Adam pass's launch, bracket 2 is that
    launch (gqe_opt_gemm_kernel) and bracket 1 the small launch that steps the d x d matrices behind it."""
    ms_fused, n_fused = eng.timing_read(0)
    ms_gemm, n_gemm = eng.timing_read(1)
    ms_opt, n_opt = eng.timing_read(2)
    if train_step and eng.split_steps() > 0:
        return split_block(eng, used, ms_per_step, d, tag, ms_fused, n_fused, ms_gemm, n_gemm, ms_opt, n_opt)
    rides = bool(deferred) and eng.gemm_rides()
    a_opt = float(np.mean([p["opt_bytes"] for p in used]))
    survey = 32.0 * float(np.mean([p["p_touched"] for p in used]))
    opt_kernel = "gqe_opt_kernel<ADAM, LISTS> (fused Adam over the touched tensors; row gradients from per-row lists)"
    if lazy:                                                       # the row launch: p, m, v of the rows it names + their contributions
        a_opt = float(np.mean([p["n_entries"] for p in used])) * world * (28.0 * d + 12.0)
        survey = None
        opt_kernel = "gqe_rows_kernel (lazy Adam: rows of the step; duplicates counted once per entry)"
    a_q = float(np.mean([p["aq_bytes"] for p in used]))
    ff = float(np.mean([p["fused_flops"] for p in used]))
    gf = float(np.mean([p["gemm_flops"] for p in used]))
    a_pass = a_opt
    if rides:   # the launch also reads every (left, right) scratch row of the step's matrix-gradient jobs once: 8 B per MFMA-contracted pair and column
        a_opt += gf * 4.0 / d
        a_opt -= 32.0 * float(np.mean([p["p_mat"] for p in used]))   # ... and the d x d matrices are stepped by gqe_matstep_kernel, not by this launch
    achieved = a_opt / (ms_opt * 1e-3) / 1e9 if ms_opt > 0 else 0.0

    def tfs(flops, ms):
        return round(flops / (ms * 1e-3) / 1e12, 2) if ms > 0 and flops > 0 else None
    rp = (lambda k: rocprof_ms(k, tag)) if (tag and not lazy and world == 1) else (lambda k: None)
    if rides:
        opt_kernel = ("gqe_opt_gemm_kernel (the Adam pass over the tables and vectors — row gradients from per-row lists — with the "
                      "step's pair-GEMM units and loss finalize in front of its chunks; bytes = the pass's + the units' operand rows)")
    rp_opt, rp_fused = rp("gqe_opt_gemm_kernel" if rides else "gqe_opt_kernel"), rp("gqe_fused_kernel")
    rp_gemm = rp("gqe_matstep_kernel") if rides else rp("gqe_pair_gemm_kernel")
    out = {
        "roofline": {"bound": "hbm", "kernel": opt_kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "frac_of_measured_copy_peak": round(achieved / HBM_COPY_GBS, 4),
                     "traffic": None, "algorithmic_bytes_per_launch": a_opt, "optimiser_pass_bytes_per_launch": a_pass, "survey_bytes_per_launch": survey,
                     "avg_launch_ms": round(ms_opt, 5), "launches": n_opt, "rocprof_avg_launch_ms": rp_opt,
                     "frac_at_rocprof_duration": round(a_opt / (rp_opt * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if rp_opt else None},
        "kernels": {"timing_note": EVENT_NOTE,
                    "sum_of_event_brackets_ms": round(ms_fused + ms_gemm + ms_opt, 5),
                    "sum_of_rocprof_durations_ms": round(rp_opt + rp_fused + rp_gemm, 5) if (rp_opt and rp_fused and rp_gemm) else None,
                    "fused_fwd_bwd": {"avg_launch_ms": round(ms_fused, 5), "rocprof_avg_launch_ms": rp_fused, "launches": n_fused, "algorithmic_bytes_per_launch": a_q,
                                      "achieved_GBs": round(a_q / (ms_fused * 1e-3) / 1e9, 1) if ms_fused > 0 else None,
                                      "mfma_flop_per_launch": ff, "mfma_TFs": tfs(ff, ms_fused),
                                      "mfma_frac_of_f32_peak": round(ff / (ms_fused * 1e-3) / 1e12 / MFMA_F32_TFS, 4) if ms_fused > 0 and ff else None},
                    "pair_gemm": ({"rides_in": "roofline.kernel (gqe_set_deferred_gemm): no launch of its own", "mfma_flop_per_launch": gf,
                                   "matrix_step_launch": ({"kernel": "gqe_matstep_kernel: Adam on the d x d matrices, behind the pass", "avg_launch_ms": round(ms_gemm, 5),
                                                           "rocprof_avg_launch_ms": rp_gemm, "launches": n_gemm} if n_gemm else None)} if rides else
                                  {"avg_launch_ms": round(ms_gemm, 5), "rocprof_avg_launch_ms": rp_gemm, "launches": n_gemm, "mfma_flop_per_launch": gf,
                                   "mfma_TFs": tfs(gf, ms_gemm),
                                   "mfma_frac_of_f32_peak": round(gf / (ms_gemm * 1e-3) / 1e12 / MFMA_F32_TFS, 4) if ms_gemm > 0 and gf else None})},
        "step_roofline": {"algorithmic_bytes_per_step": a_pass + a_q,
                          "achieved_GBs": round((a_pass + a_q) / (ms_per_step * 1e-3) / 1e9, 1),
                          "frac": round((a_pass + a_q) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
    }
    if lazy:
        for k, nm in ((3, "optimiser_other_tables"), (4, "catch_up_before_read")):
            ms, n = eng.timing_read(k)
            out["kernels"][nm] = {"avg_launch_ms": round(ms, 5), "launches": n}
        if eng.gemm_rides() > 0:
            out["kernels"]["pair_gemm"]["note"] = ("gqe_set_deferred_gemm: the pair-GEMM units and the loss finalize ride in the step's row launch "
                                                   "(gqe_rows_ride_kernel); this bracket is the small launch that steps the d x d matrices behind it "
                                                   "(and the separate pair GEMM in front of the periodic full passes)")
    return out


class Loop(object):
    """W warm-up steps, then blocks of exactly K steps (barrier + synchronize on both sides, max over ranks) until
    ``min_seconds`` have been timed; the median block is the result."""

    def __init__(self, eng, dist, world):
        self.eng, self.dist, self.world = eng, dist, world

    def fence(self):
        import torch
        torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
            torch.cuda.synchronize()

    def run(self, step, warmup, steps, min_seconds=0.5, max_blocks=400):
        import torch
        eng = self.eng
        # hipEvent pairs around every 64th launch of each kernel (recycled; > 100 samples of the dominant launch in the 0.5 s that
        # are timed).  Denser sampling perturbs what it measures: every 4th launch cost 4 us of a 88 us step
        # (profiles/r02_experiment_event_stride.log), every 16th still 1.1 us of this round's 71 us step (stride 16 / 64 / 256:
        # 71.6 / 70.5 / 70.2 us per step, two runs each on one box)
        # (several ranks: every 16th — their steps are longer, their timed regions hold fewer of them)
        # (... and never sparser than half a block: a 5-step smoke run of 8 ranks sharing one GPU may time a single block)
        eng.timing_enable(int(os.environ.get("GQE_BENCH_EVENT_STRIDE", "64" if self.world == 1 else str(min(16, max(1, steps // 2))))))
        for i in range(warmup):
            step(i)
        self.fence()
        for k in range(7):
            eng.timing_read(k)
        times, i = [], warmup
        while True:
            self.fence()
            t0 = time.perf_counter()
            for _ in range(steps):
                step(i)
                i += 1
            eng.sync()                                             # lazy Adam: deferred steps are settled inside the timed region
            self.fence()
            el = time.perf_counter() - t0
            if self.dist is not None:
                tmax = torch.tensor([el], dtype=torch.float64, device=eng.device)
                self.dist.all_reduce(tmax, op=self.dist.ReduceOp.MAX)
                el = float(tmax.item())
            times.append(el)
            if sum(times) >= min_seconds or len(times) >= max_blocks:
                break
        self.last_step = i - 1
        return times


def summarize(times, steps):
    med = float(np.median(times))
    return med, {"blocks": len(times), "block_ms_min": round(min(times) * 1e3, 4), "block_ms_median": round(med * 1e3, 4),
                 "block_ms_max": round(max(times) * 1e3, 4), "steps_per_block": steps}


def make_step(eng, prepared, dist, exchange, n_distinct, ex_events=None, train_step=False):
    from graphqembed_amd import parallel

    def step(i):
        ps = prepared[i % n_distinct]
        if train_step:                                             # the whole iteration as one library call (gqe_train_step)
            eng.run_train_step(ps, ps["adam"])
            return
        eng.run_margin(ps)
        if eng.lazy_adam and dist is None:                         # the next iteration's rows ride in this step's row launch
            eng.lazy_prefetch(prepared[(i + 1) % n_distinct])
        if dist is not None:
            ev = {} if (ex_events is not None and (i % 4) == 0) else None   # per-phase events on every 4th step
            if exchange == "sparse":                               # contribution entries all-gathered over xGMI
                parallel.exchange_sparse(eng, dist, events=ev)
            elif exchange == "dense":                              # lists -> dense arena, RCCL sum over xGMI
                parallel.exchange_gradients(eng.grads, dist, engine=eng, events=ev)
            else:                                                  # contributions to the rows' owners, small tensors all-reduced
                parallel.shard_exchange(eng, dist, ps)
            if ev:
                ex_events.append(ev)
        eng.run_adam(ps["adam"])

    posted = {"next": None}

    def sharded_step(i):
        """Row-sharded: step i + 1 is PLANNED here, inside the timed loop (gqe_shard_post: owner sort of its host feed,
        publication on the shared-memory plan board), then gqe_shard_step runs step i: serve -> all-to-all of rows -> fused
        forward / backward + pair GEMM -> all-to-all of contributions -> link -> all-reduce of the small gradients -> Adam
        on the own shards, one library call."""
        if posted["next"] != i:
            eng.shard_post(prepared[i % n_distinct])
        eng.shard_post(prepared[(i + 1) % n_distinct])
        posted["next"] = i + 1
        eng.shard_step(prepared[i % n_distinct])
    return sharded_step if exchange == "sharded" and dist is not None else step


def measure(wl, args, dist, rank, world, exchange="sparse", lazy=False, steps=None, warmup=None, min_seconds=0.5, check_replicas=False):
    """One full measurement of a workload on this process group: returns the result dict (valid on every rank)."""
    import torch
    from graphqembed_amd import parallel
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    sparse = world > 1 and exchange == "sparse"
    sharded = world > 1 and exchange == "sharded"
    eng = wl.engine(rank=rank if sparse else 0, world=world if sparse else 1, lazy=lazy, shard=(rank, world) if sharded else None)
    # one GPU, eager Adam: the step is margin_fwd_bwd + adam_step back to back and the losses are read behind the loop, so the
    # matrix-gradient units may ride in the Adam pass's launch (include/gqe.h, gqe_set_deferred_gemm; the library falls back
    # to the separate launch wherever riding does not apply: lazy Adam, tables beyond the Infinity Cache, > 1024 units)
    deferred = world == 1 and not lazy and os.environ.get("GQE_BENCH_NO_DEFERRED_GEMM") is None
    # ... and as ONE call per iteration, gqe_train_step: the library then runs Adam over the rows the batches do not name inside
    # the fused launch (the split step, csrc/gqe_split.h) wherever that applies, and the two-call sequence elsewhere.
    # GQE_BENCH_TWO_CALLS=1 measures gqe_margin_fwd_bwd + gqe_adam_step (round 4's headline step).
    train_step = deferred and os.environ.get("GQE_BENCH_TWO_CALLS") is None
    if deferred or (lazy and world == 1 and os.environ.get("GQE_BENCH_NO_DEFERRED_GEMM") is None):
        eng.set_deferred_gemm(True)                                # (lazy Adam: the units ride in the step's row launch)
    prepared = wl.prepare(eng, dist)
    session = parallel.shard_session(eng, dist, rank, world) if sharded else None
    ex_events = [] if (dist is not None and not sharded) else None
    step = make_step(eng, prepared, dist, exchange, wl.n_distinct, ex_events, train_step=train_step)
    loop = Loop(eng, dist, world)
    times = loop.run(step, warmup, steps, min_seconds=min_seconds)
    med, blocks = summarize(times, steps)
    ms_per_step = med * 1e3 / steps
    used = prepared[:min(steps, wl.n_distinct)]
    out = {"value": round(steps * wl.qpi * world / med, 1), "unit": "queries/s", "ms_per_step": round(ms_per_step, 4), "timing": blocks}
    profiled = (wl.d, wl.B, wl.decoder, wl.inter, wl.zipf, len(wl.mix)) == ((256 if wl.name == "reddit-synth" else 128), 512, "bilinear-diag", "min", None, 9)
    out.update(kernel_block(eng, prepared, used, ms_per_step, lazy=lazy, world=world if sparse else 1, d=wl.d,
                            tag=wl.name if (profiled and world == 1) else None, deferred=deferred, train_step=train_step))
    out["step_form"] = ("split" if eng.split_steps() > 0 else "two-call sequence inside gqe_train_step") if train_step else "two calls"
    eng.sync()
    out["longest_gradient_list"] = int(max(p["longest_list"] for p in used))
    out["rows_with_over_32_contributions"] = int(max(p["rows_over_32"] for p in used))
    eng.timing_enable(0)
    loss = float(prepared[loop.last_step % wl.n_distinct]["losses"][-1].item())
    if not np.isfinite(loss):
        raise SystemExit("non-finite loss")
    out["final_loss"] = round(loss, 6)
    if sharded:                                                    # hipEvents of the library around the two exchange phases (every 16th step)
        ms_rows, _ = eng.timing_read(5)
        ms_contrib, _ = eng.timing_read(6)
        out["exchange_ms_per_step"] = round(ms_rows + ms_contrib, 4)
        out["exchange_parts_ms"] = {"serve_and_fetch_rows": round(ms_rows, 4), "contributions_link_and_small_tensors": round(ms_contrib, 4)}
        out["planning"] = "inside the timed region: every step plans the next one on the host (gqe_shard_post)"
        if getattr(session, "error", None) is not None:
            raise session.error
    if ex_events:
        torch.cuda.synchronize()
        ev = ex_events[len(ex_events) // 5:]
        order = ["start", "exported", "gathered", "imported"] if exchange == "sparse" else ["start", "materialized", "reduced"]
        names = {"exported": "export_entries", "gathered": "all_gather_of_slabs", "imported": "import_entries",
                 "materialized": "lists_to_dense_gradient", "reduced": "all_reduce_of_the_arena"}
        out["exchange_ms_per_step"] = round(float(np.mean([x[order[0]][0].elapsed_time(x[order[-1]][0]) for x in ev])), 4)
        out["exchange_parts_ms"] = {names[b]: round(float(np.mean([x[a][0].elapsed_time(x[b][0]) for x in ev])), 4)
                                    for a, b in zip(order[:-1], order[1:])}
    if dist is not None:
        ones = torch.ones(1, device=eng.device)
        dist.all_reduce(ones)
        out["ranks_seen"] = int(ones.item())
        if check_replicas:
            # replicated state must be bit-identical on every rank: everything, or (row-sharded) the relation / Pre / Post tensors
            mine = torch.cat([eng.params[o:o + n] for o, n in eng.dense_spans()] +
                             [eng.layout.view(eng.params, k).reshape(-1) for k in eng.bag_keys]) if sharded else eng.params
            ref = mine.clone()
            dist.broadcast(ref, 0)
            same = torch.tensor([int(torch.equal(ref, mine))], device=eng.device)
            dist.all_reduce(same, op=dist.ReduceOp.MIN)
            out["replicas_identical"] = bool(same.item() == 1)
    return out, eng, prepared


def measure_or_fall_back(wl, args, dist, rank, world, exchange, **kw):
    """measure(); when the row-sharded session cannot be brought up on this node (plan board, the library's own RCCL
    communicator) every rank agrees on it and the replicated sparse exchange is measured instead — the line then says so
    (`fell_back`).  Only clean errors can be agreed on: a rank that fails makes the others' plan wait time out first."""
    if world == 1 or exchange != "sharded":
        return measure(wl, args, dist, rank, world, exchange=exchange, **kw) + (None,)
    import torch
    err, got = None, None
    try:
        if os.environ.get("GQE_BENCH_DEBUG_FAIL_SHARDED") in (str(rank), "all"):   # exercises this path (tests)
            raise RuntimeError("GQE_BENCH_DEBUG_FAIL_SHARDED")
        got = measure(wl, args, dist, rank, world, exchange=exchange, **kw)
    except Exception as e:                                          # noqa: BLE001 - reported in the line
        err = "%s: %s" % (type(e).__name__, e)
    ok = torch.tensor([0 if err else 1], device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 1:
        return got + (None,)
    if got is not None:
        got[1].close()
    errs = [None] * world
    dist.all_gather_object(errs, err)
    why = next((e for e in errs if e), "unknown")
    sys.stderr.write("bench.py: row-sharded step unavailable (%s): measuring the sparse exchange\n" % why)
    return measure(wl, args, dist, rank, world, exchange="sparse", **kw) + (why[:300],)


def host_fed(wl, args):
    """The same schedule driven by the native feeder (gqe_feeder_run): formula draw, wrap-around slice, negative draw and
    packing on a host core, launches and the Adam step from C++ — everything inside the timed region (SURVEY.md §8f-3,
    north star).  The index feed reaches the kernels either straight from pinned host memory (default: the kernels
    read it over PCIe; no copy, no cross-stream dependency) or through the pinned staging ring + hipMemcpyAsync on the
    library's upload stream (``pinned_hipMemcpyAsync``)."""
    import torch
    from graphqembed_amd.tensorize import FormulaPlan, table_key
    out = None
    for feed in ("zero-copy", "copy", "lazy"):
        eng = wl.engine(lazy=(feed == "lazy"))
        if os.environ.get("GQE_BENCH_NO_DEFERRED_GEMM") is None:
            eng.set_deferred_gemm(True)                            # (losses are read behind feeder_run)
        plist = []
        for t in wl.types:
            for p in wl.pools[t]:
                plist.append((FormulaPlan(p.formula, wl.layout, wl.inter), p))
        all_rows = {table_key(m): np.arange(1, wl.g.mode_sizes[m] + 1, dtype=np.int32) for m in wl.g.modes}
        feeder = eng.make_feeder(plist, all_rows, batch_size=wl.B, seed=0, feed="zero-copy" if feed == "lazy" else feed)
        eng.feeder_run(feeder, 0, max(args.warmup, 10))
        torch.cuda.synchronize()
        times, it = [], max(args.warmup, 10)
        while sum(times) < 0.5 and len(times) < 200:
            t0 = time.perf_counter()
            losses = eng.feeder_run(feeder, it, args.steps)
            eng.sync()                                             # lazy Adam: the deferred steps are settled inside the timed region
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            it += args.steps
        med, blocks = summarize(times, args.steps)
        n_batches = 1 + sum(2 if "inter" in t else 1 for t in wl.types if t != "1-chain")
        res = {"value": round(args.steps * n_batches * wl.B / med, 1), "unit": "queries/s", "ms_per_step": round(med * 1e3 / args.steps, 4),
               "timing": blocks, "final_loss": round(float(losses[n_batches].item()), 6)}
        eng.feeder_destroy(feeder)
        eng.close()
        if out is None:
            out = res
            out["feed"] = "pinned host memory read by the kernels (gqe_feeder_set_feed 1)"
            out["note"] = ("gqe_feeder_run: per iteration the host draws a formula per batch (prob ~ pool size), slices it by the "
                           "reference's wrap-around rule, draws 1-chain negatives and packs the index feed into a pinned slot")
        elif feed == "copy":
            res["note"] = ("pinned staging + hipMemcpyAsync on the library's upload stream (the transport north_star names): the feeds of "
                           "EIGHT iterations travel per copy / per pair of cross-stream events")
            out["pinned_hipMemcpyAsync"] = res
        else:
            res["note"] = ("lazy (deferred, bit-exact) Adam driven by the native feeder: every step names the next iteration's feed "
                           "(gqe_lazy_prefetch), so one row launch covers rows(t) and rows(t+1); NON-DEFAULT mode, not the headline")
            out["lazy_exact_adam"] = res
    return out


def api_path(args, d, decoder, inter, B, iterations=120):
    """The DROP-IN path itself, timed: ``train_helpers.run_train`` (the reference's loop, train_helpers.py:40-107) with a
    ``FusedAdam`` on ``Query`` objects — a bio-synth-sized graph built through the reference-shaped ``Graph`` /
    ``QueryEncoderDecoder`` classes, query lists sampled by the native sampler and converted ONCE to ``Query`` objects.
    (1) as ``run_train`` runs it: the iterations between two events of its schedule are ONE native call each
    (train_helpers._NativeLoop -> gqe_feeder_run with reference streams: the reference's formula draws and negatives replayed on
    np.random's / random's generators); (2) ``per_batch_python_path``: GQE_RUN_TRAIN_NATIVE=0 — every batch drawn and packed from
    Python, one gqe_train_step and one ``loss.item()`` per iteration — with the host's share per iteration."""
    import random
    import torch
    from graphqembed_amd import data_utils, train_helpers, utils
    from graphqembed_amd import model as model_mod
    from graphqembed_amd.graph import Graph, Query
    from graphqembed_amd.model import FusedAdam, QueryEncoderDecoder
    from graphqembed_amd.sampler import NativeSampler
    t_build = time.perf_counter()
    rel, adj, ids = data_utils.make_synthetic_graph(data_utils.BIO_SYNTH_SIZES, seed=0)
    node_maps = data_utils.make_node_maps(ids)
    dims = {m: d for m in rel}
    graph = Graph(None, dims, rel, adj)
    feats = {m: torch.nn.Embedding(len(node_maps[m]) + 1, d) for m in rel}
    for f in feats.values():
        f.weight.data.normal_(0, 1.0 / d)
    enc = utils.get_encoder(0, graph, dims, feats, True, node_maps=node_maps)
    model = QueryEncoderDecoder(graph, enc, utils.get_metapath_decoder(graph, dims, decoder), utils.get_intersection_decoder(graph, dims, inter),
                                max_queries=9 * B, max_batches=9)
    sampler = NativeSampler(graph, node_maps)
    train = {}
    edges = graph.get_all_edges(seed=0)[:60000]
    train["1-chain"] = dict(data_utils.group_by_formula([Query(("1-chain", e), None, None) for e in edges])["1-chain"])
    for k, t in enumerate(["2-chain", "3-chain", "2-inter", "3-inter", "3-inter_chain"]):
        by = data_utils.group_by_formula(sampler.sample(40000, q_type=t, neg_sample_max=20, seed=k, threads=8).to_queries(keep_graph=False))[t]
        keep = sorted(by, key=lambda f: -len(by[f]))[:6]          # the six most frequent formulas: lists of >= one batch
        train[t] = {f: by[f] for f in keep}
    # validation sets: 250 queries of each of the four most frequent formulas of every sampled type (<= 20 negatives per query)
    held = {t: {f: qs[:250] for f, qs in list(train[t].items())[:4]} for t in train if t != "1-chain"}
    test = {"one_neg": held, "full_neg": held}
    t_build = time.perf_counter() - t_build
    class Quiet(object):
        def info(self, m):
            pass

    # ---- (1) the loop as run_train runs it: its iterations between two events executed natively (train_helpers._NativeLoop) ----
    opt = FusedAdam(model, lr=0.01)
    runs = []
    orig_run = train_helpers._NativeLoop.run

    def spy_run(self, first, n, all_types):
        q0 = self.model.engine.feeder_queries(self.feeder)
        h0 = self.model.engine.feeder_host_seconds(self.feeder)
        t0 = time.perf_counter()
        res = orig_run(self, first, n, all_types)             # (returns behind the copy of the run's loss history: the device is idle)
        dt = time.perf_counter() - t0
        h1 = self.model.engine.feeder_host_seconds(self.feeder)
        runs.append((n, all_types, dt, self.model.engine.feeder_queries(self.feeder) - q0, h1[0] - h0[0], h1[1] - h0[1]))
        return res
    train_helpers._NativeLoop.run = spy_run
    evals = []
    orig_eval = train_helpers.evaluate

    def spy_eval(*a, **kw):
        t0 = time.perf_counter()
        res = orig_eval(*a, **kw)
        evals.append(time.perf_counter() - t0)
        return res
    train_helpers.evaluate = spy_eval
    random.seed(0); np.random.seed(0)
    t_run = time.perf_counter()
    try:
        train_helpers.run_train(model, opt, train, test, test, Quiet(), max_burn_in=2, batch_size=B, log_every=100, val_every=500,
                                max_iter=2001)
    finally:
        train_helpers._NativeLoop.run = orig_run
        train_helpers.evaluate = orig_eval
    t_run = time.perf_counter() - t_run
    full = [r for r in runs if r[1]][1:]                       # steady state: every query type, behind the first such run
    n_it, dt, q = sum(r[0] for r in full), sum(r[2] for r in full), sum(r[3] for r in full)
    out = {"value": round(q / dt, 1), "unit": "queries/s", "iterations": n_it, "ms_per_iteration": round(dt / n_it * 1e3, 4),
           "queries_per_iteration": round(q / n_it, 1), "split_steps": model.engine.split_steps(),
           "host_us_per_iteration": {"sampling_and_packing": round(sum(r[4] for r in full) / n_it * 1e6, 1), "inside_gqe_feeder_run": round(sum(r[5] for r in full) / n_it * 1e6, 1),
                                     "note": "one host thread (gqe_feeder_host_seconds): the reference's formula draws and negatives replayed on its MT19937 streams, packing, "
                                             "pinned upload and the three launches of every iteration; ms_per_iteration above is wall time — the loop is bound by the device when "
                                             "the two agree.  queries_per_iteration < 9 x B: the reference's wrap-around windows (train_helpers.py:102-105) cut batches short at a list's end"},
           "native_runs": [{"iterations": r[0], "all_types": bool(r[1]), "ms": round(r[2] * 1e3, 3)} for r in runs],
           "run_train_seconds": round(t_run, 2), "setup_seconds": round(t_build, 1),
           "validation": {"queries": sum(len(q) for by in held.values() for q in by.values()), "calls": len(evals),
                          "first_call_ms": round(evals[0] * 1e3, 1) if evals else None,
                          "later_calls_ms": round(float(np.median(evals[1:])) * 1e3, 1) if len(evals) > 1 else None,
                          "note": "train_helpers.evaluate (AUC with one negative + percentile over all negatives per query type, hard "
                                  "variants for intersections): the lists' rows are looked up at the first call (model.pool_rows), "
                                  "later calls work on arrays - formulas grouped per launch, one read-back per statistic"},
           "note": ("train_helpers.run_train + FusedAdam on Query objects (the reference's loop and signatures; 2 001 iterations, log lines "
                    "every 100, validation every 500): the iterations between two events of the schedule run as ONE library call "
                    "(gqe_feeder_run with reference streams: formula draws replayed on np.random's generator, negatives on random's, "
                    "packing, gqe_train_step) - the same batches as the reference's loop under the same seeds "
                    "(tests/test_gpu_api.py::test_run_train_native_runs_reproduce_the_reference_run); the phase switch, validation, "
                    "moving average and log lines stay in Python; value = queries / wall time of the native runs behind the first "
                    "(each ends with the copy of its loss history: the device is idle at both ends)")}

    # ---- (1b) the same call on a model built with lazy_adam=True (deferred, bit-exact Adam: NON-DEFAULT mode, as lazy_exact_adam) ----
    if not args.no_lazy:
        feats_l = {m: torch.nn.Embedding(len(node_maps[m]) + 1, d) for m in rel}
        for f in feats_l.values():
            f.weight.data.normal_(0, 1.0 / d)
        enc_l = utils.get_encoder(0, graph, dims, feats_l, True, node_maps=node_maps)
        model_l = QueryEncoderDecoder(graph, enc_l, utils.get_metapath_decoder(graph, dims, decoder), utils.get_intersection_decoder(graph, dims, inter),
                                      max_queries=9 * B, max_batches=9, lazy_adam=True)
        runs_l = []

        def spy_run_l(self, first, n, all_types):
            q0 = self.model.engine.feeder_queries(self.feeder)
            t0 = time.perf_counter()
            res = orig_run(self, first, n, all_types)
            self.model.engine.sync()                            # (the deferred steps are settled inside the timed region)
            torch.cuda.synchronize()
            runs_l.append((n, all_types, time.perf_counter() - t0, self.model.engine.feeder_queries(self.feeder) - q0))
            return res
        train_helpers._NativeLoop.run = spy_run_l
        random.seed(0); np.random.seed(0)
        try:
            train_helpers.run_train(model_l, FusedAdam(model_l, lr=0.01), train, test, test, Quiet(), max_burn_in=2, batch_size=B, log_every=100,
                                    val_every=500, max_iter=2001)
        finally:
            train_helpers._NativeLoop.run = orig_run
        full_l = [r for r in runs_l if r[1]][1:]
        out["lazy_exact_adam"] = {"value": round(sum(r[3] for r in full_l) / sum(r[2] for r in full_l), 1), "unit": "queries/s",
                                  "ms_per_iteration": round(sum(r[2] for r in full_l) / sum(r[0] for r in full_l) * 1e3, 4),
                                  "note": "QueryEncoderDecoder(..., lazy_adam=True): the same run_train call, NON-DEFAULT mode"}
        model_l.engine.close()

    # ---- (1c) --opt sgd (bio/train.py:59-60): the same call with a FusedSGD on the first model (its parameters keep training) ----
    from graphqembed_amd.model import FusedSGD
    runs_s = []

    def spy_run_s(self, first, n, all_types):
        q0 = self.model.engine.feeder_queries(self.feeder)
        t0 = time.perf_counter()
        res = orig_run(self, first, n, all_types)
        runs_s.append((n, all_types, time.perf_counter() - t0, self.model.engine.feeder_queries(self.feeder) - q0))
        return res
    train_helpers._NativeLoop.run = spy_run_s
    random.seed(0); np.random.seed(0)
    try:
        train_helpers.run_train(model, FusedSGD(model, lr=0.01), train, test, test, Quiet(), max_burn_in=2, batch_size=B, log_every=100,
                                val_every=500, max_iter=1501)
    finally:
        train_helpers._NativeLoop.run = orig_run
    full_s = [r for r in runs_s if r[1]][1:]
    out["opt_sgd"] = {"value": round(sum(r[3] for r in full_s) / sum(r[2] for r in full_s), 1), "unit": "queries/s",
                      "ms_per_iteration": round(sum(r[2] for r in full_s) / sum(r[0] for r in full_s) * 1e3, 4),
                      "note": "run_train with FusedSGD (torch.optim.SGD, momentum 0: the reference's --opt sgd): fused forward / backward, pair GEMM, "
                              "and a pass that touches only the rows with a gradient"}

    # ---- (2) the same loop batch by batch from Python (GQE_RUN_TRAIN_NATIVE=0; what round 5's first half measured) ----
    opt = FusedAdam(model, lr=0.01)
    clock = {"lookup": 0.0, "pack": 0.0, "launch": 0.0, "n": 0, "queries": 0, "stamps": []}

    def timed(owner, name, key, count=None):
        fn = getattr(owner, name)

        def wrapper(*a, **kw):
            t0 = time.perf_counter()
            out = fn(*a, **kw)
            clock[key] += time.perf_counter() - t0
            if count:
                count(a, out)
            return out
        setattr(owner, name, wrapper)
        return fn
    undo = [(train_helpers.FusedExecutor, "add_window", timed(train_helpers.FusedExecutor, "add_window", "lookup",
                                                             lambda a, out: clock.__setitem__("queries", clock["queries"] + (a[4] - a[3])))),
            (model_mod, "pack_margin_batches", timed(model_mod, "pack_margin_batches", "pack")),
            (model.engine, "train_step", timed(model.engine, "train_step", "launch"))]
    step0 = opt.step

    def step():
        step0()
        clock["stamps"].append((time.perf_counter(), clock["lookup"], clock["pack"], clock["launch"], clock["queries"]))
    opt.step = step
    random.seed(0); np.random.seed(0)
    os.environ["GQE_RUN_TRAIN_NATIVE"] = "0"
    try:
        train_helpers.run_train(model, opt, train, test, test, Quiet(), max_burn_in=2, batch_size=B, log_every=10 ** 9, val_every=10 ** 9,
                                max_iter=iterations)
    finally:
        os.environ.pop("GQE_RUN_TRAIN_NATIVE", None)
        for owner, name, fn in undo:
            setattr(owner, name, fn)
    st = clock["stamps"]
    a, b = st[len(st) // 3], st[-1]                                 # steady state: the last two thirds (all query types)
    n = len(st) - 1 - len(st) // 3
    dt = b[0] - a[0]
    q = b[4] - a[4]
    out["per_batch_python_path"] = {
        "value": round(q / dt, 1), "unit": "queries/s", "iterations": n, "ms_per_iteration": round(dt / n * 1e3, 4),
        "host_us_per_iteration": {"row_lookup_and_negative_draw": round((b[1] - a[1]) / n * 1e6, 1), "packing": round((b[2] - a[2]) / n * 1e6, 1),
                                  "library_call": round((b[3] - a[3]) / n * 1e6, 1),
                                  "rest (formula draw, loss .item() = waiting for the device, loop)": round((dt - (b[1] - a[1]) - (b[2] - a[2]) - (b[3] - a[3])) / n * 1e6, 1)},
        "note": "GQE_RUN_TRAIN_NATIVE=0: every batch drawn and packed from Python, one gqe_train_step and one loss.item() per iteration"}
    model.engine.close()
    return out


def cpu_baseline(eng, decoder, inter, item_sets, budget_s, queries_per_iter):
    """Time the torch-CPU port on the same parameters / batches (bounded sample).  torch's default
    (one thread per hardware thread) is far from its best on a 2x64-core host, so a short sweep over
    thread counts picks the fastest setting and the remaining budget is spent measuring it."""
    import torch
    from oracle.netquery_numpy import make_plan
    from oracle.netquery_torch import TorchPort
    torch.cuda.synchronize()
    host = eng.params.cpu().numpy()
    params = {k: host[off:off + int(np.prod(shape))].reshape(shape).copy() for k, (off, shape) in eng.layout.entries.items()}
    port = TorchPort(params, decoder, inter)
    sets = [[(make_plan(f.query_type, f.rels), t, g, a, w, m) for (f, t, g, a, w, m) in items] for items in item_sets]
    default_threads = torch.get_num_threads()
    try:
        import psutil
        physical = int(psutil.cpu_count(logical=False) or default_threads)
    except Exception:                                              # noqa: BLE001
        physical = default_threads
    port.train_iteration(sets[0])                                  # warm-up (allocator, Adam state)

    def rate(nt, iters):
        torch.set_num_threads(nt)
        port.train_iteration(sets[0])
        t0 = time.time()
        for k in range(iters):
            port.train_iteration(sets[(k + 1) % len(sets)])
        return (time.time() - t0) / iters
    # the port at n = 1 and at n = all physical cores, stated next to the best of a short sweep (torch's
    # default — one thread per hardware thread — is far from its best on a 2 x 64-core host)
    fixed = {}
    for nt in sorted(set([1, physical])):
        dt = rate(nt, 2)
        fixed[nt] = {"threads": nt, "value": round(queries_per_iter / dt, 1), "unit": "queries/s", "iterations": 2}
    best = (None, 1e30)
    for nt in sorted(set([8, 16, 32, 64, default_threads])):
        if nt > default_threads:
            continue
        dt = rate(nt, 2)
        if dt < best[1]:
            best = (nt, dt)
    for nt, r in fixed.items():
        if queries_per_iter / best[1] < r["value"]:
            best = (nt, queries_per_iter / r["value"])
    torch.set_num_threads(best[0])
    n, t0 = 0, time.time()
    while True:
        port.train_iteration(sets[(n + 1) % len(sets)])
        n += 1
        el = time.time() - t0
        if el >= budget_s or n >= 400:
            break
    torch.set_num_threads(default_threads)
    return {"value": round(n * queries_per_iter / el, 1), "unit": "queries/s", "cores": int(best[0]),
            "kind": "port", "sample": "%d full-mix iterations (9x512 queries, P=%d, dense torch Adam) in %.1f s with %d threads "
            "(best of a sweep over 1 / 8 / 16 / 32 / 64 / %d physical cores / %d hardware threads); oracle/netquery_torch.py, torch %s"
            % (n, eng.layout.total, el, best[0], physical, default_threads, torch.__version__),
            "one_thread": fixed[1], "all_physical_cores": fixed[physical], "physical_cores": physical, "hardware_threads": default_threads}


def slim(res):
    """A secondary measurement as it appears in the JSON line: the headline figures + per-kernel launch times."""
    k = res["kernels"]
    out = {"value": res["value"], "unit": "queries/s", "ms_per_step": res["ms_per_step"], "timing": res["timing"],
           "final_loss": res["final_loss"], "step_form": res.get("step_form"),
           "kernels_ms": {("matrix_step (pair GEMM / loss finalize ride in the optimiser launch)" if "matrix_step_launch" in v else name):
                          ((v["matrix_step_launch"] or {"avg_launch_ms": None}) if "matrix_step_launch" in v else v)["avg_launch_ms"]
                          for name, v in k.items() if isinstance(v, dict)},
           "longest_gradient_list": res.get("longest_gradient_list"),
           "optimiser": {"avg_launch_ms": res["roofline"]["avg_launch_ms"], "achieved_GBs": res["roofline"]["achieved"],
                         "frac": res["roofline"]["frac"], "algorithmic_bytes_per_launch": res["roofline"]["algorithmic_bytes_per_launch"]},
           "fused_mfma_TFs": k["fused_fwd_bwd"]["mfma_TFs"], "pair_gemm_mfma_TFs": k["pair_gemm"].get("mfma_TFs"),
           "step_roofline": res["step_roofline"]}
    for key in ("exchange_ms_per_step", "ranks_seen", "replicas_identical"):
        if key in res:
            out[key] = res[key]
    return out


def self_launch(args, argv):
    """--gpus N > 1 without a launcher environment: become the launcher (one rank per GPU, RCCL)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")             # dmabuf IPC (RCCL / cross-process device memory)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def main():
    from graphqembed_amd import synth
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch-size", type=int, default=512)
    ap.add_argument("--dim", type=int, default=None, help="embedding dimension (default: 128 bio-synth, 256 reddit-synth)")
    ap.add_argument("--workload", default="bio-synth", choices=["bio-synth", "reddit-synth"],
                    help="main measurement: BASELINE's metric is quoted on bio-synth; reddit-synth is BASELINE config 5")
    ap.add_argument("--decoder", default="bilinear-diag")
    ap.add_argument("--inter-decoder", default="min")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only to "
                    "smoke-test the multi-rank path when several ranks share one GPU)")
    ap.add_argument("--exchange", default="sharded", choices=["sharded", "sparse", "dense"],
                    help="--gpus > 1, main measurement: row-sharded tables with owner-computes Adam (sharded: rows and gradient "
                    "contributions travel by all-to-all, optimiser bytes per rank fall as 1/N), replicated tables with an all-gather of "
                    "the gradient contribution entries (sparse), or with an all-reduce of the dense arena (dense); the other "
                    "forms are measured next to it (shorter)")
    ap.add_argument("--lazy-adam", action="store_true", help="run the MAIN measurement in lazy-Adam mode (non-default; the config "
                    "then says so).  Works with --gpus N and the sparse exchange.")
    ap.add_argument("--no-lazy", action="store_true", help="skip the secondary measurement of the lazy (deferred, bit-exact) Adam mode")
    ap.add_argument("--no-configs", action="store_true", help="skip the other SURVEY §8d configurations (N=1)")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the gqe_feeder_run measurement (N=1)")
    ap.add_argument("--no-reddit", action="store_true", help="skip the secondary reddit-synth d=256 measurement")
    ap.add_argument("--reddit-scale", type=float, default=None, help="shrink the reddit-synth world (nodes, edges, words) by this factor: "
                    "functional runs of the config-5 workload with many ranks sharing one GPU — the line says so, it is no measurement")
    ap.add_argument("--no-api-path", action="store_true", help="skip the measurement of the reference-shaped API (run_train on Query objects, N=1)")
    ap.add_argument("--only-main", action="store_true", help="main measurement only (profiling runs)")
    ap.add_argument("--min-seconds", type=float, default=0.5, help="timed blocks of --steps steps repeat until this much time was "
                    "measured (profiling runs: 0 = one block)")
    ap.add_argument("--check-replicas", action="store_true", help="kept for compatibility: replicas are always compared for --gpus > 1")
    args = ap.parse_args()
    if args.only_main:
        args.no_lazy = args.no_configs = args.no_host_fed = args.no_reddit = args.no_cpu_baseline = args.no_api_path = True

    if args.reddit_scale is not None:
        os.environ["GQE_BENCH_REDDIT_SCALE"] = repr(args.reddit_scale)      # (inherited by the ranks of a self-launch)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, sys.argv[1:]))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%s" % (args.gpus, os.environ.get("WORLD_SIZE", "1")))
    # stdout carries exactly ONE line, the JSON result: whatever libraries print on file descriptor 1 while they initialise
    # (RCCL's version banner, gloo's "Rank i is connected" lines) is sent to stderr instead
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    from graphqembed_amd import parallel
    n_dev = torch.cuda.device_count()
    backend, backend_note = args.backend, None
    if args.gpus > 1 and backend == "nccl" and n_dev < args.gpus:
        backend = "gloo"                                           # RCCL cannot put two ranks on one device
        backend_note = "%d ranks share %d GPU(s): gloo instead of RCCL (functional run, not a scaling measurement)" % (args.gpus, n_dev)
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(n_dev, 1))
    rank, world, local_rank, dist = parallel.init_from_env(backend)

    reddit = args.workload == "reddit-synth"
    d = args.dim or (256 if reddit else 128)
    B = args.batch_size
    wl = Workload(args.workload, d, args.decoder, args.inter_decoder, synth.FULL_MIX, B, rank=rank, world=world)
    res, eng, prepared, fell_back = measure_or_fall_back(wl, args, dist, rank, world, exchange=args.exchange, lazy=args.lazy_adam,
                                                         check_replicas=world > 1, min_seconds=args.min_seconds)
    if fell_back:
        args.exchange = "sparse"
    sparse = world > 1 and args.exchange == "sparse"
    sharded = world > 1 and args.exchange == "sharded"
    res["roofline"]["traffic"] = None if (world > 1 or args.lazy_adam or (d, B, args.decoder, args.inter_decoder) != ((256 if reddit else 128), 512, "bilinear-diag", "min")) \
        else pmc_traffic("gqe_fused_kernel" if res["roofline"]["kernel"].startswith("gqe_fused_kernel") else
                         "gqe_opt_gemm_kernel" if res["roofline"]["kernel"].startswith("gqe_opt_gemm_kernel") else "gqe_opt_kernel", args.workload)
    if res["roofline"].get("traffic") and res["roofline"].get("avg_launch_ms"):
        # what the memory system actually carried during the launch (PMC bytes of the last profiled run / this run's launch time)
        tr = res["roofline"]["traffic"] / (res["roofline"]["avg_launch_ms"] * 1e-3) / 1e9
        res["roofline"]["traffic_GBs"] = round(tr, 1)
        res["roofline"]["traffic_frac_of_measured_copy_peak"] = round(tr / HBM_COPY_GBS, 4)
    label = "Reddit" if reddit else "Bio"
    out = {
        "metric": "queries/sec, %s full conjunctive mix d=%d, at 1/2/4/8 MI355X" % (label, d),
        "value": res["value"], "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic", "timing": res["timing"],
        "config": {"workload": "%s full mix (1/2/3-chain, 2/3-inter x{neg,hard}, 3-inter_chain x{neg,hard}): "
                               "%d batches x B=%d per GPU per step, d=%d, %s + SetIntersection(%s), P=%d, Adam lr 0.01"
                               % (args.workload, len(wl.mix), B, d, args.decoder, args.inter_decoder, wl.layout.total),
                   "graph": wl.describe(), "queries_per_step_per_gpu": wl.qpi, "parallelism": "dp%d" % world,
                   "backend": None if world == 1 else backend,
                   "optimizer": "lazy (deferred, bit-exact) Adam — NON-DEFAULT mode" if args.lazy_adam else "eager dense Adam",
                   "step_call": "gqe_train_step (one library call per iteration)" if res.get("step_form", "two calls") != "two calls" else "gqe_margin_fwd_bwd + gqe_adam_step",
                   "step_launches": ("previous step's matrices + row stamps | fused forward/backward tiles WITH Adam over the rows the batches do not name "
                                     "(rider workgroups) | loss finalize + pair-GEMM units + Adam over the named rows and the vectors  (the split step of "
                                     "gqe_train_step, csrc/gqe_split.h)" if res["roofline"]["kernel"].startswith("gqe_fused_kernel") else
                                     "fused forward/backward | Adam pass with the pair-GEMM units and the loss finalize in front of its chunks "
                                     "(gqe_set_deferred_gemm) | Adam on the d x d matrices" if res["roofline"]["kernel"].startswith("gqe_opt_gemm_kernel")
                                     else "fused forward/backward | pair GEMM + loss finalize | [exchange] | Adam pass"),
                   "gradient_exchange": "none" if world == 1 else
                   ("row-sharded tables (rank k owns rows r %% %d == k and their Adam moments), one library call per step "
                    "(gqe_shard_step over RCCL): all-to-all of the %d rows the batch reads (%d floats each), all-to-all of their "
                    "gradient contributions back to the owners WITH every rank's relation/Pre/Post gradient as further sends / "
                    "receives of the same ncclGroup (summed in rank order by each rank: no all-reduce); the fused Adam pass "
                    "streams 1/%d of the tables per rank.  INSIDE the timed region: the host planning of every step (owner sort of "
                    "its index feed + publication on the shared-memory plan board, gqe_shard_post), the serve / link kernels, both "
                    "exchanges, the optimiser" % (world, prepared[0]["n_entries"], d, world)) if sharded else
                   ("one all-gather per step of per-rank slabs: %d contribution entries x (%d floats + row id) + the dense "
                    "relation/Pre/Post gradients" % (prepared[0]["n_entries"], d)) if sparse else
                   "all-reduce of the %d-float gradient arena" % wl.layout.total},
        "roofline": res["roofline"], "kernels": res["kernels"], "step_roofline": res["step_roofline"],
        "final_loss": res["final_loss"],
    }
    if backend_note:
        out["config"]["backend_note"] = backend_note
    if os.environ.get("GQE_BENCH_REDDIT_SCALE"):
        out["config"]["reddit_scale"] = float(os.environ["GQE_BENCH_REDDIT_SCALE"])
        out["config"]["reddit_scale_note"] = "reddit-synth shrunk by this factor (nodes, edges, words): a functional run, not the config-5 measurement"
    if fell_back:
        out["config"]["fell_back"] = "row-sharded step unavailable on this node (%s): replicated tables + sparse exchange measured" % fell_back
    for key in ("exchange_ms_per_step", "exchange_parts_ms", "planning", "ranks_seen", "replicas_identical"):
        if key in res:
            out[key] = res[key]
    short = dict(steps=max(20, min(args.steps, 100)), warmup=min(args.warmup, 10), min_seconds=0.25)
    if world > 1:
        # every exchange form in one line: the other ones, shorter
        eng.close()
        forms = {args.exchange: res}
        for other in ("sharded", "sparse", "dense"):
            if other == args.exchange or (fell_back and other == "sharded"):
                continue
            r2, e2, _ = measure(wl, args, dist, rank, world, exchange=other, lazy=False, check_replicas=True, **short)
            e2.close()
            forms[other] = r2
        out["exchange"] = {name: {"ms_per_step": r["ms_per_step"], "exchange_ms_per_step": r.get("exchange_ms_per_step"),
                                  "exchange_parts_ms": r.get("exchange_parts_ms"), "value": r["value"],
                                  "optimiser_ms": r["roofline"]["avg_launch_ms"], "optimiser_bytes_per_launch": r["roofline"]["algorithmic_bytes_per_launch"],
                                  "replicas_identical": r.get("replicas_identical")} for name, r in forms.items()}
        eng = None
        # ... and what ONE rank does without any exchange, measured in this very run on every rank's GPU at once (each on its own
        # replica of the whole model: the N = 1 line's step, gqe_train_step): the slowest rank's time is reported, so that a scaling
        # table can be read against a same-box, same-run single-GPU step instead of another box's BENCH line
        import torch
        w1 = Workload(args.workload, d, args.decoder, args.inter_decoder, synth.FULL_MIX, B)
        r1, e1, _ = measure(w1, args, None, 0, 1, **short)
        e1.close()
        t1 = torch.tensor([r1["ms_per_step"]], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t1, op=dist.ReduceOp.MAX)
        out["single_rank_step"] = {"ms_per_step": round(float(t1.item()), 4), "value": round(w1.qpi / (float(t1.item()) * 1e-3), 1), "unit": "queries/s per GPU",
                                   "step_form": r1.get("step_form"), "ranks_measured_at_once": world,
                                   "note": "no exchange: every rank steps its own full replica on its own GPU, all ranks at the same time; max over ranks"}
    if world == 1 and not args.no_lazy and not args.lazy_adam:
        rl, el, _ = measure(wl, args, None, 0, 1, lazy=True)
        el.close()
        lz = slim(rl)
        lz["kernels_ms"]["optimiser_rows_or_full_pass"] = rl["roofline"]["avg_launch_ms"]
        lz["note"] = ("same workload and step count; deferred zero-gradient Adam steps are replayed bit-exactly on demand "
                      "(tests/test_gpu_parity.py::test_lazy_adam_is_bit_identical_to_the_eager_schedule); a full pass every "
                      "<= 32 steps per table and the final gqe_optimizer_sync are inside the timed region")
        out["lazy_exact_adam"] = lz
    def secondary(name, fn):
        """A secondary one-GPU measurement must not take the headline line with it: a failure is recorded (and printed), not raised."""
        try:
            out[name] = fn()
        except Exception as e:      # noqa: BLE001
            import traceback
            traceback.print_exc(file=sys.stderr)
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
    if world == 1 and not args.no_host_fed and not args.lazy_adam:
        secondary("host_fed", lambda: host_fed(wl, args))
    if world == 1 and not args.no_api_path and not reddit:
        secondary("api_path", lambda: api_path(args, d, args.decoder, args.inter_decoder, B))
    if world == 1 and not args.no_configs and not reddit:
        cfgs = {}
        M = synth.FULL_MIX

        def run_cfg(name, decoder, mix, Bc, note, zipf=None):
            w = Workload("bio-synth", d, decoder, args.inter_decoder, mix, Bc, n_distinct=16 if Bc > 512 else 32, zipf=zipf)
            r, e, _ = measure(w, args, None, 0, 1, **(dict(short, steps=20) if Bc > 512 else short))
            e.close()
            s = slim(r)
            s["config"] = "%s; %d batches x B=%d, %s + SetIntersection(%s), P=%d" % (note, len(mix), Bc, decoder, args.inter_decoder, w.layout.total)
            cfgs[name] = s
        run_cfg("C1_1chain_only", args.decoder, (M[0],), B, "burn-in loop (train_helpers.py:51)")
        run_cfg("C2_2chain_2inter", args.decoder, (M[0], M[1], M[3], M[4]), B, "1-chain + 2-chain + 2-inter x{neg,hard}")
        run_cfg("C2_without_1chain", args.decoder, (M[1], M[3], M[4]), B, "2-chain + 2-inter x{neg,hard}")
        run_cfg("C4_full_bilinear", "bilinear", M, B, "full mix, d x d relation matrices (decoders.py:142-150): the MFMA path")
        run_cfg("C3_scaled_batch_B8192", args.decoder, M, 8192, "full mix, scaled batch (B=512 is launch-latency-bound)")
        run_cfg("C3_plus_3chain_inter", args.decoder, M + (("3-chain_inter", 0.005, False), ("3-chain_inter", 0.005, True)), B,
                "11-batch mix incl. 3-chain_inter (model.py:99-109)")
        run_cfg("C3_zipf", args.decoder, M, B, "full mix on a HEAVY-TAILED graph: node degrees ~ 1 / rank (hub proteins: graph.py:108-122 keeps "
                "the real adjacency; every other line here is on uniform-random graphs)", zipf=1.0)
        cfgs["C3_zipf"]["optimiser_vs_uniform"] = round(cfgs["C3_zipf"]["optimiser"]["avg_launch_ms"] / res["roofline"]["avg_launch_ms"], 3)
        out["configs"] = cfgs
    if not args.no_reddit and not reddit:
        if eng is not None and world > 1:
            eng.close()
            eng = None
        wr = Workload("reddit-synth", 256, args.decoder, args.inter_decoder, synth.FULL_MIX, B, rank=rank, world=world, n_distinct=16)
        rr, er, _ = measure(wr, args, dist, rank, world, exchange=args.exchange, check_replicas=world > 1, **short)
        er.close()
        rs = slim(rr)
        rs["config"] = ("BASELINE config 5 workload: reddit-synth (%s; EmbeddingBag post features over a %d-word table, bags U[5,30]), "
                        "full mix %d batches x B=%d per GPU, d=256, %s + SetIntersection(%s), P=%d"
                        % (wr.describe(), wr.g.table_rows["post"], len(wr.mix), B, args.decoder, args.inter_decoder, wr.layout.total))
        rides_r = str(rr["roofline"].get("kernel", "")).startswith("gqe_opt_gemm_kernel")   # (the pair-GEMM units spread through the pass)
        rs["optimiser"]["traffic"] = pmc_traffic("gqe_opt_gemm_kernel" if rides_r else "gqe_opt_kernel", "reddit-synth") if world == 1 else None
        if world == 1 and not args.no_lazy:
            # the same workload with lazy (deferred, bit-exact) Adam: at Reddit-sized tables the eager pass streams 3.5 GB per step
            rl, el, _ = measure(wr, args, None, 0, 1, lazy=True, **short)
            el.close()
            lz = slim(rl)
            rs["lazy_exact_adam"] = {"value": lz["value"], "unit": "queries/s", "ms_per_step": lz["ms_per_step"], "kernels_ms": lz["kernels_ms"],
                                     "note": "NON-DEFAULT mode (see lazy_exact_adam above); full pass every <= 32 steps and the final sync inside the timed region"}
        out["reddit_synth"] = rs
        if world == 1:
            # the same workload with Zipfian word frequencies and node degrees (a real vocabulary: reddit/data_utils_new.py:155,162-169)
            wz = Workload("reddit-synth", 256, args.decoder, args.inter_decoder, synth.FULL_MIX, B, n_distinct=16, zipf=1.0)
            rz, ez, _ = measure(wz, args, None, 0, 1, **short)
            hot_rows, (sub_lists, sub_on) = ez.hot_rows(), ez.hot_sub_lists()
            ez.close()
            zs = slim(rz)
            zs["hot_rows"] = {"promoted": hot_rows, "word_sub_lists": sub_lists, "fused_launches_use_them": sub_on,
                              "note": "include/gqe.h, gqe_hot_rows / gqe_hot_sub_lists; kernels_ms.fused_fwd_bwd here = the fused launch + the "
                                      "gather launch behind it (one timing slot)"}
            zs["config"] = "reddit-synth with Zipf(1) word frequencies and node degrees (%s), same mix / d / decoders" % wz.describe()
            zs["rows_with_over_32_contributions"] = rz["rows_with_over_32_contributions"]
            zs["optimiser_vs_uniform"] = round(zs["optimiser"]["avg_launch_ms"] / rs["optimiser"]["avg_launch_ms"], 3)
            zs["fused_vs_uniform"] = round(zs["kernels_ms"]["fused_fwd_bwd"] / rs["kernels_ms"]["fused_fwd_bwd"], 3)
            zs["step_vs_uniform"] = round(zs["ms_per_step"] / rs["ms_per_step"], 3)
            # The ratio above is the FIRST steps of training, where every hinge is active and every hot word collects a gradient
            # from every query that holds it; both workloads run again behind 300 training steps (tools/probes/hot_settle_probe.py:
            # the fused launch settles over ~250 steps at a constant hot set), same engines' parameters, same feeds.
            settled = dict(short, warmup=300, min_seconds=0.1)
            ru, eu, _ = measure(wr, args, None, 0, 1, **settled)
            eu.close()
            rz2, ez2, _ = measure(wz, args, None, 0, 1, **settled)
            ez2.close()
            fu, fz = slim(ru)["kernels_ms"]["fused_fwd_bwd"], slim(rz2)["kernels_ms"]["fused_fwd_bwd"]
            zs["after_300_steps"] = {"fused_fwd_bwd_ms": fz, "uniform_fused_fwd_bwd_ms": fu, "fused_vs_uniform": round(fz / fu, 3),
                                     "value": rz2["value"], "uniform_value": ru["value"], "unit": "queries/s",
                                     "note": "the same two workloads measured behind 300 training steps instead of 10: fewer active hinges, "
                                             "so fewer contributions to the hot words (DESIGN.md section 3, Hot rows; experiments 81, 101)"}
            out["reddit_synth_zipf"] = zs
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not reddit:
        out["cpu_baseline"] = cpu_baseline(eng, args.decoder, args.inter_decoder, wl.item_sets[:8], args.cpu_seconds, wl.qpi)
    elif rank == 0:
        out["cpu_baseline"] = None
    # the keys the driver's record keeps verbatim are the contract's + roofline + cpu_baseline: what a reader of that record should
    # not have to dig for rides in `roofline` as well
    out["roofline"]["step"] = dict(out["step_roofline"], ms_per_step=out["ms_per_step"], note="whole step: bytes the step has to move / ms_per_step")
    if out.get("host_fed") and "error" not in out["host_fed"]:
        hf = out["host_fed"]
        out["roofline"]["host_fed"] = {"resident_feed_value": out["value"], "pinned_host_memory_read_by_the_kernels": hf.get("value"),
                                       "pinned_hipMemcpyAsync": (hf.get("pinned_hipMemcpyAsync") or {}).get("value"), "unit": "queries/s",
                                       "note": "the same schedule with sampling + packing + the feed's transport inside the timed region (gqe_feeder_run)"}
    if out.get("api_path") and "error" not in out["api_path"]:
        ap_ = out["api_path"]
        out["roofline"]["api_path"] = {"run_train_native_runs": ap_.get("value"), "run_train_batch_by_batch": (ap_.get("per_batch_python_path") or {}).get("value"),
                                       "unit": "queries/s", "note": "the reference-shaped API (train_helpers.run_train + FusedAdam on Query objects), "
                                       "same seeds = the reference's batches; see api_path"}
    if rank == 0:
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if eng is not None:
        eng.close()


if __name__ == "__main__":
    main()
