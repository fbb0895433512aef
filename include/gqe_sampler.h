/* gqe_sampler.h — host-side query-subgraph and negative sampler (C ABI, part of libgqe.so; no GPU involved).
 *
 * North star: "query and negative sampling from netquery.graph.Graph stay on the host cores".  The reference does
 * it with Python sets and dict-of-set adjacency (≈800 queries/s/core); this is the same
 * procedure over CSR adjacency with bitset answer sets and one RNG stream per worker thread.
 *
 * What each entry point replaces (reference file:line):
 *   gqe_sampler_create        Graph.__init__ adjacency / flat_adj_lists / full_sets      netquery/graph.py:104-137
 *   gqe_sampler_sample        Graph.sample_query_subgraph[_bytype]                        netquery/graph.py:298-434
 *                             Graph.get_negative_samples                                  netquery/graph.py:240-291
 *                             Graph.sample_queries / sample_test_queries (accept loop)    netquery/graph.py:185-238
 *                             Query.__init__ negative sub-sampling (neg_sample_max)       netquery/graph.py:55-69
 *   gqe_sampler_check         Graph._is_subgraph / _is_negative (the sampler's invariants) netquery/graph.py:447-534
 *
 * Nodes are addressed as (mode, local index): relation r maps local indices of mode rel_src_mode[r] to local
 * indices of mode rel_dst_mode[r]; rel_reverse[r] is the id of the same edges read backwards (the reference
 * stores both directions, graph.py:8-11).  Reference behaviour kept on purpose: after the first edge of a
 * 3-chain / 3-chain_inter shape the reference continues from (neighbour, SOURCE mode of that edge) (graph.py:319,
 * 387), which only finds out-edges when the relation stays inside one mode — so those two shapes always start
 * with an intra-mode relation.  The RNG is not Python's Mersenne Twister: the sampled distribution is
 * the reference's, the stream is not; parity is checked set-wise (tests/test_sampler_cpu.py).
 */
#ifndef GQE_SAMPLER_H
#define GQE_SAMPLER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gqe_sampler gqe_sampler;

typedef struct {
  int32_t n_modes;
  const int64_t* mode_sizes;       /* [n_modes] nodes per mode */
  int32_t n_rels;
  const int32_t* rel_src_mode;     /* [n_rels] */
  const int32_t* rel_dst_mode;     /* [n_rels] */
  const int32_t* rel_reverse;      /* [n_rels] id of the reversed relation */
  const int64_t* const* rel_ptr;   /* [n_rels] -> int64[mode_sizes[src] + 1] CSR row pointers */
  const int32_t* const* rel_idx;   /* [n_rels] -> int32[nnz] neighbour local indices (dst mode) */
  const uint8_t* const* mode_present; /* NULL, or [n_modes] -> uint8[mode_sizes[m]]: the nodes negatives are drawn from
                                        (Graph.full_sets, graph.py:116-120); default = nodes that occur in some edge */
} gqe_graph_desc;

/* query types, numbered as in include/gqe.h (GQE_Q_*); 1-chain queries are plain edges and are not sampled here */
#define GQE_SAMPLE_ANY (-1)         /* arity-driven: graph.py:364-434 shape probabilities */

/* One call's result, owned by the library (gqe_query_batch_free).  Query i:
 *   qtype[i]; edges[9*i ..]: up to three (src, rel, dst) triples in the reference's reading order
 *   ("2-inter": (t,r1,a1),(t,r2,a2); "3-inter_chain": (t,r1,a1),(t,r2,v),(v,r3,a2); "3-chain_inter":
 *   (t,r1,v),(v,r2,a1),(v,r3,a2); chains: (t,r1,v1),(v1,r2,..)...), unused slots -1; src/dst are local indices.
 *   negatives: neg_idx[neg_ptr[i] .. neg_ptr[i+1]) — local indices in the target mode, ascending unless
 *   sub-sampled; hard negatives likewise (empty for chain queries, which have none). */
typedef struct {
  int64_t n;
  int32_t* qtype;
  int32_t* edges;
  int64_t* neg_ptr;
  int32_t* neg_idx;
  int64_t* hard_ptr;
  int32_t* hard_idx;
  int64_t attempts;   /* sampled shapes incl. rejected ones (diagnostics) */
} gqe_query_batch;

/* Copies the graph. */
int gqe_sampler_create(const gqe_graph_desc* graph, gqe_sampler** out);
int gqe_sampler_destroy(gqe_sampler* s);

/* Samples until n queries are accepted (a query needs a non-empty negative set and, for intersection types, a
 * non-empty hard-negative set — graph.py:193-197).
 *   train      NULL, or the sampler of the training graph: accept only queries whose target does NOT answer the
 *              query there (sample_test_queries, graph.py:222-226)
 *   qtype      GQE_Q_2CHAIN .. GQE_Q_3CHAIN_INTER, or GQE_SAMPLE_ANY with arity 2 or 3
 *   neg_sample_max   as Query.__init__: negatives are sub-sampled when len >= max, hard negatives when len > max
 *   threads    worker threads; the result is a deterministic function of (seed, threads)
 *   max_attempts     give up (GQE_SAMPLER_EXHAUSTED) after this many rejected shapes per accepted one on average; 0 = 10000 */
int gqe_sampler_sample(const gqe_sampler* s, const gqe_sampler* train, int32_t qtype, int32_t arity, int64_t n,
                       int32_t neg_sample_max, uint64_t seed, int32_t threads, int64_t max_attempts,
                       gqe_query_batch** out);
int gqe_query_batch_free(gqe_query_batch* b);

/* Invariant check of one query against this graph: bit 0 = every edge exists and the chains hook up
 * (_is_subgraph); bit 1 = `node` is a negative (_is_negative(.., False)); bit 2 = `node` is a hard negative. */
int gqe_sampler_check(const gqe_sampler* s, int32_t qtype, const int32_t* edges9, int32_t node);

/* The reference draws ONE negative per query with Python's `random.choice(list)` (model.py:113-120): for a list of n entries
 * that is `_randbelow(n)` — k = n.bit_length(); r = getrandbits(k) until r < n — on the Mersenne Twister behind the `random` module.
 * This replays exactly that consumption of the generator for a whole batch: `state` = the 624 words + position of
 * `random.getstate()[1]` (updated in place: hand it back with `random.setstate`), counts[i] = len(list i) >= 1,
 * choice[i] = the index `random.choice` would have picked.  A run seeded like the reference's reproduces its negatives without
 * one interpreter call per query (train_helpers.FusedExecutor).  Returns GQE_SAMPLER_ARG for a count < 1 or a bad position. */
int gqe_py_random_choices(uint32_t* state625, const int64_t* counts, int64_t n, int64_t* choice);

/* The reference picks the formula of a batch with `np.random.multinomial(1, sizes / sum(sizes))` (train_helpers.py:96-99).  This
 * replays that draw — numpy's legacy multinomial -> binomial inversion on the RandomState's MT19937 — on `state625` = the 624 key
 * words + position of `np.random.get_state()` (updated in place: hand it back with `np.random.set_state`); pvals = the float64
 * probability vector exactly as the caller would pass it; *pick = argmax of the draw.  Same value, same generator state after. */
int gqe_np_multinomial_pick(uint32_t* state625, const double* pvals, int64_t d, int64_t* pick);

const char* gqe_sampler_last_error(void);

#define GQE_SAMPLER_OK 0
#define GQE_SAMPLER_ARG 1
#define GQE_SAMPLER_EXHAUSTED 2

#ifdef __cplusplus
}
#endif
#endif
