// gqe_sampler.cpp — host-side query sampler (include/gqe_sampler.h).  Plain C++17, no GPU code.
//
// The procedure is the reference's (netquery/graph.py:185-434): pick a relation, then a start node with that
// relation, grow the query shape edge by edge over the start node's out-edges of ALL relations, compute the
// answer sets of the query's branches, and keep the query if it has negatives (and hard negatives for the
// intersection shapes).  Python sets become bitsets over a mode's local node indices; dict-of-set adjacency
// becomes CSR; every worker thread owns an RNG stream.
#include "../../include/gqe_sampler.h"

#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "gqe_mt.h"   // MT19937 as numpy / CPython run it (C++ helpers: outside the extern "C" block below)
using gqe_mt::mt_next;

namespace {

enum { Q2C = 1, Q3C = 2, Q2I = 3, Q3I = 4, Q3IC = 5, Q3CI = 6 };

thread_local std::string g_err;

struct Rng {  // xoshiro256** seeded through splitmix64
  uint64_t s[4];
  explicit Rng(uint64_t seed) {
    for (int i = 0; i < 4; ++i) {
      seed += 0x9E3779B97F4A7C15ull;
      uint64_t z = seed;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      s[i] = z ^ (z >> 31);
    }
  }
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  uint64_t next() {
    const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 45);
    return r;
  }
  uint64_t below(uint64_t n) {  // uniform in [0, n), Lemire's multiply-shift with rejection
    uint64_t x = next();
    __uint128_t m = (__uint128_t)x * n;
    uint64_t l = (uint64_t)m;
    if (l < n) {
      const uint64_t t = (0 - n) % n;
      while (l < t) {
        x = next();
        m = (__uint128_t)x * n;
        l = (uint64_t)m;
      }
    }
    return (uint64_t)(m >> 64);
  }
};

struct Bits {  // a set of local node indices of one mode
  std::vector<uint64_t> w;
  void reset(int64_t n) { w.assign((size_t)((n + 63) / 64), 0); }
  void set(int32_t i) { w[(size_t)i >> 6] |= 1ull << (i & 63); }
  bool has(int32_t i) const { return (w[(size_t)i >> 6] >> (i & 63)) & 1; }
  int64_t count() const {
    int64_t c = 0;
    for (uint64_t x : w) c += __builtin_popcountll(x);
    return c;
  }
  bool any() const {
    for (uint64_t x : w)
      if (x) return true;
    return false;
  }
};

struct Edge {
  int32_t src, rel, dst;
};

struct QueryG {
  int32_t qtype;
  Edge e[3];
  int n_edges() const { return qtype == Q2C || qtype == Q2I ? 2 : 3; }
};

}  // namespace

struct gqe_sampler {
  int32_t n_modes = 0, n_rels = 0;
  std::vector<int64_t> mode_size;
  std::vector<int32_t> src, dst, rev;
  std::vector<std::vector<int64_t>> ptr;
  std::vector<std::vector<int32_t>> idx;
  // flat_adj_lists (graph.py:120-127): every out-edge (rel, neighbour) of a node, all relations of its mode
  std::vector<std::vector<int64_t>> fptr;
  std::vector<std::vector<int32_t>> frel, fnei;
  std::vector<std::vector<int32_t>> rel_sources;  // nodes with at least one edge of the relation
  std::vector<Bits> present;                      // full_sets (graph.py:128-137): nodes that occur in some edge

  int64_t deg(int32_t r, int32_t u) const { return ptr[r][u + 1] - ptr[r][u]; }
  bool has_edge(const Edge& e) const {
    if (e.rel < 0 || e.rel >= n_rels || e.src < 0 || e.src >= mode_size[src[e.rel]]) return false;
    const int32_t* b = idx[e.rel].data() + ptr[e.rel][e.src];
    const int32_t* en = idx[e.rel].data() + ptr[e.rel][e.src + 1];
    return std::binary_search(b, en, e.dst);
  }
  void neighbours(int32_t r, int32_t u, Bits& out) const {
    for (int64_t k = ptr[r][u]; k < ptr[r][u + 1]; ++k) out.set(idx[r][k]);
  }
  // nodes reached from the set `from` (mode src[r]) over relation r
  void expand(int32_t r, const Bits& from, Bits& out) const {
    out.reset(mode_size[dst[r]]);
    for (size_t wi = 0; wi < from.w.size(); ++wi) {
      uint64_t x = from.w[wi];
      while (x) {
        const int b = __builtin_ctzll(x);
        x &= x - 1;
        neighbours(r, (int32_t)(wi * 64 + b), out);
      }
    }
  }
  // answer set of a one-edge branch (t, r, a): the t with t -r-> a, read off the reversed relation (graph.py:262)
  void branch1(const Edge& e, Bits& out) const {
    out.reset(mode_size[src[e.rel]]);
    neighbours(rev[e.rel], e.dst, out);
  }
  // two-hop branch ((t, r2, v), (v, r3, a))
  void branch2(const Edge& e2, const Edge& e3, Bits& out) const {
    Bits mid;
    branch1(e3, mid);
    expand(rev[e2.rel], mid, out);
  }
  int32_t target_mode(const QueryG& q) const { return src[q.e[0].rel]; }

  // pos = nodes that satisfy the query; upos = nodes that satisfy at least one branch (intersection shapes)
  void answer_sets(const QueryG& q, Bits& pos, Bits& upos, bool want_union) const {
    Bits a, b, c;
    switch (q.qtype) {
      case Q2C:
        branch2(q.e[0], q.e[1], pos);
        break;
      case Q3C:
        branch2(q.e[1], q.e[2], a);          // the v1 with v1 -r2-> v2 -r3-> anchor
        expand(rev[q.e[0].rel], a, pos);
        break;
      case Q2I:
      case Q3I:
      case Q3IC: {
        branch1(q.e[0], a);
        if (q.qtype == Q3IC) {
          branch2(q.e[1], q.e[2], b);
        } else {
          branch1(q.e[1], b);
          if (q.qtype == Q3I) branch1(q.e[2], c);
        }
        pos = a;
        for (size_t i = 0; i < pos.w.size(); ++i) pos.w[i] &= b.w[i];
        if (q.qtype == Q3I)
          for (size_t i = 0; i < pos.w.size(); ++i) pos.w[i] &= c.w[i];
        if (want_union) {
          upos = a;
          for (size_t i = 0; i < upos.w.size(); ++i) upos.w[i] |= b.w[i];
          if (q.qtype == Q3I)
            for (size_t i = 0; i < upos.w.size(); ++i) upos.w[i] |= c.w[i];
        }
        break;
      }
      default: {  // Q3CI: (t, r1, v), (v, r2, a1), (v, r3, a2)
        branch1(q.e[1], a);
        branch1(q.e[2], b);
        Bits both = a;
        for (size_t i = 0; i < both.w.size(); ++i) both.w[i] &= b.w[i];
        expand(rev[q.e[0].rel], both, pos);
        if (want_union) {
          for (size_t i = 0; i < a.w.size(); ++i) a.w[i] |= b.w[i];
          expand(rev[q.e[0].rel], a, upos);
        }
      }
    }
  }
};

namespace {

bool is_inter(int32_t qt) { return qt >= Q2I; }

bool hooks_up(const gqe_sampler& g, const QueryG& q) {
  const Edge* e = q.e;
  for (int i = 0; i < q.n_edges(); ++i)
    if (!g.has_edge(e[i])) return false;
  switch (q.qtype) {
    case Q2C: return e[0].dst == e[1].src && g.dst[e[0].rel] == g.src[e[1].rel];
    case Q3C: return e[0].dst == e[1].src && e[1].dst == e[2].src && g.dst[e[0].rel] == g.src[e[1].rel] && g.dst[e[1].rel] == g.src[e[2].rel];
    case Q2I: return e[0].src == e[1].src && g.src[e[0].rel] == g.src[e[1].rel];
    case Q3I: return e[0].src == e[1].src && e[0].src == e[2].src && g.src[e[0].rel] == g.src[e[1].rel] && g.src[e[0].rel] == g.src[e[2].rel];
    case Q3IC: return e[0].src == e[1].src && e[1].dst == e[2].src && g.src[e[0].rel] == g.src[e[1].rel] && g.dst[e[1].rel] == g.src[e[2].rel];
    default: return e[0].dst == e[1].src && e[1].src == e[2].src && g.dst[e[0].rel] == g.src[e[1].rel] && g.src[e[1].rel] == g.src[e[2].rel];
  }
}

struct Worker {
  const gqe_sampler& g;
  const gqe_sampler* train;
  Rng rng;
  Worker(const gqe_sampler& g_, const gqe_sampler* t, uint64_t seed) : g(g_), train(t), rng(seed) {}

  int64_t flat_deg(int32_t mode, int32_t u) const { return g.fptr[mode][u + 1] - g.fptr[mode][u]; }
  Edge flat_edge(int32_t mode, int32_t u, int64_t k) const {
    const int64_t p = g.fptr[mode][u] + k;
    return Edge{u, g.frel[mode][p], g.fnei[mode][p]};
  }
  Edge sample_edge(int32_t mode, int32_t u) { return flat_edge(mode, u, (int64_t)rng.below((uint64_t)flat_deg(mode, u))); }
  // k pairwise-distinct out-edges: the first one free, the others re-drawn until distinct (graph.py:322-356)
  void distinct_edges(int32_t mode, int32_t u, int k, Edge* out) {
    const uint64_t dg = (uint64_t)flat_deg(mode, u);
    int64_t pick[3];
    for (int i = 0; i < k; ++i) {
      for (;;) {
        pick[i] = (int64_t)rng.below(dg);
        bool dup = false;
        for (int j = 0; j < i; ++j) {
          const Edge a = flat_edge(mode, u, pick[i]), b = flat_edge(mode, u, pick[j]);
          dup = dup || (a.rel == b.rel && a.dst == b.dst);
        }
        if (!dup) break;
      }
      out[i] = flat_edge(mode, u, pick[i]);
    }
  }
  static int root_edges(int32_t qtype) { return qtype == Q2I || qtype == Q3IC ? 2 : (qtype == Q3I ? 3 : 1); }

  // graph.py:298-434; qtype < 0: the arity-driven shape lottery
  bool sample_shape(int32_t qtype, int32_t arity, int32_t mode, int32_t node, QueryG& q) {
    int num_edges;
    if (qtype >= 0) {
      num_edges = root_edges(qtype);
    } else if (arity == 3) {
      static const int lot[4] = {1, 1, 2, 3};
      num_edges = lot[rng.below(4)];
    } else {
      num_edges = 1 + (int)rng.below(2);
    }
    if (num_edges > flat_deg(mode, node)) return false;
    if (arity == 3) {
      if (num_edges == 1) {
        const Edge e = sample_edge(mode, node);
        QueryG sub;
        const int32_t sub_type = qtype < 0 ? -1 : (qtype == Q3C ? Q2C : Q2I);
        // The reference continues from (neigh, rel[0]) — the SOURCE mode of the edge just taken (graph.py:319,387).
        // Node ids being unique across modes, flat_adj_lists[rel[0]][neigh] is empty unless the relation stays
        // inside one mode, so 3-chain / 3-chain_inter queries only ever start with an intra-mode relation.  The
        // query files the reference ships were drawn this way; reproduce it.
        if (g.src[e.rel] != g.dst[e.rel]) return false;
        if (!sample_shape(sub_type, 2, g.dst[e.rel], e.dst, sub)) return false;
        q.qtype = sub.qtype == Q2C ? Q3C : Q3CI;
        q.e[0] = e;
        q.e[1] = sub.e[0];
        q.e[2] = sub.e[1];
        return true;
      }
      if (num_edges == 2) {
        Edge e[2];
        distinct_edges(mode, node, 2, e);
        if (flat_deg(g.dst[e[1].rel], e[1].dst) < 1) return false;
        q.qtype = Q3IC;
        q.e[0] = e[0];
        q.e[1] = e[1];
        q.e[2] = sample_edge(g.dst[e[1].rel], e[1].dst);
        return true;
      }
      q.qtype = Q3I;
      distinct_edges(mode, node, 3, q.e);
      return true;
    }
    if (num_edges == 1) {
      q.qtype = Q2C;
      q.e[0] = sample_edge(mode, node);
      if (flat_deg(g.dst[q.e[0].rel], q.e[0].dst) < 1) return false;
      q.e[1] = sample_edge(g.dst[q.e[0].rel], q.e[0].dst);
      return true;
    }
    q.qtype = Q2I;
    distinct_edges(mode, node, 2, q.e);
    return true;
  }

  // ascending members of `set`, or `limit` of them drawn without replacement when there are too many
  void emit(const Bits& set, int64_t count, int64_t limit, bool subsample, std::vector<int32_t>& out) {
    if (!subsample) {
      for (size_t wi = 0; wi < set.w.size(); ++wi) {
        uint64_t x = set.w[wi];
        while (x) {
          out.push_back((int32_t)(wi * 64 + __builtin_ctzll(x)));
          x &= x - 1;
        }
      }
      return;
    }
    // Floyd's algorithm for `limit` distinct ranks in [0, count), then one pass over the set
    std::vector<int64_t> ranks;
    ranks.reserve((size_t)limit);
    for (int64_t j = count - limit; j < count; ++j) {
      const int64_t t = (int64_t)rng.below((uint64_t)j + 1);
      ranks.push_back(std::find(ranks.begin(), ranks.end(), t) == ranks.end() ? t : j);
    }
    std::sort(ranks.begin(), ranks.end());
    size_t ri = 0;
    int64_t seen = 0;
    for (size_t wi = 0; wi < set.w.size() && ri < ranks.size(); ++wi) {
      uint64_t x = set.w[wi];
      const int pc = __builtin_popcountll(x);
      if (seen + pc <= ranks[ri]) {
        seen += pc;
        continue;
      }
      while (x && ri < ranks.size()) {
        if (seen == ranks[ri]) {
          out.push_back((int32_t)(wi * 64 + __builtin_ctzll(x)));
          ++ri;
        }
        x &= x - 1;
        ++seen;
      }
    }
  }

  struct Out {
    std::vector<int32_t> qtype, edges, neg, hard;
    std::vector<int64_t> neg_ptr{0}, hard_ptr{0};
    int64_t attempts = 0;
    bool exhausted = false;
  };

  void run(int32_t qtype, int32_t arity, int64_t quota, int32_t neg_max, int64_t max_attempts, Out& o) {
    Bits pos, upos, negs, tpos, tu;
    int64_t got = 0;
    while (got < quota) {
      if (o.attempts >= max_attempts * (got + 1) + 1000) {
        o.exhausted = true;
        return;
      }
      ++o.attempts;
      // _random_start (graph.py:317-320): a relation, then a node that has it
      const int32_t r = (int32_t)rng.below((uint64_t)g.n_rels);
      if (g.rel_sources[r].empty()) continue;
      const int32_t node = g.rel_sources[r][rng.below(g.rel_sources[r].size())];
      QueryG q;
      if (!sample_shape(qtype, arity, g.src[r], node, q)) continue;
      if (train) {  // test queries: the target must not answer the query in the training graph
        train->answer_sets(q, tpos, tu, false);
        if (tpos.has(q.e[0].src)) continue;
      }
      const bool inter = is_inter(q.qtype);
      g.answer_sets(q, pos, upos, inter);
      const Bits& full = g.present[g.target_mode(q)];
      negs.w.resize(full.w.size());
      for (size_t i = 0; i < full.w.size(); ++i) negs.w[i] = full.w[i] & ~pos.w[i];
      const int64_t n_neg = negs.count();
      if (n_neg == 0) continue;
      int64_t n_hard = 0;
      if (inter) {
        for (size_t i = 0; i < upos.w.size(); ++i) upos.w[i] &= ~pos.w[i];
        n_hard = upos.count();
        if (n_hard == 0) continue;
      }
      o.qtype.push_back(q.qtype);
      for (int i = 0; i < 3; ++i) {
        const bool on = i < q.n_edges();
        o.edges.push_back(on ? q.e[i].src : -1);
        o.edges.push_back(on ? q.e[i].rel : -1);
        o.edges.push_back(on ? q.e[i].dst : -1);
      }
      emit(negs, n_neg, neg_max, n_neg >= neg_max, o.neg);        // Query.__init__: `<` keeps all (graph.py:59-62)
      o.neg_ptr.push_back((int64_t)o.neg.size());
      if (inter) emit(upos, n_hard, neg_max, n_hard > neg_max, o.hard);  // `<=` keeps all (graph.py:65-68)
      o.hard_ptr.push_back((int64_t)o.hard.size());
      ++got;
    }
  }
};

template <class T>
T* dup(const std::vector<T>& v) {
  T* p = static_cast<T*>(malloc(std::max<size_t>(v.size(), 1) * sizeof(T)));
  if (p && !v.empty()) memcpy(p, v.data(), v.size() * sizeof(T));
  return p;
}

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

}  // namespace

extern "C" {

// ---- Python's random.choice on a batch of lists (include/gqe_sampler.h) ------------------------------------------------
// MT19937 as CPython's _randommodule.c runs it (gqe_mt.h): genrand_uint32 with the standard tempering; getrandbits(k <= 32) = one
// output word >> (32 - k); lists longer than 2^32 do not occur (k <= 32 is checked).
int gqe_py_random_choices(uint32_t* state625, const int64_t* counts, int64_t n, int64_t* choice) {
  if (!state625 || !counts || !choice || n < 0) return GQE_SAMPLER_ARG;
  uint32_t pos = state625[624];
  if (pos > 624) return GQE_SAMPLER_ARG;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t c = counts[i];
    if (c < 1 || c > 0xffffffffll) return GQE_SAMPLER_ARG;
    int k = 0;
    for (int64_t x = c; x; x >>= 1) ++k;          // c.bit_length()
    uint32_t r;
    do {
      r = mt_next(state625, pos) >> (32 - k);
    } while ((int64_t)r >= c);
    choice[i] = (int64_t)r;
  }
  state625[624] = pos;
  return GQE_SAMPLER_OK;
}

int gqe_np_multinomial_pick(uint32_t* state625, const double* pvals, int64_t d, int64_t* pick) {
  if (!state625 || !pvals || !pick || d < 1) return GQE_SAMPLER_ARG;
  uint32_t pos = state625[624];
  if (pos > 624) return GQE_SAMPLER_ARG;
  *pick = gqe_mt::np_multinomial_one(state625, pos, pvals, d);
  state625[624] = pos;
  return GQE_SAMPLER_OK;
}

const char* gqe_sampler_last_error(void) { return g_err.c_str(); }

int gqe_sampler_create(const gqe_graph_desc* d, gqe_sampler** out) {
  if (!d || !out) return fail(GQE_SAMPLER_ARG, "null argument");
  if (d->n_modes < 1 || d->n_rels < 1) return fail(GQE_SAMPLER_ARG, "need at least one mode and one relation");
  auto* s = new gqe_sampler;
  s->n_modes = d->n_modes;
  s->n_rels = d->n_rels;
  s->mode_size.assign(d->mode_sizes, d->mode_sizes + d->n_modes);
  s->src.assign(d->rel_src_mode, d->rel_src_mode + d->n_rels);
  s->dst.assign(d->rel_dst_mode, d->rel_dst_mode + d->n_rels);
  s->rev.assign(d->rel_reverse, d->rel_reverse + d->n_rels);
  s->ptr.resize(d->n_rels);
  s->idx.resize(d->n_rels);
  s->rel_sources.resize(d->n_rels);
  s->present.resize(d->n_modes);
  for (int m = 0; m < d->n_modes; ++m) s->present[m].reset(s->mode_size[m]);
  std::string err;
  for (int r = 0; r < d->n_rels && err.empty(); ++r) {
    const int sm = s->src[r], dm = s->dst[r], rv = s->rev[r];
    if (sm < 0 || sm >= d->n_modes || dm < 0 || dm >= d->n_modes || rv < 0 || rv >= d->n_rels) {
      err = "relation " + std::to_string(r) + ": mode or reverse id out of range";
      break;
    }
    if (d->rel_src_mode[rv] != dm || d->rel_dst_mode[rv] != sm) {
      err = "relation " + std::to_string(r) + ": its reverse does not connect the swapped modes";
      break;
    }
    const int64_t n = s->mode_size[sm];
    s->ptr[r].assign(d->rel_ptr[r], d->rel_ptr[r] + n + 1);
    if (s->ptr[r][0] != 0) err = "relation " + std::to_string(r) + ": rel_ptr[0] != 0";
    for (int64_t u = 0; u < n && err.empty(); ++u)
      if (s->ptr[r][u + 1] < s->ptr[r][u]) err = "relation " + std::to_string(r) + ": rel_ptr not monotone";
    if (!err.empty()) break;
    s->idx[r].assign(d->rel_idx[r], d->rel_idx[r] + s->ptr[r][n]);
    for (int64_t u = 0; u < n; ++u) {
      int32_t* b = s->idx[r].data() + s->ptr[r][u];
      int32_t* e = s->idx[r].data() + s->ptr[r][u + 1];
      std::sort(b, e);
      if (b != e) {
        s->rel_sources[r].push_back((int32_t)u);
        s->present[sm].set((int32_t)u);
      }
      for (int32_t* p = b; p != e; ++p) {
        if (*p < 0 || *p >= s->mode_size[dm]) {
          err = "relation " + std::to_string(r) + ": neighbour index out of range";
          break;
        }
        s->present[dm].set(*p);
      }
      if (!err.empty()) break;
    }
  }
  if (!err.empty()) {
    delete s;
    return fail(GQE_SAMPLER_ARG, err);
  }
  if (d->mode_present)
    for (int m = 0; m < d->n_modes; ++m) {
      s->present[m].reset(s->mode_size[m]);
      for (int64_t u = 0; u < s->mode_size[m]; ++u)
        if (d->mode_present[m][u]) s->present[m].set((int32_t)u);
    }
  s->fptr.resize(d->n_modes);
  s->frel.resize(d->n_modes);
  s->fnei.resize(d->n_modes);
  for (int m = 0; m < d->n_modes; ++m) {
    const int64_t n = s->mode_size[m];
    s->fptr[m].assign((size_t)n + 1, 0);
    for (int r = 0; r < d->n_rels; ++r)
      if (s->src[r] == m)
        for (int64_t u = 0; u < n; ++u) s->fptr[m][u + 1] += s->deg(r, (int32_t)u);
    for (int64_t u = 0; u < n; ++u) s->fptr[m][u + 1] += s->fptr[m][u];
    s->frel[m].resize((size_t)s->fptr[m][n]);
    s->fnei[m].resize((size_t)s->fptr[m][n]);
    std::vector<int64_t> at(s->fptr[m].begin(), s->fptr[m].end() - 1);
    for (int r = 0; r < d->n_rels; ++r)
      if (s->src[r] == m)
        for (int64_t u = 0; u < n; ++u)
          for (int64_t k = s->ptr[r][u]; k < s->ptr[r][u + 1]; ++k) {
            s->frel[m][at[u]] = r;
            s->fnei[m][at[u]++] = s->idx[r][k];
          }
  }
  *out = s;
  return GQE_SAMPLER_OK;
}

int gqe_sampler_destroy(gqe_sampler* s) {
  delete s;
  return GQE_SAMPLER_OK;
}

int gqe_sampler_sample(const gqe_sampler* s, const gqe_sampler* train, int32_t qtype, int32_t arity, int64_t n,
                       int32_t neg_sample_max, uint64_t seed, int32_t threads, int64_t max_attempts,
                       gqe_query_batch** out) {
  if (!s || !out || n < 0) return fail(GQE_SAMPLER_ARG, "null sampler / output or negative count");
  if (qtype == 0 || qtype > Q3CI || qtype < GQE_SAMPLE_ANY) return fail(GQE_SAMPLER_ARG, "qtype must be 1..6 or GQE_SAMPLE_ANY");
  if (qtype == GQE_SAMPLE_ANY && arity != 2 && arity != 3) return fail(GQE_SAMPLER_ARG, "Only arity of at most 3 is supported for queries");
  if (qtype > 0) arity = (qtype == Q2C || qtype == Q2I) ? 2 : 3;
  if (neg_sample_max < 1) return fail(GQE_SAMPLER_ARG, "neg_sample_max must be >= 1");
  if (train && (train->n_modes != s->n_modes || train->n_rels != s->n_rels || train->mode_size != s->mode_size))
    return fail(GQE_SAMPLER_ARG, "the training graph has a different schema");
  if (threads < 1) threads = 1;
  if ((int64_t)threads > std::max<int64_t>(n, 1)) threads = (int32_t)std::max<int64_t>(n, 1);
  if (max_attempts < 1) max_attempts = 10000;
  std::vector<Worker::Out> outs((size_t)threads);
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) {
    const int64_t quota = n / threads + (t < n % threads ? 1 : 0);
    auto job = [=, &outs]() {
      Worker w(*s, train, seed * 0x9E3779B97F4A7C15ull + (uint64_t)t * 0xD1B54A32D192ED03ull + 1);
      w.run(qtype, arity, quota, neg_sample_max, max_attempts, outs[(size_t)t]);
    };
    if (threads == 1) job(); else pool.emplace_back(job);
  }
  for (auto& th : pool) th.join();
  Worker::Out all;
  for (auto& o : outs) {
    if (o.exhausted) return fail(GQE_SAMPLER_EXHAUSTED, "gave up: too many rejected shapes (graph too sparse for this query type?)");
    all.attempts += o.attempts;
    all.qtype.insert(all.qtype.end(), o.qtype.begin(), o.qtype.end());
    all.edges.insert(all.edges.end(), o.edges.begin(), o.edges.end());
    const int64_t nb = (int64_t)all.neg.size(), hb = (int64_t)all.hard.size();
    all.neg.insert(all.neg.end(), o.neg.begin(), o.neg.end());
    all.hard.insert(all.hard.end(), o.hard.begin(), o.hard.end());
    for (size_t i = 1; i < o.neg_ptr.size(); ++i) all.neg_ptr.push_back(nb + o.neg_ptr[i]);
    for (size_t i = 1; i < o.hard_ptr.size(); ++i) all.hard_ptr.push_back(hb + o.hard_ptr[i]);
  }
  auto* b = static_cast<gqe_query_batch*>(malloc(sizeof(gqe_query_batch)));
  b->n = (int64_t)all.qtype.size();
  b->qtype = dup(all.qtype);
  b->edges = dup(all.edges);
  b->neg_ptr = dup(all.neg_ptr);
  b->neg_idx = dup(all.neg);
  b->hard_ptr = dup(all.hard_ptr);
  b->hard_idx = dup(all.hard);
  b->attempts = all.attempts;
  *out = b;
  return GQE_SAMPLER_OK;
}

int gqe_query_batch_free(gqe_query_batch* b) {
  if (!b) return GQE_SAMPLER_OK;
  free(b->qtype);
  free(b->edges);
  free(b->neg_ptr);
  free(b->neg_idx);
  free(b->hard_ptr);
  free(b->hard_idx);
  free(b);
  return GQE_SAMPLER_OK;
}

int gqe_sampler_check(const gqe_sampler* s, int32_t qtype, const int32_t* e9, int32_t node) {
  if (!s || !e9 || qtype < Q2C || qtype > Q3CI) return -1;
  QueryG q;
  q.qtype = qtype;
  for (int i = 0; i < 3; ++i) q.e[i] = Edge{e9[3 * i], e9[3 * i + 1], e9[3 * i + 2]};
  int res = 0;
  if (!hooks_up(*s, q)) return res;
  res |= 1;
  if (node < 0 || node >= s->mode_size[s->target_mode(q)]) return res;
  Bits pos, upos;
  s->answer_sets(q, pos, upos, is_inter(qtype));
  if (!pos.has(node)) {
    res |= 2;
    if (is_inter(qtype) && upos.has(node)) res |= 4;
  }
  return res;
}

}  // extern "C"
