// gqe_mt.h — the two host generators the reference's training loop draws from, replayed natively (test infrastructure for
// neither: both are product code behind train_helpers.run_train's unchanged signature).
//   * `random` (CPython _randommodule.c): negatives, random.choice per query (model.py:113-120);
//   * `np.random` (numpy RandomState, legacy distributions): the formula of a batch, np.random.multinomial(1, p)
//     (train_helpers.py:96-99).
// Both are MT19937; a state is 624 key words + the position (word 624), as random.getstate()[1] / np.random.get_state()[1:3]
// hold it.
#ifndef GQE_MT_H
#define GQE_MT_H
#include <cmath>
#include <cstdint>

namespace gqe_mt {

inline uint32_t mt_next(uint32_t* mt, uint32_t& pos) {
  constexpr uint32_t N = 624, M = 397;
  if (pos >= N) {
    auto twist = [](uint32_t u, uint32_t v) { return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u); };
    uint32_t kk = 0;
    for (; kk < N - M; ++kk) mt[kk] = mt[kk + M] ^ twist(mt[kk], mt[kk + 1]);
    for (; kk < N - 1; ++kk) mt[kk] = mt[kk + M - N] ^ twist(mt[kk], mt[kk + 1]);
    mt[N - 1] = mt[M - 1] ^ twist(mt[N - 1], mt[0]);
    pos = 0;
  }
  uint32_t y = mt[pos++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

// random.choice(range(c)): _randbelow_with_getrandbits — k = c.bit_length(); r = getrandbits(k) until r < c  (1 <= c < 2^32)
inline int64_t py_randbelow(uint32_t* mt, uint32_t& pos, int64_t c) {
  int k = 0;
  for (int64_t x = c; x; x >>= 1) ++k;
  uint32_t r;
  do {
    r = mt_next(mt, pos) >> (32 - k);
  } while ((int64_t)r >= c);
  return (int64_t)r;
}

// numpy's legacy double (randomkit rk_double): 53 bits out of two outputs
inline double np_double(uint32_t* mt, uint32_t& pos) {
  const uint32_t a = mt_next(mt, pos) >> 5, b = mt_next(mt, pos) >> 6;
  return (a * 67108864.0 + b) / 9007199254740992.0;
}

// numpy/random/src/legacy/legacy-distributions.c, legacy_random_binomial_inversion (the branch every draw with n * p <= 30 takes;
// its cache of (q, qn, np, bound) holds values recomputed here)
inline int64_t np_binomial_inversion(uint32_t* mt, uint32_t& pos, int64_t n, double p) {
  const double q = 1.0 - p, qn = std::exp((double)n * std::log(q)), np_ = (double)n * p;
  const double b = np_ + 10.0 * std::sqrt(np_ * q + 1);
  const int64_t bound = (int64_t)((double)n < b ? (double)n : b);
  int64_t X = 0;
  double px = qn, U = np_double(mt, pos);
  while (U > px) {
    ++X;
    if (X > bound) {
      X = 0;
      px = qn;
      U = np_double(mt, pos);
    } else {
      U -= px;
      px = ((double)(n - X + 1) * p * px) / ((double)X * q);
    }
  }
  return X;
}

// legacy_random_binomial for n * min(p, 1 - p) <= 30 (n = 1 here: always)
inline int64_t np_binomial_small(uint32_t* mt, uint32_t& pos, double p, int64_t n) {
  if (p <= 0.5) return np_binomial_inversion(mt, pos, n, p);
  return n - np_binomial_inversion(mt, pos, n, 1.0 - p);
}

// np.random.multinomial(1, pvals).argmax() (legacy_random_multinomial with n = 1): the category of the single trial
inline int64_t np_multinomial_one(uint32_t* mt, uint32_t& pos, const double* pvals, int64_t d) {
  double remaining_p = 1.0;
  for (int64_t j = 0; j < d - 1; ++j) {
    if (np_binomial_small(mt, pos, pvals[j] / remaining_p, 1) > 0) return j;
    remaining_p -= pvals[j];
  }
  return d - 1;
}

}  // namespace gqe_mt
#endif
