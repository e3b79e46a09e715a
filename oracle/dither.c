// TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's dither noise (never linked / loaded by the product).
//
// kaldi/src/feat/feature-window.cc:90-98   Dither(): a fresh RandomState per frame, data[i] += RandGauss(&rstate) * dither
// kaldi/src/base/kaldi-math.cc:59-70       RandomState::RandomState(): seed = unsigned(Rand()) + 27437, Rand() = glibc rand()
// kaldi/src/base/kaldi-math.cc:43-56       Rand(state) = rand_r(&state->seed)
// kaldi/src/base/kaldi-math.h:150-158      RandUniform = float((Rand + 1.0) / (RAND_MAX + 2.0)),
//                                          RandGauss = float(sqrtf(-2 * logf(RandUniform)) * cosf(2 * M_PI * RandUniform))
//
// The reference starts one process per utterance (tools.py:117-147), so glibc's rand() starts from its default seed (1)
// for every utterance, the model set-up consumes a number of values that depends on the model alone, and frame t of ANY
// utterance adds the same 'win' noise values: the table is a constant of (model, t, i).  glibc's rand() is the TYPE_3 additive-feedback generator of stdlib/random_r.c (r[i] = r[i-3] + r[i-31] on 34
// words seeded by the Lehmer sequence 16807 x mod 2^31 - 1, 310 outputs discarded, result >> 1); rand_r() is the
// three-step LCG of stdlib/rand_r.c.  Pinned bit for bit against the reference's own Dither() built from
// /root/reference (tests/test_oracle_golden.py::test_dither_table_is_the_reference's).
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct { uint32_t r[34]; int f, b; } glibc_rand_t;

static void glibc_srand(glibc_rand_t *g, uint32_t seed) {
  int32_t word = seed ? (int32_t)seed : 1;
  g->r[0] = (uint32_t)word;
  for (int i = 1; i < 31; i++) {
    const long hi = word / 127773, lo = word % 127773;
    long w = 16807 * lo - 2836 * hi;
    if (w < 0) w += 2147483647;
    word = (int32_t)w;
    g->r[i] = (uint32_t)word;
  }
  g->f = 3;   // front pointer: r[f] += r[b]
  g->b = 0;
  for (int i = 0; i < 310; i++) {
    g->r[g->f] += g->r[g->b];
    g->f = (g->f + 1) % 31;
    g->b = (g->b + 1) % 31;
  }
}

static int glibc_rand(glibc_rand_t *g) {
  g->r[g->f] += g->r[g->b];
  const int out = (int)(g->r[g->f] >> 1);
  g->f = (g->f + 1) % 31;
  g->b = (g->b + 1) % 31;
  return out;
}

static int glibc_rand_r(uint32_t *seed) {
  uint32_t next = *seed;
  int result;
  next = next * 1103515245u + 12345u;
  result = (int)((next / 65536u) % 2048u);
  next = next * 1103515245u + 12345u;
  result <<= 10;
  result ^= (int)((next / 65536u) % 1024u);
  next = next * 1103515245u + 12345u;
  result <<= 10;
  result ^= (int)((next / 65536u) % 1024u);
  *seed = next;
  return result;
}

static float rand_uniform(uint32_t *seed) { return (float)((glibc_rand_r(seed) + 1.0) / (2147483647 + 2.0)); }

// first 'n' values of glibc rand() in a fresh process
void oracle_glibc_rand(int n, int *out) {
  glibc_rand_t g;
  glibc_srand(&g, 1);
  for (int i = 0; i < n; i++) out[i] = glibc_rand(&g);
}

// out[t * win + i] = the value Dither() adds (for dither_value 1) to sample i of frame t, frames t0 <= t < t1, in a
// process that has called rand() 'offset' times before its first frame (the decoder binaries' model set-up does:
// nnet3's graph builder and CollapseModel draw from rand(); oracle/nnet3_rand.py restates how often)
void oracle_dither_table(long offset, int t0, int t1, int win, float *out) {
  glibc_rand_t g;
  glibc_srand(&g, 1);
  for (long i = 0; i < offset; i++) (void)glibc_rand(&g);
  for (int t = 0; t < t1; t++) {
    uint32_t seed = (uint32_t)glibc_rand(&g) + 27437u;
    if (t < t0) continue;
    float *row = out + (size_t)(t - t0) * win;
    for (int i = 0; i < win; i++) {
      // the order of the two draws inside RandGauss is unspecified in C++; the reference built with g++ draws the
      // logarithm's operand first (pinned by the bit-for-bit comparison with the reference's own Dither())
      const float u1 = rand_uniform(&seed);
      const float u2 = rand_uniform(&seed);
      row[i] = (float)(sqrtf(-2 * logf(u1)) * cosf(2 * M_PI * u2));
    }
  }
}
