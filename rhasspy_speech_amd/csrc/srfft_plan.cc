#include "srfft_plan.h"

#include <cmath>
#include <utility>

namespace rs {

namespace {
const double kTwoPi = 6.283185307179586476925286766559005;   // Kaldi's M_2PI

// srfft.cc:96-113: the tables hold cos/sin of FLOAT angles evaluated with the float libm overloads
void AppendTwiddle(std::vector<float> *tw, int n, int m) {
  float ang = (float)(n * kTwoPi / m);
  float c = std::cos(ang), s = std::sin(ang);
  tw->push_back(c);
  tw->push_back(-(s + c));
  tw->push_back(s - c);
  ang = (float)(3 * n * kTwoPi / m);
  c = std::cos(ang);
  s = std::sin(ang);
  tw->push_back(c);
  tw->push_back(-(s + c));
  tw->push_back(s - c);
}

// srfft.cc:185-209 applied to an index vector: which input element ends up at each position
std::vector<int> BitReverseGather(int logn) {
  const int N = 1 << logn;
  int lg2 = logn >> 1;
  if (logn & 1) lg2++;
  std::vector<int> seed((size_t)1 << lg2, 0);
  if (lg2 >= 1) seed[1] = 1;
  for (int j = 2; j <= lg2; j++) {
    const int imax = 1 << (j - 1);
    for (int i = 0; i < imax; i++) { seed[i] <<= 1; seed[i + imax] = seed[i] + 1; }
  }
  std::vector<int> x(N);
  for (int i = 0; i < N; i++) x[i] = i;
  if (logn > 1) {
    const int n = 1 << (logn >> 1);
    for (int off = 1; off < n; off++) {
      const int fj = n * seed[off];
      std::swap(x[off], x[fj]);
      int p = off;
      for (int gno = 1; gno < seed[off]; gno++) {
        p += n;
        std::swap(x[p], x[fj + seed[gno]]);
      }
    }
  }
  return x;
}
}  // namespace

SrfftPlan BuildSrfftPlan(int padded_window) {
  SrfftPlan pl;
  const int N = padded_window / 2;
  while ((1 << pl.logn) < N) pl.logn++;
  // ---- levels of the recursion tree (srfft.cc:212-355): a block spawns (off, logn-1), (off + m/2, logn-2),
  // (off + 3m/4, logn-2); blocks of one depth touch disjoint ranges
  std::vector<std::pair<int, int>> cur = {{0, pl.logn}}, next;
  pl.level_begin.push_back(0);
  while (!cur.empty()) {
    next.clear();
    for (const auto &b : cur) {
      const int off = b.first, lg = b.second;
      if (lg >= 3) {
        const int m = 1 << lg, m4 = m / 4, m8 = m / 8;
        for (int n = 0; n < m4; n++) {
          SrfftTask t{0 | (lg << 8), off, n, -1};
          if (n == 0) t.tw = -1;
          else if (n == m8) t.tw = -2;
          else { t.tw = (int)(pl.tw.size() / 6); AppendTwiddle(&pl.tw, n, m); }
          pl.tasks.push_back(t);
        }
        next.push_back({off, lg - 1});
        next.push_back({off + m / 2, lg - 2});
        next.push_back({off + 3 * (m / 4), lg - 2});
      } else if (lg == 2) {
        pl.tasks.push_back(SrfftTask{1 | (2 << 8), off, 0, -1});
      } else if (lg == 1) {
        pl.tasks.push_back(SrfftTask{2 | (1 << 8), off, 0, -1});
      }
    }
    if ((int)pl.tasks.size() > pl.level_begin.back()) pl.level_begin.push_back((int)pl.tasks.size());
    cur.swap(next);
  }
  // ---- A 64-lane wave runs a level in ceil(tasks / 64) rounds.  The whole small blocks (kinds 1 and 2) are leaves -- nothing
  // reads their results before the bit-reversal pass -- so the few of them that would cost a level an extra round move to a
  // later level that has lanes to spare (512-point window: levels of 64 64 64 64 68 28 3 tasks become 64 64 64 64 64 32 3:
  // seven rounds instead of eight; same operations on the same values).
  {
    const int nl = (int)pl.level_begin.size() - 1;
    std::vector<std::vector<SrfftTask>> lv(nl);
    for (int l = 0; l < nl; l++) lv[l].assign(pl.tasks.begin() + pl.level_begin[l], pl.tasks.begin() + pl.level_begin[l + 1]);
    for (int l = 0; l + 1 < nl; l++) {
      int excess = (int)lv[l].size() % 64;
      if (excess == 0 || (int)lv[l].size() < 64) continue;
      for (int to = l + 1; to < nl && excess > 0; to++) {
        int room = (64 - (int)lv[to].size() % 64) % 64;
        for (int i = (int)lv[l].size() - 1; i >= 0 && excess > 0 && room > 0; i--) {
          if ((lv[l][i].kind_logm & 0xFF) == 0) continue;
          lv[to].push_back(lv[l][i]);
          lv[l].erase(lv[l].begin() + i);
          excess--; room--;
        }
      }
    }
    pl.tasks.clear();
    pl.level_begin.assign(1, 0);
    for (int l = 0; l < nl; l++) { pl.tasks.insert(pl.tasks.end(), lv[l].begin(), lv[l].end()); pl.level_begin.push_back((int)pl.tasks.size()); }
  }
  pl.perm = BitReverseGather(pl.logn);
  // ---- srfft.cc:379-385: exp(-2 pi i k / padded) advanced by a float complex multiplication per k
  const int NR = padded_window;
  const float x = (float)(kTwoPi / NR * -1);
  const float root_re = std::cos(x), root_im = std::sin(x);
  float k_re = 1.0f, k_im = 0.0f;
  pl.kn.assign((size_t)2 * (N / 2 + 1), 0.f);
  pl.kn[0] = k_re;
  pl.kn[1] = k_im;
  for (int k = 1; 2 * k <= N; k++) {
    const float t_re = (k_re * root_re) - (k_im * root_im);
    k_im = k_re * root_im + k_im * root_re;
    k_re = t_re;
    pl.kn[2 * k] = k_re;
    pl.kn[2 * k + 1] = k_im;
  }
  return pl;
}

}  // namespace rs
