// Host-side n-best extraction from a pruned raw lattice (the tail of the reference's pipeline:
// DeterminizeLatticePhonePrunedWrapper -> lattice-to-nbest -> nbest-to-linear).
#pragma once
#include <cstdint>
#include <vector>

namespace rs {

struct RawLattice {
  int start = -1;
  int num_states = 0;
  std::vector<double> final_cost;           // per state (graph side), +inf = non-final
  struct Arc { int src, dst, olabel; double graph, acoustic; };
  std::vector<Arc> arcs;
};

struct NbestPath {
  std::vector<int32_t> words;
  double graph_cost = 0, acoustic_cost = 0;
};

// Up to n distinct word sequences in increasing (graph + acoustic_scale * acoustic) order, restricted to
// sequences whose best alignment is within lattice_beam of the best path (the determinisation beam,
// lat/determinize-lattice-pruned.cc:1488-1513); each with the (graph, acoustic) costs of its best alignment
// (LatticeWeight comparison: total, then graph part; fstext/lattice-weight.h:294-307).
std::vector<NbestPath> LatticeNbest(const RawLattice &lat, int n, double lattice_beam, double acoustic_scale);

}  // namespace rs
