// Host-side n-best extraction from a pruned raw lattice (the tail of the reference's pipeline:
// DeterminizeLatticePhonePrunedWrapper -> lattice-to-nbest -> nbest-to-linear).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace rs {

struct RawLattice {
  int start = -1;
  int num_states = 0;
  std::vector<double> final_cost;           // per state (graph side), +inf = non-final
  struct Arc { int src, dst, olabel; double graph, acoustic; int ilabel = 0; };   // ilabel = transition-id (0 = none)
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

// Word-determinised lattice in Kaldi's CompactLattice form: an acceptor over word ids whose arc / final weights carry
// (graph cost, acoustic cost) and the transition-id string of the best alignment.
struct CompactLat {
  struct Weight { double graph = 0, acoustic = 0; std::vector<int32_t> tids; };
  struct Arc { int dst, label; Weight w; };
  int start = -1;
  std::vector<std::vector<Arc>> arcs;          // per state
  std::vector<Weight> final_w;                 // per state
  std::vector<char> is_final;
};
// The role of DeterminizeLatticePhonePrunedWrapper (lat/determinize-lattice-pruned.cc:1488-1513) on the pruned raw lattice:
// every word sequence whose best alignment lies within `beam` of the best path survives as exactly one path that carries
// the (graph, acoustic) costs and the transition-ids of that alignment (LatticeWeight order: total, then graph part).
// The state numbering and the placement of weights along a path are this implementation's, not Kaldi's: equivalent
// lattices, not identical files.
CompactLat DeterminizeLattice(const RawLattice &lat, double beam);
// One binary table entry as `lattice-copy ark:- ark:-` would write it: "<key> " + VectorFst<CompactLatticeArc>.
std::string CompactLatticeArkEntry(const std::string &key, const CompactLat &clat);

}  // namespace rs
