// Host-side model description: everything the hot path reads from disk, parsed into plain
// structs (no Kaldi types).  File/line citations refer to /root/reference/kaldi/src.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "kaldi_io.h"

namespace rs {

// ---------------------------------------------------------------- feature options
// feat/feature-window.h:40-116, feat/mel-computations.h:43-75, feat/feature-mfcc.h:40-80
struct MfccOptions {
  float samp_freq = 16000, frame_shift_ms = 10, frame_length_ms = 25, dither = 1.0f, preemph = 0.97f;
  bool remove_dc = true, round_pow2 = true, snip_edges = true;
  bool allow_downsample = false, allow_upsample = false;   // feature-window.h:95-107: the reference then resamples (LinearResample); the library refuses such input
  std::string window_type = "povey";
  float blackman_coeff = 0.42f;
  int num_bins = 23;
  float low_freq = 20, high_freq = 0, vtln_low = 100, vtln_high = -500;
  int num_ceps = 13;
  bool use_energy = true, raw_energy = true, htk_compat = false;
  float energy_floor = 0, cepstral_lifter = 22;
  int WindowShift() const { return (int)(samp_freq * 0.001f * frame_shift_ms); }
  int WindowSize() const { return (int)(samp_freq * 0.001f * frame_length_ms); }
  int PaddedWindowSize() const;
};

// Tables the MFCC kernel needs, computed once on the host with the reference's float/double
// expression order (feature-window.cc:113-125, mel-computations.cc:33-142,253-259,
// matrix-functions.cc:592-608).
struct MfccTables {
  MfccOptions opts;
  int win = 0, shift = 0, padded = 0, nbins = 0, nceps = 0;
  std::vector<float> window;      // win
  std::vector<int> mel_offset;    // nbins: first FFT bin of each filter
  std::vector<int> mel_len;       // nbins
  std::vector<int> mel_start;     // nbins: offset into mel_weights
  std::vector<float> mel_weights; // concatenated
  std::vector<float> dct;         // nceps x nbins
  std::vector<float> lifter;      // nceps (1.0 if no liftering)
  float log_energy_floor = 0;
};
void ReadMfccOptions(const std::string &conf_path, MfccOptions *o);
void BuildMfccTables(const MfccOptions &o, MfccTables *t);
int NumFrames(long num_samples, const MfccOptions &o);  // feature-window.cc:42-87 (snip_edges only)

// feat/online-feature.h:200-260
struct CmvnOptions {
  int cmn_window = 600, speaker_frames = 600, global_frames = 200;
  bool normalize_mean = true, normalize_variance = false;
};

// online2/online-ivector-feature.h:60-160 + the files it names
struct IvectorExtractor {
  bool present = false;
  // config
  int ivector_period = 10, num_gselect = 5, num_cg_iters = 15;
  float min_post = 0.025f, posterior_scale = 0.1f, max_count = 0.0f, max_remembered_frames = 1000;
  bool use_most_recent_ivector = true, greedy = false, online_cmvn_iextractor = false;
  int splice_left = 0, splice_right = 0;
  CmvnOptions cmvn;
  // data
  MatF lda;                 // D_lda x (C*nsplice [+1])
  MatD global_cmvn;         // 2 x (C+1)
  std::vector<float> gconsts, weights;   // UBM (gconsts recomputed like DiagGmm::ComputeGconsts)
  MatF means_invvars, inv_vars;          // G x D_lda
  std::vector<MatD> M;                   // G x (D_lda x D_iv)
  std::vector<std::vector<double>> sigma_inv;  // G x packed(D_lda)
  double prior_offset = 0;
  // derived (ivector-extractor.cc:182-218)
  std::vector<double> U;            // G x D_iv(D_iv+1)/2
  std::vector<double> sigma_inv_M;  // G x D_lda x D_iv
  int feat_dim() const { return lda.rows; }
  int ivector_dim() const { return M.empty() ? 0 : M[0].cols; }
  int num_gauss() const { return (int)M.size(); }
  void ComputeDerived();
};

// online2/online-nnet2-feature-pipeline.h (OnlineNnet2FeaturePipelineConfig / Info)
struct FeatureConfig {
  std::string feature_type = "mfcc";
  MfccTables mfcc;
  bool use_cmvn = false;   // --cmvn-config on the nnet input branch
  CmvnOptions cmvn;
  MatD global_cmvn;
  IvectorExtractor ie;
  // Decoder / decodable options found in online.conf.  The reference registers them on the parser that reads --config
  // (online2-wav-nnet3-latgen-faster.cc:131-137, online2-cli-nnet3-decode-faster.cc:73-78), so they take effect there unless the
  // command line repeats them (util/parse-options.cc:328-345: the config file is read first).  Model::Model applies them.
  std::vector<std::pair<std::string, std::string>> decoder_conf;
  std::string conf_path;
};
void ReadFeatureConfig(const std::string &online_conf, FeatureConfig *fc);

// ---------------------------------------------------------------- transition model
// hmm/hmm-topology.cc:39-161, hmm/transition-model.cc:144-177,394-420
struct TransitionModel {
  int num_pdfs = 0;
  std::vector<int32_t> id2pdf;     // index = transition-id (0 unused)
  std::vector<int32_t> id2phone;
  // for the lattice tools of the rescoring path (lat/lattice-functions.cc:423-441, hmm/hmm-utils.cc:1065-1084):
  std::vector<int32_t> id2hmm_state;   // TransitionIdToHmmState
  std::vector<char> id2self_loop;      // IsSelfLoop
  std::vector<float> log_prob;         // GetTransitionLogProb
  std::vector<float> non_self_loop_log_prob;   // GetNonSelfLoopLogProb of the transition-id's transition-state
  // for graph construction (hmm/hmm-utils.cc:30-150,470-560): the topology and the tuple table themselves
  struct HmmState { int fwd = -1, self = -1; std::vector<std::pair<int, float>> trans; };   // pdf classes, (destination, prob)
  struct Tuple { int phone, hmm_state, fwd, self; };
  std::vector<std::vector<HmmState>> topo_entries;
  std::vector<int32_t> phone2entry;            // index = phone, -1 = no topology
  std::vector<Tuple> tuples;                   // transition-state k (1-based) = tuples[k - 1]
  std::vector<int32_t> tstate_first_tid;       // per transition-state (1-based; [0] unused): its first transition-id
  std::vector<int32_t> id2tstate;              // TransitionIdToTransitionState
  std::vector<int32_t> self_loop_of_id;        // per transition-id: SelfLoopOf(its transition-state), 0 = none
  int NumTransitionIds() const { return (int)id2pdf.size() - 1; }
  int TupleToTransitionState(int phone, int hmm_state, int fwd, int self) const;     // 0 = not found
  int NumPdfClasses(int phone) const;
  void Read(KaldiReader &r);
};

// ---------------------------------------------------------------- nnet3
// A parsed component blob.  `f` holds every basic value that followed each token, `m`/`v`/`iv` the
// matrices / vectors / integer vectors, `flags` the bare tokens.
struct Component {
  std::string type;
  std::map<std::string, std::vector<double>> f;      // basic values read as floating point
  std::map<std::string, std::vector<int64_t>> i;     // the same values read as integers (binary files do not say which)
  int Int(const std::string &k, int idx = 0) const { return (int)i.at(k).at(idx); }
  std::map<std::string, MatF> m;
  std::map<std::string, std::vector<float>> v;
  std::map<std::string, std::vector<int32_t>> iv;
  std::map<std::string, bool> b;
  bool Has(const std::string &k) const { return f.count(k) || m.count(k) || v.count(k) || iv.count(k) || b.count(k); }
};

// One term of a descriptor sum: scale * node[t + offset]; `const_t` = ReplaceIndex(x, t, 0) / Round (per-utterance row).
struct DescTerm {
  int node = -1;
  int offset = 0;
  float scale = 1.0f;
  bool const_t = false;
};
// Append(part0, part1, ...) where each part is a Sum of terms of equal dim.
struct DescPart {
  std::vector<DescTerm> terms;
  int dim = 0;
};
struct Descriptor {
  std::vector<DescPart> parts;
  int dim = 0;
};

struct NnetNode {
  enum Kind { kInput, kComponent, kOutput, kDimRange } kind = kInput;
  std::string name;
  int dim = 0;
  int component = -1;      // kComponent
  Descriptor input;        // kComponent / kOutput
  int range_node = -1, range_offset = 0;   // kDimRange
};

// Executable layer plan.  Every op produces one buffer with rows t in [-lext, T+rext) per utterance.
struct GemmSegment {
  int src_buf = -1;     // buffer index; -1 = iVector input (one row per utterance/chunk)
  int src_col = 0;      // first column inside the source buffer
  int ncols = 0;        // K extent of this segment
  int offset = 0;       // time offset
  int w_col = 0;        // first column in the weight matrix
};
struct EltStage {
  enum Kind { kRelu, kScaleOffset, kLogSoftmax, kNormalize, kScale } kind = kRelu;
  std::vector<float> scale, offset;   // kScaleOffset (BatchNorm test mode, ScaleAndOffset, PerElementScale...)
  float alpha = 1.0f;                 // kScale / kNormalize target_rms
};
struct SumTerm { int src_buf; int src_col; int offset; float scale; };
struct LayerOp {
  enum Kind { kGemm, kEltwise } kind = kGemm;
  std::string name;
  int out_buf = -1, out_dim = 0;
  // kGemm
  std::vector<GemmSegment> segs;
  MatF W;                       // out_dim x K (K = sum of segment widths, Kaldi column order)
  std::vector<float> bias;      // may be empty
  // a two-term Sum() folded into this GEMM (Nnet::Compile: the residual of a factorised TDNN layer, Sum(Scale(0.66, previous), this)):
  // out = stages(W x + b) + res_scale * res_buf[t], computed with the float operations of the elementwise op it replaces
  int res_buf = -1;
  float res_scale = 1.0f;
  // kEltwise: out = stages(sum_i scale_i * src_i[t+o_i])
  std::vector<SumTerm> terms;
  std::vector<EltStage> stages; // applied after the gemm/sum, in order
};
struct BufferInfo {
  int dim = 0, lext = 0, rext = 0;   // rows cover t in [-lext, T + rext)
  bool is_input = false;             // the MFCC ("input") buffer
  int stride = 1;                    // > 1 (--frame-subsampling-factor): only the rows t = 0 mod stride are read by anybody
};

struct Nnet {
  std::vector<NnetNode> nodes;
  std::vector<std::string> component_names;
  std::vector<Component> components;
  std::vector<float> priors;
  int input_dim = 0, ivector_dim = 0, output_dim = 0;
  // plan
  std::vector<LayerOp> ops;
  std::vector<BufferInfo> bufs;
  int input_buf = -1, output_buf = -1;
  int left_context = 0, right_context = 0;
  // What the reference's binaries do to the network before the first frame (nnet3_setup.h): the network is the one
  // CollapseModel leaves behind, and rand() has been called `setup_rand_calls` times (the dither seeds follow from that).
  long setup_rand_calls = 0;
  bool setup_rand_certain = true;
  std::string setup_rand_uncertain_why;
  void Read(KaldiReader &r, int frames_per_chunk = 24, int extra_left_context_initial = 0, int frame_subsampling_factor = 1);
  void Compile();   // builds ops/bufs for the "output" node
  void SetSubsampling(int factor);   // fills BufferInfo::stride for outputs wanted at t = 0, factor, 2 factor, ...
  int FindNode(const std::string &name) const;
};

// <NumComponents> ... </Nnet3> (nnet-nnet.cc:608-621)
void ReadNnetComponents(KaldiReader &r, std::vector<std::string> *names, std::vector<Component> *comps);

struct AcousticModel {
  TransitionModel trans;
  Nnet nnet;
  void Read(const std::string &final_mdl, int frames_per_chunk = 24, int extra_left_context_initial = 0, int frame_subsampling_factor = 1);
};

// ---------------------------------------------------------------- HCLG
// fstext/kaldi-fst-io.cc:51-91; openfst lib/fst.cc:58-82; const-fst.h:192-232; vector-fst.h
struct FstArc { int32_t ilabel, olabel; float weight; int32_t nextstate; };
struct Hclg {
  int32_t start = -1;
  std::vector<float> final_cost;     // per state (+inf = non-final)
  std::vector<uint32_t> arc_begin;   // per state, size n+1
  std::vector<uint32_t> num_ieps;    // per state: number of input-epsilon arcs (sorted first)
  std::vector<FstArc> arcs;          // ilabel-sorted within each state if the file was
  int num_states() const { return (int)final_cost.size(); }
  void Read(const std::string &path);
};

std::vector<std::string> ReadWordsTxt(const std::string &path);

}  // namespace rs
