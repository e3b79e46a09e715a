#include "engine.h"
#include "nnet3_setup.h"
#include "srfft_plan.h"
#include "lattice.h"
#include "env.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <sstream>
#include <thread>
#include <unordered_map>

namespace rs {

void HipCheck(hipError_t e, const char *what, const char *file, int line) {
  if (e != hipSuccess) {
    std::ostringstream os;
    os << "HIP error " << (int)e << " (" << hipGetErrorString(e) << ") at " << file << ":" << line << ": " << what;
    throw DeviceError(os.str());
  }
}

DeviceArena::~DeviceArena() { if (base_) (void)hipFree(base_); }

void DeviceArena::Reserve(size_t bytes, hipStream_t s) {
  if (bytes <= cap_) return;
  if (base_) {
    RS_HIP(hipStreamSynchronize(s));
    RS_HIP(hipFree(base_));
    base_ = nullptr;
    cap_ = 0;
  }
  size_t want = bytes + bytes / 8 + (1u << 20);
  RS_HIP(hipMalloc((void **)&base_, want));
  RS_HIP(hipMemsetAsync(base_, 0, want, s));
  cap_ = want;
  used_ = 0;
}

HostArena::~HostArena() { for (auto &b : blocks_) (void)hipHostFree(b.first); }

void *HostArena::Alloc(size_t bytes) {
  // blocks are never freed before destruction, so pointers handed out earlier in the same call stay valid when a new
  // block is appended; Reset() restarts from the first block
  for (;;) {
    if (cur_ < blocks_.size()) {
      const size_t a = (used_ + 63) & ~(size_t)63;
      if (a + bytes <= blocks_[cur_].second) { used_ = a + bytes; return blocks_[cur_].first + a; }
      cur_++;
      used_ = 0;
      continue;
    }
    const size_t want = std::max(bytes + bytes / 4, (size_t)1 << 20);
    char *p = nullptr;
    RS_HIP(hipHostMalloc((void **)&p, want, hipHostMallocDefault));
    blocks_.push_back({p, want});
  }
}

void *DeviceArena::Alloc(size_t bytes) {
  size_t a = (used_ + 255) & ~(size_t)255;
  if (a + bytes > cap_) Fail("internal error: device arena too small");
  used_ = a + bytes;
  return base_ + a;
}

static int RoundUp(int x, int m) { return (x + m - 1) / m * m; }

// ------------------------------------------------------------------------------------------------ model

// The reference registers the decodable's and the decoder's options on the parser that reads --config=online.conf
// (online2-wav-nnet3-latgen-faster.cc:131-137, online2-cli-nnet3-decode-faster.cc:73-78; nnet3/decodable-simple-looped.h:68-81,
// decoder/lattice-faster-decoder.h:67-85): ParseOptions reads the config file first and the command line overrides it
// (util/parse-options.cc:328-345).  Here rs_decode_opts is the command line: a field left at RS_OPT_UNSET takes online.conf's
// value, else the reference's default; a field that is set wins.  What the kernels cannot honour is refused, never dropped.
static int ConfInt(const std::string &v) {
  size_t n = 0;
  int x = 0;
  try { x = std::stoi(v, &n); } catch (...) { n = 0; }
  if (n == 0 || n != v.size()) Fail("Invalid integer option \"" + v + "\"");       // parse-options.cc:591
  return x;
}
static float ConfFloat(const std::string &v) {
  size_t n = 0;
  float x = 0;
  try { x = std::stof(v, &n); } catch (...) { n = 0; }
  if (n == 0 || n != v.size()) Fail("Invalid floating-point option \"" + v + "\"");
  return x;
}
static bool ConfBool(const std::string &v) {
  if (v == "true" || v == "t" || v == "1" || v == "True" || v == "T" || v.empty()) return true;
  if (v == "false" || v == "f" || v == "0" || v == "False" || v == "F") return false;
  Fail("Invalid format for boolean argument [expected true or false]: " + v);       // parse-options.cc:582
}

void Model::ResolveDecoderOptions() {
  const std::string where = " (online.conf: " + fc_.conf_path + ")";
  // values of online.conf, in file order (a repeated option: the last one wins, as in ParseOptions)
  bool has_beam = false, has_max = false, has_min = false, has_lb = false, has_bd = false, has_as = false, has_fpc = false, has_fsf = false;
  float c_beam = 0, c_lb = 0, c_bd = 0, c_as = 0;
  int c_max = 0, c_min = 0, c_fpc = 0, c_fsf = 0;
  // (an option of this kind given on the command line overrides the file's value: what the file says is then parsed, not refused)
  const int fixed = opts_.command_line_fixed;
  for (auto &kv : fc_.decoder_conf) {
    const std::string &k = kv.first, &v = kv.second;
    if (k == "beam") { c_beam = ConfFloat(v); has_beam = true; }
    else if (k == "max-active") { c_max = ConfInt(v); has_max = true; }
    else if (k == "min-active") { c_min = ConfInt(v); has_min = true; }
    else if (k == "lattice-beam") { c_lb = ConfFloat(v); has_lb = true; }
    else if (k == "beam-delta") { c_bd = ConfFloat(v); has_bd = true; }
    else if (k == "acoustic-scale") { c_as = ConfFloat(v); has_as = true; }
    else if (k == "frames-per-chunk") { c_fpc = ConfInt(v); has_fpc = true; }
    else if (k == "frame-subsampling-factor") { c_fsf = ConfInt(v); has_fsf = true; }
    else if (k == "extra-left-context-initial") {
      const int x = ConfInt(v);
      if (x < 0) Fail("KALDI_ASSERT: at Check:decodable-simple-looped.h:62, failed: extra_left_context_initial >= 0 && frame_subsampling_factor > 0 && frames_per_chunk > 0 && acoustic_scale > 0.0" + where);
      if (x != 0 && !(fixed & RS_FIXED_EXTRA_LEFT_CONTEXT_INITIAL)) Fail("--extra-left-context-initial=" + v + " is not supported by the HIP path (only 0, the reference's default)" + where);
    } else if (k == "prune-interval") {
      const int x = ConfInt(v);
      if (x != 25 && !(fixed & RS_FIXED_PRUNE_INTERVAL)) Fail("--prune-interval=" + v + " is not supported by the HIP path (only 25, the reference's default: lattices are pruned once, exactly, at the end)" + where);
    } else if (k == "determinize-lattice") {
      if (!ConfBool(v) && !(fixed & RS_FIXED_DETERMINIZE_LATTICE)) Fail("--determinize-lattice=false is not supported by the HIP path (lattices are always determinised)" + where);
    } else if (k == "hash-ratio") {
      if (!(ConfFloat(v) >= 1.0f)) Fail("KALDI_ASSERT: at Check:lattice-faster-decoder.h:87, failed: hash_ratio >= 1.0" + where);
    } else if (k == "online") {
      if (ConfBool(v) && !(fixed & RS_FIXED_ONLINE)) Fail("--online=true is not supported by the HIP path (the reference passes --online=false on its command line)" + where);
    } else if (k == "do-endpointing") {
      if (ConfBool(v) && !(fixed & RS_FIXED_DO_ENDPOINTING)) Fail("--do-endpointing=true is not supported by the HIP path (the reference passes --do-endpointing=false on its command line)" + where);
    } else if (k == "minimize" || k == "phone-determinize" || k == "word-determinize" || k == "debug-computation") {
      (void)ConfBool(v);      // same n-best lists either way (the emitted lattice is equivalent, not minimised)
    } else if (k == "max-mem" || k == "num-threads-startup") {
      (void)ConfInt(v);
    } else if (k == "chunk-length" || k == "delta") {
      (void)ConfFloat(v);     // chunk-length only paces the wav binary's simulated online loop with --online=true
    }
  }
  auto unset_f = [](float x) { return x == (float)RS_OPT_UNSET; };
  auto unset_i = [](int x) { return x == RS_OPT_UNSET; };
  if (unset_f(opts_.beam)) opts_.beam = has_beam ? c_beam : 16.0f;                                   // lattice-faster-decoder.h:56-63
  if (unset_i(opts_.max_active)) opts_.max_active = has_max ? c_max : 0x7fffffff;
  if (unset_i(opts_.min_active)) opts_.min_active = has_min ? c_min : 200;
  if (unset_f(opts_.lattice_beam)) opts_.lattice_beam = has_lb ? c_lb : 10.0f;
  if (unset_f(opts_.beam_delta)) opts_.beam_delta = has_bd ? c_bd : 0.5f;
  if (unset_f(opts_.acoustic_scale)) opts_.acoustic_scale = has_as ? c_as : 0.1f;                    // decodable-simple-looped.h:55-59
  if (unset_i(opts_.frames_per_chunk)) opts_.frames_per_chunk = has_fpc ? c_fpc : 24;
  if (unset_i(opts_.frame_subsampling_factor)) opts_.frame_subsampling_factor = has_fsf ? c_fsf : 1;
  // NnetSimpleLoopedComputationOptions::Check (decodable-simple-looped.h:61-65) / LatticeFasterDecoderConfig::Check
  // (lattice-faster-decoder.h:86-91): the reference's binaries abort on these
  if (!(opts_.frame_subsampling_factor > 0 && opts_.frames_per_chunk > 0 && opts_.acoustic_scale > 0.0f))
    Fail("KALDI_ASSERT: at Check:decodable-simple-looped.h:62, failed: extra_left_context_initial >= 0 && frame_subsampling_factor > 0 && frames_per_chunk > 0 && acoustic_scale > 0.0"
         " (frame-subsampling-factor " + std::to_string(opts_.frame_subsampling_factor) + ", frames-per-chunk " + std::to_string(opts_.frames_per_chunk) +
         ", acoustic-scale " + std::to_string(opts_.acoustic_scale) + ")");
  if (!(opts_.beam > 0.0f && opts_.max_active > 1 && opts_.lattice_beam > 0.0f && opts_.min_active <= opts_.max_active && opts_.beam_delta > 0.0f))
    Fail("KALDI_ASSERT: at Check:lattice-faster-decoder.h:87, failed: beam > 0.0 && max_active > 1 && lattice_beam > 0.0 && min_active <= max_active"
         " && prune_interval > 0 && beam_delta > 0.0 && hash_ratio >= 1.0 && prune_scale > 0.0 && prune_scale < 1.0"
         " (beam " + std::to_string(opts_.beam) + ", max-active " + std::to_string(opts_.max_active) + ", min-active " + std::to_string(opts_.min_active) +
         ", lattice-beam " + std::to_string(opts_.lattice_beam) + ", beam-delta " + std::to_string(opts_.beam_delta) + ")");
  // GetChunkSize (nnet-compile-looped.cc:81-94): the advised chunk rounded up to a multiple of the subsampling factor (and of the
  // network's modulus, which is 1 for every network the layer plan accepts)
  while (opts_.frames_per_chunk % opts_.frame_subsampling_factor != 0) opts_.frames_per_chunk++;
}

Model::Model(const std::string &final_mdl, const std::string &hclg, const std::string &online_conf,
             const rs_decode_opts &opts)
    : opts_(opts) {
  if (const char *e = TuneEnv("RS_FORCE_SPARSE_DECODER")) force_sparse_ = e[0] == '1';
  if (const char *e = std::getenv("RS_SUBBATCHES")) max_groups_ = std::atoi(e);
  if (const char *e = std::getenv("RS_DECODER")) {
    const std::string v(e);
    // reg / dense: the LDS-resident searches of small graphs; sparse: DecodeKernel alone (dense per-state tables in HBM); hash: the
    // token-list search with the live-state table, which is what "auto" runs on graphs the first two cannot hold
    decoder_choice_ = v == "reg" ? 1 : v == "dense" ? 2 : v == "sparse" ? 3 : v == "hash" ? 4 : 0;
    if (decoder_choice_ >= 3) force_sparse_ = true;
  }
  ReadFeatureConfig(online_conf, &fc_);
  ResolveDecoderOptions();
  am_.Read(final_mdl, opts_.frames_per_chunk, 0, opts_.frame_subsampling_factor);
  hclg_.Read(hclg);
  dither_rand_calls_ = am_.nnet.setup_rand_calls;
  if (const char *e = std::getenv("RS_DITHER_RAND_CALLS")) dither_rand_calls_ = std::atol(e);
  else if (fc_.mfcc.opts.dither != 0.0f && !am_.nnet.setup_rand_certain)
    Fail("mfcc.conf has --dither=" + std::to_string(fc_.mfcc.opts.dither) + " and the reference seeds that noise from rand() after a model "
         "set-up whose number of rand() calls cannot be derived for this network (" + am_.nnet.setup_rand_uncertain_why +
         ": its looped compilation may need a second attempt, nnet-compile-looped.cc:326-345); set --dither=0 in the mfcc config, or "
         "RS_DITHER_RAND_CALLS=<n> to the count `rs-dump randpos` reports for the reference");
  const Nnet &n = am_.nnet;
  if (n.input_dim != fc_.mfcc.nceps)
    Fail("Input feature dimension mismatch: got " + std::to_string(fc_.mfcc.nceps) + " but network expects " + std::to_string(n.input_dim));
  int ivd = fc_.ie.present ? fc_.ie.ivector_dim() : 0;
  if (n.ivector_dim != ivd) {
    if (n.ivector_dim > 0 && ivd == 0) Fail("Ivector feature dimension mismatch: got -1 but network expects " + std::to_string(n.ivector_dim));
    Fail("Ivector feature dimension mismatch: got " + std::to_string(ivd) + " but network expects " + std::to_string(n.ivector_dim));
  }
  // every emitting arc must name a valid transition-id
  int ntid = (int)am_.trans.id2pdf.size();
  for (auto &a : hclg_.arcs)
    if (a.ilabel >= ntid) Fail("HCLG refers to transition-id " + std::to_string(a.ilabel) + " but the model has only " + std::to_string(ntid - 1));
  // arcs must be ilabel-sorted within a state so that the epsilon arcs come first (mkgraph.sh output is)
  arc_ilabel_.resize(hclg_.arcs.size());
  for (int s = 0; s < hclg_.num_states(); s++) {
    auto b = hclg_.arcs.begin() + hclg_.arc_begin[s], e = hclg_.arcs.begin() + hclg_.arc_begin[s + 1];
    if (!std::is_sorted(b, e, [](const FstArc &x, const FstArc &y) { return x.ilabel < y.ilabel; }))
      std::stable_sort(b, e, [](const FstArc &x, const FstArc &y) { return x.ilabel < y.ilabel; });
    uint32_t ne = 0;
    for (auto it = b; it != e; ++it) if (it->ilabel == 0) ne++;
    hclg_.num_ieps[s] = ne;
  }
  for (size_t i = 0; i < hclg_.arcs.size(); i++) arc_ilabel_[i] = hclg_.arcs[i].ilabel;
  L_ = n.left_context;
  R_ = n.right_context;
  if (fc_.ie.present) {
    L_ = std::max(L_, fc_.ie.splice_left);
    R_ = std::max(R_, fc_.ie.splice_right);
    // splice + LDA as a segmented GEMM over the (padded) feature buffer: buffers 0 = raw, 1 = cmvn (filled in per batch)
    const IvectorExtractor &ie = fc_.ie;
    int C = fc_.mfcc.nceps, nsp = ie.splice_left + 1 + ie.splice_right;
    lda_op_.kind = LayerOp::kGemm;
    lda_op_.name = "ivector.splice+lda";
    lda_op_.out_dim = ie.lda.rows;
    lda_op_.W.Resize(ie.lda.rows, C * nsp);
    bool affine = ie.lda.cols == C * nsp + 1;
    if (affine) lda_op_.bias.resize(ie.lda.rows);
    for (int r = 0; r < ie.lda.rows; r++) {
      for (int c = 0; c < C * nsp; c++) lda_op_.W(r, c) = ie.lda(r, c);
      if (affine) lda_op_.bias[r] = ie.lda(r, C * nsp);
    }
    for (int i = 0; i < nsp; i++) {
      GemmSegment sg;
      sg.src_buf = 0; sg.src_col = 0; sg.ncols = C; sg.offset = i - ie.splice_left; sg.w_col = i * C;
      lda_op_.segs.push_back(sg);
    }
    lda_op1_ = lda_op_;
    lda_op1_.name = "ivector.splice+lda (one run)";
    lda_op1_.segs.clear();
    {
      GemmSegment sg;
      sg.src_buf = 0; sg.src_col = 0; sg.ncols = C * nsp; sg.offset = -ie.splice_left; sg.w_col = 0;
      lda_op1_.segs.push_back(sg);
    }
  }
  PruneOutputLayer();
  for (auto &op : am_.nnet.ops) {
    if (op.kind == LayerOp::kGemm && (int)op.segs.size() > kMaxSegs) Fail("nnet3: layer " + op.name + " has too many input segments");
    if ((int)op.stages.size() > kMaxStages) Fail("nnet3: layer " + op.name + " has too many fused stages");
    if (op.kind == LayerOp::kEltwise && op.terms.size() > 8) Fail("nnet3: layer " + op.name + " sums too many terms");
  }
}

// rs_decode_opts.prune_output_pdfs: the search looks log-likelihoods up by the pdf of the arc it crosses
// (DecodableNnetLoopedOnline::LogLikelihood, decodable-online-looped.cc:213-224), so a pdf that is on no HCLG arc is never
// read.  Grammar graphs use a fraction of the model's pdfs (362 of 2000 for the bench grammar): the output affine is cut
// down to those rows and the arcs' pdf ids are renumbered.  Only when the output node IS that affine (a log-softmax or
// any other row-wise stage after it needs every pdf) and only when at least 30 % of the pdfs go.
void Model::PruneOutputLayer() {
  if (!opts_.prune_output_pdfs || opts_.keep_intermediates) return;
  Nnet &n = am_.nnet;
  if (n.ops.empty()) return;
  LayerOp &op = n.ops.back();
  if (op.kind != LayerOp::kGemm || op.out_buf != n.output_buf || op.out_dim != n.output_dim) return;
  for (auto &st : op.stages)
    if (st.kind == EltStage::kLogSoftmax || st.kind == EltStage::kNormalize) return;
  const int P = n.output_dim;
  std::vector<int> remap(P, -1);
  int count = 0;
  for (auto &a : hclg_.arcs)
    if (a.ilabel > 0) {
      const int pdf = am_.trans.id2pdf[a.ilabel];
      if (pdf < 0 || pdf >= P) Fail("HCLG refers to pdf " + std::to_string(pdf) + " but the model has " + std::to_string(P));
      if (remap[pdf] < 0) remap[pdf] = 0, count++;
    }
  if (count == 0 || (long)count * 10 > (long)P * 7) return;
  count = 0;
  for (int p = 0; p < P; p++) if (remap[p] >= 0) remap[p] = count++;
  MatF W;
  W.Resize(count, op.W.cols);
  std::vector<float> bias(op.bias.empty() ? 0 : count), priors(n.priors.empty() ? 0 : count);
  for (int p = 0; p < P; p++) {
    if (remap[p] < 0) continue;
    std::memcpy(&W.d[(size_t)remap[p] * W.cols], &op.W.d[(size_t)p * op.W.cols], sizeof(float) * W.cols);
    if (!bias.empty()) bias[remap[p]] = op.bias[p];
    if (!priors.empty()) priors[remap[p]] = n.priors[p];
  }
  for (auto &st : op.stages)
    if (st.kind == EltStage::kScaleOffset) {
      std::vector<float> sc(count), of(count);
      for (int p = 0; p < P; p++) if (remap[p] >= 0) { sc[remap[p]] = st.scale[p]; of[remap[p]] = st.offset[p]; }
      st.scale.swap(sc);
      st.offset.swap(of);
    }
  op.W = std::move(W);
  op.bias.swap(bias);
  if (!priors.empty()) n.priors.swap(priors);
  op.out_dim = count;
  n.bufs[n.output_buf].dim = count;
  n.output_dim = count;
  pdf_remap_.swap(remap);
  pruned_from_ = P;
}

Model::~Model() {
  for (void *p : owned_) (void)hipFree(p);
  for (void *p : dither_bufs_) (void)hipFree(p);
  for (auto &c : ctx_) {
    if (c->h_pcm_pinned) (void)hipHostFree(c->h_pcm_pinned);
    if (c->d_pcm) (void)hipFree(c->d_pcm);
    for (auto &ab : c->lat_arcs) if (ab.d) (void)hipFree(ab.d);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream_dec) (void)hipStreamDestroy(c->stream_dec);
    for (auto &e : c->slab_ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : c->stage_ev) if (e) (void)hipEventDestroy(e);
    if (c->split_ev) (void)hipEventDestroy(c->split_ev);
    if (c->gemm_ovf) (void)hipHostFree(c->gemm_ovf);
  }
  if (stream_ctx_ && stream_ctx_->gemm_ovf) (void)hipHostFree(stream_ctx_->gemm_ovf);
}

MfccDev Model::MfccWithDither(int frames) {
  MfccDev m = mfcc_dev_;
  m.dither = nullptr;
  m.dither_value = fc_.mfcc.opts.dither;
  if (m.dither_value == 0.0f || frames <= 0) return m;
  {
    std::lock_guard<std::mutex> lk(dither_mu_);
    if (frames <= dither_frames_) { m.dither = d_dither_; return m; }
  }
  // Growth (rare: the table doubles).  One grower at a time; calls that fit the current table are not held up while the noise is
  // generated (dither_mu_ is only taken to read or publish the pointer).  Size: win floats per frame of the longest utterance or
  // stream so far -- 1.6 KB per frame, 0.58 GB per hour of audio.
  std::lock_guard<std::mutex> grow(dither_grow_mu_);
  int old;
  const float *d_old;
  {
    std::lock_guard<std::mutex> lk(dither_mu_);
    old = dither_frames_; d_old = d_dither_;
  }
  if (frames > old) {
    const int want = std::max({frames, 2 * old, 1024});
    const size_t win = (size_t)fc_.mfcc.win;
    std::vector<float> host((size_t)(want - old) * win);
    {
      // host libm like the reference (the values are those of its RandGauss to the bit); a few worker threads for long audio
      const int nthr = std::min(8, std::max(1, (want - old) / 512));
      std::vector<std::thread> thr;
      for (int k = 0; k < nthr; k++) {
        const int a = old + (int)((long)(want - old) * k / nthr), b = old + (int)((long)(want - old) * (k + 1) / nthr);
        thr.emplace_back([&, a, b]() { DitherNoise(dither_rand_calls_, a, b, (int)win, host.data() + (size_t)(a - old) * win); });
      }
      for (auto &t : thr) t.join();
    }
    float *d = nullptr;
    RS_HIP(hipMalloc((void **)&d, sizeof(float) * (size_t)want * win));
    if (old) RS_HIP(hipMemcpy(d, d_old, sizeof(float) * (size_t)old * win, hipMemcpyDeviceToDevice));
    RS_HIP(hipMemcpy(d + (size_t)old * win, host.data(), sizeof(float) * host.size(), hipMemcpyHostToDevice));
    // Launches in flight may still read the table this one supersedes, so that one stays; the ones before it go -- after a
    // device-wide wait, which a handful of growth steps in a model's life can afford.
    if (dither_bufs_.size() >= 2) {
      RS_HIP(hipDeviceSynchronize());
      while (dither_bufs_.size() >= 2) { (void)hipFree(dither_bufs_.front()); dither_bufs_.erase(dither_bufs_.begin()); }
    }
    dither_bufs_.push_back(d);
    std::lock_guard<std::mutex> lk(dither_mu_);
    d_dither_ = d;
    dither_frames_ = want;
  }
  std::lock_guard<std::mutex> lk(dither_mu_);
  m.dither = d_dither_;
  return m;
}

void *Model::UploadBytes(const void *p, size_t bytes) {
  void *d = nullptr;
  RS_HIP(hipMalloc(&d, std::max<size_t>(bytes, 16)));
  owned_.push_back(d);
  if (bytes) RS_HIP(hipMemcpy(d, p, bytes, hipMemcpyHostToDevice));
  return d;
}
template <typename T> T *Model::Upload(const std::vector<T> &v) { return static_cast<T *>(UploadBytes(v.data(), v.size() * sizeof(T))); }

void Model::BuildGemmPlan(const LayerOp &op, GemmPlan *plan) {
  plan->op = &op;
  int k = 0;
  plan->seg_k0.clear();
  for (auto &sg : op.segs) { plan->seg_k0.push_back(k); k += RoundUp(sg.ncols, kGemmBK); }
  plan->k_pad = k;
  plan->n_pad = RoundUp(op.out_dim, kGemmBN);
  std::vector<float> W((size_t)plan->n_pad * plan->k_pad, 0.0f);
  for (size_t si = 0; si < op.segs.size(); si++) {
    const GemmSegment &sg = op.segs[si];
    for (int r = 0; r < op.out_dim; r++)
      std::memcpy(&W[(size_t)r * plan->k_pad + plan->seg_k0[si]], &op.W.d[(size_t)r * op.W.cols + sg.w_col], sizeof(float) * sg.ncols);
  }
  plan->d_W = Upload(W);
  // Split-fp16 image (nnet_gemm_b3.hip): column n of W times a power of two s_n that puts its largest weight into [2^14, 2^15)
  // (both fp16 parts of every weight that matters are then normal numbers), w s_n = w1 + w2 with fp16 parts (round to nearest
  // even), stored per (16-wide k-step, 32-column tile, part) as one 1 KiB MFMA B fragment [k-group 2][column 32][8 fp16].  The
  // kernels' epilogue multiplies column n by 1 / s_n (exact).
  plan->n3 = RoundUp(op.out_dim, 256);
  plan->d_W3 = nullptr;
  plan->d_W3I = nullptr;
  plan->d_w3_inv_scale = nullptr;
  if (op.out_dim >= 96 && GemmB3PaddingOk(op.out_dim, plan->n3)) {
    auto to_f16 = [](float x) -> uint16_t {          // round to nearest even, subnormals kept; |x| < 65520 here
      uint32_t u;
      std::memcpy(&u, &x, 4);
      const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
      const int e = (int)((u >> 23) & 0xff) - 127;
      uint32_t m = u & 0x7fffffu;
      if (e < -25) return sign;
      if (e < -14) {                                  // subnormal result: value = q 2^-24
        m |= 0x800000u;
        const int sh = -e - 1;                        // q = m >> sh (14 .. 24)
        uint32_t q = m >> sh;
        const uint32_t rem = m & ((1u << sh) - 1u), halfway = 1u << (sh - 1);
        if (rem > halfway || (rem == halfway && (q & 1u))) q++;
        return (uint16_t)(sign | q);
      }
      uint32_t h = ((uint32_t)(e + 15) << 10) | (m >> 13);
      const uint32_t rem = m & 0x1fffu;
      if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;      // a carry into the exponent is the right answer
      return (uint16_t)(sign | h);
    };
    auto from_f16 = [](uint16_t h) -> float {
      const int e = (h >> 10) & 0x1f;
      const uint32_t m = h & 0x3ffu;
      const float v = e == 0 ? std::ldexp((float)m, -24) : std::ldexp((float)(m | 0x400u), e - 25);
      return (h & 0x8000u) ? -v : v;
    };
    std::vector<float> col_scale(plan->n3, 1.0f), inv_scale(plan->n3, 1.0f);
    bool finite = true;
    for (int n = 0; n < op.out_dim; n++) {
      float mx = 0.0f;
      for (int k = 0; k < plan->k_pad; k++) {
        const float a = std::fabs(W[(size_t)n * plan->k_pad + k]);
        if (!std::isfinite(a)) finite = false;
        mx = std::max(mx, a);
      }
      if (!(mx > 0.0f) || !finite) continue;
      int ex;
      std::frexp(mx, &ex);                            // mx = f 2^ex, f in [0.5, 1)
      const int sh = std::min(std::max(15 - ex, -100), 100);
      col_scale[n] = std::ldexp(1.0f, sh);
      inv_scale[n] = std::ldexp(1.0f, -sh);
    }
    const int nct = plan->n3 / 32;
    // k-step order: segment after segment, or -- when every segment is a row-shifted view of the same columns of one
    // buffer -- alternating between the segments (step t = segment t % nsegs, columns 16 (t / nsegs))
    const int nsegs = (int)op.segs.size();
    plan->interleave = nsegs > 1;
    for (auto &sg : op.segs)
      plan->interleave = plan->interleave && sg.src_buf >= 0 && sg.src_buf == op.segs[0].src_buf && sg.src_col == op.segs[0].src_col &&
                         sg.ncols == op.segs[0].ncols;
    // image of W for a list of k-steps (first W column of each): [k-step][32-column tile][part][k-group 2][column 32][8 fp16]
    constexpr int P = kActImageParts;
    auto build = [&](const std::vector<int> &step_k) {
      const int nks = (int)step_k.size();
      std::vector<uint16_t> W3((size_t)(nks + 2) * nct * P * 512, 0);      // + 2 k-steps the kernels' pipelines request past the end
      for (int n = 0; n < op.out_dim; n++)
        for (int t = 0; t < nks; t++)
          for (int kk = 0; kk < 16; kk++) {
            const float w = W[(size_t)n * plan->k_pad + step_k[t] + kk] * col_scale[n];
            if (w == 0.0f) continue;
            const uint16_t h1 = to_f16(w);
            const uint16_t h2 = to_f16(w - from_f16(h1));
            const size_t base = ((size_t)t * nct + n / 32) * P * 512 + (size_t)(kk / 8) * 256 + (size_t)(n % 32) * 8 + kk % 8;
            W3[base] = h1; W3[base + 512] = h2;
          }
      return UploadBytes(W3.data(), W3.size() * sizeof(uint16_t));
    };
    // (1) GemmKernelB3: every segment spans its width padded to kGemmBK
    if (finite) {
      plan->d_w3_inv_scale = Upload(inv_scale);
      const int nks = plan->k_pad / 16;
      std::vector<int> step_k(nks);
      for (int t = 0; t < nks; t++) step_k[t] = plan->interleave ? plan->seg_k0[t % nsegs] + (t / nsegs) * 16 : t * 16;
      plan->d_W3 = build(step_k);
    }
    // (2) GemmKernelB3I (sources stored as operand images): segments padded to the 16-wide k-step only; needs every source
    // to be a frame buffer whose first column sits on a k-step boundary
    // ... and the layer to be one the split-bf16 kernels take at all (GemmB3IUsable's padding rule: at most a quarter of the
    // 256-column tiles may be padding) -- decided HERE, because the producers of its sources stop storing plain floats
    bool imageable = finite && GemmB3PaddingOk(op.out_dim, plan->n3);
    for (auto &sg : op.segs) imageable = imageable && sg.src_buf >= 0 && sg.src_col % 16 == 0;
    if (imageable) {
      std::vector<int> step_k;
      if (plan->interleave) {
        const int per = (op.segs[0].ncols + 15) / 16;
        for (int t = 0; t < per * nsegs; t++) step_k.push_back(plan->seg_k0[t % nsegs] + (t / nsegs) * 16);
      } else {
        for (int si = 0; si < nsegs; si++)
          for (int k = 0; k < (op.segs[si].ncols + 15) / 16; k++) step_k.push_back(plan->seg_k0[si] + k * 16);
      }
      plan->d_W3I = build(step_k);
    }
  }
  plan->d_bias = op.bias.empty() ? nullptr : Upload(op.bias);
  plan->d_stage.clear();
  for (auto &st : op.stages) {
    float *s = nullptr, *o = nullptr;
    if (st.kind == EltStage::kScaleOffset) { s = Upload(st.scale); o = Upload(st.offset); }
    plan->d_stage.emplace_back(s, o);
  }
}

void Model::ToDevice() {
  std::lock_guard<std::mutex> lk(mu_);
  if (on_device_) return;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    throw DeviceError("no HIP device available: librhasspy_speech_hip has no CPU fallback (hipGetDeviceCount: " +
                      std::string(hipGetErrorString(e)) + ")");
  if (opts_.device_id < 0 || opts_.device_id >= ndev) throw DeviceError("device_id " + std::to_string(opts_.device_id) + " out of range");
  RS_HIP(hipSetDevice(opts_.device_id));
  hipDeviceProp_t prop;
  RS_HIP(hipGetDeviceProperties(&prop, opts_.device_id));
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
    throw DeviceError(std::string("this library is built for gfx950 (MI355X) only; device reports ") + prop.gcnArchName);
  {
    const char *e = std::getenv("RS_CONTEXTS");
    const int n = std::min(std::max(e ? std::atoi(e) : 4, 1), 8);      // arenas are allocated on first use
    // the search of a slab is latency-bound and must not queue behind the thousands of GEMM workgroups of the next slab
    int prio_low = 0, prio_high = 0;
    RS_HIP(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
    for (int i = 0; i < n; i++) {
      std::unique_ptr<DecodeContext> c(new DecodeContext());
      RS_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
      RS_HIP(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
      RS_HIP(hipStreamCreateWithPriority(&c->stream_dec, hipStreamNonBlocking, prio_high));
      for (auto &ev : c->slab_ev) RS_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      for (auto &ev : c->stage_ev) RS_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      RS_HIP(hipEventCreateWithFlags(&c->split_ev, hipEventDisableTiming));
      RS_HIP(hipHostMalloc((void **)&c->gemm_ovf, 64, hipHostMallocMapped));
      c->gemm_ovf[0] = c->gemm_ovf[1] = 0;
      RS_HIP(hipHostGetDevicePointer((void **)&c->gemm_ovf_dev, c->gemm_ovf, 0));
      ctx_.push_back(std::move(c));
    }
  }
  // ---- MFCC tables
  const MfccTables &t = fc_.mfcc;
  mfcc_dev_.win = t.win; mfcc_dev_.shift = t.shift; mfcc_dev_.padded = t.padded; mfcc_dev_.nbins = t.nbins; mfcc_dev_.nceps = t.nceps;
  mfcc_dev_.preemph = t.opts.preemph; mfcc_dev_.remove_dc = t.opts.remove_dc; mfcc_dev_.use_energy = t.opts.use_energy;
  mfcc_dev_.raw_energy = t.opts.raw_energy; mfcc_dev_.log_energy_floor = t.log_energy_floor;
  mfcc_dev_.window = Upload(t.window);
  mfcc_dev_.mel_offset = Upload(t.mel_offset); mfcc_dev_.mel_len = Upload(t.mel_len); mfcc_dev_.mel_start = Upload(t.mel_start);
  {
    std::vector<int> rec((size_t)t.nbins * 4, 0);
    for (int b = 0; b < t.nbins; b++) { rec[4 * b] = t.mel_offset[b]; rec[4 * b + 1] = t.mel_len[b]; rec[4 * b + 2] = t.mel_start[b]; }
    mfcc_dev_.mel_rec = static_cast<const int4 *>(UploadBytes(rec.data(), rec.size() * sizeof(int)));
  }
  mfcc_dev_.mel_weights = Upload(t.mel_weights);
  mfcc_dev_.dct = Upload(t.dct);
  mfcc_dev_.lifter = Upload(t.lifter);
  {
    const SrfftPlan pl = BuildSrfftPlan(t.padded);
    if ((int)pl.level_begin.size() - 1 > 15) Fail("FFT plan has too many levels");
    static_assert(sizeof(SrfftTask) == 16, "SrfftTask is read as an int4");
    mfcc_dev_.fft_tasks = static_cast<const int *>(UploadBytes(pl.tasks.data(), pl.tasks.size() * sizeof(SrfftTask)));
    mfcc_dev_.fft_num_levels = (int)pl.level_begin.size() - 1;
    mfcc_dev_.fft_num_tasks = (int)pl.tasks.size();
    mfcc_dev_.fft_num_tw = (int)(pl.tw.size() / 6);
    for (size_t i = 0; i < pl.level_begin.size(); i++) mfcc_dev_.fft_level_begin[i] = pl.level_begin[i];
    mfcc_dev_.fft_tw = Upload(pl.tw.empty() ? std::vector<float>(6, 0.f) : pl.tw);
    std::vector<int> pl_perm_swz = pl.perm;      // the bit-reversal gather, in the layout the transform leaves its points in
    mfcc_dev_.fft_swz[0] = mfcc_dev_.fft_swz[1] = mfcc_dev_.fft_swz[2] = 0;
    {
      bool narrow = true;
      for (size_t l = 0; l + 1 < pl.level_begin.size(); l++) narrow = narrow && pl.level_begin[l + 1] - pl.level_begin[l] <= 64;
      mfcc_dev_.fft_recs = nullptr;
      if (narrow) {
        // 64 records per level, one per lane: {byte offsets of the task's points in the swizzled layout; kind (3: none) | twiddle
        // class << 8; six factors; -}.  The swizzle (feat_kernels.hip, SrfftRunRec): exhaustive search over the linear ones for a
        // 256-point plan (profiles/micro/fft_swizzle.py); other sizes keep the plain layout.
        const size_t nl = pl.level_begin.size() - 1;
        const int N = t.padded / 2;
        if (N == 256) { mfcc_dev_.fft_swz[0] = 14; mfcc_dev_.fft_swz[1] = 21; mfcc_dev_.fft_swz[2] = 1; }
        auto swz = [&](int i) { return i ^ ((i & 32) ? mfcc_dev_.fft_swz[0] : 0) ^ ((i & 64) ? mfcc_dev_.fft_swz[1] : 0) ^ ((i & 128) ? mfcc_dev_.fft_swz[2] : 0); };
        std::vector<float> recs(nl * 64 * 12, 0.f);
        for (size_t l = 0; l < nl; l++)
          for (int j = 0; j < 64; j++) {
            int head[5] = {0, 0, 0, 0, 3};
            const int ti = pl.level_begin[l] + j;
            if (ti < pl.level_begin[l + 1]) {
              const SrfftTask &tk = pl.tasks[ti];
              const int kind = tk.kind_logm & 0xff, lg = tk.kind_logm >> 8;
              int pts[4] = {tk.off, tk.off + 1, tk.off + 2, tk.off + 3};
              if (kind == 0) { const int mm = 1 << lg, e0 = tk.off + tk.n; pts[0] = e0; pts[1] = e0 + mm / 4; pts[2] = e0 + mm / 2; pts[3] = e0 + mm / 2 + mm / 4; }
              if (kind == 2) { pts[2] = pts[0]; pts[3] = pts[1]; }
              for (int q = 0; q < 4; q++) head[q] = 4 * swz(pts[q]);
              head[4] = kind | ((tk.tw >= 0 ? 0 : tk.tw == -1 ? 1 : 2) << 8);
              if (tk.tw >= 0) std::memcpy(&recs[(l * 64 + j) * 12 + 5], &pl.tw[(size_t)tk.tw * 6], 24);
            }
            std::memcpy(&recs[(l * 64 + j) * 12], head, 20);
          }
        mfcc_dev_.fft_recs = static_cast<const float4 *>(UploadBytes(recs.data(), recs.size() * sizeof(float)));
        for (int &v : pl_perm_swz) v = swz(v);
      }
    }
    mfcc_dev_.fft_perm = Upload(pl_perm_swz);
    mfcc_dev_.fft_kn = Upload(pl.kn);
    mfcc_dev_.fft_post = nullptr;
    if (t.padded == 512) {
      std::vector<float> post(2 * 64 * 4, 0.f);
      for (int q = 0; q < 2; q++)
        for (int lane = 0; lane < 64; lane++) {
          const int k = lane + 1 + 64 * q, pk = pl_perm_swz[k], pd = pl_perm_swz[256 - k];
          float *r = &post[(size_t)(q * 64 + lane) * 4];
          std::memcpy(r, &pk, 4); std::memcpy(r + 1, &pd, 4);
          r[2] = pl.kn[2 * k]; r[3] = pl.kn[2 * k + 1];
        }
      mfcc_dev_.fft_post = static_cast<const float4 *>(UploadBytes(post.data(), post.size() * sizeof(float)));
    }
  }
  if (t.nceps > 128) Fail("more than 128 cepstral coefficients are not supported");
  // ---- CMVN on the nnet input branch
  if (fc_.use_cmvn) {
    cmvn_nnet_dev_.dim = t.nceps; cmvn_nnet_dev_.cmn_window = fc_.cmvn.cmn_window;
    cmvn_nnet_dev_.speaker_frames = fc_.cmvn.speaker_frames; cmvn_nnet_dev_.global_frames = fc_.cmvn.global_frames;
    cmvn_nnet_dev_.global_stats = Upload(fc_.global_cmvn.d);
  }
  // ---- iVector extractor
  if (fc_.ie.present) {
    const IvectorExtractor &ie = fc_.ie;
    cmvn_iv_dev_.dim = t.nceps; cmvn_iv_dev_.cmn_window = ie.cmvn.cmn_window; cmvn_iv_dev_.speaker_frames = ie.cmvn.speaker_frames;
    cmvn_iv_dev_.global_frames = ie.cmvn.global_frames;
    cmvn_iv_dev_.global_stats = Upload(ie.global_cmvn.d);
    int G = ie.num_gauss(), D = ie.feat_dim();
    if (G > 2048) Fail("iVector extractor: more than 2048 Gaussians are not supported");
    if (D > 128) Fail("iVector extractor: feature dim > 128 is not supported");
    if (ie.num_gselect > 8) Fail("iVector extractor: num-gselect > 8 is not supported");
    std::vector<float> mt((size_t)D * G), vt((size_t)D * G);
    for (int g = 0; g < G; g++)
      for (int d = 0; d < D; d++) { mt[(size_t)d * G + g] = ie.means_invvars(g, d); vt[(size_t)d * G + g] = ie.inv_vars(g, d); }
    ivec_dev_.feat_dim = D; ivec_dev_.ivec_dim = ie.ivector_dim(); ivec_dev_.num_gauss = G; ivec_dev_.num_gselect = ie.num_gselect;
    ivec_dev_.num_cg_iters = ie.num_cg_iters; ivec_dev_.min_post = ie.min_post; ivec_dev_.posterior_scale = ie.posterior_scale;
    ivec_dev_.max_count = ie.max_count; ivec_dev_.prior_offset = ie.prior_offset;
    ivec_dev_.gconsts = Upload(ie.gconsts);
    ivec_dev_.means_invvars_t = Upload(mt);
    ivec_dev_.inv_vars_t = Upload(vt);
    ivec_dev_.ubm_bm = ivec_dev_.ubm_bv = nullptr;
    ivec_dev_.ubm_kg = 0;
    if (G <= 512 && D <= 48) {      // UbmPostMfmaKernel's operand tables
      const int kg = D <= 16 ? 1 : 3, nt_real = (G + 15) / 16, nt = nt_real <= 2 ? 2 : (nt_real <= 8 ? 8 : 32);
      std::vector<float> bm((size_t)nt * kg * 256, 0.f), bv(bm.size(), 0.f);
      for (int j = 0; j < nt; j++)
        for (int k = 0; k < kg; k++)
          for (int l = 0; l < 64; l++)
            for (int i = 0; i < 4; i++) {
              const int d = 16 * k + 4 * i + (l >> 4), gi = 16 * j + (l & 15);
              if (d < D && gi < G) {
                bm[((size_t)(j * kg + k) * 64 + l) * 4 + i] = ie.means_invvars(gi, d);
                bv[((size_t)(j * kg + k) * 64 + l) * 4 + i] = ie.inv_vars(gi, d);
              }
            }
      ivec_dev_.ubm_bm = Upload(bm);
      ivec_dev_.ubm_bv = Upload(bv);
      ivec_dev_.ubm_kg = kg;
    }
    ivec_dev_.sigma_inv_M = Upload(ie.sigma_inv_M);
    ivec_dev_.U = Upload(ie.U);
    BuildGemmPlan(lda_op_, &lda_plan_);
    if (TuneEnv("RS_LDA_ONE_RUN") == nullptr || std::atoi(TuneEnv("RS_LDA_ONE_RUN")) != 0) BuildGemmPlan(lda_op1_, &lda_plan1_);
  }
  // ---- nnet
  gemm_plans_.assign(am_.nnet.ops.size(), GemmPlan());
  for (size_t i = 0; i < am_.nnet.ops.size(); i++) BuildGemmPlan(am_.nnet.ops[i], &gemm_plans_[i]);
  {
    // operand images: a buffer gets one when a split-bf16 layer reads it through GemmKernelB3I; it is still stored as plain
    // floats when anything else reads it (the nnet's input / output, an elementwise op, a layer on another kernel)
    const Nnet &nn = am_.nnet;
    buf_image_.assign(nn.bufs.size(), 0);
    buf_f32_.assign(nn.bufs.size(), 0);
    buf_f32_[nn.input_buf] = 1;
    buf_f32_[nn.output_buf] = 1;
    for (size_t i = 0; i < nn.ops.size(); i++) {
      const LayerOp &op = nn.ops[i];
      if (op.kind == LayerOp::kGemm) {
        // (the nnet's input buffer is written by the feature kernels, not by a layer: a layer that reads it splits on the fly)
        bool by_image = gemm_plans_[i].d_W3I != nullptr;
        for (auto &sg : op.segs) by_image = by_image && sg.src_buf >= 0 && sg.src_buf != nn.input_buf;
        for (auto &sg : op.segs)
          if (sg.src_buf >= 0) (by_image ? buf_image_ : buf_f32_)[sg.src_buf] = 1;
        if (op.res_buf >= 0) {
          // a folded residual: through the source's operand image when the layer is image-fed (RS_RESIDUAL_IMAGE=0, read per model
          // load: as plain floats, the elementwise op's exact operands), else as plain floats
          const char *e = std::getenv("RS_RESIDUAL_IMAGE");
          gemm_plans_[i].res_by_image = by_image && op.res_buf != nn.input_buf && !(e && std::atoi(e) == 0);
          (gemm_plans_[i].res_by_image ? buf_image_ : buf_f32_)[op.res_buf] = 1;
        }
      } else {
        for (auto &t : op.terms) buf_f32_[t.src_buf] = 1;
      }
    }
  }
  if (!am_.nnet.priors.empty()) {
    std::vector<float> lp(am_.nnet.priors.size());
    for (size_t i = 0; i < lp.size(); i++) lp[i] = logf(am_.nnet.priors[i]);
    d_log_priors_ = Upload(lp);
  }
  // ---- HCLG
  {
    size_t A = hclg_.arcs.size();
    std::vector<int4> arcs(A);
    std::vector<int> src(A);
    for (int s = 0; s < hclg_.num_states(); s++)
      for (uint32_t a = hclg_.arc_begin[s]; a < hclg_.arc_begin[s + 1]; a++) {
        const FstArc &fa = hclg_.arcs[a];
        int pdf1 = fa.ilabel == 0 ? 0 : am_.trans.id2pdf[fa.ilabel] + 1;
        if (pdf1 > 0 && !pdf_remap_.empty()) pdf1 = pdf_remap_[pdf1 - 1] + 1;
        int wbits;
        std::memcpy(&wbits, &fa.weight, 4);
        arcs[a] = make_int4(pdf1, fa.olabel, wbits, fa.nextstate);
        src[a] = s;
      }
    hclg_dev_.num_states = hclg_.num_states();
    hclg_dev_.num_arcs = (int)A;
    hclg_dev_.start = hclg_.start;
    hclg_dev_.arc_begin = Upload(hclg_.arc_begin);
    hclg_dev_.num_ieps = Upload(hclg_.num_ieps);
    hclg_has_eps_ = false;
    for (auto n : hclg_.num_ieps) hclg_has_eps_ = hclg_has_eps_ || n != 0;
    {
      const int S = hclg_.num_states();
      std::vector<uint4> rec(S);
      for (int s = 0; s < S; s++) {
        const uint32_t b = hclg_.arc_begin[s], e = hclg_.arc_begin[s + 1], ne = hclg_.num_ieps[s];
        rec[s] = make_uint4(b, ne, e - b - ne, 0u);
      }
      hclg_dev_.state_rec = static_cast<uint4 *>(UploadBytes(rec.data(), (size_t)S * sizeof(uint4)));
    }
    hclg_dev_.arcs = static_cast<int4 *>(UploadBytes(arcs.data(), A * sizeof(int4)));
    if (A < ((size_t)1 << 30)) {      // the live-state-table search's copies (decode_live.hip)
      const int S = hclg_.num_states();
      std::vector<char> eps_dst(S, 0);
      for (size_t a = 0; a < A; a++) if (arcs[a].x == 0) eps_dst[arcs[a].w] = 1;
      std::vector<int4> af(arcs);
      for (size_t a = 0; a < A; a++) {
        if (hclg_.num_ieps[af[a].w] != 0) af[a].x |= (int)0x80000000;
        if (eps_dst[af[a].w]) af[a].x |= 0x40000000;
      }
      hclg_dev_.arcs_f = static_cast<int4 *>(UploadBytes(af.data(), A * sizeof(int4)));
      std::vector<uint4> nodes((size_t)S * 4, make_uint4(0u, 0u, 0u, 0u));
      for (int s = 0; s < S; s++) {
        const uint32_t b = hclg_.arc_begin[s], e = hclg_.arc_begin[s + 1], ne = hclg_.num_ieps[s];
        nodes[(size_t)s * 4] = make_uint4(b, ne, e - b - ne, 0u);
        for (uint32_t k = 0; k < 2 && b + ne + k < e; k++) {
          const int4 &x = af[b + ne + k];
          nodes[(size_t)s * 4 + 1 + k] = make_uint4((unsigned)x.x, (unsigned)x.y, (unsigned)x.z, (unsigned)x.w);
        }
      }
      hclg_dev_.nodes = static_cast<uint4 *>(UploadBytes(nodes.data(), nodes.size() * sizeof(uint4)));
    }
    hclg_dev_.arc_src = Upload(src);
    {
      std::vector<int> srcx(A);
      for (size_t a = 0; a < A; a++) srcx[a] = src[a] | (arcs[a].x == 0 ? (int)0x80000000 : 0);
      hclg_dev_.arc_srcx = Upload(srcx);
    }
    hclg_dev_.final_cost = Upload(hclg_.final_cost);
    // reverse graph for the dense (pull) decoder: in-arcs per destination state, in forward-arc order
    dense_ok_ = DenseDecodeFits(hclg_.num_states(), am_.nnet.output_dim);
    if (dense_ok_) {
      const int S = hclg_.num_states();
      std::vector<uint32_t> be(S + 1, 0), bx(S + 1, 0);
      for (size_t a = 0; a < A; a++) (hclg_.arcs[a].ilabel == 0 ? bx : be)[hclg_.arcs[a].nextstate + 1]++;
      for (int st = 0; st < S; st++) { be[st + 1] += be[st]; bx[st + 1] += bx[st]; }
      std::vector<int4> ie(be[S]), ix(bx[S]);
      std::vector<uint32_t> fe(be.begin(), be.end() - 1), fx(bx.begin(), bx.end() - 1);
      for (size_t a = 0; a < A; a++) {
        const int4 &fa = arcs[a];
        int4 rec = make_int4(src[a], fa.x, fa.z, (int)a);
        if (fa.x == 0) ix[fx[fa.w]++] = rec; else ie[fe[fa.w]++] = rec;
      }
      std::vector<int> eps_dst;
      for (int st = 0; st < S; st++) if (bx[st + 1] > bx[st]) eps_dst.push_back(st);
      rev_dev_.in_begin_e = Upload(be);
      rev_dev_.in_begin_x = Upload(bx);
      rev_dev_.in_e = static_cast<int4 *>(UploadBytes(ie.data(), ie.size() * sizeof(int4)));
      rev_dev_.in_x = static_cast<int4 *>(UploadBytes(ix.data(), ix.size() * sizeof(int4)));
      rev_dev_.eps_dst = Upload(eps_dst);
      rev_dev_.num_eps_dst = (int)eps_dst.size();
      rev_dev_.in_begin_e_host_total = (int)ie.size();
      rev_dev_.in_begin_x_host_total = (int)ix.size();
      // register-resident variant (decode_reg.hip): arcs dealt out to threads in forward order
      const int P = am_.nnet.output_dim;
      int nt = 0, ke = 0, kx = 0;
      if (P > 0 && RegDecodeConfig(S, (int)ie.size(), (int)ix.size(), &nt, &ke, &kx)) {
        const int key_base = (int)(((size_t)(S + 1) * 4 + 15) & ~(size_t)15);
        const int pad_e = (4 * S) | ((key_base + 8 * S) << 16), pad_x = (key_base + 8 * S + 4) | ((key_base + 8 * S) << 16);
        std::vector<int4> et((size_t)ke * nt, make_int4(pad_e, 0, 0, 0)), xt((size_t)kx * nt, make_int4(pad_x, 0, 0, 0));
        std::vector<int> eaux((size_t)ke * nt, 0), xaux((size_t)kx * nt, 0);
        std::vector<int> first_e(S, -1), cnt_e(S, 0), cnt_x(S, 0);
        size_t ne = 0, nx = 0;
        for (size_t a = 0; a < A; a++) {
          const int4 &fa = arcs[a];
          const int dst_addr = (key_base + 8 * fa.w) << 16;
          if (fa.x == 0) {
            xaux[nx] = cnt_x[src[a]]++;
            xt[nx++] = make_int4((key_base + 8 * src[a] + 4) | dst_addr, 0, fa.z, (int)a);
          } else {
            if (first_e[src[a]] < 0) first_e[src[a]] = (int)ne;
            eaux[ne] = (first_e[src[a]] << 8) | (cnt_e[src[a]]++ & 255);
            et[ne++] = make_int4((4 * src[a]) | dst_addr, fa.x - 1, fa.z, (int)a);
          }
        }
        int max_e = 0, max_x = 0;
        for (int st = 0; st < S; st++) { max_e = std::max(max_e, cnt_e[st]); max_x = std::max(max_x, cnt_x[st]); }
        // longest path of the epsilon subgraph = number of closure rounds; cyclic or deep -> the kernel votes instead
        int depth = 0;
        {
          std::vector<int> indeg(S, 0), len(S, 0), order;
          std::vector<std::vector<int>> out(S);
          for (size_t a = 0; a < A; a++) if (arcs[a].x == 0) { out[src[a]].push_back(arcs[a].w); indeg[arcs[a].w]++; }
          for (int st = 0; st < S; st++) if (indeg[st] == 0) order.push_back(st);
          for (size_t i = 0; i < order.size(); i++)
            for (int d : out[order[i]]) { len[d] = std::max(len[d], len[order[i]] + 1); if (--indeg[d] == 0) order.push_back(d); }
          if ((int)order.size() < S) depth = -1;
          else { for (int st = 0; st < S; st++) depth = std::max(depth, len[st]); if (depth > 6) depth = -1; }
        }
        reg_dev_.nt = nt; reg_dev_.ke = ke; reg_dev_.kx = kx;
        reg_dev_.eps_depth = depth;
        reg_dev_.key_base = key_base;
        reg_dev_.e_tab = static_cast<int4 *>(UploadBytes(et.data(), et.size() * sizeof(int4)));
        reg_dev_.x_tab = static_cast<int4 *>(UploadBytes(xt.data(), xt.size() * sizeof(int4)));
        reg_dev_.e_aux = Upload(eaux);
        reg_dev_.x_aux = Upload(xaux);
        // the reference's token order can be followed exactly where its hash table cannot collide (it starts with 1000 buckets),
        // the closure is one round, and a state's arcs fit one 32-bit mask (decode_reg.hip: RegDecodeExactKernel)
        reg_dev_.exact_ok = (S <= 1000 && depth >= 0 && depth <= 1 && max_e <= 32 && max_x <= 32) ? 1 : 0;
      }
    }
  }
  RS_HIP(hipDeviceSynchronize());
  on_device_ = true;
}

std::string Model::Describe() const {
  std::ostringstream os;
  const MfccTables &t = fc_.mfcc;
  os << "mfcc: win=" << t.win << " shift=" << t.shift << " padded=" << t.padded << " mel_bins=" << t.nbins << " ceps=" << t.nceps
     << " dither=" << t.opts.dither << "\n";
  os << "nnet_input_cmvn: " << (fc_.use_cmvn ? 1 : 0) << "\n";
  if (fc_.ie.present)
    os << "ivector: dim=" << fc_.ie.ivector_dim() << " gauss=" << fc_.ie.num_gauss() << " lda_dim=" << fc_.ie.feat_dim()
       << " splice=" << fc_.ie.splice_left << "," << fc_.ie.splice_right << " gselect=" << fc_.ie.num_gselect
       << " max_count=" << fc_.ie.max_count << " prior_offset=" << fc_.ie.prior_offset << "\n";
  else os << "ivector: none\n";
  const Nnet &n = am_.nnet;
  os << "nnet: input_dim=" << n.input_dim << " ivector_dim=" << n.ivector_dim << " output_dim=" << n.output_dim
     << " left_context=" << n.left_context << " right_context=" << n.right_context << " priors=" << n.priors.size()
     << " components=" << n.components.size() << " ops=" << n.ops.size() << "\n";
  for (auto &op : n.ops) {
    os << "op: " << (op.kind == LayerOp::kGemm ? "gemm" : "eltwise") << " " << op.name << " out_dim=" << op.out_dim;
    if (op.kind == LayerOp::kGemm) {
      os << " k=" << op.W.cols << " segs=";
      for (auto &s : op.segs) os << "[" << s.src_buf << ":" << s.offset << ":" << s.ncols << "]";
      if (op.res_buf >= 0) os << " residual=" << op.res_scale << "*[" << op.res_buf << "]";
    } else {
      os << " terms=" << op.terms.size();
    }
    os << " stages=" << op.stages.size();
    if (n.bufs[op.out_buf].stride > 1) os << " rows=every-" << n.bufs[op.out_buf].stride;      // (--frame-subsampling-factor: nothing reads the rows between)
    os << "\n";
  }
  os << "transition_model: tids=" << am_.trans.id2pdf.size() - 1 << " pdfs=" << am_.trans.num_pdfs << "\n";
  if (pruned_from_) os << "output layer: pruned to the " << am_.nnet.output_dim << " of " << pruned_from_ << " pdfs that occur on HCLG arcs\n";
  os << "hclg: states=" << hclg_.num_states() << " arcs=" << hclg_.arcs.size() << " start=" << hclg_.start << "\n";
  os << "halo: L=" << L_ << " R=" << R_ << "\n";
  os << "decoder_opts: beam=" << opts_.beam << " max_active=" << opts_.max_active << " min_active=" << opts_.min_active << " lattice_beam=" << opts_.lattice_beam
     << " beam_delta=" << opts_.beam_delta << " acoustic_scale=" << opts_.acoustic_scale << " frames_per_chunk=" << opts_.frames_per_chunk
     << " frame_subsampling_factor=" << opts_.frame_subsampling_factor << "\n";
  // (state, not structure: calls repeated on the exact-FP32 layer GEMMs because an activation left the fp16 split's range, and
  // whether the model has changed to those kernels for good)
  os << "token_order: " << (ExactOrder() && reg_dev_.exact_ok ? "exact (the reference's running cutoff in its hash order)" : "final cutoff")
     << (ExactOrder() && !reg_dev_.exact_ok ? " (exact_token_order asked for: not applicable to this graph)" : "") << "\n";
  os << "layer_gemm: range_retries=" << range_retries_.load() << " precision_retries=" << precision_retries_.load() << " exact_fp32=" << (exact_gemm_.load() ? 1 : 0)
     << " regime=" << (exact_gemm_.load() ? "exact-fp32" : "split-fp16")
     << " (split-fp16 carries an operand row to 2^-22 of its largest element while that element lies in [2^-3, 65520): a row with |x| >= 65520, infinity"
        " or NaN, or a non-zero row whose largest |x| is below 2^-3, raises a flag and the call is repeated on the exact-FP32 kernels; the third such call"
        " makes them permanent)\n";
  return os.str();
}

// ------------------------------------------------------------------------------------------------ batch


// Decode calls may overlap on the device: calls on one model up to its number of decode contexts (RS_CONTEXTS, 4), calls
// on different models freely.  The latency-bound search of one batch leaves the CUs to the GEMMs of the next: 3.3 ms per
// headline batch with four calls in flight instead of 4.2.
//
// Overlap needed one fix that is NOT in this file.  Waves executing packed FP32 VALU instructions (v_pk_add_f32 /
// v_pk_mul_f32, which the compiler forms by itself when it vectorises scalar float code) produced wrong results in
// 16-lane groups whenever MFMA-issuing waves of another kernel (GemmKernelB3 of another call) shared their CU: isolated
// MFCC frames came out perturbed, first noticed as two models disturbing each other (DESIGN.md section 5 has the
// evidence; profiles/micro/stress_*.py reproduce it in seconds when feat_kernels.hip is built without NOPACK).  The
// Makefile therefore builds the feature and the small nnet kernels with -fno-slp-vectorize -fno-vectorize; with that,
// 60 000 results of concurrent calls (one model, two models) match their sequential values.  Two older, blunter
// protections remain as switches: RS_MFCC_EXCLUSIVE=1 (the feature kernel takes whole CUs while other calls are in
// flight) and RS_PROCESS_LOCK=1 (different models take turns).
namespace {
class ProcessTurn {
 public:
  explicit ProcessTurn(const void *model) {
    static const bool on = std::getenv("RS_PROCESS_LOCK") != nullptr;
    if (!on) return;
    std::unique_lock<std::mutex> lk(Mu());
    bool counted = false;      // a model waiting for its turn stops the owner's further calls from overlapping (no starvation)
    while (!(Owner() == nullptr || (Owner() == model && Waiters() == 0))) {
      if (!counted && Owner() != model) { Waiters()++; counted = true; }
      Cv().wait(lk);
    }
    if (counted) Waiters()--;
    Owner() = model;
    Depth()++;
    held_ = true;
  }
  ~ProcessTurn() {
    if (!held_) return;
    { std::lock_guard<std::mutex> lk(Mu()); if (--Depth() == 0) Owner() = nullptr; }
    Cv().notify_all();
  }
 private:
  static std::mutex &Mu() { static std::mutex m; return m; }
  static std::condition_variable &Cv() { static std::condition_variable c; return c; }
  static const void *&Owner() { static const void *o = nullptr; return o; }
  static int &Depth() { static int d = 0; return d; }
  static int &Waiters() { static int w = 0; return w; }
  bool held_ = false;
};
}  // namespace

std::unique_ptr<Result> Model::DecodeBatchHost(const int16_t *const *pcm, const int32_t *n_samples, int n_utts, int nbest,
                                               float lat_scale, bool streaming) {
  ToDevice();
  std::vector<int64_t> off(n_utts + 1, 0);
  for (int i = 0; i < n_utts; i++) {
    if (n_samples[i] < 0 || (n_samples[i] > 0 && pcm[i] == nullptr)) Fail("rs_decode_batch: bad sample buffer for utterance " + std::to_string(i));
    off[i + 1] = off[i] + n_samples[i];
  }
  ProcessTurn turn(this);
  DecodeContext *cx = AcquireContext();
  std::unique_ptr<Result> res;
  try {
    RS_HIP(hipSetDevice(opts_.device_id));
    size_t total = (size_t)off[n_utts] + 512;
    if (total > cx->h_pcm_cap) {
      if (cx->h_pcm_pinned) RS_HIP(hipHostFree(cx->h_pcm_pinned));
      if (cx->d_pcm) RS_HIP(hipFree(cx->d_pcm));
      cx->h_pcm_pinned = nullptr; cx->d_pcm = nullptr;
      cx->h_pcm_cap = total + total / 4;
      RS_HIP(hipHostMalloc((void **)&cx->h_pcm_pinned, cx->h_pcm_cap * sizeof(int16_t), hipHostMallocDefault));
      RS_HIP(hipMalloc((void **)&cx->d_pcm, cx->h_pcm_cap * sizeof(int16_t)));
    }
    auto t0 = std::chrono::steady_clock::now();
    // Pageable caller buffers -> pinned staging -> HBM in spans of about 16 MB: the DMA of one span runs under the host copy of
    // the next, and nothing waits here -- the feature kernel is ordered behind the last span on the context's stream.
    static const int64_t span = [] { const char *e = TuneEnv("RS_UPLOAD_SPAN_KB"); return (int64_t)(e && std::atol(e) > 0 ? std::atol(e) : 16384) * 512; }();      // samples per span (16 MB: 2 MB spans cost the mixed workload 5 % in copy calls, one span for everything 6 % in lost overlap)
    // The copies of consecutive calls are chained like the stages (engine.h: stage 2): a call's hipMemcpyAsync waits, on the
    // device, for the previous call's.  Calls that start at the same moment otherwise submit their copies at the same moment, and
    // one of them was then seen to spend 6.6-9 ms INSIDE hipMemcpyAsync (profiles/r03/soak.txt); with the copies ordered that does
    // not happen: 20 steps from a standing start 2.70 -> 2.55 ms per step on average, steady state unchanged.
    static const bool up_chain = [] { const char *e = TuneEnv("RS_STAGE_CHAIN"); return !e || std::atoi(e) != 0; }();
    std::unique_lock<std::mutex> up_lock;
    if (up_chain) {
      up_lock = std::unique_lock<std::mutex>(stage_mu_[2]);
      if (stage_tail_[2] && stage_tail_[2] != cx->stage_ev[2]) RS_HIP(hipStreamWaitEvent(cx->stream, stage_tail_[2], 0));
    }
    for (int i = 0, first = 0; i < n_utts; i++) {
      if (n_samples[i]) std::memcpy(cx->h_pcm_pinned + off[i], pcm[i], sizeof(int16_t) * (size_t)n_samples[i]);
      if (off[i + 1] - off[first] >= span || i + 1 == n_utts) {
        if (off[i + 1] > off[first])
          RS_HIP(hipMemcpyAsync(cx->d_pcm + off[first], cx->h_pcm_pinned + off[first], sizeof(int16_t) * (size_t)(off[i + 1] - off[first]),
                                hipMemcpyHostToDevice, cx->stream));
        first = i + 1;
      }
    }
    if (up_chain) {
      RS_HIP(hipEventRecord(cx->stage_ev[2], cx->stream));
      stage_tail_[2] = cx->stage_ev[2];
      up_lock.unlock();
    }
    const float h2d_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    res = DecodeInContext(*cx, cx->d_pcm, off.data(), n_utts, nbest, lat_scale, nullptr, streaming);
    res->timings[0] = h2d_ms;
    res->timings[6] += h2d_ms;
  } catch (...) {
    ReleaseContext(cx);
    throw;
  }
  ReleaseContext(cx);
  return res;
}

std::unique_ptr<Result> Model::DecodeBatchDevice(const int16_t *d_pcm, const int64_t *sample_offsets, int n_utts, int nbest,
                                                 float lat_scale, hipStream_t user_stream, bool streaming) {
  ToDevice();
  ProcessTurn turn(this);
  DecodeContext *cx = AcquireContext();
  std::unique_ptr<Result> res;
  try {
    res = DecodeInContext(*cx, d_pcm, sample_offsets, n_utts, nbest, lat_scale, user_stream, streaming);
  } catch (...) {
    ReleaseContext(cx);
    throw;
  }
  ReleaseContext(cx);
  return res;
}

static std::atomic<int> g_calls_in_flight{0};      // decode calls of any model of this process

// Which layer GEMMs the work this thread is enqueueing uses: sampled from the model once per utterance group / stream advance
// (a call must not change kernels half way: producers and consumers agree on who stores operand images).
thread_local bool tls_exact_gemm = false;
thread_local int *tls_gemm_ovf_dev = nullptr;
void SampleGemmMode(bool exact, int *ovf_dev) { tls_exact_gemm = exact; tls_gemm_ovf_dev = ovf_dev; }
bool Model::CheckGemmRange(DecodeContext &cx) {
  if (tls_exact_gemm || !cx.gemm_ovf) return false;
  // (RS_GEMM_B3_NOUNDER=1, -DRS_TUNING builds: the precision flag is ignored -- what the split kernels do to small rows, measured)
  static const bool no_under = [] { const char *e = TuneEnv("RS_GEMM_B3_NOUNDER"); return e && std::atoi(e) != 0; }();
  const int over = static_cast<volatile int *>(cx.gemm_ovf)[0], under = no_under ? 0 : static_cast<volatile int *>(cx.gemm_ovf)[1];
  if (over == 0 && under == 0) return false;
  if (under != 0) precision_retries_.fetch_add(1);
  if (range_retries_.fetch_add(1) + 1 >= 3) exact_gemm_.store(true);
  return true;
}

Model::DecodeContext *Model::AcquireContext() {
  std::unique_lock<std::mutex> lk(ctx_mu_);
  for (;;) {
    for (auto &c : ctx_) if (!c->busy) { c->busy = true; g_calls_in_flight.fetch_add(1); return c.get(); }
    ctx_cv_.wait(lk);
  }
}
// RS_MFCC_EXCLUSIVE=1: true when another decode call of this process is in flight right now; the feature kernel then keeps
// GEMM workgroups off its CUs (a call that starts alone cannot meet another call's nnet stage during its own, first,
// 0.3 ms feature stage, because that call would have to be in flight already).
bool Model::OthersInFlight() {
  static const bool on = [] { const char *e = TuneEnv("RS_MFCC_EXCLUSIVE"); return e && std::atoi(e) != 0; }();
  return on && g_calls_in_flight.load() > 1;
}
void Model::ReleaseContext(DecodeContext *cx) {
  { std::lock_guard<std::mutex> lk(ctx_mu_); cx->busy = false; g_calls_in_flight.fetch_sub(1); }
  ctx_cv_.notify_one();
}

std::unique_ptr<Result> Model::DecodeInContext(DecodeContext &cx, const int16_t *d_pcm, const int64_t *sample_offsets, int n_utts, int nbest,
                                               float lat_scale, hipStream_t user_stream, bool streaming) {
  RS_HIP(hipSetDevice(opts_.device_id));
  if (nbest < 1) Fail("nbest must be >= 1");
  auto wall0 = std::chrono::steady_clock::now();
  std::unique_ptr<Result> res(new Result());
  res->utts.resize(n_utts);
  if (n_utts == 0) return res;
  // two concurrent groups unless the caller pinned a stream, the batch is small, or RS_SUBBATCHES=1
  const int ngroups = (user_stream || n_utts < 32 || max_groups_ < 2) ? 1 : 2;
  cx.active_groups = ngroups;
  for (int attempt = 0;; attempt++) {
  cx.gemm_ovf[0] = cx.gemm_ovf[1] = 0;
  cx.force_exact = attempt > 0;
  try {
  if (ngroups == 1) {
    DecodeGroup(cx, 0, d_pcm, sample_offsets, n_utts, nbest, lat_scale, user_stream ? user_stream : cx.stream, streaming,
                res->utts.data(), res->timings);
  } else {
    const int half = (n_utts + 1) / 2;
    float t2[2][8] = {{0}, {0}};
    std::exception_ptr err[2];
    // the second group's stream starts behind whatever the first one's holds for this call (DecodeBatchHost: the copies of the
    // samples to the device, which nothing else orders group 1's feature kernel behind)
    RS_HIP(hipEventRecord(cx.split_ev, cx.stream));
    RS_HIP(hipStreamWaitEvent(cx.stream2, cx.split_ev, 0));
    auto run = [&](int gi) {
      try {
        const int u0 = gi == 0 ? 0 : half, n = gi == 0 ? half : n_utts - half;
        DecodeGroup(cx, gi, d_pcm, sample_offsets + u0, n, nbest, lat_scale, gi == 0 ? cx.stream : cx.stream2, streaming,
                    res->utts.data() + u0, t2[gi]);
      } catch (...) {
        err[gi] = std::current_exception();
      }
    };
    std::thread th(run, 1);
    run(0);
    th.join();
    for (int gi = 0; gi < 2; gi++) if (err[gi]) std::rethrow_exception(err[gi]);
    for (int k = 0; k < 8; k++) res->timings[k] = t2[0][k] + t2[1][k];   // stage times add up; they overlap in wall time
  }
  break;
  } catch (const RangeRetry &) {
    // (both groups are over: the throwing group's exception is only rethrown after the join)
    if (attempt > 0) Fail("layer GEMM range check failed on the exact-FP32 kernels");
    for (auto &u : res->utts) u = UttResult();
  }
  }
  res->timings[6] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - wall0).count();
  return res;
}

size_t Model::ImageBytes(int rows) const {
  size_t b = 0;
  const int guard = RoundUp(L_ + R_ + 8, 32);
  for (size_t i = 0; i < buf_image_.size(); i++)
    if (buf_image_[i]) b += kActImageParts * ActImagePartBytes(rows, guard, am_.nnet.bufs[i].dim) + 1024;
  return b;
}

std::vector<ActImage> Model::AllocImages(DeviceArena &arena, int rows) const {
  std::vector<ActImage> imgs(buf_image_.size(), ActImage{nullptr, 0, 0, 0});
  const int guard = RoundUp(L_ + R_ + 8, 32);
  for (size_t i = 0; i < buf_image_.size(); i++) {
    if (!buf_image_[i]) continue;
    ActImage &im = imgs[i];
    im.part_bytes = ActImagePartBytes(rows, guard, am_.nnet.bufs[i].dim);
    im.base = static_cast<unsigned char *>(arena.Alloc(kActImageParts * im.part_bytes));
    im.nks = (am_.nnet.bufs[i].dim + 15) / 16;
    im.guard = guard;
  }
  return imgs;
}

void Model::ZeroGuards(const std::vector<float *> &bufp, const std::vector<int> &buf_ld, int rows, const std::vector<ActImage> *imgs, hipStream_t s) const {
  ZeroRegions z;
  z.count = 0;
  auto add = [&](void *p, size_t bytes) {
    if (bytes == 0) return;
    if (z.count == ZeroRegions::kMax) { LaunchZeroRegions(z, s); z.count = 0; }
    z.r[z.count].p = p; z.r[z.count].bytes = bytes; z.count++;
  };
  const int guard = L_ + R_ + 8;                            // rows in front of / behind a frame buffer (DecodeGroup, StreamsAdvance: falloc)
  for (size_t b = 0; b < bufp.size(); b++) {
    if (!bufp[b]) continue;
    const size_t g = (size_t)guard * buf_ld[b] * sizeof(float);
    add(bufp[b] - (size_t)guard * buf_ld[b], g);
    add(bufp[b] + (size_t)rows * buf_ld[b], g);
  }
  if (imgs)
    for (const ActImage &im : *imgs) {
      if (!im.base) continue;
      // image rows [0, guard) and from the row block that holds image row guard + rows on (its rows of real frames are
      // written after this by their producer)
      const size_t blk = (size_t)im.nks * 1024, head = (size_t)(im.guard / 32) * blk, tail0 = (size_t)((im.guard + rows) / 32) * blk;
      for (int p = 0; p < kActImageParts; p++) {
        add(im.base + p * im.part_bytes, head);
        add(im.base + p * im.part_bytes + tail0, im.part_bytes - tail0);
      }
    }
  if (z.count) LaunchZeroRegions(z, s);
}

GemmDev Model::MakeGemm(const GemmPlan &pl, const std::vector<float *> &src, const std::vector<int> &src_ld, float *ivec, int ivec_ld, float *out,
                        int ldo, int share, const std::vector<ActImage> *imgs, int out_buf) const {
  GemmDev d;
  std::memset(&d, 0, sizeof(d));
  const LayerOp &op = *pl.op;
  const bool images_on = imgs != nullptr && GemmImagesEnabled() && !tls_exact_gemm && tls_gemm_ovf_dev != nullptr;
  d.nsegs = (int)op.segs.size();
  for (int i = 0; i < d.nsegs; i++) {
    const GemmSegment &sg = op.segs[i];
    GemmSegDev &o = d.segs[i];
    if (sg.src_buf < 0) { o.src = ivec; o.ld = ivec_ld; o.per_utt = 1; o.row_off = 0; }
    else {
      o.src = src[sg.src_buf]; o.ld = src_ld[sg.src_buf]; o.per_utt = 0; o.row_off = sg.offset;
      if (images_on) o.img = (*imgs)[sg.src_buf];
    }
    o.col0 = sg.src_col; o.ncols = sg.ncols; o.k0 = pl.seg_k0[i];
  }
  d.W = pl.d_W; d.k_pad = pl.k_pad; d.n = op.out_dim; d.n_pad = pl.n_pad; d.bias = pl.d_bias;
  d.W3 = (tls_exact_gemm || !tls_gemm_ovf_dev) ? nullptr : pl.d_W3; d.n3 = pl.n3; d.interleave = pl.interleave ? 1 : 0; d.share = share;
  d.w3_inv_scale = pl.d_w3_inv_scale; d.ovf = tls_gemm_ovf_dev;
  d.W3I = images_on ? pl.d_W3I : nullptr;
  d.write_f32 = 1;
  if (images_on && out_buf >= 0 && (*imgs)[out_buf].base) {
    d.out_img = (*imgs)[out_buf];
    d.write_f32 = buf_f32_[out_buf] ? 1 : 0;
  }
  d.exclusive = (ctx_.size() > 1 && std::getenv("RS_GEMM_B3_EXCLUSIVE")) ? 1 : 0;
  d.nstages = (int)op.stages.size();
  for (int i = 0; i < d.nstages; i++) {
    const EltStage &st = op.stages[i];
    d.stages[i].kind = st.kind == EltStage::kRelu ? 0 : (st.kind == EltStage::kScaleOffset ? 1 : 4);
    d.stages[i].scale = pl.d_stage[i].first; d.stages[i].offset = pl.d_stage[i].second; d.stages[i].alpha = st.alpha;
  }
  d.out = out; d.ldo = ldo;
  if (op.res_buf >= 0) {
    d.res = src[op.res_buf]; d.res_ld = src_ld[op.res_buf]; d.res_scale = op.res_scale;
    if (images_on && pl.res_by_image && d.W3I) d.res_img = (*imgs)[op.res_buf];
  }
  return d;
}

// The acoustic model's ops [op_begin, op_end) on one set of frame buffers (kernels.h row layout).  A layer is evaluated on the
// rows its consumers read -- t in [-lext, T + rext) of every utterance, the real frames only for the last layers -- when the
// caller supplies that row list (row_maps), else on all rows.
void Model::RunNnet(const std::vector<float *> &bufp, const std::vector<int> &buf_ld, float *d_ivec, int ld_i, const int *d_row_ivec, int rows,
                    const RowMaps &row_maps, int share, size_t op_begin, size_t op_end, hipStream_t s,
                    const std::vector<ActImage> *imgs) const {
  const Nnet &nn = am_.nnet;
  const bool images_on = imgs != nullptr && GemmImagesEnabled() && !tls_exact_gemm && tls_gemm_ovf_dev != nullptr;
  if (op_begin == 0 && !tls_exact_gemm) ZeroGuards(bufp, buf_ld, rows, images_on ? imgs : nullptr, s);
  for (size_t i = op_begin; i < op_end; i++) {
    const LayerOp &op = nn.ops[i];
    const bool img_out = images_on && (*imgs)[op.out_buf].base != nullptr;
    bool img_done = false;
    const int *conv_map = nullptr;
    int conv_rows = rows;
    if (op.kind == LayerOp::kGemm) {
      GemmDev gd = MakeGemm(gemm_plans_[i], bufp, buf_ld, d_ivec, ld_i, bufp[op.out_buf], buf_ld[op.out_buf], share, imgs, (int)op.out_buf);
      if (img_out && !GemmWritesImage(gd)) gd.write_f32 = 1;      // a kernel without the image epilogue: converted below
      else img_done = img_out;
      const BufferInfo &ob = nn.bufs[op.out_buf];
      if (const RowMaps::Entry *rm = row_maps.Find(ob.lext, ob.rext, ob.stride)) {     // only the rows somebody reads
        gd.row_map = rm->rows;
        gd.row_map_span128 = rm->span128;
        gd.row_map_span160 = rm->span160;
        LaunchGemm(gd, rm->count, d_row_ivec, s);
        conv_map = rm->rows; conv_rows = rm->count;
      } else {
        LaunchGemm(gd, rows, d_row_ivec, s);
      }
    } else {
      EltwiseDev d;
      std::memset(&d, 0, sizeof(d));
      d.nterms = (int)op.terms.size();
      for (int k = 0; k < d.nterms; k++) {
        d.terms[k].src = bufp[op.terms[k].src_buf]; d.terms[k].ld = buf_ld[op.terms[k].src_buf];
        d.terms[k].col0 = op.terms[k].src_col; d.terms[k].row_off = op.terms[k].offset; d.terms[k].scale = op.terms[k].scale;
      }
      d.dim = op.out_dim; d.out = bufp[op.out_buf]; d.ldo = buf_ld[op.out_buf];
      int ns = 0;
      for (auto &st : op.stages) {
        if (st.kind == EltStage::kLogSoftmax || st.kind == EltStage::kNormalize) {
          if (op.stages.size() != 1 || op.terms.size() != 1 || op.terms[0].scale != 1.0f)
            Fail("nnet3: row-wise component in a fused position is not supported: " + op.name);
          d.row_reduce = st.kind == EltStage::kLogSoftmax ? 2 : 3;
          d.alpha = st.alpha;
        } else {
          d.stages[ns].kind = st.kind == EltStage::kRelu ? 0 : (st.kind == EltStage::kScaleOffset ? 1 : 4);
          d.stages[ns].scale = gemm_plans_[i].d_stage[&st - &op.stages[0]].first;
          d.stages[ns].offset = gemm_plans_[i].d_stage[&st - &op.stages[0]].second;
          d.stages[ns].alpha = st.alpha;
          ns++;
        }
      }
      d.nstages = ns;
      // (the rows somebody reads, like a layer GEMM: its operands hold nothing else when THEY were evaluated through that list)
      const BufferInfo &ob = nn.bufs[op.out_buf];
      if (const RowMaps::Entry *rm = row_maps.Find(ob.lext, ob.rext, ob.stride)) {
        d.row_map = rm->rows;
        LaunchEltwise(d, rm->count, s);
        conv_map = rm->rows; conv_rows = rm->count;
      } else {
        LaunchEltwise(d, rows, s);
      }
    }
    // a producer without the image epilogue: its rows -- the ones it wrote, no others -- converted (and range-checked) here
    if (img_out && !img_done) LaunchToImage(bufp[op.out_buf], buf_ld[op.out_buf], nn.bufs[op.out_buf].dim, conv_rows, (*imgs)[op.out_buf], tls_gemm_ovf_dev, s, conv_map);
  }
  if (op_end == nn.ops.size() && (d_log_priors_ || opts_.acoustic_scale != 1.0f))
    LaunchPriorScale(bufp[nn.output_buf], buf_ld[nn.output_buf], rows, nn.output_dim, d_log_priors_, opts_.acoustic_scale, s);
}

// ------------------------------------------------------------------------------------------------ online iVector estimator, chunk chain
static int IvecChainGroup(int n, int K) { return std::max(1, std::min(K, 512 / std::max(n, 1))); }      // chunks whose statistics are computed side by side
size_t Model::IvecChunkChainBytes(const IvecDev &iv, int n, int K) {
  const size_t P = (size_t)IvecChainGroup(n, K) * n, Di = iv.ivec_dim, usz = Di * (Di + 1) / 2;
  return P * ((size_t)iv.num_gauss * 8 + (size_t)iv.num_gauss * iv.feat_dim * 8 + (Di + usz + 1) * 8) + IvecStatsScratchDoubles(iv, (int)P) * 8 + 4096;
}
void Model::IvecChunkChain(DeviceArena &arena, const BatchGeom &g, int n, int K, const float *stats_feats, int ld_l, const int *post_idx, const float *post_w,
                           const int *d_fb, const int *d_fe, const int *d_or, const int *d_ac, double *lin, double *quad, double *numf, double *x,
                           const int *slot, float *ivec_out, int ld_i, hipStream_t s) const {
  if (n <= 0 || K <= 0) return;
  const IvecDev &iv = ivec_dev_;
  const int KB = IvecChainGroup(n, K), Di = iv.ivec_dim, usz = Di * (Di + 1) / 2;
  const size_t Pmax = (size_t)KB * n;
  double *gamma = arena.AllocT<double>(Pmax * iv.num_gauss), *wfeats = arena.AllocT<double>(Pmax * iv.num_gauss * iv.feat_dim);
  double *dblock = arena.AllocT<double>(Pmax * (size_t)(Di + usz + 1));
  double *scratch = arena.AllocT<double>(IvecStatsScratchDoubles(iv, (int)Pmax));
  IvecDev iv0 = iv;
  iv0.max_count = 0.f;            // the increments carry no prior rescaling: the chain kernel applies it to the running state
  for (int k0 = 0; k0 < K; k0 += KB) {
    const int kb = std::min(KB, K - k0), P = kb * n;
    const size_t o = (size_t)k0 * n;
    BatchGeom g2 = g;
    g2.n_utts = P;
    // (every pseudo-utterance starts its sums itself -- a chunk's statistics are its own frames' -- so nothing clears gamma / wfeats)
    LaunchIvecAccumulate(iv, g2, stats_feats, ld_l, post_idx, post_w, d_fb + o, d_fe + o, gamma, wfeats, true, s, n);
    double *dlin = dblock, *dquad = dlin + (size_t)P * Di, *dtot = dquad + (size_t)P * usz;
    RS_HIP(hipMemsetAsync(dblock, 0, sizeof(double) * (size_t)P * (Di + usz + 1), s));
    LaunchIvecStats(iv0, P, gamma, wfeats, dlin, dquad, dtot, scratch, s);
    if (!LaunchIvecChain(iv, n, kb, dlin, dquad, dtot, lin, quad, numf, x, slot, ivec_out, ld_i, d_or + o, d_ac + o, s))
      Fail("internal error: iVector dimension beyond the chain kernel (callers check)");
  }
}

// ------------------------------------------------------------------------------------------------ search
// Which search kernel a call runs and the work buffers it needs (all from the call's arena).  Shared by the batch path
// (DecodeGroup) and the end of a stream (stream.cc), whose log-likelihoods live in the stream pool.
size_t Model::PlanSearch(int n_utts, int maxT, int nbest, float lat_scale, SearchPlan *sp) const {
  const int S = hclg_.num_states();
  sp->S = S; sp->n_utts = n_utts; sp->maxT = maxT; sp->max_words = 1024;
  // The reference un-scales the lattice's acoustic costs before lattice-to-nbest ranks its paths (online2-wav-nnet3-latgen-
  // faster.cc:290-293), so with a decodable --acoustic-scale other than 1 even the 1-best is chosen on the lattice.
  sp->unscale = opts_.acoustic_scale != 1.0f && opts_.acoustic_scale != 0.0f;
  sp->want_lattice = (nbest > 1 || lat_scale != 1.0f || opts_.emit_lattice != 0 || sp->unscale);
  sp->use_reg = reg_dev_.nt != 0 && !sp->want_lattice && !force_sparse_ && (decoder_choice_ == 0 || decoder_choice_ == 1);
  sp->use_dense = dense_ok_ && !sp->want_lattice && !force_sparse_ && decoder_choice_ != 3;
  // A lattice needs every token of every frame, which the token-list searches keep and the register-resident one does not (5.8 ms
  // against 1.1 for the headline batch): it leaves the costs of all (frame, state) pairs beside its back-pointer rows instead and a
  // compaction kernel writes the token lists LatticeKernel reads (RS_LATTICE_SEARCH=tokens: the token-list search, as before round 4)
  {
    const char *e = std::getenv("RS_LATTICE_SEARCH");          // (read per call: a test compares the two)
    sp->reg_lattice = sp->want_lattice && reg_dev_.nt != 0 && !force_sparse_ && (decoder_choice_ == 0 || decoder_choice_ == 1) &&
                      !(ExactOrder() && reg_dev_.exact_ok) && !(e && std::string(e) == "tokens");
    if (sp->reg_lattice) { sp->use_reg = true; sp->use_dense = true; }
  }
  int cap_pf = opts_.max_tokens_per_frame > 0 ? opts_.max_tokens_per_frame : (int)std::min<long long>(std::max(4ll * opts_.max_active, 8192ll), 0x7fffffffll);
  cap_pf = std::min(cap_pf, S);
  // (the register-resident search behind an n-best / lattice call keeps every live state of every frame -- it has no per-frame token
  // limit -- and DenseToTokensKernel writes them all: the utterance's slice of the token array holds S per frame whatever
  // max_tokens_per_frame says)
  if (sp->reg_lattice) cap_pf = S;
  const long tok_cap_l = (long)(maxT + 2) * cap_pf;
  if (tok_cap_l > 0x7fffffffL) Fail("decoder token capacity overflows; lower max_tokens_per_frame");
  sp->tok_cap = (int)tok_cap_l;
  sp->dopts.beam = opts_.beam; sp->dopts.lattice_beam = opts_.lattice_beam; sp->dopts.beam_delta = opts_.beam_delta;
  sp->dopts.max_active = opts_.max_active; sp->dopts.min_active = opts_.min_active;
  sp->dopts.exact_order = ExactOrder() ? 1 : 0;
  size_t need = (size_t)n_utts * ((size_t)(maxT + 1) * 16 + (size_t)sp->max_words * 4 + 4 + 16 + 64) + 65536;      // results, counters, frame info
  if (sp->use_dense)      // dense / register-resident search: back-pointer rows, path scratch, parked token costs
    need += (size_t)n_utts * ((size_t)(maxT + 1) * S * 4 + (size_t)(maxT + 2) * 32 + (2 * (size_t)S + 4) * 4) + 4096;
  if (sp->reg_lattice)    // ... the cost rows, the token lists made of them, LatticeKernel's two maps
    need += (size_t)n_utts * ((size_t)(maxT + 1) * S * 4 + (size_t)sp->tok_cap * 16 + (size_t)(maxT + 2) * 4 + (size_t)S * 8) + 8192;
  if (!sp->use_dense) {                  // token-list search: per-state tables, queues, the token arrays of every frame
    need += (size_t)n_utts * ((size_t)S * (8 + 4 * 5) + (size_t)sp->tok_cap * 16 + (size_t)(maxT + 2) * 4) + 8192;
    sp->use_hash = decoder_choice_ != 3 && DecodeLiveUsable(hclg_dev_);
    if (sp->use_hash) need += (size_t)n_utts * ((size_t)DecodeLiveTableSize() * (8 + 4) + (size_t)DecodeLiveGlobalTable() * 4 + (size_t)DecodeLiveSlotCap() * 16 +
                                                (size_t)kHashCandCap * 8 + (size_t)kLiveQueueCap * 2 * 20 + 4) + 16384;
  }
  if (sp->want_lattice) need += sizeof(float) * (size_t)n_utts * sp->tok_cap + 4096;                                // LatticeKernel's extra_cost
  return need;
}

void Model::AllocSearch(SearchPlan *sp, DeviceArena &arena_, hipStream_t s, bool pooled_frames) const {
  const int n_utts = sp->n_utts, maxT = sp->maxT, S = sp->S, max_words = sp->max_words;
  DecodeWork &w = sp->w;
  std::memset(&w, 0, sizeof(w));
  w.max_words = max_words;
  w.out_words = arena_.AllocT<int>((size_t)n_utts * max_words);
  w.out_nwords = arena_.AllocT<int>(n_utts);
  w.out_costs = arena_.AllocT<float>((size_t)n_utts * 4);
  w.counters = arena_.AllocT<long long>((size_t)n_utts * 8);
  if (!pooled_frames) w.frame_info = arena_.AllocT<float>((size_t)n_utts * (maxT + 1) * 4);
  DenseWork &dw = sp->dw;
  std::memset(&dw, 0, sizeof(dw));
  if (sp->use_dense) {
    dw.out_words = w.out_words; dw.out_nwords = w.out_nwords; dw.out_costs = w.out_costs; dw.counters = w.counters;
    dw.frame_info = w.frame_info; dw.max_words = max_words;
    dw.path_cap = 4 * (maxT + 2);
    dw.path = arena_.AllocT<int>((size_t)n_utts * dw.path_cap * 2);
    if (!pooled_frames) {      // (streams keep back-pointer rows, frame info, parked costs and counters in their pool)
      dw.bp = arena_.AllocT<int>((size_t)n_utts * (maxT + 1) * S);
      dw.state_cost = arena_.AllocT<float>((size_t)n_utts * (2 * (size_t)S + 4));
      RS_HIP(hipMemsetAsync(w.counters, 0, sizeof(long long) * 8 * (size_t)n_utts, s));
    }
    if (sp->reg_lattice && !pooled_frames) {
      dw.cost_rows = arena_.AllocT<float>((size_t)n_utts * (maxT + 1) * S);
      w.tok_cap = sp->tok_cap;
      w.tokens = arena_.AllocT<int4>((size_t)n_utts * sp->tok_cap);
      w.frame_tok_off = arena_.AllocT<int>((size_t)n_utts * (maxT + 2));
      w.map_a = arena_.AllocT<int>((size_t)n_utts * S);
      w.map_b = arena_.AllocT<int>((size_t)n_utts * S);
      RS_HIP(hipMemsetAsync(w.map_a, 0xFF, sizeof(int) * (size_t)n_utts * S, s));      // LatticeKernel expects both state -> token maps empty
      RS_HIP(hipMemsetAsync(w.map_b, 0xFF, sizeof(int) * (size_t)n_utts * S, s));
    }
  }
}

void Model::LaunchSearch(SearchPlan *sp, DeviceArena &arena_, const BatchGeom &g, const float *ll, int ll_ld, hipStream_t s) const {
  const int n_utts = sp->n_utts, maxT = sp->maxT, S = sp->S;
  if (sp->use_dense) {
    if (sp->use_reg) LaunchDecodeReg(hclg_dev_, reg_dev_, sp->dopts, g, ll, ll_ld, sp->dw, -1, maxT + 1, s);
    else LaunchDecodeDense(hclg_dev_, rev_dev_, sp->dopts, g, ll, ll_ld, am_.nnet.output_dim, sp->dw, s);
    if (sp->reg_lattice) LaunchDenseToTokens(hclg_dev_, g, sp->dw, sp->w, s, /*write_tokens=*/!DenseLatticeUsable(hclg_dev_));
    return;
  }
  DecodeWork &w = sp->w;
  w.best = arena_.AllocT<unsigned long long>((size_t)n_utts * S);
  w.map_a = arena_.AllocT<int>((size_t)n_utts * S);
  w.map_b = arena_.AllocT<int>((size_t)n_utts * S);
  w.queue_a = arena_.AllocT<int>((size_t)n_utts * S);
  w.queue_b = arena_.AllocT<int>((size_t)n_utts * S);
  w.in_queue = arena_.AllocT<int>((size_t)n_utts * S);
  w.tok_cap = sp->tok_cap;
  w.tokens = arena_.AllocT<int4>((size_t)n_utts * sp->tok_cap);
  w.frame_tok_off = arena_.AllocT<int>((size_t)n_utts * (maxT + 2));
  if (sp->use_hash) {
    // live states of a frame in a two-level table (LDS, then global memory: decode_live.hip); the dense tables above are only touched
    // for utterances that outgrow it (w.redo)
    const size_t cap = (size_t)DecodeLiveSlotCap(), tab = (size_t)DecodeLiveTableSize();
    w.h_tab = (int)tab;
    w.h_keys = arena_.AllocT<unsigned long long>((size_t)n_utts * tab);
    w.h_slot_tok = arena_.AllocT<int>((size_t)n_utts * tab);
    w.h_cand_cap = kHashCandCap;
    { const char *e = std::getenv("RS_HASH_SLOT_LIMIT"); w.h_slot_limit = e ? std::atoi(e) : DecodeLiveSlotCap(); }      // (tests)
    w.h_cand = arena_.AllocT<int>((size_t)n_utts * 2 * kHashCandCap);
    w.h_gtags = arena_.AllocT<unsigned>((size_t)n_utts * DecodeLiveGlobalTable());
    w.h_qcap = kLiveQueueCap;
    w.h_q4 = arena_.AllocT<int4>((size_t)n_utts * 2 * kLiveQueueCap);
    w.h_qne = arena_.AllocT<int>((size_t)n_utts * 2 * kLiveQueueCap);
    { const char *e = std::getenv("RS_HASH_LDS_LOG"); w.h_lds_log = e ? std::atoi(e) : 0; }                                 // (tests)
    w.h_comp = arena_.AllocT<int4>((size_t)n_utts * cap);
    w.redo = arena_.AllocT<int>(n_utts);
    LaunchDecodeLive(hclg_dev_, sp->dopts, g, ll, ll_ld, w, s);
    if (sp->want_lattice) {      // LatticeKernel expects both state -> token maps empty (DecodeKernel leaves them so for the utterances it decodes)
      RS_HIP(hipMemsetAsync(w.map_a, 0xFF, sizeof(int) * (size_t)n_utts * S, s));
      RS_HIP(hipMemsetAsync(w.map_b, 0xFF, sizeof(int) * (size_t)n_utts * S, s));
    }
  }
  LaunchDecode(hclg_dev_, sp->dopts, g, ll, ll_ld, w, s);
}

// Result records to the host, and -- when the call asked for more than the traceback gives -- the reference's
// determinise | lattice-to-nbest | nbest-to-linear tail on the lattice the token-list search left behind.
void Model::CollectResults(SearchPlan &sp, DecodeContext &cx, int gi, const BatchGeom &g, const int *T, const float *ll, int ll_ld, int nbest,
                           float lat_scale, hipStream_t s, UttResult *out_utts, float *timings) {
  DeviceArena &arena_ = cx.arena[gi];
  HostArena &harena = cx.host_arena[gi];
  const int n_utts = sp.n_utts, max_words = sp.max_words, tok_cap = sp.tok_cap;
  const bool want_lattice = sp.want_lattice, unscale = sp.unscale;
  const DecodeOptsDev &dopts = sp.dopts;
  DecodeWork &w = sp.w;
  // ---- results to host
  // page-locked destination; only the first kWordsInline word ids of every utterance travel with the first copy (longer
  // transcripts fetch their row afterwards)
  constexpr int kWordsInline = 48;
  int *h_nw = harena.AllocT<int>(n_utts), *h_words = harena.AllocT<int>((size_t)n_utts * kWordsInline);
  float *h_costs = harena.AllocT<float>((size_t)n_utts * 4);
  long long *h_ctr = harena.AllocT<long long>((size_t)n_utts * 8);
  LaunchResultsToHost(w.out_nwords, w.out_costs, w.counters, w.out_words, max_words, std::min(kWordsInline, max_words), kWordsInline, n_utts, h_nw, h_costs, h_ctr, h_words, s);
  RS_HIP(hipStreamSynchronize(s));
  RS_HIP(hipGetLastError());
  if (CheckGemmRange(cx)) throw RangeRetry{};      // an activation beyond the fp16 split's range: the call is repeated on the exact-FP32 GEMMs
  for (int u = 0; u < n_utts; u++) {
    UttResult &ur = out_utts[u];
    for (int k = 0; k < 8; k++) ur.counters[k] = h_ctr[(size_t)u * 8 + k];
    if (T[u] == 0) {
      ur.status = RS_ERR_DECODE;
      ur.error = "You cannot get a lattice if you decoded no frames.";   // online-nnet3-decoding.cc:68-69
      continue;
    }
    if (h_nw[u] < 0) {
      ur.status = RS_ERR_DECODE;
      long long flags = ur.counters[7];
      if (flags & 1) ur.error = "decoder token capacity exceeded (raise rs_decode_opts.max_tokens_per_frame)";
      else if (flags & 4) ur.error = "epsilon cycle in the decoding graph";
      else if (flags & 2) ur.error = "no surviving tokens (search error)";
      else ur.error = "best path has more than " + std::to_string(max_words) + " words";
      continue;
    }
    if (ur.counters[7] & 1) {
      ur.status = RS_ERR_DECODE;
      ur.error = "decoder token capacity exceeded (raise rs_decode_opts.max_tokens_per_frame)";
      continue;
    }
    Hypothesis hy;
    if (h_nw[u] <= kWordsInline) {
      hy.words.assign(h_words + (size_t)u * kWordsInline, h_words + (size_t)u * kWordsInline + h_nw[u]);
    } else {
      hy.words.resize(h_nw[u]);
      RS_HIP(hipMemcpy(hy.words.data(), w.out_words + (size_t)u * max_words, sizeof(int) * h_nw[u], hipMemcpyDeviceToHost));
    }
    hy.graph_cost = h_costs[(size_t)u * 4 + 0];
    hy.acoustic_cost = h_costs[(size_t)u * 4 + 1];
    ur.hyps.push_back(std::move(hy));
  }
  // ---- n-best through the lattice (the reference's determinise | lattice-to-nbest | nbest-to-linear tail)
  if (want_lattice) {
    auto t_l0 = std::chrono::steady_clock::now();
    LatticeWork lw;
    std::memset(&lw, 0, sizeof(lw));
    // extra_cost comes from the call's arena; the arc buffer is the context's own, grow-only: no hipMalloc / hipFree (a
    // device-wide synchronisation that would stall the other calls in flight) in the steady state
    lw.extra_cost = arena_.AllocT<float>((size_t)n_utts * tok_cap + 64);
    int *d_count = arena_.AllocT<int>((size_t)n_utts + 4);
    const LatArc *h_arcs = nullptr;          // pinned staging (the call's host arena): a pageable destination copied at 5 GB/s
    size_t n_arcs = 0;
    LatArcBuffer &ab = cx.lat_arcs[gi];
    std::vector<int> ubegin(n_utts + 1, 0);  // the utterances' arcs in h_arcs: every utterance appends to its own region of the buffer
    // RS_LATTICE_TRACE=1: where the tail's time goes (kernel + counts, arcs to the host, -, the per-utterance jobs)
    static const bool lat_trace = [] { const char *e = TuneEnv("RS_LATTICE_TRACE"); return e && std::atoi(e) != 0; }();
    auto lt0 = std::chrono::steady_clock::now();
    float lt_ms[4] = {0, 0, 0, 0};
    auto lt_mark = [&](int i) { const auto n_ = std::chrono::steady_clock::now(); lt_ms[i] += std::chrono::duration<float, std::milli>(n_ - lt0).count(); lt0 = n_; };
    for (int attempt = 0; attempt < 8; attempt++) {
      if (ab.cap < (size_t)n_utts * 256) {
        if (ab.d) RS_HIP(hipFree(ab.d));
        ab.cap = std::max<size_t>(1u << 20, (size_t)n_utts * 1024);
        RS_HIP(hipMalloc((void **)&ab.d, sizeof(LatArc) * ab.cap));
      }
      RS_HIP(hipMemsetAsync(d_count, 0, sizeof(int) * n_utts, s));
      lw.arcs = static_cast<LatArc *>(ab.d); lw.utt_cap = (int)std::min<size_t>(ab.cap / n_utts, 0x7fffffff); lw.arcs_count = d_count;
      if (sp.reg_lattice && DenseLatticeUsable(hclg_dev_)) LaunchDenseLattice(hclg_dev_, dopts, g, ll, ll_ld, sp.dw, w, lw, hclg_has_eps_ ? reg_dev_.eps_depth : 0, s);
      else LaunchLatticePrune(hclg_dev_, dopts, g, ll, ll_ld, w, lw, s);
      // (counts and arcs travel by kernels that store into the pinned block, not by the copy engine: with other calls in flight their
      // 25 MB sample uploads are queued on that engine and these copies waited behind them -- 0.15 ms alone, 5.4 ms with four calls
      // in flight: profiles/micro/nbest_trace.sh)
      int *h_count = harena.AllocT<int>(n_utts);
      LaunchCopyRows(d_count, n_utts, nullptr, h_count, n_utts, nullptr, 1, n_utts, s);
      RS_HIP(hipStreamSynchronize(s));
      int most = 0;
      for (int u = 0; u < n_utts; u++) most = std::max(most, h_count[u]);
      lt_mark(0);
      if (most <= lw.utt_cap) {
        for (int u = 0; u < n_utts; u++) ubegin[u + 1] = ubegin[u] + h_count[u];
        n_arcs = (size_t)ubegin[n_utts];
        LatArc *dst = harena.AllocT<LatArc>(n_arcs + 1);
        if (n_arcs) LaunchCompactArcs(lw.arcs, lw.utt_cap, d_count, n_utts, dst, s);
        RS_HIP(hipStreamSynchronize(s));
        h_arcs = dst;
        lt_mark(1);
        break;
      }
      if (attempt == 7) Fail("lattice extraction: arc buffer overflow");
      RS_HIP(hipFree(ab.d));                     // rare: an utterance's lattice outgrew its region
      ab.d = nullptr;
      ab.cap = ((size_t)most + (size_t)most / 4 + 256) * (size_t)n_utts;
      RS_HIP(hipMalloc((void **)&ab.d, sizeof(LatArc) * ab.cap));
    }
    lt_mark(2);
    auto one = [&](int u) {
      UttResult &ur = out_utts[u];
      if (ur.status != RS_OK) return;
      RawLattice lat;
      // token index -> lattice state in order of first appearance (a per-thread table with stamps: a hash map per utterance was a
      // fifth of the tail's host time)
      thread_local std::vector<int> id_of, id_stamp;
      thread_local int id_epoch = 0;
      int max_tok = 0;
      for (int k = ubegin[u]; k < ubegin[u + 1]; k++) { const LatArc *a = h_arcs + k; max_tok = std::max(max_tok, std::max(a->src, a->arc >= 0 ? a->dst : 0)); }
      if ((int)id_of.size() <= max_tok) { id_of.resize((size_t)max_tok + 1); id_stamp.resize((size_t)max_tok + 1, 0); }
      if (++id_epoch == 0x7fffffff) { std::fill(id_stamp.begin(), id_stamp.end(), 0); id_epoch = 1; }
      int n_ids = 0;
      auto sid = [&](int tok) {
        if (id_stamp[tok] != id_epoch) { id_stamp[tok] = id_epoch; id_of[tok] = n_ids++; }
        return id_of[tok];
      };
      lat.start = sid(0);   // the start token is the first token of frame 0
      for (int k = ubegin[u]; k < ubegin[u + 1]; k++) { const LatArc *a = h_arcs + k; sid(a->src); if (a->arc >= 0) sid(a->dst); }
      lat.num_states = n_ids;
      lat.final_cost.assign(lat.num_states, std::numeric_limits<double>::infinity());
      lat.arcs.reserve((size_t)(ubegin[u + 1] - ubegin[u]));
      for (int k = ubegin[u]; k < ubegin[u + 1]; k++) {
        const LatArc *a = h_arcs + k;
        if (a->arc < 0) { lat.final_cost[id_of[a->src]] = a->graph; continue; }
        lat.arcs.push_back({id_of[a->src], id_of[a->dst], hclg_.arcs[a->arc].olabel, (double)a->graph, (double)a->acoustic, hclg_.arcs[a->arc].ilabel});
      }
      std::vector<NbestPath> paths = LatticeNbest(lat, nbest, opts_.lattice_beam, unscale ? lat_scale / opts_.acoustic_scale : lat_scale);
      ur.counters[4] = (int64_t)lat.arcs.size();
      if (opts_.emit_lattice) {
        ur.clat = std::make_shared<CompactLat>(DeterminizeLattice(lat, opts_.lattice_beam));
        if (unscale) {
          const double inv = 1.0 / opts_.acoustic_scale;
          for (auto &v : ur.clat->arcs) for (auto &a : v) a.w.acoustic *= inv;
          for (auto &fw : ur.clat->final_w) fw.acoustic *= inv;
        }
      }
      if (paths.empty()) return;     // keep the traceback result (cannot happen for a consistent lattice)
      ur.hyps.clear();
      for (auto &p : paths) {
        Hypothesis hy;
        hy.words = p.words;
        hy.graph_cost = (float)p.graph_cost;
        hy.acoustic_cost = (float)(unscale ? p.acoustic_cost / opts_.acoustic_scale : p.acoustic_cost);
        ur.hyps.push_back(std::move(hy));
      }
    };
    // The reference runs one determinise | lattice-to-nbest process per utterance (tools.py:117-147); here the utterances of a call
    // are independent jobs for a few host threads (lat/determinize-lattice-pruned.cc:1488-1513 and latbin/lattice-to-nbest.cc:80-110
    // per job, host code in the reference too).  The threads inherit the caller's CPU mask (rs_bind_host_thread).
    static const int max_threads = [] {
      const char *e = std::getenv("RS_LATTICE_THREADS");
      const int hw = (int)std::thread::hardware_concurrency();
      return std::max(1, e ? std::atoi(e) : std::min(hw > 0 ? hw : 1, 16));
    }();
    const int nthr = std::min(max_threads, n_utts);
    if (nthr <= 1) {
      for (int u = 0; u < n_utts; u++) one(u);
    } else {
      std::atomic<int> next{0};
      std::vector<std::exception_ptr> errs(nthr);
      auto run = [&](int k) {
        try {
          for (int u = next.fetch_add(1); u < n_utts; u = next.fetch_add(1)) one(u);
        } catch (...) { errs[k] = std::current_exception(); }
      };
      std::vector<std::thread> th;
      for (int k = 1; k < nthr; k++) th.emplace_back(run, k);
      run(0);
      for (auto &t : th) t.join();
      for (auto &e : errs) if (e) std::rethrow_exception(e);
    }
    lt_mark(3);
    if (lat_trace)
      std::fprintf(stderr, "lattice tail: kernel + count %.2f ms, %zu arcs (%.1f MB) to the host %.2f ms, grouping %.2f ms, %d utterances on %d threads %.2f ms\n",
                   lt_ms[0], n_arcs, n_arcs * sizeof(LatArc) / 1e6, lt_ms[1], lt_ms[2], n_utts, nthr, lt_ms[3]);
    timings[7] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_l0).count();
  }}

// One group of utterances, start to finish, on one stream with one arena.  DecodeBatchDevice runs two groups
// concurrently (two host threads, two streams) so that the latency-bound stages of one group (search, iVector)
// overlap the MFMA-bound stage (TDNN) of the other.
void Model::DecodeGroup(DecodeContext &cx, int gi, const int16_t *d_pcm, const int64_t *sample_offsets, int n_utts, int nbest, float lat_scale,
                        hipStream_t s, bool streaming, UttResult *out_utts, float *timings) {
  DeviceArena &arena_ = cx.arena[gi];
  HostArena &harena = cx.host_arena[gi];
  RS_HIP(hipSetDevice(opts_.device_id));
  auto wall0 = std::chrono::steady_clock::now();
  if (n_utts == 0) return;
  SampleGemmMode(exact_gemm_.load() || cx.force_exact, cx.gemm_ovf_dev);
  const Nnet &nn = am_.nnet;
  const int C = fc_.mfcc.nceps, P = nn.output_dim;
  // ---- geometry
  std::vector<int> T(n_utts), row_base(n_utts + 1, 0), frame_base(n_utts + 1, 0);
  int maxT = 0;
  for (int u = 0; u < n_utts; u++) {
    long ns = (long)(sample_offsets[u + 1] - sample_offsets[u]);
    if (ns < 0) Fail("sample offsets must be non-decreasing");
    T[u] = NumFrames(ns, fc_.mfcc.opts);
    maxT = std::max(maxT, T[u]);
    row_base[u + 1] = row_base[u] + T[u] + L_ + R_;
    frame_base[u + 1] = frame_base[u] + T[u];
  }
  // --frame-subsampling-factor f: the decoder's frames are the output rows t = 0, f, 2 f, ... (decodable-online-looped.cc:56-84:
  // (T + f - 1) / f of them once the input is finished); num_frames is what the reference's binaries log as decoded frames
  const int fsf = opts_.frame_subsampling_factor;
  std::vector<int> T_dec(n_utts), dec_base(n_utts + 1, 0);
  int maxT_dec = 0;
  for (int u = 0; u < n_utts; u++) {
    T_dec[u] = (T[u] + fsf - 1) / fsf;
    maxT_dec = std::max(maxT_dec, T_dec[u]);
    dec_base[u + 1] = dec_base[u] + T_dec[u];
    out_utts[u].num_frames = T_dec[u];
  }
  const int rows = row_base[n_utts];
  const int guard = L_ + R_ + 8;
  // ---- iVector schedule.  Offline (--online=false): one estimate per utterance from all frames.  Streaming:
  // one estimate per nnet chunk, from the frames available at the 1024-sample tick on which
  // DecodableNnetLoopedOnlineBase::AdvanceChunk runs for that chunk (decodable-online-looped.cc:56-84,186-194;
  // online2-cli-nnet3-decode-faster.cc:143-161).  The schedule only depends on sample counts, so the streaming
  // result is reproduced exactly without replaying wall-clock time.
  const bool has_iv = fc_.ie.present;
  const int chunk = opts_.frames_per_chunk;
  std::vector<int> ivrow_base(n_utts + 1, 0);               // first iVector row of each utterance
  std::vector<std::vector<int>> chunk_last(n_utts);          // streaming: last stats frame of every chunk
  int max_chunks = 1;
  for (int u = 0; u < n_utts; u++) {
    int nrows_u = 1;
    if (streaming && has_iv) {
      const long ns = (long)(sample_offsets[u + 1] - sample_offsets[u]);
      const int nch = (T[u] + chunk - 1) / chunk;
      const int Rm = nn.right_context, sr = fc_.ie.splice_right;
      const long nt = (ns + 1023) / 1024;
      int k = 0;
      for (long j = 0; j < nt && k < nch; j++) {
        const int fr = NumFrames(std::min<long>(1024 * (j + 1), ns), fc_.mfcc.opts);
        const int ready = std::max(0, fr - Rm) / chunk;
        while (k < ready && k < nch) { chunk_last[u].push_back(std::min(fr - 1, fr - sr - 1)); k++; }
      }
      while (k < nch) { chunk_last[u].push_back(T[u] - 1); k++; }
      nrows_u = std::max(nch, 1);
      max_chunks = std::max(max_chunks, nch);
    }
    ivrow_base[u + 1] = ivrow_base[u] + nrows_u;
  }
  const int n_ivrows = ivrow_base[n_utts];
  // which iVector row every frame row reads: the chunk that supplied its Round(ivector, chunk) slot
  // (nnet-compile-looped.cc:164-231: chunk 0 supplies the slots of t in [-L, chunk + R), chunk k the new ones of
  //  [k*chunk + R, (k+1)*chunk + R))
  // (offline: every row of an utterance reads its single iVector row -- filled in on the device with the row geometry)
  harena.Reset();
  const bool host_row_ivec = streaming && has_iv;
  int *row_ivec = host_row_ivec ? harena.AllocT<int>(rows) : nullptr;
  for (int u = 0; host_row_ivec && u < n_utts; u++) {
    const int nch = (int)chunk_last[u].size();
    for (int r = row_base[u]; r < row_base[u + 1]; r++) {
      int k = 0;
      if (streaming && has_iv && nch > 0) {
        const int t = r - row_base[u] - L_;
        const int slot = (t >= 0 ? t / chunk : -((-t + chunk - 1) / chunk)) * chunk;
        // smallest k whose input range [.., (k+1)*chunk + R) contains a time with this slot: slot < (k+1)*chunk + R
        k = 0;
        while (k < nch - 1 && slot >= (k + 1) * chunk + nn.right_context) k++;
      }
      row_ivec[r] = ivrow_base[u] + k;
    }
  }
  // ---- arena sizing
  auto fbytes = [&](int ld) { return ((size_t)rows + 2 * guard) * ld * sizeof(float) + 512; };
  size_t need = 0;
  need += (sizeof(int64_t) + 4 * sizeof(int)) * (size_t)(n_utts + 2) + 3 * sizeof(int) * (size_t)rows + 4096;
  need += sizeof(int) * ((size_t)frame_base[n_utts] + 8 * (size_t)n_utts + 64) + 1024;     // frame-row map
  std::vector<int> buf_ld(nn.bufs.size());
  for (size_t b = 0; b < nn.bufs.size(); b++) { buf_ld[b] = RoundUp(nn.bufs[b].dim, 4); need += fbytes(buf_ld[b]); }
  need += ImageBytes(rows);
  need += (size_t)nn.ops.size() * (sizeof(int) * ((size_t)rows + n_utts + 64) + 512);      // row lists per output extent
  const int Dl = has_iv ? fc_.ie.feat_dim() : 0, Di = has_iv ? fc_.ie.ivector_dim() : 0, G = has_iv ? fc_.ie.num_gauss() : 0;
  const int ld_c = RoundUp(C, 4), ld_l = RoundUp(std::max(Dl, 1), 4), ld_i = RoundUp(std::max(Di, 1), 4);
  const int usz = Di * (Di + 1) / 2, nsel = has_iv ? fc_.ie.num_gselect : 0;
  if (fc_.use_cmvn) need += fbytes(ld_c);
  if (has_iv) {
    need += fbytes(ld_c) + 2 * fbytes(ld_l);
    need += (size_t)rows * nsel * 8 + 1024;
    need += (size_t)n_utts * ((size_t)G * 8 + (size_t)G * Dl * 8 + (size_t)Di * 8 * 2 + (size_t)usz * 8 + 8) + (size_t)n_ivrows * ld_i * 4 + 8192 + 1024;
    need += (size_t)max_chunks * n_utts * 16 + 4096;
    need += IvecStatsScratchDoubles(ivec_dev_, n_utts) * 8 + 1024;
    if (streaming) need += IvecChunkChainBytes(ivec_dev_, n_utts, max_chunks);
  }
  SearchPlan sp;
  need += PlanSearch(n_utts, maxT_dec, nbest, lat_scale, &sp);
  if (fsf > 1) need += ((size_t)dec_base[n_utts] + 8) * RoundUp(P, 4) * sizeof(float) + (size_t)(dec_base[n_utts] + 3 * n_utts + 16) * sizeof(int) + 4096;
  const bool use_reg = sp.use_reg;
  need += 64 * 256;   // alignment slack
  arena_.Reserve(need + (1u << 20), s);
  arena_.Reset();
  // ---- geometry: ONE page-locked staging block -> one async copy, ONE launch that derives every per-row array on the device (the
  // rows' utterance / frame / iVector row and the row lists of the layers that are evaluated on fewer rows than the full halo).
  // Round 4 issued a copy + a launch per list: 8 + 8 of the ~25 launch boundaries in front of a call's first real kernel.
  int *d_row_ivec = nullptr;
  BatchGeom g;
  g.n_utts = n_utts; g.L = L_; g.R = R_; g.total_rows = rows; g.total_frames = frame_base[n_utts]; g.max_frames = maxT; g.guard = guard;
  const int total_frames = frame_base[n_utts];
#ifdef RS_TUNING
  static const int lds_poison = [] { const char *e = TuneEnv("RS_LDS_POISON"); return e ? std::atoi(e) : 0; }();
  auto poison = [&]() {
    if (!lds_poison) return;
    static unsigned *sink = [] { unsigned *p = nullptr; (void)hipMalloc((void **)&p, 64); return p; }();
    LaunchLdsPoison(sink, s);      // (profiles/micro/poison_kernels.hip)
  };
#else
  auto poison = []() {};
#endif
  // The last layer and the search can be pipelined over time slabs when the search is the register-resident kernel: the
  // output GEMM of slab k+1 (MFMA-bound) runs while slab k is searched (latency-bound) on a second, high-priority stream.
  // Measured on the bench batch: 5.07 -> 4.97 ms with 3 slabs -- the search runs at half speed while it shares the CUs
  // with GEMM waves, so most of the overlap is given back; off by default (RS_OVERLAP_SLABS=n turns it on).
  const int overlap_env = [] { const char *e = std::getenv("RS_OVERLAP_SLABS"); return e ? std::atoi(e) : 1; }();
  const bool last_is_gemm = !nn.ops.empty() && nn.ops.back().kind == LayerOp::kGemm && nn.ops.back().out_buf == nn.output_buf &&
                            nn.bufs[nn.output_buf].lext == 0 && nn.bufs[nn.output_buf].rext == 0;
  const bool pipelined = use_reg && !sp.reg_lattice && last_is_gemm && !d_log_priors_ && opts_.acoustic_scale == 1.0f && overlap_env > 1 && maxT >= 64 &&
                         s == cx.stream && fsf == 1;
  const int n_slabs = pipelined ? std::min(overlap_env, 8) : 1, slab_len = std::max(1, (maxT + n_slabs - 1) / n_slabs);
  std::vector<int> slab_off(n_slabs + 1, 0);
  int *d_frame_rows = nullptr;
  RowMaps row_maps;
  {
    // which row lists this batch needs: the real frames in slab-major order (slab k = frames [k * slab_len, (k + 1) * slab_len) of
    // every utterance: layers nothing downstream reads with a time offset are evaluated on these rows only), and per hidden layer
    // only as much halo as the layers after it reach (15 rows a side for the first, none for the last of the zamia-like net:
    // 5 % fewer rows over the stack than evaluating the full halo everywhere)
    struct ListPlan { int lext, rext, n_segs, total, L_eff, slab_len, span128; size_t seg_at; int stride = 1, first = 0, span160 = 0; };
    std::vector<ListPlan> lists;
    std::vector<int> segs;      // the lists' segment offsets, back to back
    // The physical rows 128 consecutive entries of a stride-1 list reach over, exactly: the list is one run of consecutive rows per
    // utterance with frames (t in [-lext, T + rext): T + lext + rext entries from row row_base + L - lext on); a window of 128 entries
    // that holds the last entry of run a and the first of run b crosses every gap between them, and it can do so when the runs in
    // between hold at most 126 entries.  An utterance without frames has no entries but still owns L + R rows (a too-short clip
    // inside a batch), so the gap between two runs is not bounded by one halo: GemmKernelB3J's strip form trusts this number.
    auto span_of_runs = [&](int lext, int rext, int window = 128) {
      std::vector<std::pair<int, int>> runs;      // (first physical row, entries)
      for (int u = 0; u < n_utts; u++) if (T[u] > 0) runs.emplace_back(row_base[u] + L_ - lext, T[u] + lext + rext);
      int worst = 0;
      size_t b = 0;
      long inner = 0;                             // entries of the runs strictly between a and b
      for (size_t a = 0; a + 1 < runs.size(); a++) {
        if (b <= a) { b = a + 1; inner = 0; }
        while (b + 1 < runs.size() && inner + runs[b].second <= window - 2) { inner += runs[b].second; b++; }
        const long gaps = (long)runs[b].first - (runs[a].first + runs[a].second) - inner;
        worst = std::max<long>(worst, gaps);
        if (b > a + 1) inner -= runs[a + 1].second;
      }
      return window + worst;
    };
    if (total_frames > 0) {
      ListPlan lp{0, 0, n_slabs * n_utts, total_frames, L_, slab_len, 0, segs.size()};
      int acc_rows = 0;
      for (int k = 0; k < n_slabs; k++) {
        slab_off[k] = acc_rows;
        for (int u = 0; u < n_utts; u++) { segs.push_back(acc_rows); acc_rows += std::min(std::max(T[u] - k * slab_len, 0), slab_len); }
      }
      segs.push_back(acc_rows);
      slab_off[n_slabs] = acc_rows;
      // (one slab: the list runs through the utterances in order, so a GEMM tile of 128 rows reaches over its rows + the halos it skips)
      lp.span128 = n_slabs == 1 ? span_of_runs(0, 0) : 0;
      lp.span160 = n_slabs == 1 ? span_of_runs(0, 0, 160) : 0;
      lists.push_back(lp);
      static const int trim_env = [] { const char *e = TuneEnv("RS_TRIM_HALO"); return e ? std::atoi(e) : 1; }();
      // Trimmed halos are all or nothing: an op evaluated through its list leaves the other rows of its buffer as the arena held them, so
      // everything that reads the buffer must go through a list as narrow or narrower.  Count the distinct lists first; a network with
      // more of them than a call carries evaluates every layer on all rows (of the full halo: always valid) instead.
      int trim = trim_env;
      {
        std::vector<std::array<int, 3>> distinct;
        for (auto &op : nn.ops) {
          const BufferInfo &ob = nn.bufs[op.out_buf];
          if (ob.stride == 1 && ((ob.lext == 0 && ob.rext == 0) || (ob.lext >= L_ && ob.rext >= R_) || !trim_env)) continue;
          const std::array<int, 3> key{ob.lext, ob.rext, ob.stride};
          if (std::find(distinct.begin(), distinct.end(), key) == distinct.end()) distinct.push_back(key);
        }
        if ((int)distinct.size() + 1 > BatchSetup::kMaxLists) trim = 0;
      }
      // (two passes: the lists of the strided buffers first -- a buffer evaluated on every f-th row MUST have its list, its consumers
      // read nothing else and its own sources may hold nothing else -- then, while there is room, the trimmed halos, which only save work)
      for (size_t pi = 0; pi < 2 * nn.ops.size(); pi++) {
        const size_t i = pi % nn.ops.size();
        const bool strided_pass = pi < nn.ops.size();
        const BufferInfo &ob = nn.bufs[nn.ops[i].out_buf];
        const int st = ob.stride;
        if ((st > 1) != strided_pass || (st == 1 && !trim)) continue;
        if (ob.lext > L_ || ob.rext > R_) continue;
        if (st == 1 && ((ob.lext == 0 && ob.rext == 0) || (ob.lext >= L_ && ob.rext >= R_))) continue;
        if (st > 1 && n_slabs != 1) Fail("internal error: strided layers in a slab-pipelined call");
        bool have = false;
        for (auto &l : lists) have = have || (l.lext == ob.lext && l.rext == ob.rext && l.stride == st);
        if (have) continue;
        if ((int)lists.size() >= BatchSetup::kMaxLists) {
          if (st > 1) Fail("nnet3: more distinct (context, stride) row lists than a call carries (" + std::to_string(BatchSetup::kMaxLists) + ") with --frame-subsampling-factor");
          continue;
        }
        ListPlan l2{ob.lext, ob.rext, n_utts, 0, L_ - ob.lext, std::max(maxT + ob.lext + ob.rext, 1), 0, segs.size()};
        l2.stride = st;
        l2.first = ob.lext % st;             // t = -lext + first is the first row with t = 0 mod stride
        int acc = 0;
        // (rows t = 0 mod stride of [-lext, T + rext): (T + rext - 1) / stride + lext / stride + 1 of them)
        for (int u = 0; u < n_utts; u++) { segs.push_back(acc); acc += T[u] > 0 ? (st == 1 ? T[u] + ob.lext + ob.rext : (T[u] + ob.rext - 1) / st + ob.lext / st + 1) : 0; }
        segs.push_back(acc);
        if (acc == 0) { segs.resize(l2.seg_at); continue; }
        l2.total = acc;
        // 128 consecutive rows of the list cross at most (126 / shortest run) + 1 utterance boundaries, each skipping the halo rows
        // nobody reads: the physical rows a GEMM tile reaches over (a strided list: not bounded here, the strip form is not used)
        l2.span128 = st == 1 ? span_of_runs(ob.lext, ob.rext) : 0;
        l2.span160 = st == 1 ? span_of_runs(ob.lext, ob.rext, 160) : 0;
        lists.push_back(l2);
      }
    }
    const size_t n1 = (size_t)n_utts + 1;
    const size_t geo_bytes = n1 * sizeof(int64_t) + 4 * n1 * sizeof(int), bytes = geo_bytes + segs.size() * sizeof(int);
    char *hp = static_cast<char *>(harena.Alloc(bytes));
    char *dp = static_cast<char *>(arena_.Alloc(bytes));
    int64_t *h_so = reinterpret_cast<int64_t *>(hp);
    int *h_T = reinterpret_cast<int *>(hp + n1 * sizeof(int64_t)), *h_rb = h_T + n1, *h_fb = h_rb + n1, *h_ib = h_fb + n1;
    std::memcpy(h_so, sample_offsets, n1 * sizeof(int64_t));
    std::memcpy(h_T, T.data(), sizeof(int) * n_utts);
    h_T[n_utts] = 0;
    std::memcpy(h_rb, row_base.data(), sizeof(int) * n1);
    std::memcpy(h_fb, frame_base.data(), sizeof(int) * n1);
    std::memcpy(h_ib, ivrow_base.data(), sizeof(int) * n1);
    if (!segs.empty()) std::memcpy(hp + geo_bytes, segs.data(), segs.size() * sizeof(int));
    RS_HIP(hipMemcpyAsync(dp, hp, bytes, hipMemcpyHostToDevice, s));
    int64_t *d_so = reinterpret_cast<int64_t *>(dp);
    int *d_T = reinterpret_cast<int *>(dp + n1 * sizeof(int64_t)), *d_rb = d_T + n1, *d_fb = d_rb + n1, *d_ib = d_fb + n1;
    const int *d_segs = reinterpret_cast<const int *>(dp + geo_bytes);
    int *d_ru = arena_.AllocT<int>(rows), *d_rt = arena_.AllocT<int>(rows);
    d_row_ivec = arena_.AllocT<int>(rows);
    if (host_row_ivec) RS_HIP(hipMemcpyAsync(d_row_ivec, row_ivec, sizeof(int) * rows, hipMemcpyHostToDevice, s));
    BatchSetup bs;
    std::memset(&bs, 0, sizeof(bs));
    bs.n_utts = n_utts; bs.rows = rows; bs.L = L_; bs.row_base = d_rb; bs.ivrow_base = d_ib; bs.row_utt = d_ru; bs.row_t = d_rt;
    bs.row_ivec = host_row_ivec ? nullptr : d_row_ivec;
    for (auto &l : lists) {
      int *out = arena_.AllocT<int>(l.total);
      bs.lists[bs.n_lists++] = {l.n_segs, l.total, l.L_eff, l.slab_len, d_segs + l.seg_at, out, l.stride, l.first};
      row_maps.maps.push_back({l.lext, l.rext, out, l.total, l.span128, l.stride, l.span160});
      if (l.lext == 0 && l.rext == 0 && l.stride == 1) d_frame_rows = out;
    }
    LaunchBatchSetup(bs, s);
    g.d_sample_off = d_so; g.d_num_frames = d_T; g.d_row_base = d_rb; g.d_frame_base = d_fb; g.d_row_utt = d_ru; g.d_row_t = d_rt;
  }
  auto falloc = [&](int ld) { return arena_.AllocT<float>(((size_t)rows + 2 * guard) * ld) + (size_t)guard * ld; };
  static const int chain = [] { const char *e = TuneEnv("RS_STAGE_CHAIN"); return e ? std::atoi(e) : 1; }();
  // (one chain per model: a single chain for all models of the process was no better on the two-model batch -- 10.75-10.96 ms
  // against 10.56-10.70 -- and its tail event would have to outlive the model that recorded it)
  std::mutex *const smu = stage_mu_;
  hipEvent_t *const stail = stage_tail_;
  const bool chained = chain != 0 && !streaming;
  Timer tm(s);
  tm.Mark();
  std::unique_lock<std::mutex> stage_lock;
  auto stage_begin = [&](int st) {
    if (!chained) return;
    stage_lock = std::unique_lock<std::mutex>(smu[st]);
    if (stail[st] && stail[st] != cx.stage_ev[st]) RS_HIP(hipStreamWaitEvent(s, stail[st], 0));
  };
  auto stage_end = [&](int st) {
    if (!chained) return;
    RS_HIP(hipEventRecord(cx.stage_ev[st], s));
    stail[st] = cx.stage_ev[st];
    stage_lock.unlock();
  };
  stage_begin(0);
  // ---- features
  std::vector<float *> bufp(nn.bufs.size(), nullptr);
  for (size_t b = 0; b < nn.bufs.size(); b++) bufp[b] = falloc(buf_ld[b]);
  const std::vector<ActImage> imgs = AllocImages(arena_, rows);
  float *raw = bufp[nn.input_buf];
  if (fc_.use_cmvn) {
    raw = falloc(ld_c);
    poison();
    LaunchMfcc(MfccWithDither(maxT), g, d_pcm, raw, ld_c, s, OthersInFlight());
    poison();
    LaunchOnlineCmvn(cmvn_nnet_dev_, g, raw, bufp[nn.input_buf], buf_ld[nn.input_buf], s);
  } else {
    poison();
    LaunchMfcc(MfccWithDither(maxT), g, d_pcm, raw, buf_ld[nn.input_buf], s, OthersInFlight());
  }
  const int raw_ld = fc_.use_cmvn ? ld_c : buf_ld[nn.input_buf];
  tm.Mark();
  // ---- iVector
  float *d_ivec = nullptr;
  if (has_iv) {
    float *cm = falloc(ld_c), *lda_raw = falloc(ld_l), *lda_norm = falloc(ld_l);
    poison();
    LaunchOnlineCmvn(cmvn_iv_dev_, g, raw, cm, ld_c, s);
    poison();
    LaunchGemm(MakeGemm(LdaPlan(raw_ld), {raw}, {raw_ld}, nullptr, 0, lda_raw, ld_l, cx.active_groups), rows, d_row_ivec, s);
    poison();
    LaunchGemm(MakeGemm(LdaPlan(ld_c), {cm}, {ld_c}, nullptr, 0, lda_norm, ld_l, cx.active_groups), rows, d_row_ivec, s);
    int *post_idx = arena_.AllocT<int>((size_t)rows * nsel);
    float *post_w = arena_.AllocT<float>((size_t)rows * nsel);
    poison();
    LaunchUbmPosteriors(ivec_dev_, g, lda_norm, ld_l, post_idx, post_w, s);
    double *gamma = arena_.AllocT<double>((size_t)n_utts * G), *wfeats = arena_.AllocT<double>((size_t)n_utts * G * Dl);
    double *linear = arena_.AllocT<double>((size_t)n_utts * Di), *quad = arena_.AllocT<double>((size_t)n_utts * usz);
    double *numf = arena_.AllocT<double>(n_utts), *x = arena_.AllocT<double>((size_t)n_utts * Di);
    d_ivec = arena_.AllocT<float>((size_t)n_ivrows * ld_i + 256);   // + slack: staging loads may read past a row's end
    if (streaming) {          // (whole utterances: the accumulate kernel starts the sums itself)
      RS_HIP(hipMemsetAsync(gamma, 0, sizeof(double) * (size_t)n_utts * G, s));
      RS_HIP(hipMemsetAsync(wfeats, 0, sizeof(double) * (size_t)n_utts * G * Dl, s));
    }
    RS_HIP(hipMemsetAsync(d_ivec, 0, sizeof(float) * (size_t)n_ivrows * ld_i, s));
    LaunchIvecInit(ivec_dev_, n_utts, linear, quad, x, numf, s);
    double *iv_scratch = arena_.AllocT<double>(IvecStatsScratchDoubles(ivec_dev_, n_utts));
    const float *stats_feats = fc_.ie.online_cmvn_iextractor ? lda_norm : lda_raw;
    if (!streaming) {
      poison();
      LaunchIvecAccumulate(ivec_dev_, g, stats_feats, ld_l, post_idx, post_w, nullptr, nullptr, gamma, wfeats, true, s);
      poison();
      LaunchIvecStats(ivec_dev_, n_utts, gamma, wfeats, linear, quad, numf, iv_scratch, s);
      LaunchIvecSolve(ivec_dev_, n_utts, linear, quad, numf, x, d_ivec, ld_i, nullptr, nullptr, s);
    } else {
      // per-chunk schedule tables: [step][utt] frame_begin, frame_end, out_row, active
      std::vector<int> fb((size_t)max_chunks * n_utts, 0), fe((size_t)max_chunks * n_utts, 0), orow((size_t)max_chunks * n_utts, -1),
          act((size_t)max_chunks * n_utts, 0);
      for (int u = 0; u < n_utts; u++) {
        int done = 0;
        for (size_t k = 0; k < chunk_last[u].size(); k++) {
          const size_t i = k * n_utts + u;
          const int last = chunk_last[u][k];
          orow[i] = ivrow_base[u] + (int)k;
          if (last + 1 > done) { fb[i] = done; fe[i] = last + 1; act[i] = 1; done = last + 1; }
        }
      }
      int *d_fb = arena_.AllocT<int>(fb.size()), *d_fe = arena_.AllocT<int>(fb.size()), *d_or = arena_.AllocT<int>(fb.size()),
          *d_ac = arena_.AllocT<int>(fb.size());
      RS_HIP(hipMemcpyAsync(d_fb, fb.data(), sizeof(int) * fb.size(), hipMemcpyHostToDevice, s));
      RS_HIP(hipMemcpyAsync(d_fe, fe.data(), sizeof(int) * fb.size(), hipMemcpyHostToDevice, s));
      RS_HIP(hipMemcpyAsync(d_or, orow.data(), sizeof(int) * fb.size(), hipMemcpyHostToDevice, s));
      RS_HIP(hipMemcpyAsync(d_ac, act.data(), sizeof(int) * fb.size(), hipMemcpyHostToDevice, s));
      RS_HIP(hipStreamSynchronize(s));
      if (Di <= 128) {
        IvecChunkChain(arena_, g, n_utts, max_chunks, stats_feats, ld_l, post_idx, post_w, d_fb, d_fe, d_or, d_ac, linear, quad, numf, x, nullptr, d_ivec, ld_i, s);
      } else {
        for (int k = 0; k < max_chunks; k++) {
          const size_t o = (size_t)k * n_utts;
          LaunchIvecAccumulate(ivec_dev_, g, stats_feats, ld_l, post_idx, post_w, d_fb + o, d_fe + o, gamma, wfeats, false, s);
          LaunchIvecStats(ivec_dev_, n_utts, gamma, wfeats, linear, quad, numf, iv_scratch, s);
          LaunchIvecSolve(ivec_dev_, n_utts, linear, quad, numf, x, d_ivec, ld_i, d_or + o, d_ac + o, s);
          LaunchIvecClear(ivec_dev_, n_utts, gamma, wfeats, s);
        }
      }
    }
  }
  stage_end(0);
  // ---- search work buffers (before the acoustic model: the last layer can be pipelined with the search)
  AllocSearch(&sp, arena_, s);
  const DecodeOptsDev &dopts = sp.dopts;
  DenseWork &dw = sp.dw;
  tm.Mark();
  // ---- acoustic model
  stage_begin(1);
  if (pipelined) {
    RunNnet(bufp, buf_ld, d_ivec, ld_i, d_row_ivec, rows, row_maps, cx.active_groups, 0, nn.ops.size() - 1, s, &imgs);
    // slab k: output layer on the main stream, then the search of that slab on the decode stream
    const size_t i = nn.ops.size() - 1;
    GemmDev gd = MakeGemm(gemm_plans_[i], bufp, buf_ld, d_ivec, ld_i, bufp[nn.ops[i].out_buf], buf_ld[nn.ops[i].out_buf], cx.active_groups, &imgs,
                          (int)nn.ops[i].out_buf);
    for (int k = 0; k < n_slabs; k++) {
      gd.row_map = d_frame_rows + slab_off[k];
      LaunchGemm(gd, slab_off[k + 1] - slab_off[k], d_row_ivec, s);
      RS_HIP(hipEventRecord(cx.slab_ev[k], s));
      RS_HIP(hipStreamWaitEvent(cx.stream_dec, cx.slab_ev[k], 0));
      LaunchDecodeReg(hclg_dev_, reg_dev_, dopts, g, bufp[nn.output_buf], buf_ld[nn.output_buf], dw, k == 0 ? -1 : k * slab_len,
                      k + 1 == n_slabs ? maxT + 1 : (k + 1) * slab_len, cx.stream_dec);
    }
    RS_HIP(hipEventRecord(cx.slab_ev[8], cx.stream_dec));
  } else {
    poison();
    RunNnet(bufp, buf_ld, d_ivec, ld_i, d_row_ivec, rows, row_maps, cx.active_groups, 0, nn.ops.size(), s, &imgs);
  }
  stage_end(1);
  float *ll = bufp[nn.output_buf];
  int ll_ld = buf_ld[nn.output_buf];
  BatchGeom gdec = g;                     // what the search sees: the utterances' decoder frames
  if (fsf > 1) {
    // the rows the decoder reads, gathered into a dense [decoder frame][pdf] array with its own geometry (no halo)
    const int nd = dec_base[n_utts];
    int *h_idx = harena.AllocT<int>((size_t)nd + 2 * (size_t)n_utts + 2);
    int *h_T = h_idx + nd, *h_base = h_T + n_utts;
    for (int u = 0; u < n_utts; u++) {
      for (int f = 0; f < T_dec[u]; f++) h_idx[dec_base[u] + f] = row_base[u] + L_ + f * fsf;
      h_T[u] = T_dec[u];
      h_base[u] = dec_base[u];
    }
    h_base[n_utts] = nd;
    int *d_idx = arena_.AllocT<int>((size_t)nd + 2 * (size_t)n_utts + 2);
    RS_HIP(hipMemcpyAsync(d_idx, h_idx, sizeof(int) * ((size_t)nd + 2 * (size_t)n_utts + 2), hipMemcpyHostToDevice, s));
    const int ld_dec = RoundUp(P, 4);
    float *ll_dec = arena_.AllocT<float>(((size_t)nd + 8) * ld_dec);
    if (nd > 0) LaunchCopyRows(ll, ll_ld, d_idx, ll_dec, ld_dec, nullptr, nd, P, s);
    ll = ll_dec; ll_ld = ld_dec;
    gdec.L = 0; gdec.R = 0; gdec.total_rows = nd; gdec.total_frames = nd; gdec.max_frames = maxT_dec;
    gdec.d_num_frames = d_idx + nd; gdec.d_row_base = d_idx + nd + n_utts; gdec.d_frame_base = d_idx + nd + n_utts;
    gdec.d_row_utt = nullptr; gdec.d_row_t = nullptr; gdec.d_sample_off = nullptr;
  }
  tm.Mark();
  // ---- decode
  if (pipelined) RS_HIP(hipStreamWaitEvent(s, cx.slab_ev[8], 0));       // the search of the last slab
  else { poison(); LaunchSearch(&sp, arena_, gdec, ll, ll_ld, s); }
  tm.Mark();
  CollectResults(sp, cx, gi, gdec, T_dec.data(), ll, ll_ld, nbest, lat_scale, s, out_utts, timings);
  tm.Mark();
  if (opts_.keep_intermediates) {
    for (int u = 0; u < n_utts; u++) {
      UttResult &ur = out_utts[u];
      ur.feat_dim = C; ur.num_pdfs = P; ur.ivec_dim = Di; ur.ivec_rows = has_iv ? ivrow_base[u + 1] - ivrow_base[u] : 0;
      if (T[u] == 0) continue;
      ur.feats.resize((size_t)T[u] * C);
      ur.loglikes.resize((size_t)T_dec[u] * P);
      const float *fin = bufp[nn.input_buf] + ((size_t)row_base[u] + L_) * buf_ld[nn.input_buf];
      RS_HIP(hipMemcpy2D(ur.feats.data(), sizeof(float) * C, fin, sizeof(float) * buf_ld[nn.input_buf], sizeof(float) * C, T[u], hipMemcpyDeviceToHost));
      const float *lin = ll + (fsf > 1 ? (size_t)dec_base[u] : (size_t)row_base[u] + L_) * ll_ld;
      RS_HIP(hipMemcpy2D(ur.loglikes.data(), sizeof(float) * P, lin, sizeof(float) * ll_ld, sizeof(float) * P, T_dec[u], hipMemcpyDeviceToHost));
      if (has_iv) {
        ur.ivector.resize((size_t)ur.ivec_rows * Di);
        RS_HIP(hipMemcpy2D(ur.ivector.data(), sizeof(float) * Di, d_ivec + (size_t)ivrow_base[u] * ld_i, sizeof(float) * ld_i,
                           sizeof(float) * Di, ur.ivec_rows, hipMemcpyDeviceToHost));
      }
    }
  }
  // kernel launches are not checked one by one; a failed launch of this thread surfaces here instead of as silent garbage
  { const hipError_t le = hipGetLastError(); if (le != hipSuccess) Fail(std::string("a kernel launch failed: ") + hipGetErrorString(le)); }
  timings[1] = tm.Ms(0, 1);
  timings[2] = tm.Ms(1, 2);
  timings[3] = tm.Ms(2, 3);
  timings[4] = tm.Ms(3, 4);
  timings[5] = tm.Ms(4, 5);
  timings[6] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - wall0).count();
}

}  // namespace rs
