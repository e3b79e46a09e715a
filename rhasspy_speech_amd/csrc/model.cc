#include "model.h"
#include "nnet3_setup.h"
#include "env.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <fstream>
#include <limits>
#include <set>
#include <sstream>

namespace rs {

// =============================================================================== options

static bool ParseBool(const std::string &v, const std::string &key) {
  if (v == "true" || v == "t" || v == "1" || v == "True" || v == "T" || v.empty()) return true;
  if (v == "false" || v == "f" || v == "0" || v == "False" || v == "F") return false;
  Fail("Invalid boolean value for --" + key + ": " + v);
}

int MfccOptions::PaddedWindowSize() const {
  int w = WindowSize();
  if (!round_pow2) return w;
  int p = 1;
  while (p < w) p <<= 1;
  return p;
}

void ReadMfccOptions(const std::string &conf_path, MfccOptions *o) {
  for (auto &kv : ReadConfigFile(conf_path)) {
    const std::string &k = kv.first, &v = kv.second;
    if (k == "sample-frequency") o->samp_freq = std::stof(v);
    else if (k == "frame-length") o->frame_length_ms = std::stof(v);
    else if (k == "frame-shift") o->frame_shift_ms = std::stof(v);
    else if (k == "preemphasis-coefficient") o->preemph = std::stof(v);
    else if (k == "remove-dc-offset") o->remove_dc = ParseBool(v, k);
    else if (k == "dither") o->dither = std::stof(v);
    else if (k == "window-type") o->window_type = v;
    else if (k == "blackman-coeff") o->blackman_coeff = std::stof(v);
    else if (k == "round-to-power-of-two") o->round_pow2 = ParseBool(v, k);
    else if (k == "snip-edges") o->snip_edges = ParseBool(v, k);
    else if (k == "allow-downsample") o->allow_downsample = ParseBool(v, k);
    else if (k == "allow-upsample") o->allow_upsample = ParseBool(v, k);
    else if (k == "max-feature-vectors" || k == "debug-mel") {}
    else if (k == "num-mel-bins") o->num_bins = std::stoi(v);
    else if (k == "low-freq") o->low_freq = std::stof(v);
    else if (k == "high-freq") o->high_freq = std::stof(v);
    else if (k == "vtln-low") o->vtln_low = std::stof(v);
    else if (k == "vtln-high") o->vtln_high = std::stof(v);
    else if (k == "num-ceps") o->num_ceps = std::stoi(v);
    else if (k == "use-energy") o->use_energy = ParseBool(v, k);
    else if (k == "energy-floor") o->energy_floor = std::stof(v);
    else if (k == "raw-energy") o->raw_energy = ParseBool(v, k);
    else if (k == "cepstral-lifter") o->cepstral_lifter = std::stof(v);
    else if (k == "htk-compat") o->htk_compat = ParseBool(v, k);
    else Fail("Invalid option --" + k + "=" + v + " in config file " + conf_path);
  }
}

int NumFrames(long num_samples, const MfccOptions &o) {
  long shift = o.WindowShift(), len = o.WindowSize();
  if (num_samples < len) return 0;
  return (int)(1 + (num_samples - len) / shift);
}

// mel-computations.h:81-87
static inline float MelScale(float freq) { return 1127.0f * logf(1.0f + freq / 700.0f); }

void BuildMfccTables(const MfccOptions &o, MfccTables *t) {
  t->opts = o;
  if (!o.snip_edges) Fail("mfcc: --snip-edges=false is not supported by the HIP feature kernel");
  if (o.htk_compat) Fail("mfcc: --htk-compat=true is not supported by the HIP feature kernel");
  if (o.num_ceps > o.num_bins) Fail("num-ceps cannot be larger than num-mel-bins.");
  t->win = o.WindowSize();
  t->shift = o.WindowShift();
  t->padded = o.PaddedWindowSize();
  if (t->padded != 512 && t->padded != 2048)
    Fail("mfcc: padded window size " + std::to_string(t->padded) + " unsupported by the HIP FFT (need 512 or 2048)");
  if (o.num_bins > 64 || o.num_ceps > 64) Fail("mfcc: more than 64 mel bins / cepstra are not supported by the HIP kernel");
  t->nbins = o.num_bins;
  t->nceps = o.num_ceps;
  // window (feature-window.cc:113-125): computed in double, stored as float
  t->window.resize(t->win);
  double a = 2.0 * M_PI / (t->win - 1);
  for (int i = 0; i < t->win; i++) {
    double x = (double)i, w;
    if (o.window_type == "hanning") w = 0.5 - 0.5 * cos(a * x);
    else if (o.window_type == "sine") w = sin(0.5 * a * x);
    else if (o.window_type == "hamming") w = 0.54 - 0.46 * cos(a * x);
    else if (o.window_type == "povey") w = pow(0.5 - 0.5 * cos(a * x), 0.85);
    else if (o.window_type == "rectangular") w = 1.0;
    else if (o.window_type == "blackman") w = o.blackman_coeff - 0.5 * cos(a * x) + (0.5 - o.blackman_coeff) * cos(2 * a * x);
    else Fail("Invalid window type " + o.window_type);
    t->window[i] = (float)w;
  }
  // mel banks (mel-computations.cc:33-142), float arithmetic as in the reference
  float sample_freq = o.samp_freq;
  int num_fft_bins = t->padded / 2;
  float nyquist = 0.5f * sample_freq;
  float low_freq = o.low_freq, high_freq = o.high_freq > 0.0f ? o.high_freq : nyquist + o.high_freq;
  if (low_freq < 0.0f || low_freq >= nyquist || high_freq <= 0.0f || high_freq > nyquist || high_freq <= low_freq)
    Fail("Bad values in options: low-freq " + std::to_string(low_freq) + " and high-freq " + std::to_string(high_freq));
  float fft_bin_width = sample_freq / t->padded;
  float mel_low = MelScale(low_freq), mel_high = MelScale(high_freq);
  float mel_delta = (mel_high - mel_low) / (o.num_bins + 1);
  t->mel_offset.assign(o.num_bins, 0);
  t->mel_len.assign(o.num_bins, 0);
  t->mel_start.assign(o.num_bins, 0);
  t->mel_weights.clear();
  for (int bin = 0; bin < o.num_bins; bin++) {
    float left = mel_low + bin * mel_delta, center = mel_low + (bin + 1) * mel_delta, right = mel_low + (bin + 2) * mel_delta;
    std::vector<float> this_bin(num_fft_bins, 0.0f);
    int first = -1, last = -1;
    for (int i = 0; i < num_fft_bins; i++) {
      float freq = fft_bin_width * i;
      float mel = MelScale(freq);
      if (mel > left && mel < right) {
        float w;
        if (mel <= center) w = (mel - left) / (center - left);
        else w = (right - mel) / (right - center);
        this_bin[i] = w;
        if (first == -1) first = i;
        last = i;
      }
    }
    if (first == -1) Fail("You may have set --num-mel-bins too large.");
    t->mel_offset[bin] = first;
    t->mel_len[bin] = last + 1 - first;
    t->mel_start[bin] = (int)t->mel_weights.size();
    t->mel_weights.insert(t->mel_weights.end(), this_bin.begin() + first, this_bin.begin() + last + 1);
  }
  // DCT (matrix-functions.cc:592-608): rows k of an N x N type-II DCT, first num_ceps rows kept
  int N = o.num_bins;
  t->dct.assign((size_t)o.num_ceps * N, 0.0f);
  {
    double normalizer = std::sqrt(1.0 / (double)N);
    for (int j = 0; j < N; j++) t->dct[j] = (float)normalizer;
    normalizer = std::sqrt(2.0 / (double)N);
    for (int k = 1; k < o.num_ceps; k++)
      for (int n = 0; n < N; n++) t->dct[(size_t)k * N + n] = (float)(normalizer * cos((double)M_PI / N * (n + 0.5) * k));
  }
  t->lifter.assign(o.num_ceps, 1.0f);
  if (o.cepstral_lifter != 0.0f)
    for (int i = 0; i < o.num_ceps; i++)
      t->lifter[i] = (float)(1.0 + 0.5 * o.cepstral_lifter * sin(M_PI * i / o.cepstral_lifter));
  t->log_energy_floor = o.energy_floor > 0.0f ? logf(o.energy_floor) : -std::numeric_limits<float>::infinity();
}

static void ReadCmvnOptions(const std::string &path, CmvnOptions *c) {
  for (auto &kv : ReadConfigFile(path)) {
    const std::string &k = kv.first, &v = kv.second;
    if (k == "cmn-window") c->cmn_window = std::stoi(v);
    else if (k == "speaker-frames") c->speaker_frames = std::stoi(v);
    else if (k == "global-frames") c->global_frames = std::stoi(v);
    else if (k == "norm-means" || k == "norm-mean") c->normalize_mean = ParseBool(v, k);
    else if (k == "norm-vars") c->normalize_variance = ParseBool(v, k);
    else if (k == "skip-dims") { if (!v.empty()) Fail("online cmvn: --skip-dims is not supported"); }
    else Fail("Invalid option --" + k + " in config file " + path);
  }
  if (c->normalize_variance) Fail("online cmvn: --norm-vars=true is not supported by the HIP feature kernels");
}

static void ReadDiagGmm(const std::string &path, IvectorExtractor *ie) {
  KaldiReader r(path);
  std::string tok = r.ReadToken();
  if (tok != "<DiagGMMBegin>" && tok != "<DiagGMM>") Fail(path + ": Expected <DiagGMM>, got " + tok);
  tok = r.ReadToken();
  if (tok == "<GCONSTS>") {
    r.ReadVector(&ie->gconsts);
    r.ExpectToken("<WEIGHTS>");
  } else if (tok != "<WEIGHTS>") {
    Fail(path + ": DiagGmm::Read, expected <WEIGHTS> or <GCONSTS>, got " + tok);
  }
  r.ReadVector(&ie->weights);
  r.ExpectToken("<MEANS_INVVARS>");
  r.ReadMatrix(&ie->means_invvars);
  r.ExpectToken("<INV_VARS>");
  r.ReadMatrix(&ie->inv_vars);
  tok = r.ReadToken();
  if (tok != "<DiagGMMEnd>" && tok != "</DiagGMM>") Fail(path + ": Expected </DiagGMM>, got " + tok);
  // DiagGmm::ComputeGconsts (gmm/diag-gmm.cc:114-150): gconsts are always recomputed on read.
  int G = ie->means_invvars.rows, D = ie->means_invvars.cols;
  ie->gconsts.assign(G, 0.0f);
  float offset = (float)(-0.5 * 1.8378770664093454835606594728112 * D);  // M_LOG_2PI
  for (int g = 0; g < G; g++) {
    float gc = logf(ie->weights[g]) + offset;
    for (int d = 0; d < D; d++) {
      float iv = ie->inv_vars(g, d), miv = ie->means_invvars(g, d);
      // the reference's expression mixes double literals with float operands (evaluated in double)
      gc = (float)((double)gc + (0.5 * (double)logf(iv) - 0.5 * (double)miv * (double)miv / (double)iv));
    }
    if (std::isnan(gc)) Fail(path + ": not a number in gconst computation");
    if (std::isinf(gc) && gc > 0) gc = -gc;
    ie->gconsts[g] = gc;
  }
}

static void ReadIvectorExtractorFile(const std::string &path, IvectorExtractor *ie) {
  KaldiReader r(path);
  r.ExpectToken("<IvectorExtractor>");
  r.ExpectToken("<w>");
  MatD w;
  r.ReadMatrixD(&w);
  if (w.rows != 0) Fail(path + ": iVector extractors with iVector-dependent weights are not supported");
  r.ExpectToken("<w_vec>");
  std::vector<double> wv;
  r.ReadVectorD(&wv);
  r.ExpectToken("<M>");
  int n = r.ReadInt32();
  if (n <= 0) Fail(path + ": bad number of Gaussians");
  ie->M.resize(n);
  for (int i = 0; i < n; i++) r.ReadMatrixD(&ie->M[i]);
  r.ExpectToken("<SigmaInv>");
  ie->sigma_inv.resize(n);
  for (int i = 0; i < n; i++) {
    int dim;
    r.ReadSpMatrixD(&dim, &ie->sigma_inv[i]);
    if (dim != ie->M[i].rows) Fail(path + ": SigmaInv dim mismatch");
  }
  r.ExpectToken("<IvectorOffset>");
  ie->prior_offset = r.ReadDouble();
  r.ExpectToken("</IvectorExtractor>");
}

void IvectorExtractor::ComputeDerived() {
  // ivector-extractor.cc:205-218: U_g = M_g^T Sigma_g^{-1} M_g (packed), Sigma_inv_M_g = Sigma_g^{-1} M_g
  int G = num_gauss(), D = feat_dim(), I = ivector_dim();
  size_t usz = (size_t)I * (I + 1) / 2;
  U.assign((size_t)G * usz, 0.0);
  sigma_inv_M.assign((size_t)G * D * I, 0.0);
  std::vector<double> S((size_t)D * D);
  for (int g = 0; g < G; g++) {
    const std::vector<double> &p = sigma_inv[g];
    for (int r = 0, k = 0; r < D; r++)
      for (int c = 0; c <= r; c++, k++) S[(size_t)r * D + c] = S[(size_t)c * D + r] = p[k];
    double *sim = &sigma_inv_M[(size_t)g * D * I];
    const MatD &Mg = M[g];
    for (int r = 0; r < D; r++)
      for (int c = 0; c < D; c++) {
        double s = S[(size_t)r * D + c];
        if (s == 0.0) continue;
        const double *mrow = &Mg.d[(size_t)c * I];
        double *orow = &sim[(size_t)r * I];
        for (int i = 0; i < I; i++) orow[i] += s * mrow[i];
      }
    double *u = &U[(size_t)g * usz];
    for (int i = 0, k = 0; i < I; i++)
      for (int j = 0; j <= i; j++, k++) {
        double acc = 0;
        for (int d = 0; d < D; d++) acc += Mg.d[(size_t)d * I + i] * sim[(size_t)d * I + j];
        u[k] = acc;
      }
  }
}

static std::string DirName(const std::string &p) {
  size_t s = p.find_last_of('/');
  return s == std::string::npos ? "." : p.substr(0, s);
}

static void ReadKaldiMatrixFile(const std::string &path, MatF *m) { KaldiReader r(path); r.ReadMatrix(m); }
static void ReadKaldiMatrixFileD(const std::string &path, MatD *m) { KaldiReader r(path); r.ReadMatrixD(m); }

static void ReadIvectorConfig(const std::string &path, IvectorExtractor *ie, int base_dim) {
  std::string lda, gstats, cmvn_conf, splice_conf, ubm, extractor;
  for (auto &kv : ReadConfigFile(path)) {
    const std::string &k = kv.first, &v = kv.second;
    if (k == "lda-matrix") lda = v;
    else if (k == "global-cmvn-stats") gstats = v;
    else if (k == "cmvn-config") cmvn_conf = v;
    else if (k == "online-cmvn-iextractor") ie->online_cmvn_iextractor = ParseBool(v, k);
    else if (k == "splice-config") splice_conf = v;
    else if (k == "diag-ubm") ubm = v;
    else if (k == "ivector-extractor") extractor = v;
    else if (k == "ivector-period") ie->ivector_period = std::stoi(v);
    else if (k == "num-gselect") ie->num_gselect = std::stoi(v);
    else if (k == "min-post") ie->min_post = std::stof(v);
    else if (k == "posterior-scale") ie->posterior_scale = std::stof(v);
    else if (k == "max-count") ie->max_count = std::stof(v);
    else if (k == "num-cg-iters") ie->num_cg_iters = std::stoi(v);
    else if (k == "use-most-recent-ivector") ie->use_most_recent_ivector = ParseBool(v, k);
    else if (k == "greedy-ivector-extractor") ie->greedy = ParseBool(v, k);
    else if (k == "max-remembered-frames") ie->max_remembered_frames = std::stof(v);
    else Fail("Invalid option --" + k + " in config file " + path);
  }
  const char *note = " (note: this may be needed in the file supplied to --ivector-extractor-config)";
  if (lda.empty()) Fail(std::string("--lda-matrix option must be set") + note);
  if (gstats.empty()) Fail(std::string("--global-cmvn-stats option must be set") + note);
  if (cmvn_conf.empty()) Fail(std::string("--cmvn-config option must be set") + note);
  if (splice_conf.empty()) Fail(std::string("--splice-config option must be set") + note);
  if (ubm.empty()) Fail(std::string("--diag-ubm option must be set") + note);
  if (extractor.empty()) Fail(std::string("--ivector-extractor option must be set") + note);
  if (ie->greedy) ie->use_most_recent_ivector = true;
  ReadKaldiMatrixFile(lda, &ie->lda);
  ReadKaldiMatrixFileD(gstats, &ie->global_cmvn);
  ReadCmvnOptions(cmvn_conf, &ie->cmvn);
  for (auto &kv : ReadConfigFile(splice_conf)) {
    if (kv.first == "left-context") ie->splice_left = std::stoi(kv.second);
    else if (kv.first == "right-context") ie->splice_right = std::stoi(kv.second);
    else Fail("Invalid option --" + kv.first + " in config file " + splice_conf);
  }
  ReadDiagGmm(ubm, ie);
  ReadIvectorExtractorFile(extractor, ie);
  // OnlineIvectorExtractionInfo::Check (online-ivector-feature.cc:83-99)
  if (ie->global_cmvn.rows != 2) Fail("global_cmvn_stats must have 2 rows");
  int bdim = ie->global_cmvn.cols - 1, nsp = ie->splice_left + 1 + ie->splice_right;
  if (bdim != base_dim) Fail("iVector extractor: global CMVN stats dim " + std::to_string(bdim) + " != feature dim " + std::to_string(base_dim));
  if (ie->lda.cols != bdim * nsp && ie->lda.cols != bdim * nsp + 1) Fail("iVector extractor: LDA matrix has wrong number of columns");
  if (ie->lda.rows != ie->means_invvars.cols) Fail("iVector extractor: LDA rows != UBM dim");
  if (ie->means_invvars.cols != ie->M[0].rows) Fail("iVector extractor: UBM dim != extractor feature dim");
  if ((int)ie->M.size() != ie->means_invvars.rows) Fail("iVector extractor: #Gaussians mismatch between UBM and extractor");
  if (ie->ivector_period <= 0 || ie->num_gselect <= 0 || !(ie->min_post < 0.5f) ||
      !(ie->posterior_scale > 0.0f && ie->posterior_scale <= 1.0f))
    Fail("iVector extractor: invalid options");
  ie->ComputeDerived();
  ie->present = true;
}

void ReadFeatureConfig(const std::string &online_conf, FeatureConfig *fc) {
  std::string mfcc_conf, ivec_conf, cmvn_conf, gstats;
  fc->conf_path = online_conf;
  for (auto &kv : ReadConfigFile(online_conf)) {
    const std::string &k = kv.first, &v = kv.second;
    if (k == "feature-type") fc->feature_type = v;
    else if (k == "mfcc-config") mfcc_conf = v;
    else if (k == "ivector-extraction-config") ivec_conf = v;
    else if (k == "cmvn-config") cmvn_conf = v;
    else if (k == "global-cmvn-stats") gstats = v;
    else if (k == "add-pitch") { if (ParseBool(v, k)) Fail("--add-pitch=true is not supported"); }
    else if (k == "plp-config" || k == "fbank-config" || k == "online-pitch-config") {}
    else if (k.compare(0, 9, "endpoint.") == 0 || k.compare(0, 26, "ivector-silence-weighting.") == 0) {}
    // decodable / decoder options registered on the same parser (NnetSimpleLoopedComputationOptions, decodable-simple-looped.h:68-81;
    // LatticeFasterDecoderConfig + its det_opts, lattice-faster-decoder.h:67-85; the binaries' own --online, --do-endpointing,
    // --chunk-length ...): kept, and applied or refused by Model::Model -- never dropped
    else if (k == "frame-subsampling-factor" || k == "frames-per-chunk" || k == "acoustic-scale" ||
             k == "extra-left-context-initial" || k == "beam" || k == "max-active" || k == "min-active" ||
             k == "lattice-beam" || k == "beam-delta" || k == "prune-interval" || k == "hash-ratio" ||
             k == "determinize-lattice" || k == "minimize" || k == "phone-determinize" || k == "word-determinize" ||
             k == "max-mem" || k == "debug-computation" || k == "online" || k == "do-endpointing" || k == "chunk-length" ||
             k == "delta" || k == "num-threads-startup")
      fc->decoder_conf.emplace_back(k, v);
    else Fail("Invalid option --" + k + "=" + v + " in config file " + online_conf);
  }
  if (fc->feature_type != "mfcc") {
    if (fc->feature_type == "plp" || fc->feature_type == "fbank")
      Fail("feature type " + fc->feature_type + " is not supported by the HIP path (mfcc only)");
    Fail("Invalid feature type: " + fc->feature_type + ". Supported feature types: mfcc, plp, fbank.");
  }
  MfccOptions mo;
  if (!mfcc_conf.empty()) ReadMfccOptions(mfcc_conf, &mo);
  BuildMfccTables(mo, &fc->mfcc);
  fc->use_cmvn = !cmvn_conf.empty();
  if (fc->use_cmvn) {
    ReadCmvnOptions(cmvn_conf, &fc->cmvn);
    if (gstats.empty()) Fail("--global-cmvn-stats option is required  when --cmvn-config is specified.");
    ReadKaldiMatrixFileD(gstats, &fc->global_cmvn);
    if (fc->global_cmvn.rows != 2 || fc->global_cmvn.cols != mo.num_ceps + 1) Fail("global cmvn stats have the wrong dimension");
  }
  if (!ivec_conf.empty()) ReadIvectorConfig(ivec_conf, &fc->ie, mo.num_ceps);
}

// =============================================================================== transition model

void TransitionModel::Read(KaldiReader &r) {
  r.ExpectToken("<TransitionModel>");
  r.ExpectToken("<Topology>");
  // per topology entry: per state: (forward_pdf_class, self_loop_pdf_class, transitions (dst, prob))
  std::vector<std::vector<HmmState>> entries;
  std::vector<int32_t> phones, phone2idx;
  if (r.binary()) {
    r.ReadIntVector(&phones);
    r.ReadIntVector(&phone2idx);
    int sz = r.ReadInt32();
    bool is_hmm = true;
    if (sz == -1) { is_hmm = false; sz = r.ReadInt32(); }
    entries.resize(sz);
    for (int i = 0; i < sz; i++) {
      int ns = r.ReadInt32();
      entries[i].resize(ns);
      for (int j = 0; j < ns; j++) {
        entries[i][j].fwd = r.ReadInt32();
        entries[i][j].self = is_hmm ? entries[i][j].fwd : r.ReadInt32();
        int nt = r.ReadInt32();
        entries[i][j].trans.resize(nt);
        for (int k = 0; k < nt; k++) { entries[i][j].trans[k].first = r.ReadInt32(); entries[i][j].trans[k].second = r.ReadFloat(); }
      }
    }
    r.ExpectToken("</Topology>");
  } else {
    while (true) {
      std::string tok = r.ReadToken();
      if (tok == "</Topology>") break;
      if (tok != "<TopologyEntry>") Fail("Reading HmmTopology object, expected </Topology> or <TopologyEntry>, got " + tok);
      r.ExpectToken("<ForPhones>");
      std::vector<int> these;
      while (true) {
        std::string s = r.ReadToken();
        if (s == "</ForPhones>") break;
        these.push_back(std::stoi(s));
      }
      std::vector<HmmState> entry;
      tok = r.ReadToken();
      while (tok != "</TopologyEntry>") {
        if (tok != "<State>") Fail("Expected </TopologyEntry> or <State>, got instead " + tok);
        int state = r.ReadInt32();
        if (state != (int)entry.size()) Fail("States are expected to be in order from zero");
        HmmState hs;
        tok = r.ReadToken();
        if (tok == "<PdfClass>") { hs.fwd = hs.self = r.ReadInt32(); tok = r.ReadToken(); }
        else if (tok == "<ForwardPdfClass>") {
          hs.fwd = r.ReadInt32();
          r.ExpectToken("<SelfLoopPdfClass>");
          hs.self = r.ReadInt32();
          tok = r.ReadToken();
        }
        while (tok == "<Transition>") {
          int dst = r.ReadInt32();
          float p = r.ReadFloat();
          hs.trans.emplace_back(dst, p);
          tok = r.ReadToken();
        }
        if (tok != "</State>") Fail("Expected </State>, got instead " + tok);
        entry.push_back(hs);
        tok = r.ReadToken();
      }
      int idx = (int)entries.size();
      entries.push_back(entry);
      for (int p : these) {
        if ((int)phone2idx.size() <= p) phone2idx.resize(p + 1, -1);
        if (p <= 0 || phone2idx[p] != -1) Fail("bad phone in topology");
        phone2idx[p] = idx;
        phones.push_back(p);
      }
    }
  }
  std::string tok = r.ReadToken();
  if (tok != "<Triples>" && tok != "<Tuples>") Fail("TransitionModel: expected <Triples> or <Tuples>, got " + tok);
  const bool has_self = (tok == "<Tuples>");
  int n = r.ReadInt32();
  std::vector<Tuple> tp(n);
  for (int i = 0; i < n; i++) {
    tp[i].phone = r.ReadInt32();
    tp[i].hmm_state = r.ReadInt32();
    tp[i].fwd = r.ReadInt32();
    tp[i].self = has_self ? r.ReadInt32() : tp[i].fwd;
  }
  tok = r.ReadToken();
  if (tok != "</Triples>" && tok != "</Tuples>") Fail("TransitionModel: expected </Triples> or </Tuples>");
  // ComputeDerived (transition-model.cc:144-177)
  id2pdf.assign(1, 0);
  id2phone.assign(1, 0);
  id2hmm_state.assign(1, 0);
  id2self_loop.assign(1, 0);
  std::vector<int> self_loop_of(1, 0);      // per transition-id: the self-loop transition-id of its transition-state (0 = none)
  topo_entries = entries;
  phone2entry = phone2idx;
  tuples = tp;
  tstate_first_tid.assign(1, 0);
  id2tstate.assign(1, 0);
  num_pdfs = 0;
  for (int ts = 0; ts < n; ts++) {
    const Tuple &t = tp[ts];
    if (t.phone <= 0 || t.phone >= (int)phone2idx.size() || phone2idx[t.phone] < 0) Fail("TransitionModel: phone without topology");
    const auto &entry = entries[phone2idx[t.phone]];
    if (t.hmm_state < 0 || t.hmm_state >= (int)entry.size()) Fail("TransitionModel: bad hmm-state");
    const HmmState &hs = entry[t.hmm_state];
    num_pdfs = std::max(num_pdfs, 1 + std::max(t.fwd, t.self));
    const int first_tid = (int)id2pdf.size();
    tstate_first_tid.push_back(first_tid);
    int sl_tid = 0;
    for (size_t k = 0; k < hs.trans.size(); k++) {
      bool self_loop = (hs.trans[k].first == t.hmm_state);
      if (self_loop && sl_tid == 0) sl_tid = first_tid + (int)k;       // SelfLoopOf (transition-model.cc:360-373)
      id2pdf.push_back(self_loop ? t.self : t.fwd);
      id2phone.push_back(t.phone);
      id2hmm_state.push_back(t.hmm_state);
      id2self_loop.push_back(self_loop ? 1 : 0);
      id2tstate.push_back(ts + 1);
    }
    for (size_t k = 0; k < hs.trans.size(); k++) self_loop_of.push_back(sl_tid);
  }
  self_loop_of_id = self_loop_of;
  r.ExpectToken("<LogProbs>");
  std::vector<float> lp;
  r.ReadVector(&lp);
  if (lp.size() != id2pdf.size()) Fail("TransitionModel: <LogProbs> size does not match the number of transition-ids");
  log_prob = lp;
  // ComputeDerivedOfProbs (transition-model.cc:375-392)
  non_self_loop_log_prob.assign(lp.size(), 0.0f);
  for (size_t tid = 1; tid < lp.size(); tid++) {
    if (self_loop_of[tid] == 0) continue;
    float p = 1.0f - expf(lp[self_loop_of[tid]]);
    if (p <= 0.0f) p = 1.0e-10f;
    non_self_loop_log_prob[tid] = logf(p);
  }
  r.ExpectToken("</LogProbs>");
  r.ExpectToken("</TransitionModel>");
}

int TransitionModel::TupleToTransitionState(int phone, int hmm_state, int fwd, int self) const {
  // transition-model.cc:179-196: binary search in the sorted tuple table
  size_t lo = 0, hi = tuples.size();
  auto less = [](const Tuple &a, int p, int h, int f, int s) {
    if (a.phone != p) return a.phone < p;
    if (a.hmm_state != h) return a.hmm_state < h;
    if (a.fwd != f) return a.fwd < f;
    return a.self < s;
  };
  while (lo < hi) { const size_t mid = (lo + hi) / 2; if (less(tuples[mid], phone, hmm_state, fwd, self)) lo = mid + 1; else hi = mid; }
  if (lo < tuples.size() && tuples[lo].phone == phone && tuples[lo].hmm_state == hmm_state && tuples[lo].fwd == fwd && tuples[lo].self == self)
    return (int)lo + 1;
  return 0;
}

int TransitionModel::NumPdfClasses(int phone) const {
  // hmm-topology.cc:275-285: 1 + the largest pdf class of the phone's entry
  if (phone <= 0 || phone >= (int)phone2entry.size() || phone2entry[phone] < 0) Fail("TransitionModel: phone " + std::to_string(phone) + " has no topology");
  int mx = -1;
  for (const HmmState &h : topo_entries[phone2entry[phone]]) mx = std::max(mx, std::max(h.fwd, h.self));
  return mx + 1;
}

// =============================================================================== nnet3 parsing

static const std::set<std::string> kIntVectorFields = {"<TimeOffsets>", "<ColumnMap>", "<Sizes>"};

static Component ReadComponent(KaldiReader &r) {
  Component c;
  std::string open = r.ReadToken();
  if (open.size() < 3 || open[0] != '<' || open.back() != '>') Fail(r.name() + ": expected a component opening tag, got " + open.substr(0, 40));
  c.type = open.substr(1, open.size() - 2);
  std::string close = "</" + c.type + ">";
  while (true) {
    std::string tok = r.ReadToken();
    if (tok == close) break;
    if (tok.empty() || tok[0] != '<') Fail(r.name() + ": malformed component " + c.type + " near " + tok.substr(0, 40));
    if (kIntVectorFields.count(tok)) { r.ReadIntVector(&c.iv[tok]); continue; }
    c.b[tok] = true;   // present (bare flag unless values follow)
    while (true) {
      int ch = r.PeekChar();
      if (ch < 0) Fail(r.name() + ": unexpected EOF in component " + c.type);
      if (ch == '<') break;
      if (r.binary()) {
        if (ch == 4 || ch == 8) { double fv; int64_t iv; r.ReadBasicAny(&fv, &iv); c.f[tok].push_back(fv); c.i[tok].push_back(iv); continue; }
        std::string head = r.PeekToken();
        if (head == "FM" || head == "DM" || head == "CM" || head == "CM2" || head == "CM3") { r.ReadMatrix(&c.m[tok]); continue; }
        if (head == "FV" || head == "DV") { r.ReadVector(&c.v[tok]); continue; }
        if (ch == 'T' || ch == 'F') { bool bv = r.ReadBool(); c.f[tok].push_back(bv ? 1.0 : 0.0); c.i[tok].push_back(bv); continue; }
        Fail(r.name() + ": cannot parse field " + tok + " of component " + c.type);
      } else {
        if (ch == '[') {
          // vector or matrix: decide by row count
          MatF m;
          r.ReadMatrix(&m);
          if (m.rows <= 1 && tok != "<LinearParams>" && tok != "<Params>") c.v[tok] = m.d;
          else c.m[tok] = m;
          continue;
        }
        if (ch == 'T' || ch == 'F') {
          std::string t = r.PeekToken();
          if (t == "T" || t == "F") { bool bv = r.ReadBool(); c.f[tok].push_back(bv ? 1.0 : 0.0); c.i[tok].push_back(bv); continue; }
        }
        { double fv; int64_t iv; r.ReadBasicAny(&fv, &iv); c.f[tok].push_back(fv); c.i[tok].push_back(iv); }
      }
    }
  }
  return c;
}

// ---- descriptor parser (nnet3/nnet-descriptor.cc grammar subset) ----
namespace {
struct DescParser {
  const std::string &s;
  size_t p = 0;
  const Nnet &net;
  const std::vector<int> &node_dims;
  DescParser(const std::string &str, const Nnet &n, const std::vector<int> &dims) : s(str), net(n), node_dims(dims) {}
  void Ws() { while (p < s.size() && std::isspace((unsigned char)s[p])) p++; }
  std::string Ident() {
    Ws();
    size_t b = p;
    while (p < s.size() && (std::isalnum((unsigned char)s[p]) || s[p] == '_' || s[p] == '-' || s[p] == '.')) p++;
    return s.substr(b, p - b);
  }
  void Expect(char c) {
    Ws();
    if (p >= s.size() || s[p] != c) Fail("nnet3 descriptor parse error in '" + s + "' at " + std::to_string(p) + ": expected '" + c + "'");
    p++;
  }
  double Number() {
    Ws();
    const char *b = s.c_str() + p;
    char *e;
    double v = std::strtod(b, &e);
    if (e == b) Fail("nnet3 descriptor parse error in '" + s + "': expected number");
    p += e - b;
    return v;
  }
  std::vector<DescPart> Parse() {
    size_t save = p;
    std::string id = Ident();
    Ws();
    if (p < s.size() && s[p] == '(') {
      p++;
      std::vector<DescPart> out;
      if (id == "Append") {
        while (true) {
          auto sub = Parse();
          out.insert(out.end(), sub.begin(), sub.end());
          Ws();
          if (p < s.size() && s[p] == ',') { p++; continue; }
          break;
        }
      } else if (id == "Sum") {
        out = Parse();
        while (true) {
          Ws();
          if (p < s.size() && s[p] == ',') {
            p++;
            auto b = Parse();
            if (b.size() != out.size()) Fail("nnet3 descriptor: Sum() of differently structured Appends is not supported: " + s);
            for (size_t i = 0; i < out.size(); i++) {
              if (b[i].dim != out[i].dim) Fail("nnet3 descriptor: Sum() dimension mismatch: " + s);
              out[i].terms.insert(out[i].terms.end(), b[i].terms.begin(), b[i].terms.end());
            }
            continue;
          }
          break;
        }
      } else if (id == "Offset") {
        out = Parse();
        Expect(',');
        int t = (int)Number();
        Ws();
        if (p < s.size() && s[p] == ',') { p++; if ((int)Number() != 0) Fail("nnet3 descriptor: x-offsets are not supported: " + s); }
        for (auto &pt : out) for (auto &tm : pt.terms) if (!tm.const_t) tm.offset += t;
      } else if (id == "Scale") {
        float sc = (float)Number();
        Expect(',');
        out = Parse();
        for (auto &pt : out) for (auto &tm : pt.terms) tm.scale *= sc;
      } else if (id == "ReplaceIndex") {
        out = Parse();
        Expect(',');
        std::string var = Ident();
        Expect(',');
        int val = (int)Number();
        if (var != "t" || val != 0) Fail("nnet3 descriptor: only ReplaceIndex(x, t, 0) is supported: " + s);
        for (auto &pt : out) for (auto &tm : pt.terms) { tm.const_t = true; tm.offset = 0; }
      } else if (id == "IfDefined") {
        out = Parse();
      } else {
        Fail("nnet3 descriptor: unsupported expression '" + id + "' in: " + s);
      }
      Expect(')');
      return out;
    }
    if (id.empty()) { p = save; Fail("nnet3 descriptor parse error in '" + s + "'"); }
    int n = net.FindNode(id);
    if (n < 0) Fail("nnet3 descriptor: unknown node '" + id + "' in: " + s);
    DescPart part;
    DescTerm t;
    t.node = n;
    part.terms.push_back(t);
    part.dim = node_dims[n];
    return {part};
  }
};

std::map<std::string, std::string> ParseConfigLine(const std::string &line, std::string *first) {
  // "component-node name=x component=y input=Append(a, b)" -> key/value; values run until the next " key=".
  std::map<std::string, std::string> kv;
  std::istringstream is(line);
  is >> *first;
  std::string rest;
  std::getline(is, rest);
  size_t i = 0;
  std::vector<std::pair<size_t, size_t>> keys;  // (key start, '=' pos)
  while (i < rest.size()) {
    if (std::isspace((unsigned char)rest[i])) { i++; continue; }
    size_t b = i;
    while (i < rest.size() && (std::isalnum((unsigned char)rest[i]) || rest[i] == '-' || rest[i] == '_')) i++;
    if (i < rest.size() && rest[i] == '=' && i > b && (b == 0 || std::isspace((unsigned char)rest[b - 1]))) {
      keys.emplace_back(b, i);
    }
    while (i < rest.size() && !std::isspace((unsigned char)rest[i])) i++;
  }
  for (size_t k = 0; k < keys.size(); k++) {
    size_t vb = keys[k].second + 1, ve = (k + 1 < keys.size()) ? keys[k + 1].first : rest.size();
    std::string v = rest.substr(vb, ve - vb);
    while (!v.empty() && std::isspace((unsigned char)v.back())) v.pop_back();
    kv[rest.substr(keys[k].first, keys[k].second - keys[k].first)] = v;
  }
  return kv;
}

int ComponentOutputDim(const Component &c, const std::string &name) {
  auto rows = [&](const char *k) -> int { auto it = c.m.find(k); return it == c.m.end() ? -1 : it->second.rows; };
  if (c.type == "AffineComponent" || c.type == "NaturalGradientAffineComponent" || c.type == "FixedAffineComponent" ||
      c.type == "TdnnComponent")
    return rows("<LinearParams>");
  if (c.type == "LinearComponent") return rows("<Params>");
  if (c.type == "NormalizeComponent") {
    int d = c.i.count("<InputDim>") ? c.Int("<InputDim>") : c.Int("<Dim>");
    bool add = c.f.count("<AddLogStddev>") && c.f.at("<AddLogStddev>")[0] != 0.0;
    return d + (add ? 1 : 0);
  }
  if (c.i.count("<Dim>")) return c.Int("<Dim>");
  if (c.type == "PerElementScaleComponent" || c.type == "FixedScaleComponent") {
    auto it = c.v.find(c.type == "FixedScaleComponent" ? "<Scales>" : "<Params>");
    if (it != c.v.end()) return (int)it->second.size();
  }
  if (c.type == "PerElementOffsetComponent") { auto it = c.v.find("<Offsets>"); if (it != c.v.end()) return (int)it->second.size(); }
  if (c.type == "FixedBiasComponent") { auto it = c.v.find("<Bias>"); if (it != c.v.end()) return (int)it->second.size(); }
  if (c.type == "ScaleAndOffsetComponent") { auto it = c.v.find("<Scales>"); if (it != c.v.end()) return (int)it->second.size(); }
  Fail("nnet3: cannot determine the output dimension of component '" + name + "' of type " + c.type);
}
}  // namespace

int Nnet::FindNode(const std::string &name) const {
  for (size_t i = 0; i < nodes.size(); i++) if (nodes[i].name == name) return (int)i;
  return -1;
}

void ReadNnetComponents(KaldiReader &r, std::vector<std::string> *names, std::vector<Component> *comps) {
  r.ExpectToken("<NumComponents>");
  int nc = r.ReadInt32();
  if (nc < 0 || nc >= 100000) Fail("bad <NumComponents>");
  names->resize(nc);
  comps->resize(nc);
  for (int c = 0; c < nc; c++) {
    r.ExpectToken("<ComponentName>");
    (*names)[c] = r.ReadToken();
    (*comps)[c] = ReadComponent(r);
  }
  r.ExpectToken("</Nnet3>");
}

void Nnet::Read(KaldiReader &r, int frames_per_chunk, int extra_left_context_initial, int frame_subsampling_factor) {
  r.ExpectToken("<Nnet3>");
  std::string line = r.ReadLine();
  if (!line.empty() && line.find_first_not_of(" \t") != std::string::npos) Fail("Expected newline in config file, got " + line);
  std::vector<std::string> cfg;
  while (true) {
    if (r.RawEof()) Fail(r.name() + ": EOF inside <Nnet3> config section");
    line = r.ReadLine();
    if (line.empty()) break;
    cfg.push_back(line);
  }
  ReadNnetComponents(r, &component_names, &components);
  {
    // CollapseModel + the rand() calls of the reference's set-up (nnet3_setup.h); RS_NO_COLLAPSE=1 keeps the layers as
    // written (A/B of the rounding difference; the rand() count is that of the reference either way)
    std::vector<std::string> names = component_names;
    std::vector<Component> comps = components;
    const char *e = TuneEnv("RS_NO_COLLAPSE");
    const bool keep_layers = e && e[0] == '1';
    Nnet3SetupResult su = Nnet3Setup(cfg, keep_layers ? &names : &component_names, keep_layers ? &comps : &components, frames_per_chunk,
                                     extra_left_context_initial, frame_subsampling_factor);
    setup_rand_calls = su.rand_calls;
    setup_rand_certain = su.rand_calls_certain;
    setup_rand_uncertain_why = su.uncertain_why;
    if (!keep_layers) cfg = su.config;
  }
  // first pass: create nodes (names + dims), second pass: descriptors
  std::vector<std::map<std::string, std::string>> kvs;
  std::vector<std::string> firsts;
  for (auto &l : cfg) {
    std::string first;
    auto kv = ParseConfigLine(l, &first);
    if (first == "component") continue;   // "component name=... type=..." lines do not occur in written models
    NnetNode n;
    n.name = kv["name"];
    if (first == "input-node") { n.kind = NnetNode::kInput; n.dim = std::stoi(kv["dim"]); }
    else if (first == "component-node") {
      n.kind = NnetNode::kComponent;
      auto it = std::find(component_names.begin(), component_names.end(), kv["component"]);
      if (it == component_names.end()) Fail("nnet3: component-node " + n.name + " refers to unknown component " + kv["component"]);
      n.component = (int)(it - component_names.begin());
      n.dim = ComponentOutputDim(components[n.component], kv["component"]);
    } else if (first == "output-node") n.kind = NnetNode::kOutput;
    else if (first == "dim-range-node") { n.kind = NnetNode::kDimRange; n.dim = std::stoi(kv["dim"]); n.range_offset = std::stoi(kv["dim-offset"]); }
    else Fail("nnet3: unsupported config line: " + l);
    nodes.push_back(n);
    kvs.push_back(kv);
    firsts.push_back(first);
  }
  // resolve descriptors; dims of output nodes depend on their inputs, so iterate until stable
  std::vector<int> dims(nodes.size());
  for (size_t i = 0; i < nodes.size(); i++) dims[i] = nodes[i].dim;
  for (size_t i = 0; i < nodes.size(); i++) {
    NnetNode &n = nodes[i];
    if (n.kind == NnetNode::kDimRange) {
      n.range_node = FindNode(kvs[i]["input-node"]);
      if (n.range_node < 0) Fail("nnet3: dim-range-node " + n.name + " has unknown input-node");
    }
  }
  for (size_t i = 0; i < nodes.size(); i++) {
    NnetNode &n = nodes[i];
    if (n.kind != NnetNode::kComponent && n.kind != NnetNode::kOutput) continue;
    DescParser dp(kvs[i]["input"], *this, dims);
    n.input.parts = dp.Parse();
    dp.Ws();
    if (dp.p != dp.s.size()) Fail("nnet3: trailing characters in descriptor: " + dp.s);
    n.input.dim = 0;
    for (auto &p : n.input.parts) n.input.dim += p.dim;
    if (n.kind == NnetNode::kOutput) { n.dim = n.input.dim; dims[i] = n.dim; }
  }
  int in = FindNode("input"), iv = FindNode("ivector"), out = FindNode("output");
  if (in < 0 || nodes[in].kind != NnetNode::kInput) Fail("nnet3: no input-node named 'input'");
  if (out < 0 || nodes[out].kind != NnetNode::kOutput) Fail("nnet3: no output-node named 'output'");
  input_dim = nodes[in].dim;
  ivector_dim = iv >= 0 ? nodes[iv].dim : 0;
  output_dim = nodes[out].dim;
}

// =============================================================================== nnet3 compile

namespace {
bool IsAffineLike(const std::string &t) {
  return t == "AffineComponent" || t == "NaturalGradientAffineComponent" || t == "FixedAffineComponent" ||
         t == "LinearComponent" || t == "TdnnComponent";
}
bool IsIdentity(const Component &c) {
  return c.type == "NoOpComponent" || c.type == "DropoutComponent" || c.type == "GeneralDropoutComponent" ||
         c.type == "DropoutMaskComponent" || c.type == "SpecAugmentTimeMaskComponent" || c.type == "ClipGradientComponent" ||
         c.type == "BackpropTruncationComponent";
}

// Elementwise component -> stages (empty = identity).  false if the type is not elementwise.
bool EltStagesFor(const Component &c, const std::string &name, std::vector<EltStage> *st) {
  st->clear();
  if (IsIdentity(c)) return true;
  if (c.type == "RectifiedLinearComponent") { EltStage s; s.kind = EltStage::kRelu; st->push_back(s); return true; }
  if (c.type == "LogSoftmaxComponent") { EltStage s; s.kind = EltStage::kLogSoftmax; st->push_back(s); return true; }
  if (c.type == "BatchNormComponent") {
    // test-mode scale/offset exactly as BatchNormComponent::Read + ComputeDerived
    // (nnet-normalize-component.cc:209-246,590-612), float arithmetic
    int dim = c.Int("<Dim>"), block = c.Int("<BlockDim>");
    float eps = (float)c.f.at("<Epsilon>")[0], rms = (float)c.f.at("<TargetRms>")[0];
    double count = c.f.at("<Count>")[0];
    const std::vector<float> &mean = c.v.at("<StatsMean>"), &var = c.v.at("<StatsVar>");
    if (count == 0.0) Fail("nnet3: BatchNormComponent '" + name + "' has no stats (count = 0); cannot run in test mode");
    if ((int)mean.size() != block || (int)var.size() != block) Fail("nnet3: BatchNormComponent '" + name + "' stats dim mismatch");
    EltStage s;
    s.kind = EltStage::kScaleOffset;
    s.scale.resize(dim);
    s.offset.resize(dim);
    std::vector<float> sc(block), of(block);
    for (int i = 0; i < block; i++) {
      // Read(): stats_sumsq = (var + mean*mean) * count ; stats_sum = mean * count   [CuVector<float> ops]
      float sumsq = var[i] + mean[i] * mean[i];
      float ssum = mean[i] * (float)count;
      sumsq = sumsq * (float)count;
      // ComputeDerived()
      float off = ssum * (float)(-1.0 / count);
      float scl = sumsq * (float)(1.0 / count);
      scl = scl + (-1.0f) * off * off;
      if (scl < 0.0f) scl = 0.0f;
      scl = scl + eps;
      scl = powf(scl, -0.5f);
      scl = scl * rms;
      off = off * scl;
      sc[i] = scl;
      of[i] = off;
    }
    for (int i = 0; i < dim; i++) { s.scale[i] = sc[i % block]; s.offset[i] = of[i % block]; }
    st->push_back(s);
    return true;
  }
  if (c.type == "NormalizeComponent") {
    if (c.f.count("<AddLogStddev>") && c.f.at("<AddLogStddev>")[0] != 0.0) Fail("nnet3: NormalizeComponent add-log-stddev=true is not supported");
    if (c.f.count("<BlockDim>")) {
      int d = c.i.count("<InputDim>") ? c.Int("<InputDim>") : c.Int("<Dim>");
      if (c.Int("<BlockDim>") != d) Fail("nnet3: NormalizeComponent with block-dim is not supported");
    }
    EltStage s;
    s.kind = EltStage::kNormalize;
    s.alpha = c.f.count("<TargetRms>") ? (float)c.f.at("<TargetRms>")[0] : 1.0f;
    st->push_back(s);
    return true;
  }
  if (c.type == "FixedScaleComponent" || c.type == "PerElementScaleComponent" || c.type == "NaturalGradientPerElementScaleComponent") {
    const std::vector<float> &v = c.v.at(c.type == "FixedScaleComponent" ? "<Scales>" : "<Params>");
    EltStage s; s.kind = EltStage::kScaleOffset; s.scale = v; s.offset.assign(v.size(), 0.0f);
    st->push_back(s);
    return true;
  }
  if (c.type == "FixedBiasComponent" || c.type == "PerElementOffsetComponent") {
    const std::vector<float> &v = c.v.at(c.type == "FixedBiasComponent" ? "<Bias>" : "<Offsets>");
    if (c.type == "PerElementOffsetComponent" && c.f.count("<UseNaturalGradient>") == 0 && c.i.count("<Dim>") && c.Int("<Dim>") != (int)v.size())
      Fail("nnet3: PerElementOffsetComponent with block structure is not supported");
    EltStage s; s.kind = EltStage::kScaleOffset; s.offset = v; s.scale.assign(v.size(), 1.0f);
    st->push_back(s);
    return true;
  }
  return false;
}
}  // namespace

void Nnet::Compile() {
  int out_node = FindNode("output");
  int in_node = FindNode("input"), iv_node = FindNode("ivector");
  const int N = (int)nodes.size();
  // reachable set + topological order (DFS post-order)
  std::vector<int> order, state(N, 0);
  std::function<void(int)> visit = [&](int n) {
    if (state[n] == 2) return;
    if (state[n] == 1) Fail("nnet3: recurrent networks are not supported (cycle at node " + nodes[n].name + ")");
    state[n] = 1;
    const NnetNode &nd = nodes[n];
    if (nd.kind == NnetNode::kDimRange) visit(nd.range_node);
    for (auto &p : nd.input.parts) for (auto &t : p.terms) visit(t.node);
    state[n] = 2;
    order.push_back(n);
  };
  visit(out_node);
  // consumer counts (only "plain" uses allow fusing)
  std::vector<int> consumers(N, 0);
  for (int n : order) {
    const NnetNode &nd = nodes[n];
    if (nd.kind == NnetNode::kDimRange) consumers[nd.range_node] += 2;  // never fuse through a dim-range
    for (auto &p : nd.input.parts) for (auto &t : p.terms) consumers[t.node]++;
  }
  // Tdnn components contribute their own time offsets on top of the descriptor's.
  auto tdnn_offsets = [&](const NnetNode &nd) -> std::vector<int> {
    if (nd.kind == NnetNode::kComponent && components[nd.component].type == "TdnnComponent") {
      const auto &c = components[nd.component];
      auto it = c.iv.find("<TimeOffsets>");
      if (it == c.iv.end() || it->second.empty()) Fail("nnet3: TdnnComponent without <TimeOffsets>");
      return std::vector<int>(it->second.begin(), it->second.end());
    }
    return {0};
  };
  // required extension per node: rows t in [-lext, T+rext)
  std::vector<int> lext(N, 0), rext(N, 0);
  for (auto it = order.rbegin(); it != order.rend(); ++it) {
    int n = *it;
    const NnetNode &nd = nodes[n];
    if (nd.kind == NnetNode::kDimRange) {
      lext[nd.range_node] = std::max(lext[nd.range_node], lext[n]);
      rext[nd.range_node] = std::max(rext[nd.range_node], rext[n]);
      continue;
    }
    std::vector<int> toffs = tdnn_offsets(nd);
    for (auto &p : nd.input.parts)
      for (auto &t : p.terms) {
        if (t.const_t) continue;
        for (int o2 : toffs) {
          int o = t.offset + o2;
          lext[t.node] = std::max(lext[t.node], lext[n] - o);
          rext[t.node] = std::max(rext[t.node], rext[n] + o);
        }
      }
  }
  left_context = lext[in_node];
  right_context = rext[in_node];

  // buffer assignment
  std::vector<int> node_buf(N, -1), node_col(N, 0);
  std::vector<int> node_op(N, -1);     // op producing the node's buffer (for fusing)
  ops.clear();
  bufs.clear();
  auto new_buf = [&](int dim, int l, int r) { BufferInfo b; b.dim = dim; b.lext = l; b.rext = r; bufs.push_back(b); return (int)bufs.size() - 1; };
  input_buf = new_buf(nodes[in_node].dim, lext[in_node], rext[in_node]);
  bufs[input_buf].is_input = true;
  node_buf[in_node] = input_buf;
  if (iv_node >= 0) node_buf[iv_node] = -2;  // special: iVector rows

  // materialise one descriptor part (a Sum of terms) as an eltwise op if it is not a plain reference
  auto term_src = [&](const DescTerm &t, int *buf, int *col) {
    if (node_buf[t.node] == -1) Fail("nnet3: internal error, node " + nodes[t.node].name + " used before it was computed");
    *buf = node_buf[t.node];
    *col = node_col[t.node];
  };

  for (int n : order) {
    NnetNode &nd = nodes[n];
    if (nd.kind == NnetNode::kInput) continue;
    if (nd.kind == NnetNode::kDimRange) {
      if (node_buf[nd.range_node] == -2) Fail("nnet3: dim-range-node over the iVector input is not supported");
      node_buf[n] = node_buf[nd.range_node];
      node_col[n] = node_col[nd.range_node] + nd.range_offset;
      continue;
    }
    const Component *comp = nd.kind == NnetNode::kComponent ? &components[nd.component] : nullptr;
    bool affine = comp && IsAffineLike(comp->type);
    std::vector<EltStage> stages;
    bool elt = !affine && (comp == nullptr || EltStagesFor(*comp, nd.name, &stages));
    if (!affine && !elt)
      Fail("nnet3: component type " + comp->type + " (node " + nd.name + ") is not supported by the HIP acoustic-model kernels");

    // ---- fuse an elementwise node into its producer when it is the only consumer of a plain reference
    bool plain = nd.input.parts.size() == 1 && nd.input.parts[0].terms.size() == 1;
    if (elt && plain) {
      const DescTerm &t = nd.input.parts[0].terms[0];
      if (!t.const_t && t.offset == 0 && t.scale == 1.0f && consumers[t.node] == 1 && node_op[t.node] >= 0 &&
          node_col[t.node] == 0 && nodes[t.node].kind != NnetNode::kDimRange) {
        LayerOp &op = ops[node_op[t.node]];
        bool ok = true;
        // a row-wise reduction can only be fused when the op's tile covers the whole row; keep those standalone
        for (auto &s : stages) if (s.kind == EltStage::kLogSoftmax || s.kind == EltStage::kNormalize) ok = false;
        if (ok) {
          op.stages.insert(op.stages.end(), stages.begin(), stages.end());
          op.name += "+" + nd.name;
          node_buf[n] = node_buf[t.node];
          node_col[n] = 0;
          node_op[n] = node_op[t.node];
          // the fused buffer must cover this node's extent as well (it does: same offsets), keep max
          bufs[node_buf[n]].lext = std::max(bufs[node_buf[n]].lext, lext[n]);
          bufs[node_buf[n]].rext = std::max(bufs[node_buf[n]].rext, rext[n]);
          continue;
        }
      }
    }

    LayerOp op;
    op.name = nd.name;
    op.out_dim = nd.dim;
    if (affine) {
      op.kind = LayerOp::kGemm;
      const MatF &W = comp->type == "LinearComponent" ? comp->m.at("<Params>") : comp->m.at("<LinearParams>");
      op.W = W;
      auto bit = comp->v.find("<BiasParams>");
      if (bit != comp->v.end() && !bit->second.empty()) op.bias = bit->second;
      std::vector<int> toffs = tdnn_offsets(nd);
      int wcol = 0;
      for (int o2 : toffs) {
        for (auto &p : nd.input.parts) {
          int sbuf, scol, soff;
          if (p.terms.size() == 1 && p.terms[0].scale == 1.0f) {
            const DescTerm &t = p.terms[0];
            term_src(t, &sbuf, &scol);
            soff = t.const_t ? 0 : t.offset;
            if (sbuf == -2) { sbuf = -1; soff = 0; }
            else if (t.const_t) Fail("nnet3: ReplaceIndex on a non-iVector node is not supported (node " + nd.name + ")");
          } else {
            // materialise the sum
            LayerOp sop;
            sop.kind = LayerOp::kEltwise;
            sop.name = nd.name + ".sum";
            sop.out_dim = p.dim;
            int l = 0, r = 0;
            for (int oo : toffs) { l = std::max(l, lext[n] - oo); r = std::max(r, rext[n] + oo); }
            sop.out_buf = new_buf(p.dim, std::max(l, 0), std::max(r, 0));
            for (auto &t : p.terms) {
              int b, c;
              term_src(t, &b, &c);
              if (b == -2 || t.const_t) Fail("nnet3: Sum() over the iVector input is not supported");
              sop.terms.push_back({b, c, t.offset, t.scale});
            }
            ops.push_back(sop);
            sbuf = sop.out_buf; scol = 0; soff = 0;
          }
          GemmSegment sg;
          sg.src_buf = sbuf; sg.src_col = scol; sg.ncols = p.dim; sg.offset = soff + (sbuf == -1 ? 0 : o2); sg.w_col = wcol;
          op.segs.push_back(sg);
          wcol += p.dim;
        }
      }
      if (wcol != W.cols) Fail("nnet3: component of node " + nd.name + " expects input dim " + std::to_string(W.cols) + " but its descriptor provides " + std::to_string(wcol));
      if (!op.bias.empty() && (int)op.bias.size() != W.rows) Fail("nnet3: bias dim mismatch in node " + nd.name);
    } else {
      op.kind = LayerOp::kEltwise;
      if (nd.input.parts.size() != 1) {
        // Append feeding an elementwise component / the output: copy each part into its column range
        Fail("nnet3: Append() feeding a non-affine component (node " + nd.name + ") is not supported");
      }
      for (auto &t : nd.input.parts[0].terms) {
        int b, c;
        term_src(t, &b, &c);
        if (b == -2 || t.const_t) Fail("nnet3: elementwise component over the iVector input is not supported");
        op.terms.push_back({b, c, t.offset, t.scale});
      }
      op.stages = stages;
      if (nd.kind == NnetNode::kOutput && op.terms.size() == 1 && op.terms[0].offset == 0 && op.terms[0].scale == 1.0f &&
          op.terms[0].src_col == 0 && bufs[op.terms[0].src_buf].dim == nd.dim && op.stages.empty()) {
        // output-node that simply names a buffer: alias it
        node_buf[n] = op.terms[0].src_buf;
        node_op[n] = -1;
        continue;
      }
    }
    op.out_buf = new_buf(nd.dim, lext[n], rext[n]);
    ops.push_back(op);
    node_buf[n] = op.out_buf;
    node_col[n] = 0;
    node_op[n] = (int)ops.size() - 1;
  }
  output_buf = node_buf[out_node];
  if (output_buf < 0 || node_col[out_node] != 0 || bufs[output_buf].dim != output_dim)
    Fail("nnet3: could not resolve the output node to a buffer");
  // ---- a two-term sum whose one term is the (otherwise unread) result of a GEMM becomes that GEMM's epilogue: the residual of a
  // factorised TDNN layer, NoOp(Sum(Scale(0.66, previous layer), dropout(batchnorm(relu(affine))))) -- as a separate pass over two
  // 1024-wide buffers it took as long as the layer's two GEMMs together (profiles/r06/tdnnf_notes.txt).  RS_FUSE_RESIDUAL=0 keeps
  // the elementwise op (tests compare the two bit for bit).
  {
    const char *e = std::getenv("RS_FUSE_RESIDUAL");
    const bool fuse = !(e && std::atoi(e) == 0);
    auto readers = [&](int buf) {
      int n = buf == output_buf ? 1 : 0;
      for (auto &op : ops) {
        for (auto &sg : op.segs) n += sg.src_buf == buf;
        for (auto &t : op.terms) n += t.src_buf == buf;
        n += op.res_buf == buf;
      }
      return n;
    };
    for (size_t ei = 0; fuse && ei < ops.size(); ei++) {
      LayerOp &eo = ops[ei];
      if (eo.kind != LayerOp::kEltwise || eo.terms.size() != 2 || !eo.stages.empty()) continue;
      for (int k = 0; k < 2; k++) {
        const SumTerm &t = eo.terms[k], &r = eo.terms[1 - k];
        if (t.scale != 1.0f || t.offset != 0 || r.offset != 0 || t.src_col != 0 || r.src_col != 0 || t.src_buf == r.src_buf) continue;
        if (bufs[t.src_buf].dim != eo.out_dim || bufs[r.src_buf].dim != eo.out_dim || readers(t.src_buf) != 1) continue;
        size_t gi = ops.size(), ri = 0;
        bool r_made = bufs[r.src_buf].is_input;
        for (size_t i = 0; i < ei; i++) {
          if (ops[i].out_buf == t.src_buf) gi = i;
          if (ops[i].out_buf == r.src_buf) { ri = i; r_made = true; }
        }
        if (gi == ops.size() || ops[gi].kind != LayerOp::kGemm || ops[gi].res_buf >= 0 || !r_made || (ri >= gi && !bufs[r.src_buf].is_input)) continue;
        for (auto &st : ops[gi].stages) if (st.kind == EltStage::kLogSoftmax || st.kind == EltStage::kNormalize) gi = ops.size();
        if (gi == ops.size()) continue;
        LayerOp &g = ops[gi];
        g.res_buf = r.src_buf;
        g.res_scale = r.scale;
        g.out_buf = eo.out_buf;
        g.name += "+" + eo.name;
        ops.erase(ops.begin() + ei);
        ei--;
        break;
      }
    }
  }
  // sanity: every source buffer must cover what its consumers read
  for (auto &op : ops) {
    const BufferInfo &ob = bufs[op.out_buf];
    auto check = [&](int sb, int off) {
      if (sb < 0) return;
      if (bufs[sb].lext < ob.lext - off || bufs[sb].rext < ob.rext + off)
        Fail("nnet3: internal error, buffer extents of op " + op.name + " are inconsistent");
    };
    for (auto &s : op.segs) check(s.src_buf, s.offset);
    for (auto &t : op.terms) check(t.src_buf, t.offset);
    check(op.res_buf, 0);
  }
}

// --frame-subsampling-factor f: the output is wanted at t = 0, f, 2 f, ... only (CreateComputationRequestInternal, nnet-compile-looped.cc:
// 111-128) and the reference's compiler computes of every layer just the rows something reads.  Same here, in the one form a TDNN
// stack needs: walking the ops backwards, the residues (t mod f) of every buffer that are read; a buffer read at residue 0 only is
// evaluated on every f-th row (chain models: everything above the last layer with offsets that are not multiples of f).
void Nnet::SetSubsampling(int factor) {
  for (auto &b : bufs) b.stride = 1;
  if (factor <= 1 || factor > 30 || output_buf < 0) return;
  std::vector<unsigned> need(bufs.size(), 0u);
  need[output_buf] = 1u;
  auto mod = [&](int a) { int r = a % factor; return r < 0 ? r + factor : r; };
  // (elementwise ops -- a TDNN-F layer's Sum(Scale(0.66, x), y), a row-wise component -- take row lists like the GEMMs (RunNnet), so
  // they propagate the residues they are read at like any other op.  Round 5 forced their buffers dense AFTER the walk, which left
  // a layer X that feeds a Sum dense while its own input W stayed strided: X's GEMM ran over all rows and read rows of W nobody
  // had written in that call.)
  for (auto it = ops.rbegin(); it != ops.rend(); ++it) {
    const unsigned out = need[it->out_buf];
    auto reads = [&](int src, int off) {
      if (src < 0) return;
      for (int r = 0; r < factor; r++) if (out >> r & 1u) need[src] |= 1u << mod(r + off);
    };
    for (auto &sg : it->segs) reads(sg.src_buf, sg.offset);
    for (auto &t : it->terms) reads(t.src_buf, t.offset);
    reads(it->res_buf, 0);
  }
  for (size_t b = 0; b < bufs.size(); b++) if (!bufs[b].is_input && need[b] == 1u) bufs[b].stride = factor;
}

void AcousticModel::Read(const std::string &final_mdl, int frames_per_chunk, int extra_left_context_initial, int frame_subsampling_factor) {
  KaldiReader r(final_mdl);
  trans.Read(r);
  nnet.Read(r, frames_per_chunk, extra_left_context_initial, frame_subsampling_factor);
  r.ExpectToken("<LeftContext>");
  r.ReadInt32();
  r.ExpectToken("<RightContext>");
  r.ReadInt32();
  r.ExpectToken("<Priors>");
  r.ReadVector(&nnet.priors);
  if (!nnet.priors.empty() && (int)nnet.priors.size() != nnet.output_dim) nnet.priors.clear();  // am-nnet-simple.cc:63-68
  if (trans.num_pdfs != nnet.output_dim)
    Fail(final_mdl + ": transition model has " + std::to_string(trans.num_pdfs) + " pdfs but the nnet output dim is " + std::to_string(nnet.output_dim));
  nnet.Compile();
  nnet.SetSubsampling(frame_subsampling_factor);
}

// =============================================================================== HCLG

namespace {
struct ByteReader {
  const std::string &b;
  size_t p = 0;
  const std::string &name;
  ByteReader(const std::string &bytes, const std::string &n) : b(bytes), name(n) {}
  template <typename T> T Get() {
    if (p + sizeof(T) > b.size()) Fail(name + ": truncated FST file");
    T v;
    std::memcpy(&v, b.data() + p, sizeof(T));
    p += sizeof(T);
    return v;
  }
  std::string Str() {
    int32_t n = Get<int32_t>();
    if (n < 0 || p + (size_t)n > b.size()) Fail(name + ": bad string in FST header");
    std::string s = b.substr(p, n);
    p += n;
    return s;
  }
  void SkipSymbolTable() {
    // openfst lib/symbol-table.cc SymbolTableImpl::Read: magic, name, available_key, size, then (symbol, key)*
    int32_t magic = Get<int32_t>();
    if (magic != 2125658996) Fail(name + ": bad symbol table magic");
    Str();
    Get<int64_t>();
    int64_t n = Get<int64_t>();
    for (int64_t i = 0; i < n; i++) { Str(); Get<int64_t>(); }
  }
};
}  // namespace

void Hclg::Read(const std::string &path) {
  std::string bytes = ReadFileBytes(path);
  ByteReader r(bytes, path);
  if (r.Get<int32_t>() != 2125659606) Fail("FstHeader::Read: Bad FST header: " + path);
  std::string fsttype = r.Str(), arctype = r.Str();
  int32_t version = r.Get<int32_t>(), flags = r.Get<int32_t>();
  r.Get<uint64_t>();  // properties
  int64_t st = r.Get<int64_t>(), ns = r.Get<int64_t>(), na = r.Get<int64_t>();
  if (arctype != "standard") Fail("FST with arc type " + arctype + " not supported.");   // kaldi-fst-io.cc:66-69
  if (flags & 1) r.SkipSymbolTable();
  if (flags & 2) r.SkipSymbolTable();
  if (ns <= 0 || st < 0 || st >= ns) Fail(path + ": empty FST or no start state");
  start = (int32_t)st;
  auto align = [&]() { while (r.p % 16) r.p++; };   // MappedFile::kArchAlignment
  if (fsttype == "const") {
    bool aligned = (version == 1) || (flags & 4);
    if (aligned) align();
    final_cost.resize(ns);
    arc_begin.resize(ns + 1);
    num_ieps.resize(ns);
    for (int64_t s = 0; s < ns; s++) {
      final_cost[s] = r.Get<float>();
      uint32_t pos = r.Get<uint32_t>(), narcs = r.Get<uint32_t>(), nie = r.Get<uint32_t>();
      r.Get<uint32_t>();
      arc_begin[s] = pos;
      num_ieps[s] = nie;
      if (s + 1 == ns) arc_begin[ns] = pos + narcs;
      else if (false) (void)narcs;
    }
    if (aligned) align();
    if (arc_begin[ns] != (uint64_t)na) Fail(path + ": ConstFst arc count mismatch");
    arcs.resize(na);
    if (r.p + (size_t)na * sizeof(FstArc) > bytes.size()) Fail(path + ": truncated ConstFst arcs");
    std::memcpy(arcs.data(), bytes.data() + r.p, (size_t)na * sizeof(FstArc));
    for (int64_t s = 0; s + 1 < ns; s++) if (arc_begin[s] > arc_begin[s + 1]) Fail(path + ": ConstFst states are not in arc order");
  } else if (fsttype == "vector") {
    final_cost.resize(ns);
    arc_begin.resize(ns + 1);
    num_ieps.assign(ns, 0);
    arcs.clear();
    for (int64_t s = 0; s < ns; s++) {
      final_cost[s] = r.Get<float>();
      int64_t n = r.Get<int64_t>();
      arc_begin[s] = (uint32_t)arcs.size();
      for (int64_t a = 0; a < n; a++) {
        FstArc arc;
        arc.ilabel = r.Get<int32_t>();
        arc.olabel = r.Get<int32_t>();
        arc.weight = r.Get<float>();
        arc.nextstate = r.Get<int32_t>();
        if (arc.ilabel == 0) num_ieps[s]++;
        arcs.push_back(arc);
      }
    }
    arc_begin[ns] = (uint32_t)arcs.size();
  } else {
    Fail("Reading FST: unsupported FST type: " + fsttype);   // kaldi-fst-io.cc:86-89
  }
  for (auto &a : arcs)
    if (a.nextstate < 0 || a.nextstate >= ns || a.ilabel < 0) Fail(path + ": arc with invalid next state or label");
}

std::vector<std::string> ReadWordsTxt(const std::string &path) {
  std::ifstream is(path);
  if (!is.good()) Fail("Could not read symbol table from file " + path);
  std::vector<std::string> out;
  std::string sym;
  long id;
  while (is >> sym >> id) {
    if (id < 0) continue;
    if ((long)out.size() <= id) out.resize(id + 1);
    out[id] = sym;
  }
  return out;
}

}  // namespace rs
