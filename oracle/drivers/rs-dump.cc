// TEST INFRASTRUCTURE ONLY.  Golden-vector dump driver: our own main() linked against the
// reference's Kaldi libraries (oracle/_ref/libkaldi_ref.so, built by oracle/build_ref.sh from the
// sources under /root/reference).  It drives the reference's own classes exactly the way
// online2-wav-nnet3-latgen-faster.cc:150-268 (mode "offline") and
// online2-cli-nnet3-decode-faster.cc:129-161 (mode "stream", 1024-sample ticks) do, and writes the
// intermediate values of the hot path as .npy files:
//
//   input.npy      T x C   nnet input features (OnlineNnet2FeaturePipeline::InputFeature)
//   ivector.npy    n x D   iVector handed to each nnet chunk (offline: one row)
//   chunk_tick.npy n       tick index at which each chunk was computed (stream mode)
//   loglikes.npy   T x P   DecodableNnetLoopedOnline::LogLikelihood for every frame / pdf
//   cmvn.npy, lda.npy, lda_norm.npy  (offline only) the iVector branch's intermediate features,
//                  rebuilt from the same reference classes OnlineIvectorFeature wires together
//                  (online2/online-ivector-feature.cc:412-438)
//
// Usage: rs-dump [kaldi options] <offline|stream|randpos|collapsed> <final.mdl> <wav> <out-dir>
//   dither <num-frames> <window> <out.npy>: the noise of the reference's Dither() in a fresh process
//   randpos: prints the number of rand() calls consumed before the first feature frame and exits (wav / out-dir unused)
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "base/kaldi-common.h"
#include "feat/feature-window.h"
#include "feat/wave-reader.h"
#include "nnet3/decodable-online-looped.h"
#include "nnet3/nnet-utils.h"
#include "decoder/lattice-faster-decoder.h"
#include "online2/online-endpoint.h"
#include "online2/online-nnet2-feature-pipeline.h"
#include "util/common-utils.h"

using namespace kaldi;

static void WriteNpy(const std::string &path, const Matrix<BaseFloat> &m) {
  std::string dict = "{'descr': '<f4', 'fortran_order': False, 'shape': (" +
                     std::to_string(m.NumRows()) + ", " + std::to_string(m.NumCols()) + "), }";
  size_t total = 10 + dict.size() + 1;
  size_t pad = (64 - total % 64) % 64;
  dict += std::string(pad, ' ') + "\n";
  std::ofstream os(path, std::ios::binary);
  os.write("\x93NUMPY\x01\x00", 8);
  uint16_t len = dict.size();
  os.write(reinterpret_cast<const char *>(&len), 2);
  os.write(dict.data(), dict.size());
  for (int32 r = 0; r < m.NumRows(); r++)
    os.write(reinterpret_cast<const char *>(m.RowData(r)), sizeof(BaseFloat) * m.NumCols());
}

// Number of rand() calls made so far in this process (which never calls srand() elsewhere): the position of the next
// two values in glibc's default-seed sequence.  Leaves the generator where it found it.
static long RandPosition() {
  const int r1 = rand(), r2 = rand();
  srand(1);
  long pos = 0;
  int a = rand(), b = rand();
  while (!(a == r1 && b == r2)) {
    a = b;
    b = rand();
    if (++pos > 200000000) return -1;
  }
  srand(1);
  for (long i = 0; i < pos; i++) (void)rand();
  return pos;
}

int main(int argc, char *argv[]) {
  try {
    ParseOptions po("rs-dump [options] <offline|stream> <final.mdl> <wav> <out-dir>");
    OnlineNnet2FeaturePipelineConfig feature_opts;
    nnet3::NnetSimpleLoopedComputationOptions decodable_opts;
    LatticeFasterDecoderConfig decoder_opts;   // registered so online.conf / reference argv parse
    OnlineEndpointConfig endpoint_opts;
    feature_opts.Register(&po);
    decodable_opts.Register(&po);
    decoder_opts.Register(&po);
    endpoint_opts.Register(&po);
    po.Read(argc, argv);
    if (po.NumArgs() != 4) { po.PrintUsage(); return 1; }
    std::string mode = po.GetArg(1), mdl = po.GetArg(2), wav = po.GetArg(3), out = po.GetArg(4);
    bool offline = (mode == "offline");
    if (mode == "dither") {
      // rs-dump dither <num-frames> <window> <out.npy>: what the reference's own Dither() (feat/feature-window.cc:90-98) adds,
      // for --dither=1, to the frames of a process that has not called rand() before (pins oracle/dither.c)
      const int32 T = std::atoi(mdl.c_str()), W = std::atoi(wav.c_str());
      Matrix<BaseFloat> m(T, W);
      for (int32 t = 0; t < T; t++) {
        SubVector<BaseFloat> row(m, t);
        Dither(&row, 1.0);
      }
      WriteNpy(out, m);
      return 0;
    }

    OnlineNnet2FeaturePipelineInfo feature_info(feature_opts);
    if (offline) {
      feature_info.ivector_extractor_info.use_most_recent_ivector = true;
      feature_info.ivector_extractor_info.greedy_ivector_extractor = true;
    }
    TransitionModel trans_model;
    nnet3::AmNnetSimple am_nnet;
    {
      bool binary;
      Input ki(mdl, &binary);
      trans_model.Read(ki.Stream(), binary);
      am_nnet.Read(ki.Stream(), binary);
      SetBatchnormTestMode(true, &(am_nnet.GetNnet()));
      SetDropoutTestMode(true, &(am_nnet.GetNnet()));
      nnet3::CollapseModel(nnet3::CollapseModelConfig(), &(am_nnet.GetNnet()));
    }
    if (mode == "collapsed") {   // the network's config lines after CollapseModel (what the graph builder will see)
      std::vector<std::string> lines;
      am_nnet.GetNnet().GetConfigLines(false, &lines);
      for (size_t i = 0; i < lines.size(); i++) std::printf("%s\n", lines[i].c_str());
      return 0;
    }
    nnet3::DecodableNnetSimpleLoopedInfo info(decodable_opts, &am_nnet);
    if (mode == "randpos") {
      // How many rand() calls the set-up both decoder binaries share (model read, CollapseModel, looped compilation:
      // nnet-computation-graph.cc:481-555, nnet-utils.cc:107, nnet-optimize-utils.cc:4654, AffineComponent::Init) has
      // consumed.  The dither of frame t is seeded by value number (this + t) of glibc's default-seed sequence
      // (feature-window.cc:95, kaldi-math.cc:59-70).
      std::printf("%ld\n", RandPosition());
      return 0;
    }

    WaveData wave_data;
    {
      bool binary;
      Input ki(wav, &binary);
      wave_data.Read(ki.Stream());
    }
    SubVector<BaseFloat> data(wave_data.Data(), 0);

    const long rand_pos_at_first_frame = RandPosition();
    OnlineNnet2FeaturePipeline pipeline(feature_info);
    nnet3::DecodableNnetLoopedOnline decodable(info, pipeline.InputFeature(), pipeline.IvectorFeature());
    int32 P = info.output_dim, chunk = info.frames_per_chunk;
    std::vector<Vector<BaseFloat> > loglike_rows, ivec_rows;
    std::vector<BaseFloat> chunk_tick;
    int32 done = 0, tick = 0;
    auto advance = [&]() {
      int32 ready = decodable.NumFramesReady();
      for (; done < ready; done++) {
        Vector<BaseFloat> row(P);
        for (int32 p = 0; p < P; p++) row(p) = decodable.LogLikelihood(done, p + 1);
        loglike_rows.push_back(row);
        if (done % (chunk / decodable_opts.frame_subsampling_factor) == 0 && pipeline.IvectorFeature() != NULL) {
          // the chunk holding frame 'done' has just been computed on this tick; re-query the
          // iVector the decodable used (decodable-online-looped.cc:186-194; idempotent).
          OnlineIvectorFeature *iv = pipeline.IvectorFeature();
          int32 most_recent = pipeline.InputFeature()->NumFramesReady() - 1,
                ivr = iv->NumFramesReady();
          Vector<BaseFloat> v(iv->Dim());
          if (ivr > 0) iv->GetFrame(std::min(most_recent, ivr - 1), &v);
          ivec_rows.push_back(v);
          chunk_tick.push_back(tick);
        }
      }
    };
    if (offline) {
      pipeline.AcceptWaveform(wave_data.SampFreq(), data);
      pipeline.InputFinished();
      advance();
    } else {
      const int32 kTick = 1024;  // online2-cli-nnet3-decode-faster.cc:37
      for (int32 off = 0; off < data.Dim(); off += kTick, tick++) {
        int32 n = std::min(kTick, data.Dim() - off);
        Vector<BaseFloat> part(n);
        // the CLI reads int16 from stdin and casts to float (:143-148)
        for (int32 i = 0; i < n; i++) part(i) = static_cast<BaseFloat>(static_cast<int16>(data(off + i)));
        pipeline.AcceptWaveform(16000.0, part);
        advance();
      }
      pipeline.InputFinished();
      advance();
    }
    int32 T = loglike_rows.size();
    Matrix<BaseFloat> ll(T, P);
    for (int32 t = 0; t < T; t++) ll.Row(t).CopyFromVec(loglike_rows[t]);
    WriteNpy(out + "/loglikes.npy", ll);
    {
      OnlineFeatureInterface *in = pipeline.InputFeature();
      int32 n = in->NumFramesReady();
      Matrix<BaseFloat> feats(n, in->Dim());
      for (int32 t = 0; t < n; t++) { SubVector<BaseFloat> r(feats, t); in->GetFrame(t, &r); }
      WriteNpy(out + "/input.npy", feats);
    }
    if (!ivec_rows.empty()) {
      Matrix<BaseFloat> iv(ivec_rows.size(), ivec_rows[0].Dim());
      for (size_t i = 0; i < ivec_rows.size(); i++) iv.Row(i).CopyFromVec(ivec_rows[i]);
      WriteNpy(out + "/ivector.npy", iv);
      Matrix<BaseFloat> ct(1, chunk_tick.size());
      for (size_t i = 0; i < chunk_tick.size(); i++) ct(0, i) = chunk_tick[i];
      WriteNpy(out + "/chunk_tick.npy", ct);
    }
    if (offline && feature_info.use_ivectors) {
      // Same wiring as OnlineIvectorFeature's constructor, on a fresh MFCC.
      const OnlineIvectorExtractionInfo &ii = feature_info.ivector_extractor_info;
      srand(1);   // our second pass must see the dither of the first: rewind rand() to where the first frame found it
      for (long i = 0; i < rand_pos_at_first_frame; i++) (void)rand();
      OnlineMfcc mfcc(feature_info.mfcc_opts);
      mfcc.AcceptWaveform(wave_data.SampFreq(), data);
      mfcc.InputFinished();
      OnlineCmvnState st(ii.global_cmvn_stats);
      OnlineCmvn cmvn(ii.cmvn_opts, st, &mfcc);
      OnlineSpliceFrames sp(ii.splice_opts, &mfcc), spn(ii.splice_opts, &cmvn);
      OnlineTransform lda(ii.lda_mat, &sp), ldan(ii.lda_mat, &spn);
      int32 n = mfcc.NumFramesReady();
      Matrix<BaseFloat> a(n, cmvn.Dim()), b(n, lda.Dim()), c(n, ldan.Dim());
      for (int32 t = 0; t < n; t++) {
        SubVector<BaseFloat> ra(a, t), rb(b, t), rc(c, t);
        cmvn.GetFrame(t, &ra);
        lda.GetFrame(t, &rb);
        ldan.GetFrame(t, &rc);
      }
      WriteNpy(out + "/cmvn.npy", a);
      WriteNpy(out + "/lda.npy", b);
      WriteNpy(out + "/lda_norm.npy", c);
    }
    std::fprintf(stderr, "rs-dump: %d frames, %d pdfs, chunk %d, L %d, R %d\n", T, P, chunk,
                 info.frames_left_context, info.frames_right_context);
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return -1;
  }
}
