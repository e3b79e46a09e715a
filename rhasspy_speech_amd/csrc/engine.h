// Host orchestration of the batched transcribe pipeline on one MI355X (one process per GPU).
#pragma once
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rhasspy_speech_hip.h"
#include "kernels.h"
#include "lattice.h"
#include "model.h"

namespace rs {

struct DeviceError : Error {
  explicit DeviceError(const std::string &m) : Error(m) {}
};
void HipCheck(hipError_t e, const char *what, const char *file, int line);
#define RS_HIP(x) ::rs::HipCheck((x), #x, __FILE__, __LINE__)

// Grow-only device arena; reset at the start of every batch (no hipMalloc inside the timed region once warm).
class DeviceArena {
 public:
  ~DeviceArena();
  void Reset() { used_ = 0; }
  void Reserve(size_t bytes, hipStream_t s);   // make sure capacity >= bytes (frees + reallocates if needed)
  void *Alloc(size_t bytes);                   // 256-byte aligned; throws if Reserve() was too small
  size_t capacity() const { return cap_; }
  template <typename T> T *AllocT(size_t n) { return static_cast<T *>(Alloc(n * sizeof(T))); }
 private:
  char *base_ = nullptr;
  size_t cap_ = 0, used_ = 0;
};

// Grow-only page-locked host staging: small H2D / D2H transfers of a decode call are truly asynchronous from it and never
// wait on a pageable-memory bounce buffer.  Contents stay valid until the next Reset().
class HostArena {
 public:
  ~HostArena();
  void Reset() { used_ = 0; cur_ = 0; }
  void *Alloc(size_t bytes);                   // 64-byte aligned; grows (a new block) when needed
  template <typename T> T *AllocT(size_t n) { return static_cast<T *>(Alloc(n * sizeof(T))); }
 private:
  std::vector<std::pair<char *, size_t>> blocks_;
  size_t cur_ = 0, used_ = 0;
};

struct Hypothesis {
  std::vector<int32_t> words;
  float graph_cost = 0, acoustic_cost = 0;
};
struct UttResult {
  int status = RS_OK;
  std::string error;
  int num_frames = 0;
  std::vector<Hypothesis> hyps;
  int64_t counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  std::vector<float> feats, ivector, loglikes;   // keep_intermediates only
  int feat_dim = 0, ivec_rows = 0, ivec_dim = 0, num_pdfs = 0;
  std::shared_ptr<CompactLat> clat;              // emit_lattice only
};
struct Result {
  std::vector<UttResult> utts;
  float timings[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

// Device-resident, immutable part of a model.
struct GemmPlan {        // one LayerOp of kind kGemm, uploaded
  const LayerOp *op = nullptr;
  float *d_W = nullptr, *d_bias = nullptr;
  void *d_W3 = nullptr;           // split-bf16 image of W for GemmKernelB3 (layers at least 192 columns wide)
  int k_pad = 0, n_pad = 0, n3 = 0;
  bool interleave = false;        // W3 k-steps alternate between the segments (see GemmKernelB3)
  std::vector<int> seg_k0;
  std::vector<std::pair<float *, float *>> d_stage;   // scale/offset vectors per stage
};

struct LatArcBuffer { void *d = nullptr; size_t cap = 0; };      // capacity in LatArc records

class Model {
 public:
  Model(const std::string &final_mdl, const std::string &hclg, const std::string &online_conf, const rs_decode_opts &opts);
  ~Model();
  void ToDevice();
  std::string Describe() const;
  // streaming = true reproduces online2-cli-nnet3-decode-faster: 1024-sample ticks, one iVector per nnet chunk
  // estimated from the frames available at the tick the chunk is computed on (warm-started CG).
  std::unique_ptr<Result> DecodeBatchDevice(const int16_t *d_pcm, const int64_t *sample_offsets, int n_utts, int nbest,
                                            float lat_scale, hipStream_t stream, bool streaming = false);
  std::unique_ptr<Result> DecodeBatchHost(const int16_t *const *pcm, const int32_t *n_samples, int n_utts, int nbest,
                                          float lat_scale, bool streaming = false);
  const rs_decode_opts &opts() const { return opts_; }
  const FeatureConfig &features() const { return fc_; }
  const AcousticModel &am() const { return am_; }
  const Hclg &hclg() const { return hclg_; }

 private:
  struct DecodeContext;
  void DecodeGroup(DecodeContext &cx, int gi, const int16_t *d_pcm, const int64_t *sample_offsets, int n_utts, int nbest, float lat_scale,
                   hipStream_t s, bool streaming, UttResult *out_utts, float *timings);
  template <typename T> T *Upload(const std::vector<T> &v);
  void *UploadBytes(const void *p, size_t bytes);
  void BuildGemmPlan(const LayerOp &op, GemmPlan *plan);

  rs_decode_opts opts_;
  FeatureConfig fc_;
  AcousticModel am_;
  Hclg hclg_;
  std::vector<int32_t> arc_ilabel_;   // original transition-ids (device arcs carry pdf+1)

  bool on_device_ = false;
  std::mutex mu_;                      // ToDevice
  std::vector<void *> owned_;         // persistent device allocations
  int max_groups_ = 1;                 // RS_SUBBATCHES=2: two concurrent utterance groups inside one call
  // Everything one decode call needs for itself.  A model owns a few of them so that calls from different host threads can
  // be in flight together: the search of one batch is latency-bound and leaves the device to the next batch's GEMMs.
  struct DecodeContext {
    hipStream_t stream = nullptr, stream2 = nullptr;
    hipStream_t stream_dec = nullptr;      // the search of a time slab runs here while the next slab's output layer runs on `stream`
    hipEvent_t slab_ev[9] = {};
    DeviceArena arena[2];                  // one per concurrent utterance group
    HostArena host_arena[2];
    LatArcBuffer lat_arcs[2];              // lattice arc output of LatticeKernel, grow-only, one per utterance group
    int16_t *h_pcm_pinned = nullptr;       // pinned staging for host-buffer batches
    size_t h_pcm_cap = 0;
    int16_t *d_pcm = nullptr;
    int active_groups = 1;
    bool busy = false;
  };
  std::vector<std::unique_ptr<DecodeContext>> ctx_;
  std::mutex ctx_mu_;
  std::condition_variable ctx_cv_;
  DecodeContext *AcquireContext();
  void ReleaseContext(DecodeContext *cx);
  bool OthersInFlight();
  std::unique_ptr<Result> DecodeInContext(DecodeContext &cx, const int16_t *d_pcm, const int64_t *sample_offsets, int n_utts, int nbest,
                                          float lat_scale, hipStream_t user_stream, bool streaming);
  std::vector<int> pdf_remap_;     // prune_output_pdfs: pdf id -> column of the pruned output layer (-1 = never read)
  int pruned_from_ = 0;            // number of pdfs before pruning (0 = not pruned)
  void PruneOutputLayer();

  MfccDev mfcc_dev_{};
  CmvnDev cmvn_iv_dev_{}, cmvn_nnet_dev_{};
  IvecDev ivec_dev_{};
  LayerOp lda_op_;                    // splice + LDA of the iVector branch, as a segmented GEMM
  GemmPlan lda_plan_;
  std::vector<GemmPlan> gemm_plans_;  // indexed like am_.nnet.ops (unused entries for eltwise ops)
  float *d_log_priors_ = nullptr;
  HclgDev hclg_dev_{};
  RevGraphDev rev_dev_{};
  RegGraphDev reg_dev_{};
  int decoder_choice_ = 0;      // RS_DECODER=reg|dense|sparse forces a kernel variant (tests); 0 = automatic
  bool dense_ok_ = false;
  bool force_sparse_ = false;   // RS_FORCE_SPARSE_DECODER=1: always use the general (token-list) kernel
  int L_ = 0, R_ = 0;
};

}  // namespace rs

struct rs_model { std::unique_ptr<rs::Model> m; };
struct rs_result { std::unique_ptr<rs::Result> r; };
struct rs_stream { rs_model *model = nullptr; std::vector<int16_t> pcm; bool finished = false; };
