// Host orchestration of the batched transcribe pipeline on one MI355X (one process per GPU).
#pragma once
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <atomic>
#include <cstdlib>
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
  void *d_W3 = nullptr;           // split-fp16 image of W for GemmKernelB3 (layers at least 192 columns wide)
  void *d_W3I = nullptr;          // the same for GemmKernelB3I (every source a frame buffer on a k-step boundary)
  float *d_w3_inv_scale = nullptr;   // inverse of the images' column scales (n3 floats)
  bool res_by_image = false;      // LayerOp::res_buf is read through its operand image (GemmDev::res_img) when the call runs on the image-fed kernels
  int k_pad = 0, n_pad = 0, n3 = 0;
  bool interleave = false;        // W3 k-steps alternate between the segments (see GemmKernelB3)
  std::vector<int> seg_k0;
  std::vector<std::pair<float *, float *>> d_stage;   // scale/offset vectors per stage
};

struct LatArcBuffer { void *d = nullptr; size_t cap = 0; };      // capacity in LatArc records

// HIP events around the stages of a call, on the stream the kernels are launched on
struct Timer {
  hipEvent_t ev[8];
  int n = 0;
  hipStream_t s;
  explicit Timer(hipStream_t st) : s(st) { for (auto &e : ev) (void)hipEventCreate(&e); }
  ~Timer() { for (auto &e : ev) (void)hipEventDestroy(e); }
  void Mark() { if (n < 8) (void)hipEventRecord(ev[n++], s); }
  void Reset() { n = 0; }
  float Ms(int a, int b) { float ms = 0; if (a < n && b < n) (void)hipEventElapsedTime(&ms, ev[a], ev[b]); return ms; }
};

struct StreamPool;                                                 // stream.cc
void SampleGemmMode(bool exact, int *ovf_dev);                   // engine.cc: which layer GEMMs this thread's launches use, and whose range flag they raise
struct StreamPoolDeleter { void operator()(StreamPool *p) const; };

// Row lists of a batch for the layers that need fewer rows than the full halo: entry (lext, rext) lists, utterance after
// utterance, the physical rows of t in [-lext, T + rext) (kernels.h row layout).  (0, 0) = the real frames only.
struct RowMaps {
  // span128: the physical rows any 128 consecutive rows of the list reach over (GemmDev::row_map_span128), 0 = not known
  // stride: the list holds the rows t = 0 mod stride of [-lext, T + rext) (BufferInfo::stride)
  struct Entry { int lext, rext; const int *rows; int count; int span128 = 0; int stride = 1; int span160 = 0; };
  std::vector<Entry> maps;
  const Entry *Find(int lext, int rext, int stride = 1) const {
    for (auto &e : maps) if (e.lext == lext && e.rext == rext && e.stride == stride) return &e;
    return nullptr;
  }
};

// Which search kernel a call runs and its work buffers (engine.cc: PlanSearch / AllocSearch / LaunchSearch / CollectResults).
struct SearchPlan {
  bool unscale = false, want_lattice = false, use_reg = false, use_dense = false, use_hash = false;
  bool reg_lattice = false;      // n-best / lattice call on a grammar graph: register-resident search + its rows turned into token lists
  int S = 0, tok_cap = 0, max_words = 1024, maxT = 0, n_utts = 0;
  DecodeOptsDev dopts{};
  DecodeWork w{};
  DenseWork dw{};
};

class Model {
 public:
  Model(const std::string &final_mdl, const std::string &hclg, const std::string &online_conf, const rs_decode_opts &opts);
  ~Model();
  void ToDevice();
  // The online iVector estimator over K chunks of n streams / utterances in four launches (ivector_kernels.hip: IvecChainKernel);
  // d_fb / d_fe / d_or / d_ac: [K][n] frame ranges, iVector rows, "has new frames"; state rows slot[u] (null: u) of lin / quad / numf / x
  static size_t IvecChunkChainBytes(const IvecDev &iv, int n, int K);
  void IvecChunkChain(DeviceArena &arena, const BatchGeom &g, int n, int K, const float *stats_feats, int ld_l, const int *post_idx, const float *post_w,
                      const int *d_fb, const int *d_fe, const int *d_or, const int *d_ac, double *lin, double *quad, double *numf, double *x, const int *slot,
                      float *ivec_out, int ld_i, hipStream_t s) const;
  void ResolveDecoderOptions();      // opts_ := command line (rs_decode_opts) over online.conf over the reference's defaults
  std::string Describe() const;
  // streaming = true reproduces online2-cli-nnet3-decode-faster: 1024-sample ticks, one iVector per nnet chunk
  // estimated from the frames available at the tick the chunk is computed on (warm-started CG).
  std::unique_ptr<Result> DecodeBatchDevice(const int16_t *d_pcm, const int64_t *sample_offsets, int n_utts, int nbest,
                                            float lat_scale, hipStream_t stream, bool streaming = false);
  std::unique_ptr<Result> DecodeBatchHost(const int16_t *const *pcm, const int32_t *n_samples, int n_utts, int nbest,
                                          float lat_scale, bool streaming = false);
  const rs_decode_opts &opts() const { return opts_; }
  // rs_decode_opts.exact_token_order, RS_EXACT_ORDER overriding (read per call: tests compare the two searches on one model)
  bool ExactOrder() const { const char *e = std::getenv("RS_EXACT_ORDER"); return e ? std::atoi(e) != 0 : opts_.exact_token_order != 0; }
  const FeatureConfig &features() const { return fc_; }
  const AcousticModel &am() const { return am_; }
  const Hclg &hclg() const { return hclg_; }
  // Streams (stream.cc): a stream owns a slot and a row range of the model's pool from open to close; advance runs the
  // device work the accepted audio makes possible, batched over the given streams; final = end of input, results to res.
  void StreamOpen(rs_stream *st);
  void StreamClose(rs_stream *st);
  void StreamsAdvance(rs_stream *const *streams, int n, bool final, int nbest, float lat_scale, Result *res);
  void StreamsAdvanceLocked(rs_stream *const *streams, int n, bool final, int nbest, float lat_scale, Result *res);      // pool_mu_ held
  void StreamsPoisonAll(const std::string &why = std::string());          // pool_mu_ held

 private:
  struct DecodeContext;
  void DecodeGroup(DecodeContext &cx, int gi, const int16_t *d_pcm, const int64_t *sample_offsets, int n_utts, int nbest, float lat_scale,
                   hipStream_t s, bool streaming, UttResult *out_utts, float *timings);
  template <typename T> T *Upload(const std::vector<T> &v);
  void *UploadBytes(const void *p, size_t bytes);
  void BuildGemmPlan(const LayerOp &op, GemmPlan *plan);
  // imgs (may be null): operand images of the frame buffers (kernels.h: ActImage), indexed like src; out_buf = index of the
  // buffer `out` belongs to (-1: none of them)
  GemmDev MakeGemm(const GemmPlan &pl, const std::vector<float *> &src, const std::vector<int> &src_ld, float *ivec, int ivec_ld, float *out, int ldo,
                   int share, const std::vector<ActImage> *imgs = nullptr, int out_buf = -1) const;
  // operand images for the buffers the split-bf16 GEMM reads as images: bytes needed / allocation from a call's arena
  size_t ImageBytes(int rows) const;
  std::vector<ActImage> AllocImages(DeviceArena &arena, int rows) const;
  std::vector<char> buf_image_, buf_f32_;      // per nnet buffer: has an operand image / is (also) read as plain floats
  size_t PlanSearch(int n_utts, int maxT, int nbest, float lat_scale, SearchPlan *sp) const;
  void AllocSearch(SearchPlan *sp, DeviceArena &arena, hipStream_t s, bool pooled_frames = false) const;
  void LaunchSearch(SearchPlan *sp, DeviceArena &arena, const BatchGeom &g, const float *ll, int ll_ld, hipStream_t s) const;
  void CollectResults(SearchPlan &sp, DecodeContext &cx, int gi, const BatchGeom &g, const int *T, const float *ll, int ll_ld, int nbest,
                      float lat_scale, hipStream_t s, UttResult *out_utts, float *timings);
  void RunNnet(const std::vector<float *> &bufp, const std::vector<int> &buf_ld, float *d_ivec, int ld_i, const int *d_row_ivec, int rows,
               const RowMaps &row_maps, int share, size_t op_begin, size_t op_end, hipStream_t s,
               const std::vector<ActImage> *imgs = nullptr) const;

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
    hipEvent_t stage_ev[3] = {};           // end of this call's feature + iVector stage / of its acoustic-model stage / of its sample upload (StageChain)
    int *gemm_ovf = nullptr, *gemm_ovf_dev = nullptr;      // pinned + mapped word and the device's address of it (see exact_gemm_)
    hipEvent_t split_ev = nullptr;         // what `stream` held when a call split into two utterance groups (the second group's stream waits for it)
#ifndef RS_CONTEXT_SETS
#define RS_CONTEXT_SETS 4
#endif
    static constexpr int kSets = RS_CONTEXT_SETS;
    DeviceArena arena[kSets];              // one per concurrent utterance group (batch calls use two; stream advances rotate over all)
    HostArena host_arena[kSets];
    LatArcBuffer lat_arcs[kSets];          // lattice arc output of LatticeKernel, grow-only, one per utterance group
    int16_t *h_pcm_pinned = nullptr;       // pinned staging for host-buffer batches
    size_t h_pcm_cap = 0;
    int16_t *d_pcm = nullptr;
    int active_groups = 1;
    bool force_exact = false;              // this call is the repetition of one that left the split-fp16 kernels' range
    bool busy = false;
  };
  std::vector<std::unique_ptr<DecodeContext>> ctx_;
  std::mutex ctx_mu_;
  // Calls in flight on one model pass through the feature + iVector stage and through the acoustic-model stage ONE AFTER THE
  // OTHER (each stage's kernels fill the device; two calls in the same stage only slow each other down), so calls that arrive
  // at the same moment come out staggered -- the first after one call's latency, not all of them after four -- and the pipeline
  // of stages is full from the second call on.  Device-side only: the next call's stream waits for the event the previous
  // call recorded behind its stage; the host holds the stage's mutex just while it enqueues.  RS_STAGE_CHAIN=0 switches it off.
  std::mutex stage_mu_[3];
  hipEvent_t stage_tail_[3] = {nullptr, nullptr, nullptr};      // [2]: the copies of a call's samples to the device (DecodeBatchHost)
  std::condition_variable ctx_cv_;
  DecodeContext *AcquireContext();
  void ReleaseContext(DecodeContext *cx);
  bool OthersInFlight();
  std::unique_ptr<Result> DecodeInContext(DecodeContext &cx, const int16_t *d_pcm, const int64_t *sample_offsets, int n_utts, int nbest,
                                          float lat_scale, hipStream_t user_stream, bool streaming);
  friend struct StreamPool;
  std::unique_ptr<StreamPool, StreamPoolDeleter> pool_;
  std::unique_ptr<DecodeContext> stream_ctx_;
  std::mutex pool_mu_;             // pool bookkeeping and advances of this model, one at a time
  StreamPool *Pool();
  void StreamsDrain(StreamPool *p, float *extra);
  void IssuerSync(StreamPool *p);     // everything handed to the pool's issuing thread has been queued; ITS failure poisons the open streams, then rethrows
  void StreamGrow(rs_stream *st, int need_frames);
  // The split-fp16 layer GEMMs carry activations below 65520 in magnitude (nnet_gemm_b3.hip).  A kernel that meets a larger
  // one sets the flag of the decode context it runs for (DecodeContext::gemm_ovf: host memory the device writes to); a batch
  // call that finds it set after its wait repeats itself on the exact-FP32 kernels (a model whose calls keep doing that changes
  // to them for good); a stream advance cannot be repeated: its call fails and the model changes kernels at once.
  std::atomic<bool> exact_gemm_{false};
  std::atomic<int> range_retries_{0};     // batch calls repeated so far; the third makes the change permanent
  std::atomic<int> precision_retries_{0}; // ... of them, those repeated because an operand row was too small for the split (GemmDev::ovf[1])
  struct RangeRetry {};             // thrown by a batch call that has to be repeated on the exact kernels
  void StreamsCheckRange();         // the same for stream advances (stream.cc)
  bool CheckGemmRange(DecodeContext &cx);      // after a wait: true if this call ran on the split-fp16 kernels and one of them overflowed
  // Guard rows of the frame buffers / operand images of a call: layers evaluated on all rows read up to their context beyond
  // the first and last row; what they compute from those rows is never used, but it passes through the range check above.
  void ZeroGuards(const std::vector<float *> &bufp, const std::vector<int> &buf_ld, int rows, const std::vector<ActImage> *imgs, hipStream_t s) const;
  std::vector<int> pdf_remap_;     // prune_output_pdfs: pdf id -> column of the pruned output layer (-1 = never read)
  int pruned_from_ = 0;            // number of pdfs before pruning (0 = not pruned)
  void PruneOutputLayer();

  MfccDev mfcc_dev_{};
  // Dither noise of frames [0, dither_frames_) on the device (nnet3_setup.h: the reference's draws for frame t are a constant of
  // the model).  Grows on demand by doubling; the table a growth step supersedes stays alive until the next step (launches in flight may still read it).
  MfccDev MfccWithDither(int frames);
  std::mutex dither_mu_;           // guards d_dither_ / dither_frames_ (read, publish)
  std::mutex dither_grow_mu_;      // one thread grows the table at a time
  const float *d_dither_ = nullptr;
  int dither_frames_ = 0;
  long dither_rand_calls_ = 0;
  std::vector<void *> dither_bufs_;
  CmvnDev cmvn_iv_dev_{}, cmvn_nnet_dev_{};
  IvecDev ivec_dev_{};
  LayerOp lda_op_;                    // splice + LDA of the iVector branch, as a segmented GEMM
  GemmPlan lda_plan_;
  // The same as ONE segment: with a row pitch equal to the feature dim, the spliced vector of frame t is the contiguous run of
  // floats from row t - left onwards (rows overlap), so the per-frame segments' padding to the k-tile (40 -> 64) disappears
  LayerOp lda_op1_;
  GemmPlan lda_plan1_;
  const GemmPlan &LdaPlan(int ld) const { return (lda_plan1_.op && ld == fc_.mfcc.nceps) ? lda_plan1_ : lda_plan_; }
  std::vector<GemmPlan> gemm_plans_;  // indexed like am_.nnet.ops (unused entries for eltwise ops)
  float *d_log_priors_ = nullptr;
  HclgDev hclg_dev_{};
  bool hclg_has_eps_ = false;      // the graph has input-epsilon arcs (DenseLatticeKernel skips its closure pass otherwise)
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
struct rs_stream {
  rs_model *model = nullptr;
  bool finished = false, open = false;
  std::atomic<bool> failed{false}; // an rs_streams_advance over this stream threw: schedule and device rows disagree, only close is allowed
  std::string fail_why;            // ... and, when the failure surfaced in another call (an advance's deferred work), its message (written before `failed`)
  bool keep_pcm = false;           // RS_STREAM_BATCH=1: keep every sample and replay the stream as one batch at finish (cross-check path)
  std::vector<int16_t> pcm;        // samples from absolute index pcm_start on (the ones no complete frame has consumed yet)
  long pcm_start = 0, n_samples = 0;
  int slot = -1, row0 = 0, cap = 0;   // the stream's slot and frame-row range in the model's pool
  int frames_mfcc = 0;             // MFCC (and CMVN) frames produced
  long ticks_done = 0;             // 1024-sample ticks the chunk schedule has seen
  int chunks_sched = 0;            // nnet chunks computed
  int stats_done = 0;              // frames accumulated into the iVector statistics
  int ll_done = 0;                 // frames with log-likelihoods in the pool
  int frames_decoded = 0;          // frames the incremental search has consumed
  bool dec_started = false;
};
