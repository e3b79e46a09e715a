// Incremental streaming decode: rs_stream_accept / rs_streams_advance / rs_streams_finish.
//
// Reference: online2-cli-nnet3-decode-faster.cc:129-170 reads stdin in 1024-sample ticks and, per tick, AcceptWaveform +
// AdvanceDecoding: MFCC frames are produced as their samples arrive (online-feature.cc:131-204), a nnet chunk of 24 frames
// is computed on the first tick at which 24 (k + 1) + R feature frames exist (decodable-online-looped.cc:56-84), with the
// iVector estimated from the frames available on that tick (online-ivector-feature.cc:248-275; CG warm-started from the
// previous estimate), and the search consumes the new log-likelihood rows.  After stdin closes, the last frames are flushed
// (right context = copies of the last frame) and the lattice is finalised.
//
// Here the same work is done per rs_streams_advance call, BATCHED over the streams of the call: which chunk becomes
// computable on which tick -- and therefore which frames its iVector has seen -- is a pure function of the number of samples
// accepted so far, so it does not matter how many ticks' worth of audio one advance covers: the results are those of the
// tick-by-tick reference (and equal, bit for bit, to the batch replay of DecodeGroup(streaming = true), which the tests use as
// a cross-check).  What persists between advances lives in a per-model POOL in HBM:
//   rows (one per frame; a stream owns a contiguous range):  raw MFCCs, CMVN'd MFCCs, log-likelihoods, and -- for graphs the
//       register-resident search handles -- back-pointer rows and per-frame cutoffs;
//   slots (one per stream): iVector estimator state (linear / quadratic terms, frame count, CG solution), CMVN running sums,
//       the search's parked token costs and counters;  per-chunk iVectors.
// Everything else is transient and comes from the arena of the advance.  One advance =
//   1. new samples -> device; MFCC of the frames they complete, straight into the stream's pool rows;
//   2. CMVN resumed from the parked sums (iVector branch; nnet branch if the model has one);
//   3. for the streams that got new chunks: splice + LDA + UBM posteriors over the not-yet-accumulated frames (context gathered
//      from the pool with the reference's edge clamping), then chunk by chunk accumulate -> derived terms -> CG solve;
//   4. the acoustic model over [first new frame - left context, last new frame + right context) of every such stream, gathered
//      into the batch layout of kernels.h, output rows scattered into the pool;
//   5. the search resumed over the new rows (RegDecodeKernel windows); graphs it cannot hold are searched at finish.
// rs_streams_finish = one last advance with end-of-input semantics, then traceback / lattice / result records: its device
// work is one chunk's worth when the streams were advanced as the audio came in.
#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <thread>
#include "env.h"

#include "engine.h"
#include <cstdio>

namespace rs {

// A second host thread for the second half of an advance.  Issuing an advance's ~35 launches is the host's largest share of a
// streams step (profiles/micro/streams_trace.sh: 7 ms of 21.5 per 64 x 30 s, beside 2.6 ms accepting samples, 1.5 planning, 1.3
// uploading, 2.7 finishing); the acoustic model's and the search's launches (stage B / C: queues q and qc) need nothing the caller's
// thread computes afterwards, so they are handed to this thread as a closure and the caller goes on to its bookkeeping and the next
// advance's plan and stage A (queues qa, qi).  Jobs run in submission order (the order on q / qc is the advances' order); whoever
// is about to wait for an advance's events, to queue work on q / qc itself, or to tear the pool down drains the thread first; an
// exception of a job surfaces at that drain.
struct StreamIssuer {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv, cv_done;
  std::deque<std::function<void()>> jobs;
  long submitted = 0, completed = 0;
  bool stop = false, started = false;
  std::exception_ptr err;
  int device = 0;
  void Loop() {
    (void)hipSetDevice(device);
    for (;;) {
      std::function<void()> job;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || !jobs.empty(); });
        if (jobs.empty()) return;
        job = std::move(jobs.front());
        jobs.pop_front();
      }
      std::exception_ptr e;
      try { job(); } catch (...) { e = std::current_exception(); }
      {
        std::lock_guard<std::mutex> lk(mu);
        if (e && !err) err = e;
        completed++;
      }
      cv_done.notify_all();
    }
  }
  void Submit(std::function<void()> f) {
    {
      std::lock_guard<std::mutex> lk(mu);
      if (!started) { started = true; th = std::thread([this] { Loop(); }); }
      jobs.push_back(std::move(f));
      submitted++;
    }
    cv.notify_one();
  }
  // everything submitted has been issued; a job's exception is rethrown here (once)
  void Drain(bool rethrow = true) {
    std::exception_ptr e;
    {
      std::unique_lock<std::mutex> lk(mu);
      cv_done.wait(lk, [&] { return completed == submitted; });
      e = err;
      err = nullptr;
    }
    if (e && rethrow) std::rethrow_exception(e);
  }
  ~StreamIssuer() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv.notify_all();
    if (th.joinable()) th.join();
  }
};

struct StreamPool {
  StreamIssuer issuer;
  bool use_issuer = true;        // RS_STREAM_ISSUER=0: the caller's thread issues everything (A/B: profiles/micro)
  int min_ticks = 16;            // an advance call with fewer new 1024-sample ticks than this on every stream is coalesced into the next (rs_decode_opts.stream_min_ticks; RS_STREAM_MIN_TICKS overrides)
  // Queues, so that consecutive advances overlap on the device: `qa` runs an advance's features and UBM posteriors, `qi` (below)
  // its iVector steps, `q` its acoustic model (behind events of both), `qc` its search.  Stage A of advance n + 1 touches rows and slots stage B of advance n
  // does not (new frames / new chunks vs. the ones already scheduled), so the only ordering between them is per queue.
  hipStream_t q = nullptr, qa = nullptr, qc = nullptr;      // qc: the search of an advance (behind q's event), under the next advance's acoustic model
#ifndef RS_STREAM_DEPTH
#define RS_STREAM_DEPTH 4      // (2 / 3 / 4: 35.2 / 27.3 / 24.5 ms per 64 x 30 s step, profiles/micro/stream_depth.sh; at 4 the host no longer waits for the device)
#endif
  static constexpr int kDepth = RS_STREAM_DEPTH;                               // advances in flight at most <= arena / staging sets (DecodeContext::kSets)
  hipEvent_t ev_a[kDepth] = {}, ev_b[kDepth] = {}, ev_done[kDepth] = {};       // per set: stage A issued / log-likelihoods issued / the advance finished
  // qi: the iVector steps of an advance (estimator state in, one accumulate / products / solve per new chunk, state out), behind
  // qa's event at the UBM posteriors.  The chain is the longest sequential piece of an advance (~300 us of dependent small kernels);
  // on its own queue the next advance's features and posteriors run beside it instead of behind it.
  hipStream_t qi = nullptr;
  hipEvent_t ev_f[kDepth] = {}, ev_i[kDepth] = {};                             // per set: posteriors issued (qa) / iVectors issued (qi)
  std::unique_ptr<Timer> tm_i[kDepth];
  bool pending[kDepth] = {};                                                   // an advance that used this set may still run
  std::vector<rs_stream *> open_streams;                                       // every stream that holds a slot (a device failure nobody can attribute poisons them all)
  std::unique_ptr<Timer> tm_a[kDepth], tm_b[kDepth], tm_c[kDepth];
  // device time of set `par`'s advance into the pool's totals (its done event has been waited for)
  // (device time is SAMPLED: ten event records per advance were a fifth of an advance's issue time on the host, so only every
  // kTimedEvery-th advance -- and every finishing one -- carries them and counts kTimedEvery-fold)
  static constexpr int kTimedEvery = 4;
  float time_weight[kDepth] = {};
  void Account(int par, float *extra) {
    if (time_weight[par] == 0.f) return;
    const float ms[4] = {tm_a[par]->Ms(0, 1), tm_a[par]->Ms(1, 2) + tm_i[par]->Ms(0, 1), tm_b[par]->Ms(0, 1), tm_c[par]->Ms(0, 1)};
    for (int j = 0; j < 4; j++) { stage_ms[j + 1] += ms[j] * time_weight[par]; if (extra) extra[j] += ms[j]; }
  }
  long n_adv = 0;
  bool sync_each = false;        // RS_STREAM_SYNC=1: wait for every advance before returning (the behaviour before the two queues)
  void *cx = nullptr;            // Model::DecodeContext with the arenas / staging of the advances (never handed to batch calls)
  int rows = 0;                  // capacity in frame rows
  int chunk = 24, ld_c = 0, ld_ll = 0, ld_i = 0, S = 0;
  bool reg = false;              // the register-resident search runs incrementally
  float *raw = nullptr, *cm = nullptr, *nn_in = nullptr, *ll = nullptr, *finfo = nullptr, *ivec = nullptr, *dec_state = nullptr;
  int *bp = nullptr;
  int max_slots = 0;
  double *lin = nullptr, *quad = nullptr, *numf = nullptr, *x = nullptr, *cmvn_iv = nullptr, *cmvn_nn = nullptr;
  long long *dec_ctr = nullptr;
  std::vector<int> free_slots;
  std::map<int, int> free_rows;  // start -> length
  std::vector<void *> owned;
  float host_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};       // host time of the sections of an advance (RS_STREAMS_TRACE)
  float stage_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // device time of the stages, summed over the advances since the last finish

  int AllocRows(int n) {
    for (auto it = free_rows.begin(); it != free_rows.end(); ++it)
      if (it->second >= n) {
        const int start = it->first, len = it->second;
        free_rows.erase(it);
        if (len > n) free_rows[start + n] = len - n;
        return start;
      }
    return -1;
  }
  void FreeRows(int start, int n) {
    if (n <= 0) return;
    auto it = free_rows.emplace(start, n).first;
    auto nx = std::next(it);
    if (nx != free_rows.end() && it->first + it->second == nx->first) { it->second += nx->second; free_rows.erase(nx); }
    if (it != free_rows.begin()) {
      auto pv = std::prev(it);
      if (pv->first + pv->second == it->first) { pv->second += it->second; free_rows.erase(it); }
    }
  }
};

void StreamPoolDeleter::operator()(StreamPool *p) const {
  if (!p) return;
  p->issuer.Drain(false);
  if (p->qa) (void)hipStreamSynchronize(p->qa);
  if (p->qi) (void)hipStreamSynchronize(p->qi);
  if (p->q) (void)hipStreamSynchronize(p->q);
  if (p->qc) (void)hipStreamSynchronize(p->qc);
  for (int k = 0; k < StreamPool::kDepth; k++) {
    p->tm_a[k].reset(); p->tm_b[k].reset(); p->tm_c[k].reset(); p->tm_i[k].reset();
    for (hipEvent_t e : {p->ev_a[k], p->ev_b[k], p->ev_done[k], p->ev_f[k], p->ev_i[k]}) if (e) (void)hipEventDestroy(e);
  }
  for (void *d : p->owned) (void)hipFree(d);
  if (p->q) (void)hipStreamDestroy(p->q);
  if (p->qa) (void)hipStreamDestroy(p->qa);
  if (p->qc) (void)hipStreamDestroy(p->qc);
  if (p->qi) (void)hipStreamDestroy(p->qi);
  delete p;
}

namespace {
int RoundUp(int x, int m) { return (x + m - 1) / m * m; }
int EnvInt(const char *name, int dflt) { const char *e = std::getenv(name); return e ? std::atoi(e) : dflt; }

// index arrays of one advance: collected on the host, uploaded with one copy
struct IntStage {
  std::vector<int> h;
  size_t Add(const std::vector<int> &v) {
    const size_t o = h.size();
    h.insert(h.end(), v.begin(), v.end());
    while (h.size() & 3) h.push_back(0);
    return o;
  }
  size_t Add64(const std::vector<int64_t> &v) {
    const size_t o = h.size();
    h.resize(o + 2 * v.size());
    if (!v.empty()) std::memcpy(&h[o], v.data(), 8 * v.size());
    while (h.size() & 3) h.push_back(0);
    return o;
  }
};
}  // namespace

StreamPool *Model::Pool() {
  if (pool_) return pool_.get();
  ToDevice();
  RS_HIP(hipSetDevice(opts_.device_id));
  std::unique_ptr<StreamPool, StreamPoolDeleter> p(new StreamPool());
  const Nnet &nn = am_.nnet;
  const int C = fc_.mfcc.nceps, P = nn.output_dim;
  p->chunk = opts_.frames_per_chunk;
  p->rows = RoundUp(std::max(EnvInt("RS_STREAM_POOL_ROWS", 1 << 19), 4 * p->chunk), p->chunk);      // 2^19 frames = 87 min of live audio; ~10 KB per row for a 2000-pdf model
  p->max_slots = std::max(EnvInt("RS_STREAM_SLOTS", 512), 1);
  p->ld_c = RoundUp(C, 4);
  p->ld_ll = RoundUp(P, 4);
  p->S = hclg_.num_states();
  p->reg = reg_dev_.nt != 0 && !force_sparse_ && (decoder_choice_ == 0 || decoder_choice_ == 1);
  RS_HIP(hipStreamCreateWithFlags(&p->q, hipStreamNonBlocking));
  RS_HIP(hipStreamCreateWithFlags(&p->qa, hipStreamNonBlocking));
  RS_HIP(hipStreamCreateWithFlags(&p->qc, hipStreamNonBlocking));
  RS_HIP(hipStreamCreateWithFlags(&p->qi, hipStreamNonBlocking));
  for (int k = 0; k < StreamPool::kDepth; k++) {
    RS_HIP(hipEventCreateWithFlags(&p->ev_f[k], hipEventDisableTiming));
    RS_HIP(hipEventCreateWithFlags(&p->ev_i[k], hipEventDisableTiming));
    p->tm_i[k].reset(new Timer(p->qi));
    RS_HIP(hipEventCreateWithFlags(&p->ev_a[k], hipEventDisableTiming));
    RS_HIP(hipEventCreateWithFlags(&p->ev_b[k], hipEventDisableTiming));
    RS_HIP(hipEventCreateWithFlags(&p->ev_done[k], hipEventDisableTiming));
    p->tm_a[k].reset(new Timer(p->qa));
    p->tm_b[k].reset(new Timer(p->q));
    p->tm_c[k].reset(new Timer(p->qc));
  }
  { const char *e = TuneEnv("RS_STREAM_SYNC"); p->sync_each = e && std::atoi(e) != 0; }
  { const char *e = std::getenv("RS_STREAM_ISSUER"); p->use_issuer = !(e && std::atoi(e) == 0); }
  if (opts_.stream_min_ticks >= 1) p->min_ticks = opts_.stream_min_ticks;      // rs_decode_opts (0: the default above)
  { const char *e = std::getenv("RS_STREAM_MIN_TICKS"); if (e && std::atoi(e) >= 1) p->min_ticks = std::atoi(e); }
  p->issuer.device = opts_.device_id;
  auto dalloc = [&](size_t bytes) {
    void *d = nullptr;
    RS_HIP(hipMalloc(&d, std::max<size_t>(bytes, 256)));
    RS_HIP(hipMemsetAsync(d, 0, std::max<size_t>(bytes, 256), p->q));
    p->owned.push_back(d);
    return d;
  };
  const size_t R = (size_t)p->rows + 8;
  p->raw = static_cast<float *>(dalloc(R * p->ld_c * 4));
  if (fc_.ie.present) p->cm = static_cast<float *>(dalloc(R * p->ld_c * 4));
  if (fc_.use_cmvn) p->nn_in = static_cast<float *>(dalloc(R * p->ld_c * 4));
  p->ll = static_cast<float *>(dalloc(R * p->ld_ll * 4));
  if (p->reg) {
    p->bp = static_cast<int *>(dalloc(R * p->S * 4));
    p->finfo = static_cast<float *>(dalloc(R * 16));
    p->dec_state = static_cast<float *>(dalloc((size_t)p->max_slots * (2 * (size_t)p->S + 4) * 4));
  }
  p->dec_ctr = static_cast<long long *>(dalloc((size_t)p->max_slots * 64));
  if (fc_.ie.present) {
    const int Di = fc_.ie.ivector_dim(), usz = Di * (Di + 1) / 2;
    p->ld_i = RoundUp(Di, 4);
    p->ivec = static_cast<float *>(dalloc(((size_t)p->rows / p->chunk + 8) * p->ld_i * 4 + 1024));
    p->lin = static_cast<double *>(dalloc((size_t)p->max_slots * Di * 8));
    p->quad = static_cast<double *>(dalloc((size_t)p->max_slots * usz * 8));
    p->numf = static_cast<double *>(dalloc((size_t)p->max_slots * 8));
    p->x = static_cast<double *>(dalloc((size_t)p->max_slots * Di * 8));
    p->cmvn_iv = static_cast<double *>(dalloc((size_t)p->max_slots * (C + 1) * 8));
  }
  if (fc_.use_cmvn) p->cmvn_nn = static_cast<double *>(dalloc((size_t)p->max_slots * (C + 1) * 8));
  for (int s = p->max_slots - 1; s >= 0; s--) p->free_slots.push_back(s);
  p->free_rows[0] = p->rows;
  std::unique_ptr<DecodeContext> c(new DecodeContext());
  c->stream = p->q;
  RS_HIP(hipHostMalloc((void **)&c->gemm_ovf, 64, hipHostMallocMapped));
  c->gemm_ovf[0] = c->gemm_ovf[1] = 0;
  RS_HIP(hipHostGetDevicePointer((void **)&c->gemm_ovf_dev, c->gemm_ovf, 0));
  stream_ctx_ = std::move(c);
  p->cx = stream_ctx_.get();
  RS_HIP(hipStreamSynchronize(p->q));
  RS_HIP(hipStreamSynchronize(p->qa));
  RS_HIP(hipStreamSynchronize(p->qi));
  pool_ = std::move(p);
  return pool_.get();
}

// Waits for the advances still in flight and adds their stage times to the pool's totals (and to `extra`, if given: the finishing
// call's own share).  Device errors of those advances surface here.
// A closure the issuing thread ran for an advance (its acoustic-model and search launches) threw: the advance's rows were never
// computed, its done event never recorded -- and the exception comes out in whichever later call waits for the thread, possibly one
// that has nothing to do with the streams concerned (rs_stream_open).  So wherever it is collected, every open stream of the pool is
// poisoned with its message first (the streams used to carry on with missing rows behind a stale event).
void Model::IssuerSync(StreamPool *p) {
  try {
    p->issuer.Drain();
  } catch (const std::exception &e) {
    StreamsPoisonAll(e.what());
    throw;
  } catch (...) {
    StreamsPoisonAll("an advance's deferred work failed");
    throw;
  }
}

void Model::StreamsDrain(StreamPool *p, float *extra) {
  IssuerSync(p);
  for (int k = 0; k < StreamPool::kDepth; k++) {
    const int par = (int)((p->n_adv + k) % StreamPool::kDepth);      // the oldest advance first
    if (!p->pending[par]) continue;
    RS_HIP(hipEventSynchronize(p->ev_done[par]));
    p->pending[par] = false;
    p->Account(par, extra);
  }
  const hipError_t le = hipGetLastError();
  if (le != hipSuccess) Fail(std::string("a kernel launch failed: ") + hipGetErrorString(le));
  StreamsCheckRange();
}

// An advance cannot be repeated (its rows are written, the chunk schedule has moved on): when a layer GEMM met an activation
// the fp16 split cannot carry, the model changes to the exact-FP32 kernels for everything that follows and this call fails.
void Model::StreamsCheckRange() {
  DecodeContext *cx = stream_ctx_.get();
  // (only the split-fp16 kernels raise the word, so a raised word means an advance issued on them overflowed -- whether or not a
  // batch call has moved the model to the exact kernels in the meantime)
  if (!cx || !cx->gemm_ovf) return;
  volatile int *w = static_cast<volatile int *>(cx->gemm_ovf);
  const int over = w[0], under = w[1];
  if (over != 0 || under != 0) {
    w[0] = w[1] = 0;
    exact_gemm_.store(true);
    range_retries_.fetch_add(1);
    if (under != 0) precision_retries_.fetch_add(1);
    StreamsPoisonAll();        // (any of the advances in flight may be the one: their log-likelihood rows are not numbers / not the FP32 result)
    if (over != 0)
      Fail("an activation exceeded the range of the split-fp16 layer GEMMs (|x| >= 65520) during a stream advance; the model now uses the "
           "exact-FP32 kernels (RS_GEMM_B3=0 selects them from the start)");
    Fail("an activation row below the precision range of the split-fp16 layer GEMMs (largest |x| < 2^-3) was met during a stream advance; the model "
         "now uses the exact-FP32 kernels (RS_GEMM_B3=0 selects them from the start)");
  }
}

void Model::StreamOpen(rs_stream *st) {
  std::lock_guard<std::mutex> lk(pool_mu_);
  StreamPool *p = Pool();
  RS_HIP(hipSetDevice(opts_.device_id));
  // (this call queues on qi / qc itself.  A failure of an earlier advance's deferred work belongs to the streams that are open, not to
  // the one being opened: they are poisoned with its message and refuse their next call; the open goes ahead on the drained pool)
  try { IssuerSync(p); } catch (...) {}
  if (p->free_slots.empty()) Fail("too many live streams on this model (RS_STREAM_SLOTS=" + std::to_string(p->max_slots) + ")");
  const int want = RoundUp(std::max(EnvInt("RS_STREAM_INIT_FRAMES", 4096), 2 * p->chunk), p->chunk);
  const int row0 = p->AllocRows(want);
  if (row0 < 0) Fail("stream pool exhausted (RS_STREAM_POOL_ROWS=" + std::to_string(p->rows) + " frame rows of live audio)");
  st->slot = p->free_slots.back();
  p->free_slots.pop_back();
  st->row0 = row0;
  st->cap = want;
  // fresh estimator / search state in the slot
  if (fc_.ie.present) {
    const int Di = fc_.ie.ivector_dim(), usz = Di * (Di + 1) / 2;
    LaunchIvecInit(ivec_dev_, 1, p->lin + (size_t)st->slot * Di, p->quad + (size_t)st->slot * usz, p->x + (size_t)st->slot * Di, p->numf + st->slot, p->qi);      // (the iVector steps' queue: the estimator state is its)
  }
  RS_HIP(hipMemsetAsync(p->dec_ctr + (size_t)st->slot * 8, 0, 64, p->qc));      // (the search's queue)
  st->open = true;
  p->open_streams.push_back(st);
}

void Model::StreamClose(rs_stream *st) {
  if (!st->open) return;
  std::lock_guard<std::mutex> lk(pool_mu_);
  if (!pool_) return;
  // an advance that still uses the stream's rows / slot finishes first (a device error of it is the other streams' to report:
  // this one is going away either way)
  try { StreamsDrain(pool_.get(), nullptr); } catch (...) { pool_->issuer.Drain(false); (void)hipStreamSynchronize(pool_->qa); (void)hipStreamSynchronize(pool_->qi); (void)hipStreamSynchronize(pool_->q); (void)hipStreamSynchronize(pool_->qc); }
  pool_->FreeRows(st->row0, st->cap);
  pool_->free_slots.push_back(st->slot);
  st->open = false;
  auto &os = pool_->open_streams;
  os.erase(std::remove(os.begin(), os.end(), st), os.end());
}

// A failure that cannot be pinned on one advance's streams -- a device error that surfaced at a wait, up to three advances
// after the one that caused it; a layer GEMM that left its range in any of the advances in flight: no stream of the pool can
// trust its rows.  Everything queued is waited for, every set is free again, every open stream refuses further calls.
void Model::StreamsPoisonAll(const std::string &why) {
  StreamPool *p = pool_.get();
  if (!p) return;
  p->issuer.Drain(false);
  (void)hipStreamSynchronize(p->qa); (void)hipStreamSynchronize(p->qi); (void)hipStreamSynchronize(p->q); (void)hipStreamSynchronize(p->qc);
  (void)hipGetLastError();
  for (int k = 0; k < StreamPool::kDepth; k++) p->pending[k] = false;
  for (rs_stream *st : p->open_streams) {
    if (!st->failed && !why.empty()) st->fail_why = why;
    st->failed = true;
  }
}

// A stream outgrew its row range: move it to a range twice as long (device-to-device copies on the pool's stream).
void Model::StreamGrow(rs_stream *st, int need_frames) {
  StreamPool *p = pool_.get();
  StreamsDrain(p, nullptr);                 // both queues idle: the rows move under nobody's feet
  int want = st->cap;
  while (want < need_frames) want *= 2;
  const int row0 = p->AllocRows(want);
  if (row0 < 0) Fail("stream pool exhausted (RS_STREAM_POOL_ROWS=" + std::to_string(p->rows) + " frame rows of live audio)");
  const size_t n = (size_t)st->cap;
  auto mv = [&](void *base, size_t row_bytes) {
    if (base) RS_HIP(hipMemcpyAsync(static_cast<char *>(base) + (size_t)row0 * row_bytes, static_cast<char *>(base) + (size_t)st->row0 * row_bytes, n * row_bytes,
                                    hipMemcpyDeviceToDevice, p->q));
  };
  mv(p->raw, (size_t)p->ld_c * 4); mv(p->cm, (size_t)p->ld_c * 4); mv(p->nn_in, (size_t)p->ld_c * 4); mv(p->ll, (size_t)p->ld_ll * 4);
  mv(p->bp, (size_t)p->S * 4); mv(p->finfo, 16);
  if (p->ivec)
    RS_HIP(hipMemcpyAsync(p->ivec + (size_t)(row0 / p->chunk) * p->ld_i, p->ivec + (size_t)(st->row0 / p->chunk) * p->ld_i,
                          (size_t)(st->cap / p->chunk) * p->ld_i * 4, hipMemcpyDeviceToDevice, p->q));
  RS_HIP(hipStreamSynchronize(p->q));       // ... and stage A of the next advance (other queue) finds them in place
  p->FreeRows(st->row0, st->cap);
  st->row0 = row0;
  st->cap = want;
}

// One advance over `n` streams of this model.  final: end of input for all of them; results go to `res` (n utterances).
void Model::StreamsAdvance(rs_stream *const *streams, int n, bool final, int nbest, float lat_scale, Result *res) {
  std::lock_guard<std::mutex> lk(pool_mu_);
  try {
    StreamsAdvanceLocked(streams, n, final, nbest, lat_scale, res);
  } catch (const DeviceError &) {
    StreamsPoisonAll();        // a HIP error: which advance, which streams -- unknown
    throw;
  } catch (...) {
    // The advance may have queued part of its work on a set it never marked pending: whatever is queued finishes before the set
    // can be handed out again (the call's own streams are poisoned by the caller, api.cc).
    if (pool_) { pool_->issuer.Drain(false); (void)hipStreamSynchronize(pool_->qa); (void)hipStreamSynchronize(pool_->qi); (void)hipStreamSynchronize(pool_->q); (void)hipStreamSynchronize(pool_->qc); }
    throw;
  }
}

void Model::StreamsAdvanceLocked(rs_stream *const *streams, int n, bool final, int nbest, float lat_scale, Result *res) {
  StreamPool *p = Pool();
  RS_HIP(hipSetDevice(opts_.device_id));
  // queues: qa = features + iVectors (stage A), q = acoustic model + search (stage B, behind stage A's event); consecutive
  // advances rotate over kDepth arena / staging sets, so the host plans and issues the next advances while earlier ones still run
  // Coalescing: nothing but the end of a stream reads what an advance computes, and an advance is chains of small dependent launches
  // (the iVector estimator's per-chunk chain is the longest queue of a step: ~300 us per advance whatever the number of chunks) -- so
  // a call that brings less than min_ticks ticks of new audio on every stream leaves it to the next one.  The chunk / iVector
  // schedule is a function of the sample counts, not of the calls (the tick loop below): same rows, same results as with one
  // advance per tick or one at the end (the three delivery patterns of the stream tests).  RS_STREAM_MIN_TICKS=1: every call works.
  if (!final && p->min_ticks > 1) {
    long most = 0;
    for (int i = 0; i < n; i++) most = std::max(most, (long)(streams[i]->n_samples / 1024) - streams[i]->ticks_done);
    if (most < p->min_ticks) return;
  }
  hipStream_t qa = p->qa, q = p->q, qc = p->qc, qi = p->qi;
  DecodeContext &cx = *static_cast<DecodeContext *>(p->cx);
  const int par = (int)(p->n_adv % StreamPool::kDepth);
  DeviceArena &arena = cx.arena[par];
  HostArena &harena = cx.host_arena[par];
  if (final) IssuerSync(p);      // a finishing call issues everything itself, behind what the issuing thread still holds
  if (p->pending[par]) {       // the advance kDepth calls ago used this set: it has to be over (it normally is)
    const auto w0 = std::chrono::steady_clock::now();
    IssuerSync(p);         // (its done event has been recorded)
    RS_HIP(hipEventSynchronize(p->ev_done[par]));
    p->stage_ms[7] += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - w0).count();
    p->pending[par] = false;
    p->Account(par, nullptr);
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) Fail(std::string("a kernel launch failed: ") + hipGetErrorString(le));
    StreamsCheckRange();
  }
  SampleGemmMode(exact_gemm_.load(), cx.gemm_ovf_dev);
  const Nnet &nn = am_.nnet;
  const bool has_iv = fc_.ie.present;
  const int C = fc_.mfcc.nceps, P = nn.output_dim, chunk = p->chunk, Rm = nn.right_context, fsf = opts_.frame_subsampling_factor;
  const int shift = fc_.mfcc.shift;
  const int sl = has_iv ? fc_.ie.splice_left : 0, sr = has_iv ? fc_.ie.splice_right : 0;
  const int Dl = has_iv ? fc_.ie.feat_dim() : 0, Di = has_iv ? fc_.ie.ivector_dim() : 0, G = has_iv ? fc_.ie.num_gauss() : 0;
  const int ld_c = p->ld_c, ld_l = RoundUp(std::max(Dl, 1), 4), ld_i = p->ld_i, usz = Di * (Di + 1) / 2, nsel = has_iv ? fc_.ie.num_gselect : 0;
  auto wall0 = std::chrono::steady_clock::now();
  // (RS_STREAMS_TRACE: the host's time per section -- plan | arena + uploads | issue of stage A | B | C)
  auto host_last = wall0;
#define HOST_MARK(i) do { const auto n_ = std::chrono::steady_clock::now(); p->host_ms[i] += std::chrono::duration<float, std::milli>(n_ - host_last).count(); host_last = n_; } while (0)

  // ---------------------------------------------------------------- plan (host)
  struct Plan {
    int avail = 0;                 // frames computable from the samples accepted so far
    int mf0 = 0;                   // first new MFCC frame
    std::vector<std::pair<int, int>> chunks;      // new nnet chunks: (index, last frame its iVector has seen)
    int sa = 0, sb = 0;            // frames to splice / LDA / score for the iVector statistics: [sa, sb)
    int t0 = 0, t1 = 0;            // frames that get log-likelihoods in this advance: [t0, t1)
  };
  std::vector<Plan> pl(n);
  int max_new_chunks = 0;
  for (int i = 0; i < n; i++) {
    rs_stream &st = *streams[i];
    Plan &a = pl[i];
    const long ns = st.n_samples;
    a.avail = NumFrames(ns, fc_.mfcc.opts);
    if (a.avail + 2 > st.cap) StreamGrow(&st, a.avail + 2);
    a.mf0 = st.frames_mfcc;
    // the tick schedule (DecodeGroup's, resumed): ticks completed by the samples so far; the partial last read counts at EOF
    const long ticks = final ? (ns + 1023) / 1024 : ns / 1024;
    const int nch_final = (a.avail + chunk - 1) / chunk;
    for (long j = st.ticks_done; j < ticks; j++) {
      const int fr = NumFrames(std::min<long>(1024 * (j + 1), ns), fc_.mfcc.opts);
      const int ready = std::max(0, fr - Rm) / chunk;
      while (st.chunks_sched < ready && (!final || st.chunks_sched < nch_final)) {
        a.chunks.push_back({st.chunks_sched, std::min(fr - 1, fr - sr - 1)});
        st.chunks_sched++;
      }
    }
    st.ticks_done = ticks;
    if (final) while (st.chunks_sched < nch_final) { a.chunks.push_back({st.chunks_sched, a.avail - 1}); st.chunks_sched++; }
    max_new_chunks = std::max(max_new_chunks, (int)a.chunks.size());
    a.sa = a.sb = st.stats_done;
    for (auto &c : a.chunks) a.sb = std::max(a.sb, c.second + 1);
    a.t0 = st.ll_done;
    a.t1 = final ? a.avail : std::min(chunk * st.chunks_sched, a.avail);
  }
  // ---------------------------------------------------------------- transient geometry of the stages, index arrays
  IntStage is;
  std::vector<int> slots(n), row0s(n), avails(n);
  for (int i = 0; i < n; i++) { slots[i] = streams[i]->slot; row0s[i] = streams[i]->row0; avails[i] = (pl[i].avail + fsf - 1) / fsf; }
  // stage 1: MFCC over the new frames (dense rows, no halo), rows -> pool
  std::vector<int> m_T, m_rb{0}, m_out, m_f0;
  std::vector<int64_t> m_so{0};
  size_t pcm_total = 0;
  for (int i = 0; i < n; i++) {
    const int tn = pl[i].avail - pl[i].mf0;
    if (tn <= 0) continue;
    rs_stream &st = *streams[i];
    const long first = (long)pl[i].mf0 * shift, cnt = st.n_samples - first;
    m_T.push_back(tn);
    m_f0.push_back(pl[i].mf0);
    m_rb.push_back(m_rb.back() + tn);
    m_so.push_back(m_so.back() + cnt);
    { const size_t b0 = m_out.size(); m_out.resize(b0 + tn); int *po = m_out.data() + b0; const int r0 = st.row0 + pl[i].mf0; for (int t = 0; t < tn; t++) po[t] = r0 + t; }
    pcm_total += (size_t)cnt;
  }
  const int nM = (int)m_T.size(), rowsM = m_rb.back();
  m_T.push_back(0);
  m_f0.push_back(0);
  // stage 2: CMVN resumed (same streams)
  std::vector<int> c_T, c_rb, c_tb, c_slot;
  for (int i = 0; i < n; i++)
    if (pl[i].avail > pl[i].mf0) { c_T.push_back(pl[i].avail); c_rb.push_back(streams[i]->row0); c_tb.push_back(pl[i].mf0); c_slot.push_back(streams[i]->slot); }
  c_rb.push_back(0);
  // stage 3: iVector segments
  std::vector<int> I_idx;
  for (int i = 0; i < n; i++) if (has_iv && !pl[i].chunks.empty()) I_idx.push_back(i);
  const int nI = (int)I_idx.size();
  std::vector<int> i_T(nI + 1, 0), i_rb(nI + 1, 0), i_src, i_slot(nI);
  for (int u = 0; u < nI; u++) {
    const Plan &a = pl[I_idx[u]];
    const rs_stream &st = *streams[I_idx[u]];
    i_T[u] = a.sb - a.sa;
    i_rb[u + 1] = i_rb[u] + i_T[u] + sl + sr;
    i_slot[u] = st.slot;
    {
      const int nr = i_T[u] + sl + sr, hi = std::max(a.avail - 1, 0), t_first = a.sa - sl;
      const size_t b0 = i_src.size();
      i_src.resize(b0 + nr);
      int *ps = i_src.data() + b0;
      for (int r = 0; r < nr; r++) { const int t = t_first + r; ps[r] = st.row0 + (t < 0 ? 0 : (t > hi ? hi : t)); }
    }
  }
  const int rowsI = i_rb[nI];
  std::vector<int> s_fb((size_t)max_new_chunks * std::max(nI, 1), 0), s_fe(s_fb.size(), 0), s_or(s_fb.size(), -1), s_ac(s_fb.size(), 0);
  for (int u = 0; u < nI; u++) {
    const Plan &a = pl[I_idx[u]];
    const rs_stream &st = *streams[I_idx[u]];
    int done = a.sa;
    for (size_t k = 0; k < a.chunks.size(); k++) {
      const size_t o = k * nI + u;
      s_or[o] = st.row0 / chunk + a.chunks[k].first;
      if (a.chunks[k].second + 1 > done) { s_fb[o] = done - a.sa; s_fe[o] = a.chunks[k].second + 1 - a.sa; s_ac[o] = 1; done = a.chunks[k].second + 1; }
    }
  }
  // stage 4: nnet segments
  std::vector<int> N_idx;
  for (int i = 0; i < n; i++) if (pl[i].t1 > pl[i].t0) N_idx.push_back(i);
  const int nN = (int)N_idx.size();
  std::vector<int> n_T(nN + 1, 0), n_rb(nN + 1, 0), n_fb(nN + 1, 0), n_src, n_riv, n_lldst, n_llsrc;
  {
    size_t rows_est = 0;
    for (int u = 0; u < nN; u++) rows_est += (size_t)(pl[N_idx[u]].t1 - pl[N_idx[u]].t0) + L_ + R_;
    n_src.reserve(rows_est); n_riv.reserve(rows_est); n_lldst.reserve(rows_est);
  }
  int maxTn = 0;
  for (int u = 0; u < nN; u++) {
    const Plan &a = pl[N_idx[u]];
    const rs_stream &st = *streams[N_idx[u]];
    n_T[u] = a.t1 - a.t0;
    maxTn = std::max(maxTn, n_T[u]);
    n_rb[u + 1] = n_rb[u] + n_T[u] + L_ + R_;
    n_fb[u + 1] = n_fb[u] + n_T[u];
    {
      // per row: the pool row it is gathered from (context rows clamped to the stream's frames) and the chunk whose iVector its
      // Round(ivector, chunk) slot was supplied by (DecodeGroup: row_ivec) = the number of chunks j >= 1 with j * chunk + Rm <= slot_t,
      // at most the last scheduled one.  slot_t / chunk is carried along instead of divided out per row (the plan was 3.7 ms of a
      // 24 ms step on the host).
      const int nr = n_T[u] + L_ + R_, hi = std::max(a.avail - 1, 0), t_first = a.t0 - L_, kmax = std::max(st.chunks_sched - 1, 0), ivrow0 = st.row0 / chunk;
      const size_t b0 = n_src.size();
      n_src.resize(b0 + nr);
      n_riv.resize(b0 + nr);
      int *ps = n_src.data() + b0, *pr = n_riv.data() + b0;
      int q = t_first >= 0 ? t_first / chunk : -((-t_first + chunk - 1) / chunk), rem = t_first - q * chunk;      // t = q * chunk + rem, 0 <= rem < chunk
      auto k_of = [&](int qq) { const int d = qq * chunk - Rm; const int k = d >= 0 ? d / chunk : 0; return ivrow0 + (k > kmax ? kmax : k); };
      int kv = k_of(q);
      for (int r = 0; r < nr; r++) {
        const int t = t_first + r;
        ps[r] = st.row0 + (t < 0 ? 0 : (t > hi ? hi : t));
        pr[r] = kv;
        if (++rem == chunk) { rem = 0; kv = k_of(++q); }
      }
      if (fsf == 1) {
        const size_t l0 = n_lldst.size();
        n_lldst.resize(l0 + n_T[u]);
        int *pd = n_lldst.data() + l0;
        for (int t = 0; t < n_T[u]; t++) pd[t] = st.row0 + a.t0 + t;
      } else {
        // --frame-subsampling-factor: the decoder's frame t / fsf is the output row of t = 0, fsf, 2 fsf, ...; only those go to the pool
        for (int t = (a.t0 + fsf - 1) / fsf * fsf; t < a.t1; t += fsf) { n_llsrc.push_back(n_rb[u] + L_ + (t - a.t0)); n_lldst.push_back(st.row0 + t / fsf); }
      }
    }
  }
  const int rowsN = n_rb[nN], framesN = n_fb[nN];
  // stage 5: search windows (every stream of the call: a stream that ends without new rows still needs its traceback)
  SearchPlan sp;
  int maxT = 0;
  auto dec_frames = [&](int t) { return (t + fsf - 1) / fsf; };       // decoder frames of the first t feature frames (decodable-online-looped.cc:56-84)
  int max_feat_frames = 0;                                               // (the dither table is indexed by FEATURE frames)
  for (int i = 0; i < n; i++) { maxT = std::max(maxT, dec_frames(pl[i].avail)); max_feat_frames = std::max(max_feat_frames, pl[i].avail); }
  size_t search_bytes = 0;
  if (final) search_bytes = PlanSearch(n, maxT, nbest, lat_scale, &sp);
  const bool reg_windows = p->reg && !(final && sp.want_lattice);
  std::vector<int> d_T(n + 1, 0), d_rb(n + 1, 0), w_b(n), w_e(n), w_f(n, final ? 1 : 0);
  for (int i = 0; i < n; i++) {
    d_T[i] = dec_frames(pl[i].t1); d_rb[i] = streams[i]->row0;
    w_b[i] = streams[i]->dec_started ? streams[i]->frames_decoded : -1;
    w_e[i] = dec_frames(pl[i].t1);
  }
  const size_t o_mT = is.Add(m_T), o_mrb = is.Add(m_rb), o_mout = is.Add(m_out), o_mf0 = is.Add(m_f0), o_mso = is.Add64(m_so);
  const size_t o_cT = is.Add(c_T), o_crb = is.Add(c_rb), o_ctb = is.Add(c_tb), o_cslot = is.Add(c_slot);
  const size_t o_iT = is.Add(i_T), o_irb = is.Add(i_rb), o_isrc = is.Add(i_src), o_islot = is.Add(i_slot);
  const size_t o_sfb = is.Add(s_fb), o_sfe = is.Add(s_fe), o_sor = is.Add(s_or), o_sac = is.Add(s_ac);
  is.Add(n_T);
  const size_t o_nrb = is.Add(n_rb), o_nfb = is.Add(n_fb), o_nsrc = is.Add(n_src), o_nriv = is.Add(n_riv), o_nll = is.Add(n_lldst), o_nlls = is.Add(n_llsrc);
  const size_t o_dT = is.Add(d_T), o_drb = is.Add(d_rb), o_wb = is.Add(w_b), o_we = is.Add(w_e), o_wf = is.Add(w_f), o_slots = is.Add(slots), o_row0 = is.Add(row0s);
  // ---------------------------------------------------------------- arena
  HOST_MARK(0);
  const int guard = L_ + R_ + 8;
  auto fbytes = [&](int rows, int ld) { return ((size_t)rows + 2 * guard) * ld * sizeof(float) + 512; };
  std::vector<int> buf_ld(nn.bufs.size());
  size_t need = is.h.size() * 4 + pcm_total * 2 + 4096 + 3 * sizeof(int) * (size_t)(rowsM + rowsI + rowsN + 64) + sizeof(int) * (size_t)(framesN + 64);
  for (size_t b = 0; b < nn.bufs.size(); b++) { buf_ld[b] = RoundUp(nn.bufs[b].dim, 4); need += fbytes(rowsN, buf_ld[b]); }
  need += ImageBytes(rowsN);
  if (has_iv) {
    need += 2 * fbytes(rowsI, ld_c) + 2 * fbytes(rowsI, ld_l) + (size_t)rowsI * nsel * 8 + 4096;
    need += (size_t)std::max(nI, 1) * ((size_t)G * 8 + (size_t)G * Dl * 8 + (size_t)Di * 16 + (size_t)usz * 8 + 64) + 8192;
    need += IvecStatsScratchDoubles(ivec_dev_, std::max(nI, 1)) * 8 + 1024;
    need += IvecChunkChainBytes(ivec_dev_, std::max(nI, 1), std::max(max_new_chunks, 1));
  }
  need += search_bytes + (size_t)n * 64 * 8 + 64 * 256 + (1u << 20);
  arena.Reserve(need, qa);
  arena.Reset();
  harena.Reset();
  // ---------------------------------------------------------------- uploads: index arrays, new samples
  int *d_is = arena.AllocT<int>(is.h.size() + 16);
  {
    int *hp = harena.AllocT<int>(is.h.size() + 16);
    std::memcpy(hp, is.h.data(), is.h.size() * 4);
    RS_HIP(hipMemcpyAsync(d_is, hp, is.h.size() * 4, hipMemcpyHostToDevice, qa));
  }
  auto D = [&](size_t off) { return d_is + off; };
  int16_t *d_pcm = arena.AllocT<int16_t>(pcm_total + 512);
  if (pcm_total) {
    int16_t *hp = harena.AllocT<int16_t>(pcm_total + 512);
    size_t o = 0;
    for (int i = 0; i < n; i++) {
      rs_stream &st = *streams[i];
      if (pl[i].avail <= pl[i].mf0) continue;
      const long first = (long)pl[i].mf0 * shift, cnt = st.n_samples - first;
      std::memcpy(hp + o, st.pcm.data() + (first - st.pcm_start), sizeof(int16_t) * (size_t)cnt);
      o += (size_t)cnt;
    }
    RS_HIP(hipMemcpyAsync(d_pcm, hp, sizeof(int16_t) * pcm_total, hipMemcpyHostToDevice, qa));
  }
  Timer &tma = *p->tm_a[par], &tmb = *p->tm_b[par], &tmc = *p->tm_c[par], &tmi = *p->tm_i[par];
  tma.Reset(); tmb.Reset(); tmc.Reset(); tmi.Reset();
  const bool timed = final || p->n_adv % StreamPool::kTimedEvery == 0;
  p->time_weight[par] = final ? 1.f : (timed ? (float)StreamPool::kTimedEvery : 0.f);
#define TM_MARK(t) do { if (timed) (t).Mark(); } while (0)
  TM_MARK(tma);
  // ---------------------------------------------------------------- 1. MFCC
  HOST_MARK(1);
  if (nM > 0) {
    BatchGeom g;
    g.n_utts = nM; g.total_rows = rowsM; g.total_frames = rowsM;
    g.d_sample_off = reinterpret_cast<const int64_t *>(D(o_mso)); g.d_num_frames = D(o_mT); g.d_row_base = D(o_mrb);
    g.d_frame0 = D(o_mf0);
    int *ru = arena.AllocT<int>(rowsM), *rt = arena.AllocT<int>(rowsM);
    LaunchRowGeometry(nM, rowsM, 0, D(o_mrb), nullptr, ru, rt, nullptr, qa);
    g.d_row_utt = ru; g.d_row_t = rt;
    LaunchMfcc(MfccWithDither(max_feat_frames), g, d_pcm, p->raw, ld_c, qa, false, D(o_mout));
    // ---------------------------------------------------------------- 2. CMVN, resumed
    BatchGeom gc;
    gc.n_utts = nM; gc.d_num_frames = D(o_cT); gc.d_row_base = D(o_crb);
    if (has_iv) LaunchOnlineCmvn(cmvn_iv_dev_, gc, p->raw, p->cm, ld_c, qa, D(o_ctb), p->cmvn_iv, D(o_cslot));
    if (fc_.use_cmvn) LaunchOnlineCmvn(cmvn_nnet_dev_, gc, p->raw, p->nn_in, ld_c, qa, D(o_ctb), p->cmvn_nn, D(o_cslot));
  }
  TM_MARK(tma);
  // ---------------------------------------------------------------- 3. iVectors of the new chunks
  auto falloc = [&](int rows, int ld) { return arena.AllocT<float>(((size_t)rows + 2 * guard) * ld) + (size_t)guard * ld; };
  if (nI > 0) {
    float *seg_raw = falloc(rowsI, ld_c), *seg_cm = falloc(rowsI, ld_c), *lda_raw = falloc(rowsI, ld_l), *lda_norm = falloc(rowsI, ld_l);
    LaunchCopyRows(p->raw, ld_c, D(o_isrc), seg_raw, ld_c, nullptr, rowsI, C, qa);
    LaunchCopyRows(p->cm, ld_c, D(o_isrc), seg_cm, ld_c, nullptr, rowsI, C, qa);
    BatchGeom g;
    g.n_utts = nI; g.L = sl; g.R = sr; g.total_rows = rowsI; g.guard = guard;
    g.d_num_frames = D(o_iT); g.d_row_base = D(o_irb);
    int *ru = arena.AllocT<int>(rowsI + 8), *rt = arena.AllocT<int>(rowsI + 8);
    LaunchRowGeometry(nI, rowsI, sl, D(o_irb), nullptr, ru, rt, nullptr, qa);
    g.d_row_utt = ru; g.d_row_t = rt;
    LaunchGemm(MakeGemm(LdaPlan(ld_c), {seg_raw}, {ld_c}, nullptr, 0, lda_raw, ld_l, 1), rowsI, ru, qa);
    LaunchGemm(MakeGemm(LdaPlan(ld_c), {seg_cm}, {ld_c}, nullptr, 0, lda_norm, ld_l, 1), rowsI, ru, qa);
    int *post_idx = arena.AllocT<int>((size_t)rowsI * nsel + 64);
    float *post_w = arena.AllocT<float>((size_t)rowsI * nsel + 64);
    LaunchUbmPosteriors(ivec_dev_, g, lda_norm, ld_l, post_idx, post_w, qa);
    // the estimator's steps on their own queue, behind the posteriors
    RS_HIP(hipEventRecord(p->ev_f[par], qa));
    RS_HIP(hipStreamWaitEvent(qi, p->ev_f[par], 0));
    TM_MARK(tmi);
    const float *stats_feats = fc_.ie.online_cmvn_iextractor ? lda_norm : lda_raw;
    if (Di <= 128) {
      // the new chunks of all streams: statistics side by side, then one launch that walks every stream's chunks in order on its
      // estimator state in place (round 4: state rows gathered, a five-launch chain per chunk, state rows scattered)
      IvecChunkChain(arena, g, nI, max_new_chunks, stats_feats, ld_l, post_idx, post_w, D(o_sfb), D(o_sfe), D(o_sor), D(o_sac), p->lin, p->quad, p->numf,
                     p->x, D(o_islot), p->ivec, ld_i, qi);
    } else {
      double *gamma = arena.AllocT<double>((size_t)nI * G), *wfeats = arena.AllocT<double>((size_t)nI * G * Dl);
      double *linear = arena.AllocT<double>((size_t)nI * Di), *quad = arena.AllocT<double>((size_t)nI * usz);
      double *numf = arena.AllocT<double>(nI), *x = arena.AllocT<double>((size_t)nI * Di);
      double *scratch = arena.AllocT<double>(IvecStatsScratchDoubles(ivec_dev_, nI));
      // estimator state: slots -> dense, the steps, dense -> slots (one launch each way for the four arrays)
      CopyRowsSet in_set{{{p->lin, linear, 2L * Di, 2 * Di}, {p->quad, quad, 2L * usz, 2 * usz}, {p->numf, numf, 2, 2}, {p->x, x, 2L * Di, 2 * Di}}, 4};
      LaunchCopyRowsMulti(in_set, D(o_islot), nullptr, nI, qi);
      for (int k = 0; k < max_new_chunks; k++) {
        const size_t o = (size_t)k * nI;
        LaunchIvecAccumulate(ivec_dev_, g, stats_feats, ld_l, post_idx, post_w, D(o_sfb) + o, D(o_sfe) + o, gamma, wfeats, true, qi);
        LaunchIvecStats(ivec_dev_, nI, gamma, wfeats, linear, quad, numf, scratch, qi);
        LaunchIvecSolve(ivec_dev_, nI, linear, quad, numf, x, p->ivec, ld_i, D(o_sor) + o, D(o_sac) + o, qi);
      }
      CopyRowsSet out_set{{{linear, p->lin, 2L * Di, 2 * Di}, {quad, p->quad, 2L * usz, 2 * usz}, {numf, p->numf, 2, 2}, {x, p->x, 2L * Di, 2 * Di}}, 4};
      LaunchCopyRowsMulti(out_set, nullptr, D(o_islot), nI, qi);
    }
    TM_MARK(tmi);
  }
  TM_MARK(tma);
  // the acoustic model waits for both: the features (qa) and, where there is an extractor, the iVectors (qi, itself behind qa)
  RS_HIP(hipEventRecord(p->ev_a[par], qa));
  if (nI > 0) RS_HIP(hipEventRecord(p->ev_i[par], qi));
  // ---------------------------------------------------------------- 4. acoustic model over the new chunks (+ context), 5. search
  // Everything below queues on q / qc and needs nothing this thread computes later: a closure over values, run here by a finishing
  // call and handed to the pool's issuing thread otherwise (StreamIssuer).  Buffers come out of the arena here, in this thread.
  HOST_MARK(2);
  std::vector<float *> bufp(nn.bufs.size(), nullptr);
  std::vector<ActImage> imgs;
  int *frame_rows = nullptr;
  if (nN > 0) {
    for (size_t b = 0; b < nn.bufs.size(); b++) bufp[b] = falloc(rowsN, buf_ld[b]);
    imgs = AllocImages(arena, rowsN);
    frame_rows = arena.AllocT<int>(framesN + 8);
  }
  BatchGeom gd;
  gd.n_utts = n; gd.max_frames = maxT; gd.d_num_frames = D(o_dT); gd.d_row_base = D(o_drb);
  DenseWork dw;
  std::memset(&dw, 0, sizeof(dw));
  dw.bp = p->bp; dw.frame_info = p->finfo; dw.state_cost = p->dec_state; dw.counters = p->dec_ctr;
  dw.win_begin = D(o_wb); dw.win_end = D(o_we); dw.win_final = D(o_wf); dw.pool_row = D(o_row0); dw.slot = D(o_slots);
  DecodeOptsDev dopts;
  dopts.beam = opts_.beam; dopts.lattice_beam = opts_.lattice_beam; dopts.beam_delta = opts_.beam_delta;
  dopts.max_active = opts_.max_active; dopts.min_active = opts_.min_active;
  dopts.exact_order = ExactOrder() ? 1 : 0;
  {
    const float *nn_src = fc_.use_cmvn ? p->nn_in : p->raw;
    const int *d_nsrc = D(o_nsrc), *d_nfb = D(o_nfb), *d_nrb = D(o_nrb), *d_nriv = D(o_nriv), *d_nll = D(o_nll), *d_nlls = D(o_nlls), *d_slots = D(o_slots);
    const int n_sub = (int)n_lldst.size();
    const bool have_sub = !n_llsrc.empty(), exact_now = exact_gemm_.load();
    int *const ovf_dev = cx.gemm_ovf_dev;
    Timer *const tb = &tmb, *const tc = &tmc;
    SearchPlan *const spp = &sp;            // (a finishing call only; it runs the closure itself)
    DeviceArena *const arp = &arena;
    hipEvent_t ev_a = p->ev_a[par], ev_i = p->ev_i[par], ev_b = p->ev_b[par], ev_done = p->ev_done[par];
    const int maxTn_c = std::max(maxTn, 1);
    const Nnet *const nnp = &nn;             // (a pointer: [=] on the reference would copy the network)
    auto issue_bc = [=]() {
      SampleGemmMode(exact_now, ovf_dev);      // (thread-local: the issuing thread's launches run in the mode this advance was planned in)
      RS_HIP(hipStreamWaitEvent(q, ev_a, 0));
      if (nI > 0) RS_HIP(hipStreamWaitEvent(q, ev_i, 0));
      if (timed) tb->Mark();
      if (nN > 0) {
        LaunchCopyRows(nn_src, ld_c, d_nsrc, bufp[nnp->input_buf], buf_ld[nnp->input_buf], nullptr, rowsN, C, q);
        LaunchFrameRows(nN, nN, framesN, L_, maxTn_c, d_nfb, d_nrb, frame_rows, q);
        RowMaps row_maps;
        row_maps.maps.push_back({0, 0, frame_rows, framesN});
        if (fsf > 1 && have_sub) row_maps.maps.push_back({0, 0, d_nlls, n_sub, 0, fsf});      // the layers only the decoder's frames read
        RunNnet(bufp, buf_ld, p->ivec, ld_i, d_nriv, rowsN, row_maps, 1, 0, nnp->ops.size(), q, &imgs);
        if (fsf == 1) LaunchCopyRows(bufp[nnp->output_buf], buf_ld[nnp->output_buf], frame_rows, p->ll, p->ld_ll, d_nll, framesN, P, q);
        else if (n_sub > 0) LaunchCopyRows(bufp[nnp->output_buf], buf_ld[nnp->output_buf], d_nlls, p->ll, p->ld_ll, d_nll, n_sub, P, q);
      }
      if (timed) tb->Mark();
      // the search on its own queue: the next advance's acoustic model does not wait for it
      RS_HIP(hipEventRecord(ev_b, q));
      RS_HIP(hipStreamWaitEvent(qc, ev_b, 0));
      if (timed) tc->Mark();
      if (final) AllocSearch(spp, *arp, qc, /*pooled_frames=*/reg_windows);
      if (reg_windows && (final || nN > 0)) {      // (an advance without new log-likelihood rows has nothing to search)
        DenseWork dw2 = dw;
        if (final) { dw2 = spp->dw; dw2.bp = dw.bp; dw2.frame_info = dw.frame_info; dw2.state_cost = dw.state_cost; dw2.counters = dw.counters;
                     dw2.win_begin = dw.win_begin; dw2.win_end = dw.win_end; dw2.win_final = dw.win_final; dw2.pool_row = dw.pool_row; dw2.slot = dw.slot; }
        LaunchDecodeReg(hclg_dev_, reg_dev_, dopts, gd, p->ll, p->ld_ll, dw2, 0, 0, qc, final);
        if (final) LaunchCopyRows(p->dec_ctr, 16, d_slots, spp->w.counters, 16, nullptr, n, 16, qc);
      } else if (final) {
        LaunchSearch(spp, *arp, gd, p->ll, p->ld_ll, qc);
      }
      if (timed) tc->Mark();
      RS_HIP(hipEventRecord(ev_done, qc));
      const hipError_t le = hipGetLastError();      // (per thread: a failed launch of this closure is seen here)
      if (le != hipSuccess) Fail(std::string("a kernel launch failed: ") + hipGetErrorString(le));
    };
    if (final || !p->use_issuer) issue_bc();
    else p->issuer.Submit(issue_bc);
  }
  HOST_MARK(3);
  // ---------------------------------------------------------------- host bookkeeping
  HOST_MARK(4);
  for (int i = 0; i < n; i++) {
    rs_stream &st = *streams[i];
    const Plan &a = pl[i];
    st.frames_mfcc = a.avail;
    st.stats_done = a.sb;
    st.ll_done = a.t1;
    if (reg_windows && (final || nN > 0)) { st.frames_decoded = dec_frames(a.t1); st.dec_started = true; }      // (the search was launched)
    // samples before the first frame that is not complete yet are not needed again (online-feature.cc:186-203)
    const long keep_from = (long)a.avail * shift;
    if (!st.keep_pcm && keep_from > st.pcm_start) {
      st.pcm.erase(st.pcm.begin(), st.pcm.begin() + std::min<long>(keep_from - st.pcm_start, (long)st.pcm.size()));
      st.pcm_start = keep_from;
    }
  }
  p->pending[par] = true;         // (its done event is recorded by the closure above)
  p->n_adv++;
  if (!final) {
    // No wait here: the next advance is planned and issued while this one runs.  What it did on the device is accounted for --
    // and a device error of it reported -- by whichever later call of this model waits for it.
    if (p->sync_each) StreamsDrain(p, nullptr);
    p->stage_ms[6] += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    return;
  }
  float own[4] = {0.f, 0.f, 0.f, 0.f};
  {
    // every earlier advance first (their time is the stream's, not this call's), then this one
    for (int k = 1; k < StreamPool::kDepth; k++) {
      const int other = (par + k) % StreamPool::kDepth;       // oldest first
      if (!p->pending[other]) continue;
      RS_HIP(hipEventSynchronize(p->ev_done[other]));
      p->pending[other] = false;
      p->Account(other, nullptr);
    }
    RS_HIP(hipEventSynchronize(p->ev_done[par]));
    p->pending[par] = false;
    p->Account(par, own);
    StreamsCheckRange();
  }
  // ---------------------------------------------------------------- results
  res->utts.resize(n);
  for (int i = 0; i < n; i++) res->utts[i].num_frames = dec_frames(pl[i].avail);
  Timer tmr(qc);
  tmr.Mark();
  CollectResults(sp, cx, par, gd, avails.data(), p->ll, p->ld_ll, nbest, lat_scale, qc, res->utts.data(), res->timings);
  tmr.Mark();
  if (opts_.keep_intermediates) {
    const float *fin = fc_.use_cmvn ? p->nn_in : p->raw;
    for (int i = 0; i < n; i++) {
      UttResult &ur = res->utts[i];
      const rs_stream &st = *streams[i];
      const int T = pl[i].avail, nch = std::max((T + chunk - 1) / chunk, 1);
      ur.feat_dim = C; ur.num_pdfs = P; ur.ivec_dim = Di; ur.ivec_rows = has_iv ? nch : 0;
      if (T == 0) continue;
      ur.feats.resize((size_t)T * C);
      ur.loglikes.resize((size_t)dec_frames(T) * P);
      RS_HIP(hipMemcpy2D(ur.feats.data(), sizeof(float) * C, fin + (size_t)st.row0 * ld_c, sizeof(float) * ld_c, sizeof(float) * C, T, hipMemcpyDeviceToHost));
      RS_HIP(hipMemcpy2D(ur.loglikes.data(), sizeof(float) * P, p->ll + (size_t)st.row0 * p->ld_ll, sizeof(float) * p->ld_ll, sizeof(float) * P, dec_frames(T),
                         hipMemcpyDeviceToHost));
      if (has_iv) {
        ur.ivector.resize((size_t)nch * Di);
        RS_HIP(hipMemcpy2D(ur.ivector.data(), sizeof(float) * Di, p->ivec + (size_t)(st.row0 / chunk) * ld_i, sizeof(float) * ld_i, sizeof(float) * Di, nch,
                           hipMemcpyDeviceToHost));
      }
    }
  }
  { const hipError_t le = hipGetLastError(); if (le != hipSuccess) Fail(std::string("a kernel launch failed: ") + hipGetErrorString(le)); }
  // stage times of the whole stream(s): this call plus the advances since the previous finish on this model; [7] = this call alone
  res->timings[7] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - wall0).count();
  for (int k = 0; k < 4; k++) res->timings[k + 1] = p->stage_ms[k + 1];
  res->timings[5] = tmr.Ms(0, 1);
  res->timings[6] = p->stage_ms[6] + res->timings[7];
  // RS_STREAMS_TRACE=1: where the host's time went since the last finish -- inside the advance calls, and of that waiting for the
  // advance three calls back to leave its arena set (the device is the bottleneck when that is most of it)
  static const bool trace = [] { const char *e = TuneEnv("RS_STREAMS_TRACE"); return e && std::atoi(e) != 0; }();
  if (trace) {
    std::fprintf(stderr, "streams: %.2f ms in advance calls, %.2f ms of it waiting for the device; plan %.2f, arena + uploads %.2f, issue of stage A %.2f, B %.2f, C %.2f\n",
                 p->stage_ms[6], p->stage_ms[7], p->host_ms[0], p->host_ms[1], p->host_ms[2], p->host_ms[3], p->host_ms[4]);
    for (float &v : p->host_ms) v = 0.f;
  }
  for (float &v : p->stage_ms) v = 0.f;
}

}  // namespace rs
