// rs_decode_batch_sharded: the multi-GPU entry point of SURVEY.md section 8(b)/(e).
//
// The reference has no parallelism (one process per utterance, rhasspy_speech/tools.py:117-147), so utterances are
// independent: utterance i belongs to rank i % world, every rank runs the whole path on its utterances with replicated
// models (one device batch per model, the models' batches concurrently from one host thread each), and ONE collective --
// ncclAllGather of fixed 272-byte records over the caller's RCCL communicator -- returns every utterance's 1-best to every
// rank.  Failures travel inside the records (status field), never around the collective: a rank whose decode failed still
// takes part in the gather, so no rank is left waiting.
//
// RCCL is bound at first use with dlopen("librccl.so.1"): a process that already carries RCCL (torch.distributed's "nccl"
// backend IS RCCL on ROCm) gets that same copy, so a communicator created there (ProcessGroupNCCL._comm_ptr()) is valid
// here; a process that never shards never loads it.
#include <dlfcn.h>

#include <cstring>
#include <exception>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "engine.h"

namespace rs {
namespace {

struct Rccl {
  using AllGatherFn = int (*)(const void *, void *, size_t, int, void *, hipStream_t);
  using CountFn = int (*)(void *, int *);
  using ErrFn = const char *(*)(int);
  AllGatherFn all_gather = nullptr;
  CountFn comm_count = nullptr, comm_rank = nullptr;
  ErrFn error_string = nullptr;
  std::string load_error;
  static const Rccl &Get() {
    static const Rccl r = [] {
      Rccl x;
      void *h = nullptr;
      for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
      }
      if (!h) { x.load_error = std::string("cannot load RCCL: ") + dlerror(); return x; }
      x.all_gather = reinterpret_cast<AllGatherFn>(dlsym(h, "ncclAllGather"));
      x.comm_count = reinterpret_cast<CountFn>(dlsym(h, "ncclCommCount"));
      x.comm_rank = reinterpret_cast<CountFn>(dlsym(h, "ncclCommUserRank"));
      x.error_string = reinterpret_cast<ErrFn>(dlsym(h, "ncclGetErrorString"));
      if (!x.all_gather || !x.comm_count || !x.comm_rank) x.load_error = "RCCL library lacks ncclAllGather / ncclCommCount / ncclCommUserRank";
      return x;
    }();
    return r;
  }
};
constexpr int kNcclInt32 = 2;      // ncclInt32 (rccl.h: ncclDataType_t)

// grow-only device + pinned staging for the gather (a handful of KB; never freed: hipFree would stall every stream)
struct GatherBuffers {
  std::mutex mu;
  int device = -1;
  int32_t *d_send = nullptr, *d_recv = nullptr, *h_stage = nullptr;
  size_t send_cap = 0, recv_cap = 0;
  hipStream_t stream = nullptr;
};
GatherBuffers &Buffers() { static GatherBuffers b; return b; }

void FillRecord(int32_t *rec, int utt, const UttResult &ur) {
  std::memset(rec, 0, sizeof(int32_t) * RS_SHARD_RECORD_INTS);
  rec[0] = utt;
  rec[1] = ur.status;
  if (ur.status != RS_OK || ur.hyps.empty()) { if (ur.status == RS_OK) rec[1] = RS_ERR_DECODE; return; }
  const Hypothesis &h = ur.hyps[0];
  rec[2] = (int32_t)h.words.size();                       // the full length: > RS_SHARD_MAX_WORDS tells the reader it was cut
  const size_t n = std::min<size_t>(h.words.size(), RS_SHARD_MAX_WORDS);
  std::memcpy(rec + 3, h.words.data(), sizeof(int32_t) * n);
  std::memcpy(rec + 3 + RS_SHARD_MAX_WORDS, &h.graph_cost, 4);
  std::memcpy(rec + 4 + RS_SHARD_MAX_WORDS, &h.acoustic_cost, 4);
}

// The exchange step's resources for `per` records per rank on device `dev`; called with gb.mu held.  Everything that can fail
// (device, stream, buffers) happens here, BEFORE a rank decodes: a rank that cannot take part in the collective fails where
// every rank of a broken node fails alike, and after the decode nothing stands between a rank and the collective.
void PrepareGather(GatherBuffers &gb, int dev, size_t send_bytes, size_t recv_bytes) {
  auto release = [&]() {
    if (gb.stream) (void)hipStreamSynchronize(gb.stream);      // no gather is still reading the old buffers
    if (gb.d_send) (void)hipFree(gb.d_send);
    if (gb.d_recv) (void)hipFree(gb.d_recv);
    if (gb.h_stage) (void)hipHostFree(gb.h_stage);
    gb.d_send = gb.d_recv = gb.h_stage = nullptr;
    gb.send_cap = gb.recv_cap = 0;
  };
  if (gb.device != dev) {                                      // the buffers of another device go back to that device
    if (gb.device >= 0 && hipSetDevice(gb.device) == hipSuccess) {
      release();
      if (gb.stream) (void)hipStreamDestroy(gb.stream);
    }
    gb.d_send = gb.d_recv = gb.h_stage = nullptr; gb.send_cap = gb.recv_cap = 0; gb.stream = nullptr;
    gb.device = dev;
  }
  RS_HIP(hipSetDevice(dev));
  if (!gb.stream) RS_HIP(hipStreamCreateWithFlags(&gb.stream, hipStreamNonBlocking));
  if (send_bytes > gb.send_cap || recv_bytes > gb.recv_cap) {
    release();
    RS_HIP(hipMalloc((void **)&gb.d_send, send_bytes * 2));
    RS_HIP(hipMalloc((void **)&gb.d_recv, recv_bytes * 2));
    RS_HIP(hipHostMalloc((void **)&gb.h_stage, recv_bytes * 2, hipHostMallocDefault));
    gb.send_cap = send_bytes * 2; gb.recv_cap = recv_bytes * 2;
  }
}

void CheckComm(void *comm, int rank, int world, const char *who) {
  // a communicator that contradicts rank / world is refused before anything is launched (nobody would be left waiting)
  const Rccl &nc = Rccl::Get();
  if (!nc.load_error.empty()) throw DeviceError(nc.load_error);
  int cn = 0, cr = -1;
  if (nc.comm_count(comm, &cn) != 0 || nc.comm_rank(comm, &cr) != 0 || cn != world || cr != rank)
    throw Error(std::string(who) + ": rank/world (" + std::to_string(rank) + "/" + std::to_string(world) +
                ") do not match the communicator's (" + std::to_string(cr) + "/" + std::to_string(cn) + ")");
}

void InitRecords(int32_t *records, int n_utts) {
  for (int i = 0; i < n_utts; i++) {
    int32_t *rec = records + (size_t)i * RS_SHARD_RECORD_INTS;
    std::memset(rec, 0, sizeof(int32_t) * RS_SHARD_RECORD_INTS);
    rec[0] = i; rec[1] = RS_SHARD_ABSENT;
  }
}

void Scatter(const int32_t *block, int nrec, int n_utts, int32_t *records) {
  for (int k = 0; k < nrec; k++) {
    const int32_t *rec = block + (size_t)k * RS_SHARD_RECORD_INTS;
    if (rec[0] >= 0 && rec[0] < n_utts) std::memcpy(records + (size_t)rec[0] * RS_SHARD_RECORD_INTS, rec, sizeof(int32_t) * RS_SHARD_RECORD_INTS);
  }
}

// ONE ncclAllGather of `per` records per rank; gb.mu held, PrepareGather done.  `local`: per * RS_SHARD_RECORD_INTS ints.
void Gather(GatherBuffers &gb, const int32_t *local, int per, int n_utts, int world, void *comm, int32_t *records) {
  const Rccl &nc = Rccl::Get();
  const size_t n_ints = (size_t)per * RS_SHARD_RECORD_INTS, send_bytes = n_ints * sizeof(int32_t), recv_bytes = send_bytes * world;
  (void)hipSetDevice(gb.device);
  std::memcpy(gb.h_stage, local, send_bytes);
  // Should the upload fail, this rank still joins the collective -- with records marked absent -- and reports afterwards.
  const hipError_t up = hipMemcpyAsync(gb.d_send, gb.h_stage, send_bytes, hipMemcpyHostToDevice, gb.stream);
  if (up != hipSuccess) (void)hipMemsetAsync(gb.d_send, 0xFF, send_bytes, gb.stream);      // utterance index -1: skipped by every reader
  const int nr = nc.all_gather(gb.d_send, gb.d_recv, n_ints, kNcclInt32, comm, gb.stream);
  if (nr != 0) throw DeviceError(std::string("ncclAllGather failed: ") + (nc.error_string ? nc.error_string(nr) : std::to_string(nr).c_str()));
  RS_HIP(hipMemcpyAsync(gb.h_stage, gb.d_recv, recv_bytes, hipMemcpyDeviceToHost, gb.stream));
  RS_HIP(hipStreamSynchronize(gb.stream));
  if (up != hipSuccess) throw DeviceError(std::string("uploading this rank's records for the gather failed: ") + hipGetErrorString(up));
  InitRecords(records, n_utts);
  Scatter(gb.h_stage, per * world, n_utts, records);
}

}  // namespace

// returns RS_OK, or the first per-model failure of this rank (after the collective has run)
int DecodeBatchSharded(rs_model *const *models, int n_models, const int32_t *utt_model, const int16_t *const *pcm, const int32_t *n_samples,
                       int n_utts, int rank, int world, void *comm, int32_t *records, std::string *error) {
  if (comm) CheckComm(comm, rank, world, "rs_decode_batch_sharded");
  const int per = (n_utts + world - 1) / world;           // records per rank in the gather (short shards are padded)
  const size_t send_bytes = (size_t)per * RS_SHARD_RECORD_INTS * sizeof(int32_t);
  GatherBuffers &gb = Buffers();
  if (comm) {
    std::lock_guard<std::mutex> lk(gb.mu);
    PrepareGather(gb, models[0]->m->opts().device_id, send_bytes, send_bytes * world);
  }
  std::vector<std::vector<int>> mine(n_models);
  for (int i = rank; i < n_utts; i += world) mine[utt_model[i]].push_back(i);
  // ---- this rank's utterances: one device batch per model, the batches of different models concurrently
  std::vector<std::unique_ptr<Result>> results(n_models);
  std::vector<std::string> errs(n_models);
  std::vector<int> codes(n_models, RS_OK);
  auto run = [&](int m) {
    if (mine[m].empty()) return;
    try {
      std::vector<const int16_t *> p;
      std::vector<int32_t> n;
      for (int i : mine[m]) { p.push_back(pcm[i]); n.push_back(n_samples[i]); }
      results[m] = models[m]->m->DecodeBatchHost(p.data(), n.data(), (int)p.size(), 1, 1.0f);
    } catch (const DeviceError &e) { errs[m] = e.what(); codes[m] = RS_ERR_DEVICE;
    } catch (const Error &e) { errs[m] = e.what(); codes[m] = RS_ERR_MODEL;
    } catch (const std::exception &e) { errs[m] = e.what(); codes[m] = RS_ERR_ARG; }
  };
  {
    std::vector<std::thread> th;
    int first = -1;
    for (int m = 0; m < n_models; m++) if (!mine[m].empty()) { if (first < 0) first = m; else th.emplace_back(run, m); }
    if (first >= 0) run(first);
    for (auto &t : th) t.join();
  }
  std::vector<int32_t> local((size_t)per * RS_SHARD_RECORD_INTS, 0);
  for (int k = 0; k < per; k++) { local[(size_t)k * RS_SHARD_RECORD_INTS] = -1; local[(size_t)k * RS_SHARD_RECORD_INTS + 1] = RS_SHARD_ABSENT; }
  int rc = RS_OK;
  for (int m = 0; m < n_models; m++) {
    if (codes[m] != RS_OK && rc == RS_OK) { rc = codes[m]; if (error) *error = errs[m]; }
    for (size_t k = 0; k < mine[m].size(); k++) {
      const int i = mine[m][k];
      int32_t *rec = &local[(size_t)((i - rank) / world) * RS_SHARD_RECORD_INTS];
      if (codes[m] != RS_OK) { std::memset(rec, 0, sizeof(int32_t) * RS_SHARD_RECORD_INTS); rec[0] = i; rec[1] = codes[m]; }
      else FillRecord(rec, i, results[m]->utts[k]);
    }
  }
  // ---- the exchange step
  if (comm == nullptr) {     // one rank without a communicator, or the caller gathers later (rs_shard_gather) or by other means
    InitRecords(records, n_utts);
    Scatter(local.data(), per, n_utts, records);
    return rc;
  }
  std::lock_guard<std::mutex> lk(gb.mu);
  // again, under the lock that is held until the gather is over: another thread's call may have re-sized the buffers since (for
  // another device, or a larger `per` with a smaller world: both capacities are checked; nothing happens when they suffice)
  PrepareGather(gb, models[0]->m->opts().device_id, send_bytes, send_bytes * world);
  Gather(gb, local.data(), per, n_utts, world, comm, records);
  return rc;
}

// The exchange step alone: `records` holds this rank's records at their utterance indices (what rs_decode_batch_sharded leaves
// when it is called without a communicator); on return it holds every rank's.  For callers that keep several decode calls in
// flight: the collectives of one communicator have to be issued in the same order on every rank, so such a caller decodes from
// its worker threads and gathers from one thread in step order.
void ShardGather(int device_id, int n_utts, int rank, int world, void *comm, int32_t *records) {
  CheckComm(comm, rank, world, "rs_shard_gather");
  const int per = (n_utts + world - 1) / world;
  const size_t send_bytes = (size_t)per * RS_SHARD_RECORD_INTS * sizeof(int32_t);
  GatherBuffers &gb = Buffers();
  std::lock_guard<std::mutex> lk(gb.mu);
  PrepareGather(gb, device_id, send_bytes, send_bytes * world);
  std::vector<int32_t> local((size_t)per * RS_SHARD_RECORD_INTS, 0);
  for (int k = 0; k < per; k++) {
    int32_t *rec = &local[(size_t)k * RS_SHARD_RECORD_INTS];
    const int i = rank + k * world;
    if (i < n_utts) std::memcpy(rec, records + (size_t)i * RS_SHARD_RECORD_INTS, sizeof(int32_t) * RS_SHARD_RECORD_INTS);
    else { rec[0] = -1; rec[1] = RS_SHARD_ABSENT; }
  }
  Gather(gb, local.data(), per, n_utts, world, comm, records);
}

}  // namespace rs
