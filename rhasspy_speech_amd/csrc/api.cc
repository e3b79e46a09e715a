// extern "C" surface of librhasspy_speech_hip.so (see include/rhasspy_speech_hip.h).  No exception crosses
// the boundary: every failure becomes a status code + thread-local message, mirroring how the reference's
// binaries report KALDI_ERR text on stderr with a non-zero exit status (tools.py:138-145).
#include <sched.h>

#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "engine.h"
#include "fuzzy.h"
#include "graph_build.h"
#include "nnet3_setup.h"
#include "rescore.h"

// Loaded before the HIP runtime has started (the usual case for a host program that links or dlopens the library first): ask the
// runtime for 8 hardware queues instead of 4, unless the environment already says otherwise.  A model keeps up to four calls in
// flight on three streams each, and streams that share a hardware queue serialise on each other's event waits (headline step
// 2.45 -> 2.31 ms).  Without effect when HIP is already initialised.  This changes the environment of the host
// PROCESS (every HIP user in it sees the variable) from inside dlopen: RS_NO_HW_QUEUES_DEFAULT=1 switches it off, a host that sets
// GPU_MAX_HW_QUEUES itself is left alone either way; INTEGRATION.md section 4 says so where an integrator reads first.
namespace {
struct HipQueuesDefault {
  HipQueuesDefault() {
    const char *off = std::getenv("RS_NO_HW_QUEUES_DEFAULT");
    if (!(off && std::atoi(off) != 0)) (void)setenv("GPU_MAX_HW_QUEUES", "8", 0);
  }
} g_hip_queues_default;
}  // namespace

namespace rs {
int DecodeBatchSharded(rs_model *const *models, int n_models, const int32_t *utt_model, const int16_t *const *pcm, const int32_t *n_samples,
                       int n_utts, int rank, int world, void *comm, int32_t *records, std::string *error);
void ShardGather(int device_id, int n_utts, int rank, int world, void *comm, int32_t *records);
}

namespace {
thread_local std::string g_last_error;

template <typename F>
int Guard(F &&f) {
  try {
    g_last_error.clear();
    return f();
  } catch (const rs::DeviceError &e) {
    g_last_error = e.what();
    return RS_ERR_DEVICE;
  } catch (const rs::Error &e) {
    g_last_error = e.what();
    return RS_ERR_MODEL;
  } catch (const std::exception &e) {
    g_last_error = std::string("unexpected error: ") + e.what();
    return RS_ERR_ARG;
  }
}
int ArgError(const char *msg) {
  g_last_error = msg;
  return RS_ERR_ARG;
}
}  // namespace

extern "C" {

int rs_default_opts(rs_decode_opts *o) {
  if (!o) return ArgError("rs_default_opts: null pointer");
  std::memset(o, 0, sizeof(*o));
  o->beam = 24.0f;           // transcribe_wav.py:24
  o->max_active = 7000;      // transcribe_wav.py:21
  o->min_active = RS_OPT_UNSET;       // not on rhasspy's command line: online.conf's value, else 200 (lattice-faster-decoder.h:61)
  o->lattice_beam = 8.0f;    // transcribe_wav.py:22
  o->beam_delta = (float)RS_OPT_UNSET;         // ... else 0.5 (lattice-faster-decoder.h:66)
  o->acoustic_scale = 1.0f;  // transcribe_wav.py:53 ("--acoustic-scale=1.0")
  o->frames_per_chunk = RS_OPT_UNSET;          // ... else 24 (decodable-simple-looped.h:57)
  o->frame_subsampling_factor = RS_OPT_UNSET;  // ... else 1  (decodable-simple-looped.h:56)
  o->device_id = 0;
  // The search reads log-likelihoods only for the pdfs that occur on HCLG arcs (decodable-online-looped.cc:213-224): the output
  // layer is cut down to those rows when that saves at least 30 % of it -- same words, same costs (engine.cc: PruneOutputLayer).
  // Off automatically with keep_intermediates and for nets that end in a log-softmax; 0 computes every pdf like the reference.
  o->prune_output_pdfs = 1;
  o->command_line_fixed = RS_FIXED_ONLINE | RS_FIXED_DO_ENDPOINTING;      // transcribe_wav.py:48-49 ("--online=false", "--do-endpointing=false")
  return RS_OK;
}

const char *rs_last_error(void) { return g_last_error.c_str(); }

int rs_model_load_files(const char *final_mdl, const char *hclg_fst, const char *online_conf, const rs_decode_opts *opts,
                        rs_model **out) {
  if (!final_mdl || !hclg_fst || !online_conf || !out) return ArgError("rs_model_load_files: null argument");
  return Guard([&]() {
    rs_decode_opts o;
    if (opts) o = *opts; else rs_default_opts(&o);
    rs_model *m = new rs_model();
    try {
      m->m.reset(new rs::Model(final_mdl, hclg_fst, online_conf, o));
    } catch (...) {
      delete m;
      throw;
    }
    *out = m;
    return RS_OK;
  });
}

int rs_model_load(const char *model_dir, const char *graph_dir, const rs_decode_opts *opts, rs_model **out) {
  if (!model_dir || !graph_dir || !out) return ArgError("rs_model_load: null argument");
  std::string md(model_dir), gd(graph_dir);
  return rs_model_load_files((md + "/model/model/final.mdl").c_str(), (gd + "/HCLG.fst").c_str(),
                             (md + "/model/online/conf/online.conf").c_str(), opts, out);
}

int rs_model_to_device(rs_model *model) {
  if (!model) return ArgError("rs_model_to_device: null model");
  return Guard([&]() { model->m->ToDevice(); return RS_OK; });
}

void rs_model_free(rs_model *model) { delete model; }

int rs_model_describe(const rs_model *model, char *buf, size_t len) {
  if (!model) return ArgError("rs_model_describe: null model");
  std::string d = model->m->Describe();
  if (buf && len) {
    size_t n = d.size() < len - 1 ? d.size() : len - 1;
    std::memcpy(buf, d.data(), n);
    buf[n] = 0;
  }
  return (int)d.size();
}

// OnlineGenericBaseFeature<C>::MaybeCreateResampler (feat/online-feature.cc:86-101)
int rs_model_check_sample_rate(const rs_model *model, float sample_rate) {
  if (!model) return ArgError("rs_model_check_sample_rate: null model");
  return Guard([&]() {
    const rs::MfccOptions &o = model->m->features().mfcc.opts;
    if (sample_rate == o.samp_freq) return RS_OK;
    std::ostringstream e;
    if ((sample_rate > o.samp_freq && o.allow_downsample) || (sample_rate < o.samp_freq && o.allow_upsample)) {
      e << "Sampling frequency mismatch, expected " << o.samp_freq << ", got " << sample_rate << ": the model's mfcc.conf sets --allow-"
        << (sample_rate > o.samp_freq ? "downsample" : "upsample") << ", with which the reference resamples the waveform (LinearResample); "
        << "this library does not resample: convert the audio to " << o.samp_freq << " Hz first";
    } else {
      e << "Sampling frequency mismatch, expected " << o.samp_freq << ", got " << sample_rate << "\nPerhaps you want to use the options --allow_{upsample,downsample}";
    }
    rs::Fail(e.str());
    return RS_OK;
  });
}

int rs_decode_batch(rs_model *model, const int16_t *const *pcm, const int32_t *n_samples, int32_t n_utts, int32_t nbest,
                    float lattice_acoustic_scale, rs_result **out) {
  if (!model || !out || n_utts < 0 || (n_utts > 0 && (!pcm || !n_samples))) return ArgError("rs_decode_batch: bad argument");
  return Guard([&]() {
    auto r = model->m->DecodeBatchHost(pcm, n_samples, n_utts, nbest, lattice_acoustic_scale);
    rs_result *res = new rs_result();
    res->r = std::move(r);
    *out = res;
    return RS_OK;
  });
}

int rs_decode_batch_device(rs_model *model, const int16_t *d_pcm, const int64_t *sample_offsets, int32_t n_utts, int32_t nbest,
                           float lattice_acoustic_scale, void *stream, rs_result **out) {
  if (!model || !out || n_utts < 0 || (n_utts > 0 && (!d_pcm || !sample_offsets))) return ArgError("rs_decode_batch_device: bad argument");
  return Guard([&]() {
    auto r = model->m->DecodeBatchDevice(d_pcm, sample_offsets, n_utts, nbest, lattice_acoustic_scale, (hipStream_t)stream);
    rs_result *res = new rs_result();
    res->r = std::move(r);
    *out = res;
    return RS_OK;
  });
}

int rs_decode_batch_sharded(rs_model *const *models, int32_t n_models, const int32_t *utt_model, const int16_t *const *pcm,
                            const int32_t *n_samples, int32_t n_utts, int32_t rank, int32_t world, void *rccl_comm, int32_t *records) {
  if (!models || n_models <= 0 || n_utts < 0 || world < 1 || rank < 0 || rank >= world || !records ||
      (n_utts > 0 && (!utt_model || !pcm || !n_samples)))
    return ArgError("rs_decode_batch_sharded: bad argument");
  for (int m = 0; m < n_models; m++) if (!models[m]) return ArgError("rs_decode_batch_sharded: null model");
  for (int i = 0; i < n_utts; i++)
    if (utt_model[i] < 0 || utt_model[i] >= n_models) return ArgError("rs_decode_batch_sharded: utt_model entry names no model");
  return Guard([&]() {
    std::string err;
    const int rc = rs::DecodeBatchSharded(models, n_models, utt_model, pcm, n_samples, n_utts, rank, world, rccl_comm, records, &err);
    if (rc != RS_OK) g_last_error = err;
    return rc;
  });
}

int rs_shard_gather(int32_t device_id, int32_t n_utts, int32_t rank, int32_t world, void *rccl_comm, int32_t *records) {
  if (n_utts < 0 || world < 1 || rank < 0 || rank >= world || !rccl_comm || (n_utts > 0 && !records) || device_id < 0)
    return ArgError("rs_shard_gather: bad argument");
  return Guard([&]() {
    rs::ShardGather(device_id, n_utts, rank, world, rccl_comm, records);
    return RS_OK;
  });
}

int rs_bind_host_thread(int32_t device_id) {
  return Guard([&]() {
    std::string list;
    if (const char *e = std::getenv("RS_BIND_CPULIST")) {
      list = e;
    } else {
      if (device_id < 0) return ArgError("rs_bind_host_thread: bad device id");
      char bus[64] = {0};
      const hipError_t he = hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device_id);
      if (he != hipSuccess) throw rs::DeviceError(std::string("rs_bind_host_thread: hipDeviceGetPCIBusId: ") + hipGetErrorString(he));
      std::string id(bus);
      for (char &c : id) c = (char)std::tolower((unsigned char)c);
      if (FILE *f = std::fopen(("/sys/bus/pci/devices/" + id + "/local_cpulist").c_str(), "r")) {
        char buf[4096] = {0};
        if (std::fgets(buf, sizeof(buf), f)) list = buf;
        std::fclose(f);
      }
    }
    // "a-b,c,d-e"
    cpu_set_t set;
    CPU_ZERO(&set);
    int n = 0;
    const char *p = list.c_str();
    while (*p) {
      while (*p == ',' || *p == ' ' || *p == '\n' || *p == '\t') p++;
      if (!*p) break;
      char *end = nullptr;
      const long a = std::strtol(p, &end, 10);
      if (end == p || a < 0) return ArgError("rs_bind_host_thread: cannot parse the CPU list");
      long b = a;
      p = end;
      if (*p == '-') {
        b = std::strtol(p + 1, &end, 10);
        if (end == p + 1 || b < a) return ArgError("rs_bind_host_thread: cannot parse the CPU list");
        p = end;
      }
      for (long c = a; c <= b && c < CPU_SETSIZE; c++) if (!CPU_ISSET((int)c, &set)) { CPU_SET((int)c, &set); n++; }
    }
    if (n == 0) return 0;
    if (sched_setaffinity(0, sizeof(set), &set) != 0) {
      g_last_error = std::string("rs_bind_host_thread: sched_setaffinity: ") + std::strerror(errno);
      return (int)RS_ERR_ARG;
    }
    return n;
  });
}

int rs_stream_open(rs_model *model, rs_stream **out) {
  if (!model || !out) return ArgError("rs_stream_open: null argument");
  return Guard([&]() {
    std::unique_ptr<rs_stream> st(new rs_stream());
    st->model = model;
    // RS_STREAM_BATCH=1: the cross-check path -- buffer everything, replay the stream as one batch at finish
    const char *e = std::getenv("RS_STREAM_BATCH");
    st->keep_pcm = e && std::atoi(e) != 0;
    if (!st->keep_pcm) model->m->StreamOpen(st.get());
    *out = st.release();
    return RS_OK;
  });
}

static std::string FailedStream(const char *who, const rs_stream *st) {
  return std::string(who) + ": stream was part of an advance that failed; its device state is undefined, close it" +
         (st->fail_why.empty() ? std::string() : " (" + st->fail_why + ")");
}

int rs_stream_accept(rs_stream *stream, const int16_t *pcm, int32_t n_samples) {
  if (!stream || n_samples < 0 || (n_samples > 0 && !pcm)) return ArgError("rs_stream_accept: bad argument");
  if (stream->finished) return ArgError("rs_stream_accept: stream already finished");
  if (stream->failed) return ArgError(FailedStream("rs_stream_accept", stream).c_str());
  return Guard([&]() {
    stream->pcm.insert(stream->pcm.end(), pcm, pcm + n_samples);
    stream->n_samples += n_samples;
    return RS_OK;
  });
}

int rs_streams_accept(rs_stream *const *streams, const int16_t *const *pcm, const int32_t *n_samples, int32_t n_streams) {
  if (n_streams < 0 || (n_streams > 0 && (!streams || !pcm || !n_samples))) return ArgError("rs_streams_accept: bad argument");
  for (int i = 0; i < n_streams; i++) {
    if (!streams[i] || n_samples[i] < 0 || (n_samples[i] > 0 && !pcm[i])) return ArgError("rs_streams_accept: bad argument");
    if (streams[i]->finished) return ArgError("rs_streams_accept: stream already finished");
    if (streams[i]->failed) return ArgError(FailedStream("rs_streams_accept", streams[i]).c_str());
  }
  return Guard([&]() {
    for (int i = 0; i < n_streams; i++) {
      streams[i]->pcm.insert(streams[i]->pcm.end(), pcm[i], pcm[i] + n_samples[i]);
      streams[i]->n_samples += n_samples[i];
    }
    return RS_OK;
  });
}

static int CheckStreams(rs_stream *const *streams, int32_t n_streams, const char *who) {
  if (n_streams < 0 || (n_streams > 0 && !streams)) return ArgError((std::string(who) + ": bad argument").c_str());
  for (int i = 0; i < n_streams; i++) {
    if (!streams[i] || !streams[i]->model) return ArgError((std::string(who) + ": null stream").c_str());
    if (streams[i]->model != streams[0]->model) return ArgError((std::string(who) + ": all streams must belong to one model").c_str());
    if (streams[i]->finished) return ArgError((std::string(who) + ": stream already finished").c_str());
    if (streams[i]->failed) return ArgError(FailedStream(who, streams[i]).c_str());
    if (streams[i]->keep_pcm != streams[0]->keep_pcm) return ArgError((std::string(who) + ": streams opened in different modes").c_str());
    for (int j = 0; j < i; j++) if (streams[j] == streams[i]) return ArgError((std::string(who) + ": a stream is listed twice").c_str());
  }
  return RS_OK;
}

int rs_streams_advance(rs_stream *const *streams, int32_t n_streams) {
  const int rc = CheckStreams(streams, n_streams, "rs_streams_advance");
  if (rc != RS_OK) return rc;
  if (n_streams == 0 || streams[0]->keep_pcm) { g_last_error.clear(); return RS_OK; }
  return Guard([&]() {
    // An advance that throws (pool exhausted for a later stream, arena growth, a HIP error) has already moved the chunk schedule
    // of some streams forward without their iVector / log-likelihood rows being written: none of the call's streams may go on.
    try {
      streams[0]->model->m->StreamsAdvance(streams, n_streams, /*final=*/false, 1, 1.0f, nullptr);
    } catch (...) {
      for (int i = 0; i < n_streams; i++) streams[i]->failed = true;
      throw;
    }
    return RS_OK;
  });
}

int rs_streams_finish(rs_stream *const *streams, int32_t n_streams, int32_t nbest, float lattice_acoustic_scale, rs_result **out) {
  if (!out) return ArgError("rs_streams_finish: bad argument");
  const int rc = CheckStreams(streams, n_streams, "rs_streams_finish");
  if (rc != RS_OK) return rc;
  if (n_streams == 0) return ArgError("rs_streams_finish: no streams");
  if (nbest < 1) return ArgError("rs_streams_finish: nbest must be >= 1");
  return Guard([&]() {
    std::unique_ptr<rs_result> res(new rs_result());
    if (streams[0]->keep_pcm) {
      std::vector<const int16_t *> ptr(n_streams);
      std::vector<int32_t> len(n_streams);
      for (int i = 0; i < n_streams; i++) { ptr[i] = streams[i]->pcm.data(); len[i] = (int32_t)streams[i]->pcm.size(); }
      res->r = streams[0]->model->m->DecodeBatchHost(ptr.data(), len.data(), n_streams, nbest, lattice_acoustic_scale, /*streaming=*/true);
    } else {
      res->r.reset(new rs::Result());
      for (int i = 0; i < n_streams; i++) streams[i]->finished = true;      // whatever happens below, these streams are over
      streams[0]->model->m->StreamsAdvance(streams, n_streams, /*final=*/true, nbest, lattice_acoustic_scale, res->r.get());
    }
    for (int i = 0; i < n_streams; i++) {
      streams[i]->finished = true;
      std::vector<int16_t>().swap(streams[i]->pcm);
      if (!streams[i]->keep_pcm) streams[i]->model->m->StreamClose(streams[i]);
    }
    *out = res.release();
    return RS_OK;
  });
}

int rs_stream_finish(rs_stream *stream, int32_t nbest, float lattice_acoustic_scale, rs_result **out) {
  rs_stream *one[1] = {stream};
  return rs_streams_finish(one, 1, nbest, lattice_acoustic_scale, out);
}

void rs_stream_free(rs_stream *stream) {
  if (!stream) return;
  if (stream->open && stream->model) { try { stream->model->m->StreamClose(stream); } catch (...) {} }
  delete stream;
}

int32_t rs_result_num_utts(const rs_result *r) { return r ? (int32_t)r->r->utts.size() : 0; }

static const rs::UttResult *Utt(const rs_result *r, int32_t utt) {
  if (!r || utt < 0 || utt >= (int32_t)r->r->utts.size()) return nullptr;
  return &r->r->utts[utt];
}

int32_t rs_result_num_hyps(const rs_result *r, int32_t utt) {
  const rs::UttResult *u = Utt(r, utt);
  return u ? (int32_t)u->hyps.size() : 0;
}
int32_t rs_result_num_frames(const rs_result *r, int32_t utt) {
  const rs::UttResult *u = Utt(r, utt);
  return u ? u->num_frames : 0;
}

int rs_result_words(const rs_result *r, int32_t utt, int32_t k, const int32_t **ids, int32_t *n) {
  const rs::UttResult *u = Utt(r, utt);
  if (!u || !ids || !n) return ArgError("rs_result_words: bad argument");
  if (u->status != RS_OK) { g_last_error = u->error; return u->status; }
  if (k < 0 || k >= (int32_t)u->hyps.size()) return ArgError("rs_result_words: hypothesis index out of range");
  *ids = u->hyps[k].words.data();
  *n = (int32_t)u->hyps[k].words.size();
  return RS_OK;
}

int rs_result_costs(const rs_result *r, int32_t utt, int32_t k, float *graph_cost, float *acoustic_cost) {
  const rs::UttResult *u = Utt(r, utt);
  if (!u) return ArgError("rs_result_costs: bad argument");
  if (u->status != RS_OK) { g_last_error = u->error; return u->status; }
  if (k < 0 || k >= (int32_t)u->hyps.size()) return ArgError("rs_result_costs: hypothesis index out of range");
  if (graph_cost) *graph_cost = u->hyps[k].graph_cost;
  if (acoustic_cost) *acoustic_cost = u->hyps[k].acoustic_cost;
  return RS_OK;
}

int rs_result_text(const rs_result *r, int32_t utt, const char *key, char *buf, size_t len) {
  const rs::UttResult *u = Utt(r, utt);
  if (!u) return ArgError("rs_result_text: bad argument");
  if (u->status != RS_OK) { g_last_error = u->error; return u->status; }
  // BasicVectorHolder text form as nbest-to-linear writes it: "<key>-<k> id id ... \n" (trailing space before newline)
  std::string s;
  const std::string ky = key ? key : "utt";
  for (size_t k = 0; k < u->hyps.size(); k++) {
    s += ky + "-" + std::to_string(k + 1) + " ";
    for (int32_t w : u->hyps[k].words) s += std::to_string(w) + " ";
    s += "\n";
  }
  if (buf && len) {
    size_t n = s.size() < len - 1 ? s.size() : len - 1;
    std::memcpy(buf, s.data(), n);
    buf[n] = 0;
  }
  return (int)s.size();
}

int64_t rs_result_lattice(const rs_result *r, int32_t utt, const char *key, char *buf, int64_t cap) {
  const rs::UttResult *u = Utt(r, utt);
  if (!u || cap < 0 || (cap > 0 && !buf)) return ArgError("rs_result_lattice: bad argument");
  if (u->status != RS_OK) { g_last_error = u->error; return u->status; }
  if (!u->clat) return ArgError("rs_result_lattice: the model was not opened with rs_decode_opts.emit_lattice = 1");
  try {
    const std::string s = rs::CompactLatticeArkEntry(key ? key : "utt", *u->clat);
    if (buf && cap) std::memcpy(buf, s.data(), std::min<size_t>(s.size(), (size_t)cap));
    return (int64_t)s.size();
  } catch (const std::exception &e) {
    g_last_error = e.what();
    return RS_ERR_DECODE;
  }
}

int rs_result_pack(const rs_result *r, int32_t max_words, int32_t *out) {
  if (!r || !out || max_words < 0) return ArgError("rs_result_pack: bad argument");
  const int stride = max_words + 4;
  for (size_t u = 0; u < r->r->utts.size(); u++) {
    const rs::UttResult &ur = r->r->utts[u];
    int32_t *rec = out + u * stride;
    std::memset(rec, 0, sizeof(int32_t) * stride);
    rec[0] = ur.status;
    if (ur.status != RS_OK || ur.hyps.empty()) { rec[1] = -1; continue; }
    const rs::Hypothesis &h = ur.hyps[0];
    const int n = (int)std::min<size_t>(h.words.size(), (size_t)max_words);
    rec[1] = n;
    std::memcpy(rec + 2, h.words.data(), sizeof(int32_t) * n);
    std::memcpy(rec + 2 + max_words, &h.graph_cost, 4);
    std::memcpy(rec + 3 + max_words, &h.acoustic_cost, 4);
  }
  g_last_error.clear();
  return RS_OK;
}

int rs_result_matrix(const rs_result *r, int32_t utt, int32_t kind, const float **data, int32_t *rows, int32_t *cols) {
  const rs::UttResult *u = Utt(r, utt);
  if (!u || !data || !rows || !cols) return ArgError("rs_result_matrix: bad argument");
  const std::vector<float> *v = nullptr;
  int c = 0, rws = 0;
  if (kind == 0) { v = &u->feats; c = u->feat_dim; rws = c ? (int)(v->size() / c) : 0; }
  else if (kind == 1) { v = &u->ivector; c = u->ivec_dim; rws = c ? (int)(v->size() / c) : 0; }
  else if (kind == 2) { v = &u->loglikes; c = u->num_pdfs; rws = c ? (int)(v->size() / c) : 0; }
  else return ArgError("rs_result_matrix: unknown kind");
  if (v->empty() && c == 0) return ArgError("rs_result_matrix: intermediates were not kept (set rs_decode_opts.keep_intermediates)");
  *data = v->data();
  *rows = rws;
  *cols = c;
  return RS_OK;
}

int rs_result_counters(const rs_result *r, int32_t utt, int64_t out[8]) {
  const rs::UttResult *u = Utt(r, utt);
  if (!u || !out) return ArgError("rs_result_counters: bad argument");
  for (int i = 0; i < 8; i++) out[i] = u->counters[i];
  return RS_OK;
}

int rs_result_timings(const rs_result *r, float out[8]) {
  if (!r || !out) return ArgError("rs_result_timings: bad argument");
  for (int i = 0; i < 8; i++) out[i] = r->r->timings[i];
  return RS_OK;
}

void rs_result_free(rs_result *r) { delete r; }

struct rs_fuzzy {
  rs::FuzzyMatcher m;
  explicit rs_fuzzy(const std::string &path) : m(path) {}
};

int rs_fuzzy_open(const char *fuzzy_fst_path, rs_fuzzy **out) {
  if (!fuzzy_fst_path || !out) return ArgError("rs_fuzzy_open: null argument");
  return Guard([&]() { *out = new rs_fuzzy(fuzzy_fst_path); return RS_OK; });
}

int rs_fuzzy_match(const rs_fuzzy *f, const char *nbest_text, int32_t *olabels, int32_t cap, int32_t *n_out, double *cost) {
  if (!f || !nbest_text || !n_out || !cost || (cap > 0 && !olabels)) return ArgError("rs_fuzzy_match: null argument");
  return Guard([&]() {
    const rs::FuzzyResult r = f->m.Match(nbest_text);
    *cost = r.cost;
    if (!r.matched) { *n_out = -1; return RS_OK; }
    *n_out = (int32_t)r.olabels.size();
    for (int32_t i = 0; i < *n_out && i < cap; i++) olabels[i] = r.olabels[i];
    return RS_OK;
  });
}

int rs_result_fuzzy(const rs_result *r, int32_t utt, const rs_fuzzy *f, int32_t *olabels, int32_t cap, int32_t *n_out, double *cost) {
  const rs::UttResult *u = Utt(r, utt);
  if (!u || !f || !n_out || !cost || (cap > 0 && !olabels)) return ArgError("rs_result_fuzzy: bad argument");
  if (u->status != RS_OK) { g_last_error = u->error; return u->status; }
  std::string s;                       // the hypotheses in rank order, as the fan of get_fuzzy_text reads them
  for (size_t k = 0; k < u->hyps.size(); k++) {
    s += "utt-" + std::to_string(k + 1) + " ";
    for (int32_t w : u->hyps[k].words) s += std::to_string(w) + " ";
    s += "\n";
  }
  return rs_fuzzy_match(f, s.c_str(), olabels, cap, n_out, cost);
}

void rs_fuzzy_free(rs_fuzzy *f) { delete f; }

// ---- graph construction
int rs_mkgraph(const char *lang_dir, const char *model_dir, const char *graph_dir, float transition_scale, float self_loop_scale,
               const char *dump_dir) {
  if (!lang_dir || !model_dir || !graph_dir) return ArgError("rs_mkgraph: null argument");
  return Guard([&]() {
    rs::gb::MkgraphOptions o;
    o.transition_scale = transition_scale;
    o.self_loop_scale = self_loop_scale;
    if (dump_dir) o.dump_dir = dump_dir;
    rs::gb::Mkgraph(lang_dir, model_dir, graph_dir, o);
    return RS_OK;
  });
}

int rs_fst_tool(const char *tool, const char *in1, const char *in2, const char *out, const char *aux, float param) {
  if (!tool) return ArgError("rs_fst_tool: null argument");
  return Guard([&]() {
    namespace gb = rs::gb;
    const std::string t = tool;
    auto need = [&](const char *p, const char *what) { if (!p) rs::Fail("rs_fst_tool " + t + ": missing " + what); return std::string(p); };
    if (t == "fsttablecompose") {
      gb::WriteFst(gb::Compose(gb::ReadFst(need(in1, "in1")), gb::ReadFst(need(in2, "in2"))), need(out, "out"), false);
    } else if (t == "fstdeterminizestar") {
      gb::WriteFst(gb::DeterminizeStar(gb::ReadFst(need(in1, "in1")), param != 0.0f), need(out, "out"), false);
    } else if (t == "fstminimizeencoded") {
      gb::Fst f = gb::ReadFst(need(in1, "in1"));
      gb::MinimizeEncoded(&f);
      gb::WriteFst(f, need(out, "out"), false);
    } else if (t == "fstpushspecial") {
      gb::Fst f = gb::ReadFst(need(in1, "in1"));
      gb::PushSpecial(&f);
      gb::WriteFst(f, need(out, "out"), false);
    } else if (t == "fstrmsymbols") {
      gb::Fst f = gb::ReadFst(need(in1, "in1"));
      gb::RemoveInputSymbols(&f, gb::ReadIntList(need(aux, "symbol list")));
      gb::WriteFst(f, need(out, "out"), false);
    } else if (t == "fstrmepslocal") {
      gb::Fst f = gb::ReadFst(need(in1, "in1"));
      gb::RemoveEpsLocal(&f, true);
      gb::WriteFst(f, need(out, "out"), false);
    } else if (t == "fstarcsort") {
      gb::Fst f = gb::ReadFst(need(in1, "in1"));
      gb::ArcSort(&f, !(aux && std::string(aux) == "olabel"));
      gb::WriteFst(f, need(out, "out"), false);
    } else if (t == "fstcomposecontext") {
      // in2 = disambig list, aux = ilabels output, param = context width * 16 + central position
      std::vector<std::vector<int32_t>> ilabels;
      const int code = (int)param;
      gb::Fst f = gb::ComposeContext(gb::ReadIntList(need(in2, "disambig list")), code / 16, code % 16, gb::ReadFst(need(in1, "in1")), &ilabels);
      gb::WriteILabelInfo(ilabels, need(aux, "ilabels output"));
      gb::WriteFst(f, need(out, "out"), false);
    } else if (t == "make-h-transducer") {
      // in1 = ilabels, in2 = tree, aux = final.mdl, out = Ha.fst (+ out + ".disambig" = the disambiguation transition-ids), param = transition scale
      gb::ContextDependency tree;
      tree.Read(need(in2, "tree"));
      rs::TransitionModel tm;
      { rs::KaldiReader r(need(aux, "model")); tm.Read(r); }
      std::vector<int32_t> dis;
      gb::Fst h = gb::MakeHTransducer(gb::ReadILabelInfo(need(in1, "ilabels")), tree, tm, param, &dis);
      gb::WriteFst(h, need(out, "out"), false);
      std::string list;
      for (int32_t d : dis) list += std::to_string(d) + "\n";
      FILE *fp = std::fopen((std::string(out) + ".disambig").c_str(), "w");
      if (!fp) rs::Fail("make-h-transducer: cannot write the disambiguation symbols");
      std::fputs(list.c_str(), fp);
      std::fclose(fp);
    } else if (t == "add-self-loops") {
      rs::TransitionModel tm;
      { rs::KaldiReader r(need(aux, "model")); tm.Read(r); }
      gb::Fst f = gb::ReadFst(need(in1, "in1"));
      gb::AddSelfLoops(tm, param, &f);
      gb::WriteFst(f, need(out, "out"), false);
    } else if (t == "fstisomorphic" || t == "fstequivalent") {
      const gb::Fst a = gb::ReadFst(need(in1, "in1")), b = gb::ReadFst(need(in2, "in2"));
      const std::string d = t == "fstisomorphic" ? gb::Isomorphic(a, b, param) : gb::RandEquivalent(a, b, param, 200, 40, 1);
      if (!d.empty()) rs::Fail(t + ": " + d);
    } else {
      rs::Fail("rs_fst_tool: unknown tool " + t);
    }
    return RS_OK;
  });
}

int rs_nnet3_setup(const char *final_mdl, int32_t frames_per_chunk, int64_t *rand_calls, int32_t *certain, char *collapsed_config,
                   size_t buf_len) {
  return rs_nnet3_setup_subsampled(final_mdl, frames_per_chunk, 1, rand_calls, certain, collapsed_config, buf_len);
}

int rs_nnet3_setup_subsampled(const char *final_mdl, int32_t frames_per_chunk, int32_t frame_subsampling_factor, int64_t *rand_calls,
                              int32_t *certain, char *collapsed_config, size_t buf_len) {
  if (!final_mdl || !rand_calls) return ArgError("rs_nnet3_setup: null argument");
  return Guard([&]() {
    // the network section of final.mdl, read the way Model does, without compiling a layer plan
    rs::KaldiReader r(final_mdl);
    rs::TransitionModel trans;
    trans.Read(r);
    r.ExpectToken("<Nnet3>");
    (void)r.ReadLine();
    std::vector<std::string> cfg;
    for (;;) {
      if (r.RawEof()) rs::Fail(std::string(final_mdl) + ": EOF inside <Nnet3> config section");
      std::string line = r.ReadLine();
      if (line.empty()) break;
      cfg.push_back(line);
    }
    std::vector<std::string> names;
    std::vector<rs::Component> comps;
    rs::ReadNnetComponents(r, &names, &comps);
    const rs::Nnet3SetupResult su = rs::Nnet3Setup(cfg, &names, &comps, frames_per_chunk > 0 ? frames_per_chunk : 24, 0, frame_subsampling_factor > 0 ? frame_subsampling_factor : 1);
    *rand_calls = su.rand_calls;
    if (certain) *certain = su.rand_calls_certain ? 1 : 0;
    if (collapsed_config && buf_len) {
      std::string s;
      for (auto &l : su.config) s += l + "\n";
      const size_t n = std::min(buf_len - 1, s.size());
      std::memcpy(collapsed_config, s.data(), n);
      collapsed_config[n] = 0;
    }
    return RS_OK;
  });
}

int rs_dither_noise(int64_t rand_calls, int32_t t0, int32_t t1, int32_t window, float *out) {
  if (!out || t0 < 0 || t1 < t0 || window <= 0 || rand_calls < 0) return ArgError("rs_dither_noise: bad argument");
  return Guard([&]() { rs::DitherNoise((long)rand_calls, t0, t1, window, out); return RS_OK; });
}

struct rs_rescorer {
  rs::Rescorer r;
  const rs_model *model;
  rs_rescorer(const std::string &dir, const rs_model *m) : r(dir), model(m) {}
};

int rs_rescorer_open(const rs_model *model, const char *new_lang_dir, rs_rescorer **out) {
  if (!model || !new_lang_dir || !out) return ArgError("rs_rescorer_open: null argument");
  return Guard([&]() { *out = new rs_rescorer(new_lang_dir, model); return RS_OK; });
}

static int RenderRescored(const std::vector<rs::NbestPath> &paths, const char *key, char *buf, size_t len, float *graph_cost,
                          float *acoustic_cost, int32_t *n_out) {
  std::string s;
  const std::string ky = key ? key : "utt";
  for (size_t k = 0; k < paths.size(); k++) {
    s += ky + "-" + std::to_string(k + 1) + " ";
    for (int32_t w : paths[k].words) s += std::to_string(w) + " ";
    s += "\n";
    if (graph_cost) graph_cost[k] = (float)paths[k].graph_cost;
    if (acoustic_cost) acoustic_cost[k] = (float)paths[k].acoustic_cost;
  }
  if (n_out) *n_out = (int32_t)paths.size();
  if (buf && len) {
    const size_t n = s.size() < len - 1 ? s.size() : len - 1;
    std::memcpy(buf, s.data(), n);
    buf[n] = 0;
  }
  return (int)s.size();
}

int rs_rescore_result(const rs_rescorer *r, const rs_result *res, int32_t utt, int32_t nbest, float acoustic_scale, const char *key,
                      char *buf, size_t len, float *graph_cost, float *acoustic_cost, int32_t *n_out) {
  const rs::UttResult *u = Utt(res, utt);
  if (!r || !u || nbest < 1) return ArgError("rs_rescore_result: bad argument");
  if (u->status != RS_OK) { g_last_error = u->error; return u->status; }
  if (!u->clat) return ArgError("rs_rescore_result: the model was not opened with rs_decode_opts.emit_lattice = 1");
  return Guard([&]() { return RenderRescored(r->r.Rescore(*u->clat, r->model->m->am().trans, nbest, acoustic_scale), key, buf, len, graph_cost, acoustic_cost, n_out); });
}

int rs_rescore_lattice(const rs_rescorer *r, const char *lattice_entry, size_t n_bytes, int32_t nbest, float acoustic_scale, const char *key,
                       char *buf, size_t len, float *graph_cost, float *acoustic_cost, int32_t *n_out) {
  if (!r || !lattice_entry || nbest < 1) return ArgError("rs_rescore_lattice: bad argument");
  return Guard([&]() {
    const rs::CompactLat clat = rs::ParseCompactLatticeEntry(lattice_entry, n_bytes, nullptr);
    return RenderRescored(r->r.Rescore(clat, r->model->m->am().trans, nbest, acoustic_scale), key, buf, len, graph_cost, acoustic_cost, n_out);
  });
}

void rs_rescorer_free(rs_rescorer *r) { delete r; }

int64_t rs_lattice_entry_from_raw(int32_t num_states, int32_t start, const float *final_cost, int32_t n_arcs, const int32_t *arc_src,
                                  const int32_t *arc_dst, const int32_t *arc_word, const int32_t *arc_tid, const float *arc_graph,
                                  const float *arc_acoustic, float beam, const char *key, char *buf, int64_t cap) {
  if (num_states <= 0 || start < 0 || start >= num_states || !final_cost || n_arcs < 0 || cap < 0 || (cap > 0 && !buf) ||
      (n_arcs > 0 && (!arc_src || !arc_dst || !arc_word || !arc_tid || !arc_graph || !arc_acoustic)))
    return ArgError("rs_lattice_entry_from_raw: bad argument");
  try {
    rs::RawLattice lat;
    lat.start = start;
    lat.num_states = num_states;
    lat.final_cost.assign(final_cost, final_cost + num_states);
    for (int i = 0; i < n_arcs; i++) {
      if (arc_src[i] < 0 || arc_src[i] >= num_states || arc_dst[i] < 0 || arc_dst[i] >= num_states)
        return ArgError("rs_lattice_entry_from_raw: arc state out of range");
      lat.arcs.push_back({arc_src[i], arc_dst[i], arc_word[i], (double)arc_graph[i], (double)arc_acoustic[i], arc_tid[i]});
    }
    const std::string s = rs::CompactLatticeArkEntry(key ? key : "utt", rs::DeterminizeLattice(lat, beam));
    if (buf && cap) std::memcpy(buf, s.data(), std::min<size_t>(s.size(), (size_t)cap));
    return (int64_t)s.size();
  } catch (const std::exception &e) {
    g_last_error = e.what();
    return RS_ERR_DECODE;
  }
}

}  // extern "C"
