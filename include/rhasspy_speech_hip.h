/* rhasspy_speech_hip.h -- C ABI of librhasspy_speech_hip.so
 *
 * Drop-in boundary for the transcribe hot path of rhasspy/rhasspy-speech.  The reference crosses a
 * *process* boundary here: rhasspy_speech/transcribe_wav.py:45-75 pipes
 *     online2-wav-nnet3-latgen-faster | lattice-to-nbest | nbest-to-linear
 * and rhasspy_speech/transcribe_stream.py:53-99 feeds s16le PCM to online2-cli-nnet3-decode-faster and
 * then runs the same two lattice tools.  Each entry point below names the piece of that contract it
 * replaces.  Plain C types only; every call returns 0 on success and a negative status on failure, with
 * the message available from rs_last_error() (the reference's convention: non-zero exit status + stderr
 * text, rhasspy_speech/tools.py:138-145).  Handles are opaque.  The caller owns every input buffer; the
 * library owns results until rs_result_free().
 *
 * There is NO CPU fallback: model parsing runs on the host, every compute entry point needs an MI355X
 * (gfx950) device and fails with RS_ERR_DEVICE otherwise.
 */
#ifndef RHASSPY_SPEECH_HIP_H_
#define RHASSPY_SPEECH_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RS_OK 0
#define RS_ERR_ARG -1     /* bad argument */
#define RS_ERR_MODEL -2   /* model/graph/config file could not be read (Kaldi's KALDI_ERR paths) */
#define RS_ERR_DEVICE -3  /* no usable HIP device / HIP runtime error */
#define RS_ERR_DECODE -4  /* decoding failed (e.g. no frames) */

typedef struct rs_model rs_model;
typedef struct rs_result rs_result;
typedef struct rs_stream rs_stream;

/* Options = the command-line flags the reference passes (transcribe_wav.py:46-55,
 * transcribe_stream.py:55-60) plus the Kaldi defaults it relies on
 * (decoder/lattice-faster-decoder.h:38-92, nnet3/decodable-simple-looped.h:50-60). */
/* rs_decode_opts is the command line of the reference's binaries.  The reference registers the decoder's and the decodable's
 * options on the parser that reads --config=online.conf (online2-wav-nnet3-latgen-faster.cc:131-137,
 * online2-cli-nnet3-decode-faster.cc:73-78) and ParseOptions reads the config file FIRST, the command line overriding it
 * (util/parse-options.cc:328-345).  Same here: a field left at RS_OPT_UNSET takes online.conf's value if the file sets the option
 * and the reference's default otherwise; any other value wins over online.conf.  rs_default_opts() sets what rhasspy passes on
 * the command line (transcribe_wav.py:46-55: --max-active --lattice-beam --acoustic-scale --beam) and leaves the rest unset.
 * Options of online.conf the kernels cannot honour (--extra-left-context-initial != 0,
 * --prune-interval != 25, --determinize-lattice=false, --online=true, --do-endpointing=true) fail the model load. */
#define RS_OPT_UNSET (-1)
typedef struct rs_decode_opts {
  float beam;                  /* --beam            (24.0 as rhasspy runs it; reference default 16.0) */
  int32_t max_active;          /* --max-active      (7000 as rhasspy runs it; reference default INT32_MAX) */
  int32_t min_active;          /* --min-active      (unset; reference default 200) */
  float lattice_beam;          /* --lattice-beam    (8.0 as rhasspy runs it; reference default 10.0) */
  float beam_delta;            /* --beam-delta      (unset; reference default 0.5) */
  float acoustic_scale;        /* --acoustic-scale of the decodable (1.0 as rhasspy runs it; reference default 0.1) */
  int32_t frames_per_chunk;    /* --frames-per-chunk (unset; reference default 24; only changes streaming iVector timing) */
  int32_t frame_subsampling_factor; /* --frame-subsampling-factor (unset; reference default 1.  f > 1: the decoder's frames are the output
                                     * rows t = 0, f, 2f, ...; frames_per_chunk is rounded up to a multiple of f: nnet-compile-looped.cc:81-94) */
  int32_t device_id;           /* HIP device ordinal */
  int32_t keep_intermediates;  /* 1: results keep features / iVectors / log-likelihoods for parity tests */
  int32_t max_tokens_per_frame;/* capacity of the per-frame token arrays on the device (0 = automatic) */
  int32_t emit_lattice;        /* 1: results keep the determinised lattice (rs_result_lattice); forces the lattice path */
  int32_t prune_output_pdfs;   /* 1 (default): evaluate the output layer only for the pdfs that occur on HCLG arcs (the search
                                * can read no others; transcripts and costs unchanged).  0 computes every pdf like the
                                * reference does.  Ignored with keep_intermediates and when the net ends in a log-softmax. */
  int32_t exact_token_order;   /* 1: the grammar-graph search creates tokens in the reference's order (the running `next_cutoff` of
                                * lattice-faster-decoder.cc:774-787 over its HashList order, hash-list-inl.h:125-165) -- costs equal the
                                * reference's on the frames where min-active / max-active binds, about 40 % more search time; graphs it
                                * does not apply to (more than 1000 states, chained epsilon arcs) are searched as with 0.  Default 0;
                                * RS_EXACT_ORDER=0|1 overrides. */
  int32_t command_line_fixed;  /* RS_FIXED_* bits: options whose only supported value was given ON THE COMMAND LINE (--online=false,
                                * --do-endpointing=false, --extra-left-context-initial=0, --prune-interval=25, --determinize-lattice=true).
                                * ParseOptions reads --config first and the command line overrides it (util/parse-options.cc:328-345), so
                                * an unsupported value of such an option in online.conf is then not an error.  rs_default_opts sets
                                * ONLINE | DO_ENDPOINTING: rhasspy's command line carries both (transcribe_wav.py:48-49). */
  int32_t stream_min_ticks;    /* rs_streams_advance coalescing: a call that brings fewer than this many new 1024-sample ticks (64 ms each) on EVERY
                                * listed stream does nothing and leaves the audio to the next call (the chunk / iVector schedule is a function
                                * of the samples accepted, not of the calls: results are the same for any value).  1: every call advances, the
                                * reference binary's per-tick cadence (online2-cli-nnet3-decode-faster.cc:143-161) -- lowest latency of
                                * rs_streams_finish, most launches; 0 = the library's default, 16 (about a second of audio per advance: the
                                * throughput setting).  RS_STREAM_MIN_TICKS=<n> overrides. */
  int32_t reserved[2];
} rs_decode_opts;
#define RS_FIXED_ONLINE 1
#define RS_FIXED_DO_ENDPOINTING 2
#define RS_FIXED_EXTRA_LEFT_CONTEXT_INITIAL 4
#define RS_FIXED_PRUNE_INTERVAL 8
#define RS_FIXED_DETERMINIZE_LATTICE 16

/* Fills `opts` with the values the reference's Python passes / Kaldi defaults. */
int rs_default_opts(rs_decode_opts *opts);

/* Thread-local message of the last failing call on this thread ("" if none). */
const char *rs_last_error(void);

/* Replaces the per-process model loading of both binaries
 * (online2-wav-nnet3-latgen-faster.cc:150-190: OnlineNnet2FeaturePipelineInfo from --config=online.conf,
 * TransitionModel + AmNnetSimple from final.mdl, ReadFstKaldiGeneric(HCLG.fst)).  Parses on the host only;
 * the device copy is made on first use or by rs_model_to_device().  Immutable afterwards, shareable
 * between threads. */
int rs_model_load_files(const char *final_mdl, const char *hclg_fst, const char *online_conf,
                        const rs_decode_opts *opts, rs_model **out);
/* Same, using the directory layout the reference hard-codes (transcribe_wav.py:43-44,56-57):
 * <model_dir>/model/model/final.mdl, <model_dir>/model/online/conf/online.conf, <graph_dir>/HCLG.fst. */
int rs_model_load(const char *model_dir, const char *graph_dir, const rs_decode_opts *opts, rs_model **out);
int rs_model_to_device(rs_model *model);
void rs_model_free(rs_model *model);
/* Writes a one-line-per-item description of the parsed model (dims, layer plan, graph size) into buf;
 * returns the number of bytes needed (like snprintf). */
int rs_model_describe(const rs_model *model, char *buf, size_t len);
/* The check the reference makes when a waveform arrives (OnlineGenericBaseFeature::MaybeCreateResampler, feat/online-feature.cc:
 * 86-101; online2-wav-nnet3-latgen-faster.cc:233 hands it WaveData::SampFreq()): RS_OK when `sample_rate` is the model's
 * --sample-frequency, else an error whose text is Kaldi's "Sampling frequency mismatch, expected 16000, got 8000 ...".  The PCM entry
 * points below take samples, not files: a caller that read a wav header calls this first (rhasspy_speech_amd.transcribe_wav and the
 * online2-wav-nnet3-latgen-faster shim do).  --allow-downsample / --allow-upsample in mfcc.conf (the reference then resamples) are
 * refused with a message: the library does not resample. */
int rs_model_check_sample_rate(const rs_model *model, float sample_rate);

/* Offline batch decode = N invocations of the reference's 3-process pipeline with --online=false
 * (online2-wav-nnet3-latgen-faster.cc:196-300 + lattice-to-nbest.cc:80-110 + nbest-to-linear.cc:67-87).
 * pcm[i] points to n_samples[i] mono 16 kHz int16 samples (WaveData::Read hands Kaldi the same values as
 * floats, unscaled).  nbest = lattice-to-nbest --n; lattice_acoustic_scale = its --acoustic-scale. */
int rs_decode_batch(rs_model *model, const int16_t *const *pcm, const int32_t *n_samples, int32_t n_utts,
                    int32_t nbest, float lattice_acoustic_scale, rs_result **out);
/* Same with the samples already resident in device memory (HBM): d_pcm is one device buffer holding all
 * utterances back to back, sample_offsets[i] (host array, n_utts+1 entries) delimits utterance i.
 * stream = hipStream_t to run on (NULL = the model's own stream).  This is what bench.py times. */
int rs_decode_batch_device(rs_model *model, const int16_t *d_pcm, const int64_t *sample_offsets,
                           int32_t n_utts, int32_t nbest, float lattice_acoustic_scale, void *stream,
                           rs_result **out);

/* Streaming decode = online2-cli-nnet3-decode-faster (stdin s16le until EOF, :143-161) followed by
 * lattice-to-nbest | nbest-to-linear (transcribe_stream.py:85-99).  Audio may arrive in arbitrary chunk
 * sizes; the library re-chunks to the binary's fixed 1024-sample ticks (:37) so results do not depend on
 * the caller's chunking, exactly like the reference. */
int rs_stream_open(rs_model *model, rs_stream **out);
int rs_stream_accept(rs_stream *stream, const int16_t *pcm, int32_t n_samples);
int rs_stream_finish(rs_stream *stream, int32_t nbest, float lattice_acoustic_scale, rs_result **out);
void rs_stream_free(rs_stream *stream);
/* Many concurrent streams (BASELINE.json config 5).  The reference's streaming result is a deterministic function
 * of the sample sequence: which frames each nnet chunk's iVector has seen follows from the 1024-sample tick
 * schedule alone (decodable-online-looped.cc:56-84,186-194), not from wall-clock time.  The library therefore
 * reproduces it exactly from the samples accepted so far: rs_streams_advance does, batched over the listed streams, all the
 * device work those samples make possible (MFCC of the completed frames, the iVector estimates and nnet chunks of the ticks
 * reached, the search over the new rows -- what online2-cli-nnet3-decode-faster.cc:143-161 does per tick), so that
 * rs_streams_finish (stdin EOF for all listed streams) only has the tail left; result utterance i belongs to streams[i].
 * A call that finds less than 16 ticks (1 s) of new audio on every listed stream leaves them to the next call: an advance is
 * chains of small dependent launches whose length hardly depends on the rows, and nothing but the end of a stream reads its
 * results -- same rows, same results, a third fewer milliseconds per hour of audio (RS_STREAM_MIN_TICKS=1: every call works).
 * An advance that fails leaves the streams it listed unusable (every later call on them except rs_stream_free is refused). */
int rs_streams_accept(rs_stream *const *streams, const int16_t *const *pcm, const int32_t *n_samples, int32_t n_streams);
    /* rs_stream_accept for many streams in one call: pcm[i] / n_samples[i] go to streams[i] (a host that serves hundreds of streams
     * hands over a round of audio per call; the samples are copied, like stdin is read, online2-cli-nnet3-decode-faster.cc:143-148) */
int rs_streams_advance(rs_stream *const *streams, int32_t n_streams);
int rs_streams_finish(rs_stream *const *streams, int32_t n_streams, int32_t nbest, float lattice_acoustic_scale,
                      rs_result **out);

/* Result access.  Hypotheses of utterance `utt` are ordered best first, like the keys utt-1..utt-n that
 * lattice-to-nbest writes (lattice-to-nbest.cc:100-106). */
int32_t rs_result_num_utts(const rs_result *r);
int32_t rs_result_num_hyps(const rs_result *r, int32_t utt);
int32_t rs_result_num_frames(const rs_result *r, int32_t utt);
/* Word ids (HCLG olabels, epsilons removed) of hypothesis k; *ids stays valid until rs_result_free. */
int rs_result_words(const rs_result *r, int32_t utt, int32_t k, const int32_t **ids, int32_t *n);
/* The utterance's lattice as one binary CompactLattice table entry ("<key> " + VectorFst<CompactLatticeArc> with no "\0B" marker,
 * lat/kaldi-lattice.cc:62-70,:478-500, fstext/lattice-weight.h:141-145,:471-475,:532-540): what online2-wav-nnet3-latgen-faster
 * writes to its `ark:` wspecifier (online2-wav-nnet3-latgen-faster.cc:286-300), i.e. the bytes the reference pipes into
 * lattice-to-nbest.  Needs rs_decode_opts.emit_lattice = 1.  The lattice is determinised on words within lattice_beam and
 * is equivalent to the reference's (same word sequences, costs and best alignments), not byte-identical: state numbering
 * is this library's.  Returns the entry's size in bytes (copies min(size, cap) bytes to buf), or a negative error. */
int64_t rs_result_lattice(const rs_result *res, int32_t utt, const char *key, char *buf, int64_t cap);

/* (graph cost, acoustic cost) of hypothesis k = the 4th/5th outputs of nbest-to-linear. */
int rs_result_costs(const rs_result *r, int32_t utt, int32_t k, float *graph_cost, float *acoustic_cost);
/* Renders exactly the bytes `nbest-to-linear ark:- ark:/dev/null ark,t:-` prints for this utterance:
 * "utt-<k> <id> <id> ... \n" per hypothesis (key prefix `key`, "utt" in the reference).  Returns the number
 * of bytes needed, like snprintf. */
int rs_result_text(const rs_result *r, int32_t utt, const char *key, char *buf, size_t len);
/* Fixed-size result records of the best hypothesis of every utterance, for the multi-GPU gather (one RCCL
 * all_gather of these records is the path's only exchange step): out[u * (max_words + 4)] =
 * {status (0 ok), n_words, word ids (max_words slots, truncated), graph cost bits, acoustic cost bits}. */
int rs_result_pack(const rs_result *r, int32_t max_words, int32_t *out);
/* Multi-GPU entry point (BASELINE.json configs[3]: a batch whose utterances name different
 * models, sharded over the GPUs of one node).  The reference runs one process per utterance with no shared state
 * (rhasspy_speech/tools.py:117-147), so the partition is free: utterance i belongs to rank i % world.  Every rank calls this
 * with the same utt_model[] / n_utts (one process per GPU; models[m] resident on this rank's device); pcm[i] / n_samples[i]
 * need only be valid for the rank's own utterances.  The rank decodes its utterances -- one device batch per model, the
 * models' batches concurrently -- and ONE ncclAllGather over `rccl_comm` (an ncclComm_t whose rank / size are `rank` /
 * `world`; RCCL over xGMI) fills `records` (n_utts x RS_SHARD_RECORD_INTS int32) for EVERY utterance on every rank:
 *   [0] utterance index, [1] status (RS_OK, a negative RS_ERR_*, or RS_SHARD_ABSENT), [2] number of words of the 1-best (if it
 *   exceeds RS_SHARD_MAX_WORDS the ids were cut), [3 .. 3+RS_SHARD_MAX_WORDS) word ids, then graph / acoustic cost (float bits).
 * Failures travel in the status field and the collective always runs, so a failing rank cannot leave the others waiting;
 * the return value is this rank's first failure (after the gather).  With rccl_comm = NULL no collective is
 * issued and only the rank's own records are filled (the others read RS_SHARD_ABSENT): for callers that gather themselves. */
#define RS_SHARD_MAX_WORDS 63
#define RS_SHARD_RECORD_INTS (3 + RS_SHARD_MAX_WORDS + 2)   /* 68 int32 = 272 bytes */
#define RS_SHARD_ABSENT 1
int rs_decode_batch_sharded(rs_model *const *models, int32_t n_models, const int32_t *utt_model, const int16_t *const *pcm,
                            const int32_t *n_samples, int32_t n_utts, int32_t rank, int32_t world, void *rccl_comm,
                            int32_t *records);
/* The exchange step of rs_decode_batch_sharded on its own: `records` holds this rank's records at their utterance indices (what
 * rs_decode_batch_sharded leaves when called with rccl_comm = NULL); ONE ncclAllGather over `rccl_comm` later it holds every
 * rank's.  For hosts that keep several decode calls in flight: the collectives of one communicator must be issued in the same
 * order on every rank, so such a host decodes from its worker threads and gathers from one thread in step order.  The
 * reference has no counterpart (one process per utterance, tools.py:117-147); `device_id` = the rank's GPU. */
int rs_shard_gather(int32_t device_id, int32_t n_utts, int32_t rank, int32_t world, void *rccl_comm, int32_t *records);
/* Host-side placement for one-process-per-GPU hosts (no counterpart in the reference, which never shares a box between GPUs):
 * restricts the CALLING thread -- and every thread it creates afterwards, the library's own helper threads included -- to the
 * CPUs local to GPU `device_id` (`/sys/bus/pci/devices/<bus id>/local_cpulist`: the cores of its NUMA node), so that a rank's
 * staging copies (24.6 MB of PCM per 2.3 ms step on the headline workload) read and write node-local memory instead of crossing
 * the socket interconnect at eight ranks.  Returns the number of CPUs in the mask (0: the system names none -- nothing changed),
 * or a negative RS_ERR_*.  RS_BIND_CPULIST=<list> (e.g. "0-15,64-79") overrides the list and needs no device. */
int rs_bind_host_thread(int32_t device_id);

/* Parity taps (only with opts.keep_intermediates): kind 0 = nnet input features (T x C), 1 = iVector
 * (n x D_iv: one row offline, one row per nnet chunk for streams), 2 = log-likelihoods (T x P). */
int rs_result_matrix(const rs_result *r, int32_t utt, int32_t kind, const float **data, int32_t *rows, int32_t *cols);
/* Decoder work counters of one utterance, for the algorithmic-bytes figure of SURVEY.md section 8(d):
 * out[0] = tokens expanded, [1] = arcs examined, [2] = token insertions (FindOrAddToken calls),
 * [3] = tokens alive summed over frames, [4] = lattice arcs after pruning, [5] = frames where
 * max_active bound, [6] = frames where min_active bound, [7] = token-capacity overflows. */
int rs_result_counters(const rs_result *r, int32_t utt, int64_t out[8]);
/* Wall-clock milliseconds of the stages of the call that produced r: out[0] = H2D, [1] = MFCC,
 * [2] = iVector, [3] = nnet, [4] = decode, [5] = lattice+n-best (host), [6] = total.  For a result of rs_streams_finish the
 * stages are summed over the finishing call and the rs_streams_advance calls that preceded it on the model, and out[7] is the
 * wall time of the finishing call alone. */
int rs_result_timings(const rs_result *r, float out[8]);
void rs_result_free(rs_result *r);

/* ---- fuzzy matching of an n-best list against <lang_dir>/G.fuzzy.fst (host side, no GPU involved).
 * Replaces rhasspy_speech/transcribe_util.py:11-88 (`get_fuzzy_text`: the n-best text piped through fstcompile |
 * fstcompose - G.fuzzy.fst | fstshortestpath | fstrmepsilon | fsttopsort | fstproject | fstprint, then the printed arcs
 * summed in Python).  rs_fuzzy_open parses the FST once (the reference re-reads it per utterance). */
typedef struct rs_fuzzy rs_fuzzy;
int rs_fuzzy_open(const char *fuzzy_fst_path, rs_fuzzy **out);
/* nbest_text = the bytes rs_result_text() renders (`utt-k id id ...` lines).  On a match *n_out = number of output labels
 * (<= cap written to olabels, word ids of <lang_dir>/words.txt incl. `__output:` meta words, epsilons removed) and *cost =
 * the value the reference compares with max_fuzzy_cost; *n_out = -1 when the reference would return None. */
int rs_fuzzy_match(const rs_fuzzy *f, const char *nbest_text, int32_t *olabels, int32_t cap, int32_t *n_out, double *cost);
/* The same match taken straight from a decode result (utterance `utt` of r): what transcribe_wav.py:87-93 /
 * transcribe_stream.py:117-123 do with the pipeline's stdout, without rendering and re-parsing the n-best text. */
int rs_result_fuzzy(const rs_result *r, int32_t utt, const rs_fuzzy *f, int32_t *olabels, int32_t cap, int32_t *n_out, double *cost);
void rs_fuzzy_free(rs_fuzzy *f);

/* ---- decoding-graph construction (host side).
 * Replaces `bash utils/mkgraph.sh --self-loop-scale 1.0 <lang_dir> <model_dir>/model <graph_dir>` (rhasspy_speech/kaldi.py:409-425;
 * kaldi/egs/wsj/s5/utils/mkgraph.sh:72-170), i.e. the process chain
 *   fsttablecompose L_disambig.fst G.fst | fstdeterminizestar --use-log=true | fstminimizeencoded | fstpushspecial   (LG)
 *   fstcomposecontext --context-size=N --central-position=P ... | fstarcsort --sort_type=ilabel                      (CLG, N / P from `tree`)
 *   make-h-transducer --transition-scale=<transition_scale> ilabels tree final.mdl                                   (Ha)
 *   fsttablecompose Ha CLG | fstdeterminizestar --use-log=true | fstrmsymbols | fstrmepslocal | fstminimizeencoded   (HCLGa)
 *   add-self-loops --self-loop-scale=<self_loop_scale> --reorder=true final.mdl | fstconvert --fst_type=const        (HCLG.fst)
 * Reads lang_dir/{L_disambig.fst,G.fst,words.txt,phones/disambig.int} and model_dir/{tree,final.mdl}; writes
 * graph_dir/{HCLG.fst,words.txt,disambig_tid.int,phones/...}.  dump_dir (may be NULL): LG.fst, CLG.fst, ilabels, Ha.fst and
 * HCLGa.fst are written there as well.  The result is the same weighted transducer as the reference chain's (equal weights for
 * every input / output label sequence up to the 1/1024 quantisation both apply), with this library's own state numbering. */
int rs_mkgraph(const char *lang_dir, const char *model_dir, const char *graph_dir, float transition_scale, float self_loop_scale,
               const char *dump_dir);
/* One step of that chain on files, named like the Kaldi / OpenFst executable it stands for: fsttablecompose (in1, in2 -> out),
 * fstdeterminizestar (param != 0: --use-log=true), fstminimizeencoded, fstpushspecial, fstrmepslocal, fstrmsymbols (aux = symbol
 * list), fstarcsort (aux = "ilabel" | "olabel"), fstcomposecontext (in2 = disambig list, aux = ilabels output, param = 16 * N + P),
 * make-h-transducer (in1 = ilabels, in2 = tree, aux = final.mdl, param = transition scale; the disambiguation transition-ids go to
 * out + ".disambig"), add-self-loops (aux = final.mdl, param = self-loop scale; --reorder=true), and two checks that fail with a
 * description of the first difference: fstisomorphic (same transducer up to state numbering, weights within param) and
 * fstequivalent (fstequivalent --random=true: equal tropical weight, within param, of the label pairs of random paths). */
int rs_fst_tool(const char *tool, const char *in1, const char *in2, const char *out, const char *aux, float param);

/* ---- what the reference's decoder binaries do to a model before the first frame (host side; no device needed).
 * Both binaries (kaldi/src/online2bin/online2-wav-nnet3-latgen-faster.cc:160-176, online2-cli-nnet3-decode-faster.cc:97-111)
 * run CollapseModel on the network they read and, while computing the network's context and compiling its looped computation,
 * call glibc's rand() a model-dependent number of times -- which decides the noise Dither() adds to every frame afterwards
 * (kaldi/src/feat/feature-window.cc:90-98; the default mfcc configuration has --dither=1 and rhasspy starts one process per
 * utterance, so the noise of frame t is a constant of the model).  rs_model_load replays both; these two entry points expose
 * the replay for inspection and tests:
 *   rs_nnet3_setup: *rand_calls = number of rand() calls before the first frame for `final_mdl` under --frames-per-chunk;
 *     *certain = 0 when the count assumes a looped compilation the reference may have to retry (rs_model_load refuses such a
 *     model unless its mfcc.conf says --dither=0); collapsed_config (may be NULL) receives the config lines of the collapsed
 *     network, '\n'-separated and NUL-terminated, truncated to buf_len.
 *   rs_dither_noise: out[(t - t0) * window + i] = the value Dither() adds, for --dither=1, to sample i of frame t in a process
 *     that called rand() `rand_calls` times before its first frame. */
int rs_nnet3_setup(const char *final_mdl, int32_t frames_per_chunk, int64_t *rand_calls, int32_t *certain, char *collapsed_config,
                   size_t buf_len);
/*   rs_nnet3_setup_subsampled: the same under --frame-subsampling-factor (the looped computation is compiled for the output frames
 *     t = 0, f, 2 f, ... and a chunk rounded up to a multiple of f: nnet-compile-looped.cc:81-94,111-128). */
int rs_nnet3_setup_subsampled(const char *final_mdl, int32_t frames_per_chunk, int32_t frame_subsampling_factor, int64_t *rand_calls,
                              int32_t *certain, char *collapsed_config, size_t buf_len);
int rs_dither_noise(int64_t rand_calls, int32_t t0, int32_t t1, int32_t window, float *out);

/* ---- rescoring against a NEW language directory (host side; the lattices come from decodes with rs_decode_opts.emit_lattice = 1).
 * Replaces the tool chain of `async_transcribe_rescore` (rhasspy_speech/transcribe_wav.py:107-232, transcribe_stream.py:131-274):
 *   [Ldet.fst from L_disambig.fst: fstprint | awk | fstcompile | fstdeterminizestar | fstrmsymbols]   lattice-scale --lm-scale=0.0 |
 *   lattice-to-phone-lattice | lattice-compose - Ldet.fst | lattice-determinize | lattice-compose --phi-label=#0 - G.fst |
 *   lattice-add-trans-probs --transition-scale=1.0 --self-loop-scale=0.1 | lattice-to-nbest --n --acoustic-scale | nbest-to-linear
 * rs_rescorer_open reads <new_lang_dir>/{L_disambig.fst, G.fst, words.txt, phones/disambig.int} once (the reference rebuilds
 * Ldet.fst and re-reads everything per utterance); it fails like the reference when words.txt has no #0. */
typedef struct rs_rescorer rs_rescorer;
int rs_rescorer_open(const rs_model *model, const char *new_lang_dir, rs_rescorer **out);
/* Rescores utterance `utt` of a result; renders the bytes `nbest-to-linear ... ark,t:-` prints at the end of the chain
 * ("<key>-<k> id id ... \n", ids of the NEW words.txt; nothing when no path survives) into buf, returns the number of bytes
 * needed (like snprintf) or a negative status.  graph_cost / acoustic_cost (may be null): up to `nbest` entries, the 4th / 5th
 * outputs of nbest-to-linear; *n_out = number of hypotheses. */
int rs_rescore_result(const rs_rescorer *r, const rs_result *res, int32_t utt, int32_t nbest, float acoustic_scale, const char *key,
                      char *buf, size_t len, float *graph_cost, float *acoustic_cost, int32_t *n_out);
/* Same on a lattice given as one binary CompactLattice table entry (what online2-wav-nnet3-latgen-faster writes, and what
 * rs_result_lattice renders): no GPU involved -- for tools and for the host-only parity tests against the reference's chain. */
int rs_rescore_lattice(const rs_rescorer *r, const char *lattice_entry, size_t n_bytes, int32_t nbest, float acoustic_scale, const char *key,
                       char *buf, size_t len, float *graph_cost, float *acoustic_cost, int32_t *n_out);
void rs_rescorer_free(rs_rescorer *r);

/* Host-side entry for tests and tools (no GPU needed): determinises a raw lattice -- the state-level lattice the search
 * leaves behind, given as n_arcs arcs (src, dst, word label, transition-id; graph and acoustic cost), a start state and
 * per-state final costs (+inf = not final) -- with the same code rs_result_lattice uses (lattice beam `beam`) and renders
 * the binary CompactLattice table entry.  Returns the entry's size (copies min(size, cap) bytes), or a negative error. */
int64_t rs_lattice_entry_from_raw(int32_t num_states, int32_t start, const float *final_cost, int32_t n_arcs, const int32_t *arc_src,
                                  const int32_t *arc_dst, const int32_t *arc_word, const int32_t *arc_tid, const float *arc_graph,
                                  const float *arc_acoustic, float beam, const char *key, char *buf, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* RHASSPY_SPEECH_HIP_H_ */
