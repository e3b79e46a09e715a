// Launchers of the hand-written gfx950 kernels (one process per GPU; all launches go to the caller's
// hipStream_t).  Device pointers unless noted.  Row layout shared by every per-frame buffer of a batch
// ("ragged time-major with halo"): utterance u owns rows [row_base[u], row_base[u] + T_u + L + R);
// frame t of u is row row_base[u] + L + t; L/R halo rows replicate the edge frames of the *input*
// features so that every nnet layer can be evaluated on all rows with constant row offsets.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace rs {

// ---------------------------------------------------------------- batch geometry (device copies)
struct BatchGeom {
  int n_utts = 0;
  int L = 0, R = 0;            // halo
  int total_rows = 0;          // sum of (T_u + L + R)
  int total_frames = 0;        // sum of T_u
  int max_frames = 0;
  int guard = 0;               // zeroed guard rows before/after every frame buffer
  const int64_t *d_sample_off = nullptr;  // n_utts + 1
  const int *d_num_frames = nullptr;      // n_utts
  const int *d_row_base = nullptr;        // n_utts + 1
  const int *d_frame_base = nullptr;      // n_utts + 1 (prefix sum of T_u; compact per-frame arrays)
  const int *d_row_utt = nullptr;         // total_rows: utterance of each row
  const int *d_row_t = nullptr;           // total_rows: frame index of each row, row - row_base[u] - L: negative / >= T_u in the halo
  // streams: the launch covers frames [frame0[u], frame0[u] + T_u) of stream u, its sample_off points at frame0[u]'s first
  // sample; what depends on the absolute frame index (the dither noise) adds it back.  null = 0.
  const int *d_frame0 = nullptr;          // n_utts
};

// ---------------------------------------------------------------- MFCC
struct MfccDev {
  int win, shift, padded, nbins, nceps;
  float preemph;
  int remove_dc, use_energy, raw_energy;
  float log_energy_floor;
  const float *window;       // win
  const int *mel_offset, *mel_len, *mel_start;
  const float *mel_weights;
  const float *dct;          // nceps x nbins
  const float *lifter;       // nceps
  // split-radix FFT plan (srfft_plan.h)
  const int *fft_tasks;      // SrfftTask records (int4 each), level-major
  int fft_num_levels, fft_num_tasks, fft_num_tw;
  int fft_level_begin[16];   // task range of each level
  const float *fft_tw;       // 6 floats per twiddled butterfly
  // the same plan as 64 48-byte records per level, one per lane {byte offsets of the task's points in xr / xi; kind (3: no task) |
  // twiddle class << 8, the task's six twiddle factors; -} when no level has more than 64 tasks (a 512-point window): a lane's
  // record of the NEXT level is requested while the current one runs, at a fixed offset from the level's base, and the factors come
  // with it instead of by a second, dependent load (null: plans with wider levels).  With it the points live in a bank-conflict-
  // reducing layout: point i at index i ^ g(i >> 5), g(b) = fft_swz[0] (bit 0 of b) ^ fft_swz[1] (bit 1) ^ fft_swz[2] (bit 2);
  // fft_perm is then the gather in that layout (all zero: the plain layout)
  int fft_swz[3];
  const float4 *fft_recs;
  const int *fft_perm;       // padded/2: bit-reversal pass as a gather
  const float *fft_kn;       // (re, im) of the post-processing factor, k = 0 .. padded/4
  // (512-point window) the same two tables as one 16-byte record per lane and round of the post-processing pass, k = lane + 1 + 64 q:
  // {fft_perm[k], fft_perm[256 - k], kn re, kn im} -- one request instead of four; and the lane's mel filter {offset, length, start, -}
  const float4 *fft_post;    // [2][64]
  const int4 *mel_rec;       // nbins
  // Dither (feature-window.cc:90-98): sample i of frame t += dither[t * win + i] * dither_value, before DC removal.  The
  // table holds the reference's own RandGauss draws for frame t of a fresh decoder process (nnet3_setup.h); null = off.
  const float *dither;
  float dither_value;
};
// feats: total_rows x ld (C columns used).  Writes every row (halo rows replicate edge frames).
// out_rows (null = the row itself): physical row of `feats` that receives dense row i (streams write into pool rows).
void LaunchMfcc(const MfccDev &m, const BatchGeom &g, const int16_t *pcm, float *feats, int ld, hipStream_t s, bool exclusive = false,
                const int *out_rows = nullptr);
// dst row dst_row[i] (null = i) <- src row src_row[i] (null = i), width_words 4-byte words, leading dimensions in words.
// the same for up to four arrays that share the row lists and whose rows are as wide as their pitch (the per-stream estimator state)
struct CopyRowsSet {
  struct One { const void *src; void *dst; long ld; int width; } a[4];
  int count;
};
void LaunchCopyRowsMulti(const CopyRowsSet &set, const int *src_row, const int *dst_row, int n, hipStream_t s);
// h_*: page-locked host memory the device can store into (hipHostMalloc); the first inline_words word ids of utterance u go to
// h_words[u * h_stride ...] (copy count and destination pitch are separate: ADVICE r05)
void LaunchResultsToHost(const int *nw, const float *costs, const long long *ctr, const int *words, int max_words, int inline_words, int h_stride, int n,
                         int *h_nw, float *h_costs, long long *h_ctr, int *h_words, hipStream_t s);
void LaunchCopyRows(const void *src, long src_ld_words, const int *src_row, void *dst, long dst_ld_words, const int *dst_row, int n, int width_words,
                    hipStream_t s);

// zero fill of a list of 16-byte aligned regions in one launch
struct ZeroRegions {
  static constexpr int kMax = 96;
  struct One { void *p; size_t bytes; } r[kMax];
  int count;
};
void LaunchZeroRegions(const ZeroRegions &z, hipStream_t s);

// frame_rows[i] = physical row of the i-th frame in slab-major order: entry (k, u) of seg_off (n_segs + 1 offsets, n_segs =
// n_slabs * n_utts) starts the frames [k * slab_len, ...) of utterance u.
void LaunchFrameRows(int n_utts, int n_segs, int total, int L, int slab_len, const int *seg_off, const int *row_base, int *frame_rows,
                     hipStream_t s);
void LaunchRowGeometry(int n_utts, int rows, int L, const int *row_base, const int *ivrow_base, int *row_utt, int *row_t, int *row_ivec,
                       hipStream_t s);
// The two above for a whole batch in ONE launch: the per-row geometry and up to kMaxLists row lists (a copy and a launch per list was
// 16 of the launch boundaries in front of a decode call's first real kernel).
struct BatchSetup {
  static constexpr int kMaxLists = 24;
  int n_utts, rows, L;
  const int *row_base, *ivrow_base;
  int *row_utt, *row_t, *row_ivec;      // row_ivec null: not written
  // element i of segment (k, u): row_base[u] + L + k * slab_len + first + stride * (i - seg_off)
  struct List { int n_segs, total, L, slab_len; const int *seg_off; int *out; int stride, first; } lists[kMaxLists];
  int n_lists;
};
void LaunchBatchSetup(const BatchSetup &b, hipStream_t s);

// ---------------------------------------------------------------- online CMVN (sliding window, causal)
struct CmvnDev {
  int dim, cmn_window, speaker_frames, global_frames;
  const double *global_stats;   // 2 x (dim+1), row 0 used
};
// in/out: total_rows x ld, same layout.  Halo rows of `out` replicate its edge frames.
// Streams: t_begin[u] = first frame to produce (frames before it were produced by earlier launches), state = parked running
// sums + window count, (dim + 1) doubles per slot, slot of utterance u = state_slot[u].  No halo rows are written then.
void LaunchOnlineCmvn(const CmvnDev &c, const BatchGeom &g, const float *in, float *out, int ld, hipStream_t s, const int *t_begin = nullptr,
                      double *state = nullptr, const int *state_slot = nullptr);

// ---------------------------------------------------------------- generic segmented GEMM (FP32 MFMA)
constexpr int kGemmBM = 128, kGemmBN = 128, kGemmBK = 32;
// A frame buffer stored a second time in the operand order of the fp16 matrix cores, already split into the two fp16
// parts of nnet_gemm_b3.hip: part p at base + p * part_bytes, inside a part [row block of 32][16-wide k-step][k-group 2]
// [row 32][8 fp16] -- one 1 KiB block is one A fragment of v_mfma_f32_32x32x16_f16 (lane l: row l & 31, k-group l >> 5).
// Image row = buffer row + guard (a multiple of 32); columns beyond the buffer's width up to 16 * nks are zero.
struct ActImage {
  unsigned char *base;
  size_t part_bytes;
  int nks;
  int guard;
};
struct GemmSegDev {
  ActImage img;       // the source buffer's operand image (base null: none)
  const float *src;   // source buffer base (row 0 of the frame buffer), or iVector matrix if per_utt
  int ld;             // leading dimension of the source
  int col0;           // first source column
  int ncols;          // valid K extent
  int row_off;        // constant row offset (time offset)
  int k0;             // first (padded) K index of this segment inside W
  int per_utt;        // 1: row index = row_ivec[row] (iVector input: one row per utterance, or per nnet chunk when streaming)
};
constexpr int kMaxSegs = 16;
struct EltStageDev {
  int kind;                 // 0 relu, 1 scale+offset, 4 scalar scale
  const float *scale, *offset;
  float alpha;
};
constexpr int kMaxStages = 6;
struct GemmDev {
  int nsegs;
  GemmSegDev segs[kMaxSegs];
  const float *W;     // n_pad x k_pad, row-major, zero padded (k_pad = sum of segment widths rounded to kGemmBK)
  int k_pad, n, n_pad;
  const void *W3;     // the same weights, every output column scaled by a power of two (w3_inv_scale) and split into two fp16 parts in MFMA fragment order (nnet_gemm_b3.hip), or null
  const void *W3I;    // the same for GemmKernelB3I: segments padded to the 16-wide k-step instead of to kGemmBK, or null
  const float *w3_inv_scale;   // n3 floats: what the accumulators of column c are multiplied by (the inverse of W3's column scale)
  int *ovf;           // two words.  [0]: set to 1 by a kernel that met an activation the fp16 split cannot carry (|x| >= 65520 or not a number);
                      // [1]: ... that split an operand row whose largest element is below 2^-3 (the split would carry it to 2^-25
                      // absolute only, nnet_b3_common.h).  Either way the host repeats the call on the exact-FP32 kernels
  int n3;             // columns of W3 (n rounded up to 256)
  int interleave;     // 1: W3's k-steps alternate between the segments (all segments shifted views of one buffer)
  int exclusive;      // 1: GemmKernelB3 keeps every other workgroup off its CU (several decode pipelines in flight)
  int share;          // launches of this kind that run side by side on the device (sub-batch groups): tile planning hint
  const float *bias;  // n (may be null)
  int nstages;
  EltStageDev stages[kMaxStages];
  float *out;
  int ldo;
  // null, or the rows of a buffer as wide as the result: out = stages(...) + res_scale * res[same row] (a residual sum folded into the
  // layer, LayerOp::res_buf), the product and the sum rounded like the elementwise kernel's (__fmul_rn, __fadd_rn)
  const float *res;
  int res_ld;
  float res_scale;
  // base non-null (split-fp16 kernels fed by operand images): the residual's rows are taken from ITS operand image instead -- the sum of
  // the two fp16 parts, i.e. the value the next layer's GEMM multiplies, within 2^-22 of the FP32 number (nnet_b3_common.h) -- so a
  // chain of residual layers needs no FP32 copy of its activations at all (half the epilogue's traffic, no pass through LDS)
  ActImage res_img;
  ActImage out_img;    // base non-null: the result is (also) written as an operand image for the layers that consume it
  int write_f32;       // 0: nobody reads `out` as floats (every consumer takes the image): skip that store
  const int *row_map;  // null, or rows entries: GEMM row i reads / writes physical row row_map[i] (e.g. only the real frames)
  int row_map_span128; // with a row map: an upper bound of row_map[i + 127] - row_map[i] + 1 over the list when the list is ascending (128 rows of
                       // a tile reach over that many physical rows: GemmKernelB3J stages them as one strip), 0 = not known
  int row_map_span160; // the same for 160 consecutive rows of the list (the 160-row tile)
};
// f32 frame buffer (rows x ld, `dim` columns) -> operand image (nnet_gemm_b3i.hip); for producers without a fused image epilogue.
// row_map (null: rows 0 .. rows - 1): the `rows` physical rows to convert -- the rows the producer wrote (a layer evaluated through a
// row list leaves the others as the arena held them: not numbers, possibly, and they would pass through the range checks)
void LaunchToImage(const float *src, int ld, int dim, int rows, const ActImage &img, int *ovf, hipStream_t s, const int *row_map = nullptr);
constexpr int kActImageParts = 2;
size_t ActImagePartBytes(int rows, int guard, int dim);      // bytes of one part for a buffer of `rows` rows
void LaunchGemm(const GemmDev &d, int rows, const int *row_ivec, hipStream_t s);
// The split-fp16 kernels work on 256-column tiles: a layer takes them only while the padding stays below a share of the
// padded width (45 %; above it the exact-FP32 kernel with its 128-column tiles wins).  RS_GEMM_B3_PAD overrides (percent).
bool GemmB3PaddingOk(int n, int n3);
bool GemmWritesImage(const GemmDev &d);      // the kernel LaunchGemm picks writes d.out_img (else: LaunchToImage afterwards)
bool GemmImagesEnabled();                     // RS_GEMM_B3I / RS_GEMM_B3 (read per call)

struct SumTermDev { const float *src; int ld, col0, row_off; float scale; };
struct EltwiseDev {
  int nterms;
  SumTermDev terms[8];
  int nstages;
  EltStageDev stages[kMaxStages];
  int dim;
  float *out;
  int ldo;
  int row_reduce;     // 0 none, 2 log-softmax, 3 normalize (alpha = target rms)
  float alpha;
  const int *row_map; // null, or `rows` entries: element i works on physical row row_map[i] of every operand and of the result
};
void LaunchEltwise(const EltwiseDev &d, int rows, hipStream_t s);
// out[row][:] = (in[row][:] + neg_log_prior[:]) * scale  (decodable-online-looped.cc:218-223), in place
void LaunchPriorScale(float *x, int ld, int rows, int dim, const float *log_priors, float scale, hipStream_t s);

// ---------------------------------------------------------------- iVector
struct IvecDev {
  int feat_dim, ivec_dim, num_gauss, num_gselect, num_cg_iters;
  float min_post, posterior_scale, max_count;
  double prior_offset;
  const float *gconsts;        // G
  const float *means_invvars_t; // D x G (transposed for coalesced lane-per-Gaussian reads)
  const float *inv_vars_t;      // D x G
  // the same two tables in the B-fragment order of v_mfma_f32_16x16x4_f32 (UbmPostMfmaKernel): [Gaussian tile of 16][group of
  // four k-steps][lane 64][4]: lane l, element i = parameter of dimension 16 kg + 4 i + (l >> 4), Gaussian 16 j + (l & 15);
  // ubm_kg groups (1 for D <= 16, else 3: D <= 48), tiles padded to the kernel instantiation's count; null: not prepared
  const float *ubm_bm, *ubm_bv;
  int ubm_kg;
  const double *sigma_inv_M;    // G x D x I
  const double *U;              // G x I(I+1)/2
};
// UBM posteriors per frame: post_idx/post_w: total_frames x num_gselect (idx -1 = unused)
void LaunchUbmPosteriors(const IvecDev &iv, const BatchGeom &g, const float *lda_norm, int ld,
                         int *post_idx, float *post_w, hipStream_t s);
// Accumulates, per (utterance, Gaussian): gamma (n_utts x G) and weighted feature sums (n_utts x G x D),
// in frame order (double), for frames [frame_begin[u], frame_end[u]) of each utterance (null = all).  geo_mod > 0: g.n_utts pseudo-
// utterances, number u over the rows of utterance u % geo_mod (the chunks of a round of streams side by side).
void LaunchIvecAccumulate(const IvecDev &iv, const BatchGeom &g, const float *lda, int ld, const int *post_idx,
                          const float *post_w, const int *frame_begin, const int *frame_end,
                          double *gamma, double *wfeats, bool fresh, hipStream_t s, int geo_mod = 0);
// linear (n_utts x I) += sum_g Sigma_inv_M_g^T wfeats_g ; quadratic (n_utts x I(I+1)/2) += sum_g gamma_g U_g,
// plus the max_count prior rescaling of OnlineIvectorEstimationStats::AccStats; num_frames (n_utts, double).
// scratch: IvecStatsScratchDoubles() doubles of workspace.
void LaunchIvecStats(const IvecDev &iv, int n_utts, const double *gamma, const double *wfeats,
                     double *linear, double *quadratic, double *num_frames, double *scratch, hipStream_t s);
size_t IvecStatsScratchDoubles(const IvecDev &iv, int n_utts);
// Fresh estimator state per utterance: quadratic = I, linear = x = [prior_offset, 0, ...], num_frames = 0.
void LaunchIvecInit(const IvecDev &iv, int n_utts, double *linear, double *quadratic, double *x, double *num_frames, hipStream_t s);
// Conjugate-gradient solve per utterance (LinearCgd, <= num_cg_iters), x in/out (double, n_utts x I);
// ivec_out (float, n_utts x ldo) = x with prior_offset subtracted from element 0.
// out_row (null = u): row of ivec_out receiving utterance u's estimate, -1 = skip; active (null = all): 0 = no new
// frames since the last estimate, just re-emit it (the reference does not re-run CG then).
void LaunchIvecSolve(const IvecDev &iv, int n_utts, const double *linear, const double *quadratic,
                     const double *num_frames, double *x, float *ivec_out, int ldo, const int *out_row, const int *active,
                     hipStream_t s);
void LaunchIvecClear(const IvecDev &iv, int n_utts, double *gamma, double *wfeats, hipStream_t s);
// The chunks of a round of streams in one launch: dlin / dquad / dtot = the increments LaunchIvecStats leaves for K x n_utts
// pseudo-utterances started from zero ([k * n_utts + u]); the kernel applies them in chunk order to stream u's estimator state
// (row slot[u], or u, of lin / quad / numf / x), solves, writes iVector row out_row[k * n_utts + u] (-1: no such chunk; active 0: re-emit).
// false: ivec_dim beyond the kernel's LDS matrix -- the caller falls back to one LaunchIvecStats + LaunchIvecSolve per chunk.
bool LaunchIvecChain(const IvecDev &iv, int n_utts, int K, const double *dlin, const double *dquad, const double *dtot, double *lin, double *quad,
                     double *numf, double *x, const int *slot, float *ivec_out, int ldo, const int *out_row, const int *active, hipStream_t s);

// ---------------------------------------------------------------- decoder
struct HclgDev {
  int num_states, num_arcs, start;
  const uint32_t *arc_begin;   // S + 1
  const uint32_t *num_ieps;    // S
  const uint4 *state_rec;      // S : {first arc, epsilon arcs, emitting arcs, 0}: one load where the token-list search needs the ranges
  const int4 *arcs;            // {pdf+1 (0 = epsilon), olabel, weight bits, nextstate}
  const int *arc_src;          // source state of each arc
  const int *arc_srcx;         // source state | (1 << 31 if the arc is an epsilon arc): one load per traceback hop
  // decode_live.hip: the arcs with two flags in .x (bit 31: the destination state has epsilon arcs; bit 30: it is the destination of
  // an epsilon arc), and one 64-byte record per state {first arc, epsilon arcs, emitting arcs, 0}, emitting arc 0, emitting arc 1, -
  const int4 *arcs_f;
  const uint4 *nodes;
  const float *final_cost;     // S
};
struct DecodeOptsDev {
  float beam, lattice_beam, beam_delta;
  int max_active, min_active;
  int exact_order;            // rs_decode_opts.exact_token_order: the reference's order-dependent token creation (decode_reg.hip), where the graph allows it
  int no_commit_hist = 0;     // RS_REG_NO_HIST=1 (tests): RegDecodeKernel's GetCutoff always selects the slow way (KthFromHist)
};
struct DecodeWork {
  // per utterance
  unsigned long long *best;   // n_utts x S : packed (ordered cost bits << 32 | arc), ~0 = empty
  int *map_a, *map_b;         // n_utts x S : state -> token index (frame-local), -1 = none
  int tok_cap;                // token capacity per utterance (all frames)
  int4 *tokens;               // n_utts x tok_cap : {state, cost bits, backpointer (frame-local idx), arc (-1 start)}
  int *frame_tok_off;         // n_utts x (max_frames + 2): start offset of each frame's tokens
  float *frame_info;          // n_utts x (max_frames + 1) x 4 : {cost_offset, cur_cutoff, next_cutoff, adaptive_beam}
  int *queue_a, *queue_b;     // n_utts x S work lists for the epsilon closure
  int *in_queue;              // n_utts x S : round stamp of the state's last push onto a closure queue (token-list search)
  // live-state-table search (LiveDecodeKernel, decode_live.hip): a slot is the position of a state's entry in the utterance's table
  // (LDS part, then the global part); null = not in use
  int h_tab;                  // DecodeLiveTableSize(): length of the slot-indexed arrays
  unsigned long long *h_keys; // n_utts x h_tab : packed (cost, arc) of the states in the global part of the table (its first entries)
  int *h_slot_tok;            // n_utts x h_tab : slot -> token index in the frame under construction, for the states the closure can reach
  unsigned *h_gtags;          // n_utts x DecodeLiveGlobalTable() : state ids of the global part of the table
  int *h_cand;                // n_utts x 2 x h_cand_cap : candidate records of a frame {arc | flags, slot << 16 | source token}
  int h_cand_cap;
  int h_slot_limit;           // live states a frame may hold before the utterance is handed to DecodeKernel (<= DecodeLiveSlotCap())
  int4 *h_q4;                 // n_utts x 2 x h_qcap : closure work lists {slot | writer token << 16, first epsilon arc, key low, key high}
  int *h_qne;                 // n_utts x 2 x h_qcap : ... and the number of epsilon arcs of the entry's state
  int h_qcap;
  int4 *h_comp;               // n_utts x slot cap : the frame's tokens with more than two emitting arcs {first arc left, cost bits, token index, arcs left}
  int h_lds_log;              // > 1: use only 2^h_lds_log entries of the LDS part (tests: forces states into the global part)
  int *redo;                  // n_utts : 1 = the utterance outgrew the live-state table, DecodeKernel decodes it (null: DecodeKernel decodes all)
  // results
  int *out_words;             // n_utts x max_words
  int *out_nwords;            // n_utts
  float *out_costs;           // n_utts x 4 : graph, acoustic, total, final-reached flag
  long long *counters;        // n_utts x 8
  int max_words;
};
void LaunchDecode(const HclgDev &h, const DecodeOptsDev &o, const BatchGeom &g, const float *loglikes, int ld,
                  const DecodeWork &w, hipStream_t s);
// The same search with the live states of a frame in a table that follows the beam, not the graph (decode_live.hip): tags and
// recombination keys in LDS, a second level in global memory behind them.  Utterances that outgrow it (more than
// DecodeLiveSlotCap() live states in a frame, more records than the lists hold) are flagged in w.redo and decoded by LaunchDecode,
// which the caller issues behind it on the same stream.
constexpr int kHashCandCap = 65536;      // candidate records per utterance and frame (the ARPA workload's largest frame: 24 k)
constexpr int kLiveQueueCap = 65536;     // closure work-list entries per round
bool DecodeLiveUsable(const HclgDev &h);
int DecodeLiveSlotCap();
int DecodeLiveTableSize();
int DecodeLiveGlobalTable();
void LaunchDecodeLive(const HclgDev &h, const DecodeOptsDev &o, const BatchGeom &g, const float *loglikes, int ld,
                      const DecodeWork &w, hipStream_t s);

// Dense ("pull") variant for graphs whose per-state tables fit in LDS: one thread per destination state walks
// its incoming arcs (reverse graph), so there are no atomics and no token lists; the frame's token costs live
// in LDS.  Produces exactly the sparse kernel's costs and back-pointers (same float expressions, same
// lowest-arc-index tie rule).  Back-pointer rows go to HBM for the traceback.
struct RevGraphDev {
  const uint32_t *in_begin_e;   // S + 1 : emitting in-arcs of each state
  const uint32_t *in_begin_x;   // S + 1 : epsilon in-arcs
  const int4 *in_e;             // {src state, pdf + 1, weight bits, forward arc index}, sorted by forward arc index
  const int4 *in_x;
  const int *eps_dst;           // states with at least one epsilon in-arc
  int num_eps_dst;
  int in_begin_e_host_total;    // number of emitting / epsilon in-arcs (= in_begin_*[S])
  int in_begin_x_host_total;
};
struct DenseWork {
  int *bp;                    // n_utts x (max_frames + 1) x S : back-pointer arc of each (frame, state), -2 = no token, -1 = start
  int *out_words, *out_nwords;
  float *out_costs;           // n_utts x 4
  long long *counters;        // n_utts x 8
  float *frame_info;          // n_utts x (max_frames + 1) x 4
  int max_words;
  int *path;                  // n_utts x path_cap x 2 scratch: best path as (arc, frame) pairs
  int path_cap;
  // n-best / lattice calls on grammar graphs (RegDecodeKernel + LaunchDenseToTokens): the costs of every (frame, state) beside the
  // back-pointer rows, +inf = no token; null = not kept
  float *cost_rows = nullptr; // n_utts x (max_frames + 1) x S
  // resumable decoding (decode_reg.hip): token costs and scalars carried between the time slabs of one utterance
  float *state_cost;          // n_utts x (2 S + 4): S costs, then {closure cutoff, error flag, -, -}, then S list positions (exact token order)
  // Streams (decode_reg.hip, win_begin != null): utterance u decodes frames [win_begin[u], win_end[u]) (win_begin -1 starts the
  // stream), of which d_num_frames[u] exist so far; win_final[u] != 0 ends it (traceback + results).  Its back-pointer / frame
  // info rows live at pool row pool_row[u] onwards (bp, frame_info = the pools' bases), its parked costs and counters in slot
  // slot[u] of state_cost / counters; out_* and path stay indexed by u.
  const int *win_begin, *win_end, *win_final, *pool_row, *slot;
};
// Register-resident variant (decode_reg.hip): the arcs are dealt out to the threads of an NT-thread workgroup (arc i ->
// thread i % NT, register slot i / NT) and live in VGPRs for the whole utterance.  Tables are [slot][thread] so that
// loading them is coalesced; LDS addresses are baked in on the host (cost_cur at byte 0, keys at key_base).
constexpr int kRegMaxStates = 5000;      // 16-bit LDS byte addresses: key_base + 8 * (S + 1) < 65536
struct RegGraphDev {
  int nt = 0;                 // 0 = graph does not fit this variant
  int ke = 0, kx = 0;         // register slots per thread for emitting / epsilon arcs (a kernel instantiation)
  int eps_depth = 0;          // longest epsilon path (rounds to the fixpoint); 0 = no epsilon arcs; -1 = cyclic or deep -> vote
  int key_base = 0;           // byte offset of key_next[] in the dynamic LDS region
  const int4 *e_tab;          // [ke][nt] {4*src | (key_base + 8*dst) << 16, pdf, weight bits, forward arc index}
  const int4 *x_tab;          // [kx][nt] {(key_base + 8*src + 4) | (key_base + 8*dst) << 16, 0, weight bits, forward arc index}
  // RegDecodeExactKernel: per emitting arc (table position of its source state's first emitting arc) << 8 | its number among
  // them; per epsilon arc its number among its source's epsilon arcs.  exact_ok: <= 1000 states, epsilon depth <= 1, <= 32 arcs of
  // either kind per state.
  const int *e_aux, *x_aux;
  int exact_ok = 0;
};
bool RegDecodeConfig(int num_states, int num_emitting, int num_eps, int *nt, int *ke, int *kx);
// Decodes frames [f_begin, f_end) of every utterance (f_begin = -1 starts an utterance; the slab that contains an
// utterance's last frame also does its traceback).  w.counters must be zeroed before the first slab.
// With w.win_begin set (streams) f_begin / f_end are ignored; `any_final` then says whether some utterance ends in this launch
// (LDS for the traceback staging is only requested then).
bool LaunchDecodeReg(const HclgDev &h, const RegGraphDev &r, const DecodeOptsDev &o, const BatchGeom &g,
                     const float *loglikes, int ld, const DenseWork &w, int f_begin, int f_end, hipStream_t s, bool any_final = true);
// Token lists (DecodeWork: tokens {state, cost bits, -, back-pointer arc}, frame_tok_off) out of the dense cost / back-pointer rows a
// register-resident search with DenseWork::cost_rows left behind: what LatticeKernel reads.  Frame 0's first token is the start state's.
void LaunchDenseToTokens(const HclgDev &h, const BatchGeom &g, const DenseWork &dw, const DecodeWork &w, hipStream_t s, bool write_tokens = true);
// the lattice's links straight from the dense rows (graphs of at most 2048 states / 8192 arcs; RS_LATTICE_KERNEL=tokens: never)
struct LatticeWork;
bool DenseLatticeUsable(const HclgDev &h);
void LaunchDenseLattice(const HclgDev &h, const DecodeOptsDev &o, const BatchGeom &g, const float *loglikes, int ld, const DenseWork &dw,
                        const DecodeWork &w, const LatticeWork &lw, int eps_rounds, hipStream_t s);   // eps_rounds: RegGraphDev::eps_depth
size_t DenseDecodeSmemBytes(int num_states, int num_pdfs);
bool DenseDecodeFits(int num_states, int num_pdfs);
void LaunchDecodeDense(const HclgDev &h, const RevGraphDev &r, const DecodeOptsDev &o, const BatchGeom &g,
                       const float *loglikes, int ld, int num_pdfs, const DenseWork &w, hipStream_t s);

// Lattice extraction = FinalizeDecoding (lattice-faster-decoder.cc:625-640): backward pass over the stored
// token lists that recomputes every forward link, derives the exact extra_cost of every token
// (PruneForwardLinksFinal / PruneForwardLinks with delta = 0) and emits the links within lattice_beam.
struct LatArc { int utt, src, dst, arc; float graph, acoustic; };   // arc = -1: final-cost record of token `src`
struct LatticeWork {
  float *extra_cost;          // n_utts x tok_cap
  // Every utterance appends to its own region [u * utt_cap, (u + 1) * utt_cap) of the buffer and counts in arcs_count[u] (which may
  // exceed utt_cap: then the caller retries with a larger buffer).  One counter for the whole call was 220 000 - 300 000 returning
  // atomics on one address per call -- most of the time of a lattice pass.
  LatArc *arcs;
  int utt_cap;
  int *arcs_count;            // n_utts
};
// region u's first min(count[u], utt_cap) records -> dst, regions back to back (dst: page-locked host memory or device memory)
void LaunchCompactArcs(const LatArc *arcs, int utt_cap, const int *counts, int n_utts, LatArc *dst, hipStream_t s);
void LaunchLatticePrune(const HclgDev &h, const DecodeOptsDev &o, const BatchGeom &g, const float *loglikes, int ld,
                        const DecodeWork &w, const LatticeWork &lw, hipStream_t s);

void LaunchLdsPoison(unsigned *sink, hipStream_t s);   // -DRS_TUNING builds only: profiles/micro/poison_kernels.hip

}  // namespace rs
