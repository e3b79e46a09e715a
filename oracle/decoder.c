/* TEST INFRASTRUCTURE ONLY -- CPU oracle, part 3: sequential restatement of the reference's lattice-generating
 * beam search.  Never linked into librhasspy_speech_hip.so; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg load it (through oracle/decoder.py).
 *
 * Follows kaldi/src/decoder/lattice-faster-decoder.cc step by step, with its single-threaded, order-dependent
 * semantics kept on purpose (the GPU kernel is order-independent; this file is what it is checked against):
 *   InitDecoding :56-73          -> init_decoding()
 *   FindOrAddToken :253-293      -> find_or_add()
 *   GetCutoff :644-711           -> get_cutoff()        (nth_element values via a sorted copy)
 *   ProcessEmitting :714-804     -> process_emitting()  (running next_cutoff, hash-list iteration order)
 *   ProcessNonemitting :820-887  -> process_nonemitting() (LIFO work list, forward links regenerated)
 *   PruneForwardLinks :299-370, PruneForwardLinksFinal :376-458, PruneTokensForFrame :479-503,
 *   PruneActiveTokens :510-533, ComputeFinalCosts :536-577, FinalizeDecoding :625-640, AdvanceDecoding :580-619
 *   GetRawLattice :106-189       -> rs_oracle_lattice_*
 * The token hash reproduces util/hash-list-inl.h:125-165: bucket = state % hash_size, buckets are chained in
 * the order they were first occupied, elements append at their bucket's end; the list is walked in that order.
 * All cost arithmetic is float with the reference's association; compile with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int state;
  float tot_cost, extra_cost;
  int links;        /* head of forward-link list (-1 none) */
  int next;         /* next token on the same frame (-1 end) */
  int backpointer;
  int alive;
  int frame;            /* trace only: the frame (plus one) the token lives on */
  float born_margin;    /* trace only: tot_cost - the frame's FINAL next_cutoff after ProcessEmitting (>= 0: an order-dependent extra); NAN: made by the closure */
} Tok;

typedef struct {
  int next_tok, ilabel, olabel;
  float graph_cost, acoustic_cost;
  int next;
} Link;

typedef struct {
  int toks;                 /* head of token list */
  int must_prune_forward_links, must_prune_tokens;
} TokList;

typedef struct { int key, val, tail; } Elem;         /* hash-list element: state -> token */
typedef struct { int last_elem, prev_bucket; } Bucket;

typedef struct {
  /* graph */
  int num_states, start;
  const float *final;
  const int64_t *arc_begin;
  const int32_t *ilabel, *olabel, *nextstate;
  const float *weight;
  const int64_t *num_ieps;
  /* acoustic */
  const float *loglikes;
  int T, P;
  const int32_t *id2pdf;
  /* config */
  float beam, lattice_beam, beam_delta, prune_scale, hash_ratio;
  int max_active, min_active, prune_interval;
  /* token / link arenas */
  Tok *tok; int ntok, captok;
  Link *lnk; int nlnk, caplnk, free_lnk;
  TokList *frames; int nframes, capframes;
  float *cost_offsets;
  int num_toks;
  /* hash list */
  Elem *elem; int nelem, capelem, free_elem;
  Bucket *buckets; int nbuckets; size_t hash_size;
  int list_head, bucket_list_tail;
  /* misc */
  int *queue; int nqueue, capqueue;
  float *tmp; int captmp;
  /* final */
  int finalized; float final_relative_cost, final_best_cost;
  float *final_costs;   /* per token id, NAN = absent */ int have_final_costs;
  int64_t counters[8];
  /* diagnostics (profiles/micro/c2_arpa_162.py), off unless the environment asks: */
  int final_cutoff_mode;    /* RS_ORACLE_FINAL_CUTOFF=1: ProcessEmitting prunes with the frame's final next_cutoff (what the kernels do) */
  FILE *trace;              /* RS_ORACLE_TRACE=<file>: one line per frame + the best path's order-dependent tokens */
} Dec;

static const float INF = INFINITY;

/* ---------------------------------------------------------------- arenas */
static int new_tok(Dec *d, float tot, float extra, int links, int next, int bp, int state) {
  if (d->ntok == d->captok) { d->captok = d->captok ? d->captok * 2 : 4096; d->tok = (Tok *)realloc(d->tok, sizeof(Tok) * d->captok); }
  Tok *t = &d->tok[d->ntok];
  t->state = state; t->tot_cost = tot; t->extra_cost = extra; t->links = links; t->next = next; t->backpointer = bp; t->alive = 1;
  t->frame = d->nframes - 1; t->born_margin = NAN;
  d->num_toks++;
  return d->ntok++;
}
static int new_link(Dec *d, int next_tok, int il, int ol, float g, float a, int next) {
  int id;
  if (d->free_lnk >= 0) { id = d->free_lnk; d->free_lnk = d->lnk[id].next; }
  else {
    if (d->nlnk == d->caplnk) { d->caplnk = d->caplnk ? d->caplnk * 2 : 8192; d->lnk = (Link *)realloc(d->lnk, sizeof(Link) * d->caplnk); }
    id = d->nlnk++;
  }
  Link *l = &d->lnk[id];
  l->next_tok = next_tok; l->ilabel = il; l->olabel = ol; l->graph_cost = g; l->acoustic_cost = a; l->next = next;
  return id;
}
static void free_link(Dec *d, int id) { d->lnk[id].next = d->free_lnk; d->free_lnk = id; }
static void delete_forward_links(Dec *d, int tok) {
  int l = d->tok[tok].links;
  while (l >= 0) { int m = d->lnk[l].next; free_link(d, l); l = m; }
  d->tok[tok].links = -1;
}

/* ---------------------------------------------------------------- hash list */
static void hl_set_size(Dec *d, size_t size) {
  d->hash_size = size;
  if ((int)size > d->nbuckets) {
    d->buckets = (Bucket *)realloc(d->buckets, sizeof(Bucket) * size);
    for (size_t i = d->nbuckets; i < size; i++) { d->buckets[i].last_elem = -1; d->buckets[i].prev_bucket = 0; }
    d->nbuckets = (int)size;
  }
}
static int hl_new(Dec *d) {
  if (d->free_elem >= 0) { int e = d->free_elem; d->free_elem = d->elem[e].tail; return e; }
  if (d->nelem == d->capelem) { d->capelem = d->capelem ? d->capelem * 2 : 4096; d->elem = (Elem *)realloc(d->elem, sizeof(Elem) * d->capelem); }
  return d->nelem++;
}
static void hl_delete(Dec *d, int e) { d->elem[e].tail = d->free_elem; d->free_elem = e; }
static int hl_clear(Dec *d) {
  for (int b = d->bucket_list_tail; b != -1; b = d->buckets[b].prev_bucket) d->buckets[b].last_elem = -1;
  d->bucket_list_tail = -1;
  int ans = d->list_head;
  d->list_head = -1;
  return ans;
}
/* Insert(key, val): returns the element; existing element is returned untouched. */
static int hl_insert(Dec *d, int key, int val) {
  size_t index = (size_t)key % d->hash_size;
  Bucket *bk = &d->buckets[index];
  if (bk->last_elem != -1) {
    int head = (bk->prev_bucket == -1) ? d->list_head : d->elem[d->buckets[bk->prev_bucket].last_elem].tail;
    int tail = d->elem[bk->last_elem].tail;
    for (int e = head; e != tail; e = d->elem[e].tail) if (d->elem[e].key == key) return e;
  }
  int e = hl_new(d);
  bk = &d->buckets[index];
  d->elem[e].key = key; d->elem[e].val = val;
  if (bk->last_elem == -1) {
    if (d->bucket_list_tail == -1) d->list_head = e;
    else d->elem[d->buckets[d->bucket_list_tail].last_elem].tail = e;
    d->elem[e].tail = -1;
    bk->last_elem = e;
    bk->prev_bucket = d->bucket_list_tail;
    d->bucket_list_tail = (int)index;
  } else {
    d->elem[e].tail = d->elem[bk->last_elem].tail;
    d->elem[bk->last_elem].tail = e;
    bk->last_elem = e;
  }
  return e;
}
static void delete_elems(Dec *d, int list) {
  for (int e = list, t; e != -1; e = t) { t = d->elem[e].tail; hl_delete(d, e); }
}

/* ---------------------------------------------------------------- decoding */
static void push_frame(Dec *d) {
  if (d->nframes == d->capframes) {
    d->capframes = d->capframes ? d->capframes * 2 : 512;
    d->frames = (TokList *)realloc(d->frames, sizeof(TokList) * d->capframes);
    d->cost_offsets = (float *)realloc(d->cost_offsets, sizeof(float) * d->capframes);
  }
  d->frames[d->nframes].toks = -1;
  d->frames[d->nframes].must_prune_forward_links = 1;
  d->frames[d->nframes].must_prune_tokens = 1;
  d->cost_offsets[d->nframes] = 0.0f;
  d->nframes++;
}

static int find_or_add(Dec *d, int state, int frame_plus_one, float tot_cost, int backpointer, int *changed) {
  int e = hl_insert(d, state, -1);
  d->counters[2]++;
  if (d->elem[e].val == -1) {
    int t = new_tok(d, tot_cost, 0.0f, -1, d->frames[frame_plus_one].toks, backpointer, state);
    d->frames[frame_plus_one].toks = t;
    d->elem[e].val = t;
    if (changed) *changed = 1;
  } else {
    Tok *tk = &d->tok[d->elem[e].val];
    if (tk->tot_cost > tot_cost) { tk->tot_cost = tot_cost; tk->backpointer = backpointer; if (changed) *changed = 1; }
    else if (changed) *changed = 0;
  }
  return e;
}

static int cmp_float(const void *a, const void *b) { float x = *(const float *)a, y = *(const float *)b; return (x > y) - (x < y); }

static float get_cutoff(Dec *d, int list_head, size_t *tok_count, float *adaptive_beam, int *best_elem) {
  float best_weight = INF;
  size_t count = 0;
  for (int e = list_head; e != -1; e = d->elem[e].tail, count++) {
    if ((int)count >= d->captmp) { d->captmp = d->captmp ? d->captmp * 2 : 4096; d->tmp = (float *)realloc(d->tmp, sizeof(float) * d->captmp); }
    float w = d->tok[d->elem[e].val].tot_cost;
    d->tmp[count] = w;
    if (w < best_weight) { best_weight = w; *best_elem = e; }
  }
  *tok_count = count;
  float beam_cutoff = best_weight + d->beam, min_active_cutoff = INF, max_active_cutoff = INF;
  int sorted = 0;
  if (count > (size_t)d->max_active) {
    qsort(d->tmp, count, sizeof(float), cmp_float); sorted = 1;
    max_active_cutoff = d->tmp[d->max_active];
  }
  if (max_active_cutoff < beam_cutoff) {
    *adaptive_beam = max_active_cutoff - best_weight + d->beam_delta;
    d->counters[5]++;
    return max_active_cutoff;
  }
  if (count > (size_t)d->min_active) {
    if (d->min_active == 0) min_active_cutoff = best_weight;
    else {
      if (!sorted) qsort(d->tmp, count, sizeof(float), cmp_float);
      min_active_cutoff = d->tmp[d->min_active];
    }
  }
  if (min_active_cutoff > beam_cutoff) {
    *adaptive_beam = min_active_cutoff - best_weight + d->beam_delta;
    if (count > (size_t)d->min_active) d->counters[6]++;
    return min_active_cutoff;
  }
  *adaptive_beam = d->beam;
  return beam_cutoff;
}

static inline float loglike(const Dec *d, int frame, int tid) { return d->loglikes[(size_t)frame * d->P + d->id2pdf[tid]]; }

static float process_emitting(Dec *d) {
  int frame = d->nframes - 1;
  push_frame(d);
  int final_toks = hl_clear(d);
  int best_elem = -1;
  float adaptive_beam;
  size_t tok_cnt;
  float cur_cutoff = get_cutoff(d, final_toks, &tok_cnt, &adaptive_beam, &best_elem);
  size_t new_sz = (size_t)((float)tok_cnt * d->hash_ratio);
  if (new_sz > d->hash_size) hl_set_size(d, new_sz);
  float next_cutoff = INF, cost_offset = 0.0f;
  if (best_elem != -1) {
    int state = d->elem[best_elem].key;
    const Tok *tk = &d->tok[d->elem[best_elem].val];
    cost_offset = -tk->tot_cost;
    for (int64_t a = d->arc_begin[state]; a < d->arc_begin[state + 1]; a++) {
      if (d->ilabel[a] != 0) {
        float new_weight = d->weight[a] + cost_offset - loglike(d, frame, d->ilabel[a]) + tk->tot_cost;
        if (new_weight + adaptive_beam < next_cutoff) next_cutoff = new_weight + adaptive_beam;
      }
    }
  }
  d->cost_offsets[frame] = cost_offset;
  if (d->final_cutoff_mode) {     /* diagnostic variant: the cutoff every token of the frame is pruned with is the one the loop below ends with */
    for (int e = final_toks; e != -1; e = d->elem[e].tail) {
      int state = d->elem[e].key, tki = d->elem[e].val;
      if (d->tok[tki].tot_cost > cur_cutoff) continue;
      for (int64_t a = d->arc_begin[state]; a < d->arc_begin[state + 1]; a++)
        if (d->ilabel[a] != 0) {
          float tot_cost = d->tok[tki].tot_cost + (cost_offset - loglike(d, frame, d->ilabel[a])) + d->weight[a];
          if (tot_cost + adaptive_beam < next_cutoff) next_cutoff = tot_cost + adaptive_beam;
        }
    }
  }
  float best_in = best_elem != -1 ? d->tok[d->elem[best_elem].val].tot_cost : 0.0f;
  for (int e = final_toks, e_tail; e != -1; e = e_tail) {
    int state = d->elem[e].key, tki = d->elem[e].val;
    if (d->tok[tki].tot_cost <= cur_cutoff) {
      d->counters[0]++;
      for (int64_t a = d->arc_begin[state]; a < d->arc_begin[state + 1]; a++) {
        if (d->ilabel[a] != 0) {
          d->counters[1]++;
          float ac_cost = cost_offset - loglike(d, frame, d->ilabel[a]), graph_cost = d->weight[a],
                cur_cost = d->tok[tki].tot_cost, tot_cost = cur_cost + ac_cost + graph_cost;
          if (tot_cost >= next_cutoff) continue;
          else if (tot_cost + adaptive_beam < next_cutoff) next_cutoff = tot_cost + adaptive_beam;
          int e_next = find_or_add(d, d->nextstate[a], frame + 1, tot_cost, tki, NULL);
          d->tok[tki].links = new_link(d, d->elem[e_next].val, d->ilabel[a], d->olabel[a], graph_cost, ac_cost, d->tok[tki].links);
        }
      }
    }
    e_tail = d->elem[e].tail;
    hl_delete(d, e);
  }
  if (d->trace) {
    int made = 0, extras = 0;
    float worst = 0.0f;
    for (int t = d->frames[frame + 1].toks; t != -1; t = d->tok[t].next) {
      float m = d->tok[t].tot_cost - next_cutoff;
      d->tok[t].born_margin = m;
      made++;
      if (m >= 0.0f) { extras++; if (m > worst) worst = m; }
    }
    fprintf(d->trace, "frame %d tokens_in %zu cutoff-best %.4f adaptive_beam %.4f made %d extras %d worst_margin %.4f\n", frame, tok_cnt,
            cur_cutoff - best_in, adaptive_beam, made, extras, worst);
  }
  return next_cutoff;
}

static void process_nonemitting(Dec *d, float cutoff) {
  int frame = d->nframes - 2;
  d->nqueue = 0;
  for (int e = d->list_head; e != -1; e = d->elem[e].tail) {
    if (d->num_ieps[d->elem[e].key] != 0) {
      if (d->nqueue == d->capqueue) { d->capqueue = d->capqueue ? d->capqueue * 2 : 4096; d->queue = (int *)realloc(d->queue, sizeof(int) * d->capqueue); }
      d->queue[d->nqueue++] = e;
    }
  }
  while (d->nqueue > 0) {
    int e = d->queue[--d->nqueue];
    int state = d->elem[e].key, tki = d->elem[e].val;
    float cur_cost = d->tok[tki].tot_cost;
    if (cur_cost >= cutoff) continue;
    d->counters[0]++;
    delete_forward_links(d, tki);
    for (int64_t a = d->arc_begin[state]; a < d->arc_begin[state + 1]; a++) {
      if (d->ilabel[a] == 0) {
        d->counters[1]++;
        float graph_cost = d->weight[a], tot_cost = cur_cost + graph_cost;
        if (tot_cost < cutoff) {
          int changed;
          int e_new = find_or_add(d, d->nextstate[a], frame + 1, tot_cost, tki, &changed);
          d->tok[tki].links = new_link(d, d->elem[e_new].val, 0, d->olabel[a], graph_cost, 0.0f, d->tok[tki].links);
          if (changed && d->num_ieps[d->nextstate[a]] != 0) {
            if (d->nqueue == d->capqueue) { d->capqueue = d->capqueue ? d->capqueue * 2 : 4096; d->queue = (int *)realloc(d->queue, sizeof(int) * d->capqueue); }
            d->queue[d->nqueue++] = e_new;
          }
        }
      }
    }
  }
}

static void prune_forward_links(Dec *d, int f, int *extra_costs_changed, int *links_pruned, float delta) {
  *extra_costs_changed = 0; *links_pruned = 0;
  int changed = 1;
  while (changed) {
    changed = 0;
    for (int t = d->frames[f].toks; t != -1; t = d->tok[t].next) {
      Tok *tk = &d->tok[t];
      int prev = -1;
      float tok_extra = INF;
      for (int l = tk->links; l != -1;) {
        Link *lk = &d->lnk[l];
        const Tok *nt = &d->tok[lk->next_tok];
        float link_extra = nt->extra_cost + ((tk->tot_cost + lk->acoustic_cost + lk->graph_cost) - nt->tot_cost);
        if (link_extra > d->lattice_beam) {
          int nl = lk->next;
          if (prev != -1) d->lnk[prev].next = nl; else tk->links = nl;
          free_link(d, l);
          l = nl;
          *links_pruned = 1;
        } else {
          if (link_extra < 0.0f) link_extra = 0.0f;
          if (link_extra < tok_extra) tok_extra = link_extra;
          prev = l;
          l = lk->next;
        }
      }
      if (fabsf(tok_extra - tk->extra_cost) > delta) changed = 1;
      tk->extra_cost = tok_extra;
    }
    if (changed) *extra_costs_changed = 1;
  }
}

static void compute_final_costs(Dec *d, int fill, float *final_relative_cost, float *final_best_cost) {
  float best_cost = INF, best_with_final = INF;
  if (fill) {
    d->final_costs = (float *)realloc(d->final_costs, sizeof(float) * (d->ntok + 1));
    for (int i = 0; i < d->ntok; i++) d->final_costs[i] = NAN;
    d->have_final_costs = 0;
  }
  for (int e = d->list_head; e != -1; e = d->elem[e].tail) {
    int state = d->elem[e].key, t = d->elem[e].val;
    float fc = d->final[state], cost = d->tok[t].tot_cost, cwf = cost + fc;
    if (cost < best_cost) best_cost = cost;
    if (cwf < best_with_final) best_with_final = cwf;
    if (fill && fc != INF) { d->final_costs[t] = fc; d->have_final_costs = 1; }
  }
  if (final_relative_cost) *final_relative_cost = (best_cost == INF && best_with_final == INF) ? INF : best_with_final - best_cost;
  if (final_best_cost) *final_best_cost = (best_with_final != INF) ? best_with_final : best_cost;
}

static int approx_equal(float a, float b, float rel) {
  if (a == b) return 1;
  float diff = fabsf(a - b);
  if (diff == INF || diff != diff) return 0;
  return diff <= rel * (fabsf(a) + fabsf(b));
}

static void prune_forward_links_final(Dec *d) {
  int f = d->nframes - 1;
  compute_final_costs(d, 1, &d->final_relative_cost, &d->final_best_cost);
  d->finalized = 1;
  delete_elems(d, hl_clear(d));
  int changed = 1;
  const float delta = 1.0e-05f;
  while (changed) {
    changed = 0;
    for (int t = d->frames[f].toks; t != -1; t = d->tok[t].next) {
      Tok *tk = &d->tok[t];
      float final_cost;
      if (!d->have_final_costs) final_cost = 0.0f;
      else final_cost = (d->final_costs[t] == d->final_costs[t]) ? d->final_costs[t] : INF;
      float tok_extra = tk->tot_cost + final_cost - d->final_best_cost;
      int prev = -1;
      for (int l = tk->links; l != -1;) {
        Link *lk = &d->lnk[l];
        const Tok *nt = &d->tok[lk->next_tok];
        float link_extra = nt->extra_cost + ((tk->tot_cost + lk->acoustic_cost + lk->graph_cost) - nt->tot_cost);
        if (link_extra > d->lattice_beam) {
          int nl = lk->next;
          if (prev != -1) d->lnk[prev].next = nl; else tk->links = nl;
          free_link(d, l);
          l = nl;
        } else {
          if (link_extra < 0.0f) link_extra = 0.0f;
          if (link_extra < tok_extra) tok_extra = link_extra;
          prev = l;
          l = lk->next;
        }
      }
      if (tok_extra > d->lattice_beam) tok_extra = INF;
      if (!approx_equal(tk->extra_cost, tok_extra, delta)) changed = 1;
      tk->extra_cost = tok_extra;
    }
  }
}

static void prune_tokens_for_frame(Dec *d, int f) {
  int prev = -1;
  for (int t = d->frames[f].toks, nx; t != -1; t = nx) {
    nx = d->tok[t].next;
    if (d->tok[t].extra_cost == INF) {
      if (prev != -1) d->tok[prev].next = nx; else d->frames[f].toks = nx;
      delete_forward_links(d, t);
      d->tok[t].alive = 0;
      d->num_toks--;
    } else prev = t;
  }
}

static void prune_active_tokens(Dec *d, float delta) {
  int cur = d->nframes - 1;
  for (int f = cur - 1; f >= 0; f--) {
    if (d->frames[f].must_prune_forward_links) {
      int ecc = 0, lp = 0;
      prune_forward_links(d, f, &ecc, &lp, delta);
      if (ecc && f > 0) d->frames[f - 1].must_prune_forward_links = 1;
      if (lp) d->frames[f].must_prune_tokens = 1;
      d->frames[f].must_prune_forward_links = 0;
    }
    if (f + 1 < cur && d->frames[f + 1].must_prune_tokens) {
      prune_tokens_for_frame(d, f + 1);
      d->frames[f + 1].must_prune_tokens = 0;
    }
  }
}

/* ---------------------------------------------------------------- public API */
Dec *rs_oracle_decode(int num_states, int start, const float *final, const int64_t *arc_begin, const int64_t *num_ieps,
                      const int32_t *ilabel, const int32_t *olabel, const float *weight, const int32_t *nextstate,
                      const float *loglikes, int T, int P, const int32_t *id2pdf,
                      float beam, int max_active, int min_active, float lattice_beam, float beam_delta) {
  Dec *d = (Dec *)calloc(1, sizeof(Dec));
  d->num_states = num_states; d->start = start; d->final = final; d->arc_begin = arc_begin; d->num_ieps = num_ieps;
  d->ilabel = ilabel; d->olabel = olabel; d->weight = weight; d->nextstate = nextstate;
  d->loglikes = loglikes; d->T = T; d->P = P; d->id2pdf = id2pdf;
  d->beam = beam; d->max_active = max_active; d->min_active = min_active; d->lattice_beam = lattice_beam; d->beam_delta = beam_delta;
  d->prune_scale = 0.1f; d->hash_ratio = 2.0f; d->prune_interval = 25;
  d->free_lnk = -1; d->free_elem = -1; d->list_head = -1; d->bucket_list_tail = -1;
  { const char *e = getenv("RS_ORACLE_FINAL_CUTOFF"); d->final_cutoff_mode = e && e[0] == '1';
    e = getenv("RS_ORACLE_TRACE"); d->trace = e && e[0] ? fopen(e, "w") : NULL; }
  hl_set_size(d, 1000);      /* LatticeFasterDecoderTpl constructor: toks_.SetSize(1000) */
  /* InitDecoding */
  push_frame(d);
  int st = new_tok(d, 0.0f, 0.0f, -1, -1, -1, start);
  d->frames[0].toks = st;
  hl_insert(d, start, st);
  process_nonemitting(d, d->beam);
  for (int t = d->frames[0].toks; t != -1; t = d->tok[t].next) d->counters[7]++;      /* tokens of frame 0 (the start state's closure) */
  /* AdvanceDecoding */
  while (d->nframes - 1 < T) {
    if ((d->nframes - 1) % d->prune_interval == 0) prune_active_tokens(d, d->lattice_beam * d->prune_scale);
    float cutoff = process_emitting(d);
    process_nonemitting(d, cutoff);
    for (int t = d->frames[d->nframes - 1].toks; t != -1; t = d->tok[t].next) d->counters[3]++;
  }
  /* FinalizeDecoding */
  int final_frame_plus_one = d->nframes - 1;
  prune_forward_links_final(d);
  if (d->trace) {       /* the best path, by back-pointers from the best token of the last frame: which of its tokens were order-dependent extras */
    int best = -1; float bc = INF;
    for (int t = d->frames[final_frame_plus_one].toks; t != -1; t = d->tok[t].next) {
      float fc = !d->have_final_costs ? 0.0f : (d->final_costs[t] == d->final_costs[t] ? d->final_costs[t] : INF);
      if (d->tok[t].tot_cost + fc < bc) { bc = d->tok[t].tot_cost + fc; best = t; }
    }
    for (int t = best; t != -1; t = d->tok[t].backpointer)
      if (d->tok[t].born_margin >= 0.0f)
        fprintf(d->trace, "best_path frame %d state %d born %.4f ABOVE the frame's final cutoff\n", d->tok[t].frame, d->tok[t].state, d->tok[t].born_margin);
    fprintf(d->trace, "best_path end\n");
    fclose(d->trace); d->trace = NULL;
  }
  for (int f = final_frame_plus_one - 1; f >= 0; f--) {
    int b1, b2;
    prune_forward_links(d, f, &b1, &b2, 0.0f);
    prune_tokens_for_frame(d, f + 1);
  }
  prune_tokens_for_frame(d, 0);
  return d;
}

void rs_oracle_free(Dec *d) {
  if (!d) return;
  free(d->tok); free(d->lnk); free(d->frames); free(d->cost_offsets); free(d->elem); free(d->buckets); free(d->queue); free(d->tmp);
  free(d->final_costs); free(d);
}

/* Raw lattice (GetRawLattice :106-189): one state per surviving token, arcs = surviving forward links with the
 * per-frame cost offset removed from the acoustic cost, final weights from the final costs (or One() for all
 * last-frame tokens if no final state was reached).  Returns counts; fill the arrays on the second call. */
int rs_oracle_lattice_size(const Dec *d, int *num_arcs) {
  int ns = 0, na = 0;
  for (int f = 0; f < d->nframes; f++)
    for (int t = d->frames[f].toks; t != -1; t = d->tok[t].next) {
      ns++;
      for (int l = d->tok[t].links; l != -1; l = d->lnk[l].next) na++;
    }
  *num_arcs = na;
  return ns;
}

/* states are numbered frame by frame in token-list order; tok_id[i] = internal token index of lattice state i */
void rs_oracle_lattice_fill(const Dec *d, int32_t *state_frame, float *state_final, int32_t *arc_src, int32_t *arc_dst,
                            int32_t *arc_ilabel, int32_t *arc_olabel, float *arc_graph, float *arc_acoustic, int32_t *start_state) {
  int *map = (int *)malloc(sizeof(int) * (d->ntok + 1));
  int ns = 0;
  for (int f = 0; f < d->nframes; f++)
    for (int t = d->frames[f].toks; t != -1; t = d->tok[t].next) { map[t] = ns; state_frame[ns] = f; ns++; }
  int na = 0, last = d->nframes - 1;
  *start_state = -1;
  for (int f = 0; f < d->nframes; f++)
    for (int t = d->frames[f].toks; t != -1; t = d->tok[t].next) {
      int s = map[t];
      if (f == 0 && d->tok[t].backpointer == -1 && d->tok[t].state == d->start && *start_state == -1) *start_state = s;
      for (int l = d->tok[t].links; l != -1; l = d->lnk[l].next) {
        const Link *lk = &d->lnk[l];
        float off = lk->ilabel != 0 ? d->cost_offsets[f] : 0.0f;
        arc_src[na] = s; arc_dst[na] = map[lk->next_tok]; arc_ilabel[na] = lk->ilabel; arc_olabel[na] = lk->olabel;
        arc_graph[na] = lk->graph_cost; arc_acoustic[na] = lk->acoustic_cost - off;
        na++;
      }
      state_final[s] = INF;
      if (f == last) {
        if (d->have_final_costs) { if (d->final_costs[t] == d->final_costs[t]) state_final[s] = d->final_costs[t]; }
        else state_final[s] = 0.0f;
      }
    }
  free(map);
}

void rs_oracle_counters(const Dec *d, int64_t out[8]) { for (int i = 0; i < 8; i++) out[i] = d->counters[i]; }
