// oracle/orc_decoder.cc -- TEST INFRASTRUCTURE ONLY (CPU oracle; never shipped).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load it.
//
// CPU restatement of kaldi::LatticeFasterDecoderTpl<ConstFst<StdArc>, StdToken>
// (decoder/lattice-faster-decoder.{h,cc}) and of the HashList it iterates
// (util/hash-list-inl.h).  PARITY UNPINNED: the reference decoder cannot be built
// in this image (needs OpenFst 1.6.7 headers, tools/Makefile:10, absent) and the
// reference has no decoder test vectors (decoder/Makefile: TESTFILES empty), so
// this file is pinned only by (a) line-by-line correspondence, (b) brute-force
// path enumeration on tiny graphs (tests/test_oracle_decoder.py) and (c) the
// invariants the reference itself asserts.
//
// Two modes:
//   mode 0 "faithful": same token visiting order (HashList bucket order), same
//     running next_cutoff, same periodic PruneActiveTokens(delta) as the reference.
//   mode 1 "canonical": the order-independent restatement the HIP kernels are
//     bit-exact against: next_cutoff is the value the reference ends the frame
//     with (min over all arcs), arcs are kept iff tot <= that final cutoff, no
//     intermediate pruning (it is provably conservative), final pruning iterated
//     to the exact fixpoint.  See DESIGN.md "Decoder parity".
//   mode 2 "canonical-loose": as mode 1, but on the frames where max_active /
//     min_active made the adaptive beam differ from the beam an arc is kept iff
//     tot <= the SEED cutoff (best token's arcs + adaptive beam, :757-772), the
//     loosest value the reference's running bound takes; such a frame's tokens are
//     then a superset of what the reference creates in ANY visiting order.  On the
//     other frames the tokens between the two bounds can never be expanded (the
//     next frame's cutoff best + beam IS this frame's final bound) and mode 2 = 1.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>

#include "../include/kaldi_amd.h"

namespace {

const float kInf = std::numeric_limits<float>::infinity();

struct Tok {    // decoder::StdToken (lattice-faster-decoder.h:121-161)
  float tot_cost, extra_cost;
  int links;    // head of ForwardLink list
  int next;     // next token on this frame
  int state;    // HCLG state (annotation only; the reference discards it)
  bool deleted;
};
struct Link {   // decoder::ForwardLink (lattice-faster-decoder.h:103-118)
  int next_tok, ilabel, olabel;
  float graph_cost, acoustic_cost;
  int next;
};
struct TokenList {  // lattice-faster-decoder.h:361-369
  int toks;
  bool must_prune_forward_links, must_prune_tokens;
  TokenList() : toks(-1), must_prune_forward_links(true), must_prune_tokens(true) {}
};

// util/hash-list-inl.h restated with indices instead of pointers.
struct HashList {
  struct Elem { int key, val, tail; };
  struct Bucket { size_t prev_bucket; int last_elem; };
  std::vector<Elem> elems;
  std::vector<Bucket> buckets;
  size_t hash_size, bucket_list_tail;
  int list_head;
  HashList() : hash_size(0), bucket_list_tail(static_cast<size_t>(-1)), list_head(-1) {}
  void SetSize(size_t size) {  // :38-44
    hash_size = size;
    if (size > buckets.size()) {
      Bucket b; b.prev_bucket = 0; b.last_elem = -1;
      buckets.resize(size, b);
    }
  }
  size_t Size() const { return hash_size; }
  int Clear() {  // :46-60
    for (size_t cur = bucket_list_tail; cur != static_cast<size_t>(-1);
         cur = buckets[cur].prev_bucket)
      buckets[cur].last_elem = -1;
    bucket_list_tail = static_cast<size_t>(-1);
    int ans = list_head;
    list_head = -1;
    return ans;
  }
  int Find(int key) const {  // :73-88
    size_t index = static_cast<size_t>(key) % hash_size;
    const Bucket &b = buckets[index];
    if (b.last_elem == -1) return -1;
    int head = (b.prev_bucket == static_cast<size_t>(-1)
                    ? list_head
                    : elems[buckets[b.prev_bucket].last_elem].tail);
    int tail = elems[b.last_elem].tail;
    for (int e = head; e != tail; e = elems[e].tail)
      if (elems[e].key == key) return e;
    return -1;
  }
  void Insert(int key, int val) {  // :127-157
    size_t index = static_cast<size_t>(key) % hash_size;
    Bucket &b = buckets[index];
    Elem el; el.key = key; el.val = val; el.tail = -1;
    int e = static_cast<int>(elems.size());
    elems.push_back(el);
    if (b.last_elem == -1) {
      if (bucket_list_tail == static_cast<size_t>(-1)) list_head = e;
      else elems[buckets[bucket_list_tail].last_elem].tail = e;
      elems[e].tail = -1;
      b.last_elem = e;
      b.prev_bucket = bucket_list_tail;
      bucket_list_tail = index;
    } else {
      elems[e].tail = elems[b.last_elem].tail;
      elems[b.last_elem].tail = e;
      b.last_elem = e;
    }
  }
  // the elems vector is reset whenever the list is handed over and consumed
  void ResetPoolIfEmpty() { if (list_head == -1) elems.clear(); }
};

// The decoding graph, shared read-only by every decoder built on it (the reference hands one
// const fst::Fst& to all decoder objects, nnet-batch-compute.h:610-611).
struct GraphStore {
  int num_states, start;
  std::vector<int64_t> arc_off;
  std::vector<kamd_arc> arcs;
  std::vector<float> final_cost;
  std::vector<uint8_t> has_eps;  // NumInputEpsilons(state) != 0
};

struct Decoder {
  // graph (OpenFst arc order preserved: the faithful mode depends on it)
  int num_states, start;
  const int64_t *arc_off;
  const kamd_arc *arcs;
  const float *final_cost;
  const uint8_t *has_eps;
  GraphStore *own_graph = NULL;   // set when the decoder made its own copy (orc_decoder_create)
  kamd_decoder_config cfg;
  std::vector<int32_t> tid2pdf;
  bool identity_map;
  int mode;

  std::vector<Tok> toks;
  std::vector<Link> links;
  std::vector<TokenList> active_toks;
  HashList hl;
  std::vector<HashList::Elem> prev_elems;  // list handed over by Clear()
  std::vector<int> queue;
  std::vector<float> cost_offsets, tmp_array;
  int num_toks;
  bool decoding_finalized;
  std::vector<std::pair<int, float> > final_costs;  // tok -> final cost
  std::vector<float> final_cost_of_tok;             // dense copy, kInf = absent
  bool final_costs_empty;
  float final_relative_cost, final_best_cost;
  // traces / counters
  std::vector<int32_t> trace_ntok;
  std::vector<float> trace_cutoff, trace_offset;
  int64_t counters[8];

  int NumFramesDecoded() const { return static_cast<int>(active_toks.size()) - 1; }

  float LogLike(const float *row, int ilabel) const {
    // DecodableMatrixMapped::LogLikelihood (decoder/decodable-matrix.cc:62-69)
    int pdf = identity_map ? ilabel - 1 : tid2pdf[ilabel];
    return row[pdf];
  }

  int NewTok(float tot, float extra, int next, int state) {
    Tok t; t.tot_cost = tot; t.extra_cost = extra; t.links = -1; t.next = next;
    t.state = state; t.deleted = false;
    toks.push_back(t);
    return static_cast<int>(toks.size()) - 1;
  }
  int NewLink(int next_tok, int il, int ol, float g, float a, int next) {
    Link l; l.next_tok = next_tok; l.ilabel = il; l.olabel = ol; l.graph_cost = g;
    l.acoustic_cost = a; l.next = next;
    links.push_back(l);
    if (mode == 0 || il != 0) counters[4]++;
    return static_cast<int>(links.size()) - 1;
  }

  // lattice-faster-decoder.cc:56-73
  void InitDecoding() {
    toks.clear(); links.clear(); active_toks.clear(); cost_offsets.clear();
    hl.Clear(); hl.elems.clear();  // hash size persists across utterances (:35 is ctor-only)
    num_toks = 0; decoding_finalized = false; final_costs.clear();
    trace_ntok.clear(); trace_cutoff.clear(); trace_offset.clear();
    memset(counters, 0, sizeof(counters));
    active_toks.resize(1);
    int start_tok = NewTok(0.0f, 0.0f, -1, start);
    active_toks[0].toks = start_tok;
    hl.Insert(start, start_tok);
    num_toks++;
    counters[5]++;
    ProcessNonemitting(cfg.beam);
  }

  // lattice-faster-decoder.cc:266-306
  int FindOrAddToken(int state, int frame_plus_one, float tot_cost, bool *changed) {
    int &head = active_toks[frame_plus_one].toks;
    int e = hl.Find(state);
    if (e == -1) {
      int t = NewTok(tot_cost, 0.0f, head, state);
      head = t;
      num_toks++;
      counters[5]++;
      hl.Insert(state, t);
      if (changed) *changed = true;
      return t;
    }
    int t = hl.elems[e].val;
    if (toks[t].tot_cost > tot_cost) {
      toks[t].tot_cost = tot_cost;
      if (changed) *changed = true;
    } else if (changed) {
      *changed = false;
    }
    return t;
  }

  // lattice-faster-decoder.cc:657-724.  'list' = elems handed over by Clear().
  float GetCutoff(int list_head, size_t *tok_count, float *adaptive_beam,
                  int *best_elem) {
    float best_weight = kInf;
    size_t count = 0;
    *best_elem = -1;
    const std::vector<HashList::Elem> &E = hl.elems;
    if (cfg.max_active == std::numeric_limits<int32_t>::max() && cfg.min_active == 0) {
      for (int e = list_head; e != -1; e = E[e].tail, count++) {
        float w = toks[E[e].val].tot_cost;
        if (BetterBest(w, best_weight, E[e].key, *best_elem)) {
          best_weight = w; *best_elem = e;
        }
      }
      *tok_count = count;
      *adaptive_beam = cfg.beam;
      return best_weight + cfg.beam;
    }
    tmp_array.clear();
    for (int e = list_head; e != -1; e = E[e].tail, count++) {
      float w = toks[E[e].val].tot_cost;
      tmp_array.push_back(w);
      if (BetterBest(w, best_weight, E[e].key, *best_elem)) {
        best_weight = w; *best_elem = e;
      }
    }
    *tok_count = count;
    float beam_cutoff = best_weight + cfg.beam, min_active_cutoff = kInf,
          max_active_cutoff = kInf;
    if (tmp_array.size() > static_cast<size_t>(cfg.max_active)) {
      std::nth_element(tmp_array.begin(), tmp_array.begin() + cfg.max_active,
                       tmp_array.end());
      max_active_cutoff = tmp_array[cfg.max_active];
    }
    if (max_active_cutoff < beam_cutoff) {
      *adaptive_beam = max_active_cutoff - best_weight + cfg.beam_delta;
      return max_active_cutoff;
    }
    if (tmp_array.size() > static_cast<size_t>(cfg.min_active)) {
      if (cfg.min_active == 0) min_active_cutoff = best_weight;
      else {
        std::nth_element(tmp_array.begin(), tmp_array.begin() + cfg.min_active,
                         tmp_array.size() > static_cast<size_t>(cfg.max_active)
                             ? tmp_array.begin() + cfg.max_active
                             : tmp_array.end());
        min_active_cutoff = tmp_array[cfg.min_active];
      }
    }
    if (min_active_cutoff > beam_cutoff) {
      *adaptive_beam = min_active_cutoff - best_weight + cfg.beam_delta;
      return min_active_cutoff;
    }
    *adaptive_beam = cfg.beam;
    return beam_cutoff;
  }
  // faithful: first strictly-better in list order (:666,679).  canonical: ties
  // broken by smallest HCLG state so the choice is order independent.
  bool BetterBest(float w, float best, int key, int best_elem) const {
    if (w < best) return true;
    if (mode >= 1 && w == best && best_elem != -1 && key < hl.elems[best_elem].key)
      return true;
    return false;
  }

  // lattice-faster-decoder.cc:727-817
  float ProcessEmitting(const float *loglike_row) {
    int frame = static_cast<int>(active_toks.size()) - 1;
    active_toks.resize(active_toks.size() + 1);
    int final_toks = hl.Clear();
    int best_elem;
    float adaptive_beam;
    size_t tok_cnt;
    float cur_cutoff = GetCutoff(final_toks, &tok_cnt, &adaptive_beam, &best_elem);
    // PossiblyResizeHash (:226-232)
    size_t new_sz = static_cast<size_t>(static_cast<float>(tok_cnt) * cfg.hash_ratio);
    if (new_sz > hl.Size()) hl.SetSize(new_sz);
    // the handed-over list lives in hl.elems; copy it out so the pool can be
    // reused for the new frame's list (the reference recycles Elems via Delete()).
    prev_elems.clear();
    for (int e = final_toks; e != -1; e = hl.elems[e].tail) prev_elems.push_back(hl.elems[e]);
    int best_pos = -1;
    {
      int pos = 0;
      for (int e = final_toks; e != -1; e = hl.elems[e].tail, pos++)
        if (e == best_elem) best_pos = pos;
    }
    hl.elems.clear();

    float next_cutoff = kInf, cost_offset = 0.0f;
    if (best_pos >= 0) {  // :757-772
      int state = prev_elems[best_pos].key;
      const Tok &tok = toks[prev_elems[best_pos].val];
      cost_offset = -tok.tot_cost;
      for (int64_t a = arc_off[state]; a < arc_off[state + 1]; a++) {
        const kamd_arc &arc = arcs[a];
        if (arc.ilabel != 0) {
          float new_weight = arc.weight + cost_offset - LogLike(loglike_row, arc.ilabel) +
                             tok.tot_cost;
          if (new_weight + adaptive_beam < next_cutoff)
            next_cutoff = new_weight + adaptive_beam;
        }
      }
    }
    cost_offsets.resize(frame + 1, 0.0f);
    cost_offsets[frame] = cost_offset;
    trace_ntok.push_back(static_cast<int32_t>(tok_cnt));
    trace_cutoff.push_back(cur_cutoff);
    trace_offset.push_back(cost_offset);

    // mode 2 (canonical-loose): an arc is kept iff its cost is within the SEED bound -- the loosest value the
    // reference's running next_cutoff can have when the arc is visited (it starts there, :757-772, and only
    // tightens) -- so every token the reference creates, in whatever order it visits them, is created too.
    const float seed_cutoff = next_cutoff;
    // ... on the frames where the extras can matter: max_active / min_active made the adaptive beam differ from the beam.
    // With adaptive_beam == beam the next frame's cutoff (best + beam) is this frame's final bound: extras are never expanded.
    const bool loose = mode == 2 && adaptive_beam != cfg.beam;
    if (mode >= 1) {
      // canonical: the value the reference's running next_cutoff ends with.
      for (size_t i = 0; i < prev_elems.size(); i++) {
        const Tok &tok = toks[prev_elems[i].val];
        if (tok.tot_cost <= cur_cutoff) {
          int state = prev_elems[i].key;
          for (int64_t a = arc_off[state]; a < arc_off[state + 1]; a++) {
            const kamd_arc &arc = arcs[a];
            if (arc.ilabel != 0) {
              float ac_cost = cost_offset - LogLike(loglike_row, arc.ilabel),
                    graph_cost = arc.weight, cur_cost = tok.tot_cost,
                    tot_cost = cur_cost + ac_cost + graph_cost;
              if (tot_cost + adaptive_beam < next_cutoff)
                next_cutoff = tot_cost + adaptive_beam;
            }
          }
        }
      }
    }

    for (size_t i = 0; i < prev_elems.size(); i++) {  // :783-815
      int state = prev_elems[i].key;
      int t = prev_elems[i].val;
      if (toks[t].tot_cost <= cur_cutoff) {
        counters[0]++;
        for (int64_t a = arc_off[state]; a < arc_off[state + 1]; a++) {
          const kamd_arc &arc = arcs[a];
          if (mode == 0) counters[1]++;
          if (arc.ilabel != 0) {
            if (mode >= 1) counters[1]++;
            counters[2]++;
            float ac_cost = cost_offset - LogLike(loglike_row, arc.ilabel),
                  graph_cost = arc.weight, cur_cost = toks[t].tot_cost,
                  tot_cost = cur_cost + ac_cost + graph_cost;
            if (tot_cost > (loose ? seed_cutoff : next_cutoff)) continue;
            else if (tot_cost + adaptive_beam < next_cutoff)
              next_cutoff = tot_cost + adaptive_beam;
            counters[3]++;
            int next_tok = FindOrAddToken(arc.nextstate, frame + 1, tot_cost, NULL);
            toks[t].links = NewLink(next_tok, arc.ilabel, arc.olabel, graph_cost,
                                    ac_cost, toks[t].links);
          }
        }
      }
    }
    return next_cutoff;
  }

  void DeleteForwardLinks(int t) { toks[t].links = -1; }

  // lattice-faster-decoder.cc:833-899
  void ProcessNonemitting(float cutoff) {
    int frame = static_cast<int>(active_toks.size()) - 2;
    queue.clear();
    for (int e = hl.list_head; e != -1; e = hl.elems[e].tail) {
      int state = hl.elems[e].key;
      if (has_eps[state]) queue.push_back(state);
    }
    while (!queue.empty()) {
      int state = queue.back();
      queue.pop_back();
      int t = hl.elems[hl.Find(state)].val;
      float cur_cost = toks[t].tot_cost;
      if (cur_cost > cutoff) continue;
      DeleteForwardLinks(t);
      for (int64_t a = arc_off[state]; a < arc_off[state + 1]; a++) {
        const kamd_arc &arc = arcs[a];
        if (arc.ilabel == 0) {
          if (mode == 0) counters[1]++;
          float graph_cost = arc.weight, tot_cost = cur_cost + graph_cost;
          if (tot_cost < cutoff) {
            bool changed;
            int new_tok = FindOrAddToken(arc.nextstate, frame + 1, tot_cost, &changed);
            toks[t].links = NewLink(new_tok, 0, arc.olabel, graph_cost, 0.0f,
                                    toks[t].links);
            if (changed && has_eps[arc.nextstate]) queue.push_back(arc.nextstate);
          }
        }
      }
    }
    if (mode >= 1) {
      // canonical work counters (order independent): epsilon arcs of every token
      // that ends the frame within the cutoff, and the epsilon links it keeps.
      for (int e = hl.list_head; e != -1; e = hl.elems[e].tail) {
        int state = hl.elems[e].key, t = hl.elems[e].val;
        if (!has_eps[state] || toks[t].tot_cost > cutoff) continue;
        for (int64_t a = arc_off[state]; a < arc_off[state + 1]; a++)
          if (arcs[a].ilabel == 0) counters[1]++;
        for (int l = toks[t].links; l != -1; l = links[l].next) counters[4]++;
      }
    }
  }

  // lattice-faster-decoder.cc:312-383
  void PruneForwardLinks(int frame_plus_one, bool *extra_costs_changed,
                         bool *links_pruned, float delta) {
    *extra_costs_changed = false;
    *links_pruned = false;
    bool changed = true;
    while (changed) {
      changed = false;
      for (int t = active_toks[frame_plus_one].toks; t != -1; t = toks[t].next) {
        Tok &tok = toks[t];
        int prev_link = -1;
        float tok_extra_cost = kInf;
        for (int l = tok.links; l != -1;) {
          Link &link = links[l];
          const Tok &next_tok = toks[link.next_tok];
          float link_extra_cost =
              next_tok.extra_cost +
              ((tok.tot_cost + link.acoustic_cost + link.graph_cost) - next_tok.tot_cost);
          if (link_extra_cost > cfg.lattice_beam) {
            int next_link = link.next;
            if (prev_link != -1) links[prev_link].next = next_link;
            else tok.links = next_link;
            l = next_link;
            *links_pruned = true;
          } else {
            if (link_extra_cost < 0.0f) link_extra_cost = 0.0f;
            if (link_extra_cost < tok_extra_cost) tok_extra_cost = link_extra_cost;
            prev_link = l;
            l = link.next;
          }
        }
        if (std::fabs(tok_extra_cost - tok.extra_cost) > delta) changed = true;
        tok.extra_cost = tok_extra_cost;
      }
      if (changed) *extra_costs_changed = true;
    }
  }

  // lattice-faster-decoder.cc:549-590
  void ComputeFinalCosts(std::vector<std::pair<int, float> > *fc, float *rel, float *bestc) {
    if (fc) fc->clear();
    float best_cost = kInf, best_cost_with_final = kInf;
    for (int e = hl.list_head; e != -1; e = hl.elems[e].tail) {
      int state = hl.elems[e].key, t = hl.elems[e].val;
      float fcst = final_cost[state];
      float cost = toks[t].tot_cost, cost_with_final = cost + fcst;
      best_cost = std::min(cost, best_cost);
      best_cost_with_final = std::min(cost_with_final, best_cost_with_final);
      if (fc && fcst != kInf) fc->push_back(std::make_pair(t, fcst));
    }
    if (rel) {
      if (best_cost == kInf && best_cost_with_final == kInf) *rel = kInf;
      else *rel = best_cost_with_final - best_cost;
    }
    if (bestc) *bestc = (best_cost_with_final != kInf) ? best_cost_with_final : best_cost;
  }

  static bool ApproxEqual(float a, float b, float tol) {  // base/kaldi-math.h:262-269
    if (a == b) return true;
    float diff = std::fabs(a - b);
    if (diff == kInf || diff != diff) return false;
    return diff <= tol * (std::fabs(a) + std::fabs(b));
  }

  void DenseFinalCosts() {
    final_cost_of_tok.assign(toks.size(), kInf);
    for (size_t i = 0; i < final_costs.size(); i++)
      final_cost_of_tok[final_costs[i].first] = final_costs[i].second;
    final_costs_empty = final_costs.empty();
  }

  // lattice-faster-decoder.cc:389-471
  void PruneForwardLinksFinal() {
    int frame_plus_one = static_cast<int>(active_toks.size()) - 1;
    ComputeFinalCosts(&final_costs, &final_relative_cost, &final_best_cost);
    DenseFinalCosts();
    decoding_finalized = true;
    hl.Clear();
    bool changed = true;
    float delta = 1.0e-05f;
    while (changed) {
      changed = false;
      for (int t = active_toks[frame_plus_one].toks; t != -1; t = toks[t].next) {
        Tok &tok = toks[t];
        int prev_link = -1;
        float fcst = final_costs_empty ? 0.0f : final_cost_of_tok[t];
        float tok_extra_cost = tok.tot_cost + fcst - final_best_cost;
        for (int l = tok.links; l != -1;) {
          Link &link = links[l];
          const Tok &next_tok = toks[link.next_tok];
          float link_extra_cost =
              next_tok.extra_cost +
              ((tok.tot_cost + link.acoustic_cost + link.graph_cost) - next_tok.tot_cost);
          if (link_extra_cost > cfg.lattice_beam) {
            int next_link = link.next;
            if (prev_link != -1) links[prev_link].next = next_link;
            else tok.links = next_link;
            l = next_link;
          } else {
            if (link_extra_cost < 0.0f) link_extra_cost = 0.0f;
            if (link_extra_cost < tok_extra_cost) tok_extra_cost = link_extra_cost;
            prev_link = l;
            l = link.next;
          }
        }
        if (tok_extra_cost > cfg.lattice_beam) tok_extra_cost = kInf;
        bool same = (mode >= 1) ? (tok.extra_cost == tok_extra_cost)
                                : ApproxEqual(tok.extra_cost, tok_extra_cost, delta);
        if (!same) changed = true;
        tok.extra_cost = tok_extra_cost;
      }
    }
  }

  // lattice-faster-decoder.cc:492-511
  void PruneTokensForFrame(int frame_plus_one) {
    int &head = active_toks[frame_plus_one].toks;
    int prev = -1, next;
    for (int t = head; t != -1; t = next) {
      next = toks[t].next;
      if (toks[t].extra_cost == kInf) {
        if (prev != -1) toks[prev].next = next;
        else head = next;
        toks[t].deleted = true;
        num_toks--;
      } else {
        prev = t;
      }
    }
  }

  // lattice-faster-decoder.cc:519-546
  void PruneActiveTokens(float delta) {
    int cur_frame_plus_one = NumFramesDecoded();
    for (int f = cur_frame_plus_one - 1; f >= 0; f--) {
      if (active_toks[f].must_prune_forward_links) {
        bool extra_costs_changed = false, links_pruned = false;
        PruneForwardLinks(f, &extra_costs_changed, &links_pruned, delta);
        if (extra_costs_changed && f > 0) active_toks[f - 1].must_prune_forward_links = true;
        if (links_pruned) active_toks[f].must_prune_tokens = true;
        active_toks[f].must_prune_forward_links = false;
      }
      if (f + 1 < cur_frame_plus_one && active_toks[f + 1].must_prune_tokens) {
        PruneTokensForFrame(f + 1);
        active_toks[f + 1].must_prune_tokens = false;
      }
    }
  }

  // lattice-faster-decoder.cc:593-632 (the decodable is a dense matrix here).
  void AdvanceDecoding(const float *loglikes, int ld, int n_frames) {
    for (int i = 0; i < n_frames; i++) {
      if (mode == 0 && NumFramesDecoded() % cfg.prune_interval == 0)
        PruneActiveTokens(cfg.lattice_beam * cfg.prune_scale);
      float cost_cutoff = ProcessEmitting(loglikes + static_cast<int64_t>(i) * ld);
      ProcessNonemitting(cost_cutoff);
      counters[6]++;
    }
  }

  // lattice-faster-decoder.cc:638-653
  void FinalizeDecoding() {
    int final_frame_plus_one = NumFramesDecoded();
    PruneForwardLinksFinal();
    for (int f = final_frame_plus_one - 1; f >= 0; f--) {
      bool b1, b2;
      PruneForwardLinks(f, &b1, &b2, 0.0f);
      PruneTokensForFrame(f + 1);
    }
    PruneTokensForFrame(0);
  }

  // ---- GetRawLattice (lattice-faster-decoder.cc:113-196) with canonical state
  // numbering: frame by frame, by HCLG state inside a frame.
  struct RawLat {
    std::vector<int32_t> frame, hclg;
    std::vector<float> cost, final;
    std::vector<kamd_lat_arc> arcs;
    int start;
  };
  bool GetRawLattice(RawLat *out, bool use_final_probs = true) {
    std::vector<std::pair<int, float> > fc_local;
    const std::vector<std::pair<int, float> > *fc = &final_costs;
    if (!decoding_finalized) {
      ComputeFinalCosts(&fc_local, NULL, NULL);
      fc = &fc_local;
    }
    std::vector<float> fdense(toks.size(), kInf);
    for (size_t i = 0; i < fc->size(); i++) fdense[(*fc)[i].first] = (*fc)[i].second;
    int num_frames = static_cast<int>(active_toks.size()) - 1;
    std::vector<int> tok2state(toks.size(), -1);
    out->frame.clear(); out->hclg.clear(); out->cost.clear(); out->final.clear();
    out->arcs.clear();
    out->start = -1;
    for (int f = 0; f <= num_frames; f++) {
      if (active_toks[f].toks == -1) return false;  // :145-149
      std::vector<std::pair<int, int> > v;  // (hclg state, tok)
      for (int t = active_toks[f].toks; t != -1; t = toks[t].next)
        v.push_back(std::make_pair(toks[t].state, t));
      std::sort(v.begin(), v.end());
      for (size_t i = 0; i < v.size(); i++) {
        tok2state[v[i].second] = static_cast<int>(out->frame.size());
        if (f == 0 && v[i].first == start) out->start = static_cast<int>(out->frame.size());
        out->frame.push_back(f);
        out->hclg.push_back(v[i].first);
        out->cost.push_back(toks[v[i].second].tot_cost);
        float fin = kInf;
        if (f == num_frames) {  // :183-192
          if (use_final_probs && !fc->empty()) fin = fdense[v[i].second];
          else fin = 0.0f;  // LatticeWeight::One()
        }
        out->final.push_back(fin);
      }
    }
    for (int f = 0; f <= num_frames; f++)
      for (int t = active_toks[f].toks; t != -1; t = toks[t].next)
        for (int l = toks[t].links; l != -1; l = links[l].next) {
          float cost_offset = 0.0f;
          if (links[l].ilabel != 0) cost_offset = cost_offsets[f];  // :173-177
          kamd_lat_arc a;
          a.src = tok2state[t];
          a.dst = tok2state[links[l].next_tok];
          a.ilabel = links[l].ilabel; a.olabel = links[l].olabel;
          a.graph_cost = links[l].graph_cost;
          a.acoustic_cost = links[l].acoustic_cost - cost_offset;
          out->arcs.push_back(a);
        }
    std::sort(out->arcs.begin(), out->arcs.end(), ArcLess);
    return !out->frame.empty();
  }
  static bool ArcLess(const kamd_lat_arc &a, const kamd_lat_arc &b) {
    if (a.src != b.src) return a.src < b.src;
    if (a.dst != b.dst) return a.dst < b.dst;
    if (a.ilabel != b.ilabel) return a.ilabel < b.ilabel;
    if (a.olabel != b.olabel) return a.olabel < b.olabel;
    if (a.graph_cost != b.graph_cost) return a.graph_cost < b.graph_cost;
    return a.acoustic_cost < b.acoustic_cost;
  }
};

// fstext/lattice-weight.h: Compare(w1, w2): by (v1+v2), then v1.  Returns true if
// a is strictly better (smaller) than b in the natural order.
inline bool LatBetter(float a1, float a2, float b1, float b2) {
  float fa = a1 + a2, fb = b1 + b2;
  if (fa < fb) return true;
  if (fa > fb) return false;
  return a1 < b1;
}

}  // namespace

extern "C" {

typedef struct Decoder orc_decoder;

typedef struct GraphStore orc_graph;
orc_graph *orc_graph_create(int32_t num_states, int32_t start, const int64_t *arc_off, const kamd_arc *arcs, const float *final_cost) {
  GraphStore *g = new GraphStore();
  g->num_states = num_states; g->start = start;
  g->arc_off.assign(arc_off, arc_off + num_states + 1);
  g->arcs.assign(arcs, arcs + arc_off[num_states]);
  g->final_cost.assign(final_cost, final_cost + num_states);
  g->has_eps.assign(num_states, 0);
  for (int s = 0; s < num_states; s++)
    for (int64_t a = arc_off[s]; a < arc_off[s + 1]; a++)
      if (arcs[a].ilabel == 0) { g->has_eps[s] = 1; break; }
  return g;
}
void orc_graph_destroy(orc_graph *g) { delete g; }

// a decoder on a shared graph (the graph must outlive it)
orc_decoder *orc_decoder_create_on(const orc_graph *g, const kamd_decoder_config *cfg, const int32_t *tid2pdf,
                                   int32_t num_tids, int mode) {
  Decoder *d = new Decoder();
  d->num_states = g->num_states; d->start = g->start;
  d->arc_off = g->arc_off.data(); d->arcs = g->arcs.data(); d->final_cost = g->final_cost.data(); d->has_eps = g->has_eps.data();
  d->cfg = *cfg;
  d->identity_map = (tid2pdf == NULL);
  if (tid2pdf) d->tid2pdf.assign(tid2pdf, tid2pdf + num_tids + 1);
  d->mode = mode;
  d->hl.SetSize(1000);  // lattice-faster-decoder.cc:35
  d->num_toks = 0; d->decoding_finalized = false;
  d->final_relative_cost = kInf; d->final_best_cost = kInf;
  memset(d->counters, 0, sizeof(d->counters));
  return d;
}
orc_decoder *orc_decoder_create(int32_t num_states, int32_t start, const int64_t *arc_off,
                                const kamd_arc *arcs, const float *final_cost,
                                const kamd_decoder_config *cfg, const int32_t *tid2pdf,
                                int32_t num_tids, int mode) {
  GraphStore *g = orc_graph_create(num_states, start, arc_off, arcs, final_cost);
  Decoder *d = orc_decoder_create_on(g, cfg, tid2pdf, num_tids, mode);
  d->own_graph = g;
  return d;
}
void orc_decoder_destroy(orc_decoder *d) { if (d) { delete d->own_graph; delete d; } }
void orc_decoder_init(orc_decoder *d) { d->InitDecoding(); }
void orc_decoder_advance(orc_decoder *d, const float *loglikes, int ld, int n_frames) {
  d->AdvanceDecoding(loglikes, ld, n_frames);
}
void orc_decoder_finalize(orc_decoder *d) { d->FinalizeDecoding(); }
int orc_decoder_num_frames_decoded(orc_decoder *d) { return d->NumFramesDecoded(); }
float orc_decoder_final_relative_cost(orc_decoder *d) {  // :474-484
  if (!d->decoding_finalized) {
    float rel;
    d->ComputeFinalCosts(NULL, &rel, NULL);
    return rel;
  }
  return d->final_relative_cost;
}
int orc_decoder_num_toks(orc_decoder *d) { return d->num_toks; }

static thread_local Decoder::RawLat g_lat;  // scratch between _size and _get (both called from one thread)

int orc_decoder_lattice_size(orc_decoder *d, kamd_lattice_size *sz) {
  bool ok = d->GetRawLattice(&g_lat);
  sz->num_states = ok ? static_cast<int32_t>(g_lat.frame.size()) : 0;
  sz->num_arcs = ok ? static_cast<int32_t>(g_lat.arcs.size()) : 0;
  sz->num_frames = d->NumFramesDecoded();
  sz->start = ok ? g_lat.start : -1;
  return ok ? 0 : -1;
}
// the same with GetRawLattice's use_final_probs argument (false is only legal before FinalizeDecoding, :117-120)
int orc_decoder_lattice_size_ufp(orc_decoder *d, int use_final_probs, kamd_lattice_size *sz) {
  if (d->decoding_finalized && !use_final_probs) return -2;
  bool ok = d->GetRawLattice(&g_lat, use_final_probs != 0);
  sz->num_states = ok ? static_cast<int32_t>(g_lat.frame.size()) : 0;
  sz->num_arcs = ok ? static_cast<int32_t>(g_lat.arcs.size()) : 0;
  sz->num_frames = d->NumFramesDecoded();
  sz->start = ok ? g_lat.start : -1;
  return ok ? 0 : -1;
}
int orc_decoder_get_raw_lattice_ufp(orc_decoder *d, int use_final_probs, int32_t *state_frame, int32_t *state_hclg,
                                    float *state_cost, float *state_final, kamd_lat_arc *arcs) {
  Decoder::RawLat lat;
  if (!d->GetRawLattice(&lat, use_final_probs != 0)) return -1;
  size_t n = lat.frame.size();
  memcpy(state_frame, lat.frame.data(), n * 4);
  memcpy(state_hclg, lat.hclg.data(), n * 4);
  memcpy(state_cost, lat.cost.data(), n * 4);
  memcpy(state_final, lat.final.data(), n * 4);
  if (!lat.arcs.empty()) memcpy(arcs, lat.arcs.data(), lat.arcs.size() * sizeof(kamd_lat_arc));
  return 0;
}
int orc_decoder_get_raw_lattice(orc_decoder *d, int32_t *state_frame, int32_t *state_hclg,
                                float *state_cost, float *state_final, kamd_lat_arc *arcs) {
  Decoder::RawLat lat;
  if (!d->GetRawLattice(&lat)) return -1;
  size_t n = lat.frame.size();
  memcpy(state_frame, lat.frame.data(), n * 4);
  memcpy(state_hclg, lat.hclg.data(), n * 4);
  memcpy(state_cost, lat.cost.data(), n * 4);
  memcpy(state_final, lat.final.data(), n * 4);
  if (!lat.arcs.empty()) memcpy(arcs, lat.arcs.data(), lat.arcs.size() * sizeof(kamd_lat_arc));
  return 0;
}

// ShortestPath over a raw lattice given as arrays (canonical numbering).
// GetBestPath (lattice-faster-decoder.cc:102-108) + GetLinearSymbolSequence
// (fstext/fstext-utils-inl.h:178-215).  Forward Viterbi in LatticeWeight
// (Times = componentwise +, Plus = natural-order min, fstext/lattice-weight.h).
static int LatticeBestPath(int num_states, int start, const float *state_final, int num_arcs,
                           const kamd_lat_arc *arcs, int32_t *alignment, int ali_cap,
                           int *ali_len, int32_t *words, int words_cap, int *words_len,
                           float *graph_cost, float *acoustic_cost, std::vector<int> *path_out) {
  *ali_len = 0; *words_len = 0; *graph_cost = kInf; *acoustic_cost = kInf;
  if (num_states == 0) return -1;
  // Kahn topological order
  std::vector<int> indeg(num_states, 0), first(num_states + 1, 0);
  for (int i = 0; i < num_arcs; i++) { indeg[arcs[i].dst]++; first[arcs[i].src + 1]++; }
  for (int s = 0; s < num_states; s++) first[s + 1] += first[s];
  std::vector<int> order;  // arcs are sorted by src already, but do not rely on it
  std::vector<int> by_src(num_arcs), pos(first.begin(), first.end() - 1);
  for (int i = 0; i < num_arcs; i++) by_src[pos[arcs[i].src]++] = i;
  std::vector<int> stack;
  for (int s = num_states - 1; s >= 0; s--) if (indeg[s] == 0) stack.push_back(s);
  std::vector<float> d1(num_states, kInf), d2(num_states, kInf);
  std::vector<int> back(num_states, -1);
  if (start < 0 || start >= num_states) return -1;
  d1[start] = 0.0f; d2[start] = 0.0f;  // the start token's state (:155-157)
  size_t visited = 0;
  while (!stack.empty()) {
    int s = stack.back(); stack.pop_back(); visited++;
    for (int k = first[s]; k < first[s + 1]; k++) {
      const kamd_lat_arc &a = arcs[by_src[k]];
      if (d1[s] != kInf) {
        float n1 = d1[s] + a.graph_cost, n2 = d2[s] + a.acoustic_cost;
        if (d1[a.dst] == kInf || LatBetter(n1, n2, d1[a.dst], d2[a.dst])) {
          d1[a.dst] = n1; d2[a.dst] = n2; back[a.dst] = by_src[k];
        }
      }
      if (--indeg[a.dst] == 0) stack.push_back(a.dst);
    }
  }
  if (visited != static_cast<size_t>(num_states)) return -2;  // cycle
  int best = -1; float b1 = kInf, b2 = kInf;
  for (int s = 0; s < num_states; s++) {
    if (state_final[s] == kInf || d1[s] == kInf) continue;
    float t1 = d1[s] + state_final[s], t2 = d2[s];
    if (best == -1 || LatBetter(t1, t2, b1, b2)) { best = s; b1 = t1; b2 = t2; }
  }
  if (best == -1) return -1;
  std::vector<int> path;
  for (int s = best; back[s] != -1; s = arcs[back[s]].src) path.push_back(back[s]);
  std::reverse(path.begin(), path.end());
  for (size_t i = 0; i < path.size(); i++) {
    const kamd_lat_arc &a = arcs[path[i]];
    if (a.ilabel != 0) { if (*ali_len < ali_cap) alignment[*ali_len] = a.ilabel; (*ali_len)++; }
    if (a.olabel != 0) { if (*words_len < words_cap) words[*words_len] = a.olabel; (*words_len)++; }
  }
  *graph_cost = b1; *acoustic_cost = b2;
  if (path_out) *path_out = path;
  return 0;
}

int orc_lattice_best_path(int num_states, int start, const float *state_final, int num_arcs,
                          const kamd_lat_arc *arcs, int32_t *alignment, int ali_cap,
                          int *ali_len, int32_t *words, int words_cap, int *words_len,
                          float *graph_cost, float *acoustic_cost) {
  return LatticeBestPath(num_states, start, state_final, num_arcs, arcs, alignment, ali_cap, ali_len, words, words_cap, words_len,
                         graph_cost, acoustic_cost, NULL);
}

// the arcs of that path, start -> end (indices into arcs): what TraceBackBestPath visits, in the other direction
int orc_lattice_best_path_arcs(int num_states, int start, const float *state_final, int num_arcs, const kamd_lat_arc *arcs,
                               int32_t *path, int cap, int *n) {
  std::vector<int32_t> ali(num_arcs + 1), words(num_arcs + 1);
  int na = 0, nw = 0; float g = 0, a = 0;
  std::vector<int> p;
  const int rc = LatticeBestPath(num_states, start, state_final, num_arcs, arcs, ali.data(), num_arcs + 1, &na, words.data(), num_arcs + 1, &nw,
                                 &g, &a, &p);
  *n = static_cast<int>(p.size());
  if (rc != 0) return rc;
  if (*n > cap) return -3;
  for (size_t i = 0; i < p.size(); i++) path[i] = p[i];
  return 0;
}

int orc_decoder_get_trace(orc_decoder *d, int32_t *ntok, float *cutoff, float *cost_offset,
                          int cap) {
  int n = std::min<int>(cap, static_cast<int>(d->trace_ntok.size()));
  for (int i = 0; i < n; i++) {
    ntok[i] = d->trace_ntok[i]; cutoff[i] = d->trace_cutoff[i];
    cost_offset[i] = d->trace_offset[i];
  }
  return n;
}
void orc_decoder_get_counters(orc_decoder *d, int64_t c[8]) { memcpy(c, d->counters, 64); }

// ---- endpointing (online2/online-endpoint.{h,cc}), BaseFloat arithmetic
// rules[5][4] = {must_contain_nonsilence, min_trailing_silence, max_relative_cost, min_utterance_length}
int orc_endpoint_detected(const float *rules, int num_frames_decoded, int trailing_silence_frames,
                          float frame_shift_in_seconds, float final_relative_cost) {
  float utterance_length = num_frames_decoded * frame_shift_in_seconds,        // online-endpoint.cc:53-54
        trailing_silence = trailing_silence_frames * frame_shift_in_seconds;
  for (int r = 0; r < 5; r++) {                                                 // RuleActivated, :25-44
    const float *R = rules + 4 * r;
    bool contains_nonsilence = (utterance_length > trailing_silence);
    bool ans = (contains_nonsilence || R[0] == 0.0f) && trailing_silence >= R[1] &&
               final_relative_cost <= R[2] && utterance_length >= R[3];
    if (ans) return 1;
  }
  return 0;
}
// TrailingSilenceLength (:71-102) given the best path's transition-ids in time order (the iterator walks them
// backwards, skipping epsilon arcs): count silence phones from the end, stop at the first other phone.
int orc_trailing_silence_length(const int32_t *alignment, int n, const int32_t *tid2phone,
                                const int32_t *silence_phones, int n_sil) {
  int num_silence_frames = 0;
  for (int i = n - 1; i >= 0; i--) {
    int phone = tid2phone[alignment[i]];
    bool is_sil = false;
    for (int k = 0; k < n_sil; k++) is_sil = is_sil || silence_phones[k] == phone;
    if (!is_sil) break;
    num_silence_frames++;
  }
  return num_silence_frames;
}

}  // extern "C"
