// Lattice determinization on the host: what
// DecodeUtteranceLatticeFaster does with the raw lattice after the decoder returns
// (decoder/decoder-wrappers.cc:262-285): Invert, TopSort, ArcSort, then
// DeterminizeLatticePhonePruned = [insert phone labels, pruned determinization to a Lattice,
// delete phone labels] + pruned determinization to a CompactLattice, + Connect
// (lat/determinize-lattice-pruned.cc:1484-1509, :1412-1470, :1389-1409, :1292-1345).
//
// The core is a restatement of LatticeDeterminizerPruned (same file, :43-1189): subset
// construction over (state, output string, LatticeWeight) triples with a "take the better
// path" Plus, epsilon closure, removal of the common string prefix and common weight of
// each subset, and a best-first agenda of (output state, label) tasks ordered by the best
// complete-path cost through them, cut off at best path + beam.  Data structures are this
// file's own: strings live in a label trie addressed by integer ids (id equality = sequence
// equality, the property the reference's LatticeStringRepository provides), subsets are
// hashed on their (state, string) sequence and compared with the reference's delta rule.
//
// PARITY UNPINNED: OpenFst is not available here, lat/determinize-lattice-pruned-test.cc is
// a randomized equivalence test that needs it, and no reference test holds a lattice
// fixture.  tests/test_determinize.py checks the defining properties instead: the output is
// deterministic on word labels, every word sequence of the pruned input has exactly one
// path whose weight is the best input path's weight and whose transition-id string is that
// path's alignment, and nothing outside the beam is required.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <queue>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace {

const float kInfF = std::numeric_limits<float>::infinity();
const double kInfD = std::numeric_limits<double>::infinity();

// ---- LatticeWeight (fstext/lattice-weight.h:50-395)
struct W { float g, a; };
inline W One() { W w = {0.0f, 0.0f}; return w; }
inline W Zero() { W w = {kInfF, kInfF}; return w; }
inline bool IsZero(const W &w) { return w.g == kInfF && w.a == kInfF; }
inline bool Same(const W &x, const W &y) { return x.g == y.g && x.a == y.a; }
inline int Cmp(const W &x, const W &y) {            // :294-307: +1 = x is better
  const float f1 = x.g + x.a, f2 = y.g + y.a;
  if (f1 < f2) return 1;
  if (f1 > f2) return -1;
  if (x.g < y.g) return 1;
  if (x.g > y.g) return -1;
  return 0;
}
inline W Times(const W &x, const W &y) { W w = {x.g + y.g, x.a + y.a}; return w; }
inline W Divide(const W &x, const W &y) {           // :370-385
  const float a = x.g - y.g, b = x.a - y.a;
  if (a != a || b != b || a == -kInfF || b == -kInfF) return Zero();
  if (a == kInfF || b == kInfF) return Zero();
  W w = {a, b};
  return w;
}
inline double Cost(const W &w) { return static_cast<double>(w.g) + static_cast<double>(w.a); }   // :847
inline bool Approx(const W &x, const W &y, float delta) {                                        // :389-394
  if (x.g == y.g && x.a == y.a) return true;
  return std::fabs((x.g + x.a) - (y.g + y.a)) <= delta;
}

// ---- label sequences as nodes of a trie: id 0 = empty, node = (parent, label)
class Strings {
 public:
  Strings() { parent_.push_back(-1); label_.push_back(0); len_.push_back(0); }
  int Successor(int s, int label) {
    const unsigned long long key = (static_cast<unsigned long long>(static_cast<unsigned>(s)) << 32) | static_cast<unsigned>(label);
    std::unordered_map<unsigned long long, int>::iterator it = child_.find(key);
    if (it != child_.end()) return it->second;
    const int id = static_cast<int>(parent_.size());
    parent_.push_back(s); label_.push_back(label); len_.push_back(len_[s] + 1);
    child_[key] = id;
    return id;
  }
  int Size(int s) const { return len_[s]; }
  void ToVector(int s, std::vector<int> *out) const {
    out->resize(len_[s]);
    for (int i = len_[s] - 1; i >= 0; i--) { (*out)[i] = label_[s]; s = parent_[s]; }
  }
  int FromVector(const std::vector<int> &v) { int s = 0; for (size_t i = 0; i < v.size(); i++) s = Successor(s, v[i]); return s; }
  void ReduceToCommonPrefix(int s, std::vector<int> *prefix) const {
    std::vector<int> v;
    ToVector(s, &v);
    size_t n = 0;
    while (n < v.size() && n < prefix->size() && v[n] == (*prefix)[n]) n++;
    prefix->resize(n);
  }
  int RemovePrefix(int s, size_t n) {
    if (n == 0) return s;
    std::vector<int> v;
    ToVector(s, &v);
    int r = 0;
    for (size_t i = n; i < v.size(); i++) r = Successor(r, v[i]);
    return r;
  }
  int Concatenate(int a, int b) {
    if (b == 0) return a;
    std::vector<int> v;
    ToVector(b, &v);
    for (size_t i = 0; i < v.size(); i++) a = Successor(a, v[i]);
    return a;
  }
  size_t MemBytes() const { return parent_.size() * 40; }
 private:
  std::vector<int> parent_, label_, len_;
  std::unordered_map<unsigned long long, int> child_;
};

struct Arc { int ilabel, olabel; W w; int next; };
struct Fst {                         // topologically sorted input, arcs sorted by ilabel
  std::vector<int> off;              // [S+1]
  std::vector<Arc> arcs;
  std::vector<W> fin;
  int start;
  int NumStates() const { return static_cast<int>(fin.size()); }
};

struct Element { int state; int str; W w; };
inline bool ElemDiffers(const Element &x, const Element &y) { return x.state != y.state || x.str != y.str || !Same(x.w, y.w); }

struct TempArc { int ilabel; int str; int next; W w; };   // next = -1: final weight
struct OutState { std::vector<Element> subset; std::vector<TempArc> arcs; double fwd; };
struct Task { int state, label; std::vector<Element> subset; double priority; };
struct TaskWorse { bool operator()(const Task *a, const Task *b) const { return a->priority > b->priority; } };

struct SubsetHash {
  size_t operator()(const std::vector<Element> *s) const {
    size_t h = 0, f = 1;
    for (size_t i = 0; i < s->size(); i++) { h *= f; h += static_cast<size_t>((*s)[i].state) + 104729u * static_cast<size_t>((*s)[i].str); f *= 23531; }
    return h;
  }
};
struct SubsetEq {
  float delta;
  explicit SubsetEq(float d) : delta(d) {}
  bool operator()(const std::vector<Element> *a, const std::vector<Element> *b) const {
    if (a->size() != b->size()) return false;
    for (size_t i = 0; i < a->size(); i++)
      if ((*a)[i].state != (*b)[i].state || (*a)[i].str != (*b)[i].str || !Approx((*a)[i].w, (*b)[i].w, delta)) return false;
    return true;
  }
};

struct Opts { float delta; int max_mem, max_states, max_arcs, max_loop; float retry_cutoff; };

class Determinizer {
 public:
  Determinizer(const Fst &f, double beam, const Opts &o)
      : ifst_(f), beam_(beam), opts_(o), num_arcs_(0), num_elems_(0), cutoff_(kInfD),
        minimal_(16, SubsetHash(), SubsetEq(o.delta)), initial_(16, SubsetHash(), SubsetEq(o.delta)) {}
  ~Determinizer() {
    for (size_t i = 0; i < states_.size(); i++) delete states_[i];
    for (InitialMap::iterator it = initial_.begin(); it != initial_.end(); ++it) delete it->first;
    while (!queue_.empty()) { delete queue_.top(); queue_.pop(); }
  }

  bool Determinize(double *effective_beam) {          // :329-376
    Initialize();
    bool complete = true;
    while (!queue_.empty()) {
      const size_t ns = states_.size();
      if ((opts_.max_states > 0 && static_cast<int>(ns) > opts_.max_states) ||
          (opts_.max_arcs > 0 && num_arcs_ > opts_.max_arcs) || (ns % 10 == 0 && !MemoryOk())) { complete = false; break; }
      Task *t = queue_.top();
      queue_.pop();
      ProcessTransition(t->state, t->label, &t->subset);
      delete t;
    }
    if (effective_beam) *effective_beam = queue_.empty() ? beam_ : queue_.top()->priority - back_[ifst_.start];
    return complete && queue_.empty();
  }

  // compact (acceptor) form: arcs carry a label, a weight and a transition-id string (:56-103)
  void OutputCompact(std::vector<std::vector<TempArc> > *arcs, Strings **strings) {
    arcs->resize(states_.size());
    for (size_t s = 0; s < states_.size(); s++) (*arcs)[s] = states_[s]->arcs;
    *strings = &repo_;
  }
  // state-level form with extra states spelling the strings out on olabels (:107-176)
  void OutputLattice(Fst *o) {
    const int n = static_cast<int>(states_.size());
    std::vector<std::vector<Arc> > out(n);
    std::vector<W> fin(n, Zero());
    if (n == 0) { o->start = -1; o->off.assign(1, 0); o->arcs.clear(); o->fin.clear(); return; }
    for (int s = 0; s < n; s++) {
      const std::vector<TempArc> &v = states_[s]->arcs;
      for (size_t k = 0; k < v.size(); k++) {
        std::vector<int> seq;
        repo_.ToVector(v[k].str, &seq);
        int cur = s;
        if (v[k].next < 0) {
          for (size_t i = 0; i < seq.size(); i++) {
            const int nx = static_cast<int>(out.size());
            out.push_back(std::vector<Arc>()); fin.push_back(Zero());
            Arc a; a.next = nx; a.w = i == 0 ? v[k].w : One(); a.ilabel = 0; a.olabel = seq[i];
            out[cur].push_back(a);
            cur = nx;
          }
          fin[cur] = seq.empty() ? v[k].w : One();
        } else {
          for (size_t i = 0; i + 1 < seq.size(); i++) {
            const int nx = static_cast<int>(out.size());
            out.push_back(std::vector<Arc>()); fin.push_back(Zero());
            Arc a; a.next = nx; a.w = i == 0 ? v[k].w : One(); a.ilabel = i == 0 ? v[k].ilabel : 0; a.olabel = seq[i];
            out[cur].push_back(a);
            cur = nx;
          }
          Arc a; a.next = v[k].next; a.w = seq.size() <= 1 ? v[k].w : One();
          a.ilabel = seq.size() <= 1 ? v[k].ilabel : 0; a.olabel = seq.empty() ? 0 : seq.back();
          out[cur].push_back(a);
        }
      }
    }
    o->start = 0; o->fin = fin; o->off.assign(out.size() + 1, 0); o->arcs.clear();
    for (size_t s = 0; s < out.size(); s++) { o->off[s] = static_cast<int>(o->arcs.size()); o->arcs.insert(o->arcs.end(), out[s].begin(), out[s].end()); }
    o->off[out.size()] = static_cast<int>(o->arcs.size());
  }

 private:
  typedef std::unordered_map<const std::vector<Element> *, int, SubsetHash, SubsetEq> MinimalMap;
  typedef std::unordered_map<const std::vector<Element> *, Element, SubsetHash, SubsetEq> InitialMap;

  bool MemoryOk() {                                   // :288-327 (no repository compaction here)
    if (opts_.max_mem <= 0) return true;
    const double total = static_cast<double>(repo_.MemBytes()) + 24.0 * num_arcs_ + 16.0 * num_elems_;
    return total <= static_cast<double>(opts_.max_mem);
  }

  int StrCompare(const W &aw, int as, const W &bw, int bs) const {   // :613-640
    const int c = Cmp(aw, bw);
    if (c != 0) return c;
    if (as == bs) return 0;
    std::vector<int> av, bv;
    repo_.ToVector(as, &av); repo_.ToVector(bs, &bv);
    if (av.size() > bv.size()) return -1;
    if (av.size() < bv.size()) return 1;
    for (size_t i = 0; i < av.size(); i++) { if (av[i] < bv[i]) return -1; if (av[i] > bv[i]) return 1; }
    return 0;
  }

  bool EmitsOrFinal(int s) {                          // :986-1010
    if (static_cast<int>(eof_.size()) <= s) eof_.resize(s + 1, 0);
    if (eof_[s]) return eof_[s] == 2;
    eof_[s] = 1;
    if (!IsZero(ifst_.fin[s])) eof_[s] = 2;
    for (int k = ifst_.off[s]; k < ifst_.off[s + 1]; k++)
      if (ifst_.arcs[k].ilabel != 0 && !IsZero(ifst_.arcs[k].w)) { eof_[s] = 2; break; }
    return eof_[s] == 2;
  }

  void EpsilonClosure(std::vector<Element> *subset) {  // :633-726
    struct ByState { bool operator()(const Element &x, const Element &y) const { return x.state > y.state; } };
    std::priority_queue<Element, std::vector<Element>, ByState> q;
    std::unordered_map<int, Element> cur;
    for (size_t i = 0; i < subset->size(); i++) { q.push((*subset)[i]); cur[(*subset)[i].state] = (*subset)[i]; }
    bool replaced = false;
    int counter = 0;
    while (!q.empty()) {
      const Element e = q.top();
      q.pop();
      if (replaced && ElemDiffers(cur[e.state], e)) continue;
      if (opts_.max_loop > 0 && counter++ > opts_.max_loop) { loop_error_ = true; break; }
      for (int k = ifst_.off[e.state]; k < ifst_.off[e.state + 1]; k++) {
        const Arc &arc = ifst_.arcs[k];
        if (arc.ilabel != 0) break;                    // arcs are sorted on ilabel
        if (IsZero(arc.w)) continue;
        Element nx;
        nx.state = arc.next; nx.w = Times(e.w, arc.w); nx.str = 0;
        std::unordered_map<int, Element>::iterator it = cur.find(nx.state);
        if (it == cur.end()) {
          nx.str = arc.olabel == 0 ? e.str : repo_.Successor(e.str, arc.olabel);
          cur[nx.state] = nx;
          q.push(nx);
        } else {
          int c = Cmp(nx.w, it->second.w);
          if (c == 0) {
            nx.str = arc.olabel == 0 ? e.str : repo_.Successor(e.str, arc.olabel);
            c = StrCompare(nx.w, nx.str, it->second.w, it->second.str);
          }
          if (c == 1) {
            nx.str = arc.olabel == 0 ? e.str : repo_.Successor(e.str, arc.olabel);
            it->second.str = nx.str; it->second.w = nx.w;
            q.push(nx);
            replaced = true;
          }
        }
      }
    }
    subset->clear();
    for (std::unordered_map<int, Element>::iterator it = cur.begin(); it != cur.end(); ++it) subset->push_back(it->second);
    std::sort(subset->begin(), subset->end(), [](const Element &x, const Element &y) { return x.state < y.state; });
  }

  void ToMinimal(std::vector<Element> *subset) {       // :501-513
    size_t o = 0;
    for (size_t i = 0; i < subset->size(); i++) if (EmitsOrFinal((*subset)[i].state)) (*subset)[o++] = (*subset)[i];
    subset->resize(o);
  }

  void Normalize(std::vector<Element> *e, W *tot, int *common) {   // :775-803
    if (e->empty()) { *common = 0; *tot = Zero(); return; }
    std::vector<int> prefix;
    repo_.ToVector((*e)[0].str, &prefix);
    W w = (*e)[0].w;
    for (size_t i = 1; i < e->size(); i++) {
      w = Cmp(w, (*e)[i].w) >= 0 ? w : (*e)[i].w;      // Plus
      repo_.ReduceToCommonPrefix((*e)[i].str, &prefix);
    }
    for (size_t i = 0; i < e->size(); i++) {
      (*e)[i].w = Divide((*e)[i].w, w);
      (*e)[i].str = repo_.RemovePrefix((*e)[i].str, prefix.size());
    }
    *common = repo_.FromVector(prefix);
    *tot = w;
  }

  void MakeUnique(std::vector<Element> *s) {           // :808-838 (input sorted on state)
    size_t o = 0, i = 0;
    while (i < s->size()) {
      (*s)[o] = (*s)[i++];
      while (i < s->size() && (*s)[i].state == (*s)[o].state) {
        if (StrCompare((*s)[i].w, (*s)[i].str, (*s)[o].w, (*s)[o].str) == 1) { (*s)[o].str = (*s)[i].str; (*s)[o].w = (*s)[i].w; }
        i++;
      }
      o++;
    }
    s->resize(o);
  }

  void ProcessFinal(int sid) {                         // :732-770
    OutState &st = *states_[sid];
    int fstr = 0; W fw = Zero(); bool is_final = false;
    for (size_t i = 0; i < st.subset.size(); i++) {
      const Element &e = st.subset[i];
      const W w = Times(e.w, ifst_.fin[e.state]);
      if (!IsZero(w) && (!is_final || StrCompare(w, e.str, fw, fstr) == 1)) { is_final = true; fw = w; fstr = e.str; }
    }
    if (is_final && Cost(fw) + st.fwd <= cutoff_) {
      TempArc t; t.ilabel = 0; t.next = -1; t.str = fstr; t.w = fw;
      st.arcs.push_back(t);
      num_arcs_++;
    }
  }

  void ProcessTransitions(int sid) {                   // :901-983
    std::vector<std::pair<int, Element> > all;
    const std::vector<Element> &sub = states_[sid]->subset;
    for (size_t i = 0; i < sub.size(); i++) {
      const Element &e = sub[i];
      for (int k = ifst_.off[e.state]; k < ifst_.off[e.state + 1]; k++) {
        const Arc &arc = ifst_.arcs[k];
        if (arc.ilabel == 0 || IsZero(arc.w)) continue;
        Element nx; nx.state = arc.next; nx.w = Times(e.w, arc.w);
        nx.str = arc.olabel == 0 ? e.str : repo_.Successor(e.str, arc.olabel);
        all.push_back(std::make_pair(arc.ilabel, nx));
      }
    }
    std::sort(all.begin(), all.end(), [](const std::pair<int, Element> &x, const std::pair<int, Element> &y) {
      if (x.first != y.first) return x.first < y.first;
      return x.second.state < y.second.state;
    });
    size_t c = 0;
    while (c < all.size()) {
      Task *t = new Task;
      t->state = sid; t->label = all[c].first; t->priority = kInfD;
      while (c < all.size() && all[c].first == t->label) {
        t->subset.push_back(all[c].second);
        t->priority = std::min(t->priority, Cost(all[c].second.w) + back_[all[c].second.state]);
        c++;
      }
      t->priority += states_[sid]->fwd;
      if (t->priority > cutoff_) { delete t; continue; }
      MakeUnique(&t->subset);
      queue_.push(t);
    }
  }

  int MinimalToState(const std::vector<Element> &subset, double fwd) {   // :520-547
    MinimalMap::const_iterator it = minimal_.find(&subset);
    if (it != minimal_.end()) return it->second;
    const int id = static_cast<int>(states_.size());
    OutState *ns = new OutState;
    ns->subset = subset; ns->fwd = fwd;
    states_.push_back(ns);
    minimal_[&ns->subset] = id;
    num_elems_ += static_cast<int>(subset.size());
    ProcessFinal(id);
    ProcessTransitions(id);
    return id;
  }

  int InitialToState(const std::vector<Element> &in, double fwd, W *rem, int *prefix) {   // :552-598
    InitialMap::const_iterator it = initial_.find(&in);
    if (it != initial_.end()) { *rem = it->second.w; *prefix = it->second.str; return it->second.state; }
    std::vector<Element> subset(in);
    EpsilonClosure(&subset);
    ToMinimal(&subset);
    Element e;
    Normalize(&subset, &e.w, &e.str);
    fwd += Cost(e.w);
    const int ans = MinimalToState(subset, fwd);
    *rem = e.w; *prefix = e.str;
    e.state = ans;
    std::vector<Element> *key = new std::vector<Element>(in);
    initial_[key] = e;
    num_elems_ += static_cast<int>(key->size());
    return ans;
  }

  void ProcessTransition(int sid, int label, std::vector<Element> *subset) {   // :845-880
    double fwd = states_[sid]->fwd;
    int common; W tot;
    Normalize(subset, &tot, &common);
    fwd += Cost(tot);
    W ntot; int ncommon;
    const int next = InitialToState(*subset, fwd, &ntot, &ncommon);
    TempArc t;
    t.ilabel = label; t.next = next; t.str = repo_.Concatenate(common, ncommon); t.w = Times(tot, ntot);
    states_[sid]->arcs.push_back(t);
    num_arcs_++;
  }

  void Initialize() {                                  // :1012-1099
    const int S = ifst_.NumStates();
    back_.assign(S, kInfD);
    for (int s = S - 1; s >= 0; s--) {
      double c = Cost(ifst_.fin[s]);
      for (int k = ifst_.off[s]; k < ifst_.off[s + 1]; k++) c = std::min(c, Cost(ifst_.arcs[k].w) + back_[ifst_.arcs[k].next]);
      back_[s] = c;
    }
    if (ifst_.start < 0) return;
    cutoff_ = back_[ifst_.start] + beam_;
    std::vector<Element> subset(1);
    subset[0].state = ifst_.start; subset[0].w = One(); subset[0].str = 0;
    EpsilonClosure(&subset);
    ToMinimal(&subset);
    OutState *st = new OutState;
    st->subset = subset; st->fwd = 0.0;
    states_.push_back(st);
    num_elems_ += static_cast<int>(subset.size());
    minimal_[&st->subset] = 0;
    ProcessFinal(0);
    ProcessTransitions(0);
  }

  const Fst &ifst_;
  double beam_;
  Opts opts_;
  int num_arcs_, num_elems_;
  double cutoff_;
  std::vector<double> back_;
  std::vector<OutState *> states_;
  std::vector<char> eof_;
  Strings repo_;
  MinimalMap minimal_;
  InitialMap initial_;
  std::priority_queue<Task *, std::vector<Task *>, TaskWorse> queue_;
 public:
  bool loop_error_ = false;
};

// ---- helpers around the core
// stable topological order (Kahn, smallest state id first); false on a cycle
bool TopSort(Fst *f) {
  const int S = f->NumStates();
  std::vector<int> indeg(S, 0), order, pos(S, -1);
  for (size_t k = 0; k < f->arcs.size(); k++) indeg[f->arcs[k].next]++;
  std::priority_queue<int, std::vector<int>, std::greater<int> > ready;
  for (int s = 0; s < S; s++) if (indeg[s] == 0) ready.push(s);
  while (!ready.empty()) {
    const int s = ready.top(); ready.pop();
    pos[s] = static_cast<int>(order.size()); order.push_back(s);
    for (int k = f->off[s]; k < f->off[s + 1]; k++) if (--indeg[f->arcs[k].next] == 0) ready.push(f->arcs[k].next);
  }
  if (static_cast<int>(order.size()) != S) return false;
  Fst o;
  o.start = f->start >= 0 ? pos[f->start] : -1;
  o.fin.resize(S); o.off.assign(S + 1, 0);
  for (int i = 0; i < S; i++) {
    const int s = order[i];
    o.fin[i] = f->fin[s];
    o.off[i] = static_cast<int>(o.arcs.size());
    for (int k = f->off[s]; k < f->off[s + 1]; k++) { Arc a = f->arcs[k]; a.next = pos[a.next]; o.arcs.push_back(a); }
  }
  o.off[S] = static_cast<int>(o.arcs.size());
  *f = o;
  return true;
}
void ArcSortIlabel(Fst *f) {
  for (int s = 0; s < f->NumStates(); s++)
    std::stable_sort(f->arcs.begin() + f->off[s], f->arcs.begin() + f->off[s + 1], [](const Arc &x, const Arc &y) { return x.ilabel < y.ilabel; });
}
// lat/lattice-functions.cc PruneLattice: keep what lies on a path within beam of the best
void PruneFst(double beam, Fst *f) {
  const int S = f->NumStates();
  if (S == 0 || f->start < 0) return;
  std::vector<double> fw(S, kInfD), bw(S, kInfD);
  fw[f->start] = 0.0;
  for (int s = 0; s < S; s++)
    for (int k = f->off[s]; k < f->off[s + 1]; k++) fw[f->arcs[k].next] = std::min(fw[f->arcs[k].next], fw[s] + Cost(f->arcs[k].w));
  for (int s = S - 1; s >= 0; s--) {
    bw[s] = Cost(f->fin[s]);
    for (int k = f->off[s]; k < f->off[s + 1]; k++) bw[s] = std::min(bw[s], Cost(f->arcs[k].w) + bw[f->arcs[k].next]);
  }
  const double cutoff = bw[f->start] + beam;
  std::vector<int> map(S, -1);
  int n = 0;
  for (int s = 0; s < S; s++) if (fw[s] + bw[s] <= cutoff) map[s] = n++;
  Fst o;
  o.start = map[f->start]; o.fin.resize(n); o.off.assign(n + 1, 0);
  for (int s = 0; s < S; s++) {
    if (map[s] < 0) continue;
    o.fin[map[s]] = (fw[s] + Cost(f->fin[s]) <= cutoff) ? f->fin[s] : Zero();
    o.off[map[s]] = static_cast<int>(o.arcs.size());
    for (int k = f->off[s]; k < f->off[s + 1]; k++) {
      const Arc &a = f->arcs[k];
      if (map[a.next] >= 0 && fw[s] + Cost(a.w) + bw[a.next] <= cutoff) { Arc b = a; b.next = map[a.next]; o.arcs.push_back(b); }
    }
  }
  o.off[n] = static_cast<int>(o.arcs.size());
  *f = o;
}

struct CompactArcOut { int src, dst, label; W w; int str; };

// DeterminizeLatticePruned with the beam-retry loop (:1191-1289); out = compact arcs (or, when
// lattice_out != NULL, the state-level form)
bool DeterminizePruned(const Fst &in, double beam, const Opts &o, std::vector<std::vector<TempArc> > *carcs,
                       Strings *strings_out, Fst *lattice_out, bool *loop_error) {
  Fst tmp;
  for (int iter = 0; iter < 10; iter++) {
    Determinizer det(iter == 0 ? in : tmp, beam, o);
    double eff;
    const bool ans = det.Determinize(&eff);
    if (det.loop_error_) { *loop_error = true; return false; }
    if (eff >= beam * o.retry_cutoff || beam == kInfD || iter + 1 == 10) {
      if (lattice_out) det.OutputLattice(lattice_out);
      else {
        Strings *s;
        det.OutputCompact(carcs, &s);
        *strings_out = *s;
      }
      return ans;
    }
    if (eff < 0.0) eff = 0.0;
    double nb = beam * std::sqrt(eff / beam);
    if (nb < 0.5 * beam) nb = 0.5 * beam;
    beam = nb;
    if (iter == 0) tmp = in;
    PruneFst(beam, &tmp);
  }
  return false;
}

struct CompactLattice {
  int start;
  std::vector<int> off;                 // [S+1] into arcs
  std::vector<kamd_clat_arc> arcs;      // str_begin/str_len index 'strings'
  std::vector<float> fin;               // [2S]
  std::vector<int32_t> fin_str_begin, fin_str_len;
  std::vector<int32_t> strings;
  bool complete;
};

}  // namespace

extern "C" {

void kamd_determinize_opts_default(kamd_determinize_opts *o) {   // lat/determinize-lattice-pruned.h:214-245, 126-141
  o->delta = 0.0009765625f; o->max_mem = 50000000; o->phone_determinize = 1; o->word_determinize = 1;
  o->max_loop = 0; o->retry_cutoff = 0.5f;
}

kamd_compact_lattice *kamd_lattice_determinize_phone_pruned(int32_t num_states, int32_t start, const float *state_final,
                                                            const kamd_lat_arc *arcs, int32_t num_arcs,
                                                            const int32_t *tid_phone, int32_t num_tids, double beam,
                                                            const kamd_determinize_opts *opts) {
  kamd_determinize_opts od;
  if (opts) od = *opts; else kamd_determinize_opts_default(&od);
  if (num_states <= 0 || start < 0 || start >= num_states) { kamd::SetError(KAMD_ERR_ARG, "empty lattice"); return NULL; }
  if (!(beam > 0.0)) { kamd::SetError(KAMD_ERR_ARG, "beam must be positive"); return NULL; }
  if (od.phone_determinize && !tid_phone) { kamd::SetError(KAMD_ERR_ARG, "phone_determinize needs the transition-id -> phone table"); return NULL; }
  Opts o; o.delta = od.delta; o.max_mem = od.max_mem; o.max_states = -1; o.max_arcs = -1; o.max_loop = od.max_loop; o.retry_cutoff = od.retry_cutoff;
  // ---- Invert (words on the input side), TopSort, ArcSort (:1490-1503)
  Fst f;
  f.start = start; f.fin.resize(num_states); f.off.assign(num_states + 1, 0);
  for (int32_t i = 0; i < num_arcs; i++) {
    if (arcs[i].src < 0 || arcs[i].src >= num_states || arcs[i].dst < 0 || arcs[i].dst >= num_states || (i > 0 && arcs[i].src < arcs[i - 1].src)) {
      kamd::SetError(KAMD_ERR_ARG, "lattice arcs must be in range and sorted by source state"); return NULL;
    }
    f.off[arcs[i].src + 1]++;
  }
  for (int s = 0; s < num_states; s++) {
    f.off[s + 1] += f.off[s];
    W w = {state_final[2 * s], state_final[2 * s + 1]};
    f.fin[s] = state_final[2 * s] == kInfF ? Zero() : w;
  }
  f.arcs.resize(num_arcs);
  for (int32_t i = 0; i < num_arcs; i++) {
    Arc a; a.ilabel = arcs[i].olabel; a.olabel = arcs[i].ilabel; a.w.g = arcs[i].graph_cost; a.w.a = arcs[i].acoustic_cost; a.next = arcs[i].dst;
    f.arcs[i] = a;
  }
  if (!TopSort(&f)) { kamd::SetError(KAMD_ERR_ARG, "topological sorting of the state-level lattice failed (epsilon cycle)"); return NULL; }
  ArcSortIlabel(&f);
  bool ans = true, loop_error = false;
  if (od.phone_determinize) {
    // ---- DeterminizeLatticeInsertPhones (:1292-1345)
    int highest = 0;
    for (size_t k = 0; k < f.arcs.size(); k++) highest = std::max(highest, f.arcs[k].ilabel);
    const int first_phone = highest + 1;
    std::vector<std::vector<Arc> > st(f.NumStates());
    for (int s = 0; s < f.NumStates(); s++) st[s].assign(f.arcs.begin() + f.off[s], f.arcs.begin() + f.off[s + 1]);
    std::vector<W> fin = f.fin;
    const int S0 = f.NumStates();
    for (int s = 0; s < S0; s++) {
      if (s == f.start) continue;
      for (size_t k = 0; k < st[s].size(); k++) {
        Arc &a = st[s][k];
        const int tid = a.olabel;
        if (tid != 0 && tid <= num_tids && tid_phone[tid] > 0) {
          const int phone = tid_phone[tid];
          if (a.ilabel == 0) a.ilabel = first_phone + phone;
          else {
            const int extra = static_cast<int>(st.size());
            Arc b; b.ilabel = first_phone + phone; b.olabel = 0; b.w = One(); b.next = a.next;
            a.next = extra;
            st.push_back(std::vector<Arc>(1, b)); fin.push_back(Zero());
          }
        }
      }
    }
    Fst g;
    g.start = f.start; g.fin = fin; g.off.assign(st.size() + 1, 0);
    for (size_t s = 0; s < st.size(); s++) { g.off[s] = static_cast<int>(g.arcs.size()); g.arcs.insert(g.arcs.end(), st[s].begin(), st[s].end()); }
    g.off[st.size()] = static_cast<int>(g.arcs.size());
    if (!TopSort(&g)) { kamd::SetError(KAMD_ERR_ARG, "TopSort failed after phone insertion"); return NULL; }
    ArcSortIlabel(&g);
    Fst first;
    ans = DeterminizePruned(g, beam, o, NULL, NULL, &first, &loop_error) && ans;
    if (loop_error) { kamd::SetError(KAMD_ERR_STATE, "lattice determinization aborted: epsilon-closure loop limit"); return NULL; }
    for (size_t k = 0; k < first.arcs.size(); k++) if (first.arcs[k].ilabel >= first_phone) first.arcs[k].ilabel = 0;   // DeletePhones
    if (!TopSort(&first)) { kamd::SetError(KAMD_ERR_ARG, "TopSort failed after the phone pass"); return NULL; }
    ArcSortIlabel(&first);
    f = first;
  }
  CompactLattice *cl = new CompactLattice;
  std::vector<std::vector<TempArc> > carcs;
  Strings strings;
  if (od.word_determinize) {
    ans = DeterminizePruned(f, beam, o, &carcs, &strings, NULL, &loop_error) && ans;
    if (loop_error) { delete cl; kamd::SetError(KAMD_ERR_STATE, "lattice determinization aborted: epsilon-closure loop limit"); return NULL; }
  } else {
    // ConvertLattice(ifst, ofst, false): one compact arc per arc, the tid (if any) as a
    // length-one string (lat/kaldi-lattice / fstext/lattice-utils-inl.h)
    carcs.resize(f.NumStates());
    for (int s = 0; s < f.NumStates(); s++) {
      for (int k = f.off[s]; k < f.off[s + 1]; k++) {
        TempArc t; t.ilabel = f.arcs[k].ilabel; t.next = f.arcs[k].next; t.w = f.arcs[k].w;
        t.str = f.arcs[k].olabel ? strings.Successor(0, f.arcs[k].olabel) : 0;
        carcs[s].push_back(t);
      }
      if (!IsZero(f.fin[s])) { TempArc t; t.ilabel = 0; t.next = -1; t.w = f.fin[s]; t.str = 0; carcs[s].push_back(t); }
    }
  }
  // ---- Connect (:1507): keep states that are reachable from 0 and reach a final state
  const int n = static_cast<int>(carcs.size());
  std::vector<char> acc(n, 0), coacc(n, 0);
  if (n > 0) {
    std::vector<int> stack(1, 0);
    acc[0] = 1;
    while (!stack.empty()) { const int s = stack.back(); stack.pop_back(); for (size_t k = 0; k < carcs[s].size(); k++) { const int d = carcs[s][k].next; if (d >= 0 && !acc[d]) { acc[d] = 1; stack.push_back(d); } } }
    bool changed = true;
    for (int s = 0; s < n; s++) for (size_t k = 0; k < carcs[s].size(); k++) if (carcs[s][k].next < 0) coacc[s] = 1;
    while (changed) {
      changed = false;
      for (int s = n - 1; s >= 0; s--) {
        if (coacc[s]) continue;
        for (size_t k = 0; k < carcs[s].size(); k++) if (carcs[s][k].next >= 0 && coacc[carcs[s][k].next]) { coacc[s] = 1; changed = true; break; }
      }
    }
  }
  std::vector<int> map(n, -1);
  int m = 0;
  for (int s = 0; s < n; s++) if (acc[s] && coacc[s]) map[s] = m++;
  cl->complete = ans;
  cl->start = (n > 0 && map[0] >= 0) ? 0 : -1;
  cl->off.assign(m + 1, 0); cl->fin.assign(2 * static_cast<size_t>(m), kInfF);
  cl->fin_str_begin.assign(m, 0); cl->fin_str_len.assign(m, 0);
  for (int s = 0; s < n; s++) {
    if (map[s] < 0) continue;
    cl->off[map[s]] = static_cast<int>(cl->arcs.size());
    for (size_t k = 0; k < carcs[s].size(); k++) {
      const TempArc &t = carcs[s][k];
      std::vector<int> seq;
      strings.ToVector(t.str, &seq);
      const int32_t begin = static_cast<int32_t>(cl->strings.size());
      cl->strings.insert(cl->strings.end(), seq.begin(), seq.end());
      if (t.next < 0) {
        cl->fin[2 * map[s]] = t.w.g; cl->fin[2 * map[s] + 1] = t.w.a;
        cl->fin_str_begin[map[s]] = begin; cl->fin_str_len[map[s]] = static_cast<int32_t>(seq.size());
      } else if (map[t.next] >= 0) {
        kamd_clat_arc a;
        a.src = map[s]; a.dst = map[t.next]; a.label = t.ilabel; a.graph_cost = t.w.g; a.acoustic_cost = t.w.a;
        a.str_begin = begin; a.str_len = static_cast<int32_t>(seq.size());
        cl->arcs.push_back(a);
      }
    }
  }
  cl->off[m] = static_cast<int>(cl->arcs.size());
  return reinterpret_cast<kamd_compact_lattice *>(cl);
}

void kamd_compact_lattice_destroy(kamd_compact_lattice *h) { delete reinterpret_cast<CompactLattice *>(h); }

int kamd_compact_lattice_sizes(const kamd_compact_lattice *h, int32_t *num_states, int32_t *num_arcs, int32_t *num_labels,
                               int32_t *start, int32_t *reached_beam) {
  const CompactLattice *cl = reinterpret_cast<const CompactLattice *>(h);
  *num_states = static_cast<int32_t>(cl->fin.size() / 2); *num_arcs = static_cast<int32_t>(cl->arcs.size());
  *num_labels = static_cast<int32_t>(cl->strings.size()); *start = cl->start; *reached_beam = cl->complete ? 1 : 0;
  return KAMD_OK;
}

int kamd_compact_lattice_get(const kamd_compact_lattice *h, float *state_final, int32_t *final_str_begin, int32_t *final_str_len,
                             kamd_clat_arc *arcs, int32_t *strings) {
  const CompactLattice *cl = reinterpret_cast<const CompactLattice *>(h);
  const size_t S = cl->fin.size() / 2;
  if (S) {
    memcpy(state_final, cl->fin.data(), sizeof(float) * 2 * S);
    memcpy(final_str_begin, cl->fin_str_begin.data(), sizeof(int32_t) * S);
    memcpy(final_str_len, cl->fin_str_len.data(), sizeof(int32_t) * S);
  }
  if (!cl->arcs.empty()) memcpy(arcs, cl->arcs.data(), sizeof(kamd_clat_arc) * cl->arcs.size());
  if (!cl->strings.empty()) memcpy(strings, cl->strings.data(), sizeof(int32_t) * cl->strings.size());
  return KAMD_OK;
}

// CompactLatticeWriter entry (lat/kaldi-lattice.cc:62-94 WriteCompactLattice): binary =
// OpenFst VectorFst over "compactlattice44" arcs (weight = two floats, int32 length, the
// transition-ids; fstext/lattice-weight.h:532-540); text = acceptor lines
// "src dst label graph,acoustic,tid_tid_..." (:728-740).  acoustic_scale != 1: acoustic costs
// are divided by it (decoder-wrappers.cc:282-284).
// ScaleLattice(GraphLatticeScale(s)) (fstext/lattice-utils.h): graph costs of every arc and final weight times s
int kamd_compact_lattice_scale_graph(kamd_compact_lattice *h, float scale) {
  CompactLattice *cl = reinterpret_cast<CompactLattice *>(h);
  for (kamd_clat_arc &a : cl->arcs) a.graph_cost *= scale;
  for (size_t s = 0; s < cl->fin.size() / 2; s++)
    if (cl->fin[2 * s] != INFINITY) cl->fin[2 * s] *= scale;
  return KAMD_OK;
}

// fst::ScaleLattice with a diagonal scale (fstext/lattice-utils-inl.h:250-290; fst::AcousticLatticeScale /
// GraphLatticeScale): the NnetBatchDecoder hands out lattices whose acoustic costs are already divided by
// the acoustic scale (nnet-batch-compute.cc:1265-1267)
int kamd_compact_lattice_scale(kamd_compact_lattice *h, float graph_scale, float acoustic_scale) {
  CompactLattice *cl = reinterpret_cast<CompactLattice *>(h);
  if (!cl) return kamd::SetError(KAMD_ERR_ARG, "null compact lattice");
  for (kamd_clat_arc &a : cl->arcs) { a.graph_cost *= graph_scale; a.acoustic_cost *= acoustic_scale; }
  for (size_t s = 0; s < cl->fin.size() / 2; s++)
    if (cl->fin[2 * s] != INFINITY) { cl->fin[2 * s] *= graph_scale; cl->fin[2 * s + 1] *= acoustic_scale; }
  return KAMD_OK;
}

kamd_compact_lattice *kamd_compact_lattice_copy(const kamd_compact_lattice *h) {
  if (!h) { kamd::SetError(KAMD_ERR_ARG, "null compact lattice"); return NULL; }
  return reinterpret_cast<kamd_compact_lattice *>(new CompactLattice(*reinterpret_cast<const CompactLattice *>(h)));
}

int kamd_compact_lattice_write(const char *path, int append, const char *key, int binary, const kamd_compact_lattice *h,
                               float acoustic_scale) {
  const CompactLattice *cl = reinterpret_cast<const CompactLattice *>(h);
  const int S = static_cast<int>(cl->fin.size() / 2);
  const float inv = acoustic_scale != 0.0f && acoustic_scale != 1.0f ? 1.0f / acoustic_scale : 1.0f;
  std::string out(key);
  out.push_back(' ');
  auto put_str = [&](std::ostringstream &os, int32_t b, int32_t n) { for (int32_t i = 0; i < n; i++) { os << cl->strings[b + i]; if (i + 1 < n) os << '_'; } };
  auto put_f = [](std::ostringstream &os, float f) {
    if (f == kInfF) os << "Infinity"; else if (f == -kInfF) os << "-Infinity"; else if (f != f) os << "BadNumber"; else os << f;
  };
  if (binary) {
    std::string b;
    auto put = [&](const void *p, size_t n) { b.append(static_cast<const char *>(p), n); };
    auto put_i32 = [&](int32_t v) { put(&v, 4); };
    auto put_i64 = [&](int64_t v) { put(&v, 8); };
    auto put_u64 = [&](uint64_t v) { put(&v, 8); };
    auto put_s = [&](const char *s) { put_i32(static_cast<int32_t>(strlen(s))); put(s, strlen(s)); };
    put_i32(2125659606); put_s("vector"); put_s("compactlattice44"); put_i32(2); put_i32(0); put_u64(0x3);
    put_i64(S > 0 ? cl->start : -1); put_i64(S); put_i64(0);
    for (int s = 0; s < S; s++) {
      const bool fin = cl->fin[2 * s] != kInfF;
      float g = fin ? cl->fin[2 * s] : kInfF, a = fin ? cl->fin[2 * s + 1] * inv : kInfF;
      put(&g, 4); put(&a, 4);
      put_i32(fin ? cl->fin_str_len[s] : 0);
      if (fin) for (int32_t i = 0; i < cl->fin_str_len[s]; i++) put_i32(cl->strings[cl->fin_str_begin[s] + i]);
      put_i64(cl->off[s + 1] - cl->off[s]);
      for (int k = cl->off[s]; k < cl->off[s + 1]; k++) {
        const kamd_clat_arc &x = cl->arcs[k];
        put_i32(x.label); put_i32(x.label);
        float gg = x.graph_cost, aa = x.acoustic_cost * inv;
        put(&gg, 4); put(&aa, 4); put_i32(x.str_len);
        for (int32_t i = 0; i < x.str_len; i++) put_i32(cl->strings[x.str_begin + i]);
        put_i32(x.dst);
      }
    }
    out += b;
  } else {
    std::ostringstream os;
    os << '\n';
    auto print_state = [&](int s) {
      bool output = false;
      for (int k = cl->off[s]; k < cl->off[s + 1]; k++) {
        const kamd_clat_arc &x = cl->arcs[k];
        os << s << '\t' << x.dst << '\t' << x.label;
        if (!(x.graph_cost == 0.0f && x.acoustic_cost == 0.0f && x.str_len == 0)) {
          os << '\t'; put_f(os, x.graph_cost); os << ','; put_f(os, x.acoustic_cost * inv); os << ','; put_str(os, x.str_begin, x.str_len);
        }
        os << '\n';
        output = true;
      }
      const bool fin = cl->fin[2 * s] != kInfF;
      if (fin || !output) {
        os << s;
        if (fin && !(cl->fin[2 * s] == 0.0f && cl->fin[2 * s + 1] == 0.0f && cl->fin_str_len[s] == 0)) {
          os << '\t'; put_f(os, cl->fin[2 * s]); os << ','; put_f(os, cl->fin[2 * s + 1] * inv); os << ','; put_str(os, cl->fin_str_begin[s], cl->fin_str_len[s]);
        } else if (!fin) {
          os << '\t'; put_f(os, kInfF); os << ','; put_f(os, kInfF); os << ',';
        }
        os << '\n';
      }
    };
    if (S > 0 && cl->start >= 0) {
      print_state(cl->start);
      for (int s = 0; s < S; s++) if (s != cl->start) print_state(s);
    }
    os << '\n';
    out += os.str();
  }
  FILE *f = fopen(path, append ? "ab" : "wb");
  if (!f) return kamd::SetError(KAMD_ERR_ARG, "cannot open for writing: %s", path);
  const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
  fclose(f);
  return ok ? KAMD_OK : kamd::SetError(KAMD_ERR_ARG, "write failed: %s", path);
}

}  // extern "C"
