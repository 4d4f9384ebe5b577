// const_arpa.cc -- ConstArpaLm and lattice rescoring with it (host code, no GPU).
//
// Reference: lm/const-arpa-lm.{h,cc} (the compact in-memory n-gram LM: LmState layout :117-160 of the
// .cc, ConstArpaLmBuilder::ReadComplete :330-470, ConstArpaLm::Write / Read :484-700, GetNgramLogprob
// :741-818, GetLmState / GetChildInfo / DecodeChildInfo :820-905, ConstArpaLmDeterministicFst :1000-1062),
// lm/arpa-file-parser.cc:41-262 (the ARPA reader), lat/lattice-functions.cc:1529-1650
// (ComposeCompactLatticeDeterministic) and latbin/lattice-lmrescore-const-arpa.cc:76-110 (the binary:
// scale by 1/lm_scale, compose, determinize, scale back).  SURVEY 8(f) row 4: this is how the reference
// produces the "tglarge" numbers BASELINE quotes (decode with tgsmall, remove its LM cost, add the big LM's).
//
// The on-disk G.carpa format is the reference's, bit for bit (arpa-to-const-arpa output can be read, and
// what is written here can be read by lattice-lmrescore-const-arpa): offsets instead of pointers in
// memory, the same int32 block.  Pinned by the reference's own known answers: the n-grams of
// arpa-file-parser-test.cc and the two sentence scores of arpa-lm-compiler-test.cc (tests/test_const_arpa.py).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <map>
#include <queue>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace kamd {

static inline int32_t FloatBits(float f) { int32_t i; memcpy(&i, &f, 4); return i; }
static inline float BitsFloat(int32_t i) { float f; memcpy(&f, &i, 4); return f; }

struct ConstArpa {
  int32_t bos = -1, eos = -1, unk = -1, order = 0, num_words = 0;
  std::vector<int32_t> lm;          // <lm_states_>: {logprob, backoff, num_children, (word, child_info) x n} per state
  std::vector<int64_t> unigram;     // [num_words]: 0 = none, else offset + 1 (exactly what the file holds)
  std::vector<int64_t> overflow;    // absolute offsets + 1 of children further than 2^30 - 1 ints away

  // GetChildInfo (:849-880): binary search in the sorted children of the state at offset p
  bool ChildInfo(int32_t word, int64_t p, int32_t *info) const {
    const int32_t n = lm[p + 2];
    int32_t lo = 1, hi = n;
    while (lo <= hi) {
      const int32_t mid = (lo + hi) / 2;
      const int32_t w = lm[p + 1 + 2 * mid];
      if (w == word) { *info = lm[p + 2 + 2 * mid]; return true; }
      if (w < word) lo = mid + 1; else hi = mid - 1;
    }
    return false;
  }
  // DecodeChildInfo (:882-905): child = -1 for a leaf
  void Decode(int32_t info, int64_t parent, int64_t *child, float *logprob) const {
    if (info % 2 == 0) { *child = -1; *logprob = BitsFloat(info); return; }
    const int32_t off = info / 2;
    if (off > 0) *child = parent + off;
    else *child = overflow[-off] - 1;
    *logprob = BitsFloat(lm[*child]);
  }
  // GetLmState (:820-847): offset of the state of a word sequence, -1 if it has none
  int64_t State(const int32_t *seq, int n) const {
    if (n == 0) return -1;
    if (seq[0] < 0 || seq[0] >= num_words || unigram[seq[0]] == 0) return -1;
    int64_t parent = unigram[seq[0]] - 1;
    for (int i = 1; i < n; i++) {
      int32_t info;
      if (!ChildInfo(seq[i], parent, &info)) return -1;
      int64_t child; float lp;
      Decode(info, parent, &child, &lp);
      if (child < 0) return -1;
      parent = child;
    }
    return parent;
  }
  bool HistoryStateExists(const std::vector<int32_t> &h) const {   // :717-739
    if (h.empty()) return true;
    const int64_t s = State(h.data(), static_cast<int>(h.size()));
    return s >= 0 && lm[s + 2] > 0;
  }
  float Recurse(int32_t word, const int32_t *hist, int n) const {  // GetNgramLogprobRecurse :781-818
    if (n == 0) {
      if (word < 0 || word >= num_words || unigram[word] == 0) return std::numeric_limits<float>::min();
      return BitsFloat(lm[unigram[word] - 1]);
    }
    float backoff = 0.0f;
    const int64_t s = State(hist, n);
    if (s >= 0) {
      int32_t info;
      if (ChildInfo(word, s, &info)) { int64_t c; float lp; Decode(info, s, &c, &lp); return lp; }
      backoff = BitsFloat(lm[s + 1]);
    }
    return backoff + Recurse(word, hist + 1, n - 1);
  }
  float NgramLogprob(int32_t word, const std::vector<int32_t> &hist_in) const {   // :741-779
    std::vector<int32_t> h(hist_in);
    while (static_cast<int>(h.size()) >= order) h.erase(h.begin());
    int32_t w = word;
    if (unk != -1) {
      if (w < 0 || w >= num_words || unigram[w] == 0) w = unk;
      for (size_t i = 0; i < h.size(); i++)
        if (h[i] < 0 || h[i] >= num_words || unigram[h[i]] == 0) h[i] = unk;
    }
    return Recurse(w, h.data(), static_cast<int>(h.size()));
  }
};

// ---- ARPA reader + builder (arpa-file-parser.cc:41-262, const-arpa-lm.cc:268-470)
struct BuildState {
  bool is_unigram, child_final; float logprob, backoff; int64_t addr = 0;
  std::vector<std::pair<int32_t, int>> child_state;   // (word, index into states)
  std::vector<std::pair<int32_t, float>> child_prob;  // (word, logprob) when the children are of the final order
  bool Leaf() const { return backoff == 0.0f && child_state.empty() && child_prob.empty(); }
  int32_t Mem() const { return (Leaf() && !is_unigram) ? 0 : 3 + 2 * static_cast<int32_t>(child_state.size() + child_prob.size()); }
};

static void TrimRight(std::string *s) { while (!s->empty() && strchr(" \t\n\r", s->back())) s->pop_back(); }
static std::vector<std::string> Split(const std::string &s) {
  std::vector<std::string> out;
  size_t i = 0;
  while (i < s.size()) {
    while (i < s.size() && (s[i] == ' ' || s[i] == '\t')) i++;
    size_t j = i;
    while (j < s.size() && s[j] != ' ' && s[j] != '\t') j++;
    if (j > i) out.push_back(s.substr(i, j - i));
    i = j;
  }
  return out;
}
static bool ToInt(const std::string &s, int32_t *v) {
  if (s.empty()) return false;
  char *e = NULL;
  const long x = strtol(s.c_str(), &e, 10);
  if (*e != '\0') return false;
  *v = static_cast<int32_t>(x);
  return true;
}
static bool ToFloat(const std::string &s, float *v) {
  if (s.empty()) return false;
  char *e = NULL;
  const double x = strtod(s.c_str(), &e);
  if (*e != '\0') return false;
  *v = static_cast<float>(x);
  return true;
}

// One parsed n-gram as the reference's parser reports it (ArpaFileParser::ConsumeNGram)
struct NGram { std::vector<int32_t> words; float logprob, backoff; int line; };

// symbols: word -> id (empty: the file holds integers).  Returns "" or an error message.
static std::string ParseArpa(FILE *f, const std::unordered_map<std::string, int32_t> *symbols, std::vector<int32_t> *counts,
                             std::vector<NGram> *out) {
  char buf[1 << 16];
  std::string line;
  int line_no = 0;
  auto get = [&]() -> bool {
    line.clear();
    bool any = false;
    while (fgets(buf, sizeof(buf), f)) { any = true; line += buf; if (!line.empty() && line.back() == '\n') { line.pop_back(); break; } }
    if (any) line_no++;
    return any;
  };
  auto ref = [&]() { return "line " + std::to_string(line_no) + " [" + line + "]: "; };
  bool keyword = false, have = false;
  while ((have = get())) {
    if (line.find_first_not_of(" \t\n\r") == std::string::npos) continue;
    TrimRight(&line);
    if (!keyword) { if (line == "\\data\\") keyword = true; continue; }
    if (line[0] == '\\') break;
    const size_t eq = line.find('=');
    std::string l2 = line;
    if (eq != std::string::npos) l2.replace(eq, 1, " = ");
    const std::vector<std::string> col = Split(l2);
    if (col.size() == 4 && col[0] == "ngram" && col[2] == "=") {
      int32_t order, n = 0;
      if (!ToInt(col[1], &order) || !ToInt(col[3], &n) || order < 1) return ref() + "cannot parse ngram count";
      if (static_cast<int32_t>(counts->size()) <= order) counts->resize(order);
      (*counts)[order - 1] = n;
    }
  }
  if (counts->empty()) return ref() + "\\data\\ section missing or empty.";
  const int max_order = static_cast<int>(counts->size());
  for (int cur = 1; cur <= max_order; cur++) {
    const std::string kw = "\\" + std::to_string(cur) + "-grams:";
    if (!have || line != kw) return ref() + "invalid directive, expecting '" + kw + "'";
    int seen = 0;
    while ((have = get())) {
      if (line.find_first_not_of(" \n\t\r") == std::string::npos) continue;
      if (line[0] == '\\') {
        TrimRight(&line);
        const std::string next = "\\" + std::to_string(cur + 1) + "-grams:";
        if (line == next || line == "\\end\\") break;
      }
      const std::vector<std::string> col = Split(line);
      const int nc = static_cast<int>(col.size());
      if (nc < 1 + cur || nc > 2 + cur || (cur == max_order && nc != 1 + cur)) return ref() + "Invalid n-gram data line";
      seen++;
      NGram g;
      g.line = line_no;
      if (!ToFloat(col[0], &g.logprob)) return ref() + "invalid n-gram logprob '" + col[0] + "'";
      g.backoff = 0.0f;
      if (nc > cur + 1 && !ToFloat(col[cur + 1], &g.backoff)) return ref() + "invalid backoff weight '" + col[cur + 1] + "'";
      g.logprob *= static_cast<float>(M_LN10); g.backoff *= static_cast<float>(M_LN10);     // to natural log (:183-184)
      g.words.resize(cur);
      for (int i = 0; i < cur; i++) {
        int32_t w;
        if (symbols && !symbols->empty()) {
          auto it = symbols->find(col[1 + i]);
          if (it == symbols->end()) return ref() + "word '" + col[1 + i] + "' not in symbol table";
          w = it->second;
        } else if (!ToInt(col[1 + i], &w) || w < 0) return ref() + "invalid symbol '" + col[1 + i] + "'";
        if (w == 0) return ref() + "epsilon symbol '" + col[1 + i] + "' is illegal in ARPA LM";
        g.words[i] = w;
      }
      out->push_back(g);
    }
    if (seen > (*counts)[cur - 1])
      return ref() + "header said there would be " + std::to_string((*counts)[cur - 1]) + " n-grams of order " + std::to_string(cur) + ", but we saw more already.";
  }
  if (!have || line != "\\end\\") return ref() + "invalid or unexpected directive line, expecting \\end\\";
  return "";
}

static std::string BuildFromNgrams(const std::vector<int32_t> &counts, const std::vector<NGram> &ngrams, ConstArpa *lm) {
  const int order = static_cast<int>(counts.size());
  lm->order = order;
  std::vector<BuildState> states;
  std::map<std::vector<int32_t>, int> seq2state;
  int32_t num_words = 0;
  for (const NGram &g : ngrams) {                     // ConsumeNGram :268-327
    const int cur = static_cast<int>(g.words.size());
    int me = -1;
    if (cur != order || order == 1) {
      if (seq2state.count(g.words)) return "an n-gram appears twice in the arpa file (line " + std::to_string(g.line) + ")";
      BuildState s; s.is_unigram = cur == 1; s.child_final = cur == order - 1; s.logprob = g.logprob; s.backoff = g.backoff;
      states.push_back(s);
      me = static_cast<int>(states.size()) - 1;
      seq2state[g.words] = me;
    }
    const int32_t last = g.words[cur - 1];
    if (cur > 1) {
      const std::vector<int32_t> hist(g.words.begin(), g.words.end() - 1);
      auto it = seq2state.find(hist);
      if (it == seq2state.end()) return "In line " + std::to_string(g.line) + ": " + std::to_string(cur) + "-gram does not have a parent model " + std::to_string(cur) + "-gram.";
      if (cur != order || order == 1) states[it->second].child_state.push_back(std::make_pair(last, me));
      else states[it->second].child_prob.push_back(std::make_pair(last, g.logprob));
    } else num_words = std::max(num_words, last + 1);
  }
  lm->num_words = num_words;
  // ReadComplete :330-470.  std::map iterates word sequences in the lexicographic order the reference sorts into.
  std::vector<int> sorted;
  for (auto &kv : seq2state) if (states[kv.second].Mem() > 0) sorted.push_back(kv.second);
  std::vector<const std::vector<int32_t> *> sorted_seq;
  for (auto &kv : seq2state) if (states[kv.second].Mem() > 0) sorted_seq.push_back(&kv.first);
  int64_t total = 0;
  for (size_t i = 0; i < sorted.size(); i++) { states[sorted[i]].addr = total; total += states[sorted[i]].Mem(); }
  lm->lm.assign(static_cast<size_t>(total), 0);
  lm->unigram.assign(num_words, 0);
  lm->overflow.clear();
  const int64_t max_off = (1ll << 30) - 1;
  int64_t idx = 0;
  for (size_t i = 0; i < sorted.size(); i++) {
    BuildState &s = states[sorted[i]];
    const int64_t parent = idx;
    lm->lm[idx++] = FloatBits(s.logprob);
    lm->lm[idx++] = FloatBits(s.backoff);
    lm->lm[idx++] = static_cast<int32_t>(s.child_state.size() + s.child_prob.size());
    if (s.child_final) {
      std::sort(s.child_prob.begin(), s.child_prob.end(), [](const std::pair<int32_t, float> &a, const std::pair<int32_t, float> &b) { return a.first < b.first; });
      for (auto &c : s.child_prob) { lm->lm[idx++] = c.first; lm->lm[idx++] = FloatBits(c.second) & ~1; }
    } else {
      std::sort(s.child_state.begin(), s.child_state.end(), [](const std::pair<int32_t, int> &a, const std::pair<int32_t, int> &b) { return a.first < b.first; });
      for (auto &c : s.child_state) {
        const BuildState &ch = states[c.second];
        int32_t info;
        if (ch.Mem() == 0) info = FloatBits(ch.logprob) & ~1;     // leaf, not unigram: the logprob sits where the pointer would
        else {
          const int64_t off = ch.addr - s.addr;
          if (off <= max_off) info = static_cast<int32_t>(off * 2) | 1;
          else { lm->overflow.push_back(parent + off + 1); info = -((static_cast<int32_t>(lm->overflow.size() - 1) * 2) | 1); }
        }
        lm->lm[idx++] = c.first; lm->lm[idx++] = info;
      }
    }
    if (s.is_unigram) lm->unigram[(*sorted_seq[i])[0]] = parent + 1;
  }
  if (idx != total) return "internal error: LmState block size mismatch";
  return "";
}

static std::string CheckSymbols(const ConstArpa &lm) {      // the asserts at the end of ReadInternal :660-667
  if (lm.order <= 0) return "n-gram order must be positive";
  if (!(lm.bos < lm.num_words && lm.bos > 0)) return "bad <s> symbol";
  if (!(lm.eos < lm.num_words && lm.eos > 0)) return "bad </s> symbol";
  if (!(lm.unk < lm.num_words && (lm.unk > 0 || lm.unk == -1))) return "bad <unk> symbol";
  return "";
}

// A file's <LmStates> block is followed through child offsets at rescoring time: walk it once after reading, so that a
// truncated or corrupt G.carpa is an error of the read and not an out-of-bounds access later.
static std::string ValidateStates(const ConstArpa &lm) {
  const int64_t size = static_cast<int64_t>(lm.lm.size());
  std::vector<bool> seen(lm.lm.size(), false);
  std::vector<int64_t> stack;
  for (int64_t a : lm.unigram) if (a > 0) stack.push_back(a - 1);
  while (!stack.empty()) {
    const int64_t p = stack.back();
    stack.pop_back();
    if (p < 0 || p + 3 > size) return "LmState offset outside <LmStates>";
    if (seen[p]) continue;                      // (a corrupt file may point two parents at one state, or in a circle)
    seen[p] = true;
    const int64_t n = lm.lm[p + 2];
    if (n < 0 || p + 3 + 2 * n > size) return "LmState children run past the end of <LmStates>";
    for (int64_t k = 0; k < n; k++) {
      const int32_t info = lm.lm[p + 4 + 2 * k];
      if (info % 2 == 0) continue;              // a leaf: the log-probability itself
      const int32_t off = info / 2;
      if (off > 0) stack.push_back(p + off);
      else {
        if (static_cast<size_t>(-static_cast<int64_t>(off)) >= lm.overflow.size()) return "overflow index outside <LmOverflow>";
        stack.push_back(lm.overflow[-off] - 1);
      }
    }
  }
  return "";
}

// ---- ConstArpaLmDeterministicFst (:1000-1062): states are word histories, created on demand
struct LmFst {
  const ConstArpa &lm;
  std::vector<std::vector<int32_t>> state_to_wseq;
  std::map<std::vector<int32_t>, int> wseq_to_state;
  explicit LmFst(const ConstArpa &l) : lm(l) {
    state_to_wseq.push_back(std::vector<int32_t>(1, lm.bos));
    wseq_to_state[state_to_wseq[0]] = 0;
  }
  float Final(int s) const { return -lm.NgramLogprob(lm.eos, state_to_wseq[s]); }
  bool GetArc(int s, int32_t label, int *next, float *weight) {
    std::vector<int32_t> wseq = state_to_wseq[s];
    const float lp = lm.NgramLogprob(label, wseq);
    if (lp == std::numeric_limits<float>::min()) return false;
    wseq.push_back(label);
    while (static_cast<int>(wseq.size()) >= lm.order) wseq.erase(wseq.begin());
    while (!lm.HistoryStateExists(wseq)) wseq.erase(wseq.begin());
    auto r = wseq_to_state.insert(std::make_pair(wseq, static_cast<int>(state_to_wseq.size())));
    if (r.second) state_to_wseq.push_back(wseq);
    *next = r.first->second; *weight = -lp;
    return true;
  }
};

}  // namespace kamd
using kamd::ConstArpa;

extern "C" {

void kamd_const_arpa_destroy(kamd_const_arpa *h) { delete reinterpret_cast<ConstArpa *>(h); }

// arpa-to-const-arpa (lmbin/arpa-to-const-arpa.cc; BuildConstArpaLm const-arpa-lm.cc:1064-1073).  words_txt: optional
// "word id" symbol table for ARPA files that hold words (recipes map to integers first: utils/map_arpa_lm.pl).
kamd_const_arpa *kamd_const_arpa_build(const char *arpa_path, int32_t bos, int32_t eos, int32_t unk, const char *words_txt) {
  std::unordered_map<std::string, int32_t> sym;
  if (words_txt && words_txt[0]) {
    FILE *w = fopen(words_txt, "r");
    if (!w) { kamd::SetError(KAMD_ERR_ARG, "cannot open %s", words_txt); return NULL; }
    char word[4096]; int id;
    while (fscanf(w, "%4095s %d", word, &id) == 2) sym[word] = id;
    fclose(w);
  }
  // ArpaFileParser::Read's option checks (:41-66)
  if (bos <= 0 || eos <= 0 || bos == eos) { kamd::SetError(KAMD_ERR_ARG, "BOS and EOS symbols are required, must not be epsilons, and differ from each other. Given: BOS=%d EOS=%d", bos, eos); return NULL; }
  if (unk != -1 && (unk == 0 || unk == bos || unk == eos)) { kamd::SetError(KAMD_ERR_ARG, "UNK symbol must not be epsilon and must differ from both BOS and EOS symbols. Given: UNK=%d BOS=%d EOS=%d", unk, bos, eos); return NULL; }
  FILE *f = fopen(arpa_path, "r");
  if (!f) { kamd::SetError(KAMD_ERR_ARG, "cannot open %s", arpa_path); return NULL; }
  std::vector<int32_t> counts; std::vector<kamd::NGram> ngrams;
  std::string err = kamd::ParseArpa(f, sym.empty() ? NULL : &sym, &counts, &ngrams);
  fclose(f);
  ConstArpa *lm = new ConstArpa();
  lm->bos = bos; lm->eos = eos; lm->unk = unk;
  if (err.empty()) err = kamd::BuildFromNgrams(counts, ngrams, lm);
  if (err.empty()) err = kamd::CheckSymbols(*lm);
  if (!err.empty()) { kamd::SetError(KAMD_ERR_ARG, "%s: %s", arpa_path, err.c_str()); delete lm; return NULL; }
  return reinterpret_cast<kamd_const_arpa *>(lm);
}

// The parser alone, for the reference's parser tests: up to cap n-grams as (line, order, words[8], logprob, backoff).
int kamd_arpa_parse(const char *arpa_path, const char *words_txt, int32_t *counts, int counts_cap, int32_t *n_counts, int32_t *lines,
                    int32_t *orders, int32_t *words /* [cap][8] */, float *logprob, float *backoff, int cap, int32_t *n) {
  std::unordered_map<std::string, int32_t> sym;
  if (words_txt && words_txt[0]) {
    FILE *w = fopen(words_txt, "r");
    if (!w) return kamd::SetError(KAMD_ERR_ARG, "cannot open %s", words_txt);
    char word[4096]; int id;
    while (fscanf(w, "%4095s %d", word, &id) == 2) sym[word] = id;
    fclose(w);
  }
  FILE *f = fopen(arpa_path, "r");
  if (!f) return kamd::SetError(KAMD_ERR_ARG, "cannot open %s", arpa_path);
  std::vector<int32_t> c; std::vector<kamd::NGram> g;
  const std::string err = kamd::ParseArpa(f, sym.empty() ? NULL : &sym, &c, &g);
  fclose(f);
  if (!err.empty()) return kamd::SetError(KAMD_ERR_ARG, "%s", err.c_str());
  *n_counts = static_cast<int32_t>(c.size()); *n = static_cast<int32_t>(g.size());
  for (size_t i = 0; i < c.size() && static_cast<int>(i) < counts_cap; i++) counts[i] = c[i];
  for (size_t i = 0; i < g.size() && static_cast<int>(i) < cap; i++) {
    lines[i] = g[i].line; orders[i] = static_cast<int32_t>(g[i].words.size()); logprob[i] = g[i].logprob; backoff[i] = g[i].backoff;
    for (int k = 0; k < 8; k++) words[8 * i + k] = k < static_cast<int>(g[i].words.size()) ? g[i].words[k] : 0;
  }
  return KAMD_OK;
}

// ConstArpaLm::Write (:484-560) behind WriteKaldiObject's binary header
int kamd_const_arpa_write(const kamd_const_arpa *h, const char *path) {
  const ConstArpa *lm = reinterpret_cast<const ConstArpa *>(h);
  FILE *f = fopen(path, "wb");
  if (!f) return kamd::SetError(KAMD_ERR_ARG, "cannot open %s for writing", path);
  auto tok = [&](const char *t) { fputs(t, f); fputc(' ', f); };
  auto i32 = [&](int32_t v) { fputc(4, f); fwrite(&v, 4, 1, f); };
  auto i64 = [&](int64_t v) { fputc(8, f); fwrite(&v, 8, 1, f); };
  fputc('\0', f); fputc('B', f);
  tok("<ConstArpaLm>");
  tok("<LmInfo>"); i32(lm->bos); i32(lm->eos); i32(lm->unk); i32(lm->order); tok("</LmInfo>");
  tok("<LmStates>"); i64(static_cast<int64_t>(lm->lm.size()));
  if (!lm->lm.empty()) fwrite(lm->lm.data(), 4, lm->lm.size(), f);
  tok("</LmStates>");
  tok("<LmUnigram>"); i32(lm->num_words);
  if (lm->num_words) fwrite(lm->unigram.data(), 8, lm->unigram.size(), f);
  tok("</LmUnigram>");
  tok("<LmOverflow>"); i32(static_cast<int32_t>(lm->overflow.size()));
  if (!lm->overflow.empty()) fwrite(lm->overflow.data(), 8, lm->overflow.size(), f);
  tok("</LmOverflow>");
  tok("</ConstArpaLm>");
  const bool ok = !ferror(f);
  fclose(f);
  return ok ? KAMD_OK : kamd::SetError(KAMD_ERR_ARG, "%s: write failed", path);
}

// ConstArpaLm::Read (:562-577): new format (token <ConstArpaLm>) or the old one (starts with the size byte 4)
kamd_const_arpa *kamd_const_arpa_read(const char *path) {
  FILE *f = fopen(path, "rb");
  if (!f) { kamd::SetError(KAMD_ERR_ARG, "cannot open %s", path); return NULL; }
  ConstArpa *lm = new ConstArpa();
  bool ok = true;
  auto fail = [&](const char *why) -> kamd_const_arpa * { kamd::SetError(KAMD_ERR_ARG, "%s: %s", path, why); fclose(f); delete lm; return NULL; };
  fseek(f, 0, SEEK_END);
  const int64_t file_size = ftell(f);            // no section can hold more than the file does
  fseek(f, 0, SEEK_SET);
  try {
  if (fgetc(f) != '\0' || fgetc(f) != 'B') return fail("not a Kaldi binary file (text-mode reading is not implemented for ConstArpaLm)");
  auto expect = [&](const char *t) {
    char buf[64]; size_t n = 0; int c;
    while ((c = fgetc(f)) != EOF && c != ' ' && n < sizeof(buf) - 1) buf[n++] = static_cast<char>(c);
    buf[n] = 0;
    if (strcmp(buf, t) != 0) ok = false;
  };
  auto i32 = [&]() -> int32_t { int32_t v = 0; if (fgetc(f) != 4 || fread(&v, 4, 1, f) != 1) ok = false; return v; };
  auto i64 = [&]() -> int64_t { int64_t v = 0; if (fgetc(f) != 8 || fread(&v, 8, 1, f) != 1) ok = false; return v; };
  const int first = fgetc(f);
  ungetc(first, f);
  if (first == 4) {                                   // ReadInternalOldFormat :670-715: every value with its size byte
    lm->bos = i32(); lm->eos = i32(); lm->unk = i32(); lm->order = i32();
    const int32_t n = i32();
    if (!ok || n < 0 || 5ll * n > file_size) return fail("corrupt old-format header");
    lm->lm.resize(n);
    for (int32_t i = 0; ok && i < n; i++) lm->lm[i] = i32();
    lm->num_words = i32();
    if (!ok || lm->num_words < 0 || 9ll * lm->num_words > file_size) return fail("corrupt unigram section");
    lm->unigram.resize(lm->num_words);
    for (int32_t i = 0; ok && i < lm->num_words; i++) lm->unigram[i] = i64();
    const int32_t no = i32();
    if (!ok || no < 0 || 9ll * no > file_size) return fail("corrupt overflow section");
    lm->overflow.resize(no);
    for (int32_t i = 0; ok && i < no; i++) lm->overflow[i] = i64();
  } else {
    expect("<ConstArpaLm>"); expect("<LmInfo>");
    lm->bos = i32(); lm->eos = i32(); lm->unk = i32(); lm->order = i32();
    expect("</LmInfo>"); expect("<LmStates>");
    const int64_t n = i64();
    if (!ok || n < 0 || 4 * n > file_size) return fail("corrupt <LmStates> header");
    lm->lm.resize(static_cast<size_t>(n));
    if (n && fread(lm->lm.data(), 4, static_cast<size_t>(n), f) != static_cast<size_t>(n)) return fail("ConstArpaLm <LmStates> section reading failed.");
    expect("</LmStates>"); expect("<LmUnigram>");
    lm->num_words = i32();
    if (!ok || lm->num_words < 0 || 8ll * lm->num_words > file_size) return fail("corrupt <LmUnigram> header");
    lm->unigram.resize(lm->num_words);
    if (lm->num_words && fread(lm->unigram.data(), 8, lm->unigram.size(), f) != lm->unigram.size()) return fail("ConstArpaLm <LmUnigram> section reading failed.");
    expect("</LmUnigram>"); expect("<LmOverflow>");
    const int32_t no = i32();
    if (!ok || no < 0 || 8ll * no > file_size) return fail("corrupt <LmOverflow> header");
    lm->overflow.resize(no);
    if (no && fread(lm->overflow.data(), 8, lm->overflow.size(), f) != lm->overflow.size()) return fail("ConstArpaLm <LmOverflow> section reading failed.");
    expect("</LmOverflow>"); expect("</ConstArpaLm>");
  }
  if (!ok) return fail("unexpected token or truncated ConstArpaLm");
  const std::string err = kamd::CheckSymbols(*lm);
  if (!err.empty()) return fail(err.c_str());
  for (int64_t a : lm->unigram) if (a < 0 || a > static_cast<int64_t>(lm->lm.size())) return fail("unigram offset outside <LmStates>");
  for (int64_t a : lm->overflow) if (a <= 0 || a > static_cast<int64_t>(lm->lm.size())) return fail("overflow offset outside <LmStates>");
  const std::string bad = kamd::ValidateStates(*lm);
  if (!bad.empty()) return fail(bad.c_str());
  } catch (const std::bad_alloc &) {
    return fail("out of memory");
  }
  fclose(f);
  return reinterpret_cast<kamd_const_arpa *>(lm);
}

int kamd_const_arpa_info(const kamd_const_arpa *h, int32_t *bos, int32_t *eos, int32_t *unk, int32_t *order, int32_t *num_words, int64_t *lm_states_size) {
  const ConstArpa *lm = reinterpret_cast<const ConstArpa *>(h);
  *bos = lm->bos; *eos = lm->eos; *unk = lm->unk; *order = lm->order; *num_words = lm->num_words;
  *lm_states_size = static_cast<int64_t>(lm->lm.size());
  return KAMD_OK;
}

// ConstArpaLm::GetNgramLogprob (:741-779): natural-log probability of `word` after history hist[0..n)
float kamd_const_arpa_ngram_logprob(const kamd_const_arpa *h, int32_t word, const int32_t *hist, int n) {
  return reinterpret_cast<const ConstArpa *>(h)->NgramLogprob(word, std::vector<int32_t>(hist, hist + n));
}

// lattice-lmrescore-const-arpa on one CompactLattice (latbin/lattice-lmrescore-const-arpa.cc:76-110):
// ScaleLattice(GraphLatticeScale(1 / lm_scale)); compose with the LM as a deterministic on-demand FST
// (ComposeCompactLatticeDeterministic, lat/lattice-functions.cc:1529-1650); ConvertLattice + Invert +
// DeterminizeLattice; ScaleLattice(GraphLatticeScale(lm_scale)).  Returns NULL with "Empty lattice ..." when the
// composition is empty (the binary's n_fail case).
kamd_compact_lattice *kamd_compact_lattice_lmrescore_const_arpa(int32_t num_states, int32_t start, const float *state_final /* [2S] */,
                                                                const int32_t *final_str_begin, const int32_t *final_str_len,
                                                                const kamd_clat_arc *arcs, int32_t num_arcs, const int32_t *strings,
                                                                const kamd_const_arpa *lmh, float lm_scale) {
  const ConstArpa &lm = *reinterpret_cast<const ConstArpa *>(lmh);
  if (num_states <= 0 || start < 0 || start >= num_states) { kamd::SetError(KAMD_ERR_ARG, "empty lattice"); return NULL; }
  if (lm_scale == 0.0f) { kamd::SetError(KAMD_ERR_ARG, "lm_scale = 0: nothing to do (copy the lattice)"); return NULL; }
  const float inv = 1.0f / lm_scale;
  // arcs by source state, in label order (ArcSort(OLabelCompare) :83)
  std::vector<int> off(num_states + 1, 0), order(num_arcs);
  for (int i = 0; i < num_arcs; i++) {
    if (arcs[i].src < 0 || arcs[i].src >= num_states || arcs[i].dst < 0 || arcs[i].dst >= num_states) { kamd::SetError(KAMD_ERR_ARG, "lattice arc %d out of range", i); return NULL; }
    off[arcs[i].src + 1]++;
  }
  for (int s = 0; s < num_states; s++) off[s + 1] += off[s];
  { std::vector<int> fill(off.begin(), off.end() - 1); for (int i = 0; i < num_arcs; i++) order[fill[arcs[i].src]++] = i; }
  for (int s = 0; s < num_states; s++)
    std::stable_sort(order.begin() + off[s], order.begin() + off[s + 1], [&](int a, int b) { return arcs[a].label < arcs[b].label; });
  // ---- composition: BFS over (lattice state, LM state)
  kamd::LmFst fst(lm);
  std::map<std::pair<int, int>, int> state_map;
  std::queue<std::pair<int, int>> q;
  struct OutArc { int src, dst; int32_t label; float g, a; int sb, sl; };
  std::vector<OutArc> out_arcs;
  std::vector<float> out_fin;              // [2S']
  std::vector<std::pair<int, int>> out_fin_str;
  auto add_state = [&]() { out_fin.push_back(INFINITY); out_fin.push_back(INFINITY); out_fin_str.push_back(std::make_pair(0, 0)); return static_cast<int>(out_fin.size() / 2) - 1; };
  state_map[std::make_pair(start, 0)] = add_state();
  q.push(std::make_pair(start, 0));
  while (!q.empty()) {
    const std::pair<int, int> s = q.front(); q.pop();
    const int me = state_map[s];
    const float f1 = state_final[2 * s.first] * inv, f2 = state_final[2 * s.first + 1];
    if (state_final[2 * s.first] != INFINITY) {
      const float lf = fst.Final(s.second);
      if (lf != INFINITY) { out_fin[2 * me] = f1 + lf; out_fin[2 * me + 1] = f2; out_fin_str[me] = std::make_pair(final_str_begin[s.first], final_str_len[s.first]); }
    }
    for (int k = off[s.first]; k < off[s.first + 1]; k++) {
      const kamd_clat_arc &a = arcs[order[k]];
      int n2 = s.second; float w2 = 0.0f; bool matched = true;
      if (a.label != 0) matched = fst.GetArc(s.second, a.label, &n2, &w2);
      if (!matched) continue;
      const std::pair<int, int> np(a.dst, n2);
      auto it = state_map.find(np);
      int ns;
      if (it == state_map.end()) { ns = add_state(); state_map[np] = ns; q.push(np); } else ns = it->second;
      OutArc o; o.src = me; o.dst = ns; o.label = a.label; o.g = a.graph_cost * inv + (a.label != 0 ? w2 : 0.0f); o.a = a.acoustic_cost;
      o.sb = a.str_begin; o.sl = a.str_len;
      out_arcs.push_back(o);
    }
  }
  // ---- ConvertLattice(CompactLattice -> Lattice): a compact arc with a k-long string becomes a chain whose first arc
  // carries the word and the weight, the others epsilon words and One(); final strings likewise (fstext/lattice-utils-inl.h)
  std::vector<kamd_lat_arc> lat;
  std::vector<float> lfin(out_fin);
  auto new_lat_state = [&]() { lfin.push_back(INFINITY); lfin.push_back(INFINITY); return static_cast<int>(lfin.size() / 2) - 1; };
  for (const OutArc &o : out_arcs) {
    int cur = o.src;
    const int n = std::max(o.sl, 1);
    for (int j = 0; j < n; j++) {
      kamd_lat_arc x;
      x.src = cur; x.dst = (j + 1 == n) ? o.dst : new_lat_state();
      x.ilabel = j < o.sl ? strings[o.sb + j] : 0; x.olabel = j == 0 ? o.label : 0;
      x.graph_cost = j == 0 ? o.g : 0.0f; x.acoustic_cost = j == 0 ? o.a : 0.0f;
      lat.push_back(x); cur = x.dst;
    }
  }
  const int S1 = static_cast<int>(out_fin.size() / 2);
  for (int s = 0; s < S1; s++) {
    if (out_fin[2 * s] == INFINITY || out_fin_str[s].second == 0) continue;
    int cur = s;
    for (int j = 0; j < out_fin_str[s].second; j++) {
      kamd_lat_arc x;
      x.src = cur; x.dst = new_lat_state(); x.ilabel = strings[out_fin_str[s].first + j]; x.olabel = 0;
      x.graph_cost = j == 0 ? out_fin[2 * s] : 0.0f; x.acoustic_cost = j == 0 ? out_fin[2 * s + 1] : 0.0f;
      lat.push_back(x); cur = x.dst;
    }
    lfin[2 * s] = INFINITY; lfin[2 * s + 1] = INFINITY;
    lfin[2 * cur] = 0.0f; lfin[2 * cur + 1] = 0.0f;
  }
  bool any_final = false;
  for (size_t s = 0; s < lfin.size() / 2; s++) if (lfin[2 * s] != INFINITY) any_final = true;
  if (!any_final) { kamd::SetError(KAMD_ERR_STATE, "Empty lattice (incompatible LM?)"); return NULL; }
  std::stable_sort(lat.begin(), lat.end(), [](const kamd_lat_arc &a, const kamd_lat_arc &b) { return a.src < b.src; });
  // ---- DeterminizeLattice (no pruning: an unbounded beam, word labels only), then scale the graph costs back
  kamd_determinize_opts d;
  kamd_determinize_opts_default(&d);
  d.phone_determinize = 0; d.word_determinize = 1; d.max_mem = 0;
  kamd_compact_lattice *res = kamd_lattice_determinize_phone_pruned(static_cast<int32_t>(lfin.size() / 2), 0, lfin.data(), lat.data(),
                                                                    static_cast<int32_t>(lat.size()), NULL, 0, 1.0e10, &d);
  if (!res) return NULL;
  if (kamd_compact_lattice_scale_graph(res, lm_scale) != KAMD_OK) { kamd_compact_lattice_destroy(res); return NULL; }
  return res;
}

}  // extern "C"
