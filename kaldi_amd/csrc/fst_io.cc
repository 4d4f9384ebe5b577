// On-disk formats either side of the decoder: OpenFst binary FSTs for HCLG
// (what ReadFstKaldiGeneric accepts, fstext/kaldi-fst-io.cc:44-89: "vector" or "const" with
// StdArc) and Kaldi lattice archives (lat/kaldi-lattice.cc:62-130 WriteLattice /
// WriteCompactLattice, :366-420 readers; util/kaldi-table TableWriter entry = key, space,
// object).  Host code only: no device calls here.
//
// OpenFst 1.6.7 is not vendored in the reference tree, so the binary layout below is a
// restatement of its published file format (fst/fst.h FstHeader, fst/vector-fst.h,
// fst/const-fst.h, fst/symbol-table.h), anchored by the facts the reference itself states:
// the first byte of the magic number on little-endian machines is 214
// (lat/kaldi-lattice.cc:377), lattice arc types are "lattice4" / "compactlattice44"
// (fstext/lattice-weight.h:85-88, 471-475), HCLG is converted to "const" by the recipes.
// PARITY UNPINNED: no OpenFst file exists in the reference tree to check against; the tests
// are write/read round trips (tests/test_fst_io.py).
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

#include "common.h"

namespace {

const int32_t kFstMagic = 2125659606;        // fst/fst.h kFstMagicNumber
const int32_t kSymTabMagic = 2125658996;     // fst/symbol-table.h kSymbolTableMagicNumber
enum { kHasISymbols = 1, kHasOSymbols = 2, kIsAligned = 4 };   // FstHeader flags
const uint64_t kExpanded = 0x1, kMutable = 0x2;                // fst/properties.h
const int kArchAlignment = 16;               // fst/const-fst.h

struct Reader {
  FILE *f;
  long pos;
  bool ok;
  explicit Reader(FILE *fp) : f(fp), pos(0), ok(true) {}
  void Bytes(void *p, size_t n) {
    if (!ok) return;
    if (n && fread(p, 1, n, f) != n) ok = false;
    pos += static_cast<long>(n);
  }
  template <class T> T Get() { T v = T(); Bytes(&v, sizeof(T)); return v; }
  std::string Str() {                        // int32 length + bytes
    int32_t n = Get<int32_t>();
    if (!ok || n < 0 || n > (1 << 20)) { ok = false; return std::string(); }
    std::string s(static_cast<size_t>(n), '\0');
    Bytes(n ? &s[0] : NULL, static_cast<size_t>(n));
    return s;
  }
  void Align() { while (ok && pos % kArchAlignment) { char c; Bytes(&c, 1); } }
  void SkipSymbolTable() {                   // fst/symbol-table.cc SymbolTableImpl::Read
    if (Get<int32_t>() != kSymTabMagic) { ok = false; return; }
    Str();                                   // name
    Get<int64_t>();                          // available_key
    int64_t n = Get<int64_t>();
    for (int64_t i = 0; ok && i < n; i++) { Str(); Get<int64_t>(); }
  }
};

struct Header {
  std::string fsttype, arctype;
  int32_t version, flags;
  uint64_t properties;
  int64_t start, numstates, numarcs;
};

// The layout is restated from the published format and has never met a file written by OpenFst itself (none exists in
// this image): every field this reader was not written for is refused by name instead of being guessed at.
bool ReadHeader(Reader *r, Header *h, std::string *why = NULL) {
  auto fail = [&](const std::string &m) { if (why) *why = m; return false; };
  const int32_t magic = r->Get<int32_t>();
  if (!r->ok || magic != kFstMagic) return fail("magic number " + std::to_string(magic) + ", expected " + std::to_string(kFstMagic));
  h->fsttype = r->Str(); h->arctype = r->Str();
  h->version = r->Get<int32_t>(); h->flags = r->Get<int32_t>();
  h->properties = r->Get<uint64_t>();
  h->start = r->Get<int64_t>(); h->numstates = r->Get<int64_t>(); h->numarcs = r->Get<int64_t>();
  if (!r->ok) return fail("truncated header");
  if (h->flags & ~(kHasISymbols | kHasOSymbols | kIsAligned))
    return fail("header flags " + std::to_string(h->flags) + ": bits other than HAS_ISYMBOLS | HAS_OSYMBOLS | IS_ALIGNED are not understood");
  // vector-fst.h kFileVersion = 2, const-fst.h kFileVersion = 2 / kAlignedFileVersion = 1 (OpenFst 1.6.x)
  if (h->version != 1 && h->version != 2) return fail("file version " + std::to_string(h->version) + ", this reader knows 1 and 2 (OpenFst 1.6.x)");
  if (h->flags & kHasISymbols) { r->SkipSymbolTable(); if (!r->ok) return fail("embedded input symbol table in a layout this reader does not know"); }
  if (h->flags & kHasOSymbols) { r->SkipSymbolTable(); if (!r->ok) return fail("embedded output symbol table in a layout this reader does not know"); }
  return true;
}

struct Writer {
  std::string buf;
  void Bytes(const void *p, size_t n) { buf.append(static_cast<const char *>(p), n); }
  template <class T> void Put(T v) { Bytes(&v, sizeof(T)); }
  void Str(const std::string &s) { Put<int32_t>(static_cast<int32_t>(s.size())); Bytes(s.data(), s.size()); }
  void Align(size_t base) { while ((buf.size() - base) % kArchAlignment) buf.push_back('\0'); }
  void Header(const char *fsttype, const char *arctype, int32_t version, int32_t flags, uint64_t props,
              int64_t start, int64_t numstates, int64_t numarcs) {
    Put<int32_t>(kFstMagic); Str(fsttype); Str(arctype); Put<int32_t>(version); Put<int32_t>(flags);
    Put<uint64_t>(props); Put<int64_t>(start); Put<int64_t>(numstates); Put<int64_t>(numarcs);
  }
};

int Fail(const char *what, const char *path) { return kamd::SetError(KAMD_ERR_ARG, "%s: %s", what, path ? path : ""); }

// lattice weight text form (fstext/lattice-weight.h:162-171, 396-404)
void PutFloat(std::ostringstream &os, float f) {
  if (f == std::numeric_limits<float>::infinity()) os << "Infinity";
  else if (f == -std::numeric_limits<float>::infinity()) os << "-Infinity";
  else if (f != f) os << "BadNumber";
  else os << f;
}
bool GetFloat(const std::string &s, float *f) {      // :174-190
  if (s == "Infinity") { *f = std::numeric_limits<float>::infinity(); return true; }
  if (s == "-Infinity") { *f = -std::numeric_limits<float>::infinity(); return true; }
  if (s == "BadNumber") { *f = std::numeric_limits<float>::quiet_NaN(); return true; }
  char *end = NULL;
  errno = 0;
  double v = strtod(s.c_str(), &end);
  if (end == s.c_str() || *end != '\0') return false;
  *f = static_cast<float>(v);
  return true;
}

struct LatState { float f1, f2; std::vector<int32_t> fstr; bool is_final; };

}  // namespace

extern "C" {

// ------------------------------------------------------------------ HCLG
int kamd_openfst_read(const char *path, int32_t *num_states, int32_t *start, int64_t **arc_off,
                      kamd_arc **arcs, float **final_cost) {
  *arc_off = NULL; *arcs = NULL; *final_cost = NULL; *num_states = 0; *start = -1;
  FILE *f = fopen(path, "rb");
  if (!f) return Fail("cannot open", path);
  Reader r(f);
  Header h;
  std::string why;
  if (!ReadHeader(&r, &h, &why)) { fclose(f); return kamd::SetError(KAMD_ERR_ARG, "%s: not an OpenFst binary file this reader accepts: %s", path, why.c_str()); }
  if (h.arctype != "standard") { fclose(f); return kamd::SetError(KAMD_ERR_ARG, "%s: arc type '%s', expected 'standard'", path, h.arctype.c_str()); }
  std::vector<int64_t> off;
  std::vector<kamd_arc> a;
  std::vector<float> fin;
  if (h.fsttype == "const") {
    // fst/const-fst.h ConstFstImpl::Read: [pad] states {final, pos, narcs, niepsilons, noepsilons}, [pad] arcs
    const bool aligned = (h.flags & kIsAligned) != 0;
    if (aligned) r.Align();
    const int64_t S = h.numstates, A = h.numarcs;
    if (S < 0 || A < 0) { fclose(f); return Fail("const FST with unknown size", path); }
    off.resize(S + 1); fin.resize(S); a.resize(A);
    for (int64_t s = 0; s < S; s++) {
      fin[s] = r.Get<float>();
      const uint32_t pos = r.Get<uint32_t>(), narcs = r.Get<uint32_t>();
      r.Get<uint32_t>(); r.Get<uint32_t>();
      off[s] = pos;
      if (s + 1 == S) off[S] = static_cast<int64_t>(pos) + narcs;
      if (r.ok && s > 0 && off[s] < off[s - 1]) r.ok = false;
    }
    if (S == 0) off[0] = 0;
    if (aligned) r.Align();
    r.Bytes(A ? a.data() : NULL, static_cast<size_t>(A) * sizeof(kamd_arc));
    if (r.ok && off[S] != A) r.ok = false;
  } else if (h.fsttype == "vector") {
    // fst/vector-fst.h VectorFstImpl::Read: per state {final, int64 narcs, arcs}; numstates may be -1
    off.push_back(0);
    for (int64_t s = 0; h.numstates < 0 || s < h.numstates; s++) {
      float fc = r.Get<float>();
      if (!r.ok) { if (h.numstates < 0) { r.ok = true; break; } break; }
      const int64_t n = r.Get<int64_t>();
      if (!r.ok || n < 0) { r.ok = false; break; }
      fin.push_back(fc);
      const size_t base = a.size();
      a.resize(base + static_cast<size_t>(n));
      r.Bytes(n ? &a[base] : NULL, static_cast<size_t>(n) * sizeof(kamd_arc));
      off.push_back(static_cast<int64_t>(a.size()));
    }
  } else {
    fclose(f);
    return kamd::SetError(KAMD_ERR_ARG, "%s: FST type '%s', expected 'vector' or 'const'", path, h.fsttype.c_str());
  }
  fclose(f);
  if (!r.ok) return Fail("truncated or corrupt FST", path);
  const int64_t S = static_cast<int64_t>(fin.size());
  for (size_t i = 0; i < a.size(); i++)
    if (a[i].nextstate < 0 || a[i].nextstate >= S) return Fail("arc to a state that does not exist", path);
  if (S > 2147483647LL || h.start >= S) return Fail("bad state count / start state", path);
  *num_states = static_cast<int32_t>(S); *start = static_cast<int32_t>(h.start);
  *arc_off = static_cast<int64_t *>(malloc(sizeof(int64_t) * (S + 1)));
  *arcs = static_cast<kamd_arc *>(malloc(sizeof(kamd_arc) * (a.size() + 1)));
  *final_cost = static_cast<float *>(malloc(sizeof(float) * (S + 1)));
  if (!*arc_off || !*arcs || !*final_cost) return kamd::SetError(KAMD_ERR_ARG, "out of host memory");
  memcpy(*arc_off, off.data(), sizeof(int64_t) * (S + 1));
  if (!a.empty()) memcpy(*arcs, a.data(), sizeof(kamd_arc) * a.size());
  if (S) memcpy(*final_cost, fin.data(), sizeof(float) * S);
  return KAMD_OK;
}

void kamd_host_free(void *p) { free(p); }

// ReadFstKaldiGeneric (fstext/kaldi-fst-io.cc:44-89) straight into HBM
kamd_graph *kamd_graph_read_openfst(const char *path) {
  int32_t S = 0, start = -1;
  int64_t *off = NULL; kamd_arc *arcs = NULL; float *fin = NULL;
  kamd_graph *g = NULL;
  if (kamd_openfst_read(path, &S, &start, &off, &arcs, &fin) == KAMD_OK) {
    if (S <= 0 || start < 0) kamd::SetError(KAMD_ERR_ARG, "%s: empty FST", path);
    else g = kamd_graph_create(S, start, off, arcs, fin);
  }
  free(off); free(arcs); free(fin);
  return g;
}

int kamd_openfst_write(const char *path, int fst_type, int align, int32_t num_states, int32_t start,
                       const int64_t *arc_off, const kamd_arc *arcs, const float *final_cost) {
  Writer w;
  const int64_t A = num_states > 0 ? arc_off[num_states] : 0;
  if (fst_type == 1) {                       // fst/const-fst.h ConstFst::WriteFst
    w.Header("const", "standard", align ? 1 : 2, align ? kIsAligned : 0, kExpanded, start, num_states, A);
    if (align) w.Align(0);
    for (int32_t s = 0; s < num_states; s++) {
      uint32_t ni = 0, no = 0;
      for (int64_t k = arc_off[s]; k < arc_off[s + 1]; k++) { ni += arcs[k].ilabel == 0; no += arcs[k].olabel == 0; }
      w.Put<float>(final_cost[s]); w.Put<uint32_t>(static_cast<uint32_t>(arc_off[s]));
      w.Put<uint32_t>(static_cast<uint32_t>(arc_off[s + 1] - arc_off[s])); w.Put<uint32_t>(ni); w.Put<uint32_t>(no);
    }
    if (align) w.Align(0);
    w.Bytes(arcs, static_cast<size_t>(A) * sizeof(kamd_arc));
  } else if (fst_type == 0) {                // fst/vector-fst.h VectorFst::WriteFst
    w.Header("vector", "standard", 2, 0, kExpanded | kMutable, start, num_states, 0);
    for (int32_t s = 0; s < num_states; s++) {
      w.Put<float>(final_cost[s]); w.Put<int64_t>(arc_off[s + 1] - arc_off[s]);
      w.Bytes(arcs + arc_off[s], static_cast<size_t>(arc_off[s + 1] - arc_off[s]) * sizeof(kamd_arc));
    }
  } else {
    return kamd::SetError(KAMD_ERR_ARG, "fst_type must be 0 (vector) or 1 (const)");
  }
  FILE *f = fopen(path, "wb");
  if (!f) return Fail("cannot open for writing", path);
  const bool ok = fwrite(w.buf.data(), 1, w.buf.size(), f) == w.buf.size();
  fclose(f);
  return ok ? KAMD_OK : Fail("write failed", path);
}

// ------------------------------------------------------------------ lattices
// One archive entry: "key " + object.  Binary object = OpenFst VectorFst<LatticeArc>
// ("lattice4": weight = two floats) with NO Kaldi binary marker (lat/kaldi-lattice.h:75-80);
// text object = '\n' + FstPrinter lines (start state first, tab separated, weights
// "graph,acoustic", One() weights omitted) + '\n' (lat/kaldi-lattice.cc:71-88, 96-130).
// state_final[2*s], [2*s+1] = final weight (graph, acoustic); graph = +inf: not final.
int kamd_lattice_write(const char *path, int append, const char *key, int binary, int32_t num_states,
                       int32_t start, const float *state_final, const kamd_lat_arc *arcs, int32_t num_arcs) {
  const float kInf = std::numeric_limits<float>::infinity();
  std::vector<int64_t> off(static_cast<size_t>(num_states) + 1, 0);
  for (int32_t i = 0; i < num_arcs; i++) {
    if (arcs[i].src < 0 || arcs[i].src >= num_states || arcs[i].dst < 0 || arcs[i].dst >= num_states)
      return kamd::SetError(KAMD_ERR_ARG, "lattice arc %d out of range", i);
    if (i > 0 && arcs[i].src < arcs[i - 1].src) return kamd::SetError(KAMD_ERR_ARG, "lattice arcs must be sorted by source state");
    off[arcs[i].src + 1]++;
  }
  for (int32_t s = 0; s < num_states; s++) off[s + 1] += off[s];
  std::string out(key);
  out.push_back(' ');
  if (binary) {
    Writer w;
    w.Header("vector", "lattice4", 2, 0, kExpanded | kMutable, num_states > 0 ? start : -1, num_states, 0);
    for (int32_t s = 0; s < num_states; s++) {
      const bool fin = state_final[2 * s] != kInf;
      w.Put<float>(fin ? state_final[2 * s] : kInf); w.Put<float>(fin ? state_final[2 * s + 1] : kInf);   // Zero() = (inf, inf)
      w.Put<int64_t>(off[s + 1] - off[s]);
      for (int64_t k = off[s]; k < off[s + 1]; k++) {
        w.Put<int32_t>(arcs[k].ilabel); w.Put<int32_t>(arcs[k].olabel);
        w.Put<float>(arcs[k].graph_cost); w.Put<float>(arcs[k].acoustic_cost); w.Put<int32_t>(arcs[k].dst);
      }
    }
    out += w.buf;
  } else {
    std::ostringstream os;
    os << '\n';
    auto print_state = [&](int32_t s) {      // fst/script/print-impl.h FstPrinter::PrintState
      bool output = false;
      for (int64_t k = off[s]; k < off[s + 1]; k++) {
        os << s << '\t' << arcs[k].dst << '\t' << arcs[k].ilabel << '\t' << arcs[k].olabel;
        if (!(arcs[k].graph_cost == 0.0f && arcs[k].acoustic_cost == 0.0f)) {
          os << '\t'; PutFloat(os, arcs[k].graph_cost); os << ','; PutFloat(os, arcs[k].acoustic_cost);
        }
        os << '\n';
        output = true;
      }
      const bool fin = state_final[2 * s] != kInf;
      if (fin || !output) {
        os << s;
        if (fin && !(state_final[2 * s] == 0.0f && state_final[2 * s + 1] == 0.0f)) {
          os << '\t'; PutFloat(os, state_final[2 * s]); os << ','; PutFloat(os, state_final[2 * s + 1]);
        } else if (!fin) {
          os << '\t'; PutFloat(os, kInf); os << ','; PutFloat(os, kInf);
        }
        os << '\n';
      }
    };
    if (num_states > 0 && start >= 0) {
      print_state(start);
      for (int32_t s = 0; s < num_states; s++) if (s != start) print_state(s);
    }
    os << '\n';
    out += os.str();
  }
  FILE *f = fopen(path, append ? "ab" : "wb");
  if (!f) return Fail("cannot open for writing", path);
  const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
  fclose(f);
  return ok ? KAMD_OK : Fail("write failed", path);
}

// Reads the next archive entry at byte offset *offset (updated).  Lattice entries of either
// form; the first byte after "key " decides: 214 = OpenFst binary, else text
// (lat/kaldi-lattice.cc:366-386).  Outputs are malloc'ed (kamd_host_free).  Returns 1 at
// end of file.
int kamd_lattice_read(const char *path, int64_t *offset, char *key, int key_cap, int32_t *num_states,
                      int32_t *start, float **state_final, kamd_lat_arc **arcs, int32_t *num_arcs) {
  const float kInf = std::numeric_limits<float>::infinity();
  *state_final = NULL; *arcs = NULL; *num_states = 0; *num_arcs = 0; *start = -1;
  FILE *f = fopen(path, "rb");
  if (!f) return Fail("cannot open", path);
  if (fseek(f, static_cast<long>(*offset), SEEK_SET) != 0) { fclose(f); return Fail("seek failed", path); }
  std::string k;
  int ch;
  while ((ch = fgetc(f)) != EOF && (ch == '\n' || ch == ' ')) {}
  if (ch == EOF) { fclose(f); return 1; }
  k.push_back(static_cast<char>(ch));
  while ((ch = fgetc(f)) != EOF && ch != ' ' && ch != '\n') k.push_back(static_cast<char>(ch));
  if (ch != ' ' || static_cast<int>(k.size()) + 1 > key_cap) { fclose(f); return Fail("bad archive key", path); }
  memcpy(key, k.c_str(), k.size() + 1);
  std::vector<LatState> st;
  std::vector<kamd_lat_arc> out;
  int32_t start_state = -1;
  ch = fgetc(f);
  if (ch == 214) {
    ungetc(ch, f);
    Reader r(f);
    r.pos = 0;
    Header h;
    if (!ReadHeader(&r, &h) || h.fsttype != "vector" || h.arctype != "lattice4") { fclose(f); return Fail("binary lattice: expected vector FST of lattice4 arcs", path); }
    for (int64_t s = 0; h.numstates < 0 || s < h.numstates; s++) {
      LatState ls; ls.f1 = r.Get<float>(); ls.f2 = r.Get<float>();
      if (!r.ok && h.numstates < 0) { r.ok = true; break; }
      const int64_t n = r.Get<int64_t>();
      if (!r.ok || n < 0) { r.ok = false; break; }
      ls.is_final = !(ls.f1 == kInf && ls.f2 == kInf);
      st.push_back(ls);
      for (int64_t i = 0; i < n && r.ok; i++) {
        kamd_lat_arc a; a.src = static_cast<int32_t>(s);
        a.ilabel = r.Get<int32_t>(); a.olabel = r.Get<int32_t>();
        a.graph_cost = r.Get<float>(); a.acoustic_cost = r.Get<float>(); a.dst = r.Get<int32_t>();
        out.push_back(a);
      }
    }
    if (!r.ok) { fclose(f); return Fail("truncated binary lattice", path); }
    start_state = static_cast<int32_t>(h.start);
  } else {
    // text: the rest of the "key" line is empty; FstPrinter lines follow; empty line ends
    std::string line;
    if (ch != '\n') { fclose(f); return Fail("text lattice: newline expected after the key", path); }
    size_t nline = 0;
    while (true) {
      line.clear();
      while ((ch = fgetc(f)) != EOF && ch != '\n') line.push_back(static_cast<char>(ch));
      std::vector<std::string> col;
      std::istringstream ls(line);
      std::string tok;
      while (ls >> tok) col.push_back(tok);
      if (col.empty()) break;
      nline++;
      auto geti = [](const std::string &s, int32_t *v) { char *e; long x = strtol(s.c_str(), &e, 10); *v = static_cast<int32_t>(x); return e != s.c_str() && *e == '\0'; };
      auto getw = [&](const std::string &s, float *a, float *b) {
        size_t c = s.find(',');
        return c != std::string::npos && GetFloat(s.substr(0, c), a) && GetFloat(s.substr(c + 1), b);
      };
      int32_t s = 0, d = 0;
      bool ok = geti(col[0], &s) && s >= 0;
      if (ok) {
        while (static_cast<size_t>(s) >= st.size()) { LatState z; z.f1 = z.f2 = kInf; z.is_final = false; st.push_back(z); }
        if (nline == 1) start_state = s;       // kaldi-lattice.cc:146-149
        if (col.size() == 1) { st[s].f1 = st[s].f2 = 0.0f; st[s].is_final = true; }
        else if (col.size() == 2) { ok = getw(col[1], &st[s].f1, &st[s].f2); st[s].is_final = !(st[s].f1 == kInf && st[s].f2 == kInf); }
        else if (col.size() == 4 || col.size() == 5) {
          kamd_lat_arc a; a.src = s; a.graph_cost = 0.0f; a.acoustic_cost = 0.0f;
          ok = geti(col[1], &d) && d >= 0 && geti(col[2], &a.ilabel) && geti(col[3], &a.olabel);
          if (ok && col.size() == 5) ok = getw(col[4], &a.graph_cost, &a.acoustic_cost);
          a.dst = d;
          if (ok) {
            while (static_cast<size_t>(d) >= st.size()) { LatState z; z.f1 = z.f2 = kInf; z.is_final = false; st.push_back(z); }
            out.push_back(a);
          }
        } else ok = false;
      }
      if (!ok) { fclose(f); return kamd::SetError(KAMD_ERR_ARG, "%s: bad line in lattice text format: %s", path, line.c_str()); }
      if (ch == EOF) break;
    }
  }
  *offset = ftell(f);
  fclose(f);
  // arcs grouped by source state, file order inside a state
  std::vector<int64_t> cnt(st.size() + 1, 0);
  for (size_t i = 0; i < out.size(); i++) cnt[out[i].src + 1]++;
  for (size_t s = 0; s < st.size(); s++) cnt[s + 1] += cnt[s];
  std::vector<kamd_lat_arc> sorted(out.size());
  { std::vector<int64_t> p(cnt.begin(), cnt.end() - 1); for (size_t i = 0; i < out.size(); i++) sorted[p[out[i].src]++] = out[i]; }
  *num_states = static_cast<int32_t>(st.size()); *num_arcs = static_cast<int32_t>(out.size()); *start = start_state;
  *state_final = static_cast<float *>(malloc(sizeof(float) * 2 * (st.size() + 1)));
  *arcs = static_cast<kamd_lat_arc *>(malloc(sizeof(kamd_lat_arc) * (out.size() + 1)));
  if (!*state_final || !*arcs) return kamd::SetError(KAMD_ERR_ARG, "out of host memory");
  for (size_t s = 0; s < st.size(); s++) {
    (*state_final)[2 * s] = st[s].is_final ? st[s].f1 : kInf;
    (*state_final)[2 * s + 1] = st[s].is_final ? st[s].f2 : kInf;
  }
  if (!sorted.empty()) memcpy(*arcs, sorted.data(), sizeof(kamd_lat_arc) * sorted.size());
  return KAMD_OK;
}

}  // extern "C"
