// mdl.cc -- binary Kaldi final.mdl -> fused layers for the device (host C++; the Python twin is kaldi_amd/mdl.py).
//
// final.mdl = TransitionModel (hmm/transition-model.cc:394-420, hmm/hmm-topology.cc:129-206) + AmNnetSimple
// (nnet3/am-nnet-simple.cc:44-54) = Nnet (nnet3/nnet-nnet.cc:586-628: config lines, then the components) + priors.
// What comes out is what the device path consumes: kamd_layer_desc[] (what CollapseModel, nnet3/nnet-utils.cc:2006, and
// the decodable's "-log prior, * acoustic scale", nnet3/nnet-am-decodable-simple.cc:268-271, amount to at test time),
// id2pdf (TransitionModel::id2pdf_id_), tid_phone (the table DeterminizeLatticePhonePruned reads) and tid2phone.
//
// The component bodies go through a generic tokenizer of Kaldi's binary stream (tokens, size-prefixed basic types,
// FM / FV / DM / DV objects; base/io-funcs-inl.h, matrix/kaldi-matrix.cc:1378-1404), so fields this reader does not use
// are skipped by structure.  The graph is compiled DESCRIPTOR-driven (no layer names): every affine-like component
// (Affine, NaturalGradientAffine, FixedAffine, Linear, Tdnn) opens a fused layer whose input is x, Offset(x, t),
// Scale(s, x) or an Append of such slices of ONE producer (+ ReplaceIndex(ivector, t, 0)); what follows folds in: ReLU,
// BatchNorm (test mode), per-element maps, Sum(Scale(s, a), b) bypasses, dropout / no-op (identity), LogSoftmax.
// Pinned by the reference's own xconfig generator: tests/test_xconfig_golden.py compares both readers on
// tests/golden/nnet/*.final.config.
#include <cmath>
#include <cstdio>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "common.h"

namespace kamd {
namespace {

struct MdlError { std::string msg; };
[[noreturn]] void Fail(const std::string &m) { throw MdlError{m}; }

struct Value {                 // one field value of a component body
  enum Kind { kRaw, kBool, kVector, kMatrix, kIntVector } kind = kRaw;
  unsigned char raw[8] = {0}; int raw_len = 0;
  bool b = false;
  int rows = 0, cols = 0;
  std::vector<float> data;     // vector / matrix (doubles converted)
  std::vector<int32_t> ints;
  float AsFloat() const {
    if (kind != kRaw) Fail("number expected");
    if (raw_len == 4) { float f; memcpy(&f, raw, 4); return f; }
    double d; memcpy(&d, raw, 8); return static_cast<float>(d);
  }
};
typedef std::map<std::string, std::vector<Value> > Fields;

struct Stream {
  const std::vector<unsigned char> &b; size_t p = 0;
  explicit Stream(const std::vector<unsigned char> &buf) : b(buf) {}
  int Peek(size_t k = 0) const { return p + k < b.size() ? b[p + k] : -1; }
  const unsigned char *Take(size_t n) { if (p + n > b.size()) Fail("unexpected end of file"); const unsigned char *r = &b[p]; p += n; return r; }
  std::string Token() {
    size_t e = p;
    while (e < b.size() && b[e] != ' ') e++;
    if (e >= b.size()) Fail("unexpected end of file inside a token");
    std::string t(reinterpret_cast<const char *>(&b[p]), e - p);
    p = e + 1;
    return t;
  }
  void Expect(const char *tok) { const std::string t = Token(); if (t != tok) Fail(std::string("expected ") + tok + ", got " + t); }
  int32_t I32() { if (*Take(1) != 4) Fail("int32 expected"); int32_t v; memcpy(&v, Take(4), 4); return v; }
  float F32() { if (*Take(1) != 4) Fail("float expected"); float v; memcpy(&v, Take(4), 4); return v; }
  // a count read from the file, checked against what the file can still hold (min_bytes per element) BEFORE anything is
  // allocated for it: a corrupt header must fail here, not in a multi-gigabyte resize
  size_t Count(int64_t n, size_t min_bytes, const char *what) {
    if (n < 0) Fail(std::string("negative ") + what);
    if (static_cast<uint64_t>(n) > (b.size() - p) / (min_bytes ? min_bytes : 1)) Fail(std::string("unexpected end of file: ") + what + " exceeds what is left of it");
    return static_cast<size_t>(n);
  }
  std::vector<int32_t> IntVector() {              // WriteIntegerVector (base/io-funcs-inl.h:198-229)
    if (*Take(1) != 4) Fail("integer vector expected");
    int32_t n; memcpy(&n, Take(4), 4);
    Count(n, 4, "integer vector size");
    std::vector<int32_t> v(n);
    if (n) memcpy(v.data(), Take(4 * static_cast<size_t>(n)), 4 * static_cast<size_t>(n));
    return v;
  }
  void Floats(bool dbl, size_t n, std::vector<float> *out) {
    Count(static_cast<int64_t>(n), dbl ? 8 : 4, "matrix / vector size");
    out->resize(n);
    if (!dbl) { if (n) memcpy(out->data(), Take(4 * n), 4 * n); return; }
    const unsigned char *q = Take(8 * n);
    for (size_t i = 0; i < n; i++) { double d; memcpy(&d, q + 8 * i, 8); (*out)[i] = static_cast<float>(d); }
  }
  Value Vector() {                                // Vector<float>::Write: "FV " size data
    const std::string t = Token();
    if (t != "FV" && t != "DV") Fail("vector expected, got " + t);
    Value v; v.kind = Value::kVector;
    const int32_t n = I32();
    if (n < 0) Fail("negative vector size");
    v.rows = n; v.cols = 1;
    Floats(t == "DV", n, &v.data);
    return v;
  }
  Value Matrix() {
    const std::string t = Token();
    if (t != "FM" && t != "DM") Fail("matrix expected, got " + t);
    Value v; v.kind = Value::kMatrix;
    v.rows = I32(); v.cols = I32();
    if (v.rows < 0 || v.cols < 0) Fail("negative matrix size");
    if (v.cols > 0 && static_cast<uint64_t>(v.rows) > b.size() / static_cast<uint64_t>(v.cols)) Fail("unexpected end of file: matrix size exceeds what is left of it");
    Floats(t == "DM", static_cast<size_t>(v.rows) * v.cols, &v.data);
    return v;
  }
  // {token: [values]} until end_token: raw basic types, bools, vectors, matrices
  Fields GenericFields(const std::string &end_token) {
    Fields out;
    std::vector<Value> *cur = NULL;
    for (;;) {
      const int c = Peek();
      if (c == '<') {
        const std::string t = Token();
        if (t == end_token) return out;
        cur = &out[t];
        if (t == "<TimeOffsets>") { Value v; v.kind = Value::kIntVector; v.ints = IntVector(); cur->push_back(v); }
      } else if ((c == 'F' || c == 'D') && (Peek(1) == 'M' || Peek(1) == 'V') && Peek(2) == ' ') {
        if (!cur) Fail("value before the first field token");
        cur->push_back(Peek(1) == 'M' ? Matrix() : Vector());
      } else if (c == 'T' || c == 'F') {
        if (!cur) Fail("value before the first field token");
        Value v; v.kind = Value::kBool; v.b = *Take(1) == 'T'; cur->push_back(v);
      } else if (c == 4 || c == 8 || c == 1 || c == 2) {
        if (!cur) Fail("value before the first field token");
        Value v; v.raw_len = *Take(1); memcpy(v.raw, Take(v.raw_len), v.raw_len); cur->push_back(v);
      } else {
        Fail("cannot parse component body at byte " + std::to_string(p));
      }
    }
  }
};

// ---- descriptors: ('node', name) | Offset(d, n) | Append(..) | Sum(a, b) | Scale(s, d) | ReplaceIndex(name, t, 0)
struct Desc {
  enum Kind { kNode, kOffset, kAppend, kSum, kScale, kReplaceIndex } kind = kNode;
  std::string name; int offset = 0; double scale = 1.0;
  std::vector<Desc> args;
};
struct DescParser {
  std::vector<std::string> toks; size_t pos = 0;
  explicit DescParser(const std::string &text) {
    for (size_t i = 0; i < text.size();) {
      const char c = text[i];
      if (isspace(static_cast<unsigned char>(c))) { i++; continue; }
      if (c == '(' || c == ')' || c == ',') { toks.push_back(std::string(1, c)); i++; continue; }
      size_t j = i;
      while (j < text.size() && !isspace(static_cast<unsigned char>(text[j])) && text[j] != '(' && text[j] != ')' && text[j] != ',') j++;
      toks.push_back(text.substr(i, j - i));
      i = j;
    }
  }
  Desc Parse() {
    if (pos >= toks.size()) Fail("descriptor ends early");
    const std::string t = toks[pos++];
    if (pos < toks.size() && toks[pos] == "(") {
      pos++;
      std::vector<Desc> args;
      while (pos < toks.size() && toks[pos] != ")") {
        if (toks[pos] == ",") { pos++; continue; }
        args.push_back(Parse());
      }
      if (pos >= toks.size()) Fail("descriptor: missing )");
      pos++;
      Desc d;
      if (t == "Offset" && args.size() >= 2) { d.kind = Desc::kOffset; d.offset = atoi(args[1].name.c_str()); d.args.push_back(args[0]); }
      else if (t == "Append") { d.kind = Desc::kAppend; d.args = args; }
      else if (t == "Sum" && args.size() == 2) { d.kind = Desc::kSum; d.args = args; }
      else if (t == "Scale" && args.size() == 2) { d.kind = Desc::kScale; d.scale = atof(args[0].name.c_str()); d.args.push_back(args[1]); }
      else if (t == "ReplaceIndex" && !args.empty()) { d.kind = Desc::kReplaceIndex; d.name = args[0].name; }
      else Fail("unsupported descriptor " + t);
      return d;
    }
    Desc d; d.kind = Desc::kNode; d.name = t;
    return d;
  }
};

struct Layer {
  std::string name;
  int in_dim = 0, out_dim = 0, input_layer = -1, bypass_layer = -2, ivector_dim = 0;
  std::vector<int> offsets;
  std::vector<int> slice_layers, slice_dims;      // Append over different producers (kamd_layer_desc::multi_input); empty otherwise
  std::vector<float> W, bias, bn_scale, bn_offset, post_offset;
  bool has_bias = false, relu = false, has_bn = false, log_softmax = false, has_post = false;
  float bypass_scale = 0.f, post_scale = 1.f;
};

}  // namespace

struct Model {
  std::vector<Layer> layers;
  std::vector<kamd_layer_desc> descs;
  int input_dim = 0, ivector_dim = 0, subsampling = 3, num_pdfs = 0;
  std::vector<int32_t> id2pdf, tid_phone, tid2phone, phones;
};

namespace {

// TransitionModel::Read + ComputeDerived (hmm/transition-model.cc:144-188, 394-420; hmm-topology.cc binary form :129-206)
void ReadTransitionModel(Stream *s, Model *m) {
  s->Expect("<TransitionModel>");
  s->Expect("<Topology>");
  m->phones = s->IntVector();
  const std::vector<int32_t> phone2idx = s->IntVector();
  int32_t n = s->I32();
  bool is_hmm = true;
  if (n == -1) { is_hmm = false; n = s->I32(); }
  struct St { int fwd, slf; std::vector<std::pair<int, float> > trans; };
  std::vector<std::vector<St> > entries(s->Count(n, 5, "topology entry count"));      // (an entry is at least its state count: 5 bytes)
  for (int e = 0; e < n; e++) {
    const int ns = s->I32();
    entries[e].resize(s->Count(ns, 10, "HMM state count"));                             // (a state: pdf class + transition count)
    for (int k = 0; k < ns; k++) {
      St &st = entries[e][k];
      st.fwd = s->I32();
      st.slf = is_hmm ? st.fwd : s->I32();
      const int nt = static_cast<int>(s->Count(s->I32(), 10, "transition count"));
      for (int t = 0; t < nt; t++) { const int dst = s->I32(); const float pr = s->F32(); st.trans.push_back(std::make_pair(dst, pr)); }
    }
  }
  s->Expect("</Topology>");
  const std::string tok = s->Token();
  if (tok != "<Triples>" && tok != "<Tuples>") Fail("expected <Triples> or <Tuples>, got " + tok);
  struct Tup { int ph, hs, fp, sp; };
  std::vector<Tup> tuples(s->Count(s->I32(), 15, "tuple count"));
  for (Tup &t : tuples) { t.ph = s->I32(); t.hs = s->I32(); t.fp = s->I32(); t.sp = tok == "<Triples>" ? t.fp : s->I32(); }
  s->Expect(tok == "<Triples>" ? "</Triples>" : "</Tuples>");
  s->Expect("<LogProbs>"); s->Vector(); s->Expect("</LogProbs>");
  s->Expect("</TransitionModel>");
  m->id2pdf.assign(1, -1); m->tid_phone.assign(1, 0); m->tid2phone.assign(1, 0);
  for (const Tup &t : tuples) {
    if (t.ph < 0 || t.ph >= static_cast<int>(phone2idx.size()) || phone2idx[t.ph] < 0 || phone2idx[t.ph] >= n) Fail("transition model: phone without a topology entry");
    const std::vector<St> &e = entries[phone2idx[t.ph]];
    if (t.hs < 0 || t.hs >= static_cast<int>(e.size())) Fail("transition model: bad hmm-state");
    for (const std::pair<int, float> &tr : e[t.hs].trans) {
      const bool self_loop = tr.first == t.hs;                 // IsSelfLoop (:319-327)
      m->id2pdf.push_back(self_loop ? t.sp : t.fp);
      m->tid_phone.push_back((t.hs == 0 && !self_loop) ? t.ph : 0);
      m->tid2phone.push_back(t.ph);
    }
  }
}

const Value &Field(const Fields &f, const char *key, const std::string &comp) {
  Fields::const_iterator it = f.find(key);
  if (it == f.end() || it->second.empty()) Fail(comp + ": field " + key + " missing");
  return it->second[0];
}

std::string ConfigField(const std::string &line, const std::string &key) {
  size_t at = 0;
  const std::string k = key + "=";
  while ((at = line.find(k, at)) != std::string::npos) {
    if (at == 0 || line[at - 1] == ' ') {
      size_t e = at + k.size();
      while (e < line.size() && !isspace(static_cast<unsigned char>(line[e]))) e++;
      return line.substr(at + k.size(), e - at - k.size());
    }
    at += k.size();
  }
  return "";
}

struct Compiler {
  struct Node { std::string component; Desc desc; };
  struct Comp { std::string type; Fields f; };
  std::map<std::string, Node> nodes;
  std::map<std::string, int> inputs;
  std::map<std::string, Comp> comps;
  std::vector<Layer> layers;
  std::map<std::string, int> layer_of;      // node name -> index of the fused layer whose output it is

  int Resolve(const Desc &d) {
    if (d.kind != Desc::kNode) Fail("unsupported input descriptor");
    if (d.name == "input") return -1;
    return Build(d.name);
  }
  static void Full(const Value &v, int dim, const std::string &name, std::vector<float> *out) {
    const size_t n = v.data.size();
    if (n == 0 || dim % static_cast<int>(n)) Fail(name + ": parameter vector of " + std::to_string(n) + " elements for dimension " + std::to_string(dim));
    out->resize(dim);
    for (int i = 0; i < dim; i++) (*out)[i] = v.data[i % n];
  }
  // y = layer(x) * scale + offset folded into the fused layer
  void ApplyPerElement(Layer *L, const std::vector<float> &scale, const std::vector<float> &offset, const std::string &name) {
    if (L->bypass_layer != -2 || L->log_softmax) Fail("a per-element map after a bypass / log-softmax is not representable (" + name + ")");
    const int N = L->out_dim;
    if (!L->relu && !L->has_bn) {
      const int K = static_cast<int>(L->W.size()) / N;
      for (int n = 0; n < N; n++) for (int k = 0; k < K; k++) L->W[static_cast<size_t>(n) * K + k] *= scale[n];
      if (!L->has_bias) { L->bias.assign(N, 0.0f); L->has_bias = true; }
      for (int n = 0; n < N; n++) L->bias[n] = L->bias[n] * scale[n] + offset[n];
    } else if (!L->has_bn) {
      L->bn_scale = scale; L->bn_offset = offset; L->has_bn = true;
    } else {
      for (int n = 0; n < N; n++) { L->bn_offset[n] = L->bn_offset[n] * scale[n] + offset[n]; L->bn_scale[n] *= scale[n]; }
    }
  }
  int Build(const std::string &name) {
    std::map<std::string, int>::iterator known = layer_of.find(name);
    if (known != layer_of.end()) return known->second;
    std::map<std::string, Node>::iterator ni = nodes.find(name);
    if (ni == nodes.end()) Fail("node " + name + " is not defined");
    const std::string cname = ni->second.component;
    Desc desc = ni->second.desc;
    std::map<std::string, Comp>::iterator ci = comps.find(cname);
    if (ci == comps.end()) Fail("component " + cname + " is not defined");
    const std::string &typ = ci->second.type;
    const Fields &f = ci->second.f;
    if (typ == "FixedAffineComponent" || typ == "NaturalGradientAffineComponent" || typ == "AffineComponent" || typ == "LinearComponent" ||
        typ == "TdnnComponent") {
      const Value &Wv = Field(f, typ == "LinearComponent" ? "<Params>" : "<LinearParams>", cname);
      if (Wv.kind != Value::kMatrix) Fail(cname + ": parameter matrix expected");
      std::vector<float> bias;
      bool has_bias = false;
      if (typ != "LinearComponent") { const Value &bv = Field(f, "<BiasParams>", cname); bias = bv.data; has_bias = !bias.empty(); }
      int ivector_dim = 0;
      std::vector<double> col_scales;
      double whole_scale = 1.0;
      if (desc.kind == Desc::kScale) { whole_scale = desc.scale; const Desc inner = desc.args[0]; desc = inner; }
      std::vector<int> offsets, producers, slice_dims;
      bool multi = false;
      int src = -1;
      if (typ == "TdnnComponent") {
        const Value &tv = Field(f, "<TimeOffsets>", cname);
        offsets.assign(tv.ints.begin(), tv.ints.end());
        src = Resolve(desc);
      } else if (desc.kind == Desc::kAppend) {
        bool have_src = false;
        for (const Desc &part0 : desc.args) {
          if (part0.kind == Desc::kReplaceIndex) {
            if (!inputs.count(part0.name)) Fail("ReplaceIndex of an unknown input " + part0.name);
            ivector_dim = inputs[part0.name];
            continue;
          }
          Desc part = part0;
          double ps = 1.0;
          if (part.kind == Desc::kScale) { ps = part.scale; const Desc in = part.args[0]; part = in; }
          int off = 0;
          Desc inner = part;
          if (part.kind == Desc::kOffset) { off = part.offset; inner = part.args[0]; }
          if (inner.kind == Desc::kScale) { ps *= inner.scale; const Desc in = inner.args[0]; inner = in; }
          col_scales.push_back(ps);
          const int idx = Resolve(inner);
          if (have_src && idx != src) multi = true;
          src = idx; have_src = true;
          producers.push_back(idx);
          offsets.push_back(off);
        }
        if (multi) {                    // Append over different producers: a multi-input layer
          if (ivector_dim) Fail("Append over different producers together with an ivector is not supported (" + name + ")");
          for (int q : producers) slice_dims.push_back(q == -1 ? inputs["input"] : layers[q].out_dim);
          src = -1;
        }
      } else if (desc.kind == Desc::kOffset) {
        offsets.push_back(desc.offset); src = Resolve(desc.args[0]);
      } else {
        offsets.push_back(0); src = Resolve(desc);
      }
      int in_dim = src == -1 ? inputs["input"] : layers[src].out_dim;
      const int n_off = static_cast<int>(offsets.size());
      std::vector<int> widths(n_off, in_dim), cols(n_off + 1, 0);
      if (multi) widths = slice_dims;
      for (int o = 0; o < n_off; o++) cols[o + 1] = cols[o] + widths[o];
      if (multi) in_dim = cols[n_off];
      if (Wv.cols != cols[n_off] + ivector_dim) Fail(name + ": parameter shape does not match its input");
      Layer L;
      L.name = name; L.in_dim = in_dim; L.out_dim = Wv.rows; L.offsets = offsets; L.input_layer = src; L.ivector_dim = ivector_dim;
      if (multi) { L.slice_layers = producers; L.slice_dims = slice_dims; }
      L.W = Wv.data;
      bool scaled = whole_scale != 1.0;
      for (double c : col_scales) scaled = scaled || c != 1.0;
      if (scaled) {
        for (int o = 0; o < n_off; o++) {
          const float cs = static_cast<float>(col_scales.empty() ? whole_scale : col_scales[o]) * (col_scales.empty() ? 1.0f : static_cast<float>(whole_scale));
          for (int n = 0; n < L.out_dim; n++)
            for (int k = 0; k < widths[o]; k++) L.W[static_cast<size_t>(n) * Wv.cols + cols[o] + k] *= cs;
        }
        if (whole_scale != 1.0 && ivector_dim > 0)          // (round-2 advisor: the i-vector columns of a whole-input Scale)
          for (int n = 0; n < L.out_dim; n++)
            for (int k = 0; k < ivector_dim; k++) L.W[static_cast<size_t>(n) * Wv.cols + cols[n_off] + k] *= static_cast<float>(whole_scale);
      }
      if (has_bias) { L.bias = bias; L.has_bias = true; }
      layers.push_back(L);
      layer_of[name] = static_cast<int>(layers.size()) - 1;
      return layer_of[name];
    }
    // components that act on the output of the layer they follow
    if (typ == "NoOpComponent" && desc.kind == Desc::kSum) {       // tdnnf bypass: Sum(Scale(s, prev), this)
      Desc a = desc.args[0], b = desc.args[1];
      Desc scaled = a.kind == Desc::kScale ? a : b, plain = a.kind == Desc::kScale ? b : a;
      double sc;
      Desc byp_desc;
      if (scaled.kind != Desc::kScale) {                           // Sum(x, y): residual with scale 1; the later layer is "this"
        const int ia = Resolve(a), ib = Resolve(b);
        plain = ia > ib ? a : b; byp_desc = ia > ib ? b : a; sc = 1.0;
      } else { sc = scaled.scale; byp_desc = scaled.args[0]; }
      const int idx = Resolve(plain), byp = Resolve(byp_desc);
      Layer &L = layers[idx];
      if (L.bypass_layer != -2) Fail("two bypass connections on one layer (" + name + ")");
      L.bypass_layer = byp; L.bypass_scale = static_cast<float>(sc);
      layer_of[name] = idx;
      return idx;
    }
    const int idx = Resolve(desc);
    Layer &L = layers[idx];
    if (typ == "RectifiedLinearComponent") {
      if (L.relu || L.has_bn || L.bypass_layer != -2) Fail("ReLU after batchnorm / bypass is not representable (" + name + ")");
      L.relu = true;
    } else if (typ == "BatchNormComponent") {
      if (L.bypass_layer != -2) Fail("batchnorm after a bypass is not representable (" + name + ")");
      const Value &mean = Field(f, "<StatsMean>", cname), &var = Field(f, "<StatsVar>", cname);
      const float eps = Field(f, "<Epsilon>", cname).AsFloat(), rms = Field(f, "<TargetRms>", cname).AsFloat();
      const int N = L.out_dim;
      if (static_cast<int>(mean.data.size()) != N || static_cast<int>(var.data.size()) != N) Fail(cname + ": statistics of another dimension");
      std::vector<float> scale(N), offset(N);
      for (int n = 0; n < N; n++) {                              // ComputeDerived (nnet-normalize-component.cc:226-245), in float
        scale[n] = static_cast<float>(pow(static_cast<double>(std::max(var.data[n], 0.0f) + eps), -0.5)) * rms;   // double pow, rounded: what
                                                                                                                   // the Python twin computes too
        offset[n] = -mean.data[n] * scale[n];
      }
      if (L.has_bn) for (int n = 0; n < N; n++) { L.bn_offset[n] = L.bn_offset[n] * scale[n] + offset[n]; L.bn_scale[n] *= scale[n]; }
      else { L.bn_scale = scale; L.bn_offset = offset; L.has_bn = true; }
    } else if (typ == "GeneralDropoutComponent" || typ == "DropoutComponent" || typ == "NoOpComponent" || typ == "ClipGradientComponent" ||
               typ == "BackpropTruncationComponent") {
      // identity at test time
    } else if (typ == "FixedScaleComponent" || typ == "FixedBiasComponent" || typ == "PerElementScaleComponent" ||
               typ == "NaturalGradientPerElementScaleComponent" || typ == "PerElementOffsetComponent" || typ == "ScaleAndOffsetComponent") {
      const int N = L.out_dim;
      std::vector<float> scale(N, 1.0f), offset(N, 0.0f);
      if (typ == "FixedScaleComponent") Full(Field(f, "<Scales>", cname), N, name, &scale);
      else if (typ == "FixedBiasComponent") Full(Field(f, "<Bias>", cname), N, name, &offset);
      else if (typ == "PerElementOffsetComponent") Full(Field(f, "<Offsets>", cname), N, name, &offset);
      else if (typ == "ScaleAndOffsetComponent") { Full(Field(f, "<Scales>", cname), N, name, &scale); Full(Field(f, "<Offsets>", cname), N, name, &offset); }
      else Full(Field(f, "<Params>", cname), N, name, &scale);
      ApplyPerElement(&L, scale, offset, name);
    } else if (typ == "LogSoftmaxComponent") {
      L.log_softmax = true;
    } else {
      Fail("unsupported component type " + typ + " (" + name + ")");
    }
    layer_of[name] = idx;
    return idx;
  }
};

void ReadModel(const char *path, float acoustic_scale, int frame_subsampling_factor, Model *m) {
  FILE *fp = fopen(path, "rb");
  if (!fp) Fail(std::string("cannot open ") + path);
  std::vector<unsigned char> buf;
  {
    unsigned char tmp[1 << 16];
    size_t n;
    while ((n = fread(tmp, 1, sizeof(tmp), fp)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(fp);
  }
  Stream s(buf);
  if (buf.size() < 2 || buf[0] != 0 || buf[1] != 'B') Fail("binary Kaldi file expected (text-mode models: nnet3-am-copy --binary=true)");
  s.p = 2;
  ReadTransitionModel(&s, m);
  s.Expect("<Nnet3>");
  // config section: text lines until an empty line (nnet-nnet.cc:601-610)
  size_t end = s.p;
  while (end + 1 < buf.size() && !(buf[end] == '\n' && buf[end + 1] == '\n')) end++;
  if (end + 1 >= buf.size()) Fail("config section without an end");
  const std::string config(reinterpret_cast<const char *>(&buf[s.p]), end - s.p);
  s.p = end + 2;
  Compiler c;
  s.Expect("<NumComponents>");
  const int nc = s.I32();
  for (int i = 0; i < nc; i++) {
    s.Expect("<ComponentName>");
    const std::string name = s.Token();
    const std::string typ = s.Token();                      // "<TdnnComponent>"
    if (typ.size() < 3) Fail("bad component type token");
    Compiler::Comp comp;
    comp.type = typ.substr(1, typ.size() - 2);
    comp.f = s.GenericFields("</" + typ.substr(1));
    c.comps[name] = comp;
  }
  s.Expect("</Nnet3>");
  s.Expect("<LeftContext>"); s.I32();
  s.Expect("<RightContext>"); s.I32();
  s.Expect("<Priors>");
  const Value priors = s.Vector();
  // ---- graph: component-node name=X component=C input=DESCRIPTOR
  bool have_out = false;
  Desc out_desc;
  for (size_t at = 0; at < config.size();) {
    size_t e = config.find('\n', at);
    if (e == std::string::npos) e = config.size();
    std::string line = config.substr(at, e - at);
    at = e + 1;
    while (!line.empty() && isspace(static_cast<unsigned char>(line.back()))) line.pop_back();
    size_t b0 = 0;
    while (b0 < line.size() && isspace(static_cast<unsigned char>(line[b0]))) b0++;
    line = line.substr(b0);
    if (line.compare(0, 10, "input-node") == 0) {
      c.inputs[ConfigField(line, "name")] = atoi(ConfigField(line, "dim").c_str());
    } else if (line.compare(0, 14, "component-node") == 0) {
      const size_t ip = line.find("input=");
      if (ip == std::string::npos) Fail("component-node without input=: " + line);
      Compiler::Node nd;
      nd.component = ConfigField(line, "component");
      DescParser dp(line.substr(ip + 6));
      nd.desc = dp.Parse();
      c.nodes[ConfigField(line, "name")] = nd;
    } else if (line.compare(0, 11, "output-node") == 0 && ConfigField(line, "name") == "output") {
      const size_t ip = line.find("input=");
      if (ip == std::string::npos) Fail("output-node without input=");
      std::string d = line.substr(ip + 6);
      const size_t ob = d.find(" objective=");
      if (ob != std::string::npos) d = d.substr(0, ob);
      DescParser dp(d);
      out_desc = dp.Parse();
      have_out = true;
    }
  }
  if (!have_out || !c.inputs.count("input")) Fail("no output-node named 'output' / input-node named 'input'");
  const int out_idx = c.Resolve(out_desc);
  if (out_idx != static_cast<int>(c.layers.size()) - 1) Fail("the output node must be the last layer built");
  Layer &out = c.layers[out_idx];
  if (!priors.data.empty()) {                             // nnet-am-decodable-simple.cc:268-269
    if (static_cast<int>(priors.data.size()) != out.out_dim) Fail("priors of another dimension than the output");
    out.post_offset.resize(out.out_dim);
    for (int n = 0; n < out.out_dim; n++) out.post_offset[n] = static_cast<float>(-log(static_cast<double>(priors.data[n])));
    out.has_post = true;
  }
  out.post_scale = acoustic_scale;
  m->layers.swap(c.layers);
  m->input_dim = c.inputs["input"];
  m->ivector_dim = c.inputs.count("ivector") ? c.inputs["ivector"] : 0;
  m->subsampling = frame_subsampling_factor;
  m->num_pdfs = m->layers.back().out_dim;
  m->descs.resize(m->layers.size());
  for (size_t i = 0; i < m->layers.size(); i++) {
    const Layer &L = m->layers[i];
    kamd_layer_desc &d = m->descs[i];
    memset(&d, 0, sizeof(d));
    if (L.offsets.size() > KAMD_MAX_OFFSETS) Fail(L.name + ": more time offsets than KAMD_MAX_OFFSETS");
    d.in_dim = L.in_dim; d.out_dim = L.out_dim; d.n_offsets = static_cast<int32_t>(L.offsets.size());
    for (size_t o = 0; o < L.offsets.size(); o++) d.offsets[o] = L.offsets[o];
    d.input_layer = L.input_layer; d.ivector_dim = L.ivector_dim; d.bypass_layer = L.bypass_layer; d.bypass_scale = L.bypass_scale;
    d.relu = L.relu; d.log_softmax = L.log_softmax; d.post_scale = L.post_scale;
    d.W = L.W.data(); d.bias = L.has_bias ? L.bias.data() : NULL;
    d.bn_scale = L.has_bn ? L.bn_scale.data() : NULL; d.bn_offset = L.has_bn ? L.bn_offset.data() : NULL;
    d.post_offset = L.has_post ? L.post_offset.data() : NULL;
    if (!L.slice_layers.empty()) {
      d.multi_input = 1;
      for (size_t o = 0; o < L.slice_layers.size(); o++) { d.slice_layer[o] = L.slice_layers[o]; d.slice_dim[o] = L.slice_dims[o]; }
    }
  }
}

}  // namespace
}  // namespace kamd

extern "C" {

kamd_model *kamd_model_read(const char *path, float acoustic_scale, int frame_subsampling_factor) {
  if (!path || frame_subsampling_factor < 1) { kamd::SetError(KAMD_ERR_ARG, "kamd_model_read: bad arguments"); return NULL; }
  std::unique_ptr<kamd::Model> m(new kamd::Model());
  try {
    kamd::ReadModel(path, acoustic_scale, frame_subsampling_factor, m.get());
  } catch (const kamd::MdlError &e) {
    kamd::SetError(KAMD_ERR_ARG, "%s: %s", path, e.msg.c_str());
    return NULL;
  } catch (const std::exception &e) {
    kamd::SetError(KAMD_ERR_ARG, "%s: %s", path, e.what());
    return NULL;
  }
  return reinterpret_cast<kamd_model *>(m.release());
}

void kamd_model_destroy(kamd_model *h) { delete reinterpret_cast<kamd::Model *>(h); }

int kamd_model_info(const kamd_model *h, int32_t *num_layers, int32_t *input_dim, int32_t *ivector_dim, int32_t *num_pdfs, int32_t *num_tids,
                    int32_t *frame_subsampling_factor) {
  const kamd::Model *m = reinterpret_cast<const kamd::Model *>(h);
  if (num_layers) *num_layers = static_cast<int32_t>(m->layers.size());
  if (input_dim) *input_dim = m->input_dim;
  if (ivector_dim) *ivector_dim = m->ivector_dim;
  if (num_pdfs) *num_pdfs = m->num_pdfs;
  if (num_tids) *num_tids = static_cast<int32_t>(m->id2pdf.size()) - 1;
  if (frame_subsampling_factor) *frame_subsampling_factor = m->subsampling;
  return KAMD_OK;
}

const kamd_layer_desc *kamd_model_layers(const kamd_model *h) { return reinterpret_cast<const kamd::Model *>(h)->descs.data(); }

int kamd_model_transition_tables(const kamd_model *h, int32_t *id2pdf, int32_t *tid_phone, int32_t *tid2phone) {
  const kamd::Model *m = reinterpret_cast<const kamd::Model *>(h);
  const size_t n = m->id2pdf.size();
  if (id2pdf) memcpy(id2pdf, m->id2pdf.data(), n * 4);
  if (tid_phone) memcpy(tid_phone, m->tid_phone.data(), n * 4);
  if (tid2phone) memcpy(tid2phone, m->tid2phone.data(), n * 4);
  return KAMD_OK;
}

kamd_nnet *kamd_model_create_nnet(const kamd_model *h) {
  const kamd::Model *m = reinterpret_cast<const kamd::Model *>(h);
  return kamd_nnet_create(m->descs.data(), static_cast<int>(m->descs.size()), m->input_dim, m->subsampling);
}

}  // extern "C"
