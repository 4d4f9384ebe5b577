// ivector_io.cc -- the files of an i-vector extraction config -> kamd_ivector_desc (host C++; Python twin: kaldi_amd/ivector.py).
//
// What OnlineIvectorExtractionInfo::Init reads (online2/online-ivector-feature.cc:30-74) for the --ivector-extraction-config
// of the online2 binaries / the --config of ivector-extract-online2 (options: OnlineIvectorExtractionConfig::Register,
// online2/online-ivector-feature.h:90-137): final.mat (LDA, Matrix<float>), global_cmvn.stats (Matrix<double>), the
// OnlineCmvnOptions and OnlineSpliceOptions config files, final.dubm (DiagGmm::Read, gmm/diag-gmm.cc:766-810; the
// stored gconsts are recomputed, ComputeGconsts :114-150) and final.ie (IvectorExtractor::Read, ivector/ivector-
// extractor.cc:462-500).  Binary objects only.  With this a C / C++ host gets the recipe's online i-vectors without Python:
// kamd_ivector_info_read -> kamd_ivector_info_create_extractor -> kamd_batch_decoder_set_ivector_extractor.
#include <cmath>
#include <cstdio>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "common.h"

namespace kamd {
namespace {

struct IoError { std::string msg; };
[[noreturn]] void Fail(const std::string &m) { throw IoError{m}; }

// "--name=value" lines, '#' comments (ParseOptions::ReadConfigFile, util/parse-options.cc:500-560)
std::map<std::string, std::string> ReadConfig(const std::string &path) {
  FILE *f = fopen(path.c_str(), "r");
  if (!f) Fail("Cannot open config file: " + path);
  std::map<std::string, std::string> kv;
  char buf[8192];
  while (fgets(buf, sizeof(buf), f)) {
    std::string line(buf);
    const size_t h = line.find('#');
    if (h != std::string::npos) line.erase(h);
    while (!line.empty() && isspace(static_cast<unsigned char>(line.back()))) line.pop_back();
    size_t b = 0;
    while (b < line.size() && isspace(static_cast<unsigned char>(line[b]))) b++;
    line = line.substr(b);
    if (line.empty()) continue;
    if (line.compare(0, 2, "--") != 0) { fclose(f); Fail("Reading config file " + path + ": line does not look like --x=y: " + line); }
    const size_t eq = line.find('=');
    std::string k = line.substr(2, eq == std::string::npos ? std::string::npos : eq - 2), v = eq == std::string::npos ? "true" : line.substr(eq + 1);
    for (char &c : k) if (c == '_') c = '-';
    kv[k] = v;
  }
  fclose(f);
  return kv;
}
std::string Need(const std::map<std::string, std::string> &kv, const char *name) {
  std::map<std::string, std::string>::const_iterator it = kv.find(name);
  if (it == kv.end() || it->second.empty())
    Fail(std::string("--") + name + " option must be set (note: this may be needed in the file supplied to --ivector-extractor-config)");
  return it->second;
}
template <typename T> T Opt(const std::map<std::string, std::string> &kv, const char *name, T def);
template <> int Opt<int>(const std::map<std::string, std::string> &kv, const char *name, int def) {
  std::map<std::string, std::string>::const_iterator it = kv.find(name);
  return it == kv.end() ? def : atoi(it->second.c_str());
}
template <> double Opt<double>(const std::map<std::string, std::string> &kv, const char *name, double def) {
  std::map<std::string, std::string>::const_iterator it = kv.find(name);
  return it == kv.end() ? def : atof(it->second.c_str());
}
template <> bool Opt<bool>(const std::map<std::string, std::string> &kv, const char *name, bool def) {
  std::map<std::string, std::string>::const_iterator it = kv.find(name);
  if (it == kv.end()) return def;
  return it->second == "true" || it->second == "t" || it->second == "1" || it->second.empty();
}
void OnlyKnown(const std::map<std::string, std::string> &kv, const std::vector<const char *> &known, const std::string &file) {
  for (const std::pair<const std::string, std::string> &e : kv) {
    bool ok = false;
    for (const char *k : known) ok = ok || e.first == k;
    if (!ok) Fail("Invalid option --" + e.first + " in config file " + file);
  }
}

// a binary Kaldi object behind an rxfilename
struct In {
  std::vector<unsigned char> b; size_t p = 0; std::string name;
  explicit In(const std::string &rxfilename) : name(rxfilename) {
    char path[4096]; int temp = 0; int64_t off = 0;
    if (kamd_rx_materialize(rxfilename.c_str(), path, sizeof(path), &off, &temp) != KAMD_OK) Fail(kamd_last_error());
    FILE *f = fopen(path, "rb");
    if (!f) Fail("cannot open " + rxfilename);
    if (off) fseek(f, static_cast<long>(off), SEEK_SET);
    unsigned char tmp[1 << 16]; size_t n;
    while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) b.insert(b.end(), tmp, tmp + n);
    fclose(f);
    if (temp) remove(path);
    if (b.size() < 2 || b[0] != 0 || b[1] != 'B') Fail(rxfilename + ": only binary-mode Kaldi objects are read");
    p = 2;
  }
  const unsigned char *Take(size_t n) { if (p + n > b.size()) Fail(name + ": unexpected end of file"); const unsigned char *r = &b[p]; p += n; return r; }
  std::string Token() {
    size_t e = p;
    while (e < b.size() && b[e] != ' ') e++;
    if (e >= b.size()) Fail(name + ": unexpected end of file inside a token");
    std::string t(reinterpret_cast<const char *>(&b[p]), e - p);
    p = e + 1;
    return t;
  }
  std::string Expect(const char *a, const char *alt = NULL) {
    const std::string t = Token();
    if (t != a && !(alt && t == alt)) Fail(name + ": expected " + a + (alt ? std::string(" or ") + alt : "") + ", got " + t);
    return t;
  }
  int32_t I32() { if (*Take(1) != 4) Fail(name + ": int32 expected"); int32_t v; memcpy(&v, Take(4), 4); return v; }
  double Real() { const int n = *Take(1); if (n == 4) { float f; memcpy(&f, Take(4), 4); return f; } if (n != 8) Fail(name + ": real expected"); double d; memcpy(&d, Take(8), 8); return d; }
  template <typename T> void Data(bool dbl, size_t n, std::vector<T> *out) {
    out->resize(n);
    const unsigned char *q = Take((dbl ? 8 : 4) * n);
    for (size_t i = 0; i < n; i++) {
      if (dbl) { double d; memcpy(&d, q + 8 * i, 8); (*out)[i] = static_cast<T>(d); }
      else { float f; memcpy(&f, q + 4 * i, 4); (*out)[i] = static_cast<T>(f); }
    }
  }
  template <typename T> void Vector(std::vector<T> *out) { const std::string t = Expect("FV", "DV"); const int32_t n = I32(); if (n < 0) Fail(name + ": bad size"); Data(t == "DV", n, out); }
  template <typename T> void Matrix(std::vector<T> *out, int *rows, int *cols) {
    const std::string t = Expect("FM", "DM");
    *rows = I32(); *cols = I32();
    if (*rows < 0 || *cols < 0) Fail(name + ": bad size");
    Data(t == "DM", static_cast<size_t>(*rows) * *cols, out);
  }
  template <typename T> void Packed(std::vector<T> *out, int *n) {
    const std::string t = Expect("FP", "DP");
    *n = I32();
    if (*n < 0) Fail(name + ": bad size");
    Data(t == "DP", static_cast<size_t>(*n) * (*n + 1) / 2, out);
  }
};

}  // namespace

struct IvInfo {
  kamd_ivector_desc d;
  std::vector<float> lda, gconsts, miv, iv, weights;
  std::vector<double> cmvn, M, sinv;
};

namespace {

void ReadInfo(const char *config, IvInfo *x) {
  const std::map<std::string, std::string> po = ReadConfig(config);
  OnlyKnown(po, {"lda-matrix", "global-cmvn-stats", "cmvn-config", "splice-config", "diag-ubm", "ivector-extractor", "ivector-period", "num-gselect",
                 "min-post", "posterior-scale", "max-count", "use-most-recent-ivector", "greedy-ivector-extractor", "max-remembered-frames",
                 "num-cg-iters"}, config);
  const std::string cmvn_conf = Need(po, "cmvn-config"), splice_conf = Need(po, "splice-config");
  const std::map<std::string, std::string> cm = ReadConfig(cmvn_conf), sp = ReadConfig(splice_conf);
  OnlyKnown(cm, {"cmn-window", "global-frames", "speaker-frames", "norm-vars", "norm-means", "skip-dims", "modulus", "ring-buffer-size"}, cmvn_conf);
  OnlyKnown(sp, {"left-context", "right-context"}, splice_conf);
  if (cm.count("skip-dims") && !cm.find("skip-dims")->second.empty()) Fail("--skip-dims is not supported");
  kamd_ivector_desc &d = x->d;
  memset(&d, 0, sizeof(d));
  d.splice_left = Opt<int>(sp, "left-context", 4); d.splice_right = Opt<int>(sp, "right-context", 4);
  d.cmn_window = Opt<int>(cm, "cmn-window", 600); d.global_frames = Opt<int>(cm, "global-frames", 200); d.speaker_frames = Opt<int>(cm, "speaker-frames", 600);
  d.normalize_mean = Opt<bool>(cm, "norm-means", true) ? 1 : 0; d.normalize_variance = Opt<bool>(cm, "norm-vars", false) ? 1 : 0;
  d.ivector_period = Opt<int>(po, "ivector-period", 10); d.num_gselect = Opt<int>(po, "num-gselect", 5); d.num_cg_iters = Opt<int>(po, "num-cg-iters", 15);
  d.min_post = static_cast<float>(Opt<double>(po, "min-post", 0.025)); d.posterior_scale = static_cast<float>(Opt<double>(po, "posterior-scale", 0.1));
  d.max_count = static_cast<float>(Opt<double>(po, "max-count", 0.0));
  int r, c;
  { In s(Need(po, "lda-matrix")); s.Matrix(&x->lda, &r, &c); d.lda_rows = r; d.lda_cols = c; }
  { In s(Need(po, "global-cmvn-stats")); s.Matrix(&x->cmvn, &r, &c); if (r != 2) Fail("global CMVN stats must have two rows"); d.feat_dim = c - 1; }
  const int D = d.lda_rows;
  {                                                         // DiagGmm::Read (gmm/diag-gmm.cc:766-810)
    In s(Need(po, "diag-ubm"));
    s.Expect("<DiagGMM>", "<DiagGMMBegin>");
    std::string t = s.Token();
    if (t == "<GCONSTS>") { std::vector<float> skip; s.Vector(&skip); t = s.Token(); }
    if (t != "<WEIGHTS>") Fail("DiagGmm::Read, expected <WEIGHTS> or <GCONSTS>, got " + t);
    s.Vector(&x->weights);
    s.Expect("<MEANS_INVVARS>"); s.Matrix(&x->miv, &r, &c);
    if (r != static_cast<int>(x->weights.size()) || c != D) Fail("diagonal UBM does not match the LDA output dimension");
    s.Expect("<INV_VARS>"); s.Matrix(&x->iv, &r, &c);
    if (r != static_cast<int>(x->weights.size()) || c != D) Fail("diagonal UBM does not match the LDA output dimension");
    s.Expect("</DiagGMM>", "<DiagGMMEnd>");
  }
  const int G = static_cast<int>(x->weights.size());
  d.num_gauss = G;
  // DiagGmm::ComputeGconsts (gmm/diag-gmm.cc:114-150), the float operation sequence of kaldi_amd/ivector.py (logs through
  // libm in double, rounded: bit-identical with the Python twin)
  x->gconsts.resize(G);
  const float offset = static_cast<float>(-0.5 * 1.8378770664093453 * D);
  for (int g = 0; g < G; g++) {
    float gc = static_cast<float>(log(static_cast<double>(x->weights[g]))) + offset;
    for (int k = 0; k < D; k++) {
      const float iv = x->iv[static_cast<size_t>(g) * D + k], mi = x->miv[static_cast<size_t>(g) * D + k];
      const float a = 0.5f * static_cast<float>(log(static_cast<double>(iv)));
      const float b = 0.5f * mi * mi / iv;
      gc = (gc + a) - b;
    }
    if (std::isinf(gc) && gc > 0) gc = -gc;
    x->gconsts[g] = gc;
  }
  {                                                         // IvectorExtractor::Read (ivector/ivector-extractor.cc:462-500)
    In s(Need(po, "ivector-extractor"));
    s.Expect("<IvectorExtractor>");
    s.Expect("<w>");
    std::vector<double> w; s.Matrix(&w, &r, &c);
    if (!w.empty()) Fail("i-vector dependent weights are not supported by the online extractor");
    s.Expect("<w_vec>"); { std::vector<double> skip; s.Vector(&skip); }
    s.Expect("<M>");
    const int g_m = s.I32();
    if (g_m != G) Fail("i-vector extractor does not match the UBM");
    int I = -1;
    for (int g = 0; g < G; g++) {
      std::vector<double> m; s.Matrix(&m, &r, &c);
      if (r != D || (I >= 0 && c != I)) Fail("i-vector extractor does not match the UBM");
      I = c;
      x->M.insert(x->M.end(), m.begin(), m.end());
    }
    d.ivector_dim = I;
    s.Expect("<SigmaInv>");
    for (int g = 0; g < G; g++) {
      std::vector<double> pk; int n; s.Packed(&pk, &n);
      if (n != D) Fail("i-vector extractor does not match the UBM");
      x->sinv.insert(x->sinv.end(), pk.begin(), pk.end());
    }
    s.Expect("<IvectorOffset>");
    d.prior_offset = s.Real();
    s.Expect("</IvectorExtractor>");
  }
  // OnlineIvectorExtractionInfo::Check (online2/online-ivector-feature.cc:76-93) + what the device path needs
  const int sd = d.feat_dim * (d.splice_left + 1 + d.splice_right);
  if (d.lda_cols != sd && d.lda_cols != sd + 1) Fail("LDA matrix has " + std::to_string(d.lda_cols) + " columns, spliced features " + std::to_string(sd));
  if (!(d.ivector_period > 0 && d.num_gselect > 0 && d.min_post < 0.5f && d.posterior_scale > 0.0f && d.posterior_scale <= 1.0f)) Fail("bad i-vector extraction options");
  if (!(d.speaker_frames <= d.cmn_window && d.global_frames <= d.speaker_frames)) Fail("OnlineCmvnOptions::Check failed");
  d.lda = x->lda.data(); d.global_cmvn_stats = x->cmvn.data();
  d.ubm_gconsts = x->gconsts.data(); d.ubm_means_invvars = x->miv.data(); d.ubm_inv_vars = x->iv.data();
  d.M = x->M.data(); d.sigma_inv = x->sinv.data();
}

}  // namespace
}  // namespace kamd

extern "C" {

kamd_ivector_info *kamd_ivector_info_read(const char *config_rxfilename) {
  if (!config_rxfilename) { kamd::SetError(KAMD_ERR_ARG, "kamd_ivector_info_read: no config file"); return NULL; }
  std::unique_ptr<kamd::IvInfo> x(new kamd::IvInfo());
  try {
    kamd::ReadInfo(config_rxfilename, x.get());
  } catch (const kamd::IoError &e) {
    kamd::SetError(KAMD_ERR_ARG, "%s", e.msg.c_str());
    return NULL;
  } catch (const std::exception &e) {
    kamd::SetError(KAMD_ERR_ARG, "%s: %s", config_rxfilename, e.what());
    return NULL;
  }
  return reinterpret_cast<kamd_ivector_info *>(x.release());
}
void kamd_ivector_info_destroy(kamd_ivector_info *h) { delete reinterpret_cast<kamd::IvInfo *>(h); }
const kamd_ivector_desc *kamd_ivector_info_desc(const kamd_ivector_info *h) { return &reinterpret_cast<const kamd::IvInfo *>(h)->d; }
kamd_ivector_extractor *kamd_ivector_info_create_extractor(const kamd_ivector_info *h) {
  return kamd_ivector_extractor_create(&reinterpret_cast<const kamd::IvInfo *>(h)->d);
}

}  // extern "C"
