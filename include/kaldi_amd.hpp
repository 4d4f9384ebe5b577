// kaldi_amd.hpp -- C++ host-side mirror of Kaldi's interfaces for the decode hot path,
// layered on the C-ABI (kaldi_amd.h).  Header only; link with -lkaldi_amd.
//
// Same class names, method names, argument meaning and error behaviour as the reference
// so that call sites read like Kaldi code:
//   kaldi_amd::DecodableInterface         itf/decodable-itf.h:82-118
//   kaldi_amd::LatticeFasterDecoderConfig decoder/lattice-faster-decoder.h:38-90
//   kaldi_amd::LatticeFasterDecoder       decoder/lattice-faster-decoder.h:226-343
//   kaldi_amd::DecodableMatrixMapped      decoder/decodable-matrix.h:98-136
//   kaldi_amd::MfccOptions / Mfcc         feat/feature-mfcc.h:38-56, feature-common.h:111
//   kaldi_amd::NnetBatchDecoder           nnet3/nnet-batch-compute.h:606-833
//   kaldi_amd::TransitionModelAndNnet     nnet3bin/nnet3-latgen-faster.cc:91-104 (what is read from <nnet-in>)
//   kaldi_amd::ConstArpaLm                lm/const-arpa-lm.h:211-352
// Errors: the reference's KALDI_ERR throws kaldi::KaldiFatalError (std::runtime_error,
// base/kaldi-error.h:89-140); here every non-zero C-ABI status throws
// kaldi_amd::KaldiFatalError carrying kamd_last_error().
// What is NOT here: fst::Fst / kaldi::Lattice types (OpenFst is not available in this
// image); INTEGRATION.md shows the ~40-line adapters a Kaldi tree adds on top.
#ifndef KALDI_AMD_HPP_
#define KALDI_AMD_HPP_

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <limits>
#include <map>
#include <stdexcept>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include <unistd.h>

#include "kaldi_amd.h"

namespace kaldi_amd {

typedef float BaseFloat;
typedef int32_t int32;

class KaldiFatalError : public std::runtime_error {
 public:
  explicit KaldiFatalError(const std::string &m) : std::runtime_error(m) {}
};
inline void Check(int rc) { if (rc < 0) throw KaldiFatalError(kamd_last_error()); }
template <typename T> inline T *CheckPtr(T *p) { if (!p) throw KaldiFatalError(kamd_last_error()); return p; }

// ------------------------------------------------------------------ itf/decodable-itf.h
class DecodableInterface {
 public:
  /// Returns the log likelihood, which will be negated in the decoder.  'index' is a
  /// 1-based transition-id; 'frame' is zero-based.
  virtual BaseFloat LogLikelihood(int32 frame, int32 index) = 0;
  virtual bool IsLastFrame(int32 frame) const = 0;
  virtual int32 NumFramesReady() const { throw KaldiFatalError("NumFramesReady() not implemented for this decodable type."); }
  virtual int32 NumIndices() const = 0;
  virtual ~DecodableInterface() {}
};

// decoder/decodable-matrix.h:98-136: log-likes matrix [frames x pdfs] + tid->pdf map.
// The matrix lives on the host here; the decoder uploads it once (or, for a matrix that is
// already in HBM, use SetDevicePointer and nothing is copied).
class DecodableMatrixMapped : public DecodableInterface {
 public:
  DecodableMatrixMapped(const std::vector<int32> &id2pdf /* [num_tids+1], [0] unused */,
                        const float *likes, int32 num_frames, int32 num_pdfs, int32 frame_offset = 0)
      : id2pdf_(id2pdf), likes_(likes), num_frames_(num_frames), num_pdfs_(num_pdfs),
        frame_offset_(frame_offset), d_likes_(NULL) {}
  BaseFloat LogLikelihood(int32 frame, int32 tid) override {
    return likes_[static_cast<size_t>(frame - frame_offset_) * num_pdfs_ + id2pdf_[tid]];
  }
  bool IsLastFrame(int32 frame) const override { return frame == NumFramesReady() - 1; }
  int32 NumFramesReady() const override { return frame_offset_ + num_frames_; }
  int32 NumIndices() const override { return static_cast<int32>(id2pdf_.size()) - 1; }
  // fast path used by LatticeFasterDecoder
  const float *HostMatrix() const { return likes_; }
  int32 NumPdfs() const { return num_pdfs_; }
  int32 FrameOffset() const { return frame_offset_; }
  void SetDevicePointer(const float *d) { d_likes_ = d; }
  const float *DevicePointer() const { return d_likes_; }
  const std::vector<int32> &Id2Pdf() const { return id2pdf_; }
 private:
  std::vector<int32> id2pdf_;
  const float *likes_;
  int32 num_frames_, num_pdfs_, frame_offset_;
  const float *d_likes_;
};

// DecodableMatrixScaledMapped (decoder/decodable-matrix.h:34-84): scale * likes(frame, pdf(tid)).
// Goes through the generic DecodableInterface path of AdvanceDecoding.
class DecodableMatrixScaledMapped : public DecodableInterface {
 public:
  DecodableMatrixScaledMapped(const std::vector<int32> &id2pdf, const float *likes, int32 num_frames, int32 num_pdfs,
                              BaseFloat scale)
      : id2pdf_(id2pdf), likes_(likes), num_frames_(num_frames), num_pdfs_(num_pdfs), scale_(scale) {}
  BaseFloat LogLikelihood(int32 frame, int32 tid) override { return scale_ * likes_[static_cast<size_t>(frame) * num_pdfs_ + id2pdf_[tid]]; }
  bool IsLastFrame(int32 frame) const override { return frame == num_frames_ - 1; }
  int32 NumFramesReady() const override { return num_frames_; }
  int32 NumIndices() const override { return static_cast<int32>(id2pdf_.size()) - 1; }
 private:
  const std::vector<int32> &id2pdf_;
  const float *likes_;
  int32 num_frames_, num_pdfs_;
  BaseFloat scale_;
};

// ------------------------------------------------- decoder/lattice-faster-decoder.h:38-90
struct LatticeFasterDecoderConfig {
  BaseFloat beam;
  int32 max_active;
  int32 min_active;
  BaseFloat lattice_beam;
  int32 prune_interval;
  bool determinize_lattice;   // kept for source compatibility; determinization is host side
  BaseFloat beam_delta;
  BaseFloat hash_ratio;
  BaseFloat prune_scale;
  LatticeFasterDecoderConfig()
      : beam(16.0), max_active(std::numeric_limits<int32>::max()), min_active(200), lattice_beam(10.0),
        prune_interval(25), determinize_lattice(true), beam_delta(0.5), hash_ratio(2.0), prune_scale(0.1) {}
  void Check() const {
    if (!(beam > 0.0 && max_active > 1 && lattice_beam > 0.0 && min_active <= max_active && prune_interval > 0 &&
          beam_delta > 0.0 && hash_ratio >= 1.0 && prune_scale > 0.0 && prune_scale < 1.0))
      throw KaldiFatalError("invalid LatticeFasterDecoderConfig");
  }
  kamd_decoder_config ToC() const {
    kamd_decoder_config c = {beam, max_active, min_active, lattice_beam, prune_interval, beam_delta, hash_ratio, prune_scale};
    return c;
  }
};

// Raw lattice as plain arrays (kaldi::Lattice = fst::VectorFst<LatticeArc>; lat/kaldi-lattice.h:44).
struct LatticeArc { int32 ilabel, olabel; BaseFloat graph_cost, acoustic_cost; int32 nextstate; };
struct Lattice {
  int32 start;
  std::vector<std::vector<LatticeArc> > arcs;   // per state
  std::vector<BaseFloat> final_graph_cost;      // LatticeWeight(final, 0); +inf = not final
  std::vector<int32> state_frame, state_hclg;   // annotations the reference discards
  int32 NumStates() const { return static_cast<int32>(arcs.size()); }
};

// HCLG handle (what the reference passes as `const fst::Fst<fst::StdArc> &`).
class DecodingGraph {
 public:
  DecodingGraph(int32 num_states, int32 start, const int64_t *arc_off, const kamd_arc *arcs, const float *final_cost)
      : g_(CheckPtr(kamd_graph_create(num_states, start, arc_off, arcs, final_cost))) {}
  /// ReadFstKaldiGeneric (fstext/kaldi-fst-io.cc:44-89): "vector" or "const" OpenFst file.
  explicit DecodingGraph(const std::string &rxfilename) : g_(CheckPtr(kamd_graph_read_openfst(rxfilename.c_str()))) {}
  ~DecodingGraph() { kamd_graph_destroy(g_); }
  kamd_graph *Handle() const { return g_; }
 private:
  DecodingGraph(const DecodingGraph &);
  kamd_graph *g_;
};

// -------------------------------------------- decoder/lattice-faster-decoder.h:226-343
class LatticeFasterDecoder {
 public:
  // The reference ctor takes (fst, config); the transition-id -> pdf table comes from the
  // decodable there (DecodableMatrixMapped holds the TransitionModel).  The device decoder
  // resolves it once per arc at construction, so it is a ctor argument here.
  LatticeFasterDecoder(const DecodingGraph &fst, const LatticeFasterDecoderConfig &config,
                       const std::vector<int32> &id2pdf, const kamd_decoder_sizes *sizes = NULL)
      : config_(config), id2pdf_(id2pdf), d_buf_(NULL), d_buf_bytes_(0) {
    config.Check();
    kamd_decoder_config c = config.ToC();
    kamd_decoder_sizes s;
    if (sizes) s = *sizes; else { kamd_decoder_sizes_default(&s); s.max_lanes = 1; }
    dec_ = CheckPtr(kamd_decoder_create(fst.Handle(), &c, &s, id2pdf.empty() ? NULL : id2pdf.data(),
                                        id2pdf.empty() ? 0 : static_cast<int32>(id2pdf.size()) - 1));
  }
  ~LatticeFasterDecoder() { if (d_buf_) kamd_free(d_buf_); kamd_decoder_destroy(dec_); }

  void SetOptions(const LatticeFasterDecoderConfig &config) {
    config.Check(); config_ = config; kamd_decoder_config c = config.ToC(); Check(kamd_decoder_set_options(dec_, &c));
  }
  const LatticeFasterDecoderConfig &GetOptions() const { return config_; }

  /// Decodes until there are no more frames left in the "decodable" object.
  bool Decode(DecodableInterface *decodable) {
    InitDecoding();
    AdvanceDecoding(decodable);
    FinalizeDecoding();
    kamd_lattice_size sz;
    return kamd_decoder_lattice_size(dec_, 0, &sz) == 0 && sz.num_states > 0;
  }
  void InitDecoding() {
    int32 lane = 0;
    Check(kamd_decoder_init(dec_, &lane, 1, NULL));
    Check(kamd_decoder_sync(dec_));
  }
  /// Decodes frames NumFramesDecoded() .. decodable->NumFramesReady()-1 (at most
  /// max_num_frames of them).
  void AdvanceDecoding(DecodableInterface *decodable, int32 max_num_frames = -1) {
    const int32 done = NumFramesDecoded(), ready = decodable->NumFramesReady();
    if (ready < done) throw KaldiFatalError("AdvanceDecoding: NumFramesReady() decreased");
    int32 n = ready - done;
    if (max_num_frames >= 0 && n > max_num_frames) n = max_num_frames;
    if (n == 0) return;
    kamd_decode_task task;
    task.lane = 0; task.n_frames = n; task.reserved = 0;
    DecodableMatrixMapped *mm = dynamic_cast<DecodableMatrixMapped *>(decodable);
    if (mm && mm->Id2Pdf() == id2pdf_) {
      const int32 P = mm->NumPdfs();
      const size_t row0 = static_cast<size_t>(done - mm->FrameOffset());
      if (mm->DevicePointer()) {
        task.d_loglikes = mm->DevicePointer() + row0 * P;
      } else {
        Upload(mm->HostMatrix() + row0 * P, static_cast<size_t>(n) * P);
        task.d_loglikes = static_cast<const float *>(d_buf_);
      }
      task.ld = P;
    } else {
      // any other DecodableInterface: materialise LogLikelihood(frame, tid) for every
      // transition-id (what the reference does lazily, one virtual call per arc,
      // lattice-faster-decoder.cc:767,794).  Needs a decoder built with an empty id2pdf.
      if (!id2pdf_.empty()) throw KaldiFatalError("generic DecodableInterface needs a decoder constructed with an empty id2pdf (columns = transition-ids)");
      const int32 nt = decodable->NumIndices();
      std::vector<float> m(static_cast<size_t>(n) * nt);
      for (int32 f = 0; f < n; f++)
        for (int32 t = 1; t <= nt; t++) m[static_cast<size_t>(f) * nt + (t - 1)] = decodable->LogLikelihood(done + f, t);
      Upload(m.data(), m.size());
      task.d_loglikes = static_cast<const float *>(d_buf_);
      task.ld = nt;
    }
    Check(kamd_decoder_advance(dec_, &task, 1, NULL));
    Check(kamd_decoder_sync(dec_));
  }
  void FinalizeDecoding() {
    int32 lane = 0;
    Check(kamd_decoder_finalize(dec_, &lane, 1, NULL));
    Check(kamd_decoder_sync(dec_));
  }
  /// PruneActiveTokens (lattice-faster-decoder.h:455; protected in the reference, which calls it every prune_interval frames
  /// from AdvanceDecoding): here the caller decides when, e.g. when the lane's arena fills up; results do not change.
  void PruneActiveTokens() {
    int32 lane = 0;
    Check(kamd_decoder_compact(dec_, &lane, 1, NULL));
    Check(kamd_decoder_sync(dec_));
  }
  BaseFloat FinalRelativeCost() const { return kamd_decoder_final_relative_cost(dec_, 0); }
  bool ReachedFinal() const { return FinalRelativeCost() != std::numeric_limits<BaseFloat>::infinity(); }
  int32 NumFramesDecoded() const { return kamd_decoder_num_frames_decoded(dec_, 0); }

  /// Raw state-level lattice (lattice-faster-decoder.cc:113-196).  After FinalizeDecoding(): the pruned lattice (use_final_probs
  /// must be true, :117-120).  Before it: every token and link the live decoder holds, final costs computed on the spot;
  /// use_final_probs == false makes every token of the last frame final with weight One.
  bool GetRawLattice(Lattice *ofst, bool use_final_probs = true) const {
    kamd_lattice_size sz;
    const bool finalized = kamd_decoder_lattice_size(dec_, 0, &sz) == 0;
    if (finalized && !use_final_probs) throw KaldiFatalError("You cannot call FinalizeDecoding() and then call GetRawLattice() with use_final_probs == false");
    if (!finalized) Check(kamd_decoder_live_lattice_size(dec_, 0, use_final_probs ? 1 : 0, &sz));
    ofst->arcs.assign(sz.num_states, std::vector<LatticeArc>());
    ofst->final_graph_cost.assign(sz.num_states, 0.f);
    ofst->state_frame.assign(sz.num_states, 0); ofst->state_hclg.assign(sz.num_states, 0);
    ofst->start = sz.start;
    if (sz.num_states == 0) return false;
    std::vector<float> cost(sz.num_states);
    std::vector<kamd_lat_arc> arcs(sz.num_arcs);
    if (finalized) Check(kamd_decoder_get_raw_lattice(dec_, 0, ofst->state_frame.data(), ofst->state_hclg.data(), cost.data(),
                                                     ofst->final_graph_cost.data(), arcs.data()));
    else Check(kamd_decoder_get_live_raw_lattice(dec_, 0, use_final_probs ? 1 : 0, ofst->state_frame.data(), ofst->state_hclg.data(),
                                                cost.data(), ofst->final_graph_cost.data(), arcs.data()));
    for (size_t i = 0; i < arcs.size(); i++) {
      LatticeArc a = {arcs[i].ilabel, arcs[i].olabel, arcs[i].graph_cost, arcs[i].acoustic_cost, arcs[i].dst};
      ofst->arcs[arcs[i].src].push_back(a);
    }
    return true;
  }
  /// LatticeFasterOnlineDecoderTpl::GetRawLatticePruned (decoder/lattice-faster-online-decoder.cc:168-265): the raw lattice
  /// restricted to the paths within `beam` of the best one.  Exact pruning (kamd_lattice_prune); the reference prunes with
  /// the extra costs its last periodic PruneActiveTokens left behind.
  bool GetRawLatticePruned(Lattice *ofst, bool use_final_probs, BaseFloat beam) const {
    Lattice full;
    if (!GetRawLattice(&full, use_final_probs)) return false;
    std::vector<float> fin;
    std::vector<kamd_lat_arc> arcs;
    const float inf = std::numeric_limits<float>::infinity();
    fin.assign(full.NumStates(), inf);
    for (int32 s = 0; s < full.NumStates(); s++) {
      fin[s] = full.final_graph_cost[s];
      for (size_t k = 0; k < full.arcs[s].size(); k++) {
        const LatticeArc &a = full.arcs[s][k];
        kamd_lat_arc f = {s, a.nextstate, a.ilabel, a.olabel, a.graph_cost, a.acoustic_cost};
        arcs.push_back(f);
      }
    }
    std::vector<int32> smap(full.NumStates());
    std::vector<uint8_t> keep(arcs.size() + 1);
    int32 n_out = 0, m_out = 0;
    Check(kamd_lattice_prune(full.NumStates(), full.start, fin.data(), arcs.data(), static_cast<int32>(arcs.size()), beam, smap.data(), keep.data(),
                             &n_out, &m_out));
    ofst->arcs.assign(n_out, std::vector<LatticeArc>());
    ofst->final_graph_cost.assign(n_out, inf); ofst->state_frame.assign(n_out, 0); ofst->state_hclg.assign(n_out, 0);
    ofst->start = n_out > 0 ? smap[full.start] : -1;
    for (int32 s = 0; s < full.NumStates(); s++)
      if (smap[s] >= 0) { ofst->final_graph_cost[smap[s]] = fin[s]; ofst->state_frame[smap[s]] = full.state_frame[s]; ofst->state_hclg[smap[s]] = full.state_hclg[s]; }
    for (size_t i = 0; i < arcs.size(); i++)
      if (keep[i]) {
        LatticeArc a = {arcs[i].ilabel, arcs[i].olabel, arcs[i].graph_cost, arcs[i].acoustic_cost, smap[arcs[i].dst]};
        ofst->arcs[smap[arcs[i].src]].push_back(a);
      }
    return n_out > 0;
  }
  /// Single best path: ShortestPath(raw lattice) + GetLinearSymbolSequence.
  bool GetBestPath(std::vector<int32> *alignment, std::vector<int32> *words, BaseFloat *graph_cost,
                   BaseFloat *acoustic_cost) const {
    kamd_lattice_size sz;
    Check(kamd_decoder_lattice_size(dec_, 0, &sz));
    std::vector<int32> ali(sz.num_arcs + 1), wrd(sz.num_arcs + 1);
    int na = 0, nw = 0;
    if (kamd_decoder_best_path(dec_, 0, ali.data(), static_cast<int>(ali.size()), &na, wrd.data(),
                               static_cast<int>(wrd.size()), &nw, graph_cost, acoustic_cost) != 0)
      return false;
    alignment->assign(ali.begin(), ali.begin() + na);
    words->assign(wrd.begin(), wrd.begin() + nw);
    return true;
  }
  kamd_decoder *Handle() const { return dec_; }

 private:
  void Upload(const float *h, size_t n) {
    if (n * sizeof(float) > d_buf_bytes_) {
      if (d_buf_) kamd_free(d_buf_);
      d_buf_ = CheckPtr(kamd_malloc(n * sizeof(float)));
      d_buf_bytes_ = n * sizeof(float);
    }
    Check(kamd_memcpy_h2d(d_buf_, h, n * sizeof(float)));
  }
  LatticeFasterDecoder(const LatticeFasterDecoder &);
  LatticeFasterDecoderConfig config_;
  std::vector<int32> id2pdf_;
  kamd_decoder *dec_;
  void *d_buf_;
  size_t d_buf_bytes_;
};

// ------------------------------------------ lat/kaldi-lattice.h, lat/determinize-lattice-pruned.h
/// Flat form of a Lattice for the C-ABI (arcs grouped by source state).
inline void FlattenLattice(const Lattice &lat, std::vector<float> *state_final, std::vector<kamd_lat_arc> *arcs) {
  const float inf = std::numeric_limits<float>::infinity();
  state_final->assign(2 * lat.arcs.size(), inf);
  arcs->clear();
  for (size_t s = 0; s < lat.arcs.size(); s++) {
    if (lat.final_graph_cost[s] != inf) { (*state_final)[2 * s] = lat.final_graph_cost[s]; (*state_final)[2 * s + 1] = 0.0f; }
    for (size_t k = 0; k < lat.arcs[s].size(); k++) {
      const LatticeArc &a = lat.arcs[s][k];
      kamd_lat_arc f = {static_cast<int32>(s), a.nextstate, a.ilabel, a.olabel, a.graph_cost, a.acoustic_cost};
      arcs->push_back(f);
    }
  }
}

/// kaldi::CompactLattice: owns the determinized lattice.
class CompactLattice {
 public:
  CompactLattice() : c_(NULL) {}
  ~CompactLattice() { Reset(NULL); }
  void Reset(kamd_compact_lattice *c) { if (c_) kamd_compact_lattice_destroy(c_); c_ = c; }
  int32 NumStates() const { int32 n = 0, m, k, st, ok; if (c_) kamd_compact_lattice_sizes(c_, &n, &m, &k, &st, &ok); return n; }
  int32 NumArcs() const { int32 n, m = 0, k, st, ok; if (c_) kamd_compact_lattice_sizes(c_, &n, &m, &k, &st, &ok); return m; }
  const kamd_compact_lattice *Handle() const { return c_; }
 private:
  CompactLattice(const CompactLattice &);
  kamd_compact_lattice *c_;
};

struct DeterminizeLatticePhonePrunedOptions {   // lat/determinize-lattice-pruned.h:214-245
  kamd_determinize_opts c;
  DeterminizeLatticePhonePrunedOptions() { kamd_determinize_opts_default(&c); }
};

/// DeterminizeLatticePhonePrunedWrapper (lat/determinize-lattice-pruned.cc:1484-1509).  The
/// TransitionModel argument is replaced by what the function reads from it: tid_phone[tid] =
/// TransitionIdToPhone(tid) if TransitionIdToHmmState(tid) == 0 && !IsSelfLoop(tid), else 0.
inline bool DeterminizeLatticePhonePrunedWrapper(const std::vector<int32> &tid_phone, const Lattice &ifst, double beam,
                                                 CompactLattice *ofst,
                                                 DeterminizeLatticePhonePrunedOptions opts = DeterminizeLatticePhonePrunedOptions()) {
  std::vector<float> fin;
  std::vector<kamd_lat_arc> arcs;
  FlattenLattice(ifst, &fin, &arcs);
  if (tid_phone.empty()) opts.c.phone_determinize = 0;
  kamd_compact_lattice *c = CheckPtr(kamd_lattice_determinize_phone_pruned(
      ifst.NumStates(), ifst.start, fin.data(), arcs.data(), static_cast<int32>(arcs.size()),
      tid_phone.empty() ? NULL : tid_phone.data(), static_cast<int32>(tid_phone.size()) - 1, beam, &opts.c));
  ofst->Reset(c);
  int32 n, m, k, st, ok;
  kamd_compact_lattice_sizes(c, &n, &m, &k, &st, &ok);
  return ok != 0;
}

/// TableWriter<LatticeHolder> / <CompactLatticeHolder> / Int32VectorWriter for the wspecifier
/// forms "ark:FILE" (binary) and "ark,t:FILE" (text) (util/kaldi-table.h:277-330).
class TableWriterBase {
 public:
  /// "ark[,t]:" + a file, "-" or "| command" (ClassifyWspecifier, util/kaldi-table.cc:115-222; WX_PIPE / WX_STDOUT
  /// of util/kaldi-io.cc:85-130: the entries are spooled to a temporary file that Close() / the destructor streams out)
  explicit TableWriterBase(const std::string &wspecifier) : first_(true), sink_(0) {
    char ark[4096], scp[16]; int opts = 0;
    const int t = kamd_classify_wspecifier(wspecifier.c_str(), ark, sizeof(ark), scp, sizeof(scp), &opts);
    if (t != 1) throw KaldiFatalError("unsupported wspecifier " + wspecifier + " (an \"ark:\" one is expected)");
    binary_ = (opts & KAMD_WSPEC_BINARY) != 0;
    const int wx = kamd_classify_wxfilename(ark);
    if (wx == KAMD_WX_FILE) { path_ = ark; return; }
    if (wx != KAMD_WX_STDOUT && wx != KAMD_WX_PIPE) throw KaldiFatalError("invalid output filename " + std::string(ark));
    sink_ = wx; target_ = ark;
    char tmpl[] = "/tmp/kamd_table_XXXXXX";
    const int fd = mkstemp(tmpl);
    if (fd < 0) throw KaldiFatalError("cannot create a temporary file for " + wspecifier);
    ::close(fd);
    path_ = tmpl;
  }
  virtual ~TableWriterBase() { try { Close(); } catch (...) {} }
  bool IsOpen() const { return !path_.empty(); }
  /// streams a spooled table to its pipe / stdout; false if the command fails (TableWriter::Close)
  bool Close() {
    if (!sink_) return true;
    const int kind = sink_;
    sink_ = 0;
    bool ok = true;
    if (first_) { FILE *e = fopen(path_.c_str(), "wb"); if (e) fclose(e); }       // nothing written: an empty stream
    FILE *in = fopen(path_.c_str(), "rb");
    FILE *out = kind == KAMD_WX_STDOUT ? stdout : popen(target_.substr(target_.find('|') + 1).c_str(), "w");
    if (!in || !out) ok = false;
    else {
      char buf[1 << 16]; size_t n;
      while ((n = fread(buf, 1, sizeof(buf), in)) > 0) if (fwrite(buf, 1, n, out) != n) { ok = false; break; }
    }
    if (in) fclose(in);
    if (out && kind == KAMD_WX_PIPE) ok = pclose(out) == 0 && ok; else if (out) fflush(out);
    remove(path_.c_str());
    if (!ok) throw KaldiFatalError("error closing the table " + target_);
    return ok;
  }
 protected:
  int Append() { const int a = first_ ? 0 : 1; first_ = false; return a; }
  std::string path_, target_;
  bool binary_, first_;
  int sink_;
};
class LatticeWriter : public TableWriterBase {
 public:
  explicit LatticeWriter(const std::string &w) : TableWriterBase(w) {}
  void Write(const std::string &key, const Lattice &lat) {
    std::vector<float> fin; std::vector<kamd_lat_arc> arcs;
    FlattenLattice(lat, &fin, &arcs);
    Check(kamd_lattice_write(path_.c_str(), Append(), key.c_str(), binary_, lat.NumStates(), lat.start, fin.data(), arcs.data(),
                             static_cast<int32>(arcs.size())));
  }
};
class CompactLatticeWriter : public TableWriterBase {
 public:
  explicit CompactLatticeWriter(const std::string &w) : TableWriterBase(w) {}
  /// acoustic_scale: ScaleLattice(AcousticLatticeScale(1/acoustic_scale)) applied on output
  void Write(const std::string &key, const CompactLattice &clat, BaseFloat acoustic_scale = 1.0f) {
    Check(kamd_compact_lattice_write(path_.c_str(), Append(), key.c_str(), binary_, clat.Handle(), acoustic_scale));
  }
};
class Int32VectorWriter : public TableWriterBase {   // BasicVectorHolder<int32> (util/kaldi-holder-inl.h)
 public:
  explicit Int32VectorWriter(const std::string &w) : TableWriterBase(w) {}
  void Write(const std::string &key, const std::vector<int32> &v) {
    FILE *f = fopen(path_.c_str(), Append() ? "ab" : "wb");
    if (!f) throw KaldiFatalError("cannot open " + path_);
    fputs(key.c_str(), f); fputc(' ', f);
    if (binary_) {                       // "\0B", WriteBasicType(count), WriteBasicType(element)... (kaldi-holder-inl.h:230-250)
      fputc('\0', f); fputc('B', f);
      const int32 n = static_cast<int32>(v.size());
      fputc(4, f); fwrite(&n, 4, 1, f);
      for (size_t i = 0; i < v.size(); i++) { fputc(4, f); fwrite(&v[i], 4, 1, f); }
    } else {
      for (size_t i = 0; i < v.size(); i++) fprintf(f, "%d ", v[i]);
      fputc('\n', f);
    }
    fclose(f);
  }
};

/// DecodeUtteranceLatticeFaster (decoder/decoder-wrappers.cc:201-296): decode, word-level
/// traceback, raw lattice, optional determinization, inverse acoustic scaling, write.
/// trans_model is replaced by tid_phone (see DeterminizeLatticePhonePrunedWrapper); writers
/// may be NULL.  Returns false (after a warning on stderr) where the reference does.
inline bool DecodeUtteranceLatticeFaster(LatticeFasterDecoder &decoder, DecodableInterface &decodable,
                                         const std::vector<int32> &tid_phone, const std::string &utt, double acoustic_scale,
                                         bool determinize, bool allow_partial, Int32VectorWriter *alignment_writer,
                                         Int32VectorWriter *words_writer, CompactLatticeWriter *compact_lattice_writer,
                                         LatticeWriter *lattice_writer, double *like_ptr) {
  if (!decoder.Decode(&decodable)) { fprintf(stderr, "WARNING Failed to decode utterance with id %s\n", utt.c_str()); return false; }
  if (!decoder.ReachedFinal()) {
    if (allow_partial) fprintf(stderr, "WARNING Outputting partial output for utterance %s since no final-state reached\n", utt.c_str());
    else { fprintf(stderr, "WARNING Not producing output for utterance %s since no final-state reached and --allow-partial=false.\n", utt.c_str()); return false; }
  }
  std::vector<int32> alignment, words;
  BaseFloat g = 0, a = 0;
  if (!decoder.GetBestPath(&alignment, &words, &g, &a)) throw KaldiFatalError("Failed to get traceback for utterance " + utt);
  const int32 num_frames = static_cast<int32>(alignment.size());
  if (words_writer) words_writer->Write(utt, words);
  if (alignment_writer) alignment_writer->Write(utt, alignment);
  const double likelihood = -(static_cast<double>(g) + a);
  Lattice lat;
  if (!decoder.GetRawLattice(&lat) || lat.NumStates() == 0) throw KaldiFatalError("Unexpected problem getting lattice for utterance " + utt);
  // fst::Connect(&lat): the exactly pruned raw lattice is connected by construction
  if (determinize) {
    CompactLattice clat;
    if (!DeterminizeLatticePhonePrunedWrapper(tid_phone, lat, decoder.GetOptions().lattice_beam, &clat))
      fprintf(stderr, "WARNING Determinization finished earlier than the beam for utterance %s\n", utt.c_str());
    if (compact_lattice_writer) compact_lattice_writer->Write(utt, clat, acoustic_scale != 0.0 ? static_cast<BaseFloat>(acoustic_scale) : 1.0f);
  } else if (lattice_writer) {
    if (acoustic_scale != 0.0 && acoustic_scale != 1.0)
      for (size_t s = 0; s < lat.arcs.size(); s++)
        for (size_t k = 0; k < lat.arcs[s].size(); k++) lat.arcs[s][k].acoustic_cost /= static_cast<BaseFloat>(acoustic_scale);
    lattice_writer->Write(utt, lat);
  }
  fprintf(stderr, "LOG Log-like per frame for utterance %s is %g over %d frames.\n", utt.c_str(), likelihood / num_frames, num_frames);
  if (like_ptr) *like_ptr = likelihood;
  return true;
}

// --------------------------------------------------------- feat/feature-mfcc.h:38-56
struct FrameExtractionOptions : kamd_frame_opts {};
struct MfccOptions {
  kamd_mfcc_opts c;
  MfccOptions() { kamd_mfcc_opts_default(&c); }   // NOTE dither defaults to 0 here (1.0 in Kaldi)
};
// OfflineFeatureTpl<MfccComputer> (feat/feature-common.h:111-160)
class Mfcc {
 public:
  explicit Mfcc(const MfccOptions &opts, BaseFloat vtln_warp = 1.0f) : f_(CheckPtr(kamd_mfcc_create(&opts.c, vtln_warp))) {}
  ~Mfcc() { kamd_feat_destroy(f_); }
  int32 Dim() const { return kamd_feat_dim(f_); }
  /// wave: int16-range samples; output row-major [num_frames x Dim()].
  void ComputeFeatures(const std::vector<float> &wave, BaseFloat sample_freq, std::vector<float> *output,
                       int32 *num_frames) {
    (void)sample_freq;
    const int T = kamd_feat_num_frames(f_, static_cast<int64_t>(wave.size()));
    output->assign(static_cast<size_t>(T) * Dim(), 0.f);
    if (T > 0) Check(kamd_feat_compute(f_, wave.data(), static_cast<int64_t>(wave.size()), output->data(), T));
    *num_frames = T;
  }
 private:
  Mfcc(const Mfcc &);
  kamd_feat *f_;
};

// OfflineFeatureTpl<FbankComputer> (feat/feature-fbank.h:39-98, feat/feature-common.h:111-160)
struct FbankOptions {
  kamd_fbank_opts c;
  FbankOptions() { kamd_fbank_opts_default(&c); }
};
class Fbank {
 public:
  explicit Fbank(const FbankOptions &opts, BaseFloat vtln_warp = 1.0f) : f_(CheckPtr(kamd_fbank_create(&opts.c, vtln_warp))) {}
  ~Fbank() { kamd_feat_destroy(f_); }
  int32 Dim() const { return kamd_feat_dim(f_); }
  void ComputeFeatures(const std::vector<float> &wave, BaseFloat sample_freq, std::vector<float> *output, int32 *num_frames) {
    (void)sample_freq;
    const int T = kamd_feat_num_frames(f_, static_cast<int64_t>(wave.size()));
    output->assign(static_cast<size_t>(T) * Dim(), 0.f);
    if (T > 0) Check(kamd_feat_compute(f_, wave.data(), static_cast<int64_t>(wave.size()), output->data(), T));
    *num_frames = T;
  }
 private:
  Fbank(const Fbank &);
  kamd_feat *f_;
};

// ------------------------------------------------ online2/online-nnet3-decoding.h:52-121
/// The acoustic model handle (nnet3::AmNnetSimple after CollapseModel): kamd_layer_desc list.
/// What the decode binaries read from <nnet-in> (nnet3bin/nnet3-latgen-faster.cc:91-104: TransitionModel, then
/// AmNnetSimple, then SetBatchnormTestMode / SetDropoutTestMode / CollapseModel): kamd_model_read.  The tables are the
/// uses the path makes of the TransitionModel (TransitionIdToPdf, the phone a transition-id enters, TransitionIdToPhone).
class TransitionModelAndNnet {
 public:
  TransitionModelAndNnet(const std::string &filename, BaseFloat acoustic_scale, int32 frame_subsampling_factor)
      : m_(CheckPtr(kamd_model_read(filename.c_str(), acoustic_scale, frame_subsampling_factor))) {
    int32 n = 0;
    kamd_model_info(m_, &num_layers_, &input_dim_, &ivector_dim_, &num_pdfs_, &n, &subsampling_);
    id2pdf_.resize(n + 1); tid_phone_.resize(n + 1); tid2phone_.resize(n + 1);
    kamd_model_transition_tables(m_, id2pdf_.data(), tid_phone_.data(), tid2phone_.data());
  }
  ~TransitionModelAndNnet() { kamd_model_destroy(m_); }
  TransitionModelAndNnet(const TransitionModelAndNnet &) = delete;
  TransitionModelAndNnet &operator=(const TransitionModelAndNnet &) = delete;
  const kamd_model *Handle() const { return m_; }
  const std::vector<int32> &Id2Pdf() const { return id2pdf_; }          // index 0 unused
  const std::vector<int32> &TidPhone() const { return tid_phone_; }
  const std::vector<int32> &Tid2Phone() const { return tid2phone_; }
  int32 NumTransitionIds() const { return static_cast<int32>(id2pdf_.size()) - 1; }
  int32 NumPdfs() const { return num_pdfs_; }
  int32 InputDim() const { return input_dim_; }
  int32 IvectorDim() const { return ivector_dim_; }
  int32 NumLayers() const { return num_layers_; }
 private:
  kamd_model *m_;
  int32 num_layers_ = 0, input_dim_ = 0, ivector_dim_ = 0, num_pdfs_ = 0, subsampling_ = 0;
  std::vector<int32> id2pdf_, tid_phone_, tid2phone_;
};

class AmNnetSimple {
 public:
  AmNnetSimple(const std::vector<kamd_layer_desc> &layers, int32 input_dim, int32 frame_subsampling_factor)
      : n_(CheckPtr(kamd_nnet_create(layers.data(), static_cast<int>(layers.size()), input_dim, frame_subsampling_factor))) {}
  explicit AmNnetSimple(const TransitionModelAndNnet &model) : n_(CheckPtr(kamd_model_create_nnet(model.Handle()))) {}
  ~AmNnetSimple() { kamd_nnet_destroy(n_); }
  kamd_nnet *Handle() const { return n_; }
  int32 OutputDim() const { return kamd_nnet_output_dim(n_); }
 private:
  AmNnetSimple(const AmNnetSimple &);
  kamd_nnet *n_;
};

/// SingleUtteranceNnet3DecoderTpl<fst::Fst<fst::StdArc>>: streaming decode of one utterance.
/// OnlineNnet2FeaturePipeline is reduced to its MFCC part (no ivector / pitch), the
/// DecodableNnetLoopedOnline to the rows kamd_nnet_forward_range serves.
// ---- online2/online-endpoint.h:113-200
struct OnlineEndpointRule {
  bool must_contain_nonsilence;
  BaseFloat min_trailing_silence, max_relative_cost, min_utterance_length;
  OnlineEndpointRule(bool must_contain_nonsilence = true, BaseFloat min_trailing_silence = 1.0,
                     BaseFloat max_relative_cost = std::numeric_limits<BaseFloat>::infinity(), BaseFloat min_utterance_length = 0.0)
      : must_contain_nonsilence(must_contain_nonsilence), min_trailing_silence(min_trailing_silence),
        max_relative_cost(max_relative_cost), min_utterance_length(min_utterance_length) {}
  template <typename Opts> void RegisterWithPrefix(const std::string &prefix, Opts *opts) {
    opts->Register(prefix + ".must-contain-nonsilence", &must_contain_nonsilence,
                   "If true, for this endpointing rule to apply there must be nonsilence in the best-path traceback.");
    opts->Register(prefix + ".min-trailing-silence", &min_trailing_silence,
                   "This endpointing rule requires duration of trailing silence (in seconds) to be >= this value.");
    opts->Register(prefix + ".max-relative-cost", &max_relative_cost,
                   "This endpointing rule requires relative-cost of final-states to be <= this value.");
    opts->Register(prefix + ".min-utterance-length", &min_utterance_length,
                   "This endpointing rule requires utterance-length (in seconds) to be >= this value.");
  }
};
struct OnlineEndpointConfig {
  std::string silence_phones;   ///< e.g. 1:2:3:4, colon separated list of phones
  OnlineEndpointRule rule1, rule2, rule3, rule4, rule5;
  OnlineEndpointConfig()
      : rule1(false, 5.0, std::numeric_limits<BaseFloat>::infinity(), 0.0), rule2(true, 0.5, 2.0, 0.0), rule3(true, 1.0, 8.0, 0.0),
        rule4(true, 2.0, std::numeric_limits<BaseFloat>::infinity(), 0.0), rule5(false, 0.0, std::numeric_limits<BaseFloat>::infinity(), 20.0) {}
  template <typename Opts> void Register(Opts *opts) {
    opts->Register("endpoint.silence-phones", &silence_phones, "List of phones that are considered to be silence phones by the endpointing code.");
    rule1.RegisterWithPrefix("endpoint.rule1", opts); rule2.RegisterWithPrefix("endpoint.rule2", opts);
    rule3.RegisterWithPrefix("endpoint.rule3", opts); rule4.RegisterWithPrefix("endpoint.rule4", opts);
    rule5.RegisterWithPrefix("endpoint.rule5", opts);
  }
  kamd_endpoint_config ToC() const {
    kamd_endpoint_config c;
    const OnlineEndpointRule *r[5] = {&rule1, &rule2, &rule3, &rule4, &rule5};
    for (int i = 0; i < 5; i++) {
      c.rule[i].must_contain_nonsilence = r[i]->must_contain_nonsilence ? 1 : 0;
      c.rule[i].min_trailing_silence = r[i]->min_trailing_silence;
      c.rule[i].max_relative_cost = r[i]->max_relative_cost;
      c.rule[i].min_utterance_length = r[i]->min_utterance_length;
    }
    return c;
  }
  /// SplitStringToIntegers(silence_phones, ":", false, ...) (online-endpoint.cc:75-77)
  std::vector<int32> SilencePhones() const {
    std::vector<int32> out;
    size_t pos = 0;
    while (pos <= silence_phones.size() && !silence_phones.empty()) {
      const size_t e = silence_phones.find(':', pos);
      const std::string tok = silence_phones.substr(pos, e == std::string::npos ? std::string::npos : e - pos);
      char *end = NULL;
      const long v = strtol(tok.c_str(), &end, 10);
      if (tok.empty() || *end != 0) throw KaldiFatalError("Bad --silence-phones option in endpointing config: " + silence_phones);
      out.push_back(static_cast<int32>(v));
      if (e == std::string::npos) break;
      pos = e + 1;
    }
    return out;
  }
};
/// EndpointDetected on plain numbers (online-endpoint.cc:46-68)
inline bool EndpointDetected(const OnlineEndpointConfig &config, int32 num_frames_decoded, int32 trailing_silence_frames,
                             BaseFloat frame_shift_in_seconds, BaseFloat final_relative_cost) {
  const kamd_endpoint_config c = config.ToC();
  const int rc = kamd_endpoint_detected(&c, num_frames_decoded, trailing_silence_frames, frame_shift_in_seconds, final_relative_cost);
  Check(rc);
  return rc == 1;
}
/// EndpointDetected(config, tmodel, frame_shift, decoder) for a set of un-finalized lanes of one decoder handle, one
/// device launch (online-endpoint.cc:105-121).  tid2phone[tid], tid = 1..size()-1, is TransitionIdToPhone.
inline void EndpointDetected(const OnlineEndpointConfig &config, const std::vector<int32> &tid2phone, BaseFloat frame_shift_in_seconds,
                             kamd_decoder *decoder, const std::vector<int32> &lanes, std::vector<int32> *detected,
                             std::vector<int32> *trailing_silence_frames = NULL) {
  const std::vector<int32> sil = config.SilencePhones();
  Check(kamd_decoder_set_silence_phones(decoder, tid2phone.data(), static_cast<int32>(tid2phone.size()) - 1, sil.data(), static_cast<int>(sil.size())));
  const kamd_endpoint_config c = config.ToC();
  detected->assign(lanes.size(), 0);
  if (trailing_silence_frames) trailing_silence_frames->assign(lanes.size(), 0);
  Check(kamd_decoder_endpoint_detected(decoder, &c, lanes.data(), static_cast<int>(lanes.size()), frame_shift_in_seconds, detected->data(),
                                       trailing_silence_frames ? trailing_silence_frames->data() : NULL));
}

class SingleUtteranceNnet3Decoder {
 public:
  SingleUtteranceNnet3Decoder(const LatticeFasterDecoderConfig &decoder_opts, const std::vector<int32> &id2pdf,
                              const AmNnetSimple &am_nnet, const DecodingGraph &fst, const MfccOptions &mfcc_opts,
                              const kamd_decoder_sizes *sizes = NULL)
      : nnet_(am_nnet.Handle()), feat_(CheckPtr(kamd_mfcc_create(&mfcc_opts.c, 1.0f))),
        online_(CheckPtr(kamd_online_feat_create(feat_))), decoder_(fst, decoder_opts, id2pdf, sizes), finished_(false) {
    frame_shift_ = mfcc_opts.c.frame.frame_shift_ms * 1.0e-3f;
    decoder_.InitDecoding();                          // online-nnet3-decoding.cc:40
  }
  ~SingleUtteranceNnet3Decoder() { if (d_ll_) kamd_free(d_ll_); kamd_online_feat_destroy(online_); kamd_feat_destroy(feat_); }
  void AcceptWaveform(BaseFloat sampling_rate, const std::vector<float> &waveform) {
    Check(kamd_online_feat_accept_waveform(online_, sampling_rate, waveform.data(), static_cast<int64_t>(waveform.size())));
  }
  void InputFinished() { Check(kamd_online_feat_input_finished(online_)); finished_ = true; }
  /// advances the decoding as far as we can (online-nnet3-decoding.cc:51-53)
  void AdvanceDecoding() {
    const int32 done = decoder_.NumFramesDecoded();
    const int32 feat_ready = kamd_online_feat_num_frames_ready(online_);
    const int32 n = kamd_nnet_num_frames_ready(nnet_, feat_ready, finished_) - done;
    if (n <= 0) return;
    const int32 P = kamd_nnet_output_dim(nnet_);
    const size_t need = static_cast<size_t>(n) * P * sizeof(float);
    if (need > ll_bytes_) {            // grow-only: hipMalloc / hipFree synchronise the device, not something for every chunk
      if (d_ll_) kamd_free(d_ll_);
      d_ll_ = NULL; ll_bytes_ = 0;
      d_ll_ = static_cast<float *>(CheckPtr(kamd_malloc(need + need / 2)));
      ll_bytes_ = need + need / 2;
    }
    int ld = 0;
    const float *d_feats = kamd_online_feat_device_frames(online_, &ld);
    int rc = kamd_nnet_forward_range(nnet_, d_feats, ld, feat_ready, finished_, done, n, d_ll_, P);
    if (rc == 0) {
      kamd_decode_task t = {0, n, d_ll_, P, 0};
      rc = kamd_decoder_advance(decoder_.Handle(), &t, 1, NULL);
      if (rc == 0) rc = kamd_decoder_sync(decoder_.Handle());
    }
    Check(rc);
  }
  void FinalizeDecoding() { decoder_.FinalizeDecoding(); }
  int32 NumFramesDecoded() const { return decoder_.NumFramesDecoded(); }
  /// GetBestPath(end_of_utterance, &best_path) (online-nnet3-decoding.cc:81-85): before
  /// FinalizeDecoding this is BestPathEnd + TraceBackBestPath of the online decoder.
  bool GetBestPath(bool end_of_utterance, std::vector<int32> *alignment, std::vector<int32> *words,
                   BaseFloat *graph_cost, BaseFloat *acoustic_cost) const {
    kamd_lattice_size sz;
    if (kamd_decoder_lattice_size(decoder_.Handle(), 0, &sz) == 0)      // finalized
      return decoder_.GetBestPath(alignment, words, graph_cost, acoustic_cost);
    const int cap = 4 * (decoder_.NumFramesDecoded() + 2) + 1024;
    std::vector<int32> ali(cap), wrd(cap);
    int na = 0, nw = 0;
    if (kamd_decoder_partial_best_path(decoder_.Handle(), 0, end_of_utterance ? 1 : 0, ali.data(), cap, &na, wrd.data(), cap,
                                       &nw, graph_cost, acoustic_cost) != 0)
      return false;
    alignment->assign(ali.begin(), ali.begin() + na);
    words->assign(wrd.begin(), wrd.begin() + nw);
    return true;
  }
  /// GetLattice(end_of_utterance, &clat) (online-nnet3-decoding.cc:66-79): the raw lattice -- of the live decoder when
  /// FinalizeDecoding() has not been called, final-probs only at the end of the utterance -- through
  /// DeterminizeLatticePhonePrunedWrapper at the decoder's lattice beam.  tid_phone as in that wrapper.
  void GetLattice(bool end_of_utterance, const std::vector<int32> &tid_phone, CompactLattice *clat,
                  DeterminizeLatticePhonePrunedOptions det_opts = DeterminizeLatticePhonePrunedOptions()) const {
    if (NumFramesDecoded() == 0) throw KaldiFatalError("You cannot get a lattice if you decoded no frames.");
    Lattice raw_lat;
    decoder_.GetRawLattice(&raw_lat, end_of_utterance);
    DeterminizeLatticePhonePrunedWrapper(tid_phone, raw_lat, decoder_.GetOptions().lattice_beam, clat, det_opts);
  }
  /// EndpointDetected(config) (online-nnet3-decoding.cc:88-95); the transition model's tid -> phone table is
  /// passed here instead of at construction
  bool EndpointDetected(const OnlineEndpointConfig &config, const std::vector<int32> &tid2phone) {
    std::vector<int32> det;
    kaldi_amd::EndpointDetected(config, tid2phone, frame_shift_ * kamd_nnet_frame_subsampling_factor(nnet_), decoder_.Handle(),
                            std::vector<int32>(1, 0), &det);
    return det[0] != 0;
  }
  const LatticeFasterDecoder &Decoder() const { return decoder_; }
 private:
  SingleUtteranceNnet3Decoder(const SingleUtteranceNnet3Decoder &);
  BaseFloat frame_shift_;
  kamd_nnet *nnet_;
  kamd_feat *feat_;
  kamd_online_feat *online_;
  LatticeFasterDecoder decoder_;
  bool finished_;
  float *d_ll_ = NULL;        // the chunk's log-likelihood rows (device)
  size_t ll_bytes_ = 0;
};

// ---- util/parse-options.h:36-260: the subset the decode binaries use (typed --name=value options, bare --flag
// for bools, name normalisation, --config=file, "--" ends the options, 1-based GetArg)
class ParseOptions {
 public:
  explicit ParseOptions(const char *usage) : usage_(usage) {}
  void Register(const std::string &name, bool *p, const std::string &doc) { Add(name, 'b', p, doc); }
  void Register(const std::string &name, int32 *p, const std::string &doc) { Add(name, 'i', p, doc); }
  void Register(const std::string &name, float *p, const std::string &doc) { Add(name, 'f', p, doc); }
  void Register(const std::string &name, double *p, const std::string &doc) { Add(name, 'd', p, doc); }
  void Register(const std::string &name, std::string *p, const std::string &doc) { Add(name, 's', p, doc); }
  int Read(int argc, const char *const argv[]) {
    for (int i = 1; i < argc; i++) {                       // first pass: config files
      if (strncmp(argv[i], "--", 2) != 0 || !strcmp(argv[i], "--")) break;
      std::string k, v; bool eq; Split(argv[i], &k, &v, &eq);
      if (k == "config") ReadConfigFile(v);
    }
    int i = 1; bool dd = false;
    for (; i < argc; i++) {
      if (strncmp(argv[i], "--", 2) != 0) break;
      if (!strcmp(argv[i], "--")) { i++; dd = true; break; }
      std::string k, v; bool eq; Split(argv[i], &k, &v, &eq);
      if (k != "config" && !Set(k, v, eq)) throw KaldiFatalError(std::string("Invalid option ") + argv[i]);
    }
    for (; i < argc; i++) {
      if (!strcmp(argv[i], "--") && !dd) dd = true; else args_.push_back(argv[i]);
    }
    return i;
  }
  int NumArgs() const { return static_cast<int>(args_.size()); }
  std::string GetArg(int i) const {
    if (i < 1 || i > NumArgs()) throw KaldiFatalError("ParseOptions::GetArg, invalid index");
    return args_[i - 1];
  }
  void PrintUsage() const {
    fprintf(stderr, "\n%s\nOptions:\n", usage_.c_str());
    for (size_t k = 0; k < names_.size(); k++) fprintf(stderr, "  --%-25s : %s\n", names_[k].c_str(), docs_[k].c_str());
  }
  void ReadConfigFile(const std::string &filename) {
    FILE *f = fopen(filename.c_str(), "r");
    if (!f) throw KaldiFatalError("Cannot open config file: " + filename);
    char buf[4096];
    while (fgets(buf, sizeof(buf), f)) {
      std::string line(buf);
      const size_t h = line.find('#');
      if (h != std::string::npos) line.erase(h);
      Trim(&line);
      if (line.empty()) continue;
      if (line.compare(0, 2, "--") != 0) { fclose(f); throw KaldiFatalError("Reading config file " + filename + ": line does not look like --x=y: " + line); }
      std::string k, v; bool eq; Split(line.c_str(), &k, &v, &eq);
      if (!Set(k, v, eq)) { fclose(f); throw KaldiFatalError("Invalid option " + line + " in config file " + filename); }
    }
    fclose(f);
  }
 private:
  static void Trim(std::string *s) {
    const size_t a = s->find_first_not_of(" \t\r\n"), b = s->find_last_not_of(" \t\r\n");
    *s = a == std::string::npos ? "" : s->substr(a, b - a + 1);
  }
  static std::string Normalize(const std::string &n) {
    std::string o;
    for (size_t i = 0; i < n.size(); i++) o += n[i] == '_' ? '-' : static_cast<char>(tolower(n[i]));
    return o;
  }
  static void Split(const char *arg, std::string *k, std::string *v, bool *eq) {
    const std::string a(arg);
    const size_t p = a.find('=');
    if (p == 2) throw KaldiFatalError("Invalid option (no key): " + a);
    *eq = p != std::string::npos;
    *k = Normalize(a.substr(2, *eq ? p - 2 : std::string::npos));
    *v = *eq ? a.substr(p + 1) : "";
    Trim(v);
  }
  void Add(const std::string &name, char t, void *p, const std::string &doc) {
    names_.push_back(Normalize(name)); types_.push_back(t); ptrs_.push_back(p); docs_.push_back(doc);
  }
  bool Set(const std::string &k, const std::string &v, bool eq) {
    for (size_t i = 0; i < names_.size(); i++) {
      if (names_[i] != k) continue;
      char *e = NULL;
      switch (types_[i]) {
        case 'b': {
          if (eq && v.empty()) throw KaldiFatalError("Invalid option --" + k + "=");
          std::string l;
          for (size_t c = 0; c < v.size(); c++) l += static_cast<char>(tolower(v[c]));
          if (l == "true" || l == "t" || l == "1" || l.empty()) *static_cast<bool *>(ptrs_[i]) = true;
          else if (l == "false" || l == "f" || l == "0") *static_cast<bool *>(ptrs_[i]) = false;
          else throw KaldiFatalError("Invalid format for boolean argument [expected true or false]: " + v);
          return true;
        }
        case 'i': { const long x = strtol(v.c_str(), &e, 10); if (v.empty() || *e) throw KaldiFatalError("Invalid integer option \"" + v + "\""); *static_cast<int32 *>(ptrs_[i]) = static_cast<int32>(x); return true; }
        case 'f': { const double x = strtod(v.c_str(), &e); if (v.empty() || *e) throw KaldiFatalError("Invalid floating-point option \"" + v + "\""); *static_cast<float *>(ptrs_[i]) = static_cast<float>(x); return true; }
        case 'd': { const double x = strtod(v.c_str(), &e); if (v.empty() || *e) throw KaldiFatalError("Invalid floating-point option \"" + v + "\""); *static_cast<double *>(ptrs_[i]) = x; return true; }
        default:
          if (!eq) throw KaldiFatalError("Invalid option --" + k + " (option format is --x=y).");
          *static_cast<std::string *>(ptrs_[i]) = v;
          return true;
      }
    }
    return false;
  }
  std::string usage_;
  std::vector<std::string> names_, docs_, args_;
  std::vector<char> types_;
  std::vector<void *> ptrs_;
};

// ---- SequentialBaseFloatMatrixReader (util/kaldi-table.h:278-360) over "ark:" and "scp:" rspecifiers with files,
// file:offset entries, input pipes and standard input (kamd_classify_rspecifier / kamd_rx_materialize / kamd_ark_read_matrix)
class SequentialBaseFloatMatrixReader {
 public:
  explicit SequentialBaseFloatMatrixReader(const std::string &rspecifier) : type_(0), off_(0), temp_(false), scp_pos_(0), done_(false), rows_(0), cols_(0) {
    char rx[4096]; int opts = 0;
    type_ = kamd_classify_rspecifier(rspecifier.c_str(), rx, sizeof(rx), &opts);
    if (type_ != 1 && type_ != 2) throw KaldiFatalError("invalid rspecifier " + rspecifier);
    if (type_ == 1) Materialize(rx, &path_, &off_, &temp_);
    else {
      std::string p; int64_t o; bool t;
      Materialize(rx, &p, &o, &t);
      FILE *f = fopen(p.c_str(), "r");
      if (!f) throw KaldiFatalError("cannot open script file " + p);
      char buf[8192];
      while (fgets(buf, sizeof(buf), f)) {
        std::string line(buf);
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        const size_t sp = line.find_first_of(" \t");
        if (line.empty() || sp == std::string::npos || sp == 0) { fclose(f); throw KaldiFatalError("Invalid line in script file: \"" + line + "\""); }
        scp_.push_back(std::make_pair(line.substr(0, sp), line.substr(line.find_first_not_of(" \t", sp))));
      }
      fclose(f);
      if (t) remove(p.c_str());
    }
    Next();
  }
  ~SequentialBaseFloatMatrixReader() { if (temp_) remove(path_.c_str()); }
  bool Done() const { return done_; }
  const std::string &Key() const { return key_; }
  const std::vector<float> &Value() const { return data_; }
  int32 NumRows() const { return rows_; }
  int32 NumCols() const { return cols_; }
  void Next() {
    float *p = NULL;
    if (type_ == 1) {
      char key[1024];
      const int rc = kamd_ark_read_matrix(path_.c_str(), &off_, key, sizeof(key), &rows_, &cols_, &p);
      if (rc == 1) { done_ = true; return; }
      Check(rc);
      key_ = key;
    } else {
      if (scp_pos_ >= scp_.size()) { done_ = true; return; }
      key_ = scp_[scp_pos_].first;
      std::string path; int64_t off; bool temp;
      Materialize(scp_[scp_pos_].second.c_str(), &path, &off, &temp);
      scp_pos_++;
      const int rc = kamd_ark_read_matrix(path.c_str(), &off, NULL, 0, &rows_, &cols_, &p);
      if (temp) remove(path.c_str());
      Check(rc);
    }
    data_.assign(p, p + static_cast<size_t>(rows_) * cols_);
    kamd_host_free(p);
  }
 private:
  static void Materialize(const char *rx, std::string *path, int64_t *off, bool *temp) {
    char buf[4096]; int t = 0;
    Check(kamd_rx_materialize(rx, buf, sizeof(buf), off, &t));
    *path = buf; *temp = t != 0;
  }
  int type_;
  std::string path_, key_;
  int64_t off_;
  bool temp_;
  std::vector<std::pair<std::string, std::string> > scp_;
  size_t scp_pos_;
  bool done_;
  int32 rows_, cols_;
  std::vector<float> data_;
};

// ---- RandomAccessBaseFloatVectorReader (util/kaldi-table.h:400-470) for the --ivectors option: the table is read
// once (a vector entry comes back from kamd_ark_read_matrix as one row)
class RandomAccessBaseFloatVectorReader {
 public:
  explicit RandomAccessBaseFloatVectorReader(const std::string &rspecifier) {
    for (SequentialBaseFloatMatrixReader r(rspecifier); !r.Done(); r.Next()) {
      if (r.NumRows() > 1) throw KaldiFatalError("a vector table is expected: " + rspecifier + " (key " + r.Key() + ")");
      map_[r.Key()] = r.Value();
    }
  }
  bool HasKey(const std::string &key) const { return map_.count(key) != 0; }
  const std::vector<float> &Value(const std::string &key) const {
    std::map<std::string, std::vector<float> >::const_iterator it = map_.find(key);
    if (it == map_.end()) throw KaldiFatalError("Value() called for a key that is not in the table: " + key);
    return it->second;
  }
 private:
  std::map<std::string, std::vector<float> > map_;
};

// ---- RandomAccessTokenReader (TokenHolder, util/kaldi-holder-inl.h:600-650) for --utt2spk: text lines "key token"
class RandomAccessTokenReader {
 public:
  explicit RandomAccessTokenReader(const std::string &rspecifier) {
    char rx[4096], buf[4096]; int opts = 0, temp = 0; int64_t off = 0;
    if (kamd_classify_rspecifier(rspecifier.c_str(), rx, sizeof(rx), &opts) != 1) throw KaldiFatalError("an \"ark:\" rspecifier is expected: " + rspecifier);
    Check(kamd_rx_materialize(rx, buf, sizeof(buf), &off, &temp));
    FILE *f = fopen(buf, "r");
    if (!f) throw KaldiFatalError(std::string("cannot open ") + buf);
    if (off) fseek(f, static_cast<long>(off), SEEK_SET);
    char line[8192];
    while (fgets(line, sizeof(line), f)) {
      char key[4096], tok[4096];
      const int n = sscanf(line, "%4095s %4095s", key, tok);
      if (n <= 0) continue;
      if (n != 2) { fclose(f); throw KaldiFatalError("bad line in the token table " + rspecifier + ": " + line); }
      map_[key] = tok;
    }
    fclose(f);
    if (temp) remove(buf);
  }
  bool HasKey(const std::string &key) const { return map_.count(key) != 0; }
  const std::string &Value(const std::string &key) const {
    std::map<std::string, std::string>::const_iterator it = map_.find(key);
    if (it == map_.end()) throw KaldiFatalError("Value() called for a key that is not in the table: " + key);
    return it->second;
  }
 private:
  std::map<std::string, std::string> map_;
};

// ---- nnet3/nnet-batch-compute.h:606-833 NnetBatchDecoder, as nnet3-latgen-faster-batch drives it
// (nnet3bin/nnet3-latgen-faster-batch.cc:170-214): AcceptInput per utterance, Finished(), GetOutput until false.
// Here the utterances of the set wait in host memory until Finished(), which runs the whole set through the device
// (kamd_batch_decoder_*: features -> acoustic model in a few large passes -> ONE work-queue launch of the search ->
// best path + determinization on num_threads host threads while the search is still running); outputs then come back
// in input order.  What differs from the reference's signature and why:
//   * `trans_model` -> the two tables that are read from it: id2pdf (TransitionIdToPdfFast) and tid_phone
//     (DeterminizeLatticePhonePrunedWrapper's use of it, see that function above; empty = word determinization only);
//   * `computer` (NnetBatchComputer) -> the acoustic model itself: the device batches whole utterances, there are no
//     minibatches of chunks to configure; `acoustic_scale` is what NnetSimpleComputationOptions carried;
//   * word_syms: a vector indexed by word id (fst::SymbolTable is OpenFst's);
//   * AcceptWaveform: waveform in, features on the device (the reference computes MFCCs in a process upstream).
struct NnetBatchDecoderOptions {
  kamd_batch_opts c;            // c.det = config.det_opts of the reference (DeterminizeLatticePhonePrunedOptions)
  BaseFloat acoustic_scale;     // output lattices get their acoustic costs divided by it (nnet-batch-compute.cc:1265-1267)
  int32 search_mode;            // kamd_decoder_set_search_mode
  NnetBatchDecoderOptions() : acoustic_scale(1.0f), search_mode(2) { kamd_batch_opts_default(&c); }
};

class NnetBatchDecoder {
 public:
  NnetBatchDecoder(const DecodingGraph &fst, const LatticeFasterDecoderConfig &decoder_config, const std::vector<int32> &id2pdf,
                   const std::vector<int32> &tid_phone, const std::vector<std::string> *word_syms, bool allow_partial, int32 num_threads,
                   const AmNnetSimple &am_nnet, const MfccOptions *mfcc_opts, const kamd_decoder_sizes &sizes,
                   NnetBatchDecoderOptions opts = NnetBatchDecoderOptions())
      : config_(decoder_config), word_syms_(word_syms), allow_partial_(allow_partial), opts_(opts), feat_(NULL), dec_(NULL), h_(NULL),
        next_out_(0), finished_(false), num_success_(0), num_fail_(0), num_partial_(0), tot_like_(0.0), frame_count_(0) {
    decoder_config.Check();
    kamd_decoder_config c = decoder_config.ToC();
    opts_.c.host_threads = num_threads;
    opts_.c.determinize = decoder_config.determinize_lattice ? 1 : 0;
    opts_.c.keep_raw_lattices = decoder_config.determinize_lattice ? 0 : 1;
    opts_.c.lattice_beam = decoder_config.lattice_beam;
    if (mfcc_opts) feat_ = CheckPtr(kamd_mfcc_create(&mfcc_opts->c, 1.0f));
    dec_ = kamd_decoder_create(fst.Handle(), &c, &sizes, id2pdf.empty() ? NULL : id2pdf.data(), id2pdf.empty() ? 0 : static_cast<int32>(id2pdf.size()) - 1);
    if (dec_ && kamd_decoder_set_search_mode(dec_, opts_.search_mode) == 0)
      h_ = kamd_batch_decoder_create(feat_, am_nnet.Handle(), dec_, &opts_.c, tid_phone.empty() ? NULL : tid_phone.data(),
                                     static_cast<int32>(tid_phone.size()) - 1);
    if (!h_) {
      const std::string why = kamd_last_error();
      if (dec_) kamd_decoder_destroy(dec_);
      if (feat_) kamd_feat_destroy(feat_);
      throw KaldiFatalError(why);
    }
    input_dim_ = kamd_nnet_input_dim(am_nnet.Handle()); ivector_dim_ = kamd_nnet_ivector_dim(am_nnet.Handle());
  }
  ~NnetBatchDecoder() {
    kamd_batch_decoder_destroy(h_); kamd_decoder_destroy(dec_);
    if (dec_long_) kamd_decoder_destroy(dec_long_);
    if (feat_) kamd_feat_destroy(feat_);
  }
  /// Not in the reference: for a shard so small that the longest utterance's own chain of frames bounds the search (one
  /// rank of an 8-GPU run), a second decoder object of `sizes.max_lanes` lanes on which Finished() searches the longest
  /// utterances while the acoustic model of the others is still running (kamd_batch_decoder_set_long_decoder; outputs do
  /// not change).  Call before the first AcceptInput.
  void SetLongUtteranceDecoder(const DecodingGraph &fst, const std::vector<int32> &id2pdf, const kamd_decoder_sizes &sizes) {
    if (dec_long_) throw KaldiFatalError("NnetBatchDecoder: the long-utterance decoder is set already");
    kamd_decoder_config c = config_.ToC();
    dec_long_ = kamd_decoder_create(fst.Handle(), &c, &sizes, id2pdf.empty() ? NULL : id2pdf.data(), id2pdf.empty() ? 0 : static_cast<int32>(id2pdf.size()) - 1);
    if (!dec_long_ || kamd_decoder_set_search_mode(dec_long_, opts_.search_mode) != 0 ||
        kamd_batch_decoder_set_long_decoder(h_, dec_long_, sizes.max_lanes) != 0)
      throw KaldiFatalError(kamd_last_error());
  }
  /// Not in the reference's class (there, online i-vectors arrive as matrices extract_ivectors_online.sh wrote):
  /// i-vectors estimated on the device from every pass's features, the model evaluated chunk by chunk like
  /// DecodableNnetSimple with --online-ivectors (kamd_batch_decoder_set_ivector_extractor).  Before the first AcceptInput /
  /// AcceptWaveform; utterances then come without an i-vector of their own.
  /// batch_computer_tasks: the chunking of NnetBatchComputer::SplitUtteranceIntoTasks (nnet3/nnet-batch-compute.cc:774-829 -- what
  /// the reference's NnetBatchDecoder is fed by) instead of DecodableNnetSimple's (kamd_batch_decoder_set_chunk_rule).
  void SetIvectorExtractor(kamd_ivector_extractor *extractor, int32 frames_per_chunk = 50, bool batch_computer_tasks = false) {
    Check(kamd_batch_decoder_set_chunk_rule(h_, batch_computer_tasks ? 1 : 0));
    Check(kamd_batch_decoder_set_ivector_extractor(h_, extractor, frames_per_chunk));
    online_ivectors_ = extractor != NULL;
  }
  NnetBatchDecoder(const NnetBatchDecoder &) = delete;
  NnetBatchDecoder &operator=(const NnetBatchDecoder &) = delete;

  /// input: [num_rows x num_cols] row-major features (Matrix<BaseFloat>); ivector: NULL or [ivector_dim].
  /// online_ivectors are not taken here: the device estimates them itself (OnlineStreamBatch / kamd_pipeline_set_ivector_extractor).
  void AcceptInput(const std::string &utterance_id, const float *input, int32 num_rows, int32 num_cols, const float *ivector,
                   int32 ivector_dim, const float *online_ivectors = NULL, int32 online_ivector_period = 0) {
    (void)online_ivector_period;
    if (finished_) throw KaldiFatalError("NnetBatchDecoder: AcceptInput after Finished()");
    if (online_ivectors) throw KaldiFatalError("NnetBatchDecoder: online i-vectors as input are not supported; the device estimates them");
    if (!waves_.empty()) throw KaldiFatalError("NnetBatchDecoder: a set is either waveforms or feature matrices");
    if (num_rows <= 0) throw KaldiFatalError("Zero-length utterance: " + utterance_id);
    if (num_cols != input_dim_) throw KaldiFatalError("NnetBatchDecoder: feature dim mismatch for " + utterance_id);
    if (!(online_ivectors_ && !ivector) && (ivector ? ivector_dim : 0) != ivector_dim_)
      throw KaldiFatalError("NnetBatchDecoder: i-vector dim mismatch for " + utterance_id);
    keys_.push_back(utterance_id);
    feats_.insert(feats_.end(), input, input + static_cast<size_t>(num_rows) * num_cols);
    row_off_.push_back(row_off_.empty() ? num_rows : row_off_.back() + num_rows);
    if (ivector) ivectors_.insert(ivectors_.end(), ivector, ivector + ivector_dim);
  }
  void AcceptWaveform(const std::string &utterance_id, const std::vector<float> &wave) {
    if (finished_) throw KaldiFatalError("NnetBatchDecoder: AcceptWaveform after Finished()");
    if (!feat_) throw KaldiFatalError("NnetBatchDecoder: constructed without feature options");
    if (!feats_.empty()) throw KaldiFatalError("NnetBatchDecoder: a set is either waveforms or feature matrices");
    keys_.push_back(utterance_id);
    waves_.insert(waves_.end(), wave.begin(), wave.end());
    row_off_.push_back(static_cast<int64_t>(waves_.size()));
  }
  void UtteranceFailed() { num_fail_++; }

  /// Runs the set; returns the number of utterances successfully decoded (nnet-batch-compute.cc:1310-1343).
  int32 Finished() {
    if (finished_) throw KaldiFatalError("NnetBatchDecoder: Finished() called twice");
    finished_ = true;
    if (keys_.empty()) return 0;
    std::vector<int64_t> off(1, 0);
    off.insert(off.end(), row_off_.begin(), row_off_.end());
    const int n = static_cast<int>(keys_.size());
    // waveforms stay in this object's memory: run() uploads them pass by pass behind the features and the model
    if (!waves_.empty()) Check(kamd_batch_decoder_load_host(h_, waves_.data(), off.data(), n));
    else Check(kamd_batch_decoder_load_features(h_, feats_.data(), off.data(), input_dim_, ivectors_.empty() ? NULL : ivectors_.data(), ivector_dim_, n));
    std::vector<float>().swap(feats_);
    Check(kamd_batch_decoder_run(h_, &stats_));
    // the library page-locked waves_ in place: the lock must be gone before the memory is
    Check(kamd_batch_decoder_unload_host(h_));
    std::vector<float>().swap(waves_);
    // the per-utterance outcome, with the log lines of decoder-wrappers.cc:228-292
    ok_.assign(n, 0);
    for (int u = 0; u < n; u++) {
      kamd_queue_result rec;
      int nw = 0, na = 0; float g = 0, a = 0;
      const int rc = kamd_batch_decoder_get_output(h_, u, NULL, 0, &nw, NULL, 0, &na, &g, &a, &rec);
      if (rc != 0) { fprintf(stderr, "WARNING Decoding failed for utterance %s: %s\n", keys_[u].c_str(), kamd_last_error()); num_fail_++; continue; }
      const bool reached_final = rec.final_relative_cost != std::numeric_limits<float>::infinity();
      if (!reached_final) {
        if (!allow_partial_) {
          fprintf(stderr, "WARNING Not producing output for utterance %s since no final-state reached and --allow-partial=false.\n", keys_[u].c_str());
          num_fail_++;
          continue;
        }
        fprintf(stderr, "WARNING Outputting partial output for utterance %s since no final-state reached\n", keys_[u].c_str());
        num_partial_++;
      }
      ok_[u] = 1; num_success_++;
      tot_like_ += -static_cast<double>(g + a); frame_count_ += rec.n_frames;
      fprintf(stderr, "LOG Log-like per frame for utterance %s is %g over %d frames.\n", keys_[u].c_str(),
              rec.n_frames > 0 ? -(g + a) / rec.n_frames : 0.0, rec.n_frames);
    }
    return num_success_;
  }

  /// determinize_lattice == true version: outputs in input order, false when nothing is left.
  bool GetOutput(std::string *utterance_id, CompactLattice *clat, std::string *sentence) {
    if (!config_.determinize_lattice) throw KaldiFatalError("Don't call this version of GetOutput if you are not determinizing.");
    const int u = NextOutput();
    if (u < 0) return false;
    const kamd_compact_lattice *c = kamd_batch_decoder_get_compact_lattice(h_, u);
    if (!c) throw KaldiFatalError(kamd_last_error());
    kamd_compact_lattice *mine = CheckPtr(kamd_compact_lattice_copy(c));
    if (opts_.acoustic_scale != 0.0f && opts_.acoustic_scale != 1.0f) kamd_compact_lattice_scale(mine, 1.0f, 1.0f / opts_.acoustic_scale);
    clat->Reset(mine);
    *utterance_id = keys_[u];
    Sentence(u, sentence);
    return true;
  }
  /// determinize_lattice == false version.
  bool GetOutput(std::string *utterance_id, Lattice *lat, std::string *sentence) {
    if (config_.determinize_lattice) throw KaldiFatalError("Don't call this version of GetOutput if you are determinizing.");
    const int u = NextOutput();
    if (u < 0) return false;
    int32 ns = 0, na = 0, start = -1;
    const int32 *fr, *hc; const float *cost, *fin; const kamd_lat_arc *arcs;
    Check(kamd_batch_decoder_get_raw_lattice(h_, u, &ns, &na, &start, &fr, &hc, &cost, &fin, &arcs));
    const float inv = opts_.acoustic_scale != 0.0f && opts_.acoustic_scale != 1.0f ? 1.0f / opts_.acoustic_scale : 1.0f;
    lat->start = start;
    lat->arcs.assign(ns, std::vector<LatticeArc>());
    lat->final_graph_cost.assign(fin, fin + ns);
    lat->state_frame.assign(fr, fr + ns); lat->state_hclg.assign(hc, hc + ns);
    for (int32 i = 0; i < na; i++) {
      LatticeArc a = {arcs[i].ilabel, arcs[i].olabel, arcs[i].graph_cost, arcs[i].acoustic_cost * inv, arcs[i].dst};
      lat->arcs[arcs[i].src].push_back(a);
    }
    *utterance_id = keys_[u];
    Sentence(u, sentence);
    return true;
  }
  /// the best path of the utterance LAST returned by GetOutput (DecodeUtteranceLatticeFaster's words / alignment writers)
  bool GetBestPath(std::vector<int32> *alignment, std::vector<int32> *words, BaseFloat *graph_cost, BaseFloat *acoustic_cost) const {
    return BestPath(next_out_ - 1, alignment, words, graph_cost, acoustic_cost);
  }
  int32 NumSuccess() const { return num_success_; }
  int32 NumFail() const { return num_fail_; }
  int32 NumPartial() const { return num_partial_; }
  double TotLike() const { return tot_like_; }
  int64_t FrameCount() const { return frame_count_; }
  const kamd_batch_stats &Stats() const { return stats_; }
  kamd_decoder *DecoderHandle() { return dec_; }
 private:
  int NextOutput() {
    if (!finished_) return -1;           // nothing is ready before Finished() here: the set runs as one pass
    while (next_out_ < static_cast<int>(keys_.size()) && !ok_[next_out_]) next_out_++;
    if (next_out_ >= static_cast<int>(keys_.size())) return -1;
    return next_out_++;
  }
  bool BestPath(int u, std::vector<int32> *alignment, std::vector<int32> *words, BaseFloat *graph_cost, BaseFloat *acoustic_cost) const {
    if (u < 0 || u >= static_cast<int>(keys_.size())) return false;
    int nw = 0, na = 0;
    if (kamd_batch_decoder_get_output(h_, u, NULL, 0, &nw, NULL, 0, &na, graph_cost, acoustic_cost, NULL) != 0) return false;
    std::vector<int32> w(std::max(nw, 1)), a(std::max(na, 1));
    if (kamd_batch_decoder_get_output(h_, u, w.data(), static_cast<int>(w.size()), &nw, a.data(), static_cast<int>(a.size()), &na, graph_cost,
                                      acoustic_cost, NULL) != 0) return false;
    words->assign(w.begin(), w.begin() + nw); alignment->assign(a.begin(), a.begin() + na);
    return true;
  }
  void Sentence(int u, std::string *sentence) const {
    sentence->clear();
    if (!word_syms_) return;
    std::vector<int32> ali, words; BaseFloat g, a;
    if (!BestPath(u, &ali, &words, &g, &a)) return;
    for (size_t k = 0; k < words.size(); k++) {
      if (words[k] < 0 || words[k] >= static_cast<int32>(word_syms_->size()) || (*word_syms_)[words[k]].empty())
        throw KaldiFatalError("Word-id " + std::to_string(words[k]) + " not in symbol table.");
      if (k) sentence->push_back(' ');
      sentence->append((*word_syms_)[words[k]]);
    }
  }
  LatticeFasterDecoderConfig config_;
  const std::vector<std::string> *word_syms_;
  bool allow_partial_;
  NnetBatchDecoderOptions opts_;
  kamd_feat *feat_;
  kamd_decoder *dec_;
  kamd_decoder *dec_long_ = NULL;
  kamd_batch_decoder *h_;
  int32 input_dim_, ivector_dim_;
  bool online_ivectors_ = false;
  std::vector<std::string> keys_;
  std::vector<float> waves_, feats_, ivectors_;
  std::vector<int64_t> row_off_;
  std::vector<char> ok_;
  int next_out_;
  bool finished_;
  int32 num_success_, num_fail_, num_partial_;
  double tot_like_;
  int64_t frame_count_;
  kamd_batch_stats stats_;
};

// ---- lm/const-arpa-lm.h:211-352 ConstArpaLm + latbin/lattice-lmrescore-const-arpa.cc:76-110
class ConstArpaLm {
 public:
  ConstArpaLm() : lm_(NULL) {}
  ~ConstArpaLm() { if (lm_) kamd_const_arpa_destroy(lm_); }
  ConstArpaLm(const ConstArpaLm &) = delete;
  ConstArpaLm &operator=(const ConstArpaLm &) = delete;
  /// ReadKaldiObject(lm_rxfilename, &const_arpa): a G.carpa written by arpa-to-const-arpa (either format)
  void Read(const std::string &rxfilename) { Reset(CheckPtr(kamd_const_arpa_read(rxfilename.c_str()))); }
  void Write(const std::string &wxfilename) const { Check(kamd_const_arpa_write(lm_, wxfilename.c_str())); }
  /// BuildConstArpaLm (const-arpa-lm.cc:1064-1073); words_txt empty = the ARPA file holds integer word ids
  void Build(const std::string &arpa_rxfilename, int32 bos_symbol, int32 eos_symbol, int32 unk_symbol, const std::string &words_txt = "") {
    Reset(CheckPtr(kamd_const_arpa_build(arpa_rxfilename.c_str(), bos_symbol, eos_symbol, unk_symbol, words_txt.empty() ? NULL : words_txt.c_str())));
  }
  /// natural log; hist = the words before `word`, oldest first
  float GetNgramLogprob(int32 word, const std::vector<int32> &hist) const {
    return kamd_const_arpa_ngram_logprob(lm_, word, hist.data(), static_cast<int>(hist.size()));
  }
  int32 BosSymbol() const { int32 b, e, u, o, n; int64_t sz; Check(kamd_const_arpa_info(lm_, &b, &e, &u, &o, &n, &sz)); return b; }
  int32 EosSymbol() const { int32 b, e, u, o, n; int64_t sz; Check(kamd_const_arpa_info(lm_, &b, &e, &u, &o, &n, &sz)); return e; }
  int32 NgramOrder() const { int32 b, e, u, o, n; int64_t sz; Check(kamd_const_arpa_info(lm_, &b, &e, &u, &o, &n, &sz)); return o; }
  const kamd_const_arpa *Handle() const { return lm_; }
 private:
  void Reset(kamd_const_arpa *l) { if (lm_) kamd_const_arpa_destroy(lm_); lm_ = l; }
  kamd_const_arpa *lm_;
};

/// The body of lattice-lmrescore-const-arpa's loop for one lattice: ScaleLattice(GraphLatticeScale(1/lm_scale)),
/// ComposeCompactLatticeDeterministic with ConstArpaLmDeterministicFst, DeterminizeLattice,
/// ScaleLattice(GraphLatticeScale(lm_scale)).  false = "Empty lattice ... (incompatible LM?)".  lm_scale = 0: copy.
inline bool LatticeLmrescoreConstArpa(BaseFloat lm_scale, const ConstArpaLm &const_arpa, const CompactLattice &clat, CompactLattice *out) {
  if (lm_scale == 0.0f) { out->Reset(CheckPtr(kamd_compact_lattice_copy(clat.Handle()))); return true; }
  int32 ns = 0, na = 0, nl = 0, start = -1, ok = 0;
  Check(kamd_compact_lattice_sizes(clat.Handle(), &ns, &na, &nl, &start, &ok));
  std::vector<float> fin(2 * static_cast<size_t>(ns) + 1);
  std::vector<int32> fb(ns + 1), fl(ns + 1), strings(nl + 1);
  std::vector<kamd_clat_arc> arcs(na + 1);
  Check(kamd_compact_lattice_get(clat.Handle(), fin.data(), fb.data(), fl.data(), arcs.data(), strings.data()));
  kamd_compact_lattice *r = kamd_compact_lattice_lmrescore_const_arpa(ns, start, fin.data(), fb.data(), fl.data(), arcs.data(), na, strings.data(),
                                                                      const_arpa.Handle(), lm_scale);
  if (!r) {
    if (std::string(kamd_last_error()).find("Empty lattice") != std::string::npos) return false;
    throw KaldiFatalError(kamd_last_error());
  }
  out->Reset(r);
  return true;
}

// ---- N concurrent SingleUtteranceNnet3Decoder's behind one set of launches (kamd_stream_batch_*): stream s is decoder lane s.
// With an OnlineIvectorExtractor the streams get online i-vectors as OnlineNnet2FeaturePipeline + DecodableAmNnetLoopedOnline
// compute them (online2-wav-nnet3-latgen-faster.cc:200-290).
class OnlineIvectorExtractor;
class OnlineStreamBatch {
 public:
  OnlineStreamBatch(const LatticeFasterDecoderConfig &decoder_opts, const std::vector<int32> &id2pdf, const AmNnetSimple &am_nnet,
                    const DecodingGraph &fst, const MfccOptions &mfcc_opts, int32 max_streams, BaseFloat max_seconds,
                    const kamd_decoder_sizes &sizes)
      : feat_(CheckPtr(kamd_mfcc_create(&mfcc_opts.c, 1.0f))), ie_(NULL) {
    Init(decoder_opts, id2pdf, am_nnet, fst, mfcc_opts.c.frame, max_streams, max_seconds, sizes);
  }
  /// --feature-type=fbank (OnlineFbank as the base feature, online2/online-nnet2-feature-pipeline.cc:84-85)
  OnlineStreamBatch(const LatticeFasterDecoderConfig &decoder_opts, const std::vector<int32> &id2pdf, const AmNnetSimple &am_nnet,
                    const DecodingGraph &fst, const FbankOptions &fbank_opts, int32 max_streams, BaseFloat max_seconds,
                    const kamd_decoder_sizes &sizes)
      : feat_(CheckPtr(kamd_fbank_create(&fbank_opts.c, 1.0f))), ie_(NULL) {
    Init(decoder_opts, id2pdf, am_nnet, fst, fbank_opts.c.frame, max_streams, max_seconds, sizes);
  }
  ~OnlineStreamBatch() { kamd_stream_batch_destroy(h_); kamd_decoder_destroy(dec_); kamd_feat_destroy(feat_); }
  OnlineStreamBatch(const OnlineStreamBatch &) = delete;
  OnlineStreamBatch &operator=(const OnlineStreamBatch &) = delete;
  /// before the first Start(): --ivector-extraction-config of the online2 binaries
  void SetIvectorExtractor(kamd_ivector_extractor *extractor, int32 frames_per_chunk, int32 splice_right) {
    Check(kamd_stream_batch_set_ivector_extractor(h_, extractor, frames_per_chunk, splice_right));
    ie_ = extractor;
  }
  /// OnlineSilenceWeightingConfig (online2/online-ivector-feature.h:404-451) for every stream: tid2phone[tid] =
  /// TransitionIdToPhone(tid) (index 0 unused), silence_phones as --ivector-silence-weighting.silence-phones lists them.
  /// After SetIvectorExtractor, before the first Start.  silence_weight == 1 or no phones: off, like Active().
  void SetSilenceWeighting(const std::vector<int32> &tid2phone, const std::vector<int32> &silence_phones, BaseFloat silence_weight,
                           BaseFloat max_state_duration = -1.0f) {
    std::vector<uint8_t> is_sil(tid2phone.size(), 0);
    for (size_t t = 1; t < tid2phone.size(); t++)
      is_sil[t] = std::find(silence_phones.begin(), silence_phones.end(), tid2phone[t]) != silence_phones.end();
    Check(kamd_stream_batch_set_silence_weighting(h_, is_sil.data(), static_cast<int>(is_sil.size()), silence_weight, max_state_duration));
  }
  /// arena compaction threshold of the streams (PruneActiveTokens when a stream's arena is fuller than this); 0 = never
  void SetCompaction(BaseFloat fraction) { Check(kamd_stream_batch_set_compaction(h_, fraction)); }
  /// LatticeFasterDecoderConfig::prune_interval: PruneActiveTokens of a stream every `frames` decoded frames (0 = never)
  void SetPruneInterval(int32 frames) { Check(kamd_stream_batch_set_prune_interval(h_, frames)); }
  /// new utterances; adaptation_states (optional): one state per stream, kamd_ivector_state_size() doubles each
  void Start(const std::vector<int32> &streams, const std::vector<double> *adaptation_states = NULL) {
    if (adaptation_states && ie_) Check(kamd_stream_batch_start_adapted(h_, streams.data(), static_cast<int>(streams.size()), adaptation_states->data()));
    else Check(kamd_stream_batch_start(h_, streams.data(), static_cast<int>(streams.size())));
  }
  void AcceptWaveform(int32 stream, const float *wave, int64_t n, bool input_finished) {
    Check(kamd_stream_batch_accept(h_, stream, wave, n, input_finished ? 1 : 0));
  }
  /// AcceptWaveform of several streams with one upload: waves[offsets[i] .. offsets[i+1]) goes to streams[i]
  void AcceptWaveforms(const std::vector<int32> &streams, const float *waves, const std::vector<int64_t> &offsets,
                       const std::vector<int32> *input_finished = NULL) {
    if (offsets.size() != streams.size() + 1) throw KaldiFatalError("AcceptWaveforms: offsets must have one more entry than streams");
    Check(kamd_stream_batch_accept_many(h_, streams.data(), static_cast<int>(streams.size()), waves, offsets.data(),
                                        input_finished ? input_finished->data() : NULL));
  }
  /// one tick: AdvanceDecoding of all listed streams
  void AdvanceDecoding(const std::vector<int32> &streams, std::vector<int32> *frames_decoded = NULL) {
    if (frames_decoded) frames_decoded->resize(streams.size());
    Check(kamd_stream_batch_advance(h_, streams.data(), static_cast<int>(streams.size()), frames_decoded ? frames_decoded->data() : NULL));
  }
  void FinalizeDecoding(const std::vector<int32> &streams) {
    Check(kamd_decoder_finalize(dec_, streams.data(), static_cast<int>(streams.size()), NULL));
    Check(kamd_decoder_sync(dec_));
  }
  bool GetBestPath(int32 stream, std::vector<int32> *alignment, std::vector<int32> *words, BaseFloat *graph_cost, BaseFloat *acoustic_cost) const {
    kamd_lattice_size sz;
    Check(kamd_decoder_lattice_size(dec_, stream, &sz));
    const int cap = sz.num_arcs + 1;
    std::vector<int32> ali(cap), wrd(cap);
    int na = 0, nw = 0;
    if (kamd_decoder_best_path(dec_, stream, ali.data(), cap, &na, wrd.data(), cap, &nw, graph_cost, acoustic_cost) != 0) return false;
    alignment->assign(ali.begin(), ali.begin() + na); words->assign(wrd.begin(), wrd.begin() + nw);
    return true;
  }
  /// GetRawLattice of a finalized stream (LatticeFasterDecoder::GetRawLattice, lane `stream` of the shared decoder)
  bool GetRawLattice(int32 stream, Lattice *ofst) const {
    kamd_lattice_size sz;
    Check(kamd_decoder_lattice_size(dec_, stream, &sz));
    ofst->arcs.assign(sz.num_states, std::vector<LatticeArc>());
    ofst->final_graph_cost.assign(sz.num_states, 0.f);
    ofst->state_frame.assign(sz.num_states, 0); ofst->state_hclg.assign(sz.num_states, 0);
    ofst->start = sz.start;
    if (sz.num_states == 0) return false;
    std::vector<float> cost(sz.num_states);
    std::vector<kamd_lat_arc> arcs(sz.num_arcs);
    Check(kamd_decoder_get_raw_lattice(dec_, stream, ofst->state_frame.data(), ofst->state_hclg.data(), cost.data(), ofst->final_graph_cost.data(), arcs.data()));
    for (size_t i = 0; i < arcs.size(); i++) {
      LatticeArc a = {arcs[i].ilabel, arcs[i].olabel, arcs[i].graph_cost, arcs[i].acoustic_cost, arcs[i].dst};
      ofst->arcs[arcs[i].src].push_back(a);
    }
    return true;
  }
  /// GetAdaptationState + LimitFrames: what the speaker's next utterance starts from
  void GetAdaptationState(int32 stream, BaseFloat max_remembered_frames, std::vector<double> *state) const {
    state->resize(kamd_ivector_state_size(ie_));
    Check(kamd_stream_batch_get_adaptation_state(h_, stream, state->data()));
    Check(kamd_ivector_state_limit_frames(ie_, state->data(), max_remembered_frames));
  }
  /// partial results (GetBestPath(end_of_utterance = false)) of these streams after a tick, one launch; words[i] is empty and
  /// ok[i] false for a stream with no token alive
  /// incremental: kamd_decoder_partial_best_paths_incremental (the decoder keeps every stream's previous answer and walks
  /// back only to the first frame whose best-path token is unchanged: what a server calls after every tick)
  void GetPartialBestPaths(const std::vector<int32> &streams, std::vector<std::vector<int32> > *words,
                           std::vector<std::vector<int32> > *alignments = NULL, std::vector<char> *ok = NULL, bool incremental = false) {
    const int n = static_cast<int>(streams.size());
    int32 frames = 0;
    for (int i = 0; i < n; i++) frames = std::max(frames, kamd_decoder_num_frames_decoded(dec_, streams[i]));
    const int cap = 4 * (frames + 2) + 1024;
    std::vector<int32> ali(static_cast<size_t>(n) * cap), wrd(static_cast<size_t>(n) * cap), na(n), nw(n);
    std::vector<float> g(n), a(n);
    if (incremental) Check(kamd_decoder_partial_best_paths_incremental(dec_, streams.data(), n, ali.data(), cap, na.data(), wrd.data(), cap, nw.data(), g.data(), a.data()));
    else Check(kamd_decoder_partial_best_paths(dec_, streams.data(), n, 0, ali.data(), cap, na.data(), wrd.data(), cap, nw.data(), g.data(), a.data()));
    words->assign(n, std::vector<int32>());
    if (alignments) alignments->assign(n, std::vector<int32>());
    if (ok) ok->assign(n, 0);
    for (int i = 0; i < n; i++) {
      if (na[i] < 0) continue;
      (*words)[i].assign(wrd.begin() + static_cast<size_t>(i) * cap, wrd.begin() + static_cast<size_t>(i) * cap + std::min(nw[i], cap));
      if (alignments) (*alignments)[i].assign(ali.begin() + static_cast<size_t>(i) * cap, ali.begin() + static_cast<size_t>(i) * cap + std::min(na[i], cap));
      if (ok) (*ok)[i] = 1;
    }
  }
  /// EndpointDetected for these streams after a tick, one launch (online2-wav-nnet3-latgen-faster.cc --do-endpointing)
  void EndpointDetected(const OnlineEndpointConfig &config, const std::vector<int32> &tid2phone, const std::vector<int32> &streams,
                        std::vector<int32> *detected, std::vector<int32> *trailing_silence_frames = NULL) {
    kaldi_amd::EndpointDetected(config, tid2phone, frame_shift_, dec_, streams, detected, trailing_silence_frames);
  }
  kamd_decoder *DecoderHandle() { return dec_; }
 private:
  void Init(const LatticeFasterDecoderConfig &decoder_opts, const std::vector<int32> &id2pdf, const AmNnetSimple &am_nnet,
            const DecodingGraph &fst, const kamd_frame_opts &frame, int32 max_streams, BaseFloat max_seconds, const kamd_decoder_sizes &sizes) {
    dec_ = NULL; h_ = NULL;
    try { decoder_opts.Check(); } catch (...) { kamd_feat_destroy(feat_); throw; }
    kamd_decoder_config c = decoder_opts.ToC();
    dec_ = kamd_decoder_create(fst.Handle(), &c, &sizes, id2pdf.empty() ? NULL : id2pdf.data(), id2pdf.empty() ? 0 : static_cast<int32>(id2pdf.size()) - 1);
    if (dec_) h_ = kamd_stream_batch_create(feat_, am_nnet.Handle(), dec_, max_streams, max_seconds, frame.samp_freq);
    if (!h_) {
      const std::string msg = kamd_last_error();
      if (dec_) kamd_decoder_destroy(dec_);
      kamd_feat_destroy(feat_);
      throw KaldiFatalError(msg);
    }
    frame_shift_ = frame.frame_shift_ms * 1.0e-3f * kamd_nnet_frame_subsampling_factor(am_nnet.Handle());
  }
  BaseFloat frame_shift_;
  kamd_feat *feat_;
  kamd_decoder *dec_;
  kamd_stream_batch *h_;
  kamd_ivector_extractor *ie_;
};

// ---- online i-vectors: OnlineIvectorFeature over a whole utterance (online2/online-ivector-feature.h:246-330)
// with the speaker's OnlineIvectorExtractorAdaptationState, as ivector-extract-online2 uses it.
class OnlineIvectorExtractorAdaptationState {
 public:
  bool empty() const { return state_.empty(); }
  std::vector<double> state_;          // kamd_ivector_state_size() doubles once set
};
class OnlineIvectorExtractor {
 public:
  // desc: the fields of OnlineIvectorExtractionInfo (lda_mat, global_cmvn_stats, diag_ubm, extractor M_ / Sigma_inv_, options)
  explicit OnlineIvectorExtractor(const kamd_ivector_desc &desc, BaseFloat max_remembered_frames = 1000.0)
      : h_(kamd_ivector_extractor_create(&desc)), max_remembered_frames_(max_remembered_frames), splice_right_(desc.splice_right) {
    if (!h_) throw KaldiFatalError(kamd_last_error());
  }
  /// OnlineIvectorExtractionInfo::Init from the files of an extraction config (--ivector-extraction-config of the online2
  /// binaries, --config of ivector-extract-online2; online2/online-ivector-feature.cc:30-74): kamd_ivector_info_read
  explicit OnlineIvectorExtractor(const std::string &config_rxfilename, BaseFloat max_remembered_frames = 1000.0)
      : h_(NULL), max_remembered_frames_(max_remembered_frames), splice_right_(0) {
    kamd_ivector_info *info = CheckPtr(kamd_ivector_info_read(config_rxfilename.c_str()));
    splice_right_ = kamd_ivector_info_desc(info)->splice_right;
    h_ = kamd_ivector_info_create_extractor(info);
    kamd_ivector_info_destroy(info);
    if (!h_) throw KaldiFatalError(kamd_last_error());
  }
  ~OnlineIvectorExtractor() { kamd_ivector_extractor_destroy(h_); }
  OnlineIvectorExtractor(const OnlineIvectorExtractor &) = delete;
  OnlineIvectorExtractor &operator=(const OnlineIvectorExtractor &) = delete;
  int32 Dim() const { return kamd_ivector_dim(h_); }
  int32 NumIvectors(int32 num_frames) const { return kamd_ivector_num_ivectors(h_, num_frames); }
  // feats: [num_frames x feat_dim] row-major; ivectors: resized to [NumIvectors x Dim()].  adaptation_state (may be
  // NULL) is read if set (SetAdaptationState) and replaced by the state after this utterance with LimitFrames
  // applied (GetAdaptationState), ready for the speaker's next utterance.
  void ExtractOnline(const float *feats, int32 num_frames, std::vector<float> *ivectors,
                     OnlineIvectorExtractorAdaptationState *adaptation_state = NULL) {
    const int32 n = NumIvectors(num_frames);
    ivectors->assign(static_cast<size_t>(n) * Dim(), 0.0f);
    std::vector<double> out_state(adaptation_state ? kamd_ivector_state_size(h_) : 0);
    const double *in = (adaptation_state && !adaptation_state->empty()) ? adaptation_state->state_.data() : NULL;
    Check(kamd_ivector_extract_online_adapt(h_, feats, num_frames, ivectors->data(), n, in, adaptation_state ? out_state.data() : NULL));
    if (adaptation_state) {
      Check(kamd_ivector_state_limit_frames(h_, out_state.data(), max_remembered_frames_));
      adaptation_state->state_.swap(out_state);
    }
  }
  kamd_ivector_extractor *handle() { return h_; }
  /// right context of the splicing in front of the LDA (OnlineSpliceOptions): what OnlineStreamBatch::SetIvectorExtractor asks for
  int32 SpliceRight() const { return splice_right_; }
 private:
  kamd_ivector_extractor *h_;
  BaseFloat max_remembered_frames_;
  int32 splice_right_;
};

// ---- OnlineSilenceWeightingConfig (online2/online-ivector-feature.h:404-451), registered with the prefix
// "ivector-silence-weighting" by OnlineNnet2FeaturePipelineConfig (online-nnet2-feature-pipeline.h:89-110)
struct OnlineSilenceWeightingConfig {
  std::string silence_phones_str;
  BaseFloat silence_weight, max_state_duration;
  OnlineSilenceWeightingConfig() : silence_weight(1.0f), max_state_duration(-1.0f) {}
  bool Active() const { return !silence_phones_str.empty() && silence_weight != 1.0f; }
  template <typename Opts> void RegisterWithPrefix(const std::string &prefix, Opts *opts) {
    opts->Register(prefix + ".silence-phones", &silence_phones_str, "(RE weighting in iVector estimation for online decoding) List of integer ids of "
                   "silence phones, separated by colons (or commas).  Data that (according to the traceback of the decoder) corresponds to "
                   "these phones will be downweighted by --silence-weight.");
    opts->Register(prefix + ".silence-weight", &silence_weight, "(RE weighting in iVector estimation for online decoding) Weighting factor for "
                   "frames that the decoder trace-back identifies as silence; only relevant if the --silence-phones option is set.");
    opts->Register(prefix + ".max-state-duration", &max_state_duration, "(RE weighting in iVector estimation for online decoding) Maximum allowed "
                   "duration of a single transition-id; runs with durations longer than this will be weighted down to the silence-weight.");
  }
  /// SplitStringToIntegers(silence_phones_str, ":,", false, ...)
  std::vector<int32> SilencePhones() const {
    std::vector<int32> out;
    std::string tok;
    for (size_t i = 0; i <= silence_phones_str.size(); i++) {
      const char ch = i < silence_phones_str.size() ? silence_phones_str[i] : ':';
      if (ch != ':' && ch != ',') { tok += ch; continue; }
      if (tok.empty()) continue;
      char *end = NULL;
      const long v = strtol(tok.c_str(), &end, 10);
      if (*end != 0) throw KaldiFatalError("Bad --silence-phones option in silence-weighting config: " + silence_phones_str);
      out.push_back(static_cast<int32>(v));
      tok.clear();
    }
    return out;
  }
};

// ---- compute-mfcc-feats / compute-fbank-feats options for a ParseOptions: FrameExtractionOptions::Register
// (feat/feature-window.h:60-86), MelBanksOptions::Register (feat/mel-computations.h:58-76), MfccOptions::Register
// (feat/feature-mfcc.h:58-88), FbankOptions::Register (feat/feature-fbank.h:60-80).  Register, read, then Finish() copies
// the flags into the C struct.
struct FrameMelOptionsParser {
  kamd_frame_opts *frame;
  kamd_mel_opts *mel;
  bool remove_dc_offset, snip_edges, round_to_power_of_two;
  BaseFloat dither;
  std::string window_type;
  FrameMelOptionsParser(kamd_frame_opts *f, kamd_mel_opts *m)
      : frame(f), mel(m), remove_dc_offset(f->remove_dc_offset != 0), snip_edges(f->snip_edges != 0),
        round_to_power_of_two(f->round_to_power_of_two != 0), dither(0.0f) {
    static const char *const names[] = {"hanning", "hamming", "povey", "rectangular", "blackman"};
    window_type = names[f->window_type >= 0 && f->window_type < 5 ? f->window_type : 2];
  }
  void Register(ParseOptions *po) {
    po->Register("sample-frequency", &frame->samp_freq, "Waveform data sample frequency (must match the waveform file, if specified there)");
    po->Register("frame-length", &frame->frame_length_ms, "Frame length in milliseconds");
    po->Register("frame-shift", &frame->frame_shift_ms, "Frame shift in milliseconds");
    po->Register("preemphasis-coefficient", &frame->preemph_coeff, "Coefficient for use in signal preemphasis");
    po->Register("remove-dc-offset", &remove_dc_offset, "Subtract mean from waveform on each frame");
    po->Register("dither", &dither, "Dithering constant (0.0 means no dither); only 0 is supported here (dithering is random in the reference)");
    po->Register("window-type", &window_type, "Type of window (\"hamming\"|\"hanning\"|\"povey\"|\"rectangular\"|\"blackmann\")");
    po->Register("blackman-coeff", &frame->blackman_coeff, "Constant coefficient for generalized Blackman window.");
    po->Register("round-to-power-of-two", &round_to_power_of_two, "If true, round window size to power of two by zero-padding input to FFT.");
    po->Register("snip-edges", &snip_edges, "If true, end effects will be handled by outputting only frames that completely fit in the file");
    po->Register("num-mel-bins", &mel->num_bins, "Number of triangular mel-frequency bins");
    po->Register("low-freq", &mel->low_freq, "Low cutoff frequency for mel bins");
    po->Register("high-freq", &mel->high_freq, "High cutoff frequency for mel bins (if <= 0, offset from Nyquist)");
    po->Register("vtln-low", &mel->vtln_low, "Low inflection point in piecewise linear VTLN warping function");
    po->Register("vtln-high", &mel->vtln_high, "High inflection point in piecewise linear VTLN warping function (if negative, offset from high-mel-freq");
  }
  void Finish() {
    if (dither != 0.0f) throw KaldiFatalError("--dither: only 0 is supported");
    static const char *const names[] = {"hanning", "hamming", "povey", "rectangular", "blackman"};
    int w = -1;
    for (int i = 0; i < 5; i++) if (window_type == names[i]) w = i;
    if (w < 0) throw KaldiFatalError("Invalid window type " + window_type);
    frame->window_type = w; frame->dither = 0.0f;
    frame->remove_dc_offset = remove_dc_offset; frame->snip_edges = snip_edges; frame->round_to_power_of_two = round_to_power_of_two;
  }
};
struct MfccOptionsParser {
  MfccOptions *opts;
  FrameMelOptionsParser fm;
  bool use_energy, raw_energy, htk_compat;
  explicit MfccOptionsParser(MfccOptions *o)
      : opts(o), fm(&o->c.frame, &o->c.mel), use_energy(o->c.use_energy != 0), raw_energy(o->c.raw_energy != 0), htk_compat(o->c.htk_compat != 0) {}
  void Register(ParseOptions *po) {
    kamd_mfcc_opts &c = opts->c;
    fm.Register(po);
    po->Register("num-ceps", &c.num_ceps, "Number of cepstra in MFCC computation (including C0)");
    po->Register("use-energy", &use_energy, "Use energy (not C0) in MFCC computation");
    po->Register("energy-floor", &c.energy_floor, "Floor on energy (absolute, not relative) in MFCC computation");
    po->Register("raw-energy", &raw_energy, "If true, compute energy before preemphasis and windowing");
    po->Register("cepstral-lifter", &c.cepstral_lifter, "Constant that controls scaling of MFCCs");
    po->Register("htk-compat", &htk_compat, "If true, put energy or C0 last and use a factor of sqrt(2) on C0.");
  }
  void Finish() {
    fm.Finish();
    kamd_mfcc_opts &c = opts->c;
    c.use_energy = use_energy; c.raw_energy = raw_energy; c.htk_compat = htk_compat;
  }
};
struct FbankOptionsParser {
  FbankOptions *opts;
  FrameMelOptionsParser fm;
  bool use_energy, raw_energy, htk_compat, use_log_fbank, use_power;
  explicit FbankOptionsParser(FbankOptions *o)
      : opts(o), fm(&o->c.frame, &o->c.mel), use_energy(o->c.use_energy != 0), raw_energy(o->c.raw_energy != 0), htk_compat(o->c.htk_compat != 0),
        use_log_fbank(o->c.use_log_fbank != 0), use_power(o->c.use_power != 0) {}
  void Register(ParseOptions *po) {
    kamd_fbank_opts &c = opts->c;
    fm.Register(po);
    po->Register("use-energy", &use_energy, "Add an extra dimension with energy to the FBANK output.");
    po->Register("energy-floor", &c.energy_floor, "Floor on energy (absolute, not relative) in FBANK computation");
    po->Register("raw-energy", &raw_energy, "If true, compute energy before preemphasis and windowing");
    po->Register("htk-compat", &htk_compat, "If true, put energy last.");
    po->Register("use-log-fbank", &use_log_fbank, "If true, produce log-filterbank, else produce linear.");
    po->Register("use-power", &use_power, "If true, use power, else use magnitude.");
  }
  void Finish() {
    fm.Finish();
    kamd_fbank_opts &c = opts->c;
    c.use_energy = use_energy; c.raw_energy = raw_energy; c.htk_compat = htk_compat; c.use_log_fbank = use_log_fbank; c.use_power = use_power;
  }
};

// ---- "key rxfilename" lines of a wav.scp (an `scp:` rspecifier; the rxfilename may be an input pipe) and
// WaveHolder::Read of one entry (feat/wave-reader.h:150-200), channel 0
struct WaveScp {
  std::vector<std::pair<std::string, std::string> > entries;
  explicit WaveScp(const std::string &rspecifier) {
    char rx[4096], path[4096]; int opts = 0, temp = 0; int64_t off = 0;
    if (kamd_classify_rspecifier(rspecifier.c_str(), rx, sizeof(rx), &opts) != 2)
      throw KaldiFatalError("expected an scp: rspecifier of \"key wav-rxfilename\" lines, got " + rspecifier);
    Check(kamd_rx_materialize(rx, path, sizeof(path), &off, &temp));
    FILE *f = fopen(path, "r");
    if (!f) throw KaldiFatalError(std::string("cannot open ") + path);
    char line[8192];
    while (fgets(line, sizeof(line), f)) {
      std::string l(line);
      while (!l.empty() && (l.back() == '\n' || l.back() == '\r' || l.back() == ' ')) l.pop_back();
      const size_t sp = l.find_first_of(" \t");
      if (l.empty()) continue;
      if (sp == std::string::npos) { fclose(f); throw KaldiFatalError("Invalid line in script file: \"" + l + "\""); }
      entries.push_back(std::make_pair(l.substr(0, sp), l.substr(l.find_first_not_of(" \t", sp))));
    }
    fclose(f);
    if (temp) remove(path);
  }
  /// the rxfilename of `key`, or NULL
  const std::string *Find(const std::string &key) const {
    for (size_t i = 0; i < entries.size(); i++) if (entries[i].first == key) return &entries[i].second;
    return NULL;
  }
  static void Read(const std::string &rxfilename, float expect_freq, std::vector<float> *samples) {
    char path[4096]; int temp = 0; int64_t off = 0;
    Check(kamd_rx_materialize(rxfilename.c_str(), path, sizeof(path), &off, &temp));
    float sf = 0; int32_t nch = 0; int64_t n = 0; float *p = NULL;
    const int rc = kamd_wave_read(path, &sf, &nch, &n, &p);
    if (temp) remove(path);
    Check(rc);
    if (sf != expect_freq) { kamd_host_free(p); throw KaldiFatalError(rxfilename + ": sampling rate " + std::to_string(sf) + ", the feature config expects " + std::to_string(expect_freq)); }
    samples->assign(p, p + n);                             // channel 0
    kamd_host_free(p);
  }
};

}  // namespace kaldi_amd
#endif  // KALDI_AMD_HPP_
