// Exercises the C++ host mirror (include/kaldi_amd.hpp) the way Kaldi code drives
// LatticeFasterDecoder: Decode(DecodableInterface*) with (1) DecodableMatrixMapped and
// (2) an arbitrary DecodableInterface subclass.  Input: a fixture file written by
// tests/test_gpu_cxx_host.py.  Output: one line per decode, compared with the oracle there.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kaldi_amd.hpp"

using namespace kaldi_amd;

template <typename T> static std::vector<T> ReadVec(FILE *f) {
  int64_t n;
  if (fread(&n, 8, 1, f) != 1) { fprintf(stderr, "bad fixture\n"); exit(2); }
  std::vector<T> v(n);
  if (n && fread(v.data(), sizeof(T), n, f) != static_cast<size_t>(n)) { fprintf(stderr, "bad fixture\n"); exit(2); }
  return v;
}

// a decodable the decoder knows nothing about (like DecodableAmDiagGmmScaled would be)
class ScaledDecodable : public DecodableInterface {
 public:
  ScaledDecodable(const std::vector<int32> &id2pdf, const std::vector<float> &ll, int32 T, int32 P)
      : id2pdf_(id2pdf), ll_(ll), T_(T), P_(P) {}
  BaseFloat LogLikelihood(int32 frame, int32 tid) override { return ll_[static_cast<size_t>(frame) * P_ + id2pdf_[tid]]; }
  bool IsLastFrame(int32 frame) const override { return frame == T_ - 1; }
  int32 NumFramesReady() const override { return T_; }
  int32 NumIndices() const override { return static_cast<int32>(id2pdf_.size()) - 1; }
 private:
  const std::vector<int32> &id2pdf_;
  const std::vector<float> &ll_;
  int32 T_, P_;
};

static void Report(const char *tag, LatticeFasterDecoder &dec) {
  std::vector<int32> ali, words;
  BaseFloat g, a;
  bool ok = dec.GetBestPath(&ali, &words, &g, &a);
  Lattice lat;
  dec.GetRawLattice(&lat);
  size_t narcs = 0;
  for (size_t s = 0; s < lat.arcs.size(); s++) narcs += lat.arcs[s].size();
  printf("%s ok=%d frames=%d reached_final=%d states=%d arcs=%zu graph=%.9g acoustic=%.9g words=", tag, ok,
         dec.NumFramesDecoded(), dec.ReachedFinal(), lat.NumStates(), narcs, g, a);
  for (size_t i = 0; i < words.size(); i++) printf("%d%s", words[i], i + 1 < words.size() ? "," : "");
  printf("\n");
}

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  std::vector<int64_t> hdr = ReadVec<int64_t>(f);     // num_states, start, T, P
  std::vector<int64_t> arc_off = ReadVec<int64_t>(f);
  std::vector<kamd_arc> arcs = ReadVec<kamd_arc>(f);
  std::vector<float> final_cost = ReadVec<float>(f);
  std::vector<int32> id2pdf = ReadVec<int32>(f);
  std::vector<float> ll = ReadVec<float>(f);
  fclose(f);
  const int32 S = static_cast<int32>(hdr[0]), start = static_cast<int32>(hdr[1]), T = static_cast<int32>(hdr[2]),
              P = static_cast<int32>(hdr[3]);
  try {
    DecodingGraph fst(S, start, arc_off.data(), arcs.data(), final_cost.data());
    LatticeFasterDecoderConfig config;
    config.beam = 15.0; config.max_active = 7000; config.min_active = 200; config.lattice_beam = 8.0;
    kamd_decoder_sizes sz;
    kamd_decoder_sizes_default(&sz);
    sz.max_lanes = 1; sz.hash_capacity = 1 << 14; sz.arena_tokens = 1 << 18; sz.arena_links = 1 << 19; sz.max_frames = 1024;
    {
      LatticeFasterDecoder decoder(fst, config, id2pdf, &sz);
      DecodableMatrixMapped decodable(id2pdf, ll.data(), T, P);
      decoder.Decode(&decodable);
      Report("mapped", decoder);
      // chunked AdvanceDecoding, as online2 drives it
      decoder.InitDecoding();
      for (int32 t = 0; t < T; t += 7) {
        DecodableMatrixMapped part(id2pdf, ll.data(), std::min(T, t + 7), P);
        decoder.AdvanceDecoding(&part);
        if (t % 14 == 0) decoder.PruneActiveTokens();     // never changes the result
      }
      decoder.FinalizeDecoding();
      Report("chunked", decoder);
    }
    {
      LatticeFasterDecoder decoder(fst, config, std::vector<int32>(), &sz);
      ScaledDecodable decodable(id2pdf, ll, T, P);
      decoder.Decode(&decodable);
      Report("generic", decoder);
    }
    // the nnet3-latgen-faster tail: DecodeUtteranceLatticeFaster with and without determinization,
    // graph read back from an OpenFst file
    if (argc >= 3) {
      const std::string dir = argv[2];
      Check(kamd_openfst_write((dir + "/HCLG.fst").c_str(), 1, 0, S, start, arc_off.data(), arcs.data(), final_cost.data()));
      DecodingGraph fst2(dir + "/HCLG.fst");
      LatticeFasterDecoder decoder(fst2, config, id2pdf, &sz);
      std::vector<int32> tid_phone(id2pdf.size(), 0);
      for (size_t t = 1; t < tid_phone.size(); t += 2) tid_phone[t] = static_cast<int32>((t + 1) / 2);
      Int32VectorWriter words_writer("ark,t:" + dir + "/words.txt"), ali_writer("ark:" + dir + "/ali.ark");
      CompactLatticeWriter clat_writer("ark:" + dir + "/clat.ark");
      LatticeWriter lat_writer("ark,t:" + dir + "/lat.txt");
      double like = 0;
      for (int rep = 0; rep < 2; rep++) {
        DecodableMatrixMapped decodable(id2pdf, ll.data(), T, P);
        const bool ok = DecodeUtteranceLatticeFaster(decoder, decodable, tid_phone, rep ? "utt-raw" : "utt-det", 0.5, rep == 0, true,
                                                     &ali_writer, &words_writer, &clat_writer, &lat_writer, &like);
        printf("wrapper ok=%d like=%.6g\n", ok, like);
      }
      // lattice-lmrescore-const-arpa on the determinized lattice (G.arpa: integer word ids, <s> = 100001, </s> = 100002)
      FILE *gf = fopen((dir + "/G.arpa").c_str(), "r");
      if (gf) {
        fclose(gf);
        ConstArpaLm built, lm;
        built.Build(dir + "/G.arpa", 100001, 100002, -1);
        built.Write(dir + "/G.carpa");
        lm.Read(dir + "/G.carpa");
        DecodableMatrixMapped decodable(id2pdf, ll.data(), T, P);
        decoder.Decode(&decodable);
        Lattice raw; CompactLattice clat, rescored, same;
        decoder.GetRawLattice(&raw);
        DeterminizeLatticePhonePrunedWrapper(tid_phone, raw, config.lattice_beam, &clat);
        const bool ok = LatticeLmrescoreConstArpa(1.0f, lm, clat, &rescored) && LatticeLmrescoreConstArpa(0.0f, lm, clat, &same);
        CompactLatticeWriter in_writer("ark:" + dir + "/carpa_in.ark"), out_writer("ark:" + dir + "/carpa_out.ark");
        in_writer.Write("utt", clat);
        out_writer.Write("utt", rescored);
        std::vector<int32> hist; hist.push_back(100001);
        printf("carpa ok=%d order=%d bos=%d eos=%d states=%d copy_states=%d p=%.9g\n", ok, lm.NgramOrder(), lm.BosSymbol(), lm.EosSymbol(),
               rescored.NumStates(), same.NumStates() - clat.NumStates(), lm.GetNgramLogprob(1, hist));
      }
    }
    // OnlineStreamBatch: two streams (one of them with online i-vectors from the third fixture's extractor, which
    // reads the first 8 cepstra) against the Python mirror driving the same C-ABI (argv[3] model, argv[4] extractor)
    if (argc >= 6) {
      FILE *mf = fopen(argv[3], "rb"), *xf = fopen(argv[4], "rb");
      if (!mf || !xf) return 2;
      std::vector<int64_t> mh = ReadVec<int64_t>(mf);
      const int nl = static_cast<int>(mh[0]);
      std::vector<kamd_layer_desc> layers(nl);
      std::vector<std::vector<float> > keep(5 * nl);
      for (int l = 0; l < nl; l++) {
        std::vector<int32> li = ReadVec<int32>(mf);
        std::vector<float> lf = ReadVec<float>(mf);
        kamd_layer_desc &d = layers[l];
        memset(&d, 0, sizeof(d));
        d.in_dim = li[0]; d.out_dim = li[1]; d.n_offsets = li[2];
        for (int k = 0; k < 8; k++) d.offsets[k] = li[3 + k];
        d.input_layer = li[11]; d.ivector_dim = li[12]; d.bypass_layer = li[13]; d.relu = li[14]; d.log_softmax = li[15];
        d.bypass_scale = lf[0]; d.post_scale = lf[1];
        for (int k = 0; k < 5; k++) keep[5 * l + k] = ReadVec<float>(mf);
      }
      for (int l = 0; l < nl; l++) {
        const float **slots[5] = {&layers[l].W, &layers[l].bias, &layers[l].bn_scale, &layers[l].bn_offset, &layers[l].post_offset};
        for (int k = 0; k < 5; k++) *slots[k] = keep[5 * l + k].empty() ? NULL : keep[5 * l + k].data();
      }
      std::vector<float> wave = ReadVec<float>(mf);
      fclose(mf);
      std::vector<int32> ih = ReadVec<int32>(xf);
      std::vector<float> lda = ReadVec<float>(xf), gc = ReadVec<float>(xf), miv = ReadVec<float>(xf), iv = ReadVec<float>(xf);
      std::vector<double> gstats = ReadVec<double>(xf), M = ReadVec<double>(xf), sinv = ReadVec<double>(xf), sc = ReadVec<double>(xf);
      fclose(xf);
      kamd_ivector_desc d;
      memset(&d, 0, sizeof(d));
      d.feat_dim = ih[0]; d.splice_left = ih[1]; d.splice_right = ih[2]; d.lda_rows = ih[3]; d.lda_cols = ih[4]; d.lda = lda.data();
      d.global_cmvn_stats = gstats.data(); d.cmn_window = ih[12]; d.speaker_frames = ih[13]; d.global_frames = ih[14];
      d.normalize_mean = 1; d.num_gauss = ih[5]; d.ubm_gconsts = gc.data(); d.ubm_means_invvars = miv.data(); d.ubm_inv_vars = iv.data();
      d.ivector_dim = ih[6]; d.M = M.data(); d.sigma_inv = sinv.data(); d.prior_offset = sc[0];
      d.ivector_period = ih[7]; d.num_gselect = ih[8]; d.num_cg_iters = ih[9];
      d.min_post = static_cast<float>(sc[1]); d.posterior_scale = static_cast<float>(sc[2]); d.max_count = static_cast<float>(sc[3]);
      OnlineIvectorExtractor extractor(d, 60.0);
      AmNnetSimple am(layers, static_cast<int>(mh[1]), static_cast<int>(mh[2]));
      MfccOptions mo;
      kamd_mfcc_opts_default(&mo.c);
      mo.c.use_energy = 0; mo.c.mel.num_bins = 40; mo.c.num_ceps = 40; mo.c.mel.low_freq = 20.0f; mo.c.mel.high_freq = -400.0f;
      kamd_decoder_sizes bs = sz;
      bs.max_lanes = 2;
      {   // NnetBatchDecoder as nnet3-latgen-faster-batch drives it: AcceptInput..., Finished(), GetOutput until false
        NnetBatchDecoderOptions bo;
        bo.c.resident_lanes = 2; bo.search_mode = 1;
        NnetBatchDecoder bd(fst, config, id2pdf, std::vector<int32>(), NULL, true, 2, am, &mo, bs, bo);
        std::vector<float> w2(wave.begin(), wave.begin() + wave.size() * 2 / 3);
        bd.AcceptWaveform("utt0", wave);
        bd.AcceptWaveform("utt1", w2);
        const int32 n_ok = bd.Finished();
        std::string key, sentence; CompactLattice clat;
        int u = 0;
        while (bd.GetOutput(&key, &clat, &sentence)) {
          std::vector<int32> ali, words; BaseFloat g = 0, a = 0;
          const bool ok = bd.GetBestPath(&ali, &words, &g, &a) && n_ok == 2 && key == (u ? "utt1" : "utt0") && clat.NumStates() > 0 && sentence.empty();
          printf("offline utt=%d ok=%d frames=%d cost=%.9g words=", u, ok, static_cast<int>(ali.size()), g + a);
          for (size_t k = 0; k < words.size(); k++) printf("%s%d", k ? "," : "", words[k]);
          printf("\n");
          u++;
        }
        bool threw = false;
        try { Lattice l; bd.GetOutput(&key, &l, &sentence); } catch (const KaldiFatalError &) { threw = true; }
        if (!threw) throw KaldiFatalError("GetOutput(Lattice) must throw when determinizing");
      }
      OnlineStreamBatch batch(config, id2pdf, am, fst, mo, 2, 4.0f, bs);
      batch.SetIvectorExtractor(extractor.handle(), 20, d.splice_right);
      std::vector<int32> both = {0, 1};
      batch.Start(both);
      const size_t step[2] = {2880, 4960};
      size_t pos[2] = {0, 0};
      const size_t len[2] = {wave.size(), wave.size() * 2 / 3};
      while (pos[0] < len[0] || pos[1] < len[1]) {
        std::vector<int32> live;
        for (int s = 0; s < 2; s++) {
          if (pos[s] >= len[s]) continue;
          const size_t n = std::min(step[s], len[s] - pos[s]);
          batch.AcceptWaveform(s, wave.data() + pos[s], static_cast<int64_t>(n), pos[s] + n >= len[s]);
          pos[s] += n; live.push_back(s);
        }
        batch.AdvanceDecoding(live);
        // partial results of the live streams in one launch: one transition-id per decoded frame
        std::vector<std::vector<int32> > pw, pa; std::vector<char> pok;
        batch.GetPartialBestPaths(live, &pw, &pa, &pok);
        for (size_t k = 0; k < live.size(); k++) {
          const int32 nd = kamd_decoder_num_frames_decoded(batch.DecoderHandle(), live[k]);
          if (nd > 0 && (!pok[k] || static_cast<int32>(pa[k].size()) != nd)) throw KaldiFatalError("partial best path: alignment length != frames decoded");
        }
      }
      batch.FinalizeDecoding(both);
      for (int s = 0; s < 2; s++) {
        std::vector<int32> ali, words; BaseFloat g = 0, a = 0;
        const bool ok = batch.GetBestPath(s, &ali, &words, &g, &a);
        std::vector<double> st;
        batch.GetAdaptationState(s, 60.0f, &st);
        printf("batch stream=%d ok=%d frames=%d cost=%.9g count=%.9g lin1=%.9g words=", s, ok, static_cast<int>(ali.size()), g + a, st[d.feat_dim],
               st[2 * (d.feat_dim + 1) + d.ivector_dim * (d.ivector_dim + 1) / 2 + 1]);
        for (size_t k = 0; k < words.size(); k++) printf("%s%d", k ? "," : "", words[k]);
        printf("\n");
      }
    }
    // online i-vectors with the speaker's adaptation state: two utterances of one speaker (third fixture)
    if (argc >= 5) {
      FILE *xf = fopen(argv[4], "rb");
      if (!xf) return 2;
      std::vector<int32> ih = ReadVec<int32>(xf);   // feat_dim L R lda_rows lda_cols G I period ng cg T1 T2 cmn_window speaker_frames global_frames
      std::vector<float> lda = ReadVec<float>(xf), gc = ReadVec<float>(xf), miv = ReadVec<float>(xf), iv = ReadVec<float>(xf);
      std::vector<double> gstats = ReadVec<double>(xf), M = ReadVec<double>(xf), sinv = ReadVec<double>(xf), sc = ReadVec<double>(xf);
      std::vector<float> f1 = ReadVec<float>(xf), f2 = ReadVec<float>(xf);
      fclose(xf);
      kamd_ivector_desc d;
      memset(&d, 0, sizeof(d));
      d.feat_dim = ih[0]; d.splice_left = ih[1]; d.splice_right = ih[2]; d.lda_rows = ih[3]; d.lda_cols = ih[4]; d.lda = lda.data();
      d.global_cmvn_stats = gstats.data(); d.cmn_window = ih[12]; d.speaker_frames = ih[13]; d.global_frames = ih[14];
      d.normalize_mean = 1; d.num_gauss = ih[5]; d.ubm_gconsts = gc.data(); d.ubm_means_invvars = miv.data(); d.ubm_inv_vars = iv.data();
      d.ivector_dim = ih[6]; d.M = M.data(); d.sigma_inv = sinv.data(); d.prior_offset = sc[0];
      d.ivector_period = ih[7]; d.num_gselect = ih[8]; d.num_cg_iters = ih[9];
      d.min_post = static_cast<float>(sc[1]); d.posterior_scale = static_cast<float>(sc[2]); d.max_count = static_cast<float>(sc[3]);
      OnlineIvectorExtractor extractor(d, 60.0);
      OnlineIvectorExtractorAdaptationState state;
      std::vector<float> out;
      extractor.ExtractOnline(f1.data(), ih[10], &out, &state);
      extractor.ExtractOnline(f2.data(), ih[11], &out, &state);
      printf("ivector rows=%d dim=%d", extractor.NumIvectors(ih[11]), extractor.Dim());
      for (int k = 0; k < extractor.Dim(); k++) printf(" %.9g", out[out.size() - extractor.Dim() + k]);
      printf("\n");
    }
    // streaming: SingleUtteranceNnet3Decoder fed in 0.18 s chunks (model + waveform from a second fixture)
    if (argc >= 4) {
      FILE *mf = fopen(argv[3], "rb");
      if (!mf) return 2;
      std::vector<int64_t> mh = ReadVec<int64_t>(mf);        // n_layers, input_dim, subsampling
      const int nl = static_cast<int>(mh[0]);
      std::vector<kamd_layer_desc> layers(nl);
      std::vector<std::vector<float> > keep;
      for (int l = 0; l < nl; l++) {
        std::vector<int32> li = ReadVec<int32>(mf);          // in_dim out_dim n_off off[8] input_layer ivector_dim bypass_layer relu log_softmax
        std::vector<float> lf = ReadVec<float>(mf);          // bypass_scale post_scale
        kamd_layer_desc &d = layers[l];
        memset(&d, 0, sizeof(d));
        d.in_dim = li[0]; d.out_dim = li[1]; d.n_offsets = li[2];
        for (int k = 0; k < 8; k++) d.offsets[k] = li[3 + k];
        d.input_layer = li[11]; d.ivector_dim = li[12]; d.bypass_layer = li[13]; d.relu = li[14]; d.log_softmax = li[15];
        d.bypass_scale = lf[0]; d.post_scale = lf[1];
        const float **slots[5] = {&d.W, &d.bias, &d.bn_scale, &d.bn_offset, &d.post_offset};
        for (int k = 0; k < 5; k++) {
          keep.push_back(ReadVec<float>(mf));
          *slots[k] = keep.back().empty() ? NULL : keep.back().data();
        }
      }
      // (vectors inside 'keep' may have moved while it grew: re-point)
      for (int l = 0; l < nl; l++) {
        const float **slots[5] = {&layers[l].W, &layers[l].bias, &layers[l].bn_scale, &layers[l].bn_offset, &layers[l].post_offset};
        for (int k = 0; k < 5; k++) *slots[k] = keep[5 * l + k].empty() ? NULL : keep[5 * l + k].data();
      }
      std::vector<float> wave = ReadVec<float>(mf);
      fclose(mf);
      AmNnetSimple am(layers, static_cast<int32>(mh[1]), static_cast<int32>(mh[2]));
      MfccOptions mo;
      mo.c.frame.dither = 0.0f; mo.c.use_energy = 0; mo.c.mel.num_bins = 40; mo.c.num_ceps = 40; mo.c.mel.low_freq = 20; mo.c.mel.high_freq = -400;
      SingleUtteranceNnet3Decoder sdec(config, id2pdf, am, fst, mo, &sz);
      const size_t chunk = 2880;
      int partials = 0, live_lattices = 0;
      // endpointing: options through ParseOptions like online2-wav-nnet3-latgen-faster; every unit counts as silence
      // here (the walk goes all the way back; the stop at the first non-silence frame is covered by test_gpu_online.py)
      OnlineEndpointConfig ep;
      std::vector<int32> tid2phone(id2pdf.size(), 0);
      for (size_t t = 1; t < tid2phone.size(); t++) tid2phone[t] = static_cast<int32>((t - 1) / 2 + 1);
      std::string sil;
      for (int32 p = 1; p <= tid2phone.back(); p++)
        sil += (sil.empty() ? "" : ":") + std::to_string(p);
      {
        ParseOptions epo("endpoint options");
        ep.Register(&epo);
        const std::string a1 = "--endpoint.silence-phones=" + sil;
        const char *eargv[] = {"x", a1.c_str(), "--endpoint.rule3.min-trailing-silence=0.03", "--endpoint.rule3.max-relative-cost=inf",
                               "--endpoint.rule2.must-contain-nonsilence=false", "--endpoint.rule2.max-relative-cost=inf"};
        epo.Read(6, eargv);
      }
      std::string ep_flags, ep_sil;
      for (size_t i = 0; i < wave.size(); i += chunk) {
        std::vector<float> part(wave.begin() + i, wave.begin() + std::min(wave.size(), i + chunk));
        sdec.AcceptWaveform(16000.0f, part);
        if (i + chunk >= wave.size()) sdec.InputFinished();
        sdec.AdvanceDecoding();
        std::vector<int32> ali, words; BaseFloat g, a;
        if (sdec.NumFramesDecoded() > 0 && sdec.GetBestPath(false, &ali, &words, &g, &a)) partials++;
        // GetLattice(end_of_utterance = false) on the live decoder (online-nnet3-decoding.cc:66-79) + GetRawLatticePruned
        if (sdec.NumFramesDecoded() > 0) {
          CompactLattice partial_clat;
          sdec.GetLattice(false, std::vector<int32>(), &partial_clat);
          int32 pn = 0, pm = 0, pk = 0, pst = 0, pok = 0;
          kamd_compact_lattice_sizes(partial_clat.Handle(), &pn, &pm, &pk, &pst, &pok);
          Lattice raw, pruned;
          sdec.Decoder().GetRawLattice(&raw, false);
          sdec.Decoder().GetRawLatticePruned(&pruned, false, 2.0f);
          if (pn <= 0 || raw.NumStates() <= 0 || pruned.NumStates() <= 0 || pruned.NumStates() > raw.NumStates())
            throw KaldiFatalError("partial lattice of the live decoder is empty or the pruned one larger than the full one");
          live_lattices++;
        }
        std::vector<int32> det, tsf;
        EndpointDetected(ep, tid2phone, 0.03f, sdec.Decoder().Handle(), std::vector<int32>(1, 0), &det, &tsf);
        if (det[0] != (sdec.EndpointDetected(ep, tid2phone) ? 1 : 0)) throw KaldiFatalError("the two endpointing calls disagree");
        ep_flags += (ep_flags.empty() ? "" : ",") + std::to_string(det[0]);
        ep_sil += (ep_sil.empty() ? "" : ",") + std::to_string(tsf[0]);
      }
      printf("endpoint flags=%s silence=%s plain=%d%d\n", ep_flags.c_str(), ep_sil.c_str(), EndpointDetected(ep, 100, 17, 0.03f, 1.9f) ? 1 : 0,
             EndpointDetected(ep, 100, 1, 0.03f, 9.0f) ? 1 : 0);
      sdec.FinalizeDecoding();
      std::vector<int32> ali, words; BaseFloat g = 0, a = 0;
      const bool ok = sdec.GetBestPath(true, &ali, &words, &g, &a);
      printf("live lattices=%d\n", live_lattices > 0);
      printf("streaming ok=%d frames=%d partials=%d graph=%.9g acoustic=%.9g words=", ok, sdec.NumFramesDecoded(), partials > 0, g, a);
      for (size_t i = 0; i < words.size(); i++) printf("%d%s", words[i], i + 1 < words.size() ? "," : "");
      printf("\n");
    }
    // error convention: a bad config throws like KALDI_ERR
    try {
      LatticeFasterDecoderConfig bad; bad.beam = -1;
      LatticeFasterDecoder d2(fst, bad, id2pdf, &sz);
      printf("badconfig no-throw\n");
    } catch (const KaldiFatalError &e) { printf("badconfig threw\n"); }
  } catch (const std::exception &e) {
    fprintf(stderr, "FATAL: %s\n", e.what());
    return 1;
  }
  return 0;
}
