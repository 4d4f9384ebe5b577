// online2-wav-nnet3-latgen-faster (online2bin/online2-wav-nnet3-latgen-faster.cc:60-300) as a C++ host program over the
// C-ABI and the kaldi_amd.hpp mirror (BASELINE configs[4]): the reference's usage line, option names (online.conf with
// --feature-type, --mfcc-config / --fbank-config, --ivector-extraction-config, --endpoint.*, --ivector-silence-weighting.*) and log
// lines; chunked features, online i-vectors, the acoustic model and the lattice-generating search run on the MI355X.
//
//   online2-wav-nnet3-latgen-faster-amd [options] <nnet3-in> <fst-in> <spk2utt-rspecifier> <wav-rspecifier> <lattice-wspecifier>
//   e.g.  online2-wav-nnet3-latgen-faster-amd --config=conf/online.conf --do-endpointing=false --frames-per-chunk=20
//             --acoustic-scale=1.0 --frame-subsampling-factor=3 final.mdl HCLG.fst ark:spk2utt scp:wav.scp "ark:|gzip -c > lat.1.gz"
//
// How it runs: one stream per speaker, --batch speakers at once (OnlineStreamBatch: stream s = decoder lane s).  Audio is
// fed --chunk-length seconds at a time; every tick uploads the chunks of ALL active streams with one copy and advances them
// together (features, i-vectors, the model on the frames that became computable, the search).  A speaker's utterances
// follow each other and hand their i-vector adaptation state on (:183-191, :285-286).  The same archive, byte for byte,
// as tools/online2_wav_nnet3_latgen_faster.py (tests/test_gpu_cxx_host.py).  --feature-type=mfcc or fbank; not supported: plp and pitch features.
// Build: g++ -std=c++14 -O2 -I include examples/online2_wav_nnet3_latgen_faster.cc -L kaldi_amd/lib -lkaldi_amd -lpthread
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "kaldi_amd.hpp"

using namespace kaldi_amd;

namespace {

// "spk utt1 utt2 ..." lines of a text archive (TokenVectorHolder), in file order
std::vector<std::pair<std::string, std::vector<std::string> > > ReadSpk2Utt(const std::string &rspecifier) {
  char rx[4096], path[4096]; int opts = 0, temp = 0; int64_t off = 0;
  if (kamd_classify_rspecifier(rspecifier.c_str(), rx, sizeof(rx), &opts) != 1)
    throw KaldiFatalError("the spk2utt rspecifier must be a text archive, got " + rspecifier);
  Check(kamd_rx_materialize(rx, path, sizeof(path), &off, &temp));
  FILE *f = fopen(path, "r");
  if (!f) throw KaldiFatalError(std::string("cannot open ") + path);
  std::vector<std::pair<std::string, std::vector<std::string> > > out;
  char line[65536];
  while (fgets(line, sizeof(line), f)) {
    std::vector<std::string> tok;
    for (char *p = strtok(line, " \t\r\n"); p; p = strtok(NULL, " \t\r\n")) tok.push_back(p);
    if (tok.empty()) continue;
    out.push_back(std::make_pair(tok[0], std::vector<std::string>(tok.begin() + 1, tok.end())));
  }
  fclose(f);
  if (temp) remove(path);
  return out;
}

struct Active {              // the utterance a stream is decoding
  std::string key;
  std::vector<float> wave;
  size_t pos;
};

}  // namespace

int main(int argc, char **argv) {
  try {
    ParseOptions po("Reads in wav file(s) and simulates online decoding with neural nets\n"
                    "(nnet3 setup), with optional iVector-based speaker adaptation and\n"
                    "optional endpointing.  Note: some configuration values and inputs are\n"
                    "set via config files whose filenames are passed as options\n\n"
                    "Usage: online2-wav-nnet3-latgen-faster-amd [options] <nnet3-in> <fst-in> "
                    "<spk2utt-rspecifier> <wav-rspecifier> <lattice-wspecifier>\n"
                    "The spk2utt-rspecifier can just be <utterance-id> <utterance-id> if\n"
                    "you want to decode utterance by utterance.\n");
    LatticeFasterDecoderConfig config;
    DeterminizeLatticePhonePrunedOptions det_opts;
    bool phone_determinize = true, word_determinize = true, minimize = false, debug_computation = false;
    int32 max_mem = 50000000;
    // LatticeFasterDecoderConfig::Register (decoder/lattice-faster-decoder.h:65-88)
    po.Register("beam", &config.beam, "Decoding beam.  Larger->slower, more accurate.");
    po.Register("max-active", &config.max_active, "Decoder max active states.  Larger->slower; more accurate");
    po.Register("min-active", &config.min_active, "Decoder minimum #active states.");
    po.Register("lattice-beam", &config.lattice_beam, "Lattice generation beam.  Larger->slower, and deeper lattices");
    po.Register("prune-interval", &config.prune_interval, "Interval (in frames) at which to prune tokens");
    po.Register("determinize-lattice", &config.determinize_lattice, "(ignored: the lattices of this program are always determinized, :279-283)");
    po.Register("beam-delta", &config.beam_delta, "Increment used in decoding-- this parameter is obscure and relates to a speedup in the way "
                "the max-active constraint is applied.  Larger is more accurate.");
    po.Register("hash-ratio", &config.hash_ratio, "(ignored: the device token table is sized by the arenas)");
    po.Register("prune-scale", &config.prune_scale, "(ignored: lattice pruning is exact here)");
    po.Register("delta", &det_opts.c.delta, "Tolerance used in determinization");
    po.Register("max-mem", &max_mem, "Maximum approximate memory usage in determinization (real usage might be many times this).");
    po.Register("phone-determinize", &phone_determinize, "If true, do an initial pass of determinization on both phones and words");
    po.Register("word-determinize", &word_determinize, "If true, do a second pass of determinization on words only");
    po.Register("minimize", &minimize, "(ignored: as in the reference's default, lattices are not minimized)");
    // NnetSimpleLoopedComputationOptions::Register (nnet3/decodable-simple-looped.h:52-90)
    int32 extra_left_context = 0, extra_right_context = 0, extra_left_context_initial = -1, extra_right_context_final = -1;
    int32 frame_subsampling_factor = 1, frames_per_chunk = 20;
    BaseFloat acoustic_scale = 0.1f;
    po.Register("extra-left-context", &extra_left_context, "(ignored: TDNN models need no extra context)");
    po.Register("extra-right-context", &extra_right_context, "(ignored)");
    po.Register("extra-left-context-initial", &extra_left_context_initial, "(ignored)");
    po.Register("extra-right-context-final", &extra_right_context_final, "(ignored)");
    po.Register("frame-subsampling-factor", &frame_subsampling_factor, "Required if the frame-rate of the output (e.g. in 'chain' models) is "
                "less than the frame-rate of the original alignment.");
    po.Register("acoustic-scale", &acoustic_scale, "Scaling factor for acoustic log-likelihoods");
    po.Register("frames-per-chunk", &frames_per_chunk, "Number of frames in each chunk that is separately evaluated by the neural net; "
                "with i-vectors also their period");
    po.Register("debug-computation", &debug_computation, "(ignored)");
    // online2-wav-nnet3-latgen-faster.cc:100-125
    BaseFloat chunk_length_secs = 0.18f, max_seconds = 60.0f;
    bool do_endpointing = false, online = true;
    int32 num_threads_startup = 8, batch = 64, device = -1;
    std::string word_syms_rxfilename, feature_type = "mfcc", mfcc_config, fbank_config, ivector_config;
    po.Register("chunk-length", &chunk_length_secs, "Length of chunk size in seconds, that we process.  Set to <= 0 to use all input in one chunk.");
    po.Register("word-symbol-table", &word_syms_rxfilename, "Symbol table for words [for debug output]");
    po.Register("do-endpointing", &do_endpointing, "If true, apply endpoint detection");
    po.Register("online", &online, "(ignored: decoding is always chunk by chunk)");
    po.Register("num-threads-startup", &num_threads_startup, "(ignored)");
    // OnlineNnet2FeaturePipelineConfig::Register (online2/online-nnet2-feature-pipeline.h:89-110)
    po.Register("feature-type", &feature_type, "Base feature type [mfcc, fbank]");
    po.Register("mfcc-config", &mfcc_config, "Configuration file for MFCC features (e.g. conf/mfcc_hires.conf)");
    po.Register("fbank-config", &fbank_config, "Configuration file for filterbank features (e.g. conf/fbank.conf)");
    po.Register("ivector-extraction-config", &ivector_config, "Configuration file for online iVector extraction");
    OnlineSilenceWeightingConfig silence_weighting_config;
    silence_weighting_config.RegisterWithPrefix("ivector-silence-weighting", &po);
    OnlineEndpointConfig endpoint_opts;
    endpoint_opts.Register(&po);
    po.Register("batch", &batch, "Speakers decoded concurrently");
    po.Register("max-seconds", &max_seconds, "Longest utterance the stream slots are sized for");
    po.Register("device", &device, "HIP device to run on");
    po.Read(argc, argv);
    if (po.NumArgs() != 5) { po.PrintUsage(); return 1; }
    const std::string nnet3_rxfilename = po.GetArg(1), fst_rxfilename = po.GetArg(2), spk2utt_rspecifier = po.GetArg(3),
                      wav_rspecifier = po.GetArg(4), clat_wspecifier = po.GetArg(5);
    if (feature_type != "mfcc" && feature_type != "fbank")       // online-nnet2-feature-pipeline.cc:36-58 (plp and pitch: not on this path)
      throw KaldiFatalError("Invalid feature type: " + feature_type + " (supported: mfcc, fbank)");
    if (do_endpointing && endpoint_opts.SilencePhones().empty()) throw KaldiFatalError("--do-endpointing needs --endpoint.silence-phones");
    config.Check();
    if (device >= 0) Check(kamd_set_device(device));
    det_opts.c.max_mem = max_mem; det_opts.c.phone_determinize = phone_determinize; det_opts.c.word_determinize = word_determinize;

    MfccOptions mfcc;
    FbankOptions fbank;
    const bool use_fbank = feature_type == "fbank";
    if (use_fbank) {
      if (!fbank_config.empty()) {
        ParseOptions fpo("fbank config");
        FbankOptionsParser fp(&fbank);
        fp.Register(&fpo);
        fpo.ReadConfigFile(fbank_config);
        fp.Finish();
      }
    } else if (!mfcc_config.empty()) {
      ParseOptions mpo("mfcc config");
      MfccOptionsParser mp(&mfcc);
      mp.Register(&mpo);
      mpo.ReadConfigFile(mfcc_config);
      mp.Finish();
    } else {                                                // conf/mfcc_hires.conf
      mfcc.c.use_energy = 0; mfcc.c.mel.num_bins = 40; mfcc.c.num_ceps = 40; mfcc.c.mel.low_freq = 20.0f; mfcc.c.mel.high_freq = -400.0f;
    }
    const kamd_frame_opts &frame_opts = use_fbank ? fbank.c.frame : mfcc.c.frame;
    const float samp_freq = frame_opts.samp_freq;

    // TransitionModel + AmNnetSimple, batch-norm / dropout in test mode, collapsed (:160-170)
    TransitionModelAndNnet model(nnet3_rxfilename, acoustic_scale, frame_subsampling_factor);
    AmNnetSimple am_nnet(model);
    DecodingGraph decode_fst(fst_rxfilename);               // ReadFstKaldiGeneric (:176)
    const std::vector<std::pair<std::string, std::vector<std::string> > > spk2utt = ReadSpk2Utt(spk2utt_rspecifier);
    WaveScp wavs(wav_rspecifier);
    CompactLatticeWriter clat_writer(clat_wspecifier);

    const int32 S = std::min<int32>(batch, std::max<int32>(1, static_cast<int32>(spk2utt.size())));
    const int32 sub = frame_subsampling_factor;
    kamd_decoder_config cfg = config.ToC();
    kamd_decoder_sizes sizes;
    Check(kamd_decoder_sizes_suggest(&cfg, S, static_cast<int32>(max_seconds * 1000.0 / frame_opts.frame_shift_ms / sub) + 2, 0, 0, 0, 0, 0.5f, &sizes));
    std::unique_ptr<OnlineStreamBatch> sb_owner(use_fbank ? new OnlineStreamBatch(config, model.Id2Pdf(), am_nnet, decode_fst, fbank, S, max_seconds, sizes)
                                                          : new OnlineStreamBatch(config, model.Id2Pdf(), am_nnet, decode_fst, mfcc, S, max_seconds, sizes));
    OnlineStreamBatch &sb = *sb_owner;
    std::unique_ptr<OnlineIvectorExtractor> extractor;
    if (!ivector_config.empty()) {
      extractor.reset(new OnlineIvectorExtractor(ivector_config));
      sb.SetIvectorExtractor(extractor->handle(), frames_per_chunk, extractor->SpliceRight());
      if (silence_weighting_config.Active())                // :258-259: Active() && IvectorFeature() != NULL
        sb.SetSilenceWeighting(model.Tid2Phone(), silence_weighting_config.SilencePhones(), silence_weighting_config.silence_weight,
                               silence_weighting_config.max_state_duration);
    }
    if ((model.IvectorDim() > 0) != (extractor != NULL))
      throw KaldiFatalError(model.IvectorDim() > 0 ? "the model has an ivector input: give --ivector-extraction-config"
                                                   : "the model has no ivector input: drop --ivector-extraction-config");
    int32 num_done = 0, num_err = 0;
    double tot_like = 0.0;
    int64_t num_frames = 0;
    std::map<std::string, std::unique_ptr<CompactLattice> > results;
    for (size_t b0 = 0; b0 < spk2utt.size(); b0 += S) {
      const size_t ng = std::min<size_t>(S, spk2utt.size() - b0);
      std::vector<std::vector<double> > states(ng);        // per speaker: the adaptation state after its last utterance
      std::vector<size_t> cursor(ng, 0);                    // next utterance of each speaker
      std::map<int32, Active> active;                       // stream -> what it decodes
      for (;;) {
        for (size_t s = 0; s < ng; s++) {                   // start the next utterance on idle streams
          const std::vector<std::string> &utts = spk2utt[b0 + s].second;
          while (!active.count(static_cast<int32>(s)) && cursor[s] < utts.size()) {
            const std::string &utt = utts[cursor[s]++];
            const std::string *rx = wavs.Find(utt);
            if (!rx) { fprintf(stderr, "WARNING Did not find audio for utterance %s\n", utt.c_str()); num_err++; continue; }
            Active a;
            a.key = utt; a.pos = 0;
            WaveScp::Read(*rx, samp_freq, &a.wave);
            const std::vector<int32> one(1, static_cast<int32>(s));
            if (extractor && !states[s].empty()) sb.Start(one, &states[s]);
            else sb.Start(one);
            active[static_cast<int32>(s)] = std::move(a);
          }
        }
        if (active.empty()) break;
        // :225-231 (the reference clamps the chunk to one sample)
        const size_t chunk = chunk_length_secs > 0 ? std::max<size_t>(1, static_cast<size_t>(static_cast<int32>(samp_freq * chunk_length_secs)))
                                                   : static_cast<size_t>(1) << 62;
        std::vector<int32> live, fin;
        std::vector<float> pieces;
        std::vector<int64_t> offsets(1, 0);
        for (std::map<int32, Active>::iterator it = active.begin(); it != active.end(); ++it) {
          Active &a = it->second;
          const size_t n = std::min(chunk, a.wave.size() - a.pos);
          pieces.insert(pieces.end(), a.wave.begin() + a.pos, a.wave.begin() + a.pos + n);
          a.pos += n;
          live.push_back(it->first); fin.push_back(a.pos >= a.wave.size() ? 1 : 0);
          offsets.push_back(static_cast<int64_t>(pieces.size()));
        }
        if (pieces.empty()) pieces.push_back(0.0f);
        sb.AcceptWaveforms(live, pieces.data(), offsets, &fin);      // one upload per tick
        std::vector<int32> decoded;
        sb.AdvanceDecoding(live, &decoded);
        std::vector<char> endpointed(S, 0);
        if (do_endpointing) {                               // one traceback launch for every stream still listening (:251-254)
          std::vector<int32> cand;
          for (size_t i = 0; i < live.size(); i++)
            if (decoded[i] > 0 && active[live[i]].pos < active[live[i]].wave.size()) cand.push_back(live[i]);
          if (!cand.empty()) {
            std::vector<int32> flags;
            sb.EndpointDetected(endpoint_opts, model.Tid2Phone(), cand, &flags);
            for (size_t i = 0; i < cand.size(); i++) if (flags[i]) endpointed[cand[i]] = 1;
          }
        }
        for (size_t i = 0; i < live.size(); i++) {
          const int32 s = live[i];
          Active &a = active[s];
          // an endpoint: the reference breaks out of the chunk loop and goes straight to FinalizeDecoding
          if (a.pos < a.wave.size() && !endpointed[s]) continue;
          const std::vector<int32> one(1, s);
          sb.FinalizeDecoding(one);
          std::vector<int32> ali, words; BaseFloat g = 0, ac = 0;
          Lattice lat;
          if (!sb.GetBestPath(s, &ali, &words, &g, &ac) || !sb.GetRawLattice(s, &lat)) {
            fprintf(stderr, "WARNING Decoding failed for utterance %s\n", a.key.c_str());
            num_err++;
          } else {
            std::unique_ptr<CompactLattice> clat(new CompactLattice());
            DeterminizeLatticePhonePrunedWrapper(model.TidPhone(), lat, config.lattice_beam, clat.get(), det_opts);
            results[a.key] = std::move(clat);
            const double like = -(static_cast<double>(g) + ac);
            const int32 nf = std::max<int32>(static_cast<int32>(ali.size()), 1);
            tot_like += like; num_frames += nf; num_done++;
            fprintf(stderr, "LOG Decoded utterance %s; log-like per frame is %g over %d frames.\n", a.key.c_str(), like / nf, nf);
          }
          if (extractor) sb.GetAdaptationState(s, 1000.0f, &states[s]);
          active.erase(s);
        }
      }
      for (size_t s = 0; s < ng; s++)                       // written in spk2utt order
        for (const std::string &utt : spk2utt[b0 + s].second) {
          std::map<std::string, std::unique_ptr<CompactLattice> >::iterator it = results.find(utt);
          if (it == results.end()) continue;
          clat_writer.Write(utt, *it->second, acoustic_scale != 0.0f ? acoustic_scale : 1.0f);
          results.erase(it);
        }
    }
    clat_writer.Close();
    fprintf(stderr, "LOG Decoded %d utterances, %d with errors.\n", num_done, num_err);
    fprintf(stderr, "LOG Overall likelihood per frame was %g per frame over %lld frames.\n", num_frames ? tot_like / num_frames : 0.0,
            static_cast<long long>(num_frames));
    return num_done != 0 ? 0 : 1;
  } catch (const std::exception &e) {
    fprintf(stderr, "ERROR %s\n", e.what());
    return 255;
  }
}
