// nnet3-latgen-faster (nnet3bin/nnet3-latgen-faster.cc:28-270; the -batch variant nnet3bin/nnet3-latgen-faster-batch.cc:
// 60-240) as a C++ host program over the C-ABI and the kaldi_amd.hpp mirror: the reference's usage line, option names
// and log lines; final.mdl and HCLG.fst are read by the library (kamd_model_read, kamd_graph_read_openfst), the
// acoustic model and the lattice-generating search run on the MI355X.
//
//   nnet3-latgen-faster-amd [options] <nnet-in> <fst-in> <features-rspecifier> <lattice-wspecifier>
//                           [<words-wspecifier> [<alignments-wspecifier>]]
//   e.g.  nnet3-latgen-faster-amd --config=conf/decode.config --acoustic-scale=1.0 --frame-subsampling-factor=3
//             --ivectors=ark:ivectors.ark final.mdl HCLG.fst scp:feats.scp "ark:|gzip -c > lat.1.gz"
//
// How it runs: utterances are collected into sets (--set-frames input frames: what is resident in HBM at a time); a set
// is one NnetBatchDecoder pass (features -> the acoustic model in a few large launches -> ONE work-queue launch of the
// search -> best path and lattice determinization on --num-threads host threads while the search still runs).
// Lattices come out in input order.  The same archive, byte for byte, as tools/nnet3_latgen_faster_batch.py
// (tests/test_gpu_cxx_host.py).  One addition to the reference's options: with --wav the third argument is a wav.scp
// style rspecifier and the MFCCs are computed on the device.
// Build: g++ -std=c++14 -O2 -I include examples/nnet3_latgen_faster.cc -L kaldi_amd/lib -lkaldi_amd -lpthread
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "kaldi_amd.hpp"

using namespace kaldi_amd;

namespace {

struct Utterance {
  std::string key;
  std::vector<float> data;      // feature rows, or samples with --wav
  int32 rows;                   // input frames (10 ms)
  std::vector<float> ivector;
};

struct Totals {
  int32 num_success = 0, num_fail = 0, num_partial = 0;
  double tot_like = 0.0;
  int64_t frame_count = 0;
};

std::vector<std::string> ReadSymbolTable(const std::string &filename) {       // fst::SymbolTable::ReadText: "symbol id" lines
  std::vector<std::string> syms;
  FILE *f = fopen(filename.c_str(), "r");
  if (!f) throw KaldiFatalError("Could not read symbol table from file " + filename);
  char sym[4096]; long id;
  while (fscanf(f, "%4095s %ld", sym, &id) == 2) {
    if (id < 0) continue;
    if (static_cast<size_t>(id) >= syms.size()) syms.resize(id + 1);
    syms[id] = sym;
  }
  fclose(f);
  return syms;
}

}  // namespace

int main(int argc, char **argv) {
  try {
    ParseOptions po("Generate lattices using nnet3 neural net model.\n"
                    "Usage: nnet3-latgen-faster-amd [options] <nnet-in> <fst-in> <features-rspecifier>"
                    " <lattice-wspecifier> [ <words-wspecifier> [<alignments-wspecifier>] ]\n");
    bool allow_partial = false, wav = false;
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
    po.Register("determinize-lattice", &config.determinize_lattice, "If true, determinize the lattice (lattice-determinization, keeping only "
                "best pdf-sequence for each word-sequence).");
    po.Register("beam-delta", &config.beam_delta, "Increment used in decoding-- this parameter is obscure and relates to a speedup in the way "
                "the max-active constraint is applied.  Larger is more accurate.");
    po.Register("hash-ratio", &config.hash_ratio, "(ignored: the device token table is sized by the arenas)");
    po.Register("prune-scale", &config.prune_scale, "(ignored: lattice pruning is exact here)");
    po.Register("delta", &det_opts.c.delta, "Tolerance used in determinization");
    po.Register("max-mem", &max_mem, "Maximum approximate memory usage in determinization (real usage might be many times this).");
    po.Register("phone-determinize", &phone_determinize, "If true, do an initial pass of determinization on both phones and words");
    po.Register("word-determinize", &word_determinize, "If true, do a second pass of determinization on words only");
    po.Register("minimize", &minimize, "(ignored: as in the reference's default, lattices are not minimized)");
    // NnetSimpleComputationOptions::Register (nnet3/nnet-am-decodable-simple.h:68-105)
    int32 extra_left_context = 0, extra_right_context = 0, extra_left_context_initial = -1, extra_right_context_final = -1;
    int32 frame_subsampling_factor = 1, frames_per_chunk = 50;
    BaseFloat acoustic_scale = 0.1f;
    po.Register("extra-left-context", &extra_left_context, "(ignored: TDNN models need no extra context)");
    po.Register("extra-right-context", &extra_right_context, "(ignored)");
    po.Register("extra-left-context-initial", &extra_left_context_initial, "(ignored)");
    po.Register("extra-right-context-final", &extra_right_context_final, "(ignored)");
    po.Register("frame-subsampling-factor", &frame_subsampling_factor, "Required if the frame-rate of the output (e.g. in 'chain' models) is "
                "less than the frame-rate of the original alignment.");
    po.Register("acoustic-scale", &acoustic_scale, "Scaling factor for acoustic log-likelihoods");
    po.Register("frames-per-chunk", &frames_per_chunk, "Number of frames in each chunk that is separately evaluated by the neural net (only matters "
                "with --ivector-extraction-config: without online iVectors whole utterances are batched, which gives the same numbers)");
    po.Register("debug-computation", &debug_computation, "(ignored)");
    std::string word_syms_filename, ivector_rspecifier, online_ivector_rspecifier, utt2spk_rspecifier, mfcc_config, ivector_config, chunk_rule = "simple";
    int32 online_ivector_period = 0, num_threads = 8, set_frames = 2000000, lanes_opt = 0, search_mode = 2, device = -1;
    po.Register("word-symbol-table", &word_syms_filename, "Symbol table for words [for debug output]");
    po.Register("allow-partial", &allow_partial, "If true, produce output even if end state was not reached.");
    po.Register("ivectors", &ivector_rspecifier, "Rspecifier for iVectors as vectors (i.e. not estimated online); per utterance by default, "
                "or per speaker if you provide the --utt2spk option.");
    po.Register("utt2spk", &utt2spk_rspecifier, "Rspecifier for utt2spk option used to get ivectors per speaker");
    po.Register("online-ivectors", &online_ivector_rspecifier, "(not supported here: the device estimates online iVectors itself, see "
                "OnlineStreamBatch / kamd_pipeline_set_ivector_extractor)");
    po.Register("online-ivector-period", &online_ivector_period, "(not supported here)");
    po.Register("ivector-extraction-config", &ivector_config, "Configuration file for online iVector extraction (the one of the online2 binaries / "
                "ivector-extract-online2): the iVectors are estimated on the device from the utterances' own features and the model is evaluated "
                "in chunks of --frames-per-chunk like nnet3-latgen-faster --online-ivectors (steps/nnet3/decode.sh:105-107)");
    po.Register("chunk-rule", &chunk_rule, "With --ivector-extraction-config: simple = DecodableNnetSimple's chunks (nnet3-latgen-faster, this "
                "binary), batch_computer = NnetBatchComputer::SplitUtteranceIntoTasks (nnet3-latgen-faster-batch)");
    po.Register("num-threads", &num_threads, "Number of host threads for the tail of every utterance (best path, lattice determinization): "
                "the decoder threads of nnet3-latgen-faster-batch");
    po.Register("wav", &wav, "The third argument is an scp: rspecifier of waveforms; features are computed on the device");
    po.Register("mfcc-config", &mfcc_config, "Config file with compute-mfcc-feats options (only with --wav; default: mfcc_hires.conf values)");
    po.Register("set-frames", &set_frames, "Input frames (10 ms) resident on the device at a time");
    po.Register("lanes", &lanes_opt, "Decoder lanes kept busy by the work queue (0 = one per compute unit)");
    po.Register("search-mode", &search_mode, "kamd_decoder_set_search_mode (2: the reference's pruning when max-active binds)");
    po.Register("device", &device, "HIP device to run on; job JOB of decode.sh's --nj 8 passes --device=$[JOB-1]");
    po.Read(argc, argv);
    if (po.NumArgs() < 4 || po.NumArgs() > 6) { po.PrintUsage(); return 1; }
    const std::string model_in_filename = po.GetArg(1), fst_in_str = po.GetArg(2), feature_rspecifier = po.GetArg(3),
                      lattice_wspecifier = po.GetArg(4), words_wspecifier = po.NumArgs() > 4 ? po.GetArg(5) : "",
                      alignment_wspecifier = po.NumArgs() > 5 ? po.GetArg(6) : "";
    if (!online_ivector_rspecifier.empty()) throw KaldiFatalError("--online-ivectors is not supported here");
    config.Check();
    if (device >= 0) Check(kamd_set_device(device));
    det_opts.c.max_mem = max_mem; det_opts.c.phone_determinize = phone_determinize; det_opts.c.word_determinize = word_determinize;

    MfccOptions mfcc;                                       // conf/mfcc_hires.conf (egs/mini_librispeech/s5/conf/mfcc_hires.conf)
    mfcc.c.use_energy = 0; mfcc.c.mel.num_bins = 40; mfcc.c.num_ceps = 40; mfcc.c.mel.low_freq = 20.0f; mfcc.c.mel.high_freq = -400.0f;
    if (wav && !mfcc_config.empty()) {
      kamd_mfcc_opts_default(&mfcc.c);
      ParseOptions mpo("compute-mfcc-feats options");
      MfccOptionsParser mp(&mfcc);
      mp.Register(&mpo);
      mpo.ReadConfigFile(mfcc_config);
      mp.Finish();
    }
    const double samp = wav ? mfcc.c.frame.samp_freq : 16000.0, shift = (wav ? mfcc.c.frame.frame_shift_ms : 10.0) * 1e-3;

    // TransitionModel + AmNnetSimple, batch-norm / dropout in test mode, collapsed (nnet3-latgen-faster.cc:91-104)
    TransitionModelAndNnet model(model_in_filename, acoustic_scale, frame_subsampling_factor);
    AmNnetSimple am_nnet(model);
    DecodingGraph decode_fst(fst_in_str);                   // ReadFstKaldiGeneric (:133)
    std::unique_ptr<RandomAccessBaseFloatVectorReader> ivector_reader;
    std::unique_ptr<RandomAccessTokenReader> utt2spk;
    if (!ivector_rspecifier.empty()) ivector_reader.reset(new RandomAccessBaseFloatVectorReader(ivector_rspecifier));
    if (!utt2spk_rspecifier.empty()) utt2spk.reset(new RandomAccessTokenReader(utt2spk_rspecifier));
    std::unique_ptr<OnlineIvectorExtractor> ivector_extractor;
    if (!ivector_config.empty()) {
      if (ivector_reader) throw KaldiFatalError("--ivectors and --ivector-extraction-config are alternatives");
      ivector_extractor.reset(new OnlineIvectorExtractor(ivector_config));
    }
    if ((model.IvectorDim() > 0) != (ivector_reader != NULL || ivector_extractor != NULL))
      throw KaldiFatalError(model.IvectorDim() > 0 ? "the model has an ivector input: give --ivectors or --ivector-extraction-config"
                                                   : "the model has no ivector input: drop --ivectors / --ivector-extraction-config");
    std::vector<std::string> word_syms_storage;
    const std::vector<std::string> *word_syms = NULL;
    if (!word_syms_filename.empty()) { word_syms_storage = ReadSymbolTable(word_syms_filename); word_syms = &word_syms_storage; }

    const bool determinize = config.determinize_lattice;
    std::unique_ptr<CompactLatticeWriter> compact_lattice_writer;
    std::unique_ptr<LatticeWriter> lattice_writer;
    if (determinize) compact_lattice_writer.reset(new CompactLatticeWriter(lattice_wspecifier));
    else lattice_writer.reset(new LatticeWriter(lattice_wspecifier));
    std::unique_ptr<Int32VectorWriter> words_writer, alignment_writer;
    if (!words_wspecifier.empty()) words_writer.reset(new Int32VectorWriter(words_wspecifier));
    if (!alignment_wspecifier.empty()) alignment_writer.reset(new Int32VectorWriter(alignment_wspecifier));

    const int32 device_lanes = kamd_device_num_cus() * kamd_decoder_lanes_per_cu();
    Totals tot;
    double sized_seconds = 0.0;
    int32 sized_lanes = 0;

    // one set = one pass of the device (the arenas are sized for the longest utterance seen so far)
    auto decode_set = [&](std::vector<Utterance> *items) {
      if (items->empty()) return;
      double secs = 0.0;
      for (const Utterance &u : *items) secs = std::max(secs, wav ? u.data.size() / samp : u.rows * shift);
      secs += 0.5;
      const int32 lanes = lanes_opt > 0 ? lanes_opt : std::min<int32>(device_lanes, static_cast<int32>(items->size()));
      sized_seconds = std::max(sized_seconds, secs); sized_lanes = std::max(sized_lanes, lanes);
      const int32 max_out = static_cast<int32>(sized_seconds * (wav ? 1000.0 / mfcc.c.frame.frame_shift_ms : 100.0) / frame_subsampling_factor) + 2;
      kamd_decoder_config cfg = config.ToC();
      kamd_decoder_sizes sizes;
      Check(kamd_decoder_sizes_suggest(&cfg, sized_lanes, max_out, max_out, 0, 0, 0, 0.5f, &sizes));
      NnetBatchDecoderOptions opts;
      opts.acoustic_scale = acoustic_scale; opts.search_mode = search_mode;
      opts.c.resident_lanes = sized_lanes; opts.c.det = det_opts.c;
      NnetBatchDecoder decoder(decode_fst, config, model.Id2Pdf(), model.TidPhone(), word_syms, allow_partial, num_threads, am_nnet,
                               wav ? &mfcc : NULL, sizes, opts);
      if (chunk_rule != "simple" && chunk_rule != "batch_computer") throw KaldiFatalError("--chunk-rule: simple or batch_computer");
      if (ivector_extractor) decoder.SetIvectorExtractor(ivector_extractor->handle(), frames_per_chunk, chunk_rule == "batch_computer");
      for (const Utterance &u : *items) {
        if (wav) decoder.AcceptWaveform(u.key, u.data);
        else decoder.AcceptInput(u.key, u.data.data(), u.rows, model.InputDim(), u.ivector.empty() ? NULL : u.ivector.data(),
                                 static_cast<int32>(u.ivector.size()));
      }
      std::vector<Utterance>().swap(*items);
      decoder.Finished();
      std::string utt, sentence;
      for (;;) {
        if (determinize) {
          CompactLattice clat;
          if (!decoder.GetOutput(&utt, &clat, &sentence)) break;
          compact_lattice_writer->Write(utt, clat);
        } else {
          Lattice lat;
          if (!decoder.GetOutput(&utt, &lat, &sentence)) break;
          lattice_writer->Write(utt, lat);
        }
        if (word_syms) fprintf(stderr, "%s %s\n", utt.c_str(), sentence.c_str());
        if (words_writer || alignment_writer) {
          std::vector<int32> ali, words; BaseFloat g, a;
          if (decoder.GetBestPath(&ali, &words, &g, &a)) {
            if (words_writer) words_writer->Write(utt, words);
            if (alignment_writer) alignment_writer->Write(utt, ali);
          }
        }
      }
      tot.num_success += decoder.NumSuccess(); tot.num_fail += decoder.NumFail(); tot.num_partial += decoder.NumPartial();
      tot.tot_like += decoder.TotLike(); tot.frame_count += decoder.FrameCount();
    };

    std::vector<Utterance> items;
    int64_t frames = 0;
    auto add = [&](Utterance *u) {
      if (ivector_reader) {
        std::string ik = u->key;
        if (utt2spk) {
          if (!utt2spk->HasKey(u->key)) throw KaldiFatalError("utterance " + u->key + " not in the utt2spk map " + utt2spk_rspecifier);
          ik = utt2spk->Value(u->key);
        }
        if (!ivector_reader->HasKey(ik)) {                  // nnet3-latgen-faster.cc:192-196
          fprintf(stderr, "WARNING No iVector available for utterance %s\n", u->key.c_str());
          tot.num_fail++;
          return;
        }
        u->ivector = ivector_reader->Value(ik);
      }
      if (!items.empty() && frames + u->rows > set_frames) { decode_set(&items); frames = 0; }
      frames += u->rows;
      items.push_back(std::move(*u));
    };
    if (wav) {
      WaveScp scp(feature_rspecifier);
      for (const std::pair<std::string, std::string> &e : scp.entries) {
        Utterance u;
        u.key = e.first;
        WaveScp::Read(e.second, static_cast<float>(samp), &u.data);
        u.rows = static_cast<int32>(u.data.size() / samp / shift);
        add(&u);
      }
    } else {
      for (SequentialBaseFloatMatrixReader reader(feature_rspecifier); !reader.Done(); reader.Next()) {
        Utterance u;
        u.key = reader.Key(); u.rows = reader.NumRows();
        if (u.rows == 0) {                                  // :184-188
          fprintf(stderr, "WARNING Zero-length utterance: %s\n", u.key.c_str());
          tot.num_fail++;
          continue;
        }
        if (reader.NumCols() != model.InputDim())
          throw KaldiFatalError("feature dimension " + std::to_string(reader.NumCols()) + " of " + u.key + ", the model expects " + std::to_string(model.InputDim()));
        u.data = reader.Value();
        add(&u);
      }
    }
    decode_set(&items);
    if (compact_lattice_writer) compact_lattice_writer->Close();
    if (lattice_writer) lattice_writer->Close();
    if (words_writer) words_writer->Close();
    if (alignment_writer) alignment_writer->Close();
    // nnet-batch-compute.cc:1336-1343
    fprintf(stderr, "LOG Decoded %d utterances, %d with errors.\n", tot.num_success + tot.num_fail, tot.num_fail);
    fprintf(stderr, "LOG Overall log-likelihood per frame is %g over %lld frames.\n", tot.frame_count ? tot.tot_like / tot.frame_count : 0.0,
            static_cast<long long>(tot.frame_count));
    if (tot.num_partial) fprintf(stderr, "LOG Decoded %d utterances with partial output.\n", tot.num_partial);
    return tot.num_success != 0 ? 0 : 1;
  } catch (const std::exception &e) {
    fprintf(stderr, "ERROR %s\n", e.what());
    return 255;
  }
}
