// latgen-faster-mapped (bin/latgen-faster-mapped.cc:40-180) as a C++ host program over the C-ABI and the
// kaldi_amd.hpp mirror: the same usage line, option names and log messages; the search runs on the MI355X.
//
//   latgen-faster-mapped-amd [options] <id2pdf-rxfilename> <fst-in> <loglikes-rspecifier> <lattice-wspecifier>
//                            [<words-wspecifier> [<alignments-wspecifier>]]
//
// <id2pdf-rxfilename> stands in for the transition model: the TransitionModel's id2pdf table dumped as an
// int32-vector archive entry (kaldi_amd/mdl.py extracts it from final.mdl).  Options incl. --config and ark / scp
// rspecifiers (files, file:offset, pipes) come from kaldi_amd.hpp's ParseOptions / SequentialBaseFloatMatrixReader; build: g++ -std=c++14 -I include examples/latgen_faster_mapped.cc -L kaldi_amd/lib -lkaldi_amd
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "kaldi_amd.hpp"

using namespace kaldi_amd;

int main(int argc, char **argv) {
  try {
    ParseOptions po("Generate lattices, reading log-likelihoods as matrices (model is needed only for the integer mappings in its transition-model)\n"
                    "Usage: latgen-faster-mapped-amd [options] id2pdf-rxfilename fst-in loglikes-rspecifier lattice-wspecifier "
                    "[ words-wspecifier [alignments-wspecifier] ]");
    LatticeFasterDecoderConfig config;
    double acoustic_scale = 0.1;
    bool allow_partial = false, determinize = true;
    po.Register("beam", &config.beam, "Decoding beam.  Larger->slower, more accurate.");
    po.Register("max-active", &config.max_active, "Decoder max active states.  Larger->slower; more accurate");
    po.Register("min-active", &config.min_active, "Decoder minimum #active states.");
    po.Register("lattice-beam", &config.lattice_beam, "Lattice generation beam.  Larger->slower, and deeper lattices");
    po.Register("acoustic-scale", &acoustic_scale, "Scaling factor for acoustic likelihoods");
    po.Register("allow-partial", &allow_partial, "If true, produce output even if end state was not reached.");
    po.Register("determinize-lattice", &determinize, "If true, determinize the lattice (word level).");
    po.Read(argc, argv);
    if (po.NumArgs() < 4 || po.NumArgs() > 6) { po.PrintUsage(); return 1; }
    const std::string id2pdf_rx = po.GetArg(1), fst_rx = po.GetArg(2), ll_rspec = po.GetArg(3), lat_wspec = po.GetArg(4);
    const std::string words_wspec = po.NumArgs() > 4 ? po.GetArg(5) : "", ali_wspec = po.NumArgs() > 5 ? po.GetArg(6) : "";
    // id2pdf
    std::vector<int32> id2pdf;
    {
      int64_t off = 0; char key[256]; int32_t n = 0; int32_t *p = NULL;
      Check(kamd_ark_read_int32_vector(id2pdf_rx.c_str(), &off, key, sizeof(key), &n, &p));
      id2pdf.assign(p, p + n);
      kamd_host_free(p);
    }
    DecodingGraph fst(fst_rx);
    LatticeFasterDecoder decoder(fst, config, id2pdf);
    std::unique_ptr<CompactLatticeWriter> clat_writer;
    std::unique_ptr<LatticeWriter> lat_writer;
    if (determinize) clat_writer.reset(new CompactLatticeWriter(lat_wspec)); else lat_writer.reset(new LatticeWriter(lat_wspec));
    std::unique_ptr<Int32VectorWriter> words_writer, ali_writer;
    if (!words_wspec.empty()) words_writer.reset(new Int32VectorWriter(words_wspec));
    if (!ali_wspec.empty()) ali_writer.reset(new Int32VectorWriter(ali_wspec));
    const std::vector<int32> no_phones;                                 // word-level determinization only
    int num_done = 0, num_err = 0; double tot_like = 0; int64_t frame_count = 0;
    for (SequentialBaseFloatMatrixReader reader(ll_rspec); !reader.Done(); reader.Next()) {
      const std::string key = reader.Key();
      const int32 rows = reader.NumRows(), cols = reader.NumCols();
      if (rows == 0) { fprintf(stderr, "WARNING Zero-length utterance: %s\n", key.c_str()); num_err++; continue; }
      // DecodableMatrixScaledMapped (decoder/decodable-matrix.h:35-70): scale once, then the device gathers by id2pdf
      std::vector<float> data(reader.Value());
      for (size_t k = 0; k < data.size(); k++) data[k] *= static_cast<BaseFloat>(acoustic_scale);
      DecodableMatrixMapped decodable(id2pdf, data.data(), rows, cols);
      double like = 0;
      if (DecodeUtteranceLatticeFaster(decoder, decodable, no_phones, key, acoustic_scale, determinize, allow_partial, ali_writer.get(),
                                       words_writer.get(), clat_writer.get(), lat_writer.get(), &like)) {
        tot_like += like; frame_count += rows; num_done++;
      } else num_err++;
    }
    fprintf(stderr, "LOG Done %d utterances, failed for %d\n", num_done, num_err);
    fprintf(stderr, "LOG Overall log-likelihood per frame is %g over %lld frames.\n", frame_count ? tot_like / frame_count : 0.0,
            static_cast<long long>(frame_count));
    return num_done != 0 ? 0 : 1;
  } catch (const std::exception &e) {
    fprintf(stderr, "ERROR %s\n", e.what());
    return 255;
  }
}
