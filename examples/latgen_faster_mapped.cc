// latgen-faster-mapped (bin/latgen-faster-mapped.cc:40-180) as a C++ host program over the C-ABI and the
// kaldi_amd.hpp mirror: the same usage line, option names and log messages; the search runs on the MI355X.
//
//   latgen-faster-mapped-amd [options] <id2pdf-rxfilename> <fst-in> <loglikes-rspecifier> <lattice-wspecifier>
//                            [<words-wspecifier> [<alignments-wspecifier>]]
//
// <id2pdf-rxfilename> stands in for the transition model: the TransitionModel's id2pdf table dumped as an
// int32-vector archive entry (kaldi_amd/mdl.py extracts it from final.mdl).  Archives only ("ark:file",
// "ark,t:file"); build: g++ -std=c++14 -I include examples/latgen_faster_mapped.cc -L kaldi_amd/lib -lkaldi_amd
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
    LatticeFasterDecoderConfig config;
    double acoustic_scale = 0.1;
    bool allow_partial = false, determinize = true;
    int i = 1;
    for (; i < argc && !strncmp(argv[i], "--", 2); i++) {            // ParseOptions: --name=value
      std::string a(argv[i] + 2);
      const size_t eq = a.find('=');
      std::string k = a.substr(0, eq), v = eq == std::string::npos ? "true" : a.substr(eq + 1);
      for (char &c : k) if (c == '_') c = '-';
      const bool on = v == "true" || v == "t" || v == "1";
      if (k == "beam") config.beam = atof(v.c_str());
      else if (k == "max-active") config.max_active = atoi(v.c_str());
      else if (k == "min-active") config.min_active = atoi(v.c_str());
      else if (k == "lattice-beam") config.lattice_beam = atof(v.c_str());
      else if (k == "acoustic-scale") acoustic_scale = atof(v.c_str());
      else if (k == "allow-partial") allow_partial = on;
      else if (k == "determinize-lattice") determinize = on;
      else { fprintf(stderr, "ERROR Invalid option %s\n", argv[i]); return 255; }
    }
    if (argc - i < 4 || argc - i > 6) {
      fprintf(stderr, "Usage: latgen-faster-mapped-amd [options] id2pdf-rxfilename fst-in loglikes-rspecifier lattice-wspecifier "
                      "[ words-wspecifier [alignments-wspecifier] ]\n");
      return 1;
    }
    const std::string id2pdf_rx = argv[i], fst_rx = argv[i + 1], ll_rspec = argv[i + 2], lat_wspec = argv[i + 3];
    const std::string words_wspec = argc - i > 4 ? argv[i + 4] : "", ali_wspec = argc - i > 5 ? argv[i + 5] : "";
    // id2pdf
    std::vector<int32> id2pdf;
    {
      int64_t off = 0; char key[256]; int32_t n = 0; int32_t *p = NULL;
      Check(kamd_ark_read_int32_vector(id2pdf_rx.c_str(), &off, key, sizeof(key), &n, &p));
      id2pdf.assign(p, p + n);
      kamd_host_free(p);
    }
    char rx[4096]; int ropts = 0;
    if (kamd_classify_rspecifier(ll_rspec.c_str(), rx, sizeof(rx), &ropts) != 1)
      throw KaldiFatalError("only archive rspecifiers (ark:file) are supported here: " + ll_rspec);
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
    int64_t off = 0;
    for (;;) {
      char key[1024]; int32_t rows = 0, cols = 0; float *data = NULL;
      const int rc = kamd_ark_read_matrix(rx, &off, key, sizeof(key), &rows, &cols, &data);
      if (rc == 1) break;
      Check(rc);
      if (rows == 0) { fprintf(stderr, "WARNING Zero-length utterance: %s\n", key); num_err++; kamd_host_free(data); continue; }
      // DecodableMatrixScaledMapped (decoder/decodable-matrix.h:35-70): scale once, then the device gathers by id2pdf
      for (size_t k = 0; k < static_cast<size_t>(rows) * cols; k++) data[k] *= static_cast<BaseFloat>(acoustic_scale);
      DecodableMatrixMapped decodable(id2pdf, data, rows, cols);
      double like = 0;
      if (DecodeUtteranceLatticeFaster(decoder, decodable, no_phones, key, acoustic_scale, determinize, allow_partial, ali_writer.get(),
                                       words_writer.get(), clat_writer.get(), lat_writer.get(), &like)) {
        tot_like += like; frame_count += rows; num_done++;
      } else num_err++;
      kamd_host_free(data);
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
