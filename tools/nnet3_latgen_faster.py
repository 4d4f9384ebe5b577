"""nnet3-latgen-faster (nnet3bin/nnet3-latgen-faster.cc:37-270) on the device:

  nnet3_latgen_faster.py [options] <nnet-in> <fst-in> <features-rspecifier> <lattice-wspecifier>
                         [<words-wspecifier> [<alignments-wspecifier>]]
  e.g.  nnet3_latgen_faster.py --config=conf/decode.config --acoustic-scale=1.0 --frame-subsampling-factor=3 \\
          --online-ivectors=scp:ivector_online.scp --online-ivector-period=10 \\
          final.mdl HCLG.fst scp:feats.scp "ark:|gzip -c > lat.1.gz"

The reference's options, extended filenames and table specifiers are accepted as they are.  One
addition: with --wav the third argument is a waveform rspecifier (scp:wav.scp, entries files or
commands) and the MFCCs are computed on the device with the options of --mfcc-config (the
options of compute-mfcc-feats; default conf/mfcc_hires.conf's values).  Utterances are decoded
--batch at a time; every stage of a batch is one pass on the GPU."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from kaldi_amd import abi, mdl, options, pipeline, table
from kaldi_amd import io as kio
from kaldi_amd._lib import KamdError, lib


def main(argv):
    po = table.ParseOptions(__doc__)
    options.register_decoder(po)
    options.register_nnet_simple(po)
    po.register("word-symbol-table", str, "", "Symbol table for words [for debug output]")
    po.register("allow-partial", bool, False, "If true, produce output even if end state was not reached.")
    po.register("ivectors", str, "", "Rspecifier for iVectors as vectors (i.e. not estimated online); per utterance "
                "by default, or per speaker if you provide the --utt2spk option.")
    po.register("utt2spk", str, "", "Rspecifier for utt2spk option used to get ivectors per speaker")
    po.register("online-ivectors", str, "", "Rspecifier for iVectors estimated online, as matrices.")
    po.register("online-ivector-period", int, 0, "Number of frames between iVectors in matrices supplied to the "
                "--online-ivectors option")
    po.register("ivector-extraction-config", str, "", "Not in the reference binary: estimate the online i-vectors on the device "
                "from the utterance's own features (the config file ivector-extract-online2 and the online2 binaries take: "
                "--lda-matrix, --global-cmvn-stats, --diag-ubm, --ivector-extractor, ...), fresh state per utterance")
    po.register("wav", bool, False, "The third argument is a waveform rspecifier; features are computed on the device")
    po.register("mfcc-config", str, "", "Config file with compute-mfcc-feats options (only with --wav)")
    po.register("batch", int, 64, "Utterances decoded per pass")
    po.register("num-threads", int, max(1, min(4, (os.cpu_count() or 2) - 1)), "Host threads for lattice determinization (the host tail of a batch; "
                "the reference's nnet3-latgen-faster-parallel uses a TaskSequencer for the same purpose)")
    po.register("device", int, -1, "HIP device to run on (default: device 0 of HIP_VISIBLE_DEVICES); with decode.sh-style "
                "splitting, job JOB of --nj 8 passes --device=$[JOB-1]: utterances shard across the GPUs of a node with "
                "no communication")
    args = po.read(argv)
    if not 4 <= len(args) <= 6:
        po.print_usage()
        return 1
    model_in, fst_in, feat_rspec, lat_wspec = args[:4]
    words_wspec = args[4] if len(args) > 4 else ""
    ali_wspec = args[5] if len(args) > 5 else ""
    if po["ivectors"] and (po["online-ivectors"] or po["ivector-extraction-config"]):
        raise KamdError("--ivectors excludes --online-ivectors and --ivector-extraction-config")      # nnet3-latgen-faster.cc:88-92
    if po["online-ivectors"] and po["ivector-extraction-config"]:
        raise KamdError("--online-ivectors and --ivector-extraction-config exclude each other")
    if po["online-ivectors"] and po["online-ivector-period"] <= 0:
        raise KamdError("--online-ivector-period must be set with --online-ivectors")       # nnet3-latgen-faster.cc:94-99
    if po["device"] >= 0:
        from kaldi_amd._lib import check
        check(lib().kamd_set_device(po["device"]))
    cfg = options.decoder_config(po)
    mpo = table.ParseOptions("compute-mfcc-feats options")
    options.register_mfcc(mpo)
    if po["mfcc-config"]:
        mpo.read_config_file(po["mfcc-config"])
        mfcc = options.mfcc_opts(mpo)
    else:
        mfcc = abi.mfcc_opts_hires()
    acwt = po["acoustic-scale"]
    model, id2pdf, tid_phone = mdl.read_mdl(model_in, acwt, po["frame-subsampling-factor"])
    with table.Input(fst_in) as (path, off):
        if off:
            raise KamdError("the decoding graph cannot be read from inside an archive: " + fst_in)
        g = kio.read_openfst(path)
    g.tid2pdf, g.num_pdfs = id2pdf, model.num_pdfs
    ivecs = table.RandomAccessTableReader(po["online-ivectors"], "matrix") if po["online-ivectors"] else None
    # RandomAccessBaseFloatVectorReaderMapped(ivector_rspecifier, utt2spk_rspecifier) (nnet3-latgen-faster.cc:135-137)
    const_ivecs = table.RandomAccessTableReader(po["ivectors"], "vector") if po["ivectors"] else None
    utt2spk = {k: v[0] for k, v in table.SequentialTableReader(po["utt2spk"], "tokens")} if po["utt2spk"] else None
    extractor = None
    if po["ivector-extraction-config"]:
        from kaldi_amd import ivector
        extractor = ivector.IvectorExtractor(ivector.IvectorExtractionInfo.from_config(po["ivector-extraction-config"]))
    lat_kind = "compact_lattice" if po["determinize-lattice"] else "lattice"
    lat_w = table.TableWriter(lat_wspec, lat_kind, acoustic_scale=acwt)
    words_w = table.TableWriter(words_wspec, "int32") if words_wspec else None
    ali_w = table.TableWriter(ali_wspec, "int32") if ali_wspec else None
    det = kio.determinize_opts_default()
    det.delta, det.phone_determinize, det.word_determinize = po["delta"], int(po["phone-determinize"]), int(po["word-determinize"])
    n_done = n_fail = 0
    tot_like, tot_frames = 0.0, 0
    shift = mfcc.frame.frame_shift_ms * 1e-3
    state = {"pipe": None, "max_s": 0.0, "n": 0}

    def decode_batch(batch):
        nonlocal n_done, n_fail, tot_like, tot_frames
        keys = [k for k, _ in batch]
        def ivkey(k):
            if utt2spk is None:
                return k
            if k not in utt2spk:
                raise KamdError("utterance %s not in the utt2spk map %s" % (k, po["utt2spk"]))
            return utt2spk[k]
        if ivecs is not None or const_ivecs is not None:
            missing = [k for k in keys if (k not in ivecs if ivecs is not None else ivkey(k) not in const_ivecs)]
            for k in missing:
                print("WARNING No iVectors available for utterance " + k, file=sys.stderr)     # nnet3-latgen-faster.cc:176-181
            n_fail += len(missing)
            batch = [(k, v) for k, v in batch if k not in missing]
            keys = [k for k, _ in batch]
        if not batch:
            return
        vals = [v for _, v in batch]
        secs = max((v.size / mfcc.frame.samp_freq) if po["wav"] else v.shape[0] * shift for v in vals) + 0.5
        if state["pipe"] is None or secs > state["max_s"] or len(vals) > state["n"]:
            state["pipe"] = pipeline.Pipeline(mfcc, model, g, cfg, max_utts=max(len(vals), state["n"]),
                                              max_seconds=max(secs, state["max_s"]))
            state["max_s"], state["n"] = max(secs, state["max_s"]), max(len(vals), state["n"])
        pipe = state["pipe"]
        if po["wav"]:
            pipe.load(vals)
        else:
            pipe.load_features(vals)
        if ivecs is not None:
            pipe.set_online_ivectors([ivecs[k] for k in keys], po["online-ivector-period"], po["frames-per-chunk"])
        if const_ivecs is not None:
            pipe.set_ivectors([const_ivecs[ivkey(k)] for k in keys])
        if extractor is not None:
            pipe.set_ivector_extractor(extractor, po["frames-per-chunk"])
        pipe.run(auto_grow=4)
        results = pipe.results(lattices=True)
        # determinization (host C code, releases the GIL) of the whole batch in parallel; output stays in input order
        clats = {}
        if po["determinize-lattice"]:
            todo = [(k, r["lattice"]) for k, r in zip(keys, results) if r is not None and r["lattice"] is not None]
            with ThreadPoolExecutor(max_workers=max(1, po["num-threads"])) as pool:
                for (k, _), c in zip(todo, pool.map(lambda kv: kio.determinize_lattice(kv[1], cfg.lattice_beam, tid_phone, det), todo)):
                    clats[k] = c
        for key, res, lane in zip(keys, results, pipe._lane_of):
            if res is None:
                print("WARNING Zero-length utterance: " + key, file=sys.stderr)
                n_fail += 1
                continue
            bp, lat = res["best"], res["lattice"]
            reached = bool(lib().kamd_decoder_reached_final(pipe.dec._dec, lane))
            if bp is None or lat is None or (not reached and not po["allow-partial"]):
                print("WARNING Not producing output for utterance %s since no final-state reached%s" %
                      (key, "" if po["allow-partial"] else " and --allow-partial=false."), file=sys.stderr)
                n_fail += 1
                continue
            if not reached:
                print("WARNING Outputting partial output for utterance %s since no final-state reached" % key, file=sys.stderr)
            if po["determinize-lattice"]:
                lat_w.write(key, clats[key])
            else:
                lat_w.write(key, lat)
            if words_w:
                words_w.write(key, bp["words"])
            if ali_w:
                ali_w.write(key, bp["alignment"])
            like = -(bp["graph_cost"] + bp["acoustic_cost"])
            nf = max(len(bp["alignment"]), 1)
            tot_like += like; tot_frames += nf; n_done += 1
            print("LOG Log-like per frame for utterance %s is %g over %d frames." % (key, like / nf, nf), file=sys.stderr)

    batch = []
    for key, val in table.SequentialTableReader(feat_rspec, "wave" if po["wav"] else "matrix"):
        if po["wav"]:
            sf, data = val
            if sf != mfcc.frame.samp_freq:
                raise KamdError("%s: sampling rate %g, the feature config expects %g" % (key, sf, mfcc.frame.samp_freq))
            val = data[0]
        elif val.shape[0] == 0:
            print("WARNING Zero-length utterance: " + key, file=sys.stderr)
            n_fail += 1
            continue
        batch.append((key, val))
        if len(batch) == po["batch"]:
            decode_batch(batch)
            batch = []
    decode_batch(batch)
    for w in (lat_w, words_w, ali_w):
        if w:
            w.close()
    print("LOG Done %d utterances, failed for %d" % (n_done, n_fail), file=sys.stderr)
    print("LOG Overall log-likelihood per frame is %g over %d frames." % (tot_like / max(tot_frames, 1), tot_frames), file=sys.stderr)
    return 0 if n_done else 1


if __name__ == "__main__":
    try:
        sys.exit(main(sys.argv))
    except KamdError as e:
        print("ERROR " + str(e), file=sys.stderr)
        sys.exit(255)
