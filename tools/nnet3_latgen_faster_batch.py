"""nnet3-latgen-faster-batch (nnet3bin/nnet3-latgen-faster-batch.cc:60-240) on the device:

  nnet3_latgen_faster_batch.py [options] <nnet-in> <fst-in> <features-rspecifier> <lattice-wspecifier>
  e.g.  nnet3_latgen_faster_batch.py --config=conf/decode.config --acoustic-scale=1.0 --frame-subsampling-factor=3 \\
          --num-threads=16 final.mdl HCLG.fst scp:feats.scp "ark:|gzip -c > lat.1.gz"

The reference's NnetBatchDecoder keeps decoder threads fed by one GPU inference thread; here the utterances are
collected into sets (--set-frames input frames at most: what stays resident in HBM at a time), and a set is one pass
of kamd_batch_decoder_*: features -> the acoustic model in a few large launches -> ONE work-queue launch of the search
that keeps a lane per compute unit busy -> best path and lattice determinization on --num-threads host threads while
the search is still running.  Lattices come out in input order.  The reference's options are accepted as they are
(the minibatch options have nothing to configure here).  One addition: with --wav the third argument is a waveform
rspecifier and the MFCCs are computed on the device (--mfcc-config).  Not supported: --online-ivectors (matrices
estimated elsewhere; nnet3_latgen_faster.py takes them)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kaldi_amd import abi, batch, decoder, mdl, options, table
from kaldi_amd import io as kio
from kaldi_amd._lib import KamdError, check, lib


def main(argv):
    po = table.ParseOptions(__doc__)
    options.register_decoder(po)
    options.register_nnet_simple(po)
    for name, typ, default in (("minibatch-size", int, 128), ("edge-minibatch-size", int, 32), ("partial-minibatch-factor", float, 0.5),
                               ("ensure-exact-final-context", bool, False)):
        po.register(name, typ, default, "(NnetBatchComputerOptions; ignored: whole utterances are batched on the device)")
    po.register("word-symbol-table", str, "", "Symbol table for words [for debug output]")
    po.register("allow-partial", bool, False, "If true, produce output even if end state was not reached.")
    po.register("ivectors", str, "", "Rspecifier for iVectors as vectors (i.e. not estimated online); per utterance "
                "by default, or per speaker if you provide the --utt2spk option.")
    po.register("utt2spk", str, "", "Rspecifier for utt2spk option used to get ivectors per speaker")
    po.register("online-ivectors", str, "", "(not supported by this tool)")
    po.register("online-ivector-period", int, 0, "(not supported by this tool)")
    po.register("ivector-extraction-config", str, "", "Configuration file for online iVector extraction (the one of the online2 binaries): the "
                "iVectors are estimated on the device from the utterances' own features, the model is evaluated in tasks of "
                "--frames-per-chunk like NnetBatchComputer with --online-ivectors (--chunk-rule=simple: like nnet3-latgen-faster)")
    po.register("chunk-rule", str, "batch_computer", "With --ivector-extraction-config: batch_computer = NnetBatchComputer::SplitUtteranceIntoTasks "
                "(this binary in the reference), simple = DecodableNnetSimple's chunks (nnet3-latgen-faster)")
    po.register("num-threads", int, max(1, min(16, (os.cpu_count() or 2) - 1)), "Number of host threads for the tail of every "
                "utterance (best path, lattice determinization); the reference's decoder threads")
    po.register("use-gpu", str, "yes", "(ignored: there is no CPU path)")
    po.register("wav", bool, False, "The third argument is a waveform rspecifier; features are computed on the device")
    po.register("mfcc-config", str, "", "Config file with compute-mfcc-feats options (only with --wav)")
    po.register("set-frames", int, 2000000, "Input frames (10 ms) resident on the device at a time")
    po.register("lanes", int, 0, "Decoder lanes kept busy by the work queue (0 = one per compute unit)")
    po.register("search-mode", int, 2, "kamd_decoder_set_search_mode (2: the reference's pruning when max-active binds)")
    po.register("device", int, -1, "HIP device to run on; job JOB of decode.sh's --nj 8 passes --device=$[JOB-1]")
    args = po.read(argv)
    if len(args) != 4:
        po.print_usage()
        return 1
    model_in, fst_in, feat_rspec, lat_wspec = args
    if po["online-ivectors"]:
        raise KamdError("--online-ivectors is not supported here: use nnet3_latgen_faster.py")
    if po["device"] >= 0:
        check(lib().kamd_set_device(po["device"]))
    cfg = options.decoder_config(po)
    if po["wav"]:
        mpo = table.ParseOptions("compute-mfcc-feats options")
        options.register_mfcc(mpo)
        if po["mfcc-config"]:
            mpo.read_config_file(po["mfcc-config"])
            mfcc = options.mfcc_opts(mpo)
        else:
            mfcc = abi.mfcc_opts_hires()
    else:
        mfcc = None
    acwt = po["acoustic-scale"]
    model, id2pdf, tid_phone = mdl.read_mdl(model_in, acwt, po["frame-subsampling-factor"])
    with table.Input(fst_in) as (path, off):
        if off:
            raise KamdError("the decoding graph cannot be read from inside an archive: " + fst_in)
        g = kio.read_openfst(path)
    g.tid2pdf, g.num_pdfs = id2pdf, model.num_pdfs
    G = decoder.Graph(g)
    const_ivecs = table.RandomAccessTableReader(po["ivectors"], "vector") if po["ivectors"] else None
    utt2spk = {k: v[0] for k, v in table.SequentialTableReader(po["utt2spk"], "tokens")} if po["utt2spk"] else None
    extractor = None
    if po["ivector-extraction-config"]:
        from kaldi_amd import ivector
        if const_ivecs is not None:
            raise KamdError("--ivectors and --ivector-extraction-config are alternatives")
        extractor = ivector.IvectorExtractor(ivector.IvectorExtractionInfo.from_config(po["ivector-extraction-config"]))
    if (model.ivector_dim > 0) != (const_ivecs is not None or extractor is not None):
        raise KamdError("the model %s an ivector input: %s --ivectors / --ivector-extraction-config" %
                        (("has", "give") if model.ivector_dim else ("has no", "drop")))
    words = None
    if po["word-symbol-table"]:
        words = {}
        for line in open(po["word-symbol-table"], encoding="utf-8"):
            f = line.split()
            if len(f) == 2:
                words[int(f[1])] = f[0]
    determinize = bool(po["determinize-lattice"])
    lat_w = table.TableWriter(lat_wspec, "compact_lattice" if determinize else "lattice", acoustic_scale=acwt)
    samp = mfcc.frame.samp_freq if mfcc is not None else 16000.0
    shift = (mfcc.frame.frame_shift_ms if mfcc is not None else 10.0) * 1e-3
    cus = lib().kamd_device_num_cus() * lib().kamd_decoder_lanes_per_cu()
    state = {"bd": None, "max_s": 0.0, "lanes": 0}
    n_done = n_fail = n_partial = 0
    tot_like, tot_frames = 0.0, 0

    def decode_set(items):
        nonlocal n_done, n_fail, n_partial, tot_like, tot_frames
        if not items:
            return
        keys = [k for k, _, _ in items]
        vals = [v for _, v, _ in items]
        secs = max((v.size / samp) if po["wav"] else v.shape[0] * shift for v in vals) + 0.5
        lanes = po["lanes"] or min(cus, len(vals))
        if state["bd"] is None or secs > state["max_s"] or lanes > state["lanes"]:
            state["bd"] = None                      # release the old arenas first
            state["max_s"], state["lanes"] = max(secs, state["max_s"]), max(lanes, state["lanes"])
            state["bd"] = batch.NnetBatchDecoder(mfcc, model, G, cfg, max_seconds=state["max_s"], resident_lanes=state["lanes"],
                                                 host_threads=po["num-threads"], determinize=determinize, keep_raw_lattices=not determinize,
                                                 tid_phone=tid_phone, search_mode=po["search-mode"],
                                                 det=dict(delta=po["delta"], phone_determinize=int(po["phone-determinize"]),
                                                          word_determinize=int(po["word-determinize"])))
            if extractor is not None:
                state["bd"].set_chunk_rule(po["chunk-rule"])
                state["bd"].set_ivector_extractor(extractor, po["frames-per-chunk"])
        bd = state["bd"]
        if po["wav"]:
            bd.load(vals)
        else:
            bd.load_features(vals, None if const_ivecs is None else np.stack([iv for _, _, iv in items]))
        bd.run()
        for u, key in enumerate(keys):
            out = bd.output(u)
            rec = bd.record(u)
            reached = np.isfinite(rec.final_relative_cost)
            if out is None:
                print("WARNING Decoding failed for utterance %s (flags %d)" % (key, rec.error), file=sys.stderr)
                n_fail += 1
                continue
            if not reached and not po["allow-partial"]:
                print("WARNING Not producing output for utterance %s since no final-state reached and --allow-partial=false." % key, file=sys.stderr)
                n_fail += 1
                continue
            if not reached:
                print("WARNING Outputting partial output for utterance %s since no final-state reached" % key, file=sys.stderr)
                n_partial += 1
            lat = bd.compact_lattice(u) if determinize else bd.raw_lattice(u)
            if lat is None:
                print("WARNING Empty lattice for utterance " + key, file=sys.stderr)
                n_fail += 1
                continue
            lat_w.write(key, lat)
            if words is not None:
                print(key + " " + " ".join(words.get(int(w), "<%d>" % int(w)) for w in out["words"]), file=sys.stderr)
            like = -(out["graph_cost"] + out["acoustic_cost"])
            nf = max(len(out["alignment"]), 1)
            tot_like += like; tot_frames += nf; n_done += 1
            print("LOG Log-like per frame for utterance %s is %g over %d frames." % (key, like / nf, nf), file=sys.stderr)

    items, frames = [], 0
    for key, val in table.SequentialTableReader(feat_rspec, "wave" if po["wav"] else "matrix"):
        if po["wav"]:
            sf, data = val
            if sf != samp:
                raise KamdError("%s: sampling rate %g, the feature config expects %g" % (key, sf, samp))
            val = data[0]
            nfr = int(val.size / samp / shift)
        else:
            nfr = val.shape[0]
            if nfr == 0:
                print("WARNING Zero-length utterance: " + key, file=sys.stderr)       # :184-188, decoder.UtteranceFailed()
                n_fail += 1
                continue
        iv = None
        if const_ivecs is not None:
            ik = key
            if utt2spk is not None:
                if key not in utt2spk:
                    raise KamdError("utterance %s not in the utt2spk map %s" % (key, po["utt2spk"]))
                ik = utt2spk[key]
            if ik not in const_ivecs:
                print("WARNING No iVector available for utterance " + key, file=sys.stderr)   # :192-196
                n_fail += 1
                continue
            iv = np.asarray(const_ivecs[ik], np.float32)
        if items and frames + nfr > po["set-frames"]:
            decode_set(items)
            items, frames = [], 0
        items.append((key, val, iv))
        frames += nfr
    decode_set(items)
    lat_w.close()
    print("LOG Decoded %d utterances, %d with errors." % (n_done + n_fail, n_fail), file=sys.stderr)      # nnet-batch-compute.cc:1336-1337
    print("LOG Overall log-likelihood per frame is %g over %d frames." % (tot_like / max(tot_frames, 1), tot_frames), file=sys.stderr)
    if n_partial:
        print("LOG Decoded %d utterances with partial output." % n_partial, file=sys.stderr)
    return 0 if n_done else 1


if __name__ == "__main__":
    try:
        sys.exit(main(sys.argv))
    except KamdError as e:
        print("ERROR " + str(e), file=sys.stderr)
        sys.exit(255)
