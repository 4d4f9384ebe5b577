"""online2-wav-nnet3-latgen-faster (online2bin/online2-wav-nnet3-latgen-faster.cc:60-300) on the device:

  online2_wav_nnet3_latgen_faster.py [options] <nnet3-in> <fst-in> <spk2utt-rspecifier> <wav-rspecifier> <lattice-wspecifier>
  e.g.  online2_wav_nnet3_latgen_faster.py --config=conf/online.conf --do-endpointing=false --frames-per-chunk=20 \\
            --acoustic-scale=1.0 --frame-subsampling-factor=3 final.mdl HCLG.fst ark:spk2utt scp:wav.scp "ark:|gzip -c > lat.1.gz"

online.conf's --feature-type, --mfcc-config / --fbank-config, --ivector-extraction-config and --endpoint.* options are read as
they are.  Audio is fed --chunk-length seconds at a time; every tick advances ALL active streams together (one
stream per speaker, --batch speakers at once; a speaker's utterances follow each other and hand their i-vector
adaptation state on) and the --ivector-silence-weighting.* options (OnlineSilenceWeighting).  --feature-type=mfcc or fbank; not supported: plp and pitch features."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kaldi_amd import abi, decoder, mdl, online, options, table
from kaldi_amd import io as kio
from kaldi_amd._lib import KamdError, lib


def main(argv):
    po = table.ParseOptions(__doc__)
    options.register_decoder(po)
    options.register_nnet_simple(po)
    po.register("frames-per-chunk", int, 20, "Number of frames in each chunk that is separately evaluated by the neural net "
                "(NnetSimpleLoopedComputationOptions, nnet3/decodable-simple-looped.h:52-90); with i-vectors also their period")
    po.register("chunk-length", float, 0.18, "Length of chunk size in seconds, that we process.  Set to <= 0 to use all input in one chunk.")
    po.register("word-symbol-table", str, "", "Symbol table for words [for debug output]")
    po.register("do-endpointing", bool, False, "If true, apply endpoint detection")
    po.register("online", bool, True, "(ignored: decoding is always chunk by chunk)")
    po.register("num-threads-startup", int, 8, "(ignored)")
    po.register("feature-type", str, "mfcc", "Base feature type [mfcc, fbank]")
    po.register("mfcc-config", str, "", "Configuration file for MFCC features (e.g. conf/mfcc_hires.conf)")
    po.register("fbank-config", str, "", "Configuration file for filterbank features (e.g. conf/fbank.conf)")
    po.register("ivector-extraction-config", str, "", "Configuration file for online iVector extraction")
    online.OnlineSilenceWeightingConfig.register(po)
    po.register("endpoint.silence-phones", str, "", "List of phones that are considered to be silence phones by the endpointing code.")
    ep = online.OnlineEndpointConfig()
    for i, r in enumerate((ep.rule1, ep.rule2, ep.rule3, ep.rule4, ep.rule5), 1):
        pre = "endpoint.rule%d." % i
        po.register(pre + "must-contain-nonsilence", bool, r.must_contain_nonsilence); po.register(pre + "min-trailing-silence", float, r.min_trailing_silence)
        po.register(pre + "max-relative-cost", float, r.max_relative_cost); po.register(pre + "min-utterance-length", float, r.min_utterance_length)
    po.register("batch", int, 64, "Speakers decoded concurrently")
    po.register("max-seconds", float, 60.0, "Longest utterance the stream slots are sized for")
    args = po.read(argv)
    if len(args) != 5:
        po.print_usage()
        return 1
    if po["feature-type"] not in ("mfcc", "fbank"):        # online-nnet2-feature-pipeline.cc:36-58 (plp and pitch: not on this path)
        raise KamdError("Invalid feature type: %s (supported: mfcc, fbank)" % po["feature-type"])
    for i, r in enumerate((ep.rule1, ep.rule2, ep.rule3, ep.rule4, ep.rule5), 1):
        pre = "endpoint.rule%d." % i
        r.must_contain_nonsilence, r.min_trailing_silence = po[pre + "must-contain-nonsilence"], po[pre + "min-trailing-silence"]
        r.max_relative_cost, r.min_utterance_length = po[pre + "max-relative-cost"], po[pre + "min-utterance-length"]
    sil = [int(x) for x in po["endpoint.silence-phones"].split(":") if x]
    if po["do-endpointing"] and not sil:
        raise KamdError("--do-endpointing needs --endpoint.silence-phones")
    if po["feature-type"] == "fbank":
        fpo = table.ParseOptions("fbank config")
        options.register_fbank(fpo)
        if po["fbank-config"]:
            fpo.read_config_file(po["fbank-config"])
        mfcc = options.fbank_opts(fpo) if po["fbank-config"] else abi.fbank_opts_default()       # (the base feature's options, whichever type)
    else:
        mpo = table.ParseOptions("mfcc config")
        options.register_mfcc(mpo)
        if po["mfcc-config"]:
            mpo.read_config_file(po["mfcc-config"])
            mfcc = options.mfcc_opts(mpo)
        else:
            mfcc = abi.mfcc_opts_hires()
    cfg = options.decoder_config(po)
    acwt = po["acoustic-scale"]
    model, id2pdf, tid_phone = mdl.read_mdl(args[0], acwt, po["frame-subsampling-factor"])
    with table.Input(args[1]) as (path, off):
        g = kio.read_openfst(path)
    g.tid2pdf, g.num_pdfs = id2pdf, model.num_pdfs
    kind, rx, _ = table.classify_rspecifier(args[2])
    if kind != table.ARCHIVE:
        raise KamdError("the spk2utt rspecifier must be a text archive")
    spk2utt = [(spk, rest.split()) for spk, rest in table.read_script_file(rx)]
    wavs = table.RandomAccessTableReader(args[3], "wave")
    writer = table.TableWriter(args[4], "compact_lattice", acoustic_scale=acwt)
    S = min(po["batch"], max(1, len(spk2utt)))
    N, G = decoder.Nnet(model), decoder.Graph(g)
    sub = model.subsampling
    from kaldi_amd.pipeline import default_sizes
    sb = online.StreamBatch(mfcc, N, G, cfg, S, max_seconds=po["max-seconds"],
                            sizes=default_sizes(cfg, S, int(po["max-seconds"] * 1000.0 / mfcc.frame.frame_shift_ms / sub) + 2))
    extractor = None
    if po["ivector-extraction-config"]:
        from kaldi_amd import ivector
        info = ivector.IvectorExtractionInfo.from_config(po["ivector-extraction-config"])
        extractor = ivector.IvectorExtractor(info)
        sb.set_ivector_extractor(extractor, po["frames-per-chunk"])
        swc = online.OnlineSilenceWeightingConfig.from_options(po)
        if swc.Active():          # online2-wav-nnet3-latgen-faster.cc:258-259: Active() && IvectorFeature() != NULL
            sb.set_silence_weighting(swc, model.tid2phone)
    det = kio.determinize_opts_default()
    det.delta, det.phone_determinize, det.word_determinize = po["delta"], int(po["phone-determinize"]), int(po["word-determinize"])
    frame_shift = mfcc.frame.frame_shift_ms * 1e-3 * sub
    n_done = n_err = 0
    tot_like, tot_frames = 0.0, 0
    results = {}
    for b0 in range(0, len(spk2utt), S):
        group = spk2utt[b0:b0 + S]
        states = [None] * len(group)                       # per speaker: adaptation state
        cursor = [0] * len(group)                          # next utterance of each speaker
        active = {}                                        # stream -> dict(key, wave, pos)
        while True:
            for s, (spk, utts) in enumerate(group):        # start the next utterance on idle streams
                while s not in active and cursor[s] < len(utts):
                    utt = utts[cursor[s]]; cursor[s] += 1
                    if utt not in wavs:
                        print("WARNING Did not find audio for utterance " + utt, file=sys.stderr); n_err += 1
                        continue
                    sf, data = wavs[utt]
                    if sf != mfcc.frame.samp_freq:
                        raise KamdError("%s: sampling rate %g, the feature config expects %g" % (utt, sf, mfcc.frame.samp_freq))
                    if extractor is not None and states[s] is not None:
                        sb.start([s], states=[states[s]])
                    else:
                        sb.start([s])
                    active[s] = dict(key=utt, wave=data[0], pos=0)
            if not active:
                break
            chunk = max(1, int(po["chunk-length"] * mfcc.frame.samp_freq)) if po["chunk-length"] > 0 else 1 << 62   # the reference clamps the chunk to one sample
            live = sorted(active)
            pieces = []
            for s in live:
                a = active[s]
                pieces.append(a["wave"][a["pos"]:a["pos"] + chunk])
                a["pos"] += pieces[-1].size
            sb.accept_many(live, pieces, [active[s]["pos"] >= active[s]["wave"].size for s in live])     # one upload per tick
            decoded = sb.advance(live)
            endpointed = set()
            if po["do-endpointing"]:                       # one traceback launch for every stream still listening
                cand = [s for s, nd in zip(live, decoded) if nd > 0 and active[s]["pos"] < active[s]["wave"].size]
                if cand:
                    flags, _ = sb.endpoint_detected(ep, cand, model.tid2phone, sil, frame_shift)
                    endpointed = {s for s, f in zip(cand, flags) if f}
            for s, nd in zip(live, decoded):
                a = active[s]
                ended = a["pos"] >= a["wave"].size
                if s in endpointed:
                    # the reference breaks out of the chunk loop and goes straight to FinalizeDecoding: no InputFinished,
                    # no further AdvanceDecoding (online2-wav-nnet3-latgen-faster.cc: "if (do_endpointing && ... break;")
                    ended = True
                if ended:
                    sb.finalize([s])
                    bp = sb.best_path(s)
                    lat = sb.raw_lattice(s)
                    if bp is None or lat is None:
                        print("WARNING Decoding failed for utterance " + a["key"], file=sys.stderr); n_err += 1
                    else:
                        results[a["key"]] = kio.determinize_lattice(lat, cfg.lattice_beam, tid_phone, det)
                        like = -(bp["graph_cost"] + bp["acoustic_cost"]); nf = max(len(bp["alignment"]), 1)
                        tot_like += like; tot_frames += nf; n_done += 1
                        print("LOG Decoded utterance %s; log-like per frame is %g over %d frames." % (a["key"], like / nf, nf), file=sys.stderr)
                    if extractor is not None:
                        states[s] = sb.adaptation_state(s, max_remembered_frames=1000.0)
                    del active[s]
        for spk, utts in group:                            # written in spk2utt order
            for utt in utts:
                if utt in results:
                    writer.write(utt, results.pop(utt))
    writer.close()
    print("LOG Decoded %d utterances, %d with errors." % (n_done, n_err), file=sys.stderr)
    print("LOG Overall likelihood per frame was %g per frame over %d frames." % (tot_like / max(tot_frames, 1), tot_frames), file=sys.stderr)
    return 0 if n_done else 1


if __name__ == "__main__":
    try:
        sys.exit(main(sys.argv))
    except KamdError as e:
        print("ERROR " + str(e), file=sys.stderr)
        sys.exit(255)
