"""gmm-latgen-faster (gmmbin/gmm-latgen-faster.cc:35-180; BASELINE configs[0], the egs/yesno plumbing case) with the
GMM log-likelihoods and the search on the device:

  gmm_latgen_faster.py [options] model-in fst-in features-rspecifier lattice-wspecifier [words-wspecifier [alignments-wspecifier]]

model-in = TransitionModel + AmDiagGmm (binary final.mdl of a GMM system); features as the recipe's pipeline
delivers them (apply-cmvn | add-deltas ...).  Options and messages follow the reference."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kaldi_amd import abi, decoder, gmm, options, table
from kaldi_amd import io as kio
from kaldi_amd._lib import KamdError, lib
from kaldi_amd.pipeline import default_sizes


def main(argv):
    po = table.ParseOptions(__doc__)
    options.register_decoder(po)
    po.register("acoustic-scale", float, 0.1, "Scaling factor for acoustic likelihoods")
    po.register("word-symbol-table", str, "", "Symbol table for words [for debug output]")
    po.register("allow-partial", bool, False, "If true, produce output even if end state was not reached.")
    po.register("batch", int, 64, "utterances decoded per launch")
    a = po.read(argv)
    if not 4 <= len(a) <= 6:
        po.print_usage()
        return 1
    cfg = options.decoder_config(po)
    am, id2pdf, tid_phone, _ = gmm.read_gmm_mdl(a[0])
    with table.Input(a[1]) as (path, off):
        G = decoder.Graph.from_file(path)
    G.hclg = type("T", (), {"tid2pdf": id2pdf})()
    dec_am = gmm.DecodableAmDiagGmmScaled(am)
    acwt = po["acoustic-scale"]
    lat_w = table.TableWriter(a[3], "compact_lattice" if po["determinize-lattice"] else "lattice", acoustic_scale=acwt)
    words_w = table.TableWriter(a[4], "int32") if len(a) > 4 and a[4] else None
    ali_w = table.TableWriter(a[5], "int32") if len(a) > 5 and a[5] else None
    det = kio.determinize_opts_default()
    det.delta, det.phone_determinize, det.word_determinize = po["delta"], int(po["phone-determinize"]), int(po["word-determinize"])
    n_done = n_err = 0
    tot_like, frames = 0.0, 0

    def flush(batch):
        nonlocal n_done, n_err, tot_like, frames
        if not batch:
            return
        lls = [dec_am.loglikes(x, acwt) for _, x in batch]            # DecodableAmDiagGmmScaled
        dec = decoder.BatchDecoder(G, cfg, default_sizes(cfg, len(batch), max(l.shape[0] for l in lls) + 2))
        lats = dec.decode(lls)
        for lane, ((key, x), lat) in enumerate(zip(batch, lats)):
            bp = dec.best_path(lane)
            reached = bool(lib().kamd_decoder_reached_final(dec._dec, lane))
            if lat is None or bp is None or (not reached and not po["allow-partial"]):
                print("WARNING Not producing output for utterance %s since no final-state reached and --allow-partial=false." % key, file=sys.stderr)
                n_err += 1
                continue
            if po["determinize-lattice"]:
                lat_w.write(key, kio.determinize_lattice(lat, cfg.lattice_beam, tid_phone, det))
            else:
                lat_w.write(key, lat)
            if words_w:
                words_w.write(key, bp["words"])
            if ali_w:
                ali_w.write(key, bp["alignment"])
            like = -(bp["graph_cost"] + bp["acoustic_cost"])
            tot_like += like; frames += x.shape[0]; n_done += 1
            print("LOG Log-like per frame for utterance %s is %g over %d frames." % (key, like / x.shape[0], x.shape[0]), file=sys.stderr)
    batch = []
    for key, x in table.SequentialTableReader(a[2], "matrix"):
        if x.shape[0] == 0:
            print("WARNING Zero-length utterance: " + key, file=sys.stderr); n_err += 1
            continue
        if x.shape[1] != am.dim:
            raise KamdError("Dim mismatch: data dim = %d vs. model dim = %d" % (x.shape[1], am.dim))
        batch.append((key, x))
        if len(batch) == po["batch"]:
            flush(batch); batch = []
    flush(batch)
    for w in (lat_w, words_w, ali_w):
        if w:
            w.close()
    print("LOG Done %d utterances, failed for %d" % (n_done, n_err), file=sys.stderr)
    print("LOG Overall log-likelihood per frame is %g over %d frames." % (tot_like / max(frames, 1), frames), file=sys.stderr)
    return 0 if n_done else 1


if __name__ == "__main__":
    try:
        sys.exit(main(sys.argv))
    except KamdError as e:
        print("ERROR " + str(e), file=sys.stderr)
        sys.exit(255)
