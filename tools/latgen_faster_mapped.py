"""latgen-faster-mapped (bin/latgen-faster-mapped.cc) on the device decoder, files in / files out:

    latgen_faster_mapped.py [options] <trans-model|id2pdf> <fst-in> <loglikes-rspecifier> <lattice-wspecifier> [<words-wspecifier> [<alignments-wspecifier>]]

  trans-model  a binary final.mdl (its TransitionModel gives id2pdf and the phones for determinization), or an int32-vector
               archive entry holding the id2pdf table (index 0 unused)
  fst-in       OpenFst vector / const FST over StdArc
  loglikes     Kaldi float matrices [frames x pdfs] (e.g. from nnet3-compute); plain paths are taken as ark:<path>
Utterances are decoded in batches of --batch lanes in one launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kaldi_amd import abi, decoder, mdl, options, table
from kaldi_amd import io as kio
from kaldi_amd._lib import KamdError, lib
from kaldi_amd.pipeline import default_sizes


def spec(x, kind):
    return x if (table.classify_rspecifier(x)[0] if kind == "r" else table.classify_wspecifier(x)[0]) != table.NO_SPECIFIER else "ark:" + x


def main(argv):
    po = table.ParseOptions(__doc__)
    options.register_decoder(po)
    po.register("acoustic-scale", float, 0.1, "Scaling factor for acoustic likelihoods")
    po.register("word-symbol-table", str, "", "Symbol table for words [for debug output]")
    po.register("allow-partial", bool, False, "If true, produce output even if end state was not reached.")
    po.register("batch", int, 64, "utterances decoded per launch")
    po.register("text", bool, False, "write text-form archives (same as ark,t:)")
    a = po.read(argv)
    if not 4 <= len(a) <= 6:
        po.print_usage()
        return 1
    cfg = options.decoder_config(po)
    tid_phone = None
    with table.Input(a[0]) as (path, off):
        head = open(path, "rb").read(64)
        if b"<TransitionModel>" in head:
            s = mdl._Stream(open(path, "rb").read())
            s.take(2)
            id2pdf, tid_phone, _ = mdl.read_transition_model(s)
        else:
            (_, id2pdf), = list(kio.read_int32_vector_ark(path))
    with table.Input(a[1]) as (path, off):
        G = decoder.Graph.from_file(path)
    G.hclg = type("T", (), {"tid2pdf": id2pdf})()           # BatchDecoder takes the table from graph.hclg
    acwt = po["acoustic-scale"]
    t = ",t" if po["text"] else ""
    wspec = spec(a[3], "w")
    if po["text"] and wspec.startswith("ark:"):
        wspec = "ark,t:" + wspec[4:]
    lat_w = table.TableWriter(wspec, "compact_lattice" if po["determinize-lattice"] else "lattice", acoustic_scale=acwt)
    words_w = table.TableWriter(spec(a[4], "w").replace("ark:", "ark,t:", 1) if len(a) > 4 and table.classify_wspecifier(a[4])[0] == table.NO_SPECIFIER
                                else a[4], "int32") if len(a) > 4 and a[4] else None
    ali_w = table.TableWriter(spec(a[5], "w"), "int32") if len(a) > 5 and a[5] else None
    det = kio.determinize_opts_default()
    det.delta, det.phone_determinize, det.word_determinize = po["delta"], int(po["phone-determinize"]), int(po["word-determinize"])
    n_done = n_fail = 0
    tot_like, tot_frames = 0.0, 0

    def flush(chunk):
        nonlocal n_done, n_fail, tot_like, tot_frames
        if not chunk:
            return
        frames = max(m.shape[0] for _, m in chunk)
        dec = decoder.BatchDecoder(G, cfg, default_sizes(cfg, len(chunk), frames + 2))
        # the decoder expects negated costs = -loglike: acoustic scale is applied to the scores
        lats = dec.decode([np.ascontiguousarray(m * np.float32(acwt)) for _, m in chunk])
        for lane, ((key, m), lat) in enumerate(zip(chunk, lats)):
            bp = dec.best_path(lane)
            reached = bool(lib().kamd_decoder_reached_final(dec._dec, lane))
            if lat is None or bp is None or (not reached and not po["allow-partial"]):
                print("WARNING Not producing output for utterance %s since no final-state reached and --allow-partial=false." % key, file=sys.stderr)
                n_fail += 1
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
            tot_like += like; tot_frames += m.shape[0]; n_done += 1
            print("LOG Log-like per frame for utterance %s is %g over %d frames." % (key, like / max(m.shape[0], 1), m.shape[0]), file=sys.stderr)
    chunk = []
    for key, m in table.SequentialTableReader(spec(a[2], "r"), "matrix"):
        if m.shape[0] == 0:
            print("WARNING Zero-length utterance: " + key, file=sys.stderr); n_fail += 1
            continue
        chunk.append((key, m))
        if len(chunk) == po["batch"]:
            flush(chunk); chunk = []
    flush(chunk)
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
