"""latgen-faster-mapped (bin/latgen-faster-mapped.cc) on the device decoder, files in / files out:

    python tools/latgen_faster_mapped.py [options] id2pdf.int HCLG.fst loglikes.ark lat.ark [words.ark]

  id2pdf.int   one int32-vector entry (text or binary): TransitionModel's id2pdf_id_ (index 0 unused),
               what the reference reads from final.mdl
  HCLG.fst     OpenFst vector / const FST over StdArc
  loglikes.ark Kaldi float-matrix archive [frames x pdfs] (e.g. from nnet3-compute)
  lat.ark      raw or determinized lattices (Lattice / CompactLattice archive)
Utterances are decoded in batches of --batch lanes in one launch."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kaldi_amd import abi, decoder
from kaldi_amd import io as kio


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--beam", type=float, default=16.0)
    ap.add_argument("--max-active", type=int, default=abi.INT32_MAX)
    ap.add_argument("--min-active", type=int, default=200)
    ap.add_argument("--lattice-beam", type=float, default=10.0)
    ap.add_argument("--acoustic-scale", type=float, default=0.1)
    ap.add_argument("--determinize-lattice", type=int, default=1)
    ap.add_argument("--allow-partial", type=int, default=0)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--text", action="store_true", help="write text-form archives")
    ap.add_argument("id2pdf"), ap.add_argument("hclg"), ap.add_argument("loglikes"), ap.add_argument("lattices")
    ap.add_argument("words", nargs="?")
    a = ap.parse_args()
    (_, id2pdf), = list(kio.read_int32_vector_ark(a.id2pdf))
    cfg = abi.decoder_config_default()
    cfg.beam, cfg.max_active, cfg.min_active, cfg.lattice_beam = a.beam, a.max_active, a.min_active, a.lattice_beam
    G = decoder.Graph.from_file(a.hclg)
    G.hclg = type("T", (), {"tid2pdf": id2pdf})()           # BatchDecoder takes the table from graph.hclg
    for p in (a.lattices, a.words):
        if p and os.path.exists(p):
            os.remove(p)
    items = list(kio.read_matrix_ark(a.loglikes))
    n_done = n_fail = 0
    tot_like, tot_frames = 0.0, 0
    for b0 in range(0, len(items), a.batch):
        chunk = items[b0:b0 + a.batch]
        frames = max(m.shape[0] for _, m in chunk)
        from kaldi_amd.pipeline import default_sizes
        dec = decoder.BatchDecoder(G, cfg, default_sizes(cfg, len(chunk), frames + 2))
        # the decoder expects negated costs = -loglike: acoustic scale is applied to the scores
        lats = dec.decode([np.ascontiguousarray(m * np.float32(a.acoustic_scale)) for _, m in chunk])
        for lane, ((key, m), lat) in enumerate(zip(chunk, lats)):
            bp = dec.best_path(lane)
            reached = np.isfinite(lib_frc(dec, lane))
            if lat is None or bp is None or (not reached and not a.allow_partial):
                print("WARNING Not producing output for utterance %s" % key, file=sys.stderr)
                n_fail += 1
                continue
            if a.determinize_lattice:
                clat = kio.determinize_lattice(lat, cfg.lattice_beam, None)      # no phone table here: word pass only
                clat.write(a.lattices, key, binary=not a.text, append=True, acoustic_scale=a.acoustic_scale)
            else:
                kio.write_lattice(a.lattices, key, lat, binary=not a.text, append=True, acoustic_scale=a.acoustic_scale)
            if a.words:
                with open(a.words, "a") as f:
                    f.write(key + " " + " ".join(str(w) for w in bp["words"]) + " \n")
            like = -(bp["graph_cost"] + bp["acoustic_cost"])
            tot_like += like; tot_frames += m.shape[0]; n_done += 1
            print("LOG Log-like per frame for utterance %s is %g over %d frames." % (key, like / max(m.shape[0], 1), m.shape[0]),
                  file=sys.stderr)
    print("LOG Done %d utterances, failed for %d" % (n_done, n_fail), file=sys.stderr)
    print("LOG Overall log-likelihood per frame is %g over %d frames." % (tot_like / max(tot_frames, 1), tot_frames), file=sys.stderr)
    return 0 if n_done else 1


def lib_frc(dec, lane):
    from kaldi_amd._lib import lib
    return lib().kamd_decoder_final_relative_cost(dec._dec, lane)


if __name__ == "__main__":
    sys.exit(main())
