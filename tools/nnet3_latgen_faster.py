"""nnet3-latgen-faster (nnet3bin/nnet3-latgen-faster.cc) end to end on the device, files in / files out:

    python tools/nnet3_latgen_faster.py [options] final.mdl HCLG.fst wav.scp lat.ark [words.ark]

  final.mdl   binary chain TDNN / TDNN-F model (kaldi_amd/mdl.py)
  HCLG.fst    OpenFst vector / const FST
  wav.scp     lines "utt-id /path/to/file.wav" (16-bit PCM RIFF; pipes are not supported)
  lat.ark     CompactLattice archive (or Lattice with --determinize-lattice=0)
Features are MFCC with the options of conf/mfcc_hires.conf, computed on the device.  Utterances
are decoded --batch at a time, whole path in one pass per batch."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kaldi_amd import abi, decoder, mdl, pipeline
from kaldi_amd import io as kio


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--beam", type=float, default=15.0)
    ap.add_argument("--max-active", type=int, default=7000)
    ap.add_argument("--min-active", type=int, default=200)
    ap.add_argument("--lattice-beam", type=float, default=8.0)
    ap.add_argument("--acoustic-scale", type=float, default=1.0)
    ap.add_argument("--frame-subsampling-factor", type=int, default=3)
    ap.add_argument("--frames-per-chunk", type=int, default=50)
    ap.add_argument("--online-ivectors", default="", help="matrix archive, one entry per utterance")
    ap.add_argument("--online-ivector-period", type=int, default=10)
    ap.add_argument("--determinize-lattice", type=int, default=1)
    ap.add_argument("--allow-partial", type=int, default=0)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--text", action="store_true")
    ap.add_argument("model"), ap.add_argument("hclg"), ap.add_argument("wav_scp"), ap.add_argument("lattices")
    ap.add_argument("words", nargs="?")
    a = ap.parse_args()
    model, id2pdf, tid_phone = mdl.read_mdl(a.model, a.acoustic_scale, a.frame_subsampling_factor)
    cfg = abi.decoder_config_default()
    cfg.beam, cfg.max_active, cfg.min_active, cfg.lattice_beam = a.beam, a.max_active, a.min_active, a.lattice_beam
    g = kio.read_openfst(a.hclg)
    g.tid2pdf, g.num_pdfs = id2pdf, model.num_pdfs
    ivecs = dict(kio.read_matrix_ark(a.online_ivectors)) if a.online_ivectors else None
    utts = [l.split(None, 1) for l in open(a.wav_scp) if l.strip()]
    for p in (a.lattices, a.words):
        if p and os.path.exists(p):
            os.remove(p)
    n_done = n_fail = 0
    tot_like, tot_frames, audio = 0.0, 0, 0.0
    pipe = None
    for b0 in range(0, len(utts), a.batch):
        chunk = utts[b0:b0 + a.batch]
        waves = []
        for key, path in chunk:
            sf, data = kio.read_wave(path.strip())
            if sf != 16000.0:
                raise SystemExit("%s: sampling rate %g, the hires MFCC config expects 16000" % (key, sf))
            waves.append(data[0])
            audio += data.shape[1] / sf
        max_s = max(w.size for w in waves) / 16000.0 + 0.5
        if pipe is None or max_s > pipe_max_s or len(waves) > pipe_n:
            pipe = pipeline.Pipeline(abi.mfcc_opts_hires(), model, g, cfg, max_utts=len(waves), max_seconds=max_s)
            pipe_max_s, pipe_n = max_s, len(waves)
        pipe.load(waves)
        if ivecs is not None:
            pipe.set_online_ivectors([ivecs[k] for k, _ in chunk], a.online_ivector_period, a.frames_per_chunk)
        pipe.run(auto_grow=4)
        for (key, _), res in zip(chunk, pipe.results(lattices=True)):
            bp = None if res is None else res["best"]
            if bp is None or res["lattice"] is None:
                print("WARNING Failed to decode utterance with id " + key, file=sys.stderr)
                n_fail += 1
                continue
            lat = res["lattice"]
            if a.determinize_lattice:
                kio.determinize_lattice(lat, cfg.lattice_beam, tid_phone).write(
                    a.lattices, key, binary=not a.text, append=True, acoustic_scale=a.acoustic_scale)
            else:
                kio.write_lattice(a.lattices, key, lat, binary=not a.text, append=True, acoustic_scale=a.acoustic_scale)
            if a.words:
                with open(a.words, "a") as f:
                    f.write(key + " " + " ".join(str(w) for w in bp["words"]) + " \n")
            like = -(bp["graph_cost"] + bp["acoustic_cost"])
            nf = max(len(bp["alignment"]), 1)
            tot_like += like; tot_frames += nf; n_done += 1
            print("LOG Log-like per frame for utterance %s is %g over %d frames." % (key, like / nf, nf), file=sys.stderr)
    print("LOG Done %d utterances, failed for %d (%.1f s of audio)" % (n_done, n_fail, audio), file=sys.stderr)
    print("LOG Overall log-likelihood per frame is %g over %d frames." % (tot_like / max(tot_frames, 1), tot_frames), file=sys.stderr)
    return 0 if n_done else 1


if __name__ == "__main__":
    sys.exit(main())
