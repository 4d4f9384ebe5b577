"""Host glue after the decoder returns: DecodeUtteranceLatticeFaster
(decoder/decoder-wrappers.cc:201-296), Python mirror of the C++ one in include/kaldi_amd.hpp."""
import sys

from . import io as kio


def decode_utterance_lattice_faster(decoder, loglikes, utt, acoustic_scale=1.0, determinize=True, allow_partial=True,
                                    tid_phone=None, lattice_path=None, binary=True, words_out=None, log=sys.stderr):
    """decoder: kaldi_amd.decoder.LatticeFasterDecoder; loglikes: [frames x pdfs] float32.
    Appends the (compact) lattice to `lattice_path`; returns (ok, log-likelihood, words)."""
    decoder.Decode(loglikes)
    if not decoder.ReachedFinal():
        if allow_partial:
            print("WARNING Outputting partial output for utterance %s since no final-state reached" % utt, file=log)
        else:
            print("WARNING Not producing output for utterance %s since no final-state reached and "
                  "--allow-partial=false." % utt, file=log)
            return False, 0.0, None
    bp = decoder.GetBestPath()
    if bp is None:
        raise RuntimeError("Failed to get traceback for utterance " + utt)
    if words_out is not None:
        words_out[utt] = bp["words"].tolist()
    likelihood = -(float(bp["graph_cost"]) + float(bp["acoustic_cost"]))
    lat = decoder.GetRawLattice()
    if lat is None or lat.frame.size == 0:
        raise RuntimeError("Unexpected problem getting lattice for utterance " + utt)
    if lattice_path is not None:
        if determinize:
            clat = kio.determinize_lattice(lat, decoder.config.lattice_beam, tid_phone)
            if not clat.reached_beam:
                print("WARNING Determinization finished earlier than the beam for utterance " + utt, file=log)
            clat.write(lattice_path, utt, binary=binary, append=True,
                       acoustic_scale=acoustic_scale if acoustic_scale != 0.0 else 1.0)
        else:
            kio.write_lattice(lattice_path, utt, lat, binary=binary, append=True,
                              acoustic_scale=acoustic_scale if acoustic_scale != 0.0 else 1.0)
    n = max(len(bp["alignment"]), 1)
    print("LOG Log-like per frame for utterance %s is %g over %d frames." % (utt, likelihood / n, n), file=log)
    return True, likelihood, bp["words"].tolist()
