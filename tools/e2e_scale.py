"""Files in / files out at the bench's scale (run on the GPU box): writes the synthetic tgsmall-scale HCLG, the
mini_librispeech-size final.mdl and 64 wav files, then times tools/nnet3_latgen_faster.py --wav end to end
(process start, model + graph read, decode, determinization on host threads, compressed lattice archive)."""
import os
import subprocess
import sys
import tempfile
import time
import wave

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kaldi_amd import nnet, synth
from kaldi_amd import io as kio
from tests.mdl_writer import write_mdl

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp(prefix="kamd_e2e_")
U = 1164
t0 = time.time()
g = synth.make_hclg(num_units=U, vocab=20000, n_hist=18000, fanout=(12, 64), pron_len=(3, 7), seed=2, self_loop_prob=0.5, lm_scale=0.1)
m = nnet.tdnnf_mini_librispeech(num_pdfs=g.num_pdfs)
sys.path.insert(0, root)
from bench import calibrate
calibrate(m, 1.3)
# the graph's transition-ids follow the model file's TransitionModel numbering (tests/mdl_writer.py)
write_mdl(os.path.join(tmp, "final.mdl"), m, num_units=U)
kio.write_openfst(os.path.join(tmp, "HCLG.fst"), g, "const")
durs = synth.utterance_durations(64, seed=1000)
audio = 0.0
with open(os.path.join(tmp, "wav.scp"), "w") as scp:
    for i, d in enumerate(durs):
        w = np.round(synth.make_wave(d, seed=i)).astype("<i2")
        p = os.path.join(tmp, "u%02d.wav" % i)
        with wave.open(p, "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000); f.writeframes(w.tobytes())
        scp.write("utt%02d %s\n" % (i, p))
        audio += w.size / 16000.0
print("setup %.1f s: graph %d MB, model %d MB, %d wav files (%.0f s audio)" % (
    time.time() - t0, os.path.getsize(os.path.join(tmp, "HCLG.fst")) >> 20, os.path.getsize(os.path.join(tmp, "final.mdl")) >> 20, len(durs), audio))
for threads in (1, 8):
    t1 = time.time()
    r = subprocess.run([sys.executable, root + "/tools/nnet3_latgen_faster.py", "--wav", "--beam=15", "--max-active=7000", "--lattice-beam=8",
                        "--acoustic-scale=1.0", "--frame-subsampling-factor=3", "--batch=64", "--num-threads=%d" % threads,
                        os.path.join(tmp, "final.mdl"), os.path.join(tmp, "HCLG.fst"), "scp:" + os.path.join(tmp, "wav.scp"),
                        "ark:| gzip -c > %s" % os.path.join(tmp, "lat.%d.gz" % threads)], capture_output=True, text=True)
    dt = time.time() - t1
    tail = [l for l in r.stderr.splitlines() if l.startswith("LOG Done")]
    print("--num-threads=%d: %.2f s wall for %.0f s audio = %.0f x real time, %s, lattices %d KB" % (
        threads, dt, audio, audio / dt, tail[-1] if tail else r.stderr[-300:], os.path.getsize(os.path.join(tmp, "lat.%d.gz" % threads)) >> 10))
