"""OnlineSilenceWeighting + the delta-weight queue (online2/online-ivector-feature.{h:404-535, cc:159-174, 263-306,
447-668}): the host bookkeeping behind --ivector-silence-weighting.*.  The product code (kamd_silence_weighting_*,
C++) against the oracle's statement-by-statement Python restatement on random traceback histories, and both against
cases worked by hand from the reference's rules.  parity unpinned: the reference has no test for this class."""
import numpy as np
import pytest

from kaldi_amd import online
from kaldi_amd._lib import KamdError
from oracle import orc

TID2PHONE = np.asarray([0, 1, 1, 2, 2, 3, 3, 4, 4], np.int32)      # tid -> phone, index 0 unused; phone 1 = silence


def product(sw=0.25, max_dur=-1.0, fs=1):
    cfg = online.OnlineSilenceWeightingConfig("1", sw, max_dur)
    return online.OnlineSilenceWeighting(TID2PHONE, cfg, fs)


def newest_first(tids, toks):
    return list(zip(tids[::-1], toks[::-1]))


def test_hand_worked_cases():
    w = product(sw=0.25)
    # nothing decoded yet, 4 input frames ready: everything is provisionally silence (.cc:605-611)
    assert w.GetDeltaWeights(4) == 4
    assert w.pop_until(3) == [(0, 0.25), (1, 0.25), (2, 0.25), (3, 0.25)]
    # three frames decoded: sil sil speech; 6 frames ready -> frames 3.. copy the newest decision (speech = 1.0)
    w.ComputeCurrentTraceback(3, [3, 2, 1], [30, 20, 10])
    w.GetDeltaWeights(6)
    # frames 0,1 stay 0.25 (no delta), frame 2: +0.75, frame 3: +0.75 (was 0.25), 4, 5: +1.0; the last frame is always listed
    assert w.pop_until(5) == [(2, 0.75), (3, 0.75), (4, 1.0), (5, 1.0)]
    # the traceback changes its mind about frame 2 (now silence) and extends: frame 3 silence, 4 speech
    w.ComputeCurrentTraceback(5, [5, 1, 2, 2, 1], [51, 41, 31, 20, 10])
    w.GetDeltaWeights(6)
    # frame 2: 1.0 -> 0.25, frame 3: 1.0 -> 0.25, 4 and 5 stay 1.0; the unchanged last frame is listed with a zero delta,
    # which MergePairVectorSumming then drops
    assert w.pop_until(5) == [(2, -0.75), (3, -0.75)]
    # asking for a frame nobody weighted is the reference's assertion failure
    with pytest.raises(KamdError, match="no weight was provided"):
        w.pop_until(9)


def test_max_state_duration_turns_long_runs_into_silence():
    w = product(sw=0.0, max_dur=3.0)
    # tid 5 (phone 3, not silence) repeated 4 times >= 3 -> silence; a run of 2 stays
    tids = [3, 5, 5, 5, 5, 7, 7]
    w.ComputeCurrentTraceback(7, tids[::-1], list(range(70, 63, -1)))
    w.GetDeltaWeights(7)
    got = dict(w.pop_until(6))
    assert got == {0: 1.0, 5: 1.0, 6: 1.0}          # frames 1-4 weigh 0 (delta 0 - 0 = 0: not listed)
    o = orc.OnlineSilenceWeighting(TID2PHONE, [1], 0.0, 3.0, 1)
    o.ComputeCurrentTraceback(7, newest_first(tids, list(range(64, 71))))
    assert dict((f, w_) for f, w_ in o.GetDeltaWeights(7) if w_ != 0.0) == got


def test_frame_subsampling_and_duplicates_are_summed():
    w = product(sw=0.5, fs=3)
    w.GetDeltaWeights(4)                                # ceil(4/3) = 2 decoder frames -> input frames 0..5, all 0.5
    w.ComputeCurrentTraceback(1, [3], [9])              # decoder frame 0 is speech
    w.GetDeltaWeights(5)                                # frame 0: +0.5 (x3 input frames); frame 1 copies it: +0.5
    assert w.pop_until(4) == [(0, 1.0), (1, 1.0), (2, 1.0), (3, 1.0), (4, 1.0)]
    assert w.pop_until(5) == [(5, 1.0)]


@pytest.mark.parametrize("seed,fs,max_dur", [(0, 1, -1.0), (1, 3, -1.0), (2, 3, 4.0), (3, 1, 2.0), (4, 2, 6.0)])
def test_product_equals_the_oracle_restatement_on_random_histories(seed, fs, max_dur):
    rng = np.random.default_rng(seed)
    sw = float(rng.choice([0.0, 0.001, 0.3]))
    w = product(sw=sw, max_dur=max_dur, fs=fs)
    o = orc.OnlineSilenceWeighting(TID2PHONE, [1], sw, max_dur, fs)
    q = orc.DeltaWeightQueue()
    tids, toks = [], []
    ready, updated = 0, -1
    n_neg = 0
    for tick in range(40):
        ready += int(rng.integers(0, 9))
        # the decoder lags the input; its path may rewrite a random tail
        n_dec = min((ready + fs - 1) // fs, len(tids) + int(rng.integers(0, 4)))
        keep = len(tids) - int(rng.integers(0, min(len(tids), 5) + 1))
        tids, toks = tids[:keep], toks[:keep]
        while len(tids) < n_dec:
            rep = tids and rng.random() < 0.6
            tids.append(tids[-1] if rep else int(rng.integers(1, 9)))
            toks.append(int(rng.integers(0, 1 << 20)))
        if n_dec:
            w.ComputeCurrentTraceback(n_dec, tids[::-1], toks[::-1])
            o.ComputeCurrentTraceback(n_dec, newest_first(tids, toks))
        n = w.GetDeltaWeights(ready)
        d = o.GetDeltaWeights(ready)
        assert n == len(d)
        q.UpdateFrameWeights(d)
        if ready - 1 > updated:                         # UpdateStatsUntilFrameWeighted only acts on a new frame
            want = q.pop_until(ready - 1)
            got = w.pop_until(ready - 1)
            assert [f for f, _ in got] == [f for f, _ in want]
            np.testing.assert_array_equal(np.asarray([x for _, x in got], np.float32), np.asarray([x for _, x in want], np.float32))
            n_neg += sum(1 for _, x in got if x < 0)
            updated = ready - 1
    assert n_neg > 0                                    # frames were re-classified along the way


def test_traceback_that_ends_too_early_is_rejected():
    w = product()
    with pytest.raises(KamdError, match="before reaching a known token"):
        w.ComputeCurrentTraceback(3, [1, 2], [5, 6])
    with pytest.raises(KamdError, match="outside the model"):
        w.ComputeCurrentTraceback(1, [99], [5])
