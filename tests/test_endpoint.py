"""Endpointing rules (online2/online-endpoint.{h,cc}): pure host arithmetic on
(frames decoded, trailing silence, final relative cost)."""
import numpy as np

from kaldi_amd.online import OnlineEndpointConfig, endpoint_detected, trailing_silence_length


def test_default_rules():
    c = OnlineEndpointConfig()
    inf = float("inf")
    fs = 0.03
    # rule1: 5 s of silence even if nothing was decoded
    assert endpoint_detected(c, 200, 200, fs, inf)            # 6 s, all silence
    assert not endpoint_detected(c, 100, 100, fs, inf)        # 3 s, all silence
    # rule2: 0.5 s of trailing silence after speech with a good final cost
    assert endpoint_detected(c, 100, 17, fs, 1.9)
    assert not endpoint_detected(c, 100, 17, fs, 2.1)         # cost too high for rule2, too short for rule3
    assert not endpoint_detected(c, 100, 16, fs, 0.0)         # 0.48 s
    # rule3: 1 s and relative cost <= 8
    assert endpoint_detected(c, 100, 34, fs, 7.9)
    assert not endpoint_detected(c, 100, 34, fs, 8.1)
    # rule4: 2 s regardless of the cost, but only after speech
    assert endpoint_detected(c, 100, 67, fs, inf)
    assert not endpoint_detected(c, 67, 67, fs, inf)          # nothing but silence: only rule1 applies
    # rule5: 20 s of anything
    assert endpoint_detected(c, 667, 0, fs, inf)
    assert not endpoint_detected(c, 666, 0, fs, inf)


def test_trailing_silence_counts_frames_back_to_the_first_non_silence():
    tid2phone = np.array([0, 1, 1, 2, 2, 3, 3])               # tids 1,2 -> phone 1 (sil); 3,4 -> 2; 5,6 -> 3
    bp = {"alignment": np.array([1, 2, 3, 4, 4, 5, 1, 2, 2, 2])}
    assert trailing_silence_length(bp, tid2phone, [1]) == 4
    assert trailing_silence_length(bp, tid2phone, [1, 3]) == 5
    assert trailing_silence_length(bp, tid2phone, [2]) == 0
    assert trailing_silence_length(None, tid2phone, [1]) == 0
