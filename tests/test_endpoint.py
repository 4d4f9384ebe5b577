"""Endpointing rules (online2/online-endpoint.{h,cc}): pure host arithmetic on
(frames decoded, trailing silence, final relative cost)."""
import pytest
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


def _rules(c):
    return [[float(r.must_contain_nonsilence), r.min_trailing_silence, r.max_relative_cost, r.min_utterance_length]
            for r in (c.rule1, c.rule2, c.rule3, c.rule4, c.rule5)]


def test_c_abi_endpoint_rules_equal_the_oracle_on_a_grid():
    """kamd_endpoint_detected (host arithmetic behind the C-ABI) against the oracle's restatement, default and
    modified rules, around every threshold (float products like 17 * 0.03 decide the comparisons)."""
    from oracle import orc
    from kaldi_amd import decoder
    from kaldi_amd._lib import KamdError
    c = OnlineEndpointConfig()
    c2 = OnlineEndpointConfig()
    c2.rule1.min_trailing_silence = 1.5; c2.rule2.max_relative_cost = 0.5; c2.rule3.must_contain_nonsilence = False
    c2.rule5.min_utterance_length = 3.0; c2.rule4.min_trailing_silence = 0.6
    n = 0
    for cfg in (c, c2):
        rules = _rules(cfg)
        for fs in (0.03, 0.01, 0.04):
            for nfd in (0, 1, 16, 17, 34, 50, 67, 100, 167, 200, 666, 667):
                for tsf in sorted({0, 1, 16, 17, 33, 34, 50, 66, 67, 166, 167, nfd}):
                    if tsf > nfd:
                        continue
                    for frc in (0.0, 0.5, 1.9, 2.0, 2.1, 7.9, 8.0, 8.1, float("inf")):
                        assert endpoint_detected(cfg, nfd, tsf, fs, frc) == orc.endpoint_detected(rules, nfd, tsf, fs, frc), (nfd, tsf, fs, frc)
                        n += 1
    assert n > 3000
    d = decoder.endpoint_config_default()                     # kamd_endpoint_config_default == OnlineEndpointConfig()
    for i, r in enumerate(_rules(c)):
        got = d.rule[i]
        assert [float(got.must_contain_nonsilence), got.min_trailing_silence, got.max_relative_cost, got.min_utterance_length] == r
    with pytest.raises(KamdError):                            # KALDI_ASSERT(num_frames_decoded >= trailing_silence_frames)
        endpoint_detected(c, 5, 6, 0.03, 0.0)


def test_oracle_trailing_silence_equals_the_host_walk():
    from oracle import orc
    rng = np.random.default_rng(0)
    tid2phone = np.concatenate([[0], np.repeat(np.arange(1, 11), 2)]).astype(np.int32)
    for _ in range(50):
        ali = rng.integers(1, 21, rng.integers(0, 40)).astype(np.int32)
        sil = rng.choice(np.arange(1, 11), rng.integers(1, 8), replace=False)
        assert orc.trailing_silence_length(ali, tid2phone, sil) == trailing_silence_length({"alignment": ali}, tid2phone, sil)
