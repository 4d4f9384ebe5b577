"""kamd_ivector_info_read (csrc/ivector_io.cc: the files of an i-vector extraction config as a C / C++ host reads them)
against the Python reader of the same files: every field bit for bit, incl. the recomputed UBM gconsts."""
import numpy as np
import pytest

from kaldi_amd import ivector
from kaldi_amd._lib import KamdError


def same(a, b):
    for f in ("lda", "global_cmvn_stats", "ubm_means_invvars", "ubm_inv_vars", "M", "sigma_inv", "ubm_gconsts"):
        x, y = getattr(a, f), getattr(b, f)
        assert x.dtype == y.dtype and x.shape == y.shape, f
        np.testing.assert_array_equal(x, y, err_msg=f)
    for f in ("prior_offset", "splice_left", "splice_right", "cmn_window", "speaker_frames", "global_frames", "normalize_mean",
              "normalize_variance", "ivector_period", "num_gselect", "num_cg_iters", "feat_dim", "ivector_dim"):
        assert getattr(a, f) == getattr(b, f), f
    for f in ("min_post", "posterior_scale", "max_count"):
        assert np.float32(getattr(a, f)) == np.float32(getattr(b, f)), f


@pytest.mark.parametrize("opts", [dict(), dict(feat_dim=13, lda_dim=20, num_gauss=33, ivector_dim=17, splice_left=2, splice_right=1, ivector_period=5,
                                               num_gselect=3, min_post=0.1, posterior_scale=0.25, max_count=75.0, cmn_window=300,
                                               speaker_frames=200, global_frames=100)])
def test_native_reader_equals_the_python_reader(tmp_path, opts):
    info = ivector.make_synthetic(seed=5, **opts)
    conf = ivector.write_config_dir(tmp_path / "extractor", info)
    py = ivector.IvectorExtractionInfo.from_config(conf)
    nat = ivector.read_config_native(conf)
    same(nat, py)
    same(nat, info)                      # and both give back what was written


def test_native_reader_errors_name_the_problem(tmp_path):
    info = ivector.make_synthetic(feat_dim=8, lda_dim=6, num_gauss=8, ivector_dim=5, seed=1)
    conf = ivector.write_config_dir(tmp_path / "x", info)
    text = open(conf).read()
    with pytest.raises(KamdError, match="Cannot open config file"):
        ivector.read_config_native(tmp_path / "absent.conf")
    (tmp_path / "a.conf").write_text(text.replace("--diag-ubm=", "--diag-ubn="))
    with pytest.raises(KamdError, match="Invalid option --diag-ubn"):
        ivector.read_config_native(tmp_path / "a.conf")
    (tmp_path / "b.conf").write_text("\n".join(l for l in text.split("\n") if not l.startswith("--lda-matrix")))
    with pytest.raises(KamdError, match="--lda-matrix option must be set"):
        ivector.read_config_native(tmp_path / "b.conf")
    ie = (tmp_path / "x" / "final.ie").read_bytes()
    (tmp_path / "x" / "final.ie").write_bytes(ie[:len(ie) // 2])
    with pytest.raises(KamdError, match="unexpected end of file"):
        ivector.read_config_native(conf)
