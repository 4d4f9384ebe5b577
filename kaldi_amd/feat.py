"""Host-side mirror of the reference's feature interface over the C-ABI.

`Mfcc` / `Fbank` follow OfflineFeatureTpl<MfccComputer/FbankComputer>
(feat/feature-common.h:111-160): `Dim()`, `ComputeFeatures(wave, sample_freq, vtln_warp)`.
"""
import ctypes as C

import numpy as np

from . import abi
from ._lib import KamdError, check, lib


class _Offline:
    def __init__(self, opts, vtln_warp, create):
        self.opts, self.vtln_warp = opts, float(vtln_warp)
        self._h = create(C.byref(opts), self.vtln_warp)
        if not self._h:
            raise KamdError(lib().kamd_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None):
            lib().kamd_feat_destroy(self._h)
            self._h = None

    def Dim(self):
        return lib().kamd_feat_dim(self._h)

    def NumFrames(self, num_samples):
        return lib().kamd_feat_num_frames(self._h, int(num_samples))

    def ComputeFeatures(self, wave, sample_freq=None, vtln_warp=None):
        """OfflineFeatureTpl::ComputeFeatures (feat/feature-common-inl.h:29-58); the
        sampling rate must equal --sample-frequency (no resampling on device)."""
        if sample_freq is not None and float(sample_freq) != float(self.opts.frame.samp_freq):
            raise KamdError("sample frequency mismatch (%s vs %s); resampling is out of scope"
                            % (sample_freq, self.opts.frame.samp_freq))
        if vtln_warp is not None and float(vtln_warp) != self.vtln_warp:
            raise KamdError("vtln_warp is fixed at construction (mel banks live on device)")
        wave = np.ascontiguousarray(wave, np.float32)
        T = self.NumFrames(wave.size)
        out = np.zeros((T, self.Dim()), np.float32)
        if T > 0:
            check(lib().kamd_feat_compute(self._h, abi.fptr(wave), wave.size, abi.fptr(out), T))
        return out


class Mfcc(_Offline):
    def __init__(self, opts=None, vtln_warp=1.0):
        super().__init__(opts or abi.mfcc_opts_default(), vtln_warp, lib().kamd_mfcc_create)


class Fbank(_Offline):
    def __init__(self, opts=None, vtln_warp=1.0):
        super().__init__(opts or abi.fbank_opts_default(), vtln_warp, lib().kamd_fbank_create)
