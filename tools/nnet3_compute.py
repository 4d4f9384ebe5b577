"""nnet3-compute on the device (see kaldi_amd/featbin.py for the reference lines it follows)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kaldi_amd import featbin

featbin.run("nnet3-compute")
