"""lattice-scale (host tool; see kaldi_amd/latbin.py for the reference lines it follows)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kaldi_amd import latbin

latbin.run("lattice-scale")
