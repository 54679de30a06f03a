"""Timing of the H=512 step cells alone (configs[3] nets at the bench batch): per-launch averages from the library's cell trace.
FHVAE_CELL_DBG ablations give wrong results by design (timing only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import microbench as mb  # noqa: F401  (prints its own header lines only under __main__)
hb = mb.hb
B = int(os.environ.get("PROBE_B", "2048"))
for (I, Ic) in ((80, 32), (0, 64)):
    mb.lstm(B, 20, I, Ic, 512, 2, hb.BF16)
