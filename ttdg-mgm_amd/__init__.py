"""ttdg_mgm_amd — MI355X-native (gfx950) implementation of the TTDG-MGM
multi-graph-matching test-time-adaptation hot path.

Host side: Python mirror of the reference's GModule operator interface
(``ttdg_mgm_amd.GModule``), backed by hand-written HIP kernels behind the C ABI
declared in ``include/ttdg_mgm.h`` (``ttdg_mgm_amd._lib`` loads
``csrc/libttdg_mgm.so`` with ctypes).  There is no CPU fallback: every operator
raises if the library is missing or the tensors are not on a HIP device.
"""
import os as _os

__version__ = "0.1.0"

# The vendor convolutions (MIOpen, 75 % of an adapted batch) are chosen per shape by a heuristic unless MIOpen finds the shape in
# its find-db.  ``miopen_db/`` holds the find-db / perf-db entries of the bench shapes (4 x 3 x 800 x 800 fp32, TTA step + Dice
# pass), produced ONCE on an MI355X by ``tools/tune_miopen.sh`` (MIOpen timing its own solvers, 400 s): with it MIOpen picks the
# measured-fastest solver in immediate mode, +2.1 - 2.6 % adapted images/s, no search and no kernel compilation at run time.
# Text files keyed by device (gfx950, 256 CUs) and MIOpen version: ignored on anything else.  TTDG_MIOPEN_DB=0 (or a
# MIOPEN_USER_DB_PATH of your own) turns it off.  Must be set before the first convolution.
MIOPEN_DB = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "miopen_db")
if _os.environ.get("TTDG_MIOPEN_DB", "1") != "0" and "MIOPEN_USER_DB_PATH" not in _os.environ and _os.path.isdir(MIOPEN_DB):
    _os.environ["MIOPEN_USER_DB_PATH"] = MIOPEN_DB
