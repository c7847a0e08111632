"""ctypes wrapper over oracle/lap.c (TEST INFRASTRUCTURE, see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_lap.so")


def build():
    src = os.path.join(_HERE, "lap.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "_build/liboracle_lap.so"])
    return _SO


def lap(cost, maximize=False):
    """Returns col_of_row (int64, -1 = unassigned) for a 2-D cost array."""
    lib = ctypes.CDLL(build())
    c = np.ascontiguousarray(cost, dtype=np.float64)
    out = np.empty(c.shape[0], dtype=np.int64)
    rc = lib.ttdg_oracle_lap(c.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(c.shape[0]), ctypes.c_int(c.shape[1]),
                             ctypes.c_int(int(maximize)), out.ctypes.data_as(ctypes.c_void_p))
    if rc != 0:
        raise ValueError("cost matrix is infeasible")
    return out
