"""ttdg_mgm_amd — MI355X-native (gfx950) implementation of the TTDG-MGM
multi-graph-matching test-time-adaptation hot path.

Host side: Python mirror of the reference's GModule operator interface
(``ttdg_mgm_amd.GModule``), backed by hand-written HIP kernels behind the C ABI
declared in ``include/ttdg_mgm.h`` (``ttdg_mgm_amd._lib`` loads
``csrc/libttdg_mgm.so`` with ctypes).  There is no CPU fallback: every operator
raises if the library is missing or the tensors are not on a HIP device.
"""
__version__ = "0.1.0"
