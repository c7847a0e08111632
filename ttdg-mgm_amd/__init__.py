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


def _private_dir(path):
    """``path`` as a directory that belongs to this user and nobody else can write to (0700), or None."""
    import stat
    try:
        _os.makedirs(path, mode=0o700, exist_ok=True)
        st = _os.lstat(path)
        if not stat.S_ISDIR(st.st_mode) or st.st_uid != _os.getuid():
            return None
        if st.st_mode & 0o077:
            _os.chmod(path, 0o700)
        return path
    except OSError:
        return None


def _stage_miopen_db():
    """MIOpen also WRITES to its user db directory (records for shapes it meets, lock and time-stamp files): give it a private
    copy - under the user's cache directory (0700, ownership checked; a fresh ``mkdtemp`` if that cannot be had), named after the
    content, so a new db in the tree is picked up - and keep the tree clean (and usable from a read-only checkout).  Several ranks
    may race here: every file appears by an atomic rename, and only files of this user are trusted."""
    import hashlib
    import shutil
    import tempfile
    files = sorted(f for f in _os.listdir(MIOPEN_DB) if f.endswith(".txt"))
    if not files:
        return None
    h = hashlib.sha1()
    for f in files:
        with open(_os.path.join(MIOPEN_DB, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    cache = _os.environ.get("XDG_CACHE_HOME") or _os.path.join(_os.path.expanduser("~"), ".cache")
    base = _private_dir(_os.path.join(cache, "ttdg_mgm_amd"))
    dst = _private_dir(_os.path.join(base, "miopen_db_" + h.hexdigest()[:12])) if base else None
    if dst is None:
        dst = tempfile.mkdtemp(prefix="ttdg_miopen_db_")          # 0700, ours, unpredictable
    for f in files:
        d = _os.path.join(dst, f)
        try:
            ours = _os.lstat(d).st_uid == _os.getuid() and not _os.path.islink(d)
        except OSError:
            ours = None                                            # missing
        if ours is None or not ours:
            tmp = "%s.%d.tmp" % (d, _os.getpid())
            shutil.copyfile(_os.path.join(MIOPEN_DB, f), tmp)
            _os.replace(tmp, d)
    return dst


def _configure_miopen_db(env=_os.environ):
    if env.get("TTDG_MIOPEN_DB", "1") != "0" and "MIOPEN_USER_DB_PATH" not in env and _os.path.isdir(MIOPEN_DB):
        try:
            env["MIOPEN_USER_DB_PATH"] = _stage_miopen_db() or MIOPEN_DB
        except OSError:
            env["MIOPEN_USER_DB_PATH"] = MIOPEN_DB


_configure_miopen_db()
