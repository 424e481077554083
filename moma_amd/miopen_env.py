"""MIOpen user find-db for the backbones' convolutions on gfx950.

MIOpen's immediate mode has no tuned entries for EfficientNet's depthwise / grouped convolutions on gfx950 and
falls back to `naive_conv_*` kernels (1.5x slower train step).  `moma_amd/miopen_db/` holds the user find-db
(`*.ufdb.txt`) and perf-db (`*.udb.txt`) written by ONE run of the benchmark in find mode
(`python bench.py --miopen_find` with MIOPEN_USER_DB_PATH pointing at an empty directory): plain text records
"problem -> best solver", no code.  `use_shipped_db()` copies them into a private writable directory and points
MIOPEN_USER_DB_PATH at it; it must run before the first convolution (call it before importing torch to be safe).
Backbones are out of scope as kernels -- this only selects among MIOpen's own solvers.
"""
import os
import shutil
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
DB_DIR = os.path.join(_HERE, "miopen_db")


def use_shipped_db(tag: str = "") -> str:
    if os.environ.get("MIOPEN_USER_DB_PATH"):           # caller decided already
        return os.environ["MIOPEN_USER_DB_PATH"]
    dst = os.path.join(tempfile.gettempdir(), f"moma_miopen_db_{os.getuid()}_{tag or os.getpid()}")
    os.makedirs(dst, exist_ok=True)
    if os.path.isdir(DB_DIR):
        for f in os.listdir(DB_DIR):
            if f.endswith(".txt") and not os.path.exists(os.path.join(dst, f)):
                shutil.copy(os.path.join(DB_DIR, f), os.path.join(dst, f))
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    return dst
