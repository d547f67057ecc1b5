"""Shipped tuning data for the ROCm library calls that remain on the path (dense conv via MIOpen).

MIOpen's exhaustive "find" for the ~150 convolution problems of one 1080x1920 HRDA step takes ~11 minutes on a fresh
MI355X box and makes those convs 5-15x faster than its immediate-mode heuristic (measured: teacher 3x3 1024->256 on 40
crops 157 ms -> 10 ms).  The find results (which solver per problem, a 57 kB text file) and the few JIT-compiled kernels
(0.6 MB) were captured once and are shipped in refign_amd/miopen_db/; pointing MIOpen's user database at a writable
copy of them makes every fresh box start tuned (first step 6 s instead of 690 s).  Hardware-specific by construction:
gfx950, 256 CUs.
"""
import os
import shutil
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))


def use_shipped_miopen_db():
    """Call BEFORE the first convolution.  No-op if the user already configured MIOpen's DB paths."""
    if "MIOPEN_USER_DB_PATH" in os.environ or "MIOPEN_CUSTOM_CACHE_DIR" in os.environ:
        return os.environ.get("MIOPEN_USER_DB_PATH")
    src = os.path.join(_HERE, "miopen_db")
    if not os.path.isdir(src):
        return None
    dst = os.path.join(tempfile.gettempdir(), f"refign_amd_miopen_{os.getuid()}_{os.environ.get('LOCAL_RANK', '0')}")
    os.makedirs(dst, exist_ok=True)
    for f in os.listdir(src):                       # MIOpen appends to its user DB: work on a private copy
        if not os.path.exists(os.path.join(dst, f)):
            shutil.copy(os.path.join(src, f), os.path.join(dst, f))
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = dst
    return dst
