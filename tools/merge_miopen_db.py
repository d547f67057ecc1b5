#!/usr/bin/env python3
"""tools/merge_miopen_db.py TUNED_DIR -- merge a MIOpen user database written by a tuning run (tools/matcher_bench.py --tune,
bench.py with MIOPEN_USER_DB_PATH set) into the shipped one (refign_amd/miopen_db/): find-db / perf-db text files gain
the lines whose keys they do not have yet, the kernel cache (sqlite) gains the compiled kernels it does not have."""
import glob
import os
import sqlite3
import sys

src = sys.argv[1]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "refign_amd", "miopen_db")
for path in glob.glob(os.path.join(src, "*.ufdb.txt")) + glob.glob(os.path.join(src, "*.udb.txt")):
    out = os.path.join(dst, os.path.basename(path))
    have = {}
    if os.path.exists(out):
        for line in open(out):
            if "=" in line:
                have[line.split("=", 1)[0]] = line
    new = 0
    for line in open(path):
        if "=" in line and line.split("=", 1)[0] not in have:
            have[line.split("=", 1)[0]] = line
            new += 1
    with open(out, "w") as f:
        f.writelines(have.values())
    print(f"{os.path.basename(out)}: +{new} entries ({len(have)} total)")
for path in glob.glob(os.path.join(src, "*.ukdb")):
    out = os.path.join(dst, os.path.basename(path))
    if not os.path.exists(out):
        import shutil
        shutil.copy(path, out)
        print(f"{os.path.basename(out)}: copied")
        continue
    con = sqlite3.connect(out)
    before = con.execute("select count(*) from kern_db").fetchone()[0]
    con.execute("attach database ? as other", (path,))
    cols = [r[1] for r in con.execute("pragma table_info(kern_db)") if r[1] != "id"]
    cl = ", ".join(cols)
    con.execute(f"insert or ignore into kern_db ({cl}) select {cl} from other.kern_db")
    con.commit()
    after = con.execute("select count(*) from kern_db").fetchone()[0]
    con.execute("detach database other")
    con.execute("vacuum")
    con.close()
    print(f"{os.path.basename(out)}: {before} -> {after} kernels")
