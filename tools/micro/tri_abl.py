import os, sys, subprocess
for a in (0, 1, 2, 3, 4, 5):
    env = dict(os.environ, RFN_TRI_ABLATE=str(a), RFN_TRI_ONLY="1")
    out = subprocess.run([sys.executable, "tools/aspp_try.py"], env=env, capture_output=True, text=True).stdout
    print("ablate", a, out.strip().splitlines()[0])
