#!/usr/bin/env python3
"""tools/pmc_corr_json.py PMC.txt OUT.json -- the output of tools/pmc_corr.sh (one kernel-trace pass + separate --pmc passes of the
level-1 fused correlation) as the record bench.py's `roofline.traffic` reads: HBM-side bytes per launch, FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for 16-byte-per-lane LDS-DMA reads on gfx950 (they are tallied at half their bytes), WRITE_SIZE
as counted, both in KB."""
import json
import re
import sys

txt = open(sys.argv[1]).read()
vals = {m.group(1): float(m.group(2)) for m in re.finditer(r"^(\w+)\s+mean_per_dispatch=([0-9.e+]+)", txt, re.M)}
kt = re.search(r"kernel-trace: (.*?)\s+calls=(\d+)\s+avg=([0-9.]+) us", txt)
b, C, H, W = 2, 128, 270, 480
alg = 4 * b * H * W * (2 * C + 81)
fetch = vals["FETCH_SIZE"] * 1024 * 2
write = vals["WRITE_SIZE"] * 1024
out = {
    "kernel": (kt.group(1) if kt else "?") + " -- level 1, C=128 270x480 b=2 (bench.py's roofline kernel)",
    "FETCH_SIZE_raw_KB": vals["FETCH_SIZE"], "WRITE_SIZE_raw_KB": vals["WRITE_SIZE"],
    "fetch_bytes_corrected_x2": int(fetch), "write_bytes": write,
    "hbm_traffic_bytes_per_launch": int(fetch + write), "algorithmic_bytes_per_launch": alg,
    "traffic_over_algorithmic": round((fetch + write) / alg, 3),
    "TCC_HIT_sum": vals.get("TCC_HIT_sum"), "TCC_MISS_sum": vals.get("TCC_MISS_sum"),
    "l2_hit_rate": round(vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]), 3) if "TCC_HIT_sum" in vals else None,
    "kernel_trace_avg_us": float(kt.group(3)) if kt else None,
    "pipes": {k: int(v) for k, v in vals.items() if k.startswith("SQ_") or k.startswith("GRBM")},
    "valu_insts_per_wave": round(vals["SQ_INSTS_VALU"] / vals["SQ_WAVES"], 1) if "SQ_WAVES" in vals else None,
    "wait_share_of_wave_cycles": round(vals["SQ_WAIT_ANY"] / vals["SQ_WAVE_CYCLES"], 3) if "SQ_WAVE_CYCLES" in vals else None,
    "note": "tools/pmc_corr.sh corr_l1_fused + tools/pmc_corr_json.py: separate --pmc passes under rocprofv3, never combined with "
            "tracing domains; memory-side counters include Infinity-Cache hits",
}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: out[k] for k in ("kernel", "hbm_traffic_bytes_per_launch", "traffic_over_algorithmic", "l2_hit_rate",
                                      "kernel_trace_avg_us", "valu_insts_per_wave", "wait_share_of_wave_cycles")}))
