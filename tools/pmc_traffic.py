#!/usr/bin/env python3
"""tools/pmc_traffic.py PMC_DIR BENCH_JSON -> profiles/traffic.json

HBM-side bytes of one fast-pass k_decode launch from the separate FETCH_SIZE / WRITE_SIZE passes
of tools/prof_pmc.sh, corrected as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE (KiB)
tallies 128-byte read requests as 64 bytes, so it is doubled; WRITE_SIZE (KiB) is exact.  The
correction is calibrated in the same run on k_sync_mask, which streams the whole input once
(its doubled FETCH_SIZE must equal the input size)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

pmc, bench = sys.argv[1], sys.argv[2]
vals = defaultdict(list)
for f in glob.glob(os.path.join(pmc, "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "GRBM_GUI_ACTIVE"):
            name = row["Kernel_Name"]
            m = re.search(r"k_decode<\s*\d+\s*,\s*(\w+)\s*,\s*(\w+)", name)      # <NS, PAIRED, GENERAL[, ILV, PARSE, DUO, WAVO]>
            targs = [t.strip() for t in name[name.index("<") + 1:name.index(">")].split(",")] if m else []
            duo = len(targs) > 5 and targs[5] == "true"
            # the one-substream fast-pass kernel (the two-substream kernel of the same launch sequence exits at once on a
            # batch without two-substream streams and would halve the mean)
            key = "decode_fast" if (m and m.group(1) == "false" and m.group(2) == "false" and not duo) else \
                  "sync_mask" if "k_sync_mask" in name else None
            if key:
                vals[(key, row["Counter_Name"])].append(float(row["Counter_Value"]))
mean = lambda k: sum(vals[k]) / len(vals[k])
j = None
for line in open(bench):
    if line.startswith("{"):
        j = json.loads(line)
cfg = j["config"]
fetch = 2.0 * mean(("decode_fast", "FETCH_SIZE")) * 1024
write = mean(("decode_fast", "WRITE_SIZE")) * 1024
calib = 2.0 * mean(("sync_mask", "FETCH_SIZE")) * 1024
out = {
    "kernel": "k_decode<6,false,false,%s> (fast pass, %s PCM)" % (
        ("true", "interleaved") if "interleaved" in cfg.get("pcm_layout", "") else ("false", "planar")),
    "hbm_bytes_per_launch": int(fetch + write),
    "fetch_bytes_corrected": int(fetch), "write_bytes": int(write),
    "calibration": {"k_sync_mask_fetch_corrected": int(calib), "input_bytes": cfg["compressed_bytes_per_gpu"]},
    "samples_per_launch": cfg["samples_per_step_per_gpu"], "compressed_bytes": cfg["compressed_bytes_per_gpu"],
    "pcm_layout": "interleaved" if "interleaved" in cfg.get("pcm_layout", "") else "planar",
    "algorithmic_bytes": j["roofline"]["algorithmic_bytes_per_launch"],
    "source": os.path.basename(pmc.rstrip("/")),
}
# instruction-issue side of the same launch (wave instructions; the bench's "issue" roofline uses them)
if vals.get(("decode_fast", "SQ_INSTS_VALU")):
    out["valu_insts_per_launch"] = int(mean(("decode_fast", "SQ_INSTS_VALU")))
if vals.get(("decode_fast", "SQ_INSTS_SALU")):
    out["salu_insts_per_launch"] = int(mean(("decode_fast", "SQ_INSTS_SALU")))
if vals.get(("decode_fast", "GRBM_GUI_ACTIVE")) and j["roofline"].get("kernel_ms"):
    # shader clock during the kernel = busy cycles / its duration (the profiled run's own duration would be
    # better; the bench's kernel_ms of the same build is what is at hand)
    out["gui_active_cycles_per_launch"] = int(mean(("decode_fast", "GRBM_GUI_ACTIVE")))
    # (the counter is summed over the 8 XCDs)
    out["shader_clock_ghz"] = round(out["gui_active_cycles_per_launch"] / 8 / (j["roofline"]["kernel_ms"] * 1e-3) / 1e9, 2)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out))
