#!/usr/bin/env python3
"""tools/pmc_sub.py DIR SUB[,SUB...] -- what tools/pmc_sub.sh collected, per sub-record: every kernel of the decode call
with its mean duration (kernel trace), launches per step, and its HBM counter bytes per step -- FETCH_SIZE (KiB,
doubled: gfx950 tallies a 128-byte read request as 64 bytes, MI355X_MICROARCH.md; calibrated on k_sync_mask, which
reads the input once) and WRITE_SIZE (KiB) -- beside the sub-record's algorithmic bytes.  Writes DIR/sub_traffic.json
(entries for profiles/sub_traffic.json, which bench.py carries into the sub-records' `roofline.traffic`)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out_dir, subs = sys.argv[1], sys.argv[2].split(",")
STEPS = 4          # (replaced per sub-record by the number of its steps found in the trace)


def short(n):
    n = n.replace("void ", "").replace("mlp::", "")
    return n.split("(")[0][:52]


res = {}
for sub in subs:
    d = os.path.join(out_dir, sub)
    bench = None
    try:
        for line in open(os.path.join(d, "bench.json")):
            if line.startswith("{"):
                bench = json.loads(line)
    except OSError:
        pass
    rec = (bench or {}).get("sub", {}).get(sub)

    # ---- which dispatches are the sub-record's?  The headline batch runs in the same process and launches kernels of
    #      the same names (k_sync_mask, k_decode<6, ..>): a mean over all launches of a name mixes two batches (round 4's
    #      table did).  Every index call begins with k_init_streams: the dispatches are cut into steps there, in
    #      dispatch order; the sub-record runs last, so the steps whose k_sync_mask grid is the LAST step's are its own.
    def own_rows(rows, name_col, grid_col):
        rows = sorted(rows, key=lambda r: int(r["Dispatch_Id"]))
        steps, cur = [], []
        for r in rows:
            if "k_init_streams" in r[name_col] and cur:
                steps.append(cur)
                cur = []
            cur.append(r)
        if cur:
            steps.append(cur)

        def grid(step):
            for r in step:
                if "k_sync_mask" in r[name_col]:
                    return r[grid_col]
            return None
        steps = [st for st in steps if grid(st) is not None]
        if not steps:
            return [], 0
        g = grid(steps[-1])
        mine = [st for st in steps if grid(st) == g]
        return [r for st in mine for r in st], len(mine)

    dur_acc, n_steps = defaultdict(list), 0
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
        rows, n_steps = own_rows(list(csv.DictReader(open(f))), "Kernel_Name", "Grid_Size_X")
        for r in rows:
            dur_acc[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    dur = {k: (sum(v) / len(v), len(v)) for k, v in dur_acc.items()}
    STEPS = max(n_steps, 1)
    ctr = defaultdict(lambda: defaultdict(list))
    for which in ("fetch", "write"):
        for f in glob.glob(os.path.join(d, which, "**", "*counter_collection.csv"), recursive=True):
            rows, _ = own_rows(list(csv.DictReader(open(f))), "Kernel_Name", "Grid_Size")
            for r in rows:
                ctr[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== %s" % sub)
    if rec:
        print("   %s Msamples/s, %.3f ms/step, decode call %.3f ms on the device; algorithmic bytes per step %.3f GB" % (
            rec["value"], rec["ms_per_step"], rec["kernel_ms"], rec["algorithmic_bytes_per_launch"] / 1e9))
    per_kernel = {}
    tot = 0.0
    for k in sorted(dur, key=lambda k: -dur[k][0] * dur[k][1]):
        ms, calls = dur[k]
        if ms * calls < 0.05 or k.startswith("__amd") or k.startswith("at::"):
            continue
        fe = ctr[k].get("FETCH_SIZE")
        wr = ctr[k].get("WRITE_SIZE")
        fb = 2.0 * 1024 * sum(fe) / len(fe) if fe else None
        wb = 1024 * sum(wr) / len(wr) if wr else None
        print("   %-46s %8.3f ms x %5.1f per step   read %8s  written %8s  (GB per launch)" % (
            k, ms, calls / STEPS, "%.3f" % (fb / 1e9) if fb is not None else "-", "%.3f" % (wb / 1e9) if wb is not None else "-"))
        per_kernel[k] = {"ms": round(ms, 4), "launches_per_step": round(calls / STEPS, 2),
                         "fetch_bytes": int(fb) if fb is not None else None, "write_bytes": int(wb) if wb is not None else None}
    if rec:
        # bytes of the sub-record's own step: every kernel of ITS dispatches, weighted by how often a step launches it
        # (the chain passes: k_chain_*, the cooperative parse k_coop<true..>, and the PARSE instance of k_decode --
        #  its fifth template argument; round 5 added a sixth and the match by the name's tail missed the parse pass)
        def parse_instance(k):
            if not k.startswith("k_decode<"):
                return False
            targs = [t.strip() for t in k[len("k_decode<"):].rstrip(">").split(",")]
            return len(targs) >= 5 and targs[4] == "true"
        own = [k for k in per_kernel if "k_chain" in k or k.startswith("k_coop<true") or parse_instance(k)]
        tb = sum((per_kernel[k]["fetch_bytes"] or 0) + (per_kernel[k]["write_bytes"] or 0) for k in own)
        res[sub] = {"samples_per_step": rec["samples_per_step"], "compressed_bytes": rec["compressed_bytes"],
                    "hbm_bytes_per_step": int(tb), "kernels_counted": own, "per_kernel": per_kernel,
                    "algorithmic_bytes": rec["algorithmic_bytes_per_launch"], "source": os.path.basename(out_dir.rstrip("/"))}
        print("   passes behind the fast pass (%s): %.2f GB of counter traffic per step = %.2f x the step's algorithmic bytes" % (
            ", ".join(own), tb / 1e9, tb / max(rec["algorithmic_bytes_per_launch"], 1)))
json.dump(res, open(os.path.join(out_dir, "sub_traffic.json"), "w"), indent=1)
