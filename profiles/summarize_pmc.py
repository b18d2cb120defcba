"""Turns two rocprofv3 --pmc runs (FETCH_SIZE, WRITE_SIZE: profiles/collect_r02.sh, profiles/collect_pmc.sh) into a
per-kernel HBM traffic summary:  summarize_pmc.py ROOT [FETCH_SUBDIR WRITE_SUBDIR OUT.json "SOURCE NOTE"].

Units / corrections (MI355X_MICROARCH.md, section HBM): FETCH_SIZE and WRITE_SIZE are in KiB-ish units of 1024 B as
reported by rocprofv3 (value * 1024 = bytes); on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced
streams, i.e. reads half the bytes -> corrected fetch = 2 x raw.  WRITE_SIZE is taken as is.  Access patterns other
than 16-B-per-lane streams are uncalibrated; both raw and corrected figures are kept."""
import csv, glob, json, os, re, sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
sub_f, sub_w = (sys.argv[2], sys.argv[3]) if len(sys.argv) > 3 else ("pmc_fetch", "pmc_write")
out_path = sys.argv[4] if len(sys.argv) > 4 else "profiles/r01_pmc_traffic.json"
note = sys.argv[5] if len(sys.argv) > 5 else "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py --no-pipeline (20 Msample stream)"


def load(sub, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                name = r["Kernel_Name"]
                m = re.search(r"fx_\w+_kernel", name)     # "void fx_walk_kernel<0>(...)" -> "fx_walk_kernel"
                name = m.group(0) if m else name
                acc[name].append(float(r["Counter_Value"]))
    return acc


fetch, write = load(sub_f, "FETCH_SIZE"), load(sub_w, "WRITE_SIZE")
out = {"source": note,
       "unit_note": "counter value x 1024 = bytes; FETCH_SIZE x2 on gfx950 (see docstring)", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("fx_"):
        continue
    f = sum(fetch.get(k, [0])) / max(len(fetch.get(k, [0])), 1) * 1024.0
    w = sum(write.get(k, [0])) / max(len(write.get(k, [0])), 1) * 1024.0
    out["kernels"][k] = {"launches_sampled": len(fetch.get(k, [])), "fetch_bytes_raw": f, "write_bytes": w,
                         "hbm_bytes_per_launch_raw": f + w, "hbm_bytes_per_launch_corrected": 2 * f + w}
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out, indent=1))
