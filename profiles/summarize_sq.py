"""Per-kernel SQ counter summary of rocprofv3 --pmc runs (profiles/collect_sq.sh):  summarize_sq.py DIR [DIR ...] > out.txt
Sums every counter over all launches of a kernel (and over XCDs/SEs as rocprofv3 reports them) and prints ratios.
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave (MI355X_MICROARCH.md); ratios are unit-free."""
import csv, glob, os, re, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float)); launches = defaultdict(set)
for root in sys.argv[1:]:
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            m = re.search(r"fx_\w+_kernel(<[^>]*>)?", name)
            name = m.group(0) if m else name[:40]
            acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[name].add((root, r.get("Dispatch_Id")))
for k in sorted(acc):
    c = acc[k]
    n = max(1, len(launches[k]) // max(1, len(sys.argv) - 1))
    print("%s  (launches sampled per pass: %d)" % (k, n))
    for name in sorted(c):
        print("    %-24s %16.0f" % (name, c[name]))
    wc = c.get("SQ_WAVE_CYCLES", 0.0) / max(1, sum(1 for root in sys.argv[1:] if True))  # SQ_WAVE_CYCLES is in every pass
    def ratio(a, b="SQ_WAVE_CYCLES", scale_b=1.0):
        if c.get(a) is None or not c.get(b): return None
        return c[a] / (c[b] * scale_b)
    npass = sum(1 for root in sys.argv[1:])
    # SQ_WAVE_CYCLES and SQ_WAIT_INST_ANY were collected in both passes: halve them where they meet single-pass counters
    dup = 1.0 / npass if npass > 1 else 1.0
    for a in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"):
        if a in c and c.get("SQ_WAVE_CYCLES"):
            print("    %-24s / WAVE_CYCLES = %.3f" % (a, c[a] / (c["SQ_WAVE_CYCLES"] * dup)))
    if "SQ_WAIT_INST_ANY" in c and c.get("SQ_WAVE_CYCLES"):
        print("    %-24s / WAVE_CYCLES = %.3f" % ("SQ_WAIT_INST_ANY", c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]))
    if c.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in c:
        print("    LDS bank-conflict cycles / LDS active cycles = %.3f" % (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]))
    if c.get("SQ_INSTS_VALU") and c.get("SQ_INSTS_LDS"):
        print("    VALU insts per LDS inst = %.2f" % (c["SQ_INSTS_VALU"] / c["SQ_INSTS_LDS"]))
