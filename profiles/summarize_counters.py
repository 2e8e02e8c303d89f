"""Mean value per (kernel, counter) of rocprofv3 --pmc passes:
    python profiles/summarize_counters.py <pass dir> [<pass dir> ...]
For every *counter_collection.csv below the given directories: per kernel (name cut at the
argument list) and counter, the number of dispatches and the mean / min / max counter value over
the LAST 60 % of that kernel's dispatches (warm-up launches and set-up steps come first).
KERNEL=<substring> restricts the kernels."""
import csv
import glob
import os
import sys
from collections import defaultdict

want = os.environ.get("KERNEL", "")
for d in sys.argv[1:]:
    for f in sorted(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)):
        vals = defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if want and want not in k:
                continue
            vals[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        print(f"== {os.path.relpath(f)}")
        for (k, c), v in sorted(vals.items()):
            tail = v[len(v) * 2 // 5:]
            print(f"{k:48s} {c:28s} n={len(v):4d} mean={sum(tail) / len(tail):16.1f} min={min(tail):14.1f} max={max(tail):14.1f}")
