"""Summarise the rocprofv3 PMC passes of profiles/collect.sh for the dense leader tick:
    python profiles/summarize_pmc.py gpurun_out/prof_<tag> [skip_first]
Prints mean FETCH_SIZE / WRITE_SIZE (KB) over the timed launches of k_leader_tick_dense and the
HBM-side traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B (gfx950 tallies 128-B
requests as 64 B in FETCH_SIZE: MI355X_MICROARCH.md), and copies the per-kernel rows to stdout as CSV."""
import csv
import glob
import sys

d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 5  # warm-up launches of the PMC bench command
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{d}/pmc_{c}/**/*counter_collection.csv", recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if r["Kernel_Name"].startswith("void k_leader_tick_dense<") and r["Counter_Name"] == c]
    vals = vals[skip:]
    out[c] = sum(vals) / len(vals)
    print(f"{c}: {len(vals)} launches, mean {out[c]:.1f} KB")
print("traffic per launch:", int((2 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024), "B",
      f"(reads {2 * out['FETCH_SIZE'] * 1024 / 1e6:.2f} MB, writes {out['WRITE_SIZE'] * 1024 / 1e6:.2f} MB)")
