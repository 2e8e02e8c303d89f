"""Write HBM traffic figures into profiles/traffic.json together with what they are valid for:
    python profiles/update_traffic.py <key> <bytes> <from> [<key> <bytes> <from> ...]
e.g.  python profiles/update_traffic.py G1000000_R5_mode0 68606976 profiles/r03/pmc_FETCH_SIZE_1M.csv+pmc_WRITE_SIZE_1M.csv
Each entry is stamped with the sha of josefine_amd/csrc/jg_dense.h at collection time; bench.py refuses the
figure (roofline.traffic.bytes = null) once that file has changed."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_sha  # noqa: E402

path = os.path.join(ROOT, "profiles", "traffic.json")
data = json.load(open(path))
meta = data.setdefault("_collected", {})
args = sys.argv[1:]
for i in range(0, len(args), 3):
    key, val, frm = args[i], int(args[i + 1]), args[i + 2]
    data[key] = val
    meta[key] = {"kernel_sha": kernel_source_sha(), "from": frm}
json.dump(data, open(path, "w"), indent=1)
print("updated", [args[i] for i in range(0, len(args), 3)])
