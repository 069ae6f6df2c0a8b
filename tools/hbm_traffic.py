#!/usr/bin/env python3
"""profiles/hbm_traffic.json from the two counter passes of tools/profile_bench.sh.

    python3 tools/hbm_traffic.py gpurun_out/prof_TAG [profiles/hbm_traffic.json]

Reads the rocprofv3 counter_collection.csv of the FETCH_SIZE and the WRITE_SIZE pass, averages the
launches of the fused MFCC-39 kernel, applies the gfx950 correction (FETCH_SIZE counts a 128-byte
request of a 16-byte-per-lane streaming read as 64 bytes: doubled; WRITE_SIZE as reported) and
records the sha1 of mfcc_amd/csrc/* the library was built from, so that bench.py only quotes the
figure for the kernel it belongs to."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_revision, N_UTT, UTT_LEN, FRAMES_PER_UTT  # noqa: E402

KERNEL = "fused_kernel<0, 13, short"


def mean_counter(root, name):
    vals = []
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if KERNEL in row.get("Kernel_Name", "") and row.get("Counter_Name") == name:
                    vals.append(float(row["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {name} rows for {KERNEL} under {root}")
    return sum(vals) / len(vals), len(vals)


def main():
    prof = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "hbm_traffic.json")
    fetch_kb, nf = mean_counter(os.path.join(prof, "fetch"), "FETCH_SIZE")
    write_kb, nw = mean_counter(os.path.join(prof, "write"), "WRITE_SIZE")
    rec = {
        "kernel": "mfcc::fused_kernel<0, 13, short, mfcc::FrameGeom<400, 160>>",
        "kernel_revision": kernel_revision(),
        "workload": f"{N_UTT} x {UTT_LEN} int16 samples -> [{N_UTT * FRAMES_PER_UTT}, 39] float32 (bench.py default)",
        "FETCH_SIZE_KB_per_launch": round(fetch_kb, 1), "launches_fetch": nf,
        "WRITE_SIZE_KB_per_launch": round(write_kb, 1), "launches_write": nw,
        "correction": "gfx950: FETCH_SIZE doubled (128-byte requests are tallied at 64 B for 16 B/lane streaming reads; an upper "
                      "bound for the 2-byte predecessor loads of the staging, which are a few percent of the requests); WRITE_SIZE as reported",
        "hbm_bytes_per_launch": int(round(2 * fetch_kb * 1024 + write_kb * 1024)),
        "read_bytes_per_launch": int(round(2 * fetch_kb * 1024)), "write_bytes_per_launch": int(round(write_kb * 1024)),
        "algorithmic_bytes_per_launch": 2 * N_UTT * UTT_LEN + 156 * N_UTT * FRAMES_PER_UTT,
        "passes": "rocprofv3 --pmc FETCH_SIZE and rocprofv3 --pmc WRITE_SIZE, separate runs of "
                  "`python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline` (tools/profile_bench.sh)",
    }
    with open(out, "w") as fh:
        json.dump(rec, fh, indent=1)
        fh.write("\n")
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
