#!/usr/bin/env python3
"""profiles/<tag>_pmc_summary.csv -> profiles/pmc_traffic.json (what bench.py reads for `roofline.traffic` and the issue roofline):
per stage, HBM bytes per 64-frame launch = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 (MI355X_MICROARCH.md: FETCH_SIZE counts half of a
coalesced stream on gfx950) and wave-level VALU instructions per launch (SQ_INSTS_VALU); pyramid = sum of the seven k_resize4 launches.
The file is stamped with the digest of csrc/ (dvslam_amd._lib.kernel_source_digest) the counters were collected on — run this right after
tools/collect_profiles.sh, from the same tree — and bench.py quotes the numbers only while the digest still matches.
usage: tools/make_pmc_traffic.py profiles/r01_f_pmc_summary.csv"""
import csv, json, os, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dynamic-visual-slam_amd"))
from dvslam_amd import _lib
src = sys.argv[1]
rows = [r for r in csv.reader(l for l in open(src) if not l.startswith("#"))]
hdr = rows[0]
fi, wi, vi = hdr.index("FETCH_SIZE"), hdr.index("WRITE_SIZE"), hdr.index("SQ_INSTS_VALU")
stage = {"k_resize4": "pyramid", "k_fast_wave": "fast", "k_octree": "octree", "k_blur_stream": "blur", "k_describe": "describe", "k_match": "match"}
by, vl = {}, {}
for r in rows[1:]:
    for k, v in stage.items():
        if k in r[0]:
            by[v] = by.get(v, 0) + int((2 * float(r[fi]) + float(r[wi])) * 1024)
            vl[v] = vl.get(v, 0) + int(float(r[vi]))
try:
    commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:  # noqa: BLE001
    commit = None
json.dump({"csrc_digest": _lib.kernel_source_digest(), "commit": commit,
           "note": f"from {src}: HBM bytes per launch of 64 frames = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 (gfx950: FETCH_SIZE reads half of a "
                   "coalesced stream); valu_insts = SQ_INSTS_VALU (wave-level) per launch; pyramid = sum of the seven k_resize4 launches",
           "batch": 64, "bytes_per_launch": by, "valu_insts_per_launch": vl,
           "valu_issue_peak_G_per_s": 560.0,
           "valu_issue_peak_note": "measured: profiles/r02_valu_issue_rates.txt (packed-16 / perm / dot4 / bcnt / min3 / mad24 at 8 waves per SIMD)"},
          open("profiles/pmc_traffic.json", "w"), indent=1)
print(by, vl)
