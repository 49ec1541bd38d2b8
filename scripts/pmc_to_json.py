"""profiles/pmc_latest.json from the FETCH_SIZE / WRITE_SIZE summaries of scripts/round_profile.sh.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are in KiB and on
gfx950 FETCH_SIZE reports half of the bytes of 16-byte-per-lane reads (MI355X_MICROARCH.md §HBM), which
is what all of these kernels issue; WRITE_SIZE is used as reported (uncalibrated per the same guide)."""
import csv
import re
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
vals = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    with open("%s/%s.summary.csv" % (src, ctr)) as f:
        for row in csv.DictReader(f):
            name = row["kernel"].replace("efg::", "").replace(";", ",")
            # bench.py labels: conv_fwd_kernel<NT> -- the KV / KS template arguments are launch details; variants of
            # one NT are merged with a launch-weighted mean
            m = re.match(r"conv_fwd_kernel<(\d+), \d+(, \d+)?>$", name)
            if m:
                name = "conv_fwd_kernel<%s>" % m.group(1)
            m = re.match(r"conv_tile_kernel<(\d+), \d+, \d+(, \d+){0,2}>$", name)  # <NT, R, KS[, V4[, MODE]]>: bench.py labels by NT
            if m:
                name = "conv_tile_kernel<%s>" % m.group(1)
            m = re.match(r"box_bwd_tile_kernel<\d+>$", name)   # the tile height is a launch detail
            if m:
                name = "box_bwd_tile_kernel"
            v = vals.setdefault(name, {})
            n, mean = int(row["launches"]), float(row["mean_" + ctr])
            tot = v.get("launches_" + ctr, 0)
            v[ctr] = (v.get(ctr, 0.0) * tot + mean * n) / (tot + n)
            v["launches_" + ctr] = tot + n
out = {"unit": "bytes per launch", "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024",
       "source": "%s/{FETCH_SIZE,WRITE_SIZE}.summary.csv: rocprofv3 --kernel-trace --pmc <ctr> -- python bench.py --steps 2 "
                 "--warmup 1 --no-cpu-baseline --no-full-graph --no-arm (scripts/round_profile.sh)" % src.rstrip("/").split("/")[-1],
       "kernels": {}}
for name, v in vals.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        out["kernels"][name] = {"hbm_bytes_per_launch": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024,
                                "FETCH_SIZE_KiB": v["FETCH_SIZE"], "WRITE_SIZE_KiB": v["WRITE_SIZE"]}
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print(len(out["kernels"]), "kernels ->", dst)
