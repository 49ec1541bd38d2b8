#!/bin/bash
# In-step matrix-pipe utilisation of the sparse-conv kernels: two --pmc passes over the bench command (no other trace
# domains), summarised per kernel.  GPU box.   usage: scripts/pmc_mfma_instep.sh <tag>
tag=${1:-mfma}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
for ctr in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES; do
  rm -rf /tmp/$tag.$ctr && mkdir -p /tmp/$tag.$ctr
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/$tag.$ctr -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full-graph --profile-steps 0 > /dev/null 2> $out/$ctr.stderr
  python scripts/pmc_summary.py /tmp/$tag.$ctr/r_counter_collection.csv $ctr > $out/$ctr.csv
done
python - <<PY
import csv
d = {}
for ctr in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_WAVES"):
    for r in csv.DictReader(open("$out/%s.csv" % ctr)):
        d.setdefault(r["kernel"], {})[ctr] = float(r["mean_" + ctr])
        d[r["kernel"]]["launches"] = int(r["launches"])
print("kernel, launches, waves/launch, MFMA instr/launch, matrix pipe busy (MFMA busy SIMD-cycles / (SQ busy cycles / 32 SE x 1024 SIMD))")
for k, v in sorted(d.items()):
    if "conv" in k and all(c in v for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES")):
        simd_cycles = v["SQ_BUSY_CYCLES"] / 32.0 * 1024.0
        print("%s, %d, %.0f, %.3g, %.3f" % (k, v["launches"], v.get("SQ_WAVES", 0), v.get("SQ_INSTS_MFMA", 0), v["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles))
PY
