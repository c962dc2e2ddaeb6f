#!/bin/bash
# usage: scripts/pmc_attention.sh <tag>  -- SQ instruction / cycle counters of the decode step's kernels (one PMC pass), per kernel name
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
raw=/tmp/pmca_$tag; rm -rf $raw; mkdir -p $raw
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $raw/p1 -o p -- python3 bench.py --steps 3 --warmup 1 --ttfa-steps 3 --target-ctx 0 --ctx-extra 250 --no-cpu-baseline --no-diagnostics --device-weights > $raw/p1.log 2>&1
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("/tmp/pmca_$tag/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:90]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            cnt[k] += 1
with open("gpurun_out/pmc_${tag}_sq.txt", "w") as out:
    ours = {k: d for k, d in tot.items() if "at::native" not in k and "rocclr" not in k}
    for k, d in sorted(ours.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]:
        n = max(cnt[k], 1)
        out.write(k + "  launches=%d\n" % n)
        for c, v in sorted(d.items()):
            out.write("    %-26s %14.1f per launch" % (c, v / n))
            if c.startswith("SQ_INSTS") and d.get("SQ_WAVES"):
                out.write("   = %.0f per wave" % (v / d["SQ_WAVES"]))
            out.write("\n")
print(open("gpurun_out/pmc_${tag}_sq.txt").read()[:8000])
PY
tail -3 $raw/p1.log
