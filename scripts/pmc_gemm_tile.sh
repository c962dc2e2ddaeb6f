#!/bin/bash
# usage: scripts/pmc_gemm_tile.sh <tag> -- matrix-pipe / wait counters of omni_gemm_tile and of hipBLASLt's kernels on the four W3 prefill
# shapes (scripts/bench_gemm_tile.py), one PMC pass: where the 12-16 % between the two go
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
raw=/tmp/pmcg_$tag; rm -rf $raw; mkdir -p $raw
export PREFILL_ONLY=1 M=6438
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $raw/p1 -o p -- python3 scripts/bench_gemm_tile.py > $raw/p1.log 2>&1
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("/tmp/pmcg_$tag/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:100] + "  grid=" + r.get("Grid_Size", "?")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            cnt[k] += 1
with open("gpurun_out/pmc_${tag}_gemm_tile.txt", "w") as out:
    for k, d in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0))[:16]:
        n = max(cnt[k], 1)
        out.write(k + "  launches=%d\n" % n)
        for c, v in sorted(d.items()):
            out.write("    %-30s %16.1f per launch\n" % (c, v / n))
        if d.get("SQ_BUSY_CYCLES"):
            out.write("    matrix pipe busy / SQ busy      %.3f\n" % (d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / d["SQ_BUSY_CYCLES"]))
        if d.get("SQ_WAVE_CYCLES"):
            out.write("    wait_any / wave cycles          %.3f   wait_inst_any / wave cycles %.3f\n" % (d.get("SQ_WAIT_ANY", 0) / d["SQ_WAVE_CYCLES"], d.get("SQ_WAIT_INST_ANY", 0) / d["SQ_WAVE_CYCLES"]))
print(open("gpurun_out/pmc_${tag}_gemm_tile.txt").read()[:9000])
PY
tail -8 $raw/p1.log
