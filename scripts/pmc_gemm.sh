#!/bin/bash
# usage: scripts/pmc_gemm.sh <tag>  -- PMC counters of the decode step's kernels (separate passes), summarised per kernel name
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
raw=/tmp/pmc_$tag; rm -rf $raw; mkdir -p $raw            # raw counter CSVs stay on the box: gpurun_out/ is capped at 64 MiB
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_TAG_STALL_sum TCC_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $raw/p$i -o p -- python3 bench.py --steps 3 --warmup 1 --ttfa-steps 3 --target-ctx 0 --no-cpu-baseline --device-weights > $raw/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("/tmp/pmc_$tag/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:90]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVES", "TCC_HIT_sum", "TA_BUSY_avr", "TCC_BUSY_avr", "TCC_TAG_STALL_sum"):
            cnt[(k, r["Counter_Name"])] += 1
with open("gpurun_out/pmc_${tag}_summary.txt", "w") as out:
    for k, d in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:24]:
        n = max(cnt[(k, "SQ_WAVES")], 1)
        out.write(k + "  launches=%d\n" % n)
        for c, v in sorted(d.items()):
            out.write("    %-34s %14.1f per launch\n" % (c, v / max(cnt.get((k, c), n), n)))
print(open("gpurun_out/pmc_${tag}_summary.txt").read()[:6000])
PY
