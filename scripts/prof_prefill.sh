#!/bin/bash
# usage: scripts/prof_prefill.sh <tag> -- kernel stats + MFMA counters of three 6.4k-token W3 prefills
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf_$tag -o pf -- python3 scripts/prof_prefill.py > gpurun_out/pf_$tag.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/pf_$tag/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"][:110], r["Calls"], r["AverageNs"][:9], r["Percentage"])
PY
grep "prefill ms" gpurun_out/pf_$tag.log
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pfpmc_$tag -o pf -- python3 scripts/prof_prefill.py > gpurun_out/pfpmc_$tag.log 2>&1
python3 - <<PY
import csv,glob,collections
fs=glob.glob("gpurun_out/pfpmc_$tag/**/*counter_collection.csv",recursive=True)
if fs:
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if "prefill_mfma" in r["Kernel_Name"]:
            agg[r["Counter_Name"]]["sum"]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
    for k,v in agg.items(): print(k, "per launch", v["sum"]/n[k], "launches", n[k])
else:
    print("no counter csv"); print(open("gpurun_out/pfpmc_$tag.log").read()[-1500:])
PY
