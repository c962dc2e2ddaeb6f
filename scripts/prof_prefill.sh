#!/bin/bash
# usage: scripts/prof_prefill.sh <tag> -- kernel stats + MFMA counters of three 6.4k-token W3 prefills (PREFILL_GEMM=tile|blas)
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pf_$tag /tmp/pfpmc_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_$tag -o pf -- python3 scripts/prof_prefill.py > gpurun_out/pf_$tag.log 2>&1
cp $(find /tmp/pf_$tag -name '*kernel_stats.csv' | head -1) gpurun_out/${tag}_prefill_kernel_stats.csv
python3 - <<PY
import csv
for r in list(csv.DictReader(open("gpurun_out/${tag}_prefill_kernel_stats.csv")))[:14]:
    print(r["Name"][:110], r["Calls"], r["AverageNs"][:9], r["Percentage"])
PY
grep "prefill ms" gpurun_out/pf_$tag.log
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d /tmp/pfpmc_$tag -o pf -- python3 scripts/prof_prefill.py > gpurun_out/pfpmc_$tag.log 2>&1
python3 - <<PY > gpurun_out/${tag}_prefill_mfma_pmc.txt
import csv,glob,collections
fs=glob.glob("/tmp/pfpmc_$tag/**/*counter_collection.csv",recursive=True)
if fs:
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"]
        key="gemm_tile" if "gemm_tile" in k else ("prefill_mfma" if "prefill_mfma" in k else None)
        if key:
            agg[key][r["Counter_Name"]]+=float(r["Counter_Value"]); n[key][r["Counter_Name"]]+=1
    for key in agg:
        for c,v in agg[key].items(): print(key, c, "per launch", v/n[key][c], "launches", n[key][c])
else:
    print("no counter csv"); print(open("gpurun_out/pfpmc_$tag.log").read()[-1500:])
PY
cat gpurun_out/${tag}_prefill_mfma_pmc.txt
