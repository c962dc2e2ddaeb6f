"""Sum rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two separate passes) over the LAST decode step of a bench.py run.

usage: pmc_step_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [note] [mean_ctx]
A step = the dispatches after the previous mtp_finalize_kernel (once per step, between the code predictor and the backbone) up to
and including the last one: the backbone half of one step + the code-predictor half of the next = one step's worth of launches.
FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 wide coalesced reads are reported at 1/2); units are KB."""
import collections, csv, json, re, sys

def last_step(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    adv = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("mtp_finalize_kernel")]
    assert len(adv) >= 2, "need at least two decode steps in the trace"
    step = rows[adv[-2] + 1: adv[-1] + 1]
    by = collections.defaultdict(float)
    for r in step:
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        by[name] += float(r["Counter_Value"])
    tot = sum(by.values())
    return {"kernels": len(step), "raw_kb": tot,
            "by_kernel_mb_raw": {k: round(v / 1024, 2) for k, v in sorted(by.items(), key=lambda kv: -kv[1])}}

if __name__ == "__main__":
    f, w, out = sys.argv[1:4]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    mean_ctx = float(sys.argv[5]) if len(sys.argv) > 5 else None
    fe, wr = last_step(f, "FETCH_SIZE"), last_step(w, "WRITE_SIZE")
    rd = fe["raw_kb"] * 1024 * 2
    wb = wr["raw_kb"] * 1024
    json.dump({"fetch": fe, "write": wr, "hbm_read_bytes_corrected": rd, "hbm_write_bytes": wb,
               "traffic_bytes_per_step": rd + wb, "mean_ctx": mean_ctx, "note": note}, open(out, "w"), indent=1)
    print(f"kernels/step {fe['kernels']}  read {rd/1e9:.3f} GB (x2 corrected)  write {wb/1e9:.3f} GB")
