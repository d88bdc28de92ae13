"""profiles/<name>_pmc.json from rocprofv3 --pmc output directories (one per counter group): mean per launch and kernel,
hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (MI355X_MICROARCH.md, HBM section: gfx950 FETCH_SIZE counts 128-B requests as 64 B).
usage: make_pmc_json.py out.json dir1 [dir2 ...]"""
import collections, csv, glob, json, re, sys
out, roots = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for root in roots:
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "smr::" not in k: continue
            k = re.sub(r"<.*", "", k.split("(")[0].replace("void ", "")).strip()         # template arguments dropped
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
kernels = {}
for k in sorted(acc):
    d = {c: acc[k][c] / n[k][c] for c in sorted(acc[k])}
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d: d["hbm_bytes"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
    d["launches_sampled"] = max(n[k].values())
    kernels[k] = d
json.dump({"command": "rocprofv3 --kernel-trace --output-format csv --pmc <counters> -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline "
                      "(one pass per counter group: FETCH_SIZE | WRITE_SIZE | SQ_*), 65536 envs, space scene, N=3",
           "units": "mean per launch; FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 "
                    "(gfx950: FETCH_SIZE counts 128-B requests at 64 B, MI355X_MICROARCH.md \"HBM\"); SQ_*_CYCLES in quad-cycles",
           "kernels": kernels}, open(out, "w"), indent=1)
print(out, {k: v.get("hbm_bytes") for k, v in kernels.items()})
