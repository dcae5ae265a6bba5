"""Turn two rocprofv3 PMC passes (--pmc FETCH_SIZE / --pmc WRITE_SIZE, csv output) over
`python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline` into profiles/rNN_pmc_traffic.json.

usage: python tools/make_pmc_traffic.py FETCH_DIR WRITE_DIR OUT.json [model_points scene_points df mode tau_d]

Counters are KiB per dispatch.  FETCH_SIZE is doubled (MI355X_MICROARCH.md: on gfx950 a wide
coalesced streaming read is tallied at half its bytes); WRITE_SIZE is used as read."""
import csv, glob, hashlib, json, os, sys


def per_kernel(d, counter):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            s = acc.setdefault(k, {})
            s[r["Dispatch_Id"]] = s.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    return {k: (len(v), sum(v.values()) / len(v)) for k, v in acc.items()}


def main():
    fd, wd, out = sys.argv[1:4]
    cfg = sys.argv[4:9] if len(sys.argv) >= 9 else ["5000", "100000", "8", "exact", "0.025"]
    f, w = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
    run = {"model_points": int(cfg[0]), "scene_points": int(cfg[1]), "ref_point_df": int(cfg[2]),
           "vote_mode": cfg[3], "tau_d": float(cfg[4])}
    for k in sorted(f):
        if not k.startswith("k_"):
            continue
        run[k] = {"launches": f[k][0], "FETCH_SIZE_KiB_avg": f[k][1], "WRITE_SIZE_KiB_avg": w.get(k, (0, 0.0))[1]}
    v = run["k_vote"]
    run["hbm_bytes_per_vote_launch"] = 1024.0 * (2.0 * v["FETCH_SIZE_KiB_avg"] + v["WRITE_SIZE_KiB_avg"])
    import importlib
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sha = importlib.import_module("objective-slam_amd").ppf.kernel_source_hash()
    rec = {"note": __doc__.strip().split("\n\n")[-1].replace("\n", " "),
           # bench.py reports the traffic figure only for the kernel source it was measured with
           "kernel_source_sha16": sha, "runs": [run]}
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(run, indent=1))


if __name__ == "__main__":
    main()
