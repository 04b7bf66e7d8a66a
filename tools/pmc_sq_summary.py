"""Summarise the counter_collection CSVs of tools/pmc_sq.sh: per kernel, total and per-wave values."""
import csv, glob, os, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
waves_pass = collections.defaultdict(dict)
for p in sorted(glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True)):
    ps = p[len(out):].strip("/").split("/")[0]
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0] if "(anonymous namespace)::" not in r["Kernel_Name"] else r["Kernel_Name"].split("(anonymous namespace)::")[1].split("(")[0]
        c = r["Counter_Name"]
        if c == "SQ_WAVES":
            waves_pass[k][ps] = waves_pass[k].get(ps, 0.0) + float(r["Counter_Value"])
            continue
        tot[k][c] += float(r["Counter_Value"])
        cnt[k][c] += 1
for k in tot:
    if "xr_" not in k:
        continue
    waves = max(waves_pass[k].values()) if waves_pass[k] else 0.0
    print(f"# {k}: {max(cnt[k].values())} launches, {waves:.0f} waves")
    for c in sorted(tot[k]):
        v = tot[k][c]
        print(f"{c:28s} total {v:16.0f}   per wave {v / waves if waves else 0:12.1f}")
    wc = tot[k].get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            if c in tot[k]:
                print(f"   {c} / SQ_WAVE_CYCLES = {tot[k][c] / wc:.3f}")
    if tot[k].get("SQ_LDS_IDX_ACTIVE"):
        print(f"   SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = {tot[k]['SQ_LDS_BANK_CONFLICT'] / tot[k]['SQ_LDS_IDX_ACTIVE']:.3f}")
