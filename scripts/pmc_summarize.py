"""Aggregates the rocprofv3 counter CSVs of scripts/pmc_bench.sh into per-kernel, per-launch figures."""
import collections, csv, glob, json, re, sys

out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = agg[r["Kernel_Name"]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for f in glob.glob(out + "/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d = dur[r["Kernel_Name"]]
        d[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); d[1] += 1


def short(n):
    m = re.search(r"gemm_nt_kernelILi(\d+)ELi(\d+)ELi\d+ELi\d+ELi\d+ENS_\d+(\w+?)ELi", n)
    if m:
        return f"gemm_nt<{m.group(1)}x{m.group(2)},{m.group(3)}>"
    m = re.search(r"gemm_nt8(p?)_kernelINS_\d+(\w+?)ELi", n)
    if m:
        return f"gemm_nt8{m.group(1)}<{m.group(2)}>"
    m = re.search(r"wmi\d*(\d\d)?([a-z_0-9]+_kernel)", n)
    m2 = re.search(r"wmi::([a-z_0-9]+)", n)
    return m2.group(1) if m2 else (m.group(2) if m else n[:40])


rows = []
for k, c in agg.items():
    n = max(v[1] for v in c.values())
    g = lambda name: (c[name][0] / c[name][1]) if name in c and c[name][1] else None
    fetch_kb, write_kb = g("FETCH_SIZE"), g("WRITE_SIZE")
    mfma, busy, gui = g("SQ_VALU_MFMA_BUSY_CYCLES"), g("SQ_BUSY_CYCLES"), g("GRBM_GUI_ACTIVE")
    us = dur[k][0] / dur[k][1] / 1e3 if dur[k][1] else None
    row = {"kernel": short(k), "launches": n, "avg_us_under_counters": us,
           # gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes -> doubled for wide coalesced reads
           "fetch_MB": None if fetch_kb is None else fetch_kb * 2 * 1024 / 1e6,
           "write_MB": None if write_kb is None else write_kb * 1024 / 1e6,
           # MFMA busy is summed over SIMDs (4 per CU, 256 CUs); GRBM_GUI_ACTIVE over the 8 XCDs
           "mfma_util": None if not (mfma and gui) else mfma / (gui / 8 * 1024)}
    if row["fetch_MB"] is not None and us:
        row["fabric_read_TBps"] = row["fetch_MB"] / us
        if row["write_MB"] is not None:
            row["fabric_rw_TBps"] = (row["fetch_MB"] + row["write_MB"]) / us
    rows.append(row)
rows.sort(key=lambda r: -(r["avg_us_under_counters"] or 0) * r["launches"])
json.dump(rows, open(out + "/summary.json", "w"), indent=1)
with open(out + "/summary.md", "w") as f:
    f.write("| kernel | launches | avg us | fetch MB/launch (x2 corrected) | write MB/launch | read TB/s | MFMA busy / (cycles x 1024 SIMDs) |\n|---|---:|---:|---:|---:|---:|---:|\n")
    for r in rows[:24]:
        fm = lambda v, p=1: "-" if v is None else f"{v:.{p}f}"
        f.write(f"| `{r['kernel']}` | {r['launches']} | {fm(r['avg_us_under_counters'])} | {fm(r['fetch_MB'], 2)} | {fm(r['write_MB'], 2)} | {fm(r.get('fabric_read_TBps'), 2)} | {fm(r['mfma_util'], 3)} |\n")
print(open(out + "/summary.md").read())
