#!/usr/bin/env python3
"""Table of tools/exp/r06_pmc_footprint.sh: per number of rotating sets K, the median per-dispatch value of every counter and the kernel's
duration in the same pass (the first K + 8 launches of a pass warm caches and translations and are left out).
    python tools/exp/r06_pmc_footprint_summary.py gpurun_out/r06_pmc_footprint > summary.json"""
import csv, glob, json, os, re, statistics, sys

root = sys.argv[1]
table = {}
for d in sorted(glob.glob(os.path.join(root, "k*_p*"))):
    if not os.path.isdir(d):
        continue
    m = re.match(r"k(\d+)_p(\d+)", os.path.basename(d))
    K = int(m.group(1))
    row = table.setdefault(str(K), {"kernel_us_under_pmc": {}})
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    skip = K + 8
    if cc:
        vals = {}
        for r in csv.DictReader(open(cc[0])):
            if "lbs_skin" not in r["Kernel_Name"]:
                continue
            vals.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for name, v in vals.items():
            v = [x for _, x in sorted(v)][skip:]
            if v:
                row[name] = statistics.median(v)
    if kt:
        du = []
        for r in csv.DictReader(open(kt[0])):
            if "lbs_skin" in r["Kernel_Name"]:
                du.append((int(r["Dispatch_Id"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3))
        du = [x for _, x in sorted(du)][skip:]
        if du:
            row["kernel_us_under_pmc"]["p" + m.group(2)] = round(statistics.median(du), 2)
for K, row in table.items():
    def ratio(a, b):
        return None if a not in row or b not in row or not row[b] else row[a] / row[b]
    row["derived"] = {
        "utcl1_miss_rate": ratio("TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_REQUEST_sum"),
        "tcp_read_latency_cycles": ratio("TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum"),
        "tcp_write_latency_cycles": ratio("TCP_TCC_WRITE_REQ_LATENCY_sum", "TCP_TCC_WRITE_REQ_sum"),
        "ea_read_latency_cycles": ratio("TCC_EA0_RDREQ_LEVEL_sum", "TCC_EA0_RDREQ_sum"),
        "ea_write_latency_cycles": ratio("TCC_EA0_WRREQ_LEVEL_sum", "TCC_EA0_WRREQ_sum"),
        "utcl2_busy_of_gui_active": ratio("GRBM_UTCL2_BUSY", "GRBM_GUI_ACTIVE"),
        "tcc_hit_rate": ratio("TCC_HIT_sum", "TCC_REQ_sum"),
    }
print(json.dumps({"what": "lone lbs_skin_dyn launch (C4: 1 M vertices / 256 bones) rotating over K sets of 100 MB; median per dispatch", "by_sets": table}, indent=1))
