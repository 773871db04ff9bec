"""profiles/traffic.json from the separate rocprofv3 --pmc passes of tools/prof_bench.sh.

  python3 tools/make_traffic_json.py gpurun_out/prof_bench_<tag> [profiles/traffic.json]

Per-launch counters of the trace kernel (pt_packetkernel<false,...> or pt_queuekernel<false,...>, whichever ran): FETCH_SIZE and WRITE_SIZE (KB; FETCH_SIZE =
TCC_EA0_RDREQ x 64 B -- the guide's x2 correction is for wide coalesced streams and is NOT applied to these 16-byte
gathers), TCC_HIT / TCC_MISS.  The file is stamped with bench.source_hash(): bench.py ignores it for any other
device code."""
import collections
import csv
import glob
import json
import re
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench   # noqa: E402

src = sys.argv[1]
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(REPO, "profiles", "traffic.json")
bench_args = sys.argv[3].split() if len(sys.argv) > 3 else []      # the bench.py arguments the passes were made with (tools/prof_configs.sh): stamps the workload
agg, cnt = collections.defaultdict(float), collections.Counter()
kernels = set()
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if not re.search(r"pt_(queue|packet)kernel(_lean)?<false, true(, false(, (false|true)(, (false|true))?)?)?>", name):      # the exact-mode trace kernel only
            continue
        kernels.add(re.search(r"pt_\w+kernel\w*<[^>]*>", name).group(0).replace(" ", ""))
        k = r.get("Counter_Name")
        agg[k] += float(r.get("Counter_Value", 0)); cnt[k] += 1
per = {k: agg[k] / cnt[k] for k in agg}
need = ("FETCH_SIZE", "WRITE_SIZE")
if any(k not in per for k in need):
    print("missing counters (fabric fields will be null): have %s" % sorted(per))
    for k in need:
        per.setdefault(k, float("nan"))
out = {
    "source": "%s (separate --pmc passes of `python3 bench.py --steps 1 --warmup 0`)" % src,
    "source_hash": bench.source_hash(REPO),
    "workload": bench.workload_text(bench.parse_args(bench_args)),
    "kernel": ", ".join(sorted(kernels)),
    "FETCH_SIZE_KB_per_launch": per["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": per["WRITE_SIZE"],
    "traffic_GB_per_launch": round((per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024 / 1e9, 1) if per["FETCH_SIZE"] == per["FETCH_SIZE"] and per["WRITE_SIZE"] == per["WRITE_SIZE"] else None,
    "TCC_HIT_per_launch": per.get("TCC_HIT_sum"), "TCC_MISS_per_launch": per.get("TCC_MISS_sum"),
    "tcc_hit_rate": round(per["TCC_HIT_sum"] / (per["TCC_HIT_sum"] + per["TCC_MISS_sum"]), 4) if "TCC_HIT_sum" in per else None,
    "SQ": {k: per[k] for k in sorted(per) if k.startswith("SQ_")},
    # vector-memory front end: the CU's address unit (TA) and L1 tag look-ups; GRBM_GUI_ACTIVE is the sum of the 8 XCDs' busy cycles
    "TA_TA_BUSY_sum": per.get("TA_TA_BUSY_sum"), "TCP_TOTAL_CACHE_ACCESSES_sum": per.get("TCP_TOTAL_CACHE_ACCESSES_sum"),
    "GRBM_GUI_ACTIVE": per.get("GRBM_GUI_ACTIVE"),
    # SQ_ACTIVE_INST_VALU counts quad-cycles summed over waves; a SIMD has one vector pipe: share of SIMD-cycles with a VALU instruction in it
    "valu_active_frac": round(per["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * per["GRBM_GUI_ACTIVE"] / 8.0), 4) if per.get("SQ_ACTIVE_INST_VALU") and per.get("GRBM_GUI_ACTIVE") else None,
    "ta_busy_frac": round(per["TA_TA_BUSY_sum"] / 256.0 / (per["GRBM_GUI_ACTIVE"] / 8.0), 4) if per.get("TA_TA_BUSY_sum") and per.get("GRBM_GUI_ACTIVE") else None,
    "l1_lookups_per_cu_clock": round(per["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256.0 / (per["GRBM_GUI_ACTIVE"] / 8.0), 4) if per.get("TCP_TOTAL_CACHE_ACCESSES_sum") and per.get("GRBM_GUI_ACTIVE") else None,
    "note": "raw counters x 1024 B; uncalibrated for 16 B/lane gathers (guide: HBM section), compare between builds of this kernel",
}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
