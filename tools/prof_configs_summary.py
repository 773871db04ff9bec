"""Summary of a tools/prof_configs.sh output directory: kernel stats + per-launch counters of the trace kernel per config.
   python3 tools/prof_configs_summary.py gpurun_out/prof_configs_<tag>"""
import collections
import csv
import glob
import os
import re
import sys

out = sys.argv[1]
for cfg in ("c2", "c4", "c5"):
    log = os.path.join(out, cfg + "_kt.log")
    line = [l for l in open(log).read().splitlines() if "per render" in l] if os.path.exists(log) else []
    bl = os.path.join(out, "line_%s.json" % cfg)
    if not line and os.path.exists(bl):
        import json
        try:
            d = json.loads(open(bl).read()); line = ["%s: %.1f Mrays/s, %.3f ms per step (bench.py, un-profiled)" % (d["config"]["workload"], d["value"], d["ms_per_step"])]
        except Exception:
            pass
    print("== %s: %s" % (cfg, line[-1] if line else "(no timing line)"))
    for f in glob.glob(os.path.join(out, cfg + "_kt", "**", "*kernel_stats.csv"), recursive=True) + glob.glob(os.path.join(out, cfg, "kt", "**", "*kernel_stats.csv"), recursive=True):
        for i, row in enumerate(csv.reader(open(f))):
            if i == 0 or "pt_" in row[0] or "k_reduce" in row[0]:
                print("   " + ", ".join(c[:60] for c in row[:7]))
    per = {}
    for p in ("sq", "ta", "fetch", "write", "l2"):
        for f in glob.glob(os.path.join(out, "%s_%s" % (cfg, p), "**", "*counter_collection.csv"), recursive=True) + glob.glob(os.path.join(out, cfg, p, "**", "*counter_collection.csv"), recursive=True):
            agg, cnt = collections.defaultdict(float), collections.Counter()
            for r in csv.DictReader(open(f)):
                n = r.get("Kernel_Name", "")
                if not re.search(r"pt_(queue|packet|mega)kernel(_lean)?<false", n):
                    continue
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
            for k in agg:
                per[k] = agg[k] / cnt[k]
    for k in sorted(per):
        print("   %-30s per launch %.6g" % (k, per[k]))
    if per.get("GRBM_GUI_ACTIVE") and per.get("SQ_ACTIVE_INST_VALU"):
        cyc = per["GRBM_GUI_ACTIVE"] / 8.0
        print("   derived: VALU active %.3f of SIMD-cycles, lanes %.3f, waves waiting %.3f, L1 address unit busy %.3f, look-ups per CU-clock %.3f, TCC hit rate %s, fabric GB %s" % (
            per["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc), per["SQ_THREAD_CYCLES_VALU"] / 64 / per["SQ_ACTIVE_INST_VALU"], per["SQ_WAIT_ANY"] / per["SQ_WAVE_CYCLES"],
            per["TA_TA_BUSY_sum"] / 256 / cyc, per["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256 / cyc,
            "%.3f" % (per["TCC_HIT_sum"] / (per["TCC_HIT_sum"] + per["TCC_MISS_sum"])) if "TCC_HIT_sum" in per else "n/a",
            "%.1f" % ((per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024 / 1e9) if "FETCH_SIZE" in per and "WRITE_SIZE" in per else "n/a"))
