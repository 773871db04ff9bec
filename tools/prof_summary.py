"""Text summary of one tools/prof_bench.sh output directory (kernel-trace stats + per-launch PMC values).

  python3 tools/prof_summary.py gpurun_out/prof_bench_<tag> > profiles/<round>_bench_coffee256_rocprofv3_summary.txt"""
import collections
import csv
import glob
import os
import re
import sys

src = sys.argv[1]
print("# %s -- rocprofv3 passes of `python3 bench.py` (tools/prof_bench.sh): kernel trace with --stats, then one --pmc pass per counter group" % src)
for f in sorted(glob.glob(os.path.join(src, "*.log"))):
    for line in open(f):
        if '"metric"' in line:
            m = re.search(r'"value": ([0-9.]+).*?"ms_per_step": ([0-9.]+)', line)
            r = re.search(r'"launch_ms": ([0-9.]+), "launches_timed": ([0-9]+)', line)
            print("pass %-6s bench.py line: %s Mrays/s, %s ms/step, trace kernel %s ms per launch over %s launches (HIP events)" % (
                os.path.basename(f)[:-4], m.group(1), m.group(2), r.group(1), r.group(2)))
for f in glob.glob(os.path.join(src, "kt", "**", "*kernel_stats.csv"), recursive=True):
    print("\n== kernel trace (--stats): %s" % f)
    for i, row in enumerate(csv.reader(open(f))):
        if i == 0 or "pt::" in row[0]:
            print("  " + ", ".join(c[:70] for c in row))
print("\n== counters, per launch (kernel name shortened)")
for d in sorted(glob.glob(os.path.join(src, "*/"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg, cnt = collections.defaultdict(float), collections.Counter()
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if "queuekernel" not in name and "packetkernel" not in name and "reduce" not in name:
                continue
            short = ("pt_%skernel<%s>" % ("packet" if "packetkernel" in name else "queue", name.split("<")[1].split(">")[0])
                     if "kernel<" in name else "k_reduce_samples")
            k = (short, r.get("Counter_Name"))
            agg[k] += float(r.get("Counter_Value", 0)); cnt[k] += 1
        for k in sorted(agg):
            print("  %-36s %-24s per_launch=%.6g (launches=%d)" % (k[0], k[1], agg[k] / cnt[k], cnt[k]))
