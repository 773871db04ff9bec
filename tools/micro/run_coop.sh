set -u
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out/r03c
for w in 12 32; do ./tools/micro/gather_coop 3.1 $w; done 2>&1 | tee gpurun_out/r03c/coop.txt
for mode in 0 3; do
 (cd /tmp && rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --output-format csv -d $ROOT/gpurun_out/r03c/pmc$mode -- $ROOT/tools/micro/gather_coop 3.1 12 $mode > $ROOT/gpurun_out/r03c/pmc$mode.log 2>&1)
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r03c/pmc[03]/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(float); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = (r.get("Kernel_Name", "")[:30], r.get("Counter_Name"))
            agg[k] += float(r.get("Counter_Value", 0)); cnt[k] += 1
        for k in sorted(agg): print("   %s %-30s %-36s per_launch=%.6g (n=%d)" % (d[-8:], k[0], k[1], agg[k] / cnt[k], cnt[k]))
PY
