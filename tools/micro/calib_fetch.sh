# Calibration of FETCH_SIZE for THIS kernel's access shape (per-lane 16-byte loads of 128-byte records): the gather micro-benchmark
# on a 151 MB table (38 x the L2 of an XCD: ~97 % of the record fetches miss L2) moves a known number of 128-byte lines.
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=gpurun_out/calib_fetch
mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/$OUT/f -- $ROOT/tools/micro/gather_coop 151 12 0 > $ROOT/$OUT/f.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_HIT_sum --output-format csv -d $ROOT/$OUT/r -- $ROOT/tools/micro/gather_coop 151 12 0 > $ROOT/$OUT/r.log 2>&1)
cat $OUT/f.log | grep "mode A"
python3 - <<PY
import csv, glob, collections
per = {}
for d in ("f", "r"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(float); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            if "k_gather<0>" not in r.get("Kernel_Name", ""): continue
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
        for k in agg: per[k] = agg[k] / cnt[k]
recs = 256 * 12 // 4 * 256 * 2000          # blocks x threads x iterations: one 128-byte record each
print("records per launch %d = %.3f GB of 128-byte lines" % (recs, recs * 128 / 1e9))
for k, v in sorted(per.items()): print("   %-24s %.6g" % (k, v))
miss = per["TCC_MISS_sum"] / (per["TCC_MISS_sum"] + per["TCC_HIT_sum"])
print("L2 miss share of requests %.3f; FETCH_SIZE x 1024 B = %.3f GB -> %.3f of the lines' bytes; RDREQ x 64 B = %.3f GB" % (
    miss, per["FETCH_SIZE"] * 1024 / 1e9, per["FETCH_SIZE"] * 1024 / (recs * 128.0), per["TCC_EA0_RDREQ_sum"] * 64 / 1e9))
PY
