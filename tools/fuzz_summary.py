"""Summary of tools/gpu_fuzz.py logs: python tools/fuzz_summary.py gpurun_out/fuzz/*.txt > profiles/rNN_fuzz.txt"""
import re, sys, collections
by = collections.Counter(); verdicts = collections.Counter(); shown = []
opts_seen = collections.defaultdict(collections.Counter)
for path in sys.argv[1:]:
    for line in open(path):
        if not line.startswith("case"):
            continue
        m = re.match(r"case\s+\d+\s+(\S+)\s+.*rule (\d) (\{.*\}) ran \((\d), (\d+)\) -> (.*)$", line.strip())
        if not m:
            continue
        scene, rule, opts, var, fmt, verdict = m.groups()
        by[(scene, "variant %s" % var, "%s-byte nodes" % fmt if var == "4" else "128-byte nodes", "rule %s" % rule)] += 1
        v = verdict.split(" (")[0]
        verdicts[v] += 1
        for k, val in eval(opts).items():
            opts_seen[k][val] += 1
        if v != "ok":
            shown.append(path.split("/")[-1] + ": " + line.strip())
print("# tools/gpu_fuzz.py: every case renders a random scene / size / sample count / rank's share with the per-lane kernel (variant 0,")
print("# leaves of 4, Karras tree) and again under random scheduler, tree and node-format options, twice (the second time with the tile")
print("# history); the accumulators must be bit-identical.  A mismatch is replayed with the per-lane kernel on the CANDIDATE's own tree: equal there = tree-dependent")
print("# grazing hit (DESIGN.md section 2), else a scheduler MISMATCH.")
print("cases %d: %s" % (sum(verdicts.values()), ", ".join("%s %d" % kv for kv in sorted(verdicts.items()))))
print("\nby scene / kernel that ran / node format / shadow rule:")
for k in sorted(by):
    print("  %-20s %-10s %-15s %-7s %5d" % (k + (by[k],)))
print("\noption values drawn:")
for k in sorted(opts_seen):
    print("  %-18s %s" % (k, "  ".join("%s:%d" % kv for kv in sorted(opts_seen[k].items()))))
if shown:
    print("\ncases that were not bit-identical:")
    for s in shown:
        print("  " + s)
