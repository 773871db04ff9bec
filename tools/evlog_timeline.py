"""Timeline of an event log written by a -DPT_EVLOG device library (MOPTIX_EVLOG=file; packetkernel.hip PT_EV): what the waves of the
workgroup that holds a capped path do, bounce by bounce.  usage: evlog_timeline.py file [first_bounce n_bounces]"""
import sys, collections
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64)
n = int(raw[0]); ev = raw[1:1 + n]
t = (ev >> np.uint64(24)).astype(np.int64); val = ((ev >> np.uint64(8)) & np.uint64(0xffff)).astype(np.int64)
wave = ((ev >> np.uint64(4)) & np.uint64(3)).astype(np.int64); code = (ev & np.uint64(15)).astype(np.int64)
order = np.argsort(t, kind="stable"); t, val, wave, code = t[order], val[order], wave[order], code[order]
t0 = t[0]; us = (t - t0) / 100.0
names = {1: "visit begin (lanes)", 2: "visit loaded", 3: "visit ran", 4: "visit stored (aux lanes)", 5: "node run begin (active)", 6: "node run end (steps)",
         7: "leaf begin (lanes)", 8: "leaf end", 9: "txn begin (active)", 10: "txn end (pass+1)", 11: "idle", 12: "tagged pop"}
visits = np.where((code == 1) & (val > 0))[0]
print("%d events over %.1f us; %d visits with lanes -> %.2f us per visit" % (n, us[-1], len(visits), us[-1] / max(1, len(visits))))
# durations of the bracketed passes, per wave
dur = collections.defaultdict(list)
open_ = {}
pairs = {1: None, 2: 1, 3: 2, 4: 3, 6: 5, 8: 7, 10: 9}
label = {2: "visit: loads", 3: "visit: run", 4: "visit: stores", 6: "node run", 8: "leaf pass", 10: "transaction"}
for i in range(n):
    w, c = wave[i], code[i]
    if c in pairs and pairs[c] is not None and (w, pairs[c]) in open_:
        j = open_[(w, pairs[c])]
        if not (c == 2 and val[j] == 0):
            dur[label[c]].append(us[i] - us[j])
    open_[(w, c)] = i
for k, v in dur.items():
    v = np.array(v); print("  %-16s n %5d  mean %6.2f us  median %6.2f  sum %8.1f us (%.1f us per visit)" % (k, len(v), v.mean(), np.median(v), v.sum(), v.sum() / max(1, len(visits))))
steps = val[code == 6]; print("  node steps per run: mean %.2f, per visit %.1f" % (steps.mean(), steps.sum() / max(1, len(visits))))
b0 = int(sys.argv[2]) if len(sys.argv) > 2 else len(visits) // 2; nb = int(sys.argv[3]) if len(sys.argv) > 3 else 2
if len(visits) > b0 + nb:
    lo, hi = visits[b0], visits[b0 + nb]
    for i in range(lo, hi + 1):
        if code[i] in (9, 10) and "-v" not in sys.argv: continue
        print("%10.2f us  wave %d  %-26s %d" % (us[i] - us[lo], wave[i], names.get(int(code[i]), str(code[i])), val[i]))
