"""Per-queue view of ONE replayed training step from a rocprofv3 --kernel-trace database (the step before the last Adam launch):
busy time per hardware queue, GPU idle time, the heaviest kernels per queue and, with --dump, the launch-by-launch timeline.
usage: python tools/prof_timeline.py <results.db> [--dump [from_us [to_us]]] [--json out.json]
--json: the per-kernel table of that one step ({kernel: {"launches", "us"}} by demangled-name prefix) -- bench.py reads the committed copy
(profiles/step_kernel_table.json) to decide which kernel is the step's dominant one."""
import collections
import sqlite3
import sys


def load(db):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = c.execute(f"select d.start,d.end,d.queue_id,s.kernel_name,d.grid_size_x,d.workgroup_size_x from {kd} d join {ks} s "
                     "on d.kernel_id=s.id order by d.start").fetchall()
    adam = [i for i, r in enumerate(rows) if "adam" in r[3]]
    last = adam[-1]
    prev = [i for i in adam if i < last - 100][-1]
    return rows[prev + 1:last + 1]


def main():
    seg = load(sys.argv[1])
    t0 = seg[0][0]
    T = (seg[-1][1] - t0) / 1e3
    print("step %.0f us, %d launches" % (T, len(seg)))
    byq = collections.defaultdict(list)
    for r in seg:
        byq[r[2]].append(r)
    ev = sorted((r[0], r[1]) for r in seg)
    cs, ce = ev[0]
    tot = 0
    for s, e in ev[1:]:
        if s > ce:
            tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    tot += ce - cs
    print("GPU busy (union) %.0f us, idle %.0f us" % (tot / 1e3, T - tot / 1e3))
    for q, v in sorted(byq.items()):
        busy = sum(r[1] - r[0] for r in v) / 1e3
        print("queue %d: %d launches, busy %.0f us, %.0f .. %.0f us" % (q, len(v), busy, (v[0][0] - t0) / 1e3, (v[-1][1] - t0) / 1e3))
        agg, cnt = collections.Counter(), collections.Counter()
        for r in v:
            agg[r[3][:44]] += (r[1] - r[0]) / 1e3
            cnt[r[3][:44]] += 1
        for k, t in agg.most_common(14):
            print("      %-46s %4d %8.0f us" % (k, cnt[k], t))
    if "--json" in sys.argv:
        import json
        import re
        import subprocess
        agg, cnt = collections.Counter(), collections.Counter()
        for r in seg:
            agg[r[3]] += (r[1] - r[0]) / 1e3
            cnt[r[3]] += 1
        names = list(agg)
        try:                                                     # demangle (c++filt ships with binutils / llvm)
            dem = subprocess.run(["c++filt"], input="\n".join(n[:-3] if n.endswith(".kd") else n for n in names), capture_output=True,
                                 text=True).stdout.splitlines()
        except OSError:
            dem = names
        table = {}
        for n, d in zip(names, dem):
            key = re.sub(r"\(.*$", "", d).replace("void ", "").strip()
            e = table.setdefault(key, {"launches": 0, "us": 0.0})
            e["launches"] += cnt[n]
            e["us"] = round(e["us"] + agg[n], 1)
        out = {"step_us_profiled": round(T, 1), "launches": len(seg),
               "kernels": dict(sorted(table.items(), key=lambda kv: -kv[1]["us"]))}
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
    if "--dump" in sys.argv:
        i = sys.argv.index("--dump")
        def num(j, dflt):
            try:
                return float(sys.argv[j])
            except (IndexError, ValueError):
                return dflt
        lo, hi = num(i + 1, 0.0), num(i + 2, 1e12)
        for r in seg:
            ts = (r[0] - t0) / 1e3
            if lo <= ts <= hi:
                print("%8.1f %7.1f q%-2d wg%-5d %s" % (ts, (r[1] - r[0]) / 1e3, r[2], r[4] // max(r[5], 1), r[3][:70]))


if __name__ == "__main__":
    main()
