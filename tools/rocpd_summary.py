#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2) rocpd SQLite database: per-kernel calls,
total/avg/min/max duration and share — the `--kernel-trace --stats` table —
plus, when PMC counters were collected, the per-kernel counter sums.

usage: tools/rocpd_summary.py results.db [more.db ...] > profiles/<name>.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\[clone.*", "", name).replace("(anonymous namespace)::", "")
    m = re.match(r"void (spfe::(?:\w+::)*\w+)<(.*)>\(", name)
    if m:
        args = m.group(2).replace(" ", "")
        tag = " [conv1b]" if (m.group(1).endswith("conv_f32_kernel") and args.startswith("1,")) or \
            (m.group(1).endswith("conv_bf16_ws_kernel") and args.endswith((",1", ",2"))) else ""
        return "%s<%s>%s" % (m.group(1), args, tag)
    return name.split("(")[0][:90]


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        cur = db.cursor()
        rows = cur.execute("select name, duration, grid_x, grid_y, grid_z, workgroup_x, lds_size, "
                           "vgpr_count, accum_vgpr_count, sgpr_count from kernels").fetchall()
        agg = {}
        for name, dur, gx, gy, gz, wx, lds, vg, ag, sg in rows:
            k = (short(name), (gx, gy, gz))  # one row per kernel AND launch geometry
            a = agg.setdefault(k, dict(n=0, tot=0, mn=1 << 62, mx=0, grid=(gx, gy, gz), wg=wx, lds=lds,
                                        vgpr=vg, agpr=ag, sgpr=sg))
            a["n"] += 1
            a["tot"] += dur
            a["mn"] = min(a["mn"], dur)
            a["mx"] = max(a["mx"], dur)
        total = sum(a["tot"] for a in agg.values()) or 1
        print("# %s" % path)
        print("# rocprofv3 --kernel-trace --stats summary (durations in us)")
        print("%-74s %6s %11s %10s %10s %10s %6s  %-16s %5s %6s %5s" %
              ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%", "grid", "wg", "lds", "vgpr"))
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["tot"]):
            print("%-74s %6d %11.1f %10.2f %10.2f %10.2f %6.2f  %-16s %5d %6d %5d" %
                  (k[0][:74], a["n"], a["tot"] / 1e3, a["tot"] / a["n"] / 1e3, a["mn"] / 1e3, a["mx"] / 1e3,
                   100.0 * a["tot"] / total, "x".join(str(g) for g in a["grid"]), a["wg"], a["lds"],
                   a["vgpr"] + a["agpr"]))
        try:
            pm = cur.execute("select name, counter_name, sum(counter_value), count(*), min(counter_value), "
                             "max(counter_value) from pmc_events group by name, counter_name").fetchall()
        except sqlite3.Error:
            pm = []
        if pm:
            print("\n# PMC counters per kernel: sum over dispatches, dispatches, min, max per dispatch")
            for kname, cname, val, cnt, mn, mx in sorted(pm):
                print("%-74s %-28s %18.1f %5d %16.1f %16.1f" % (short(kname)[:74], cname, val, cnt, mn, mx))
        print()


if __name__ == "__main__":
    main()
