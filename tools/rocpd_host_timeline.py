#!/usr/bin/env python3
"""One synchronous host call seen from both sides: HIP API calls (host), memory copies and kernels (device) of a rocprofv3
--hip-trace --kernel-trace --memory-copy-trace database, merged by start time, from the anchor-th-last host-to-device copy to
the next one.  usage: tools/rocpd_host_timeline.py results.db [which=-3]"""
import sqlite3
import sys

from rocpd_summary import short


def cols(db, t):
    return [r[1] for r in db.execute("pragma table_info(%s)" % t)]


def main():
    db = sqlite3.connect(sys.argv[1])
    which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
    views = [r[0] for r in db.execute("select name from sqlite_master where type in ('view','table')")]
    ev = []
    if "kernels" in views:
        for n, s, e in db.execute("select name, start, end from kernels"):
            ev.append((s, e, "gpu  kernel", short(n)[:70]))
    mc = "memory_copies" if "memory_copies" in views else None
    if mc:
        c = cols(db, mc)
        nm = "name" if "name" in c else c[0]
        for n, s, e in db.execute("select %s, start, end from %s" % (nm, mc)):
            ev.append((s, e, "gpu  copy", str(n)[:70]))
    rg = "regions" if "regions" in views else None
    if rg:
        c = cols(db, rg)
        nm = "name" if "name" in c else c[0]
        for n, s, e in db.execute("select %s, start, end from %s" % (nm, rg)):
            ev.append((s, e, "host api", str(n)[:70]))
    if not ev:
        print("views:", views)
        return
    ev.sort()
    h2d = [i for i, x in enumerate(ev) if x[2] == "gpu  copy" and "HOST_TO_DEVICE" in x[3].upper().replace(" ", "_")]
    if len(h2d) < 4:
        h2d = [i for i, x in enumerate(ev) if x[2] == "host api" and x[3].startswith("hipMemcpyAsync")]
    a = h2d[which]
    b = h2d[which + 1] if which + 1 < 0 else len(ev)
    # start the window at the host call that precedes the anchor copy by up to 60 us
    t0 = ev[a][0]
    lo = a
    while lo > 0 and ev[lo - 1][0] > t0 - 60000:
        lo -= 1
    print("%-10s %10s %10s  %s" % ("side", "start_us", "dur_us", "what"))
    for s, e, side, what in ev[lo:b]:
        print("%-10s %10.2f %10.2f  %s" % (side, (s - t0) / 1e3, (e - s) / 1e3, what))


if __name__ == "__main__":
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    main()
