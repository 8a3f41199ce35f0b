#!/usr/bin/env python3
"""Timeline of ONE call from a rocprofv3 rocpd database: every kernel between two consecutive launches of `anchor`
(default conv1a), with start offset, duration, and the gap to the latest end among the kernels that started before it.
usage: tools/rocpd_timeline.py results.db [anchor-substring] [which-occurrence]"""
import sqlite3
import sys

from rocpd_summary import short


def main():
    db = sqlite3.connect(sys.argv[1])
    anchor = sys.argv[2] if len(sys.argv) > 2 else "conv1a"
    which = int(sys.argv[3]) if len(sys.argv) > 3 else -3
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = db.execute("select name, start, end, %s from kernels order by start" % qcol).fetchall()
    idx = [i for i, r in enumerate(rows) if anchor in r[0]]
    a = idx[which]
    b = idx[which + 1] if which + 1 < 0 or which + 1 < len(idx) else len(rows)
    if which + 1 == 0:
        b = len(rows)
    t0 = rows[a][1]
    last_end = t0
    print("# one call: kernels from launch %d of '%s' to the next; us" % (which, anchor))
    print("%-60s %9s %9s %9s %6s" % ("kernel", "start", "dur", "gap", "queue"))
    busy = 0.0
    for name, st, en, q in rows[a:b]:
        print("%-60s %9.2f %9.2f %9.2f %6s" % (short(name)[:60], (st - t0) / 1e3, (en - st) / 1e3, (st - last_end) / 1e3, q))
        last_end = max(last_end, en)
    print("# span %.2f us" % ((last_end - t0) / 1e3))


if __name__ == "__main__":
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    main()
