"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace --stats) as a per-kernel table."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
    total = sum(r[2] for r in rows)
    lines = ["%-6s %10s %12s %10s %10s %6s  %s" % ("calls", "total_ms", "avg_us", "min_us", "max_us", "pct", "kernel")]
    for name, n, tot, avg, mn, mx in rows:
        lines.append("%-6d %10.3f %12.2f %10.2f %10.2f %6.2f  %s" % (n, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3,
                                                                   100.0 * tot / total, name[:150]))
    lines.append("total kernel time %.3f ms" % (total / 1e6))
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
