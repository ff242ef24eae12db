"""Summarises a rocprofv3 rocpd sqlite database into a per-kernel stats table (name, calls, total/avg/min/max duration)."""
import sqlite3
import sys


def main(db_path, out_path=None):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    q = f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc"
    rows = list(cur.execute(q))
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total_ms | avg_us | min_us | max_us | pct |", "|---|---|---|---|---|---|---|"]
    for n, c, t, a, mn, mx in rows:
        short = n if len(n) < 110 else n[:107] + "..."
        lines.append(f"| `{short}` | {c} | {t/1e6:.3f} | {a/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*t/total:.1f} |")
    text = "\n".join(lines)
    if out_path:
        open(out_path, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main(*sys.argv[1:])
