"""Summarise a rocprofv3 rocpd SQLite database (`rocprofv3 --kernel-trace --stats`) as a per-kernel table:
calls, total / average / min / max duration, share of GPU kernel time, registers and LDS.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/r01_kernel_stats.md
"""
import sqlite3
import sys


def main(path, skip_first_frac=0.0):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"# rocprofv3 kernel-trace summary: {path}")
    print(f"\ntotal GPU kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | sgpr | lds B | max grid | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for n, c, tot, avg, mn, mx, v, a, s, lds, g, wg in rows:
        n = n.replace("|", "\\|")
        if len(n) > 110:
            n = n[:107] + "..."
        print(f"| `{n}` | {c} | {tot / 1e6:.3f} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | "
              f"{100 * tot / total:.1f} | {v} | {a} | {s} | {lds} | {g} | {wg} |")


if __name__ == "__main__":
    main(sys.argv[1])
