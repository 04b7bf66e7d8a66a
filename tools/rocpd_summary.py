#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a small text/CSV table:
   python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/r01_kernel_stats.txt"""
import sqlite3
import sys


def main(path):
    import glob, os
    if os.path.isdir(path):                     # a rocprofv3 output directory: take its database
        found = sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))
        if not found:
            raise SystemExit(f"no .db under {path}")
        path = found[-1]
    db = sqlite3.connect(path)
    c = db.cursor()
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                     "max(lds_size), max(vgpr_count), max(sgpr_count), max(workgroup_x), max(grid_x) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# source: {path}")
    print("name,calls,total_ns,avg_ns,min_ns,max_ns,pct,lds_bytes,vgpr,sgpr,workgroup_x,grid_x")
    for r in rows:
        print(f"\"{r[0]}\",{r[1]},{r[2]},{r[3]:.0f},{r[4]},{r[5]},{100.0 * r[2] / total:.2f},{r[6]},{r[7]},{r[8]},{r[9]},{r[10]}")


if __name__ == "__main__":
    main(sys.argv[1])
