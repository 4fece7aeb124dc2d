"""Summarise a rocprofv3 --kernel-trace results.db (rocpd sqlite) into a per-kernel table."""
import re
import sqlite3
import sys


def main(path, top=40):
    c = sqlite3.connect(path)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), grid_x, grid_y, grid_z, workgroup_x, "
        "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name, grid_x, grid_y, grid_z "
        "order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    print("total kernel time %.3f ms over %d dispatches" % (tot / 1e6, sum(r[1] for r in rows)))
    print("%-52s %5s %10s %10s %10s %-18s %4s %9s %7s %6s" % ("kernel", "n", "total_ms", "avg_us", "min_us", "grid(wg)", "wg", "vgpr+agpr", "lds", "%"))
    for r in rows[:top]:
        nm = re.sub(r"ttsamd::conv1d_mfma_kernel<(.*?)>.*", r"conv<\1>", r[0])
        nm = re.sub(r"\(.*", "", nm)[:52]
        grid = "%dx%dx%d" % (r[5] // max(r[8], 1), r[6], r[7])
        print("%-52s %5d %10.3f %10.1f %10.1f %-18s %4d %5d+%-3d %7d %6.1f" % (nm, r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, grid, r[8], r[9], r[10], r[11], 100.0 * r[2] / tot))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
