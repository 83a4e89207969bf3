"""Turn a rocprofv3 (rocpd sqlite) result into a per-kernel text summary for profiles/.

    python tools/rocpd_summary.py gpurun_out/prof_r01/bench_results.db > profiles/r01_bench_kernel_stats.txt

Durations come from rocpd_kernel_dispatch (end - start, nanoseconds).
"""
import sqlite3
import sys


def main(path, cmd=""):
    c = sqlite3.connect(path)
    rows = c.execute(
        "select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), "
        "max(d.end - d.start), max(s.arch_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), "
        "max(d.private_segment_size) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        "group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats  ({cmd})")
    print(f"# source db: {path}; durations in microseconds")
    print(f"{'calls':>6} {'total_us':>11} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'pct':>6} {'vgpr':>5} "
          f"{'sgpr':>5} {'lds_B':>7} {'scr_B':>6}  kernel")
    for name, calls, tot, avg, mn, mx, vg, sg, lds, scr in rows:
        print(f"{calls:6d} {tot / 1e3:11.1f} {avg / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} "
              f"{100.0 * tot / total:6.2f} {vg:5d} {sg:5d} {lds:7d} {scr:6d}  {name[:140]}")


if __name__ == "__main__":
    main(sys.argv[1], " ".join(sys.argv[2:]))
