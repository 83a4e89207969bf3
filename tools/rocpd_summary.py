"""Turn a rocprofv3 (rocpd sqlite) result into a per-kernel text summary for profiles/.

    python tools/rocpd_summary.py <results.db> [--last-ms W] [command line being profiled ...]

Durations come from rocpd_kernel_dispatch (end - start, nanoseconds).  --last-ms W keeps only
the dispatches that started in the last W milliseconds of the trace (e.g. the timed steps of
bench.py, leaving out MIOpen's find-mode trial kernels during warm-up).
"""
import sqlite3
import sys


def main(argv):
    path = argv[0]
    rest = argv[1:]
    last_ms = None
    if rest and rest[0] == "--last-ms":
        last_ms = float(rest[1])
        rest = rest[2:]
    cmd = " ".join(rest)
    c = sqlite3.connect(path)
    where = ""
    if last_ms is not None:
        tmax = c.execute("select max(end) from rocpd_kernel_dispatch").fetchone()[0]
        where = f"where d.start >= {tmax - int(last_ms * 1e6)}"
    rows = c.execute(
        "select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), "
        "max(d.end - d.start), max(s.arch_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), "
        "max(d.private_segment_size) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        f"{where} group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats  ({cmd})")
    print(f"# source db: {path}; durations in microseconds" +
          (f"; only dispatches of the last {last_ms:.0f} ms of the trace" if last_ms else ""))
    print(f"# total kernel time in window: {total / 1e3:.1f} us over {sum(r[1] for r in rows)} dispatches")
    print(f"{'calls':>6} {'total_us':>11} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'pct':>6} {'vgpr':>5} "
          f"{'sgpr':>5} {'lds_B':>7} {'scr_B':>6}  kernel")
    for name, calls, tot, avg, mn, mx, vg, sg, lds, scr in rows:
        print(f"{calls:6d} {tot / 1e3:11.1f} {avg / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} "
              f"{100.0 * tot / total:6.2f} {vg:5d} {sg:5d} {lds:7d} {scr:6d}  {name[:140]}")


if __name__ == "__main__":
    main(sys.argv[1:])
