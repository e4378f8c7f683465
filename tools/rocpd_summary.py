"""Summarise a rocprofv3 rocpd sqlite DB (ROCm 7.2 default output) as a per-kernel stats CSV —
the same columns `rocprofv3 --stats` prints (calls, total/avg/min/max ns, % of GPU time)."""
import sqlite3
import sys


def main(db_path, out_path=None):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                       "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,VGPRs,AGPRs,SGPRs,LDS,Scratch"]
    for r in rows:
        lines.append('"%s",%d,%d,%.1f,%d,%d,%.2f,%d,%d,%d,%d,%d' % (r[0], r[1], r[2], r[3], r[4], r[5],
                                                                   100.0 * r[2] / total, r[6], r[7], r[8], r[9], r[10]))
    text = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:])
