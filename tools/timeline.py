"""dev: concurrency analysis of a rocprofv3 kernel trace (csv) of the pipelined bench run."""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[-40:], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows]
ev.sort()
# the pipelined loop = the densest burst: split at idle gaps > 0.5 ms, keep the burst with most kernels, drop its edges
segs, cur = [], [ev[0]]
for e in ev[1:]:
    if e[0] - max(x[1] for x in cur[-8:]) > 500_000:
        segs.append(cur); cur = []
    cur.append(e)
segs.append(cur)
seg = max(segs, key=len)
t0, t1 = seg[len(seg) // 6][0], seg[5 * len(seg) // 6][0]
win = [e for e in seg if t0 <= e[0] < t1]
busy = sum(e[1] - e[0] for e in win)
# union of intervals = time with at least one kernel running
cur_s, cur_e, union = None, None, 0
for s, e, _, _ in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None: union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print(f"window {(t1 - t0) / 1e3:.0f} us, kernels {len(win)}, sum of durations {busy / 1e3:.0f} us, "
      f"avg concurrency {busy / (t1 - t0):.2f}, time with >=1 kernel running {union / (t1 - t0):.3f}")
# concurrency histogram
pts = sorted([(s, 1) for s, e, _, _ in win] + [(e, -1) for s, e, _, _ in win])
hist = collections.Counter(); level = 0; last = pts[0][0]
for t, d in pts:
    hist[level] += t - last; last = t; level += d
tot = sum(hist.values())
print("time share by number of kernels in flight:", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
# per-kernel duration in the pipelined run vs alone
dur = collections.defaultdict(list)
for s, e, k, _ in win: dur[k].append((e - s) / 1e3)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k:42s} n={len(v):4d} avg {sum(v) / len(v):7.1f} us  total {sum(v):9.0f}")
