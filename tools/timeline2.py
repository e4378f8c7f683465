"""dev: per-stream view of a rocprofv3 kernel trace (csv) of the pipelined bench run.

For every queue/stream: the kernels in order with start offset, duration and the gap to the previous kernel of the same
stream; gap statistics split into intra-graph gaps (between the kernels of one step) and inter-graph gaps (between the last
kernel of a step and the first of the next step on that stream).  The step's first kernel is recognised by name."""
import csv, sys, collections

path = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "osc_tile_totals"
show = int(sys.argv[3]) if len(sys.argv) > 3 else 2   # graph replays per stream printed in full
rows = [r for r in csv.DictReader(open(path))]


def short(n):
    n = n.split("(")[0].replace("void ", "").replace("golf::", "")
    return n[-34:]


ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    st = r.get("Stream_Id", "?")
    ev.append((s, e, short(r["Kernel_Name"]), (q, st), int(r.get("VGPR_Count", 0) or 0), int(r.get("Grid_Size", 0) or 0),
               int(r.get("Workgroup_Size", 0) or 0)))
ev.sort()
# the pipelined loop = the densest burst
segs, cur = [], [ev[0]]
for e in ev[1:]:
    if e[0] - max(x[1] for x in cur[-8:]) > 150_000:
        segs.append(cur); cur = []
    cur.append(e)
segs.append(cur)
segs.sort(key=len, reverse=True)
for si, seg in enumerate(segs[:int(sys.argv[4]) if len(sys.argv) > 4 else 1]):
    t0 = seg[0][0]
    t1 = max(e[1] for e in seg)
    per = collections.defaultdict(list)
    for e in seg:
        per[e[3]].append(e)
    nfirst = sum(1 for e in seg if first in e[2])
    print(f"== burst {si}: {len(seg)} kernels, {nfirst} steps, {(t1 - t0) / 1e3:.1f} us  -> {(t1 - t0) / 1e3 / max(nfirst, 1):.2f} us/step; "
          f"{len(per)} queues/streams")
    intra, inter, chain = [], [], []
    for key, lst in sorted(per.items()):
        lst.sort()
        prev_end, graph_start, n_graph = None, None, 0
        lines = []
        for s, e, k, _, vg, gs, wg in lst:
            newg = first in k
            gap = None if prev_end is None else (s - prev_end) / 1e3
            if newg:
                if graph_start is not None:
                    chain.append((prev_end - graph_start) / 1e3)
                graph_start = s
                n_graph += 1
                if gap is not None:
                    inter.append(gap)
            elif gap is not None:
                intra.append(gap)
            if n_graph <= show or n_graph == len([1 for x in lst if first in x[2]]):
                lines.append(f"    {'*' if newg else ' '} +{(s - t0) / 1e3:9.1f}  dur {(e - s) / 1e3:7.1f}  gap {gap if gap is None else round(gap, 1)!s:>7}  "
                             f"{k:36s} wg {gs // max(wg, 1):5d}x{wg:<4d} vgpr {vg}")
            prev_end = e
        if graph_start is not None:
            chain.append((prev_end - graph_start) / 1e3)
        print(f"  queue/stream {key}: {len(lst)} kernels, {n_graph} steps, first start +{(lst[0][0] - t0) / 1e3:.1f}, last end +{(lst[-1][1] - t0) / 1e3:.1f}")
        for l in lines:
            print(l)

    def stats(v):
        if not v:
            return "n/a"
        v = sorted(v)
        return f"n={len(v)} mean {sum(v) / len(v):.1f} median {v[len(v) // 2]:.1f} p90 {v[int(len(v) * 0.9)]:.1f} max {v[-1]:.1f} sum {sum(v):.0f}"

    print("  intra-step gaps (us):", stats(intra))
    print("  inter-step gaps (us):", stats(inter))
    print("  step chain first-start -> last-end (us):", stats(chain))
    dur = collections.defaultdict(list)
    for s, e, k, *_ in seg:
        dur[k].append((e - s) / 1e3)
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print(f"    {k:36s} n={len(v):4d} avg {sum(v) / len(v):7.1f}  min {min(v):6.1f} max {max(v):6.1f}")
