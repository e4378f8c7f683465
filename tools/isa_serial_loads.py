"""dev: scan the gfx950 ISA of every kernel for loads that are waited for one at a time -- a loop (or a run of branches)
whose body holds one or two global/buffer loads and an `s_waitcnt vmcnt(0)`: each trip is a full memory round trip on the
wave's critical path (DESIGN.md 4.1 "Serial round trips hiding in staging loops").
usage: python tools/isa_serial_loads.py [name-filter]     (compiles golf_amd/csrc/*.hip with --save-temps into /tmp)"""
import glob, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = "/tmp/golf_isa"
os.makedirs(OUT, exist_ok=True)
flt = sys.argv[1] if len(sys.argv) > 1 else ""
procs = []
for src in sorted(glob.glob(os.path.join(ROOT, "golf_amd", "csrc", "*.hip"))):
    base = os.path.basename(src)[:-4]
    procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-w",
                                   "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "golf_amd", "csrc"),
                                   "-c", src, "-o", os.path.join(OUT, base + ".o"), "--save-temps"], cwd=OUT,
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
for p in procs:
    p.wait()
LOAD = re.compile(r"\b(global_load|buffer_load)")
for path in sorted(glob.glob(os.path.join(OUT, "*-hip-amdgcn-amd-amdhsa-gfx950.s"))):
    funcs, cur = {}, None
    for line in open(path).read().split("\n"):
        m = re.match(r"^(_ZN4golf\w+):", line)
        if m:
            cur = m.group(1)
            funcs[cur] = []
        if cur:
            funcs[cur].append(line)
        if line.startswith(".Lfunc_end"):
            cur = None
    for name, lines in funcs.items():
        if flt and flt not in name:
            continue
        labels = {}
        for k, l in enumerate(lines):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                labels[m.group(1)] = k
        hits = []
        for k, l in enumerate(lines):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < k:
                body = lines[labels[m.group(1)]:k]
                nload = sum(1 for x in body if LOAD.search(x))
                if len(body) <= 400 and 1 <= nload <= 2 and any("s_waitcnt vmcnt(0)" in x for x in body):
                    hits.append((labels[m.group(1)], k, nload))
        # straight-line runs: a load followed within a few instructions by a full wait, repeatedly
        run = 0
        for k, l in enumerate(lines):
            if LOAD.search(l):
                for kk in range(k + 1, min(k + 8, len(lines))):
                    if LOAD.search(lines[kk]):
                        break
                    if "s_waitcnt vmcnt(0)" in lines[kk]:
                        run += 1
                        break
        if hits or run >= 4:
            print(f"{os.path.basename(path)[:14]:14s} {name[8:78]:70s} loops {hits[:4]} load->wait(0) {run}")
