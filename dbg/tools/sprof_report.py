# dbg/tools/sprof_report.py sprof.out [top]: flat profile by function (leaf) and inclusive profile (any frame), symbolised with addr2line against the mapped files
import sys, subprocess, collections, os, bisect
path = sys.argv[1]; top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
maps = []; samples = []
for line in open(path):
    if line.startswith("M "):
        f = line[2:].split()
        maps.append((int(f[0], 16), int(f[1], 16), 0, f[2] if len(f) > 2 else ""))
    elif line.startswith("S"):
        samples.append([int(x, 16) for x in line.split()[1:]])
maps.sort()
def locate(a):
    for lo, hi, off, name in maps:
        if lo <= a < hi and (lo or name == "/proc/self/exe"): return name, a - lo
    return "?", a
byfile = collections.defaultdict(set)
for s in samples:
    for k, a in enumerate(s):
        name, rel = locate(a if k == 0 else a - 1)
        byfile[name].add(rel)
sym = {}
for name, addrs in byfile.items():
    addrs = sorted(addrs)
    if not os.path.exists(name):
        for a in addrs: sym[(name, a)] = "%s+%x" % (os.path.basename(name), a)
        continue
    out = subprocess.run(["addr2line", "-f", "-C", "-e", name] + ["%x" % a for a in addrs], capture_output=True, text=True).stdout.splitlines()
    for i, a in enumerate(addrs):
        fn = out[2 * i] if 2 * i < len(out) else "?"
        if fn == "??": fn = "%s+%x" % (os.path.basename(name), a)
        sym[(name, a)] = fn[:110]
flat = collections.Counter(); incl = collections.Counter()
for s in samples:
    seen = set()
    for k, a in enumerate(s):
        name, rel = locate(a if k == 0 else a - 1)
        fn = sym[(name, rel)]
        if k == 0: flat[fn] += 1
        if fn not in seen: incl[fn] += 1; seen.add(fn)
n = len(samples)
print("%d samples" % n)
print("---- self ----")
for fn, c in flat.most_common(top): print("%6.2f%%  %s" % (100.0 * c / n, fn))
print("---- inclusive ----")
for fn, c in incl.most_common(top): print("%6.2f%%  %s" % (100.0 * c / n, fn))
