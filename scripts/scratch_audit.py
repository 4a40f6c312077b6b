"""Development helper: where the scratch (spill) instructions of a kernel sit relative to its loops.
usage: scratch_audit.py file.s kernel_symbol_prefix
Parses the AMDGPU assembly hipcc -S emits: labels, backward branches (= loops, by line range) and scratch_load / scratch_store
instructions; prints every loop that contains scratch traffic with its nesting and the instruction counts inside it
(directly, i.e. not inside a deeper loop)."""
import re, sys, collections
path, sym = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(sym) and ":" in l and not l.startswith(sym + ".") )
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end + 1]
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] <= i:
        loops.append((labels[m.group(1)], i))
loops = sorted(set(loops))
# merge loops sharing a header (several back edges)
by_head = collections.OrderedDict()
for a, b in loops:
    by_head[a] = max(by_head.get(a, b), b)
loops = sorted(by_head.items())
def depth_chain(i):
    return [k for k, (a, b) in enumerate(loops) if a <= i <= b]
stat = collections.defaultdict(lambda: collections.Counter())
for i, l in enumerate(body):
    t = l.strip().split(" ")[0] if l.strip() else ""
    kind = "ld" if t.startswith("scratch_load") else "st" if t.startswith("scratch_store") else \
           "valu" if t.startswith("v_") else "lds" if t.startswith("ds_") else "vmem" if t.startswith(("global_", "flat_", "buffer_")) else None
    if not kind:
        continue
    ch = depth_chain(i)
    key = ch[-1] if ch else -1
    stat[key][kind] += 1
print("kernel %s: %d lines, %d loops; scratch_load %d, scratch_store %d" % (sym, len(body), len(loops),
      sum(s["ld"] for s in stat.values()), sum(s["st"] for s in stat.values())))
def parents(k):
    a, b = loops[k]
    return [j for j, (c, d) in enumerate(loops) if c <= a and b <= d and j != k]
for k in [-1] + list(range(len(loops))):
    s = stat.get(k)
    if not s:
        continue
    if k < 0:
        print("  outside every loop: scratch ld %d st %d | valu %d lds %d vmem %d" % (s["ld"], s["st"], s["valu"], s["lds"], s["vmem"]))
        continue
    a, b = loops[k]
    print("  loop %2d lines %5d-%5d depth %d parents %s: scratch ld %d st %d | valu %d lds %d vmem %d" % (
        k, a, b, len(parents(k)) + 1, parents(k), s["ld"], s["st"], s["valu"], s["lds"], s["vmem"]))
