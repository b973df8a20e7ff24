"""Basic-block instruction mix of one kernel in a hipcc -S listing: for every label, the number of VALU / SALU /
SMEM / VMEM / LDS / branch instructions up to the next label, with the loop-nest comment the compiler attached.
usage: isa_blocks.py <file.s> <kernel symbol substring> [min loop depth]"""
import re, sys
src, key = sys.argv[1], sys.argv[2]
min_depth = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and re.match(r"^_Z\S+:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks, cur = [], dict(label="entry", note="", c=dict(valu=0, salu=0, smem=0, vmem=0, lds=0, br=0, wait=0), lines=[])
def kind(op):
    if op.startswith("v_"): return "valu"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "br"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait"
    if op.startswith("s_"): return "salu"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")): return "vmem"
    if op.startswith("ds_"): return "lds"
    return None
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
    if m:
        blocks.append(cur)
        cur = dict(label=m.group(1), note=(m.group(2) or "").strip("; "), c=dict(valu=0, salu=0, smem=0, vmem=0, lds=0, br=0, wait=0), lines=[])
        continue
    t = l.strip()
    if not t or t.startswith((";", ".", "//")):
        if t.startswith(";") and ("Loop" in t) and cur["lines"] == []:
            cur["note"] += " | " + t.strip("; ")
        continue
    op = t.split()[0]
    k = kind(op)
    if k:
        cur["c"][k] += 1
        cur["lines"].append(t)
blocks.append(cur)
tot = dict(valu=0, salu=0, smem=0, vmem=0, lds=0, br=0, wait=0)
for b in blocks:
    d = max([int(x) for x in re.findall(r"Depth=(\d+)", b["note"])] or [0])
    for k in tot: tot[k] += b["c"][k]
    if d >= min_depth:
        c = b["c"]
        tgt = [x.split()[-1] for x in b["lines"] if x.startswith(("s_cbranch", "s_branch"))]
        print(f"{b['label']:12s} d{d} valu {c['valu']:4d} salu {c['salu']:3d} smem {c['smem']:2d} vmem {c['vmem']:2d} lds {c['lds']:2d} br {c['br']:2d} -> {','.join(tgt):40s} {b['note'][:70]}")
print("total", tot)
