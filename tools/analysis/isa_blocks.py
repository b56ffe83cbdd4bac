"""Dev tool: per-basic-block instruction mix of one kernel in a hipcc -S listing.

    python tools/analysis/isa_blocks.py file.s <kernel name substring> [min loop depth]

Prints every basic block (label, loop depth as annotated by the compiler) with its VALU / SALU / branch / LDS / VMEM /
SMEM counts, so that hot loops can be priced statically before a GPU run."""
import re
import sys


def classify(op):
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "xlane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_endpgm", "s_barrier", "s_sleep")):
        return "wait"
    if op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_store", "s_dcache")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    mind = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and l.rstrip().endswith(":") is False and ":" in l)
    blocks = []
    cur = {"label": "entry", "depth": 0, "c": {}, "line": start}
    for i in range(start + 1, len(lines)):
        l = lines[i]
        if l.startswith("\t.section") or l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", l) or re.match(r"^; %bb\.(\d+):", l)
        if m:
            blocks.append(cur)
            d = re.search(r"Depth=(\d+)", l)
            cur = {"label": m.group(1), "depth": int(d.group(1)) if d else 0, "c": {}, "line": i, "note": ""}
            continue
        d = re.search(r"Depth=(\d+)", l)
        if d and l.strip().startswith(";") and not cur["c"]:
            cur["depth"] = max(cur["depth"], int(d.group(1)))
        s = l.strip()
        if s.startswith(";") and "ASMSTART" not in s and "ASMEND" not in s and "Loop" not in s and "Depth" not in s and len(s) > 2:
            cur["note"] = cur.get("note", "") + s[1:].strip()[:40] + " "
        if not s or s.startswith((";", ".")):
            continue
        op = s.split()[0]
        k = classify(op)
        cur["c"][k] = cur["c"].get(k, 0) + 1
    blocks.append(cur)
    tot = {}
    print("%-14s %5s %5s %5s %5s %5s %5s %5s %5s  %s" % ("block", "line", "depth", "valu", "salu", "br", "lds", "vmem", "xlane", "note"))
    for b in blocks:
        for k, v in b["c"].items():
            tot[k] = tot.get(k, 0) + v
        if b["depth"] < mind:
            continue
        c = b["c"]
        print("%-14s %5d %5d %5d %5d %5d %5d %5d %5d  %s" % (b["label"], b["line"] + 1, b["depth"], c.get("valu", 0), c.get("salu", 0), c.get("branch", 0),
                                                           c.get("lds", 0), c.get("vmem", 0), c.get("xlane", 0), b.get("note", "")))
    print("total", tot)


if __name__ == "__main__":
    main()
