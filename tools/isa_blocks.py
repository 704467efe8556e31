"""Per-basic-block instruction mix of one kernel in a build/obj/*.s file (a reading aid for the hand-scheduled loops).
usage: python tools/isa_blocks.py <file.s> <symbol prefix> [first block] [last block]"""
import collections
import re
import sys


def main():
    path, sym = sys.argv[1], sys.argv[2]
    first = sys.argv[3] if len(sys.argv) > 3 else None
    last = sys.argv[4] if len(sys.argv) > 4 else None
    txt = open(path).read().split("\n")
    st = [i for i, l in enumerate(txt) if l.startswith(sym)][0]
    en = [i for i, l in enumerate(txt) if i > st and ".Lfunc_end" in l][0]
    blocks, cur = [], ["entry", collections.Counter(), []]
    for ln in txt[st:en]:
        t = ln.strip()
        if re.match(r"^\.LBB\d+_\d+:", t):
            blocks.append(cur)
            cur = [t.split(":")[0], collections.Counter(), []]
            continue
        if re.match(r"^; %bb\.\d+", t):
            blocks.append(cur)
            cur = [t.split()[1], collections.Counter(), []]
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        op = t.split()[0]
        cur[1]["mfma" if "mfma" in op else op.split("_")[0]] += 1
        if op.startswith("s_cbranch") or op.startswith("s_branch"):
            cur[2].append(op[2:] + "->" + t.split()[1])
    blocks.append(cur)
    on = first is None
    for n, c, br in blocks:
        if n == first:
            on = True
        if on:
            print(n, sum(c.values()), dict(c), br)
        if n == last:
            break


if __name__ == "__main__":
    main()
