"""One character per instruction of a kernel in a build/obj/*.s file, a line per basic block: M mfma, e v_exp, v other VALU,
d LDS, g global / buffer, s SALU, w s_waitcnt, n s_nop, | s_barrier. A reading aid for interleaved loops.
usage: python tools/isa_stream.py <file.s> <symbol prefix>"""
import re
import sys

txt = open(sys.argv[1]).read().split("\n")
st = [i for i, l in enumerate(txt) if l.startswith(sys.argv[2])][0]
en = [i for i, l in enumerate(txt) if i > st and ".Lfunc_end" in l][0]
out = []
for l in txt[st:en]:
    t = l.strip()
    if re.match(r"^\.LBB", t):
        out.append("\n" + t.split(":")[0] + ": ")
        continue
    if not t or t.startswith(";") or t.startswith("."):
        continue
    op = t.split()[0]
    out.append("M" if "mfma" in op else "e" if op.startswith("v_exp") else "v" if op.startswith("v_") else
               "d" if op.startswith("ds_") else "w" if op.startswith("s_waitcnt") else "|" if op.startswith("s_barrier") else
               "n" if op.startswith("s_nop") else "s" if op.startswith("s_") else
               "g" if op.startswith("global") or op.startswith("buffer") or op.startswith("scratch") else "?")
print("".join(out))
