"""Instruction mix per basic block of one kernel in a hipcc -S listing: python scripts/asm_blocks.py file.s <mangled-name-substring> [min_instructions]."""
import re
import sys
from collections import Counter

src, name = open(sys.argv[1]).read(), sys.argv[2]
min_ins = int(sys.argv[3]) if len(sys.argv) > 3 else 60
m = re.search(r"^(_Z\w*" + re.escape(name) + r"\w*):[^\n]*\n(.*?)\n\s*s_endpgm", src, re.S | re.M)
body = m.group(2)
parts = re.split(r"\n(\.LBB[0-9_]+):", body)
print(m.group(1), "lines", body.count("\n"))
cur = "entry"
for i, b in enumerate(parts):
    if i % 2 == 1:
        cur = b
        continue
    ins = [l.strip().split()[0] for l in b.split("\n") if l.strip() and not l.strip().startswith((";", "."))]
    if len(ins) < min_ins:
        continue
    c = Counter()
    for x in ins:
        key = ("mfma" if x.startswith("v_mfma") else "exp" if x.startswith("v_exp") else "vpk" if x.startswith("v_pk_") else "acc_mov" if x.startswith("v_accvgpr")
               else "valu" if x.startswith("v_") else "ds" if x.startswith("ds_") else "vmem" if x.startswith(("global_", "buffer_")) else "scratch"
               if x.startswith("scratch_") else "wait" if x.startswith("s_waitcnt") else "nop" if x.startswith("s_nop") else "salu" if x.startswith("s_") else "other")
        c[key] += 1
    print("  block", cur, len(ins), dict(c))
