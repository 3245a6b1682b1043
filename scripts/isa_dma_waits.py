"""Compiler-inserted `s_waitcnt vmcnt(0)` in kernels that use LDS-DMA (global_load_lds / buffer_load ... lds): the wait-count pass drains every
DMA in flight in front of an LDS read it cannot prove disjoint (see ppf_common.h lds_dma16_hidden).  Lists, per kernel of one .hip file, each
vmcnt(0) that is NOT inside an inline-assembly block, with the instruction that follows it.   python scripts/isa_dma_waits.py csrc/x.hip [filter]"""
import os, re, subprocess, sys, tempfile

src, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(tempfile.mkdtemp(), "k.s")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-I", os.path.join(root, "include"),
                "-I", os.path.join(root, "protopformer_amd", "csrc"), "-S", "--cuda-device-only", "-o", out, src], check=True, stderr=subprocess.DEVNULL)
txt = open(out).read()
for m in re.finditer(r"^(\S+):[^\n]*\n(.*?)\.Lfunc_end", txt, re.S | re.M):
    name, body = m.group(1), m.group(2).splitlines()
    if flt not in name or not any("load_lds" in l or (" lds" in l and "buffer_load" in l) for l in body):
        continue
    in_asm, hits = False, []
    for i, l in enumerate(body):
        s = l.strip()
        if "#ASMSTART" in s: in_asm = True
        if "#ASMEND" in s: in_asm = False
        if not in_asm and re.match(r"s_waitcnt vmcnt\(0\)", s):
            nxt = next((b.strip() for b in body[i + 1:i + 6] if b.strip() and not b.strip().startswith((";", "."))), "")
            inloop = any("Loop" in b for b in body[max(0, i - 400):i] if b.strip().startswith(".LBB"))
            hits.append((i, nxt[:60], inloop))
    ndma = sum(1 for l in body if "load_lds" in l or (" lds" in l and "buffer_load" in l))
    print(f"{name[:96]}  ({len(body)} lines, {ndma} DMA issues)")
    for i, nxt, inloop in hits:
        print(f"    line {i:5d} vmcnt(0) before: {nxt}")
