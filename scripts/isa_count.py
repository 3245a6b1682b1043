"""Static instruction mix of the kernels of one .hip file (cross-compiled to gfx950 assembly): VALU / MFMA / LDS / exp / packed-fp32 counts and
register use per kernel whose name contains FILTER.   python scripts/isa_count.py protopformer_amd/csrc/attention.hip attn_fwd16"""
import collections, os, re, subprocess, sys, tempfile

src, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(tempfile.mkdtemp(), "k.s")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-I", os.path.join(root, "include"),
                "-I", os.path.join(root, "protopformer_amd", "csrc"), "-S", "--cuda-device-only", "-o", out, src] + sys.argv[3:], check=True, stderr=subprocess.DEVNULL)
txt = open(out).read()
meta = {}
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    b = m.group(2)
    g = lambda k: (re.search(k + r" (\d+)", b) or [None, "?"])[1]
    meta[m.group(1)] = (g("next_free_vgpr"), g("accum_offset"), g("private_segment_fixed_size"), g("group_segment_fixed_size"))
for m in re.finditer(r"^(\S+):[^\n]*\n(.*?)\.Lfunc_end", txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if name not in meta or flt not in name:
        continue
    c = collections.Counter(re.sub(r"_(e32|e64|dpp|sdwa)$", "", op) for op in re.findall(r"^\s+([vsd]\w+|global_\w+|buffer_\w+)", body, re.M))
    valu = sum(v for k, v in c.items() if k.startswith("v_") and "mfma" not in k)
    pick = {k: c[k] for k in ("v_exp_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_fma_f32", "v_add_f32", "v_mul_f32", "v_cndmask_b32", "v_max3_f32", "v_max_f32",
                              "v_cvt_pk_bf16_f32", "v_accvgpr_write_b32", "v_accvgpr_read_b32", "v_mov_b32") if c[k]}
    print(name[:90])
    print("   vgpr/accum_offset/scratch/lds", meta[name], "| VALU", valu, "MFMA", sum(v for k, v in c.items() if "mfma" in k), "LDS", sum(v for k, v in c.items() if k.startswith("ds_")),
          "VMEM", sum(v for k, v in c.items() if k.startswith(("global_", "buffer_"))), "SALU", sum(v for k, v in c.items() if k.startswith("s_")))
    print("  ", pick)
