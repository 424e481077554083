"""Ablation builds of the exact-fp32 one-pass K2 kernel (timing experiments only: the variants compute garbage).  Each variant is a
text substitution on a copy of csrc/infonce_f32.hip, linked with the product's other objects into
moma_amd/lib/variants/libmoma_f32_<name>.so; select one with MOMA_HIP_LIB=<path>.  None of the variants drops a wait that covers an
inline-asm read (kernel-change checklist): they remove whole phases -- the exchange, its barriers, the exponentials, the
tile refill -- so no register is re-used under a load in flight and no address is derived from data.
usage: python scripts/build_k2_f32_variants.py [names...]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "moma_amd", "csrc", "infonce_f32.hip")
OBJ = os.path.join(ROOT, "moma_amd", "lib", "obj")
OUT = os.path.join(ROOT, "moma_amd", "lib", "variants")

XCH_WRITE = '''#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *reinterpret_cast<float4*>(&xs[((wave * 4 + g4) * 64 + lane) * 4]) = make_float4(x[4 * g4], x[4 * g4 + 1], x[4 * g4 + 2], x[4 * g4 + 3]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
'''
XCH_READ_HEAD = '''#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float4 s = *reinterpret_cast<const float4*>(&xs[((0 * 4 + g4) * 64 + lane) * 4]);
'''
BAR2 = '''            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                   // xs is free for the next tile
'''
EXP = "                x[r] = __builtin_amdgcn_exp2f(x[r] - m_ref);\n"
ADVANCE = '''        if (tn < t1) {
            dma_seg(tn, seg_of(jn), (((tn - t0) * JOBS + jn) & 1));
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
'''
VARIANTS = {
    "base": [],
    # the two barriers of the exchange only (LDS writes and reads stay)
    "nobar": [(XCH_WRITE, XCH_WRITE.replace("            __builtin_amdgcn_s_barrier();\n", "")),
              (BAR2, BAR2.replace("            __builtin_amdgcn_s_barrier();                                   // xs is free for the next tile\n", ""))],
    # no exchange at all: every wave goes on with its own partial tile
    "noexch": [(XCH_WRITE, ""), (XCH_READ_HEAD, "#pragma unroll\n            for (int g4 = 0; g4 < 0; ++g4) {\n                float4 s = *reinterpret_cast<const float4*>(&xs[((0 * 4 + g4) * 64 + lane) * 4]);\n"),
               (BAR2, "")],
    "noexp": [(EXP, "                x[r] = x[r] - m_ref;\n")],
    # the tile refill: no LDS-DMA, no vmcnt wait (the products run on whatever the first tile left in LDS)
    "nodma": [(ADVANCE, "        (void)tn; (void)jn;\n")],
}
# the second barrier only (what a double-buffered exchange would save), and the refill's waits only (issue cost vs landing)
VARIANTS["onebar"] = [(BAR2, BAR2.replace("            __builtin_amdgcn_s_barrier();                                   // xs is free for the next tile\n", ""))]
VARIANTS["novmwait"] = [(ADVANCE, ADVANCE.replace('            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");\n', "")
                         .replace('            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n', "            ;\n"))]
VARIANTS["noexch_noexp"] = VARIANTS["noexch"] + VARIANTS["noexp"]
VARIANTS["noexch_noexp_nodma"] = VARIANTS["noexch_noexp"] + VARIANTS["nodma"]

names = sys.argv[1:] or list(VARIANTS)
os.makedirs(OUT, exist_ok=True)
text = open(SRC).read()
objs = [os.path.join(OBJ, f) for f in os.listdir(OBJ) if f.endswith(".o") and f != "infonce_f32.o"]
for name in names:
    t = text
    for old, new in VARIANTS[name]:
        assert t.count(old) == 1, (name, old[:70], t.count(old))
        t = t.replace(old, new, 1)
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "infonce_f32.hip")
        open(src, "w").write(t)
        obj = os.path.join(tmp, "v.o")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
                        "-Wno-unused-variable", f"-I{os.path.dirname(SRC)}", "-c", src, "-o", obj], check=True)
        lib = os.path.join(OUT, f"libmoma_f32_{name}.so")
        subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib, obj, *objs], check=True)
    print("built", name, flush=True)
