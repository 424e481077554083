"""Ablation builds of the one-pass K2 kernel (timing experiments only: the variants compute garbage).  Each variant is a text
substitution on a copy of csrc/infonce_fused.hip, linked with the product's other objects into moma_amd/lib/variants/libmoma_<name>.so;
select one with MOMA_HIP_LIB=<path>.   usage: python scripts/build_k2_variants.py [names...]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "moma_amd", "csrc", "infonce_fused.hip")
OBJ = os.path.join(ROOT, "moma_amd", "lib", "obj")
OUT = os.path.join(ROOT, "moma_amd", "lib", "variants")
LOOP_END = '''            if (t + NBUF - 1 < t1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");     // (one two-way branch, not the general switch)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // the next iteration's first A fragments'''
VARIANTS = {
    "base": [],
    "nobar": [(LOOP_END, LOOP_END.replace("            __builtin_amdgcn_s_barrier();\n", ""))],
    "nobar_novm": [(LOOP_END, "            // the next iteration's first A fragments")],
    "noscorewait": [("            wait_lgkm(ahead + pre);\n", "")],
    "nopvwait": [('                if (ahead == 3) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");\n'
                  '                else if (ahead == 2) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");\n'
                  '                else if (ahead == 1) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");\n'
                  '                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");\n                __builtin_amdgcn_sched_barrier(0);\n                s16x4* k4', "                s16x4* k4")],
    "norefill": [("                if ((ks & 3) == 1) dma_piece<D, REFILL == 2>(ks >> 2, dl, queue, rkey0, K, rbuf, wave, lane);\n", "")],
    "nosoftmax": [('                   if (c >= SM_SHIFT && c - SM_SHIFT < 16) sm_step_a(c - SM_SHIFT);\n', ""),
                  ('                   if (c >= SM_SHIFT && c - SM_SHIFT < 16) sm_step_b(c - SM_SHIFT);\n', "")],
}
# CAUTION (round 5): variants that drop lgkmcnt waits are not just numerically garbage.  The LDS reads are inline asm whose destination
# registers the compiler considers free again after their last use; without the wait a late LDS return can land in a register that
# has since been re-used -- e.g. as the address of the next LDS-DMA piece.  A burst-read variant built this way (four row reads per
# wait, fragment ring overwritten early) died with HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION on its first launch; the variants
# below happened to run.  Run such builds once, under `timeout`, and never in a loop.
VARIANTS["nowaits"] = VARIANTS["nobar_novm"] + VARIANTS["noscorewait"] + VARIANTS["nopvwait"]
VARIANTS["nowaits_norefill"] = VARIANTS["nowaits"] + VARIANTS["norefill"]
VARIANTS["noscorereads"] = [('        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[ks % RD]) : "v"(aa[ks & 7]), "i"((ks >> 3) * 8192) : "memory");',
                             '        asm volatile("; no read %0 %1" : "=v"(kf[ks % RD]) : "v"(aa[ks & 7]) : "memory");')]
VARIANTS["nowaits_norefill_noscorereads"] = VARIANTS["nowaits_norefill"] + VARIANTS["noscorereads"]
VARIANTS["nowaits_noscorereads"] = VARIANTS["nowaits"] + VARIANTS["noscorereads"]
# score accumulator in AGPRs (is the LDS return path contending with the MFMA's VGPR result writes?)
VARIANTS["scoreacc_agpr"] = [
    ('asm volatile("s_nop 1\\n\\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(x) : "v"(kf[ks % RD]), "v"(qf[ks]), "v"(c0));',
     'asm volatile("s_nop 1\\n\\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&a"(x) : "v"(kf[ks % RD]), "v"(qf[ks]), "a"(c0));'),
    ('asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\\n\\ts_nop 15\\n\\ts_nop 3" : "+v"(x) : "v"(kf[ks % RD]), "v"(qf[ks]));',
     'asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\\n\\ts_nop 15\\n\\ts_nop 3" : "+a"(x) : "v"(kf[ks % RD]), "v"(qf[ks]));'),
    ('                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(x) : "v"(kf[ks % RD]), "v"(qf[ks]));',
     '                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(x) : "v"(kf[ks % RD]), "v"(qf[ks]));'),
]
# combine experiments: two column tiles per block (half the blocks, the row statistics repeated half as often) / no q.k dot in the combine
COMB_OLD = ('    hipExtLaunchKernelGGL((infonce_combine_kernel<1>), dim3(p.Bpad / 8, dq ? d / 32 : 1), dim3(256), 0, st, (hipEvent_t) nullptr, ev_call_end, 0, q,\n'
            '                          k, B, d, inv_T, p.nchunk, p.Bpad, o_part, m_part, l_part, x_part, loss_rows, lse, top1, dq, 0L, 1, 1, m_part);')
VARIANTS["comb_tpb2"] = [(COMB_OLD, COMB_OLD.replace("dq ? d / 32 : 1", "dq ? d / 64 : 1").replace("0L, 1, 1, m_part", "0L, 1, 2, m_part"))]
VARIANTS["comb_tpb4"] = [(COMB_OLD, COMB_OLD.replace("dq ? d / 32 : 1", "dq ? d / 128 : 1").replace("0L, 1, 1, m_part", "0L, 1, 4, m_part"))]
NOS0 = [('        for (int c = l32 * 4; c < D; c += 128) {\n            const float4 qa = *reinterpret_cast<const float4*>(q + (long)bb * D + c);',
         '        for (int c = l32 * 4; c < D && inv_T < 0.f; c += 128) {\n            const float4 qa = *reinterpret_cast<const float4*>(q + (long)bb * D + c);')]
VARIANTS["comb_nos0"] = NOS0
VARIANTS["comb_tpb2_nos0"] = VARIANTS["comb_tpb2"] + NOS0
# two interleaved accumulation chains in the score product (timing only: the odd k-steps accumulate into the registers of the
# row-constant tuple) -- is the single dependent chain on a VGPR accumulator what the score product pays for?
VARIANTS["score_two_chains"] = [
    ('                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(x) : "v"(kf[ks % RD]), "v"(qf[ks]));\n            if (ks + RD < KS) rd(ks + RD);',
     '                { if (ks & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(kf[ks % RD]), "v"(qf[ks]));\n'
     '                  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(x) : "v"(kf[ks % RD]), "v"(qf[ks])); }\n            if (ks + RD < KS) rd(ks + RD);')]
for _sd in (2, 4, 5, 6):
    VARIANTS[f"wpv_sd{_sd}"] = [("#define MOMA_K2_WPV_SD 3 ", f"#define MOMA_K2_WPV_SD {_sd} ")]
# ---- round 5, hypothesis test: TWO workgroups per CU = two waves per SIMD for the one-pass kernel at d <= 256
# -- does a second wave hide the first one's issue stream?  Grid doubled (512 workgroups, 8 tiles each).  At d = 256 the body as
# shipped needs ~310 registers; the timing-only DIET (read distance 2, one column tile of transposed reads in flight, no
# row-constant tuple, 4 row-read addresses, no max / overflow tracking: garbage results, the same instruction mix per tile minus
# ~3 VALU) brings it to 256 with no spill inside the loop.  Controls: the same diet at ONE workgroup per CU (LDS padded to
# 96 KiB so that the hardware cannot co-schedule two), on the shipped grid and on the doubled one.
OCC2 = ("template <int D, bool WITH_DQ>\n__global__ __launch_bounds__(256, 1) void infonce_flash_kernel(",
        "template <int D, bool WITH_DQ>\n__global__ __launch_bounds__(256, D <= 256 ? 2 : 1) void infonce_flash_kernel(")
GRID512 = ("    int want = 256 / (p.nbt * nterms);         // ~1 workgroup per CU over all terms",
           "    int want = 512 / (p.nbt * nterms);")
LDSPAD = ("    const size_t lds = (size_t)NBUF * KT * d * 2 + 16;      // ring + the 4 overflow words",
          "    const size_t lds = (size_t)NBUF * KT * d * 2 + 16 + (d <= 256 ? 98304 - (size_t)NBUF * KT * d * 2 : 0);")
DIET = [("#define MOMA_K2_RD 4 ", "#define MOMA_K2_RD 2 "), ("#define MOMA_K2_PF 2\n", "#define MOMA_K2_PF 1\n"),
        ('asm volatile("s_nop 1\\n\\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(x) : "v"(kf[ks % RD]), "v"(qf[ks]), "v"(c0));',
         'asm volatile("s_nop 1\\n\\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(x) : "v"(kf[ks % RD]), "v"(qf[ks]));'),
        ('asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[ks % RD]) : "v"(aa[ks & 7]), "i"((ks >> 3) * 8192) : "memory");',
         'asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[ks % RD]) : "v"(aa[ks & 3]), "i"((ks >> 3) * 8192) : "memory");', 1),
        ("        for (int c = 0; c < 8; ++c) aa[c] = a0 ^ (c << 5);", "        for (int c = 0; c < 4; ++c) aa[c] = a0 ^ (c << 5);", 1),
        ("            ovf |= (tmax > OVERFLOW_THR) ? 1 : 0;\n            mx = fmaxf(mx, tmax + m_ref);                       // back to absolute log2 units\n", "")]
VARIANTS["occ2"] = [OCC2, GRID512]                       # d = 128 fits 256 registers as shipped
VARIANTS["grid512"] = [GRID512, LDSPAD]
VARIANTS["occ1pad"] = [LDSPAD]
VARIANTS["diet_occ2"] = DIET + [OCC2, GRID512]
VARIANTS["diet_occ1"] = DIET + [LDSPAD]
VARIANTS["diet_occ1_grid512"] = DIET + [GRID512, LDSPAD]
# round 5: non-temporal queue stream of the small-batch kernel (each tile is read by exactly one workgroup)
VARIANTS["small_nt"] = [("#define MOMA_K2_SMALL_AUX 0 ", "#define MOMA_K2_SMALL_AUX 2 ")]
# round 5: de-synchronise the four waves of a workgroup after the end-of-iteration barrier (they run the same stream in lockstep and
# meet at the LDS with the same ds_read_b128 of every k-step): wave w idles w x N cycles behind the barrier
for _n in (3, 7, 15):
    VARIANTS[f"wavestagger{_n + 1}"] = [(LOOP_END, LOOP_END.replace(
        "            __builtin_amdgcn_s_barrier();\n",
        "            __builtin_amdgcn_s_barrier();\n"
        f"            if (wave >= 1) asm volatile(\"s_nop {_n}\" ::: \"memory\");\n"
        f"            if (wave >= 2) asm volatile(\"s_nop {_n}\" ::: \"memory\");\n"
        f"            if (wave >= 3) asm volatile(\"s_nop {_n}\" ::: \"memory\");\n"))]
# round 5: column tiles per block of the wide-row combine
for _t in (2, 8, 10):
    VARIANTS[f"wcomb_tpb{_t}"] = [("            int tpb = 4;\n", f"            int tpb = {_t};\n")]
# round 5: wall-clock stamps inside the combine kernel (one lane per workgroup -> a __device__ array read back by
# moma_debug_combine_stamps; scripts/diag_combine_stamps.py prints the timeline).  Diagnostic build only.
VARIANTS["combine_stamps"] = [
    ("namespace moma {\nnamespace {\n\ntypedef __attribute__((ext_vector_type(4))) short s16x4;",
     "namespace moma {\n__device__ unsigned long long g_cmb_stamps[4096 * 8];\nnamespace {\n\ntypedef __attribute__((ext_vector_type(4))) short s16x4;"),
    ("    const int tid = threadIdx.x;\n    const int wb = blockIdx.x >> 2, g = (blockIdx.x >> 1) & 1, h = blockIdx.x & 1;\n",
     "    const int tid = threadIdx.x;\n    const int wb = blockIdx.x >> 2, g = (blockIdx.x >> 1) & 1, h = blockIdx.x & 1;\n"
     "    int stamp_i = 0;\n"
     "    auto stamp = [&]() __attribute__((always_inline)) {\n"
     "        unsigned long long tt;\n"
     "        asm volatile(\"s_waitcnt vmcnt(0) lgkmcnt(0)\\n\\ts_memrealtime %0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(tt) :: \"memory\");\n"
     "        const int wgid = blockIdx.y * gridDim.x + blockIdx.x;\n"
     "        if (tid == 0 && wgid < 4096) g_cmb_stamps[wgid * 8 + stamp_i] = tt;\n"
     "        ++stamp_i;\n"
     "    };\n"
     "    stamp();\n"),
    ("    if (dq == nullptr) return;\n    __syncthreads();                                                     // wts[][], rowc[][] complete\n",
     "    if (dq == nullptr) return;\n    stamp();\n    __syncthreads();                                                     // wts[][], rowc[][] complete\n    stamp();\n"),
    ("#pragma unroll\n    for (int i = 0; i < 8; ++i) accs[cgrp][i][col] = acc[i];\n    __syncthreads();\n    {\n        const int i = cgrp;                                              // 8 rows x 32 columns of output, one per thread",
     "    stamp();\n#pragma unroll\n    for (int i = 0; i < 8; ++i) accs[cgrp][i][col] = acc[i];\n    __syncthreads();\n    {\n        const int i = cgrp;                                              // 8 rows x 32 columns of output, one per thread"),
    ("    __syncthreads();                                                     // accs[] is reused by the next column tile\n",
     "    stamp();\n    __syncthreads();                                                     // accs[] is reused by the next column tile\n"),
    ("}  // namespace moma\n", "}  // namespace moma\nextern \"C\" int moma_debug_combine_stamps(void* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(moma::g_cmb_stamps), sizeof(unsigned long long) * 4096 * 8); }\n"),
]
# round 5: the split-K partials of the one-pass kernel stored write-through (sc1) instead of left dirty in the XCD's L2 for the
# kernel-end write-back that the combine's first loads then wait for (combine timeline: 4.9 us until its row statistics are in)
SC1 = ("                dst[(c * 2 + g) * 64] = v;     // (non-temporal stores: same kernel time, +3 us on the combine that reads them back)",
       "                { typedef unsigned u32x4_t __attribute__((ext_vector_type(4))); const u32x4_t vv = {v.x, v.y, v.z, v.w};\n"
       "                  asm volatile(\"global_store_dwordx4 %0, %1, off sc1\" :: \"v\"(dst + (c * 2 + g) * 64), \"v\"(vv) : \"memory\"); }")
VARIANTS["opart_sc1"] = [SC1]
VARIANTS["opart_sc0sc1"] = [(SC1[0], SC1[1].replace("off sc1", "off sc0 sc1"))]
VARIANTS["opart_sc1_stamps"] = [SC1] + VARIANTS["combine_stamps"]
VARIANTS["combine_stamps2"] = VARIANTS["combine_stamps"] + [
    ("        float s = 0.f;                                                   // positive logit\n",
     "        stamp();\n        float s = 0.f;                                                   // positive logit\n"),
    ("        const float s0 = sum32(s) * inv_T;\n", "        stamp();\n        const float s0 = sum32(s) * inv_T;\n"),
]
names = sys.argv[1:] or list(VARIANTS)
os.makedirs(OUT, exist_ok=True)
text = open(SRC).read()
objs = [os.path.join(OBJ, f) for f in os.listdir(OBJ) if f.endswith(".o") and f != "infonce_fused.o"]
for name in names:
    t = text
    for sub in VARIANTS[name]:
        old, new = sub[0], sub[1]
        first_only = len(sub) > 2                # (old, new, 1): the text also occurs in another kernel; the first is the flash body's
        assert t.count(old) == 1 or (first_only and t.count(old) >= 1), (name, old[:60], t.count(old))
        t = t.replace(old, new, 1)
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "infonce_fused.hip")
        open(src, "w").write(t)
        obj = os.path.join(tmp, "v.o")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
                        f"-I{os.path.dirname(SRC)}", "-c", src, "-o", obj], check=True)
        lib = os.path.join(OUT, f"libmoma_{name}.so")
        subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib, obj, *objs], check=True)
    print("built", name, flush=True)
