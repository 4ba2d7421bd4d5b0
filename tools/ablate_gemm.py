"""Development aid: build timing-only ablations of the bf16x3 GEMM main loop (their RESULTS ARE WRONG by construction)
to see which resource bounds it.  Variants -> build/ablate/lib_<name>.so; time them with
XVECTOR_HIP_LIB=build/ablate/lib_<name>.so python tools/layer_bench.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "x-vector-kaldi-tf_amd", "csrc")
OUT = os.path.join(ROOT, "build", "ablate")
os.makedirs(OUT, exist_ok=True)
base = open(os.path.join(SRC, "xv_kernels.hip")).read()
A_LINE = "XV_GLDS16(ag + (size_t)(piece * 8) * xrow_bytes, adst + piece * 1024);"
B_LINE = "for (int j = 0; j < 4; ++j) XV_GLDS16(bnext + j * 1024, dst + j * 1024);"
AG_LINE = "const uint8_t *ag = abase + (size_t)ca * SROW;"
BN_LINE = "bnext += (s + 3 < n_stages) ? B3_BYTES : 0;"
assert all(l in base for l in (A_LINE, B_LINE, AG_LINE, BN_LINE))
variants = {
    "base": base,
    "noA": base.replace(A_LINE, ";"),                                 # in-loop A DMA removed (LDS keeps stale data)
    "noB": base.replace(B_LINE, "for (int j = 0; j < 4; ++j) ;"),
    "noAB": base.replace(A_LINE, ";").replace(B_LINE, "for (int j = 0; j < 4; ++j) ;"),
    "hotA": base.replace(AG_LINE, "const uint8_t *ag = abase;"),      # A always from slab 0 of the tile (cache-hot)
    "hotB": base.replace(BN_LINE, ";"),                                # B always the same 16 KB tile (cache-hot)
    "hotAB": base.replace(AG_LINE, "const uint8_t *ag = abase;").replace(BN_LINE, ";"),
}
# per-workgroup timeline: p.ypre is re-used as a trace buffer [n_wg][8] of int64 (wall clock 100 MHz, HW_ID)
def traced(text):
    i0 = text.index("void tdnn_gemm_bf16x3_kernel(const Gemm3Params p)")
    i1 = text.index("int launch_gemm3") + len("int launch_gemm3")
    return text[:i0] + _traced(text[i0:i1]) + text[i1:]


def _traced(text):
    t = text.replace("if (p.ypre) {", "if (false) {")
    assert t.count("if (p.y && p.y_split && !p.ypre && n0 + BN <= p.cout) {") == 1
    t = t.replace("if (p.y && p.y_split && !p.ypre && n0 + BN <= p.cout) {", "if (p.y && p.y_split && n0 + BN <= p.cout) {")
    a0 = "    if (tid < BM) {\n        const long gr = m0 + tid;\n        Ms[tid] ="
    assert t.count(a0) == 1
    t = t.replace(a0, "    long long *trc = reinterpret_cast<long long *>(p.ypre) + (size_t)blockIdx.x * 8;  /*clk*/\n"
                      "    if (tid == 0) { trc[0] = wall_clock64(); trc[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); trc[5] = (long long)wg | ((long long)(__builtin_readcyclecounter() & 0xffffffffffll) << 24); trc[6] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)); }\n" + a0)
    a1 = "    Frags F = {}, G = {};\n    load_frags(F, 0, 0, 0, 0);"
    assert t.count(a1) == 1
    t = t.replace(a1, "    if (tid == 0) trc[1] = wall_clock64();\n" + a1)
    a2 = "    float *T = reinterpret_cast<float *>(lds);\n    {\n        const int col = wc * 64 + (lane & 31);"
    assert t.count(a2) == 1
    t = t.replace(a2, "    if (tid == 0) trc[2] = wall_clock64();\n" + a2)
    # end of kernel: the non-POOL epilogue loop closes with "        }\n    }\n}\n\nint launch_gemm3"
    a3 = "        }\n    }\n}\n\nint launch_gemm3"
    assert t.count(a3) == 1
    t = t.replace(a3, "        }\n    }\n    __builtin_amdgcn_s_waitcnt(0);\n    if (tid == 0) trc[3] = wall_clock64();\n}\n\nint launch_gemm3")
    b1 = "    const int cg = tid & 15;                            // 8-channel group of the 128-column tile"
    assert t.count(b1) == 1
    t = t.replace(b1, "    if (tid == 0) trc[7] = wall_clock64();\n" + b1)
    f1 = "        else rows(std::false_type{});\n        return;"
    assert t.count(f1) == 1
    t = t.replace(f1, "        else rows(std::false_type{});\n        if (tid == 0) trc[6] = (trc[6] & 0xf) | (wall_clock64() << 8);\n"
                      "        __builtin_amdgcn_s_waitcnt(0);\n        if (tid == 0) { trc[3] = wall_clock64(); trc[4] = (trc[4] & 0xffffff) | ((long long)(__builtin_readcyclecounter() & 0xffffffffffll) << 24); }\n        return;")
    t = t.replace("    __builtin_amdgcn_s_waitcnt(0);\n    if (tid == 0) trc[3] = wall_clock64();", "    if (tid == 0) trc[6] = (trc[6] & 0xf) | (wall_clock64() << 8);\n    __builtin_amdgcn_s_waitcnt(0);\n    if (tid == 0) trc[3] = wall_clock64();")
    return t
variants["trace"] = traced(base)
variants["trace_noAB"] = traced(variants["noAB"])
variants["trace_hotAB"] = traced(variants["hotAB"])
ST1 = "\n                __builtin_nontemporal_store(hi, reinterpret_cast<bf16x8 *>(row + ((slot ^ sw) << 4)));"
ST2 = "\n                __builtin_nontemporal_store(lo, reinterpret_cast<bf16x8 *>(row + (((4 + slot) ^ sw) << 4)));"
assert base.count(ST1) == 1 and base.count(ST2) == 1
variants["trace_nostore"] = traced(base.replace(ST1, "\n                if (Ms[lr] == 77) {" + ST1).replace(ST2, ST2 + "\n                }"))
variants["trace_plainstore"] = traced(base.replace(ST1, "\n                *reinterpret_cast<bf16x8 *>(row + ((slot ^ sw) << 4)) = hi;")
                                      .replace(ST2, "\n                *reinterpret_cast<bf16x8 *>(row + (((4 + slot) ^ sw) << 4)) = lo;"))
W = '            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n            __syncthreads();                  // B(s): stage s fully read by everybody, stage s+1 landed'
assert base.count(W) == 1
variants["nowait"] = base.replace(W, "            __syncthreads();")
variants["nobarrier"] = base.replace(W, '            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");')
variants["nowait_nobarrier"] = base.replace(W, "            ;")
H = "        __builtin_amdgcn_sched_barrier(0);\n        auto rows = [&](auto LRELU) {"
assert base.count(H) == 2
variants["nohoist"] = base.replace(H, "        auto rows = [&](auto LRELU) {")
RB = ["        F.bh0 = *reinterpret_cast<const bf16x8 *>(Bb + boff0 + ((t ^ bsw0) << 4));",
      "        F.bh1 = *reinterpret_cast<const bf16x8 *>(Bb + boff1 + ((t ^ bsw1) << 4));",
      "        F.bl0 = *reinterpret_cast<const bf16x8 *>(Bb + B3_PLANE + boff0 + ((t ^ bsw0) << 4));",
      "        F.bl1 = *reinterpret_cast<const bf16x8 *>(Bb + B3_PLANE + boff1 + ((t ^ bsw1) << 4));"]
RA = ["        F.al0 = *reinterpret_cast<const bf16x8 *>(a0 + (((t + 4) ^ sw0) << 4));",
      "        F.al1 = *reinterpret_cast<const bf16x8 *>(a1 + (((t + 4) ^ sw1) << 4));",
      "        F.ah0 = *reinterpret_cast<const bf16x8 *>(a0 + ((t ^ sw0) << 4));",
      "        F.ah1 = *reinterpret_cast<const bf16x8 *>(a1 + ((t ^ sw1) << 4));"]
t = base
for l in RB:
    assert t.count(l) == 1
    t = t.replace(l, "        if (ks == 0) {" + l.strip() + " }")
variants["halfBreads"] = t.replace("            load_frags(G, s, c0, t0, 1);\n            mma(F);", "            load_frags(G, s, c0, t0, 1); G.bh0 = F.bh0; G.bh1 = F.bh1; G.bl0 = F.bl0; G.bl1 = F.bl1;\n            mma(F);")
t2 = t
for l in RA:
    assert t2.count(l) == 1
    t2 = t2.replace(l, "        if (ks == 0) {" + l.strip() + " }")
variants["halfABreads"] = t2.replace("            load_frags(G, s, c0, t0, 1);\n            mma(F);", "            G = F;\n            mma(F);")
variants["halfB"] = base.replace(B_LINE, "if (s & 1) for (int j = 0; j < 4; ++j) XV_GLDS16(bnext + j * 1024, dst + j * 1024);")
variants["immB"] = base.replace(B_LINE, "for (int once = 0; once < 1; ++once) { __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(bnext), (__attribute__((address_space(3))) void *)(dst), 16, 0, 0); __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(bnext), (__attribute__((address_space(3))) void *)(dst), 16, 1024, 0); __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(bnext), (__attribute__((address_space(3))) void *)(dst), 16, 2048, 0); __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(bnext), (__attribute__((address_space(3))) void *)(dst), 16, 3072, 0); }")
procs = []
for name, text in variants.items():
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    f = os.path.join(OUT, "xv_kernels_%s.hip" % name)
    open(f, "w").write(text)
    procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                                   "-I" + os.path.join(ROOT, "include"), "-I" + SRC, "-Wno-unused-function",
                                   "-o", os.path.join(OUT, "lib_%s.so" % name), f, os.path.join(SRC, "xv_train.hip"), os.path.join(SRC, "xv_frontend.hip")]))
assert all(p.wait() == 0 for p in procs)
print("built:", sorted(os.listdir(OUT)))
