"""Development aid: build timing-only ablations of the bf16x3 GEMM main loop (their RESULTS ARE WRONG by construction)
to see which resource bounds it.  Variants -> build/ablate/lib_<name>.so; time them with
XVECTOR_HIP_LIB=build/ablate/lib_<name>.so python tools/layer_bench.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "x-vector-kaldi-tf_amd", "csrc")
OUT = os.path.join(ROOT, "build", "ablate")
os.makedirs(OUT, exist_ok=True)
base = open(os.path.join(SRC, "xv_kernels.hip")).read()
A_LINES = ["XV_GLDS16(ag + ag_off[0][j], adst + al_off[0][j]);", "XV_GLDS16(anext + ag_off[t][j], adst_n + al_off[t][j]);"]
B_LINES = ["XV_GLDS16_OFF(bnext, dst, %d);" % o for o in (0, 1024, 2048, 3072)]
assert all(base.count(l) == 1 for l in A_LINES + B_LINES)


def drop(text, lines):
    for l in lines:
        text = text.replace(l, ";")
    return text


G_LOAD = "                load_a_frags(G, pa[t] + abuf, 1);\n                load_b_frags(G, pb[1] + bbuf);"
W = '                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n                __syncthreads();                  // B(s): stage s fully read by everybody, stage s+1 landed'
assert base.count(G_LOAD) == 1 and base.count(W) == 1
variants = {
    "base": base,
    "noA": drop(base, A_LINES),                      # in-loop A DMA removed (LDS keeps stale data)
    "noB": drop(base, B_LINES),
    "noAB": drop(base, A_LINES + B_LINES),
    "halfreads": base.replace(G_LOAD, "                G = F;"),      # k-step 1 re-uses the fragments of k-step 0: half the ds_reads
    "nowait": base.replace(W, "                __syncthreads();"),
    "nobarrier": base.replace(W, '                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");'),
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
    f1 = "        else run(std::integral_constant<int, 0>{});\n        return;"
    assert t.count(f1) == 1
    t = t.replace(f1, "        else run(std::integral_constant<int, 0>{});\n        if (tid == 0) trc[6] = (trc[6] & 0xf) | (wall_clock64() << 8);\n"
                      "        __builtin_amdgcn_s_waitcnt(0);\n        if (tid == 0) { trc[3] = wall_clock64(); trc[4] = (trc[4] & 0xffffff) | ((long long)(__builtin_readcyclecounter() & 0xffffffffffll) << 24); }\n        return;")
    t = t.replace("    __builtin_amdgcn_s_waitcnt(0);\n    if (tid == 0) trc[3] = wall_clock64();", "    if (tid == 0) trc[6] = (trc[6] & 0xf) | (wall_clock64() << 8);\n    __builtin_amdgcn_s_waitcnt(0);\n    if (tid == 0) trc[3] = wall_clock64();")
    return t
variants["trace"] = traced(base)
MMA_OLD = base[base.index("    auto mma = [&](const Frags &F) {"):base.index("    // Register-level software pipeline at k-step granularity")]
def _m(a, b, acc):
    return "        %s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.%s, F.%s, %s, 0, 0, 0);\n" % (acc, a, b, acc)
snake = [("al0", "bh0", "acc00"), ("al0", "bh1", "acc01"), ("al1", "bh1", "acc11"), ("al1", "bh0", "acc10"),
         ("ah0", "bh0", "acc00"), ("ah0", "bh1", "acc01"), ("ah1", "bh1", "acc11"), ("ah1", "bh0", "acc10"),
         ("ah0", "bl0", "acc00"), ("ah0", "bl1", "acc01"), ("ah1", "bl1", "acc11"), ("ah1", "bl0", "acc10")]
variants["snake"] = base.replace(MMA_OLD, "    auto mma = [&](const Frags &F) {\n" + "".join(_m(*t) for t in snake) + "    };\n\n")
# K=1 only: half the A pieces (odd stages keep stale data) / half the B pieces -- what a two-tile workgroup would save
variants["k1halfA"] = base.replace(A_LINES[0], "if (s & 1) " + A_LINES[0])
variants["k1halfB"] = base.replace(B_LINES[2], ";").replace(B_LINES[3], ";")
EPI = "    float *T = reinterpret_cast<float *>(lds);\n    {\n        const int col = wc * 64 + (lane & 31);"
assert base.count(EPI) == 1
variants["epiprio"] = base.replace(EPI, "    __builtin_amdgcn_s_setprio(3);\n" + EPI)
variants["trace_epiprio"] = traced(variants["epiprio"])

procs = []
for name, text in variants.items():
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    f = os.path.join(OUT, "xv_kernels_%s.hip" % name)
    open(f, "w").write(text)
    procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                                   "-I" + os.path.join(ROOT, "include"), "-I" + SRC, "-Wno-unused-function",
                                   "-o", os.path.join(OUT, "lib_%s.so" % name), f, os.path.join(SRC, "xv_train.hip"), os.path.join(SRC, "xv_frontend.hip"), os.path.join(SRC, "xv_attention.hip")]))
assert all(p.wait() == 0 for p in procs)
print("built:", sorted(os.listdir(OUT)))
