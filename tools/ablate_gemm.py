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
