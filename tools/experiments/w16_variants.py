"""Timing-only variants of tdnn_gemm_f16bf8_wide16_kernel (WRONG results on purpose), built into build/variants/libxv_<name>.so and run
through XVECTOR_HIP_LIB by tools/wide_bench.py -- which part of a pair the waves wait for at full clock (all-zero data):
  noax3   J10 does not reload AX for X q3 (the one X job with a single H job in front of it)
  nobh    J12 does not refill BH tile by tile (the refill no X job can cover)
  nobar   no workgroup barriers in the loop
  nodma   no LDS-DMA in the loop
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "x-vector-kaldi-tf_amd", "csrc", "xv_gemm8.hip")
OUT = os.path.join(ROOT, "build", "variants")
src = open(SRC).read()


def variant(name):
    s = src
    if name == "noax3":
        old = "job_h(I0{}, I2{}, ld_axl(I0{}, px, 3), ld_axh(I0{}, px2, 3), ld_axl(I1{}, px, 3), ld_axh(I1{}, px2, 3));"
        assert old in s
        s = s.replace(old, "job_h(I0{}, I2{});")
    elif name == "nobh":
        a = s.index("            job_h(I1{}, I3{}, none, ld_bh(I0{}, 0), none, ld_bh(I1{}, 0)")
        b = s.index("// J12", a)
        s = s[:a] + "            job_h(I1{}, I3{}, none, none, none, none, none, none, ld_ah(I1{}, I0{}, ahn, bn, 1), ld_ah(I1{}, I1{}, ahn, bn, 1)); " + s[b:]
    elif name == "nobar":
        a = s.index("tdnn_gemm_f16bf8_wide16_kernel(const Gemm8Params p)")
        b = s.index("#undef XV_LD", a)
        s = s[:a] + s[a:b].replace("            __builtin_amdgcn_s_barrier();\n", "") + s[b:]
    elif name == "nodma":
        a = s.index("            auto dx = [&](auto I)")
        b = s.index("            job_h(I0{}, I0{}, ld_bxl(I0{})", a)
        s = s[:a] + "            auto dx = [&](auto I) { return [] {}; };\n            auto dh0 = dx, dh1 = dx;\n            auto halo1 = [] {};\n            auto halo2 = [] {};\n" + s[b:]
    else:
        raise SystemExit("unknown variant " + name)
    return s


for name in sys.argv[1:]:
    path = os.path.join(OUT, "xv_gemm8_%s.hip" % name)
    open(path, "w").write(variant(name))
    obj = os.path.join(OUT, "xv_gemm8_%s.o" % name)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.dirname(SRC), "-Wno-unused-function", "-c", "-o", obj, path])
    objs = [os.path.join(ROOT, "build", "obj", f) for f in os.listdir(os.path.join(ROOT, "build", "obj")) if f.endswith(".o") and f != "xv_gemm8.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, "libxv_%s.so" % name), obj] + objs)
    print("built", name)
