"""Timing-only variants of tdnn_pair_pool_f16bf8_kernel (WRONG results on purpose) -> build/variants/libxv_p8_<name>.so, run through
XVECTOR_HIP_LIB by tools/pair8_bench.py: what the exchange of partial tiles through LDS, the pooling arithmetic, the stage barriers
and the weight DMA cost (the LDS budget of a stage: 512 cycles of fragment reads + ~256 of DMA writes + ~256 of exchange writes +
~128 of pooling reads against 1024 MFMA cycles).
  noexch   the partner's half is neither written nor read back
  nopool   no pooling arithmetic, exchange reads or statistics stores
  noswap   no v_permlane32_swap of the half a wave finishes itself
  noconv   no bias / activation / BatchNorm / split8 encoding between the two GEMMs
  nobar    no stage barriers
  nodma    no weight / frames DMA in the loops
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "x-vector-kaldi-tf_amd", "csrc", "xv_pair8.hip")
OUT = os.path.join(ROOT, "build", "variants")
src = open(SRC).read()


def rep(s, old, new, count=None):
    assert old in s, old
    return s.replace(old, new) if count is None else s.replace(old, new, count)


def variant(name):
    s = src
    if name == "noexch":        # the partner's half neither written nor read back (the KEEP half stays in registers anyway since round 4)
        s = rep(s, "            if constexpr (q == 2) red_write(I0{}, ct);", "            if constexpr (q == 2) { float t0 = yR[0], t1 = yR[9]; asm volatile(\"\" :: \"v\"(t0), \"v\"(t1)); }")
        s = rep(s, "            if constexpr (q == 2) red_write(I1{}, ct);\n", "")
        s = rep(s, "        pn[i] = from[off];\n", "        pn[i] = prm[2] + (float)off;\n")
    elif name == "nopool":      # no pooling arithmetic, exchange reads or statistics stores
        s = rep(s, "            pool_row(std::integral_constant<int, 8 * q + 2 * j>{}, I0{}, ct - 1);\n            pool_row(std::integral_constant<int, 8 * q + 2 * j + 1>{}, I1{}, ct - 1);\n", "")
        s = rep(s, "        pn[i] = from[off];\n", "")
    elif name == "noswap":      # the KEEP half pooled as it lies (wrong rows): what the eight v_permlane32_swap cost
        s = rep(s, '            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));\n', "")
    elif name == "nobar":
        s = rep(s, '        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");\n        __builtin_amdgcn_s_barrier();\n', '        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");\n')
    elif name == "nodma":
        s = rep(s, "        XV_BLDS16(wrs, lds + fill + wave * 4096, wvoff, wsoff, k * 1024);\n", "        if (wleft < 0) XV_BLDS16(wrs, lds + fill + wave * 4096, wvoff, wsoff, k * 1024);\n")
    elif name == "noconv":      # H = the raw accumulators' bits: no bias / activation / BN / split8 encoding between the GEMMs
        s = rep(s, "                xv_split8_encode8<true>(v, hi, x8[g], amax);\n", "                hi = __builtin_bit_cast(xv_f16x8, (f32x4){v[0], v[1], v[2], v[3]}); x8[g] = __builtin_bit_cast(xv_i32x4, (f32x4){v[4], v[5], v[6], v[7]});\n")
        s = rep(s, "                    for (int e = 0; e < 4; ++e) v[4 * j + e] = act_fn<MODE>(t[8 * g + 4 * j + e] + b[e], a[e]) * s[e] + o[e];\n", "                    for (int e = 0; e < 4; ++e) v[4 * j + e] = t[8 * g + 4 * j + e];\n")
    else:
        raise SystemExit("unknown variant " + name)
    return s


os.makedirs(OUT, exist_ok=True)
for name in sys.argv[1:]:
    path = os.path.join(OUT, "xv_pair8_%s.hip" % name)
    open(path, "w").write(variant(name))
    obj = os.path.join(OUT, "xv_pair8_%s.o" % name)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.dirname(SRC), "-Wno-unused-function", "-Wno-unused-variable", "-c", "-o", obj, path])
    objs = [os.path.join(ROOT, "build", "obj", f) for f in os.listdir(os.path.join(ROOT, "build", "obj")) if f.endswith(".o") and f != "xv_pair8.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, "libxv_p8_%s.so" % name), obj] + objs)
    print("built", name)
