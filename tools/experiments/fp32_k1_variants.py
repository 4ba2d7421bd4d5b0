"""Timing-only ablations of the exact-fp32 DMA-fed GEMM on the K = 1 layer shapes (WRONG RESULTS by construction): what do the
epilogue and the prologue of a 16-stage tile cost?  Variants -> build/k1/lib_<name>.so; `run` times them with tools/fp32_layer_bench.py."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "x-vector-kaldi-tf_amd", "csrc")
OBJ = os.path.join(ROOT, "build", "obj")
OUT = os.path.join(ROOT, "build", "k1")
base = open(os.path.join(SRC, "xv_kernels.hip")).read()


def rep(text, a, b):
    assert text.count(a) == 1, (a, text.count(a))
    return text.replace(a, b)


EPI = "    gemm_epilogue_rows<BM>(p, reinterpret_cast<float *>(flds), Ms, m0, n0, wr, wc, tid, acc00, acc01, acc10, acc11);\n}\n\nstd::atomic<int> g_fp32_form{0};"
variants = {
    "base": base,
    # no epilogue at all: one store per thread keeps the accumulators alive
    "noepi": rep(base, EPI, "    if (acc00[0] + acc01[1] + acc10[2] + acc11[3] == 12345.f) p.y[tid] = 1.f;\n}\n\nstd::atomic<int> g_fp32_form{0};"),
    # the prologue does not wait for its DMA (stale LDS): the first-fetch latency of a tile
    "noprowait": rep(base, "    if (KT == 1) dma_a_all(1);\n    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n    __syncthreads();", "    if (KT == 1) dma_a_all(1);\n    __syncthreads();"),
    # the epilogue's arithmetic and LDS traffic without its global stores (a store that never happens)
    "nostore": rep(base, "            __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p.y + (size_t)gr * p.ldy + gc));   // streamed once: no L2 write-allocate",
                   "            if (v[0] == 12345.f) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p.y + (size_t)gr * p.ldy + gc));"),
    # ordinary stores instead of nontemporal ones
    "plainstore": rep(base, "            __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p.y + (size_t)gr * p.ldy + gc));   // streamed once: no L2 write-allocate",
                      "            *reinterpret_cast<f32x4 *>(p.y + (size_t)gr * p.ldy + gc) = v;"),
    # the stores without the arithmetic between the LDS tile and them
    "nomath": rep(base, "                const float t = act_t<ACT>(z[i], al[i]) * sc[i] + sh[i];\n                v[i] = keep ? t : 0.f;",
                  "                v[i] = z[i];"),
}


def build():
    os.makedirs(OUT, exist_ok=True)
    others = [os.path.join(OBJ, f) for f in sorted(os.listdir(OBJ)) if f.endswith(".o") and f != "xv_kernels.o"]
    procs = []
    for name, text in variants.items():
        src = os.path.join(OUT, "xv_kernels_%s.hip" % name)
        open(src, "w").write(text)
        obj, so = os.path.join(OUT, "xv_kernels_%s.o" % name), os.path.join(OUT, "lib_%s.so" % name)
        cmd = ("/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -I%s/include -I%s -c -o %s %s && "
               "/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o %s %s %s" % (ROOT, SRC, obj, src, so, obj, " ".join(others)))
        procs.append((name, subprocess.Popen(cmd, shell=True)))
    for name, p in procs:
        assert p.wait() == 0, name


def run():
    for name in variants:
        env = dict(os.environ, XVECTOR_HIP_LIB=os.path.join(OUT, "lib_%s.so" % name))
        for layer in ("0", "3", "4"):
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fp32_layer_bench.py"), "262144", layer], env=env, capture_output=True, text=True)
            for line in out.stdout.splitlines():
                if "K=" in line and "->" in line:
                    print("%-10s %s" % (name, line.split("(output")[0]), flush=True)


if __name__ == "__main__":
    run() if len(sys.argv) > 1 and sys.argv[1] == "run" else build()
