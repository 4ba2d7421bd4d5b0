"""Development aid: timing-only ablations / scheduling variants of the Toom-Cook fp32 kernel (csrc/xv_toom.hip).  Ablations give
WRONG RESULTS by construction.  Variants -> build/toom/lib_<name>.so (xv_toom.o replaced, the other objects as built by make);
`python tools/experiments/toom_variants.py run` times each with tools/toom_layer_bench.py (on the GPU box)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "x-vector-kaldi-tf_amd", "csrc")
OBJ = os.path.join(ROOT, "build", "obj")
OUT = os.path.join(ROOT, "build", "toom")
base = open(os.path.join(SRC, "xv_toom.hip")).read()


def rep(text, a, b, count=1):
    assert text.count(a) == count, (a, text.count(a))
    return text.replace(a, b)


def variants():
    v = {"base": base}
    # the transform gone: V_j = the first raw fragment the row reads
    v["noxform"] = rep(base, "                    for (int e = 0; e < 4; ++e) r[e] = __builtin_fmaf(bt, raw[i][e], r[e]);", "                    for (int e = 0; e < 4; ++e) ;")
    # no in-loop DMA at all (stale LDS): is the loop waiting for data?
    t = rep(base, "            dma_b(s + 2, s & 1);\n            if constexpr (j < 5) dma_a_slot(c + 1, j);", "            ;")
    v["nodma"] = t
    v["nobarrier"] = rep(base, "            __builtin_amdgcn_s_barrier();\n            dma_b(s + 2", "            dma_b(s + 2")
    return v


def build():
    os.makedirs(OUT, exist_ok=True)
    others = [os.path.join(OBJ, f) for f in sorted(os.listdir(OBJ)) if f.endswith(".o") and f != "xv_toom.o"]
    procs = []
    for name, text in variants().items():
        src = os.path.join(OUT, "xv_toom_%s.hip" % name)
        if not os.path.exists(src) or open(src).read() != text:
            open(src, "w").write(text)
        obj = os.path.join(OUT, "xv_toom_%s.o" % name)
        so = os.path.join(OUT, "lib_%s.so" % name)
        cmd = ("/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I%s/include -I%s -c -o %s %s && "
               "/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o %s %s %s" % (ROOT, SRC, obj, src, so, obj, " ".join(others)))
        procs.append((name, subprocess.Popen(cmd, shell=True)))
    for name, p in procs:
        assert p.wait() == 0, name


def run():
    names = sys.argv[2:] or sorted(f[4:-3] for f in os.listdir(OUT) if f.startswith("lib_") and f.endswith(".so"))
    for name in names:
        env = dict(os.environ, XVECTOR_HIP_LIB=os.path.join(OUT, "lib_%s.so" % name))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "toom_layer_bench.py")], env=env, capture_output=True, text=True)
        for line in out.stdout.splitlines():
            if line.startswith("K="):
                print("%-12s %s" % (name, line), flush=True)
        if out.returncode:
            print(name, "FAILED", out.stderr[-500:])


if __name__ == "__main__":
    run() if len(sys.argv) > 1 and sys.argv[1] == "run" else build()
