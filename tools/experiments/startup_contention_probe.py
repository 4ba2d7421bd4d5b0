"""What N worker processes starting at once on ONE GPU wait for (the 8 x 125 k rehearsal's 6-7 s): each of N processes times, in order,
`import torch`, the HIP context, 1.5 GB of device memory, 400 MB of pinned host memory in the worker's own pieces (5 x 64 MB arenas +
3 x 25 MB staging sets), the load of libxvector_hip.so's code objects (first launch), a second allocation round.
    python tools/experiments/startup_contention_probe.py [N ...]      (default: 1 8)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    t = [time.time()]
    import torch
    t.append(time.time())
    torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
    t.append(time.time())
    dev = [torch.empty(384 * 1024 * 1024 // 4, device="cuda:0") for _ in range(4)]; torch.cuda.synchronize()
    t.append(time.time())
    pins = [torch.empty(64 * 1024 * 1024, dtype=torch.uint8).pin_memory() for _ in range(5)] + \
           [torch.empty(25 * 1024 * 1024, dtype=torch.uint8).pin_memory() for _ in range(3)]
    t.append(time.time())
    sys.path[:0] = [os.path.join(ROOT, "x-vector-kaldi-tf_amd")]
    from xvector_amd import hiplib
    hiplib.require_gpu()
    x = torch.randn(64, 512, device="cuda:0")
    hiplib.fold_bn(torch.ones(512, device="cuda:0"), torch.zeros(512, device="cuda:0"), torch.zeros(512, device="cuda:0"),
                   torch.ones(512, device="cuda:0"), 1e-3)
    torch.cuda.synchronize()
    t.append(time.time())
    more = [torch.empty(256 * 1024 * 1024 // 4, device="cuda:0") for _ in range(4)]; torch.cuda.synchronize()
    t.append(time.time())
    names = ["import torch", "hip context", "1.5 GB device", "400 MB pinned", "code objects + first launch", "1 GB device more"]
    print("rank %s: " % os.environ.get("R", "?") + ", ".join("%s %.2f" % (n, b - a) for n, a, b in zip(names, t, t[1:])) + ", total %.2f" % (t[-1] - t[0]))
    sys.stdout.flush()
    os._exit(0)

for n in [int(a) for a in sys.argv[1:]] or [1, 8]:
    t0 = time.time()
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, R=str(r))) for r in range(n)]
    for p in ps:
        p.wait()
    print("== %d processes at once: %.2f s wall" % (n, time.time() - t0))
    sys.stdout.flush()
