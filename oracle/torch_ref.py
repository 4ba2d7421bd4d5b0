"""torch-CPU fp32 port of the reference's extraction forward -- the reported CPU BASELINE only.

TEST/BENCH INFRASTRUCTURE: imported by bench.py's ``cpu_baseline`` leg (and tests); never by the product.
TensorFlow is absent here and on the GPU box, so "the reference's CPU path" is represented by this port
of local/tf/models.py:50-94 + tf_block.py:25-26 on torch's oneDNN/MKL CPU kernels, run the way the
reference runs: batch 1, one forward per <=chunk_size chunk (models.py:401-414).
"""
import time

import numpy as np
import torch
import torch.nn.functional as F

BN_EPSILON = 1e-3
VAR2STD_EPSILON = 1e-5


class TorchCpuModel(object):
    def __init__(self, weights, topo):
        self.topo = topo
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32))
        self.layers = []
        for i, (K, d) in enumerate(zip(topo["kernel_sizes"], topo["dilations"])):
            sc = "frame_level_info_layer-%d" % i
            s = weights[sc + "/gamma:0"] / np.sqrt(weights[sc + "/variance:0"].astype(np.float64) + BN_EPSILON)
            self.layers.append(dict(w=t(weights[sc + "/w:0"]).permute(2, 1, 0).contiguous(), b=t(weights[sc + "/b:0"]),
                                    scale=t(s).view(1, -1, 1),
                                    shift=t(weights[sc + "/beta:0"] - weights[sc + "/mean:0"] * s).view(1, -1, 1),
                                    pad=(K - 1) * d // 2, dil=d))
        self.w0 = t(weights["embed_layer-0/w:0"])
        self.b0 = t(weights["embed_layer-0/b:0"])
        assert topo.get("activation", "relu") == "relu"

    @torch.no_grad()
    def forward(self, x):
        h = torch.as_tensor(x).T.unsqueeze(0)
        for L in self.layers:
            h = F.relu(F.conv1d(h, L["w"], L["b"], padding=L["pad"], dilation=L["dil"])) * L["scale"] + L["shift"]
        mu = h.mean(dim=2)
        var = h.var(dim=2, unbiased=False)
        pooled = torch.cat([mu, torch.sqrt(var + VAR2STD_EPSILON)], dim=1)
        return (pooled @ self.w0 + self.b0)[0].numpy()


def time_baseline(weights, topo, mats, threads, budget_s=12.0):
    """Run batch-1 forwards over ``mats`` (cycling) for ~budget_s seconds with ``threads`` torch threads.
    Returns (utterances/s, utterances done, frames done)."""
    prev = torch.get_num_threads()
    torch.set_num_threads(int(threads))
    try:
        m = TorchCpuModel(weights, topo)
        m.forward(mats[0])                       # warm-up (oneDNN primitive creation)
        t0 = time.time()
        n = frames = 0
        while True:
            x = mats[n % len(mats)]
            m.forward(x)
            n += 1
            frames += x.shape[0]
            if time.time() - t0 >= budget_s or n >= 100000:
                break
        dt = time.time() - t0
        return n / dt, n, frames
    finally:
        torch.set_num_threads(prev)


def effective_cores():
    """Host cores this process may actually use: the scheduler affinity mask capped by the cgroup CPU quota (a container can
    show 256 logical cores and be granted 16)."""
    import math
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(math.ceil(int(txt[0]) / float(txt[1])))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, int(math.ceil(quota / float(period)))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def time_baseline_processes(weights, topo, mats, processes, threads, budget_s):
    """The reference's DEPLOYMENT shape (run.sh:229-247 -> extract_xvectors.sh:83-88): ``processes`` independent extractor
    jobs, each a batch-1 forward loop with ``threads`` intra-op threads (models.py:361-363), all running at once on this
    host.  Returns (aggregate utterances/s, per-process rates)."""
    import json
    import os
    import subprocess
    import sys
    import tempfile
    tmp = tempfile.mkdtemp(prefix="xv_cpu_base_")
    try:
        np.savez(os.path.join(tmp, "w.npz"), **{k: np.asarray(v) for k, v in weights.items()})
        np.savez(os.path.join(tmp, "m.npz"), *mats)
        json.dump(topo, open(os.path.join(tmp, "topo.json"), "wt"))
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
        cmd = [sys.executable, os.path.abspath(__file__), "--worker", tmp, str(threads), str(budget_s)]
        procs = [subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for _ in range(processes)]
        rates = []
        for p in procs:
            out = p.communicate(timeout=budget_s * 4 + 180)[0].decode().split()
            if p.returncode == 0 and out:
                rates.append(float(out[-1]))
        return sum(rates), rates
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    import json
    import os
    import sys
    if len(sys.argv) == 5 and sys.argv[1] == "--worker":
        d, threads, budget = sys.argv[2], int(sys.argv[3]), float(sys.argv[4])
        with np.load(os.path.join(d, "w.npz")) as z:
            weights = {k: z[k] for k in z.files}
        with np.load(os.path.join(d, "m.npz")) as z:
            mats = [z[k] for k in z.files]
        topo = json.load(open(os.path.join(d, "topo.json")))
        print(time_baseline(weights, topo, mats, threads, budget)[0])
