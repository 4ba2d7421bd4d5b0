"""N>1 path on CPU: world_size-2 gloo processes run the sharding + single-gather logic of
xvector_amd.dist with a stand-in per-utterance function (the GPU forward itself is covered by -m gpu)."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from conftest import PKG
from xvector_amd import dist as xdist


def test_partition_lpt_is_balanced_and_deterministic():
    lens = np.random.default_rng(0).integers(25, 10001, size=1000)
    for world in (1, 2, 4, 8):
        shards = xdist.partition_lpt(lens, world)
        allidx = np.concatenate(shards)
        assert sorted(allidx) == list(range(1000))
        loads = [int(lens[s].sum()) for s in shards]
        assert max(loads) - min(loads) <= lens.max()
        assert all(np.array_equal(a, b) for a, b in zip(shards, xdist.partition_lpt(lens, world)))
        assert all(np.all(np.diff(s) > 0) for s in shards if len(s) > 1)
    assert [len(s) for s in xdist.partition_lpt([5, 5, 5], 8)] == [1, 1, 1, 0, 0, 0, 0, 0]


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from xvector_amd import dist as xdist
    rank, world = xdist.init_process_group("gloo")
    lens = np.random.default_rng(3).integers(25, 400, size=37)
    dim = 16
    def fake_extract(idx):      # stand-in for the GPU forward: row i = f(utterance i)
        return torch.stack([torch.full((dim,), float(i)) + torch.arange(dim) * float(lens[i]) for i in idx]) if len(idx) else torch.zeros((0, dim))
    out = xdist.sharded_extract(lens, fake_extract, dim, torch.device("cpu"))
    if rank == 0:
        want = torch.stack([torch.full((dim,), float(i)) + torch.arange(dim) * float(lens[i]) for i in range(len(lens))])
        assert torch.equal(out, want), "gathered result is not in input order"
        print("GATHER_OK", world)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()
""")


def test_two_rank_gloo_sharded_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % PKG)
    port = 29000 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHER_OK 2" in outs[0]
