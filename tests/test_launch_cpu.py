"""xvector_amd/launch.py -- the package's own rank launcher (one process per GPU, the RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* contract of torch.distributed.run) and bench.py's self-launch for ``--gpus N``.  CPU only: the
ranks talk over gloo."""
import os
import subprocess
import sys
import textwrap

from conftest import PKG, ROOT

ENV = dict(os.environ, PYTHONPATH=os.pathsep.join([PKG, os.environ.get("PYTHONPATH", "")]))


def test_ranks_get_the_torchrun_contract_and_a_working_group(tmp_path):
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent('''
        import os, sys
        import torch, torch.distributed as dist
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["MASTER_ADDR"] == "127.0.0.1"
        assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and int(os.environ["MASTER_PORT"]) > 0
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        open(os.path.join(sys.argv[1], "rank%d.txt" % rank), "w").write("%d %d %s" % (rank, world, t.item()))
        dist.destroy_process_group()
    '''))
    run = subprocess.run([sys.executable, "-m", "xvector_amd.launch", "--nproc", "3", str(script), str(tmp_path)], env=ENV,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert run.returncode == 0, run.stdout.decode()
    for r in range(3):
        assert (tmp_path / ("rank%d.txt" % r)).read_text() == "%d 3 6.0" % r


def test_a_failing_rank_ends_the_job_with_its_exit_code(tmp_path):
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent('''
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(120)                       # the launcher must not wait for this
    '''))
    from xvector_amd import launch
    import time
    t0 = time.time()
    rc = launch.spawn_ranks([sys.executable, str(script)], 2, warm=False)
    assert rc == 7 and time.time() - t0 < 60


def test_relaunch_only_outside_a_launcher(monkeypatch):
    from xvector_amd import launch
    monkeypatch.delenv("RANK", raising=False)
    assert launch.relaunch_self_as_ranks(1) is None                     # one GPU: run in place
    monkeypatch.setenv("RANK", "0")
    assert launch.relaunch_self_as_ranks(8) is None                     # already a rank (torch.distributed.run or launch.py)
    seen = []
    monkeypatch.delenv("RANK")
    monkeypatch.setattr(launch, "spawn_ranks", lambda argv, n: seen.append((argv, n)) or 0)
    assert launch.relaunch_self_as_ranks(4) == 0 and seen[0][1] == 4 and seen[0][0][0] == sys.executable


def test_bench_py_gpus_2_starts_two_ranks_and_fails_on_the_gpu_count():
    """`python bench.py --gpus 2` is the form the driver uses at N = 1: it must start its ranks itself.  This box has no GPU, so
    every rank ends with "2 GPUs requested, 0 visible" -- AFTER having been spawned -- and the job's exit code is non-zero."""
    env = dict(ENV)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    log = run.stdout.decode()
    assert run.returncode != 0, log
    assert "2 GPUs requested, 0 visible" in log and "launch: rank" in log, log
