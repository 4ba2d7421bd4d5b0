"""One process per GPU of this node, started directly.

The reference spreads extraction over ``nj`` independent jobs with ``$cmd JOB=1:$nj`` and concatenates their outputs
(local/tf/extract_xvectors.sh:83-95).  Here the jobs are the ranks of one ``torch.distributed`` group (RCCL over xGMI): this
module starts them -- the same environment contract as ``torch.distributed.run`` (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR,
MASTER_PORT), without its elastic agent, rendezvous store and per-rank log plumbing, which cost seconds of start-up that a
job-level number (utterances per second of wall clock) has no use for.

    python -m xvector_amd.launch --nproc 8 local/tf/extract_embedding.py --feature-rspecifier scp:feats.scp ...
    python bench.py --gpus 8          # bench.py calls spawn_ranks() on itself when it is not already a rank

Semantics: the ranks inherit stdin/stdout/stderr; the first rank that exits non-zero ends the job -- the others get SIGTERM
(then SIGKILL) by PID -- and its exit code is returned; otherwise 0 when all have finished.
"""
import os
import signal
import socket
import subprocess
import sys
import time


def free_port():
    """A TCP port that was free a moment ago on 127.0.0.1 (the group's rendezvous address)."""
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    try:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]
    finally:
        s.close()


def rank_env(rank, nproc, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(nproc), LOCAL_WORLD_SIZE=str(nproc),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # dmabuf IPC: what RCCL needs on this driver
    # one token per job, the same in every rank: what XVECTOR_SHARD_OUTPUT=files names its parts by (extract_embedding.py)
    env.setdefault("XVECTOR_JOB_TOKEN", "%s-%d" % (port, os.getpid()))
    return env


def warm_collective_library():
    """Read torch's librccl.so (0.57 GB, most of it device code for every architecture) through the page cache once, from THIS
    otherwise idle process: the first communicator of a rank loads its kernels from that file under the HIP runtime's lock --
    3.5 s on a cold cache against 1.0 s warm, during which the rank's own kernel launches wait."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "librccl.so")
        fd = os.open(path, os.O_RDONLY)
    except Exception:
        return
    try:
        if hasattr(os, "posix_fadvise"):
            os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_SEQUENTIAL)
        while os.read(fd, 8 << 20):
            pass
    except OSError:
        pass
    finally:
        os.close(fd)


def spawn_ranks(argv, nproc, env=None, port=None, poll=0.05, grace=5.0, warm=True):
    """Run ``argv`` (a full command line, e.g. ``[sys.executable, "bench.py", "--gpus", "8"]``) as ``nproc`` ranks and wait.
    Returns the job's exit code."""
    nproc = int(nproc)
    assert nproc >= 1
    port = free_port() if port is None else int(port)
    procs = [subprocess.Popen(list(argv), env=rank_env(r, nproc, port, env)) for r in range(nproc)]
    if warm:
        import threading
        threading.Thread(target=warm_collective_library, daemon=True).start()
    rc = 0
    try:
        alive = set(range(nproc))
        while alive and rc == 0:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if code != 0:
                    rc = code if code > 0 else 128 - code
                    sys.stderr.write("launch: rank %d exited with code %d; stopping the other ranks\n" % (r, code))
                    break
            if alive and rc == 0:
                time.sleep(poll)
    except KeyboardInterrupt:
        rc = 130
    finally:
        for p in procs:
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
        deadline = time.time() + grace
        for p in procs:
            while p.poll() is None and time.time() < deadline:
                time.sleep(poll)
            if p.poll() is None:
                p.kill()
                p.wait()
    return rc


def relaunch_self_as_ranks(gpus):
    """For scripts with a ``--gpus N`` flag: when N > 1 and this process is not already a rank (no RANK in the environment),
    run the same command line as N ranks and return the job's exit code; otherwise return None and let the caller go on."""
    if int(gpus) <= 1 or "RANK" in os.environ:
        return None
    return spawn_ranks([sys.executable] + sys.argv, int(gpus))


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="Start one process per GPU of this node (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set).")
    ap.add_argument("--nproc", type=int, required=True)
    ap.add_argument("--master-port", type=int, default=None)
    ap.add_argument("script")
    ap.add_argument("args", nargs=argparse.REMAINDER)
    a = ap.parse_args(argv)
    return spawn_ranks([sys.executable, a.script] + a.args, a.nproc, port=a.master_port)


if __name__ == "__main__":
    sys.exit(main())
