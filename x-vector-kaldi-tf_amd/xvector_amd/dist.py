"""Utterance sharding across the GPUs of one node + the single gather at the end.

The reference's only multi-worker mechanism on this path is ``nj`` independent processes over disjoint
utterance shards whose outputs are concatenated (local/tf/extract_xvectors.sh:63-65,83-88,92-95).  The
MI355X equivalent: one process per GPU (``torch.distributed``, backend "nccl" == RCCL over xGMI), a
deterministic frame-balanced partition every rank can recompute, NO collective on the data path, and
ONE gather of the ``[N_r, E]`` fp32 embedding blocks to rank 0, which restores input order and writes
the ark.  Keys and rejected-utterance flags need no communication: they follow from the lengths.
The CLI worker's gather is RCCL too (round 6): ``extract_embedding.py`` pre-loads RCCL's device code under ``import torch``
(``rccl_prewarm``), which removed the ~1 s bring-up that had made gloo the better transport for small jobs.  A process that
could not pre-load (a library user of ``Model.make_embedding``, XVECTOR_RCCL_PREWARM=0) still picks the transport of that
one gather by its size (``gather_backend``: gloo for a few hundred MB, RCCL above); everything else -- the training step's
all-reduces, bench.py, the stream / byte-range modes -- is RCCL.
"""
import heapq
import os

import numpy as np


def env_world():
    """(rank, world_size, local_rank) from the torchrun environment (1-process defaults)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def group_wanted():
    """True when this process is a rank of a job that needs a process group: WORLD_SIZE > 1, or a forced 1-rank group
    (XV_FORCE_DIST=1 under a launcher: exercises the RCCL path on the one GPU a test box has)."""
    rank, world, _ = env_world()
    return world > 1 or (os.environ.get("XV_FORCE_DIST") == "1" and "RANK" in os.environ)


def group_shape():
    """(rank, world) of the job as the launcher's environment states it -- known before the group is up."""
    rank, world, _ = env_world()
    return (rank, world) if world > 1 else (0, 1)


def init_process_group(backend=None):
    """Initialise torch.distributed from the environment if WORLD_SIZE > 1.  Returns (rank, world)."""
    import torch
    import torch.distributed as dist
    rank, world, local = env_world()
    force = os.environ.get("XV_FORCE_DIST") == "1" and "RANK" in os.environ      # 1-rank group (tests the RCCL path on one GPU)
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # XVECTOR_DIST_BACKEND=gloo: the control flow of a multi-rank job on a box with ONE GPU (all ranks share it; RCCL
            # refuses two ranks on one device) -- what tests/test_gpu_bench_contract.py uses to run bench.py at N = 2
            backend = os.environ.get("XVECTOR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        # RCCL prints a five-line version banner (and gloo its "[Gloo] Rank r is connected to ..." line) through C stdio's
        # stdout when its first communicator is created; callers of this package promise machine-readable stdout (bench.py:
        # ONE JSON line), so fd 1 points at stderr while the communicator is brought up (a barrier forces it) and the C
        # buffers are flushed
        import ctypes
        import sys
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
            dist.barrier()
            ctypes.CDLL(None).fflush(None)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
    return rank, world


_ASYNC = {}
HOST_GATHER_MAX_MB = 512.0


def gather_backend(payload_bytes):
    """Transport of a job whose ONLY exchange is the final gather of ``payload_bytes`` (all ranks together) of x-vectors that already
    lie in host memory: XVECTOR_DIST_BACKEND if set; else "gloo" up to XVECTOR_HOST_GATHER_MAX_MB (default 512) -- the first RCCL
    communicator of a process loads ~1 s of device code under the HIP runtime's lock (2.4 s more when librccl is cold on the box),
    which a 50 k-utterance job's 100 MB of vectors never earn back over loopback; above it "nccl" (RCCL over xGMI: a configs[3]
    share is 2 GB into rank 0).  Every rank computes the same answer from the same shard table."""
    import torch
    forced = os.environ.get("XVECTOR_DIST_BACKEND")
    if forced:
        return forced
    if not torch.cuda.is_available():
        return "gloo"
    from . import rccl_prewarm
    if os.environ.get(rccl_prewarm.ENV_FLAG) == "1":
        # the worker pre-loaded RCCL's device code under `import torch` (rccl_prewarm): the bring-up that made gloo the better choice
        # for small payloads is gone.  Decided from the environment alone -- the same on every rank, whatever became of its pre-load
        return "nccl"
    limit = float(os.environ.get("XVECTOR_HOST_GATHER_MAX_MB", HOST_GATHER_MAX_MB))
    return "gloo" if payload_bytes <= limit * 1e6 else "nccl"


def set_gather_payload(payload_bytes):
    """Tell a pending ``init_process_group_async`` what the job will gather (before its start mark fires): picks the transport."""
    if "box" in _ASYNC and "thread" in _ASYNC:
        if _ASYNC["go"].is_set():
            # a rank whose bring-up has started keeps the default transport while its peers may pick another: a rendezvous that hangs.
            # The callers state the payload before the first window is launched; anything else is a programming error, said loudly
            raise RuntimeError("set_gather_payload after the process group's bring-up has started: the ranks could disagree on the transport")
        _ASYNC["backend"] = gather_backend(payload_bytes)


def init_process_group_async(backend=None, after_mark=None):
    """``init_process_group`` on a side thread: the communicator (RCCL: ~1 s for the first one of a process) comes up while the
    caller extracts; ``wait_process_group()`` joins it in front of the first collective.  Only for callers that write nothing to
    stdout before that point (see the redirection in ``init_process_group``).
    ``after_mark``: the bring-up starts when the job clock marks that name (jobclock.on; extract_embedding.py: "first window
    launched") -- RCCL's kernel-load phase (0.8 s of its second) holds the HIP runtime's lock, and a model load next to it takes
    0.9 s instead of 0.12: with the model on the device and the first window's kernels queued before the lock is taken, what
    stalls next to the bring-up is the rest of the extraction, which it outlasts anyway.  ``wait_process_group`` releases a
    thread that is still waiting for its mark (an input without a single window)."""
    import threading
    if "thread" in _ASYNC:
        return
    box = {}
    go = threading.Event()
    if after_mark is None:
        go.set()
    else:
        from . import jobclock
        jobclock.on(after_mark, go.set)

    def run():
        import time
        from . import jobclock
        go.wait()
        t0 = time.time()
        try:
            chosen = _ASYNC.get("backend") or backend
            from . import rccl_prewarm
            if rccl_prewarm.started():                  # torch's communicator after the throw-away one: it finds the device code resident
                state = rccl_prewarm.join(60.0)
                jobclock.note("RCCL pre-load (%s) joined after" % state, time.time() - t0)
                if state != "ok":
                    import logging
                    logging.getLogger("xvector_amd.dist").warning("RCCL pre-load: %s -- the group comes up the plain way", rccl_prewarm.report())
            box["value"] = init_process_group(chosen)
            jobclock.note("process group (%s) up on its side thread after" % (chosen or "default"), time.time() - t0)
        except BaseException as e:          # noqa: B902 -- re-raised by wait_process_group
            box["error"] = e
    t = threading.Thread(target=run, name="xv-process-group", daemon=True)
    _ASYNC.update(thread=t, box=box, go=go)
    t.start()


def wait_process_group(backend=None):
    """(rank, world) once the group is up: joins the side thread of ``init_process_group_async`` if there is one, else
    initialises here."""
    if "thread" not in _ASYNC:
        return init_process_group(backend)
    _ASYNC["go"].set()
    _ASYNC["thread"].join()
    box = _ASYNC["box"]
    if "error" in box:
        raise box["error"]
    return box["value"]


def finish_process_group(destroy=False):
    """Barrier at the end of a job: no rank leaves (and takes the communicator with it) while the root still reads.  The group is
    only destroyed on request: a worker process that exits right afterwards saves the 0.2 s the teardown takes."""
    import torch.distributed as dist
    if "thread" in _ASYNC:
        wait_process_group()
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        if destroy:
            dist.destroy_process_group()
    _ASYNC.clear()


def partition_lpt(lengths, world):
    """Greedy longest-processing-time partition of utterances over ``world`` ranks balancing the number
    of frames.  Deterministic (ties: lower index first, lower rank first).  Returns a list of ``world``
    int64 index arrays, each sorted ascending (so every shard keeps input order)."""
    lengths = np.asarray(lengths, dtype=np.int64)
    order = np.lexsort((np.arange(len(lengths)), -lengths))       # by length desc, then index asc
    heap = [(0, r) for r in range(world)]
    heapq.heapify(heap)
    shards = [[] for _ in range(world)]
    for i in order:
        load, r = heapq.heappop(heap)
        shards[r].append(int(i))
        heapq.heappush(heap, (load + int(lengths[i]), r))
    return [np.array(sorted(s), dtype=np.int64) for s in shards]


def gather_blocks(local, counts, dst=0, group=None):
    """ONE collective: gather the per-rank ``[counts[r], E]`` blocks on ``dst``.  Blocks are padded to
    ``max(counts)`` rows so that a single fixed-shape ``dist.gather`` moves everything (each peer->root
    transfer rides its own xGMI link).  Returns the list of un-padded blocks on ``dst``, else None."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    assert local.shape[0] == counts[rank]
    if not dist.is_initialized():
        return [local]
    cap = int(max(counts))
    padded = local
    if local.shape[0] != cap:
        padded = torch.zeros((cap, local.shape[1]), dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
    padded = padded.contiguous()
    staged = dist.get_backend(group) == "gloo" and padded.is_cuda       # gloo gathers host tensors only
    send = padded.cpu() if staged else padded
    bucket = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, bucket, dst=dst, group=group)
    if rank != dst:
        return None
    return [(b.to(padded.device) if staged else b)[: counts[r]] for r, b in enumerate(bucket)]


def unshard(blocks, shards, total, dim):
    """Inverse of the partition on the root: blocks[r][j] is utterance shards[r][j]."""
    import torch
    out = torch.empty((total, dim), dtype=blocks[0].dtype, device=blocks[0].device)
    for blk, idx in zip(blocks, shards):
        if len(idx):
            out[torch.as_tensor(idx, device=out.device)] = blk
    return out


def sharded_extract(lengths, extract_shard, dim, device, group=None):
    """Run ``extract_shard(indices) -> tensor[len(indices), dim]`` on this rank's shard of the
    utterances and gather everything on rank 0 in input order.  Returns the ``[N, dim]`` tensor on rank
    0 and None elsewhere.  ``extract_shard`` must return a row for every index (rejected utterances:
    any filler; the caller drops them by length)."""
    import torch.distributed as dist
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_initialized() else (0, 1)
    shards = partition_lpt(lengths, world)
    local = extract_shard(shards[rank])
    assert local.shape == (len(shards[rank]), dim)
    blocks = gather_blocks(local.to(device), [len(s) for s in shards], 0, group)
    if rank != 0:
        return None
    return unshard(blocks, shards, len(lengths), dim)
