"""RCCL's device code loaded UNDER ``import torch`` instead of after it -- for worker processes that leave through ``os._exit``.

The first RCCL communicator of a process loads ~1 s of device code under the HIP runtime's lock (3.5 s when librccl is cold on the
box); a 50 k-utterance extraction job is ~2 s of wall clock in all, so a job whose one exchange is "a single RCCL gather" (the
north star; local/tf/extract_xvectors.sh:83-95 is the reference's mechanism) paid a third of itself for the bring-up.  ``start()``
-- called by the ``extract_embedding.py`` worker as its first act, before anything imports torch -- opens torch's own ``librccl.so``
/ ``libamdhip64.so`` through ctypes on a side thread and brings up a throw-away ONE-rank communicator on the rank's device with one
256-float all-reduce.  That load then runs beside ``import torch`` (0.8 s, mostly dlopen) and the model load; torch.distributed's own
communicator finds the code resident: its first barrier takes ~0.05 s instead of ~1 s.

The price, and why this module is only ever started from the worker's ``__main__``: a process that dlopens librccl BEFORE torch does
aborts in its exit handlers (``double free or corruption``; bisected in tools/experiments/rccl_prewarm_bisect.py to the dlopen alone).
The worker ends in ``os._exit`` on every path (its outputs are closed and renamed by then), so those handlers never run.  A library
user of ``Model.make_embedding`` never gets here.  ``XVECTOR_RCCL_PREWARM=0`` disables it.
"""
import os
import threading
import time

_STATE = {"state": "off", "log": [], "thread": None}
ENV_FLAG = "XVECTOR_RCCL_PREWARMED"        # set for the rest of the process: dist.gather_backend then keeps RCCL for every payload


def wanted():
    """True for a rank of a grouped GPU job whose transport may be RCCL and that did not opt out."""
    if os.environ.get("XVECTOR_RCCL_PREWARM", "1") == "0":
        return False
    if os.environ.get("XVECTOR_DIST_BACKEND", "nccl") != "nccl":
        return False
    if os.environ.get("XVECTOR_SHARD_OUTPUT", "gather") == "files":          # that mode has no process group at all
        return False
    if not os.path.exists("/dev/kfd"):                                       # no AMD GPU driver node: the job will run over gloo
        return False
    world = int(os.environ.get("WORLD_SIZE", "1"))
    return world > 1 or (os.environ.get("XV_FORCE_DIST") == "1" and "RANK" in os.environ)


def _run(local_rank):
    log = _STATE["log"]
    try:
        import ctypes
        import importlib.util
        t = time.time()
        libdir = os.path.join(list(importlib.util.find_spec("torch").submodule_search_locations)[0], "lib")
        hip = ctypes.CDLL(os.path.join(libdir, "libamdhip64.so"), mode=ctypes.RTLD_GLOBAL)
        rccl = ctypes.CDLL(os.path.join(libdir, "librccl.so"), mode=ctypes.RTLD_GLOBAL)
        log.append("dlopen %.3f" % (time.time() - t)); t = time.time()
        if hip.hipSetDevice(int(local_rank)) != 0:
            raise RuntimeError("hipSetDevice(%d) failed" % local_rank)
        comm = ctypes.c_void_p()
        devs = (ctypes.c_int * 1)(int(local_rank))
        rc = rccl.ncclCommInitAll(ctypes.byref(comm), 1, devs)
        if rc != 0:
            raise RuntimeError("ncclCommInitAll -> %d" % rc)
        log.append("communicator %.3f" % (time.time() - t)); t = time.time()
        buf = ctypes.c_void_p()
        if hip.hipMalloc(ctypes.byref(buf), 4096) != 0:
            raise RuntimeError("hipMalloc failed")
        rc = rccl.ncclAllReduce(buf, buf, ctypes.c_size_t(256), 7, 0, comm, None)        # ncclFloat32 = 7, ncclSum = 0, null stream
        hip.hipDeviceSynchronize()
        if rc != 0:
            raise RuntimeError("ncclAllReduce -> %d" % rc)
        log.append("first all-reduce %.3f" % (time.time() - t))
        _STATE["state"] = "ok"                      # the communicator is left alive: the process ends in os._exit
    except BaseException as e:                      # noqa: B902 -- the job goes on with the plain bring-up
        log.append("failed: %r" % (e,))
        _STATE["state"] = "failed"


def start(local_rank=None):
    """Start the pre-load thread (once).  Returns True when started.  MUST be called before torch is imported, and only by a process
    that ends in os._exit (see the module text)."""
    if _STATE["thread"] is not None or not wanted():
        return False
    import sys
    if "torch" in sys.modules:                       # too late to be in front of torch's own dlopen: nothing to win, nothing risked
        return False
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ[ENV_FLAG] = "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    _STATE["state"] = "running"
    t = threading.Thread(target=_run, args=(local_rank,), name="xv-rccl-prewarm", daemon=True)
    _STATE["thread"] = t
    t.start()
    return True


def started():
    return _STATE["thread"] is not None


def join(timeout=None):
    """Wait for the pre-load (torch's communicator is brought up after it: two bring-ups at once would serialise on the HIP lock
    anyway).  -> "off" | "ok" | "failed" | "running" (timed out)."""
    t = _STATE["thread"]
    if t is not None:
        t.join(timeout)
    return _STATE["state"]


def report():
    return "%s (%s)" % (_STATE["state"], "; ".join(_STATE["log"])) if _STATE["thread"] is not None else "off"
