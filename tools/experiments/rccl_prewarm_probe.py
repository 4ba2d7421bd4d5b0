"""Can RCCL's device code be loaded BEFORE / WHILE `import torch` runs, so that the job's own communicator comes up fast? (EXPERIMENT)
A side thread, started before torch is imported, opens torch's librccl.so through ctypes and brings up a throw-away ONE-rank
communicator on the job's device (ncclCommInitAll) + one tiny all-reduce; the main thread meanwhile imports torch, initialises HIP,
then times dist.init_process_group + the first barrier.  argv[1]: "prewarm" | "plain"."""
import ctypes, importlib.util, os, sys, threading, time
T0 = time.time()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29613")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
mode = sys.argv[1] if len(sys.argv) > 1 else "prewarm"
log = []


def prewarm():
    t = time.time()
    libdir = os.path.join(list(importlib.util.find_spec("torch").submodule_search_locations)[0], "lib")
    hip = ctypes.CDLL(os.path.join(libdir, "libamdhip64.so"), mode=ctypes.RTLD_GLOBAL)
    rccl = ctypes.CDLL(os.path.join(libdir, "librccl.so"), mode=ctypes.RTLD_GLOBAL)
    log.append("dlopen %.3f" % (time.time() - t)); t = time.time()
    assert hip.hipSetDevice(int(os.environ["LOCAL_RANK"])) == 0
    comm = ctypes.c_void_p()
    devs = (ctypes.c_int * 1)(int(os.environ["LOCAL_RANK"]))
    rc = rccl.ncclCommInitAll(ctypes.byref(comm), 1, devs)
    log.append("ncclCommInitAll rc=%d %.3f" % (rc, time.time() - t)); t = time.time()
    buf = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(buf), 4096) == 0
    rc = rccl.ncclAllReduce(buf, buf, ctypes.c_size_t(256), 7, 0, comm, None)      # ncclFloat = 7, ncclSum = 0, null stream
    hip.hipDeviceSynchronize()
    log.append("first all-reduce rc=%d %.3f" % (rc, time.time() - t)); t = time.time()
    if mode.endswith("destroy"):
        rccl.ncclCommDestroy(comm)
        log.append("destroy %.3f" % (time.time() - t))
    elif mode.endswith("abort"):
        rccl.ncclCommAbort(comm)
        log.append("abort %.3f" % (time.time() - t))


th = None
if mode.startswith("prewarm"):
    th = threading.Thread(target=prewarm); th.start()
t = time.time()
import torch, torch.distributed as dist
print("import torch %.3f" % (time.time() - t)); t = time.time()
torch.cuda.init(); x = torch.zeros(1024, device="cuda:0"); torch.cuda.synchronize()
print("cuda init + first kernel %.3f" % (time.time() - t)); t = time.time()
if th is not None:
    th.join()
    print("prewarm thread joined after another %.3f: %s" % (time.time() - t, "; ".join(log))); t = time.time()
dist.init_process_group("nccl", rank=0, world_size=1)
t1 = time.time()
dist.barrier(); torch.cuda.synchronize()
print("init_process_group %.3f  first barrier %.3f" % (t1 - t, time.time() - t1)); t = time.time()
y = [torch.zeros(1024, device="cuda:0")]
dist.gather(x, y, dst=0); torch.cuda.synchronize()
print("first gather %.3f   total since start %.3f" % (time.time() - t, time.time() - T0))
dist.destroy_process_group()
print("clean exit")
