"""How long does the first RCCL communicator of a process take, and which HIP calls of another thread stall meanwhile? (EXPERIMENT)"""
import os, sys, threading, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
t0 = time.time()
import torch, torch.distributed as dist
print("import torch %.3f" % (time.time() - t0)); t0 = time.time()
torch.cuda.init(); x = torch.zeros(1024, device="cuda:0"); torch.cuda.synchronize()
print("cuda init %.3f" % (time.time() - t0))
mode = sys.argv[1] if len(sys.argv) > 1 else "sync"
def init():
    t = time.time()
    dist.init_process_group("nccl", rank=0, world_size=1)
    t1 = time.time()
    dist.barrier()
    torch.cuda.synchronize()
    print("init_process_group %.3f  first barrier %.3f" % (t1 - t, time.time() - t1), flush=True)
    t = time.time(); dist.barrier(); torch.cuda.synchronize(); print("second barrier %.4f" % (time.time() - t), flush=True)
if mode == "sync":
    init()
else:
    th = threading.Thread(target=init); th.start()
    worst = {}
    while th.is_alive():
        for name, fn in (("malloc", lambda: torch.empty(1 << 20, device="cuda:0")), ("launch", lambda: x.add_(1)),
                         ("sync", torch.cuda.synchronize), ("pin", lambda: torch.empty(1 << 16).pin_memory())):
            t = time.time(); fn(); d = time.time() - t
            worst[name] = max(worst.get(name, 0), d)
        time.sleep(0.005)
    print("worst latency of main-thread calls during the init:", {k: round(v, 3) for k, v in worst.items()})
