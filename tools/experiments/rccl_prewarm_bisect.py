import ctypes, importlib.util, os, sys, threading, time
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
mode = sys.argv[1]
libdir = os.path.join(list(importlib.util.find_spec("torch").submodule_search_locations)[0], "lib")
def work():
    if mode == "none": return
    hip = ctypes.CDLL(os.path.join(libdir, "libamdhip64.so"), mode=ctypes.RTLD_GLOBAL)
    if mode == "hip_only":
        hip.hipSetDevice(0); return
    rccl = ctypes.CDLL(os.path.join(libdir, "librccl.so"), mode=ctypes.RTLD_GLOBAL)
    if mode == "dlopen_rccl": return
    hip.hipSetDevice(0)
    comm = ctypes.c_void_p(); devs = (ctypes.c_int * 1)(0)
    print("init rc", rccl.ncclCommInitAll(ctypes.byref(comm), 1, devs))
    if mode == "init_destroy": print("destroy rc", rccl.ncclCommDestroy(comm))
if mode.endswith("_main"):
    mode = mode[:-5]; work()
else:
    th = threading.Thread(target=work); th.start(); th.join()
if "notorch" not in sys.argv:
    import torch
    torch.cuda.init(); x = torch.zeros(4, device="cuda:0"); torch.cuda.synchronize()
print("done", mode)
