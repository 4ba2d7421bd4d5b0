import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "x-vector-kaldi-tf_amd"))
from xvector_amd import hiplib
dev = torch.device("cuda:0")
vals = np.array([[1 + 0.97 / 2048, 1 + 0.93 / 2048, 1 + 0.9 / 2048, 1 - 0.97 / 4096, 3.3, 0.7, 1 + 0.81/2048, 1 + 0.80/2048] + [0.0] * 24], np.float32)
buf = hiplib.SplitBuf(1, 32, dev, hiplib.FMT_SPLIT8)
hiplib.split_encode(torch.from_numpy(vals).to(dev), buf)
back = hiplib.split_decode(buf, 1).cpu().numpy()
print("lo*2048 in :", ((vals - vals.astype(np.float16).astype(np.float32)) * 2048)[0, :8])
print("lo*2048 out:", ((back - vals.astype(np.float16).astype(np.float32)) * 2048)[0, :8])
raw = buf.base.cpu().numpy()[hiplib.SPLIT_PAD_BEFORE * buf.row_bytes:][:128]
cross = raw[64:80]
print("h8 of 3.3 / 0.7:", (cross[8 + 4:8 + 6].astype(np.uint16) << 8).view(np.float16))
