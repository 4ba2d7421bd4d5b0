"""Seeded inputs shared by tests/golden/make_golden.py (which records the reference's outputs for them)
and the tests (which regenerate the same inputs and compare against the recorded outputs)."""
import numpy as np

CONTROL_LENGTHS = [0, 24, 25, 99, 100, 200, 1000, 10000, 10001, 10024, 10025, 20024, 25000, 30010]
CONTROL_SETTINGS = [(25, 10000), (100, -1), (25, -1), (100, 10000), (25, 300)]     # (min_chunk, chunk)
CONTROL_SEED, CONTROL_FEAT = 4321, 5

FWD_SEED, FWD_T = 2024, [25, 200, 400, 1000]


def control_inputs():
    rng = np.random.default_rng(CONTROL_SEED)
    return [("key%02d-T%d" % (i, T), (rng.standard_normal((T, CONTROL_FEAT)) * 3.0).astype(np.float32))
            for i, T in enumerate(CONTROL_LENGTHS)]


def encode_cm_record(key, mat):
    """One binary Kaldi ark record "<key> \\0BCM ..." holding ``mat`` as a CompressedMatrix of the speech-feature kind (the format
    steps/make_mfcc.sh writes by default; Kaldi's CompressedMatrix::CopyFromMat with kSpeechFeature, restated: global min / range,
    per-column percentiles 0 / 25 / 75 / 100 as uint16, one byte per element on the 3-segment piecewise-linear scale).  Test input
    only: the readers are compared with each other on these bytes, not with Kaldi's writer."""
    import struct
    m = np.asarray(mat, np.float32)
    rows, cols = m.shape
    gmin, gmax = float(m.min()), float(m.max())
    if gmax == gmin:
        gmax = gmin + 1.0 + abs(gmin)
    grange = np.float32(gmax - gmin)
    gmin = np.float32(gmin)

    def to_u16(v):
        return np.clip(np.floor((np.asarray(v, np.float64) - float(gmin)) / float(grange) * 65535.0 + 0.499), 0, 65535).astype(np.uint16)

    def from_u16(q):
        return (gmin + grange * np.float32(1.52590218966964e-05) * q.astype(np.float32)).astype(np.float32)
    srt = np.sort(m, axis=0)
    q = np.stack([to_u16(srt[0]), to_u16(srt[rows // 4]), to_u16(srt[(3 * rows) // 4]), to_u16(srt[rows - 1])], axis=1).astype(np.int64)
    # Kaldi keeps the four percentiles strictly increasing
    q[:, 0] = np.minimum(q[:, 0], 65532)
    q[:, 1] = np.minimum(np.maximum(q[:, 1], q[:, 0] + 1), 65533)
    q[:, 2] = np.minimum(np.maximum(q[:, 2], q[:, 1] + 1), 65534)
    q[:, 3] = np.maximum(q[:, 3], q[:, 2] + 1)
    q = q.astype(np.uint16)
    p0, p25, p75, p100 = (from_u16(q[:, i]).astype(np.float64)[None, :] for i in range(4))
    v = m.astype(np.float64)
    lo = np.clip(np.floor((v - p0) / (p25 - p0) * 64.0 + 0.5), 0, 64)
    mid = np.clip(64 + np.floor((v - p25) / (p75 - p25) * 128.0 + 0.5), 64, 192)
    hi = np.clip(192 + np.floor((v - p75) / (p100 - p75) * 63.0 + 0.5), 192, 255)
    u = np.where(v < p25, lo, np.where(v < p75, mid, hi)).astype(np.uint8)
    return key.encode() + b" \0BCM " + struct.pack("<ffii", float(gmin), float(grange), rows, cols) + q.tobytes() + \
        np.ascontiguousarray(u.T).tobytes()


def refgraph_training_case(g, cls):
    """Inputs of one class's record in tests/golden/train_refgraph.npz, regenerated from its seed: (topology, weights, batches,
    dropout masks per step or None)."""
    from xvector_amd import synthetic, topology
    seed, nc = int(g["seed"]), int(g["num_classes"])
    topo = topology.get(cls)
    w = synthetic.trained_like(topo, 23, num_classes=nc, seed=seed)
    rng = np.random.default_rng(seed + 1)
    batches = [((rng.standard_normal((6, 40 + 5 * i, 23)) * 3).astype(np.float16), rng.integers(0, nc, 6).astype(np.int32)) for i in range(3)]
    masks = None
    if cls == "Model":
        masks = []
        widths = topo["layer_sizes"][:4] + topo["embedding_sizes"][:1]
        for step in range(3):
            per = {}
            for site in range(5):
                key = [k for k in g.files if k.startswith("Model/dropout/%d/%d/" % (step, site))][0]
                scope = key.split("/")[-1].split("|")[0]
                shape = (6, 40 + 5 * step, widths[site]) if site < 4 else (6, widths[site])
                per[scope] = (np.unpackbits(g[key])[:int(np.prod(shape))].reshape(shape).astype(np.float64), 0.8)
            masks.append(per)
    return topo, w, batches, masks


def compact(a, stride, small_stride=1):
    a = np.asarray(a, np.float64).reshape(-1)
    return a[::small_stride] if a.size <= 1536 else a[::stride]
