"""The C-ABI shared library loads on a CPU-only box and exports every symbol include/xvector_hip.h
declares; compute entry points refuse to run without a GPU (no silent fallback)."""
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "xvector_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(xv_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    syms = _declared_symbols()
    for s in ("xv_tdnn_layer_f32", "xv_stats_pool_f32", "xv_fc_f32", "xv_chunk_average_f32", "xv_version", "xv_last_error"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    from xvector_amd import hiplib
    lib = hiplib.load()
    declared = _declared_symbols()
    assert sorted(hiplib.SYMBOLS) == declared            # the binding list and the header agree
    for s in declared:
        assert getattr(lib, s) is not None
    assert lib.xv_version() == hiplib.ABI_VERSION
    assert lib.xv_stats_pool_workspace_bytes(1536, 10, 400, 512) == 0
    assert lib.xv_stats_pool_workspace_bytes(1536, 10, 10000, 512) == 10 * 20 * 2 * 1536 * 4


def test_no_cpu_fallback():
    import torch
    from xvector_amd import hiplib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hiplib.XvectorHipError):
        hiplib.require_gpu()
    with pytest.raises(hiplib.XvectorHipError):
        hiplib.pack_weights(torch.zeros((4, 4)))
    from xvector_amd import engine, synthetic
    topo = synthetic.SMALL_TOPOLOGY
    with pytest.raises(hiplib.XvectorHipError):
        engine.DeviceModel(synthetic.trained_like(topo, 5, seed=0), topo)


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py may touch oracle/."""
    pkg = os.path.join(ROOT, "x-vector-kaldi-tf_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(d, f), errors="ignore").read()
                assert "import oracle" not in src and "from oracle" not in src and "xv_oracle" not in src, os.path.join(d, f)


def test_host_library_loads_and_scans():
    """libxvector_host.so (native ark scanner) is built by build() and agrees with the pure-Python reader."""
    import io
    import numpy as np
    import kaldi_io
    lib = kaldi_io._host_lib()
    assert lib is not None and lib.xv_host_version() == 11 and hasattr(lib, "xv_scp_line_index") and hasattr(lib, "xv_vec_records_write_fd") and hasattr(lib, "xv_raw_row_plan") and hasattr(lib, "xv_ark_decode_cm") and hasattr(lib, "xv_ark_index_fd") and hasattr(lib, "xv_copy_bytes") and hasattr(lib, "xv_ark_keys") and hasattr(lib, "xv_ark_gather_fm") and hasattr(lib, "xv_pack_rows_f32")
    rng = np.random.default_rng(0)
    bio = io.BytesIO()
    mats = [("k%d" % i, rng.standard_normal((int(rng.integers(0, 40)), 23)).astype(np.float32)) for i in range(50)]
    for i, (k, m) in enumerate(mats):
        kaldi_io.write_mat(bio, m.astype(np.float64) if i == 17 else m, key=k)        # one DM record in the middle
    got = list(kaldi_io.read_mat_ark(io.BytesIO(bio.getvalue())))
    assert [k for k, _ in got] == [k for k, _ in mats]
    assert all(np.array_equal(a, b) for (_, a), (_, b) in zip(got, mats))
    assert got[17][1].dtype == np.float64 and got[16][1].dtype == np.float32


def test_integration_doc_names_every_entry_point():
    """INTEGRATION.md maps each C-ABI symbol to the reference interface it replaces: none may be missing."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [s for s in _declared_symbols() if s not in doc]
    assert not missing, missing
