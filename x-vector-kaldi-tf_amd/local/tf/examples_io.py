"""Training examples ("egs") in the reference's tar format -- the data side of ``Model.train_one_iteration`` / ``eval``.

The reference's trainer reads one ``egs.<n>.tar`` per job (``train_dnn.py:258-300`` passes ``--tar-file``): a tar whose
members ``minibatch_<i>.npy`` are float16 ``[B, T, F]`` arrays (one length T per minibatch) written by
``examples_io.save_data_info_tar`` (examples_io.py:156-185), next to ``egs.<n>.npy`` holding the int labels ``[n_minibatches,
B]`` (read at examples_io.py:226).  ``TarFileDataLoader`` (examples_io.py:223-255) serves them through a bounded queue
filled by a background thread; the model pops ``(data, labels)`` with a timeout and gets ``(None, None)`` once the archive is
exhausted.  This module keeps that protocol (class name, ``count``, ``pop(timeout)``) and adds the matching writer, so
training runs end to end without the rest of the reference's example-generation tooling (ranges files, h5 export: control
plane, out of scope).
"""
import io
import queue
import tarfile
import threading

import numpy as np

__all__ = ["TarFileDataLoader", "write_egs_tar"]


def write_egs_tar(tar_path, minibatches, labels):
    """``minibatches``: list of arrays [B, T_i, F] (stored as float16, like the reference); ``labels``: int array
    [len(minibatches), B].  Writes ``tar_path`` and the label file ``tar_path`` with ``.tar`` -> ``.npy``."""
    labels = np.asarray(labels)
    assert tar_path.endswith(".tar") and labels.shape[0] == len(minibatches)
    with tarfile.open(tar_path, "w") as tar:
        for i, m in enumerate(minibatches):
            assert m.ndim == 3 and m.shape[0] == labels.shape[1]
            buf = io.BytesIO()
            np.save(buf, np.asarray(m, dtype=np.float16))
            info = tarfile.TarInfo(name="minibatch_%d.npy" % i)
            info.size = buf.tell()
            buf.seek(0)
            tar.addfile(info, buf)
    np.save(tar_path[:-4] + ".npy", labels)


class TarFileDataLoader(object):
    """Same duck type as the reference's loader: ``count`` minibatches, ``pop(timeout)`` -> ``(float16 [B,T,F], labels[B])``
    in tar order, ``(None, None)`` when there is nothing (left) to read; may raise ``queue.Empty`` on timeout."""

    def __init__(self, tar_file, logger=None, queue_size=5):
        self._labels = np.load(tar_file.replace(".tar", ".npy"))
        self._tar = tarfile.open(tar_file, "r")
        self._names = self._tar.getnames()
        self.count = len(self._names)
        assert self.count == self._labels.shape[0], "label file does not match the archive"
        self._left = self.count
        self._logger = logger
        self.queue = queue.Queue(queue_size)
        self._thread = threading.Thread(target=self._fill, daemon=True)
        self._thread.start()

    def _fill(self):
        for name in self._names:
            index = int(name[:-4].split("_")[1])              # minibatch_<index>.npy
            # read the member fully first: numpy cannot memory-map a tar member and (numpy 2.x) probes fileno()
            data = np.load(io.BytesIO(self._tar.extractfile(name).read()))
            self.queue.put((data, self._labels[index]))

    def pop(self, timeout=30):
        if self._left == 0:
            return None, None
        item = self.queue.get(block=True, timeout=timeout)
        self._left -= 1
        return item
