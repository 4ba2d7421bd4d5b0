"""Training examples ("egs") in the reference's tar format -- the data side of ``Model.train_one_iteration`` / ``eval``.

The reference's trainer reads one ``egs.<n>.tar`` per job (``train_dnn.py:258-300`` passes ``--tar-file``): a tar whose
members ``minibatch_<i>.npy`` are float16 ``[B, T, F]`` arrays (one length T per minibatch) written by
``examples_io.save_data_info_tar`` (examples_io.py:156-185), next to ``egs.<n>.npy`` holding the int labels ``[n_minibatches,
B]`` (read at examples_io.py:226).  ``TarFileDataLoader`` (examples_io.py:223-255) serves them through a bounded queue
filled by a background thread; the model pops ``(data, labels)`` with a timeout and gets ``(None, None)`` once the archive is
exhausted.  This module keeps that protocol (class name, ``count``, ``pop(timeout)``) and adds the matching writer, so
training runs end to end without the rest of the reference's example-generation tooling (ranges files, h5 export: control
plane, out of scope).

``RangesDataLoader`` is the other input mode of the reference's trainer (train_dnn_one_iteration.py:177-200; always on the
command line ``train_dnn.py:258-262`` builds, used when no tar exists): minibatches are cut on the fly from a feature
table according to a Kaldi ranges file, one line per chunk

    <utt-id> <minibatch-index> <ignored> <first-frame> <num-frames> <label>

(examples_io.py:20-23).  Pinned against what the reference's own loader serves: tests/golden/egs_ranges.npz.
"""
import io
import queue
import tarfile
import threading

import numpy as np

__all__ = ["TarFileDataLoader", "RangesDataLoader", "write_egs_tar"]


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
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._fill, daemon=True)
        self._thread.start()

    def _fill(self):
        for name in self._names:
            index = int(name[:-4].split("_")[1])              # minibatch_<index>.npy
            # read the member fully first: numpy cannot memory-map a tar member and (numpy 2.x) probes fileno()
            data = np.load(io.BytesIO(self._tar.extractfile(name).read()))
            while not self._stop.is_set():
                try:
                    self.queue.put((data, self._labels[index]), timeout=0.2)
                    break
                except queue.Full:
                    continue
            if self._stop.is_set():
                return

    def close(self):
        """Stop the reader thread and release the archive (a consumer that pops fewer than ``count`` minibatches)."""
        self._stop.set()
        self._thread.join()
        self._tar.close()

    def pop(self, timeout=30):
        if self._left == 0:
            return None, None
        item = self.queue.get(block=True, timeout=timeout)
        self._left -= 1
        return item


class RangesDataLoader(object):
    """Minibatches cut from ``scp_file`` as listed in ``ranges_file`` (float32 [B, T, F], int32 labels [B]).

    Semantics of the reference pipeline process_range_file -> load_ranges_data -> [shuffle] -> DataLoader
    (examples_io.py:12-75,188-221; train_dnn_one_iteration.py:177-200): every chunk of a minibatch has the same length;
    a minibatch must hold exactly ``minibatch_size`` chunks, filled in the order the utterances come out of the scp and,
    within an utterance, in ranges-file order; ``shuffle`` permutes the minibatches with ``np.random.permutation`` (seed it
    with ``np.random.seed`` beforehand, as the trainer does); the loader then serves them LAST FIRST, because the
    reference pops from the end of its list.  An utterance of the scp without a ranges entry is an error there too."""

    def __init__(self, ranges_file, scp_file, minibatch_count, minibatch_size, feature_dim, shuffle=False, logger=None):
        import kaldi_io
        chunks, length, filled = {}, [None] * minibatch_count, [0] * minibatch_count
        total = [0] * minibatch_count
        with open(ranges_file, "rt") as fid:
            for line in fid:
                f = line.split()
                if not f:
                    continue
                mb, first, n, label = int(f[1]), int(f[3]), int(f[4]), int(f[5])
                chunks.setdefault(f[0], []).append((mb, first, n, label))
                if length[mb] is None:
                    length[mb] = n
                assert length[mb] == n, "minibatch %d mixes chunk lengths %d and %d" % (mb, length[mb], n)
                total[mb] += 1
        for mb in range(minibatch_count):
            assert length[mb] is not None and total[mb] == minibatch_size, \
                "minibatch %d holds %d chunks, expected %d" % (mb, total[mb], minibatch_size)
        data = [np.zeros((minibatch_size, length[mb], feature_dim), np.float32) for mb in range(minibatch_count)]
        labels = [np.zeros(minibatch_size, np.int32) for _ in range(minibatch_count)]
        for key, mat in kaldi_io.read_mat_scp(scp_file):
            for mb, first, n, label in chunks[key]:
                piece = mat[first:first + n]
                assert piece.shape == (length[mb], feature_dim), "chunk of %s does not fit its utterance / feature dim" % key
                data[mb][filled[mb]] = piece
                labels[mb][filled[mb]] = label
                filled[mb] += 1
        order = np.arange(minibatch_count)
        if shuffle:
            order = np.random.permutation(order)
        self._queue = [(data[i], labels[i]) for i in order]
        self.count = minibatch_count
        if logger is not None:
            logger.info("Loaded %d minibatches from %d utterances." % (minibatch_count, len(chunks)))

    def pop(self, timeout=30):
        if not self._queue:
            return None, None
        return self._queue.pop()
