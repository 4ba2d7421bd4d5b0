"""Kaldi table I/O for the MI355X x-vector extractor (drop-in for the reference module of the
same name on the extraction path).

Mirrors the subset of ``local/tf/kaldi_io.py`` of BUTSpeechFIT/x-vector-kaldi-tf that
``Model.make_embedding`` / ``extract_embedding.py`` touch -- same function names, argument
meaning and exceptions -- re-implemented around a buffered exact-read helper and vectorised
NumPy decoding (the reference reads keys a byte at a time and decodes compressed matrices column
by column in Python):

=====================  =========================================  ===========================
here                   replaces (reference file:line)              notes
=====================  =========================================  ===========================
open_or_fd             local/tf/kaldi_io.py:50-80                 ark:/scp: prefixes, :offset,
                                                                   ``cmd |`` / ``| cmd`` pipes, .gz
popen                  local/tf/kaldi_io.py:84-117                SubprocessFailed on rc>0
read_key               local/tf/kaldi_io.py:120-133
read_mat_ark/read_mat  local/tf/kaldi_io.py:372-410
_read_mat_binary       local/tf/kaldi_io.py:413-437               FM / DM / CM (+CM2, CM3)
_read_mat_ascii        local/tf/kaldi_io.py:440-452
_read_compressed_mat   local/tf/kaldi_io.py:455-502               float32 arithmetic as Kaldi
read_mat_scp           local/tf/kaldi_io.py:350-369
write_mat              local/tf/kaldi_io.py:506-542
read_vec_flt[_ark|_scp] local/tf/kaldi_io.py:225-305
write_vec_flt          local/tf/kaldi_io.py:309-343               'FV '/'DV ' framing
=====================  =========================================  ===========================

Int vectors, posteriors, confusion-network times and segment bool-vectors (reference lines
139-218, 553-697) are ASR types the x-vector path never reads: out of scope.
"""
import ctypes
import gzip
import io
import os
import re
import struct
import subprocess
import threading

import numpy as np

__all__ = ["open_or_fd", "popen", "TableWriter", "read_key", "read_mat", "read_mat_ark", "read_mat_scp", "write_mat",
           "read_vec_flt", "read_vec_flt_ark", "read_vec_flt_scp", "write_vec_flt", "write_vec_flt_batch",
           "UnsupportedDataType", "UnknownVectorHeader", "UnknownMatrixHeader", "BadSampleSize",
           "BadInputFormat", "SubprocessFailed"]


class UnsupportedDataType(Exception):
    pass


class UnknownVectorHeader(Exception):
    pass


class UnknownMatrixHeader(Exception):
    pass


class BadSampleSize(Exception):
    pass


class BadInputFormat(Exception):
    pass


class SubprocessFailed(Exception):
    pass


_SPECIFIER = re.compile(r"^(ark|scp)(,scp|,b|,t|,n?f|,n?p|,b?o|,n?s|,n?cs)*:")
_OFFSET = re.compile(r":[0-9]+$")
_KEY_OK = re.compile(r"^[\.\/a-zA-Z0-9_-]+$")


# ------------------------------------------------------------------------------------------------
# opening things
# ------------------------------------------------------------------------------------------------
def popen(cmd, mode="rb"):
    """Run ``cmd`` through the shell and return the pipe end matching ``mode``.  A helper thread
    waits for the child and raises SubprocessFailed when it exits with a positive status."""
    if not isinstance(cmd, str):
        raise TypeError("invalid cmd type (%s, expected string)" % type(cmd))
    if mode not in ("r", "w", "rb", "wb"):
        raise ValueError("invalid mode %s" % mode)
    reading = mode[0] == "r"
    proc = subprocess.Popen(cmd, shell=True,
                            stdout=subprocess.PIPE if reading else None,
                            stdin=None if reading else subprocess.PIPE)

    def _reap():
        rc = proc.wait()
        if rc > 0:
            raise SubprocessFailed("cmd %s returned %d !" % (cmd, rc))

    threading.Thread(target=_reap, daemon=True).start()
    end = proc.stdout if reading else proc.stdin
    return io.TextIOWrapper(end) if len(mode) == 1 else end


def open_or_fd(file, mode="rb"):
    """Open a Kaldi rxfilename / wxfilename (or pass an already opened stream through).

    Accepts an optional ``ark:`` / ``scp:`` (with options) prefix, ``path:offset``, a trailing
    ``|`` (read from command), a leading ``|`` (write to command) and ``.gz`` files."""
    if not isinstance(file, str):
        return file                       # already a stream
    offset = None
    if _SPECIFIER.search(file):
        file = file.split(":", 1)[1]
    if _OFFSET.search(file):
        file, offset = file.rsplit(":", 1)
    if file.endswith("|"):
        fd = popen(file[:-1], "rb")
    elif file.startswith("|"):
        fd = popen(file[1:], "wb")
    elif file.rsplit(".", 1)[-1] == "gz":
        fd = gzip.open(file, mode)
    else:
        fd = open(file, mode)
    if offset is not None:
        fd.seek(int(offset))
    return fd


class _BufferedStream(object):
    """Read-ahead wrapper used by the ark generators: the reference reads keys one byte at a time through
    ``fd.read(1)``; this pulls the stream in 4 MiB blocks and serves ``read`` / ``readline`` / key scans from memory.
    Whatever was read ahead is handed back to a seekable ``fd`` (``seek`` to the logical position) on ``detach``."""

    BLOCK = 1 << 22

    def __init__(self, fd):
        self.fd = fd
        self.buf = b""
        self.pos = 0

    def _fill(self, need):
        """Make at least ``need`` bytes available after pos (fewer only at end of stream)."""
        avail = len(self.buf) - self.pos
        if avail >= need:
            return
        parts = [self.buf[self.pos:]]
        while avail < need:
            blk = self.fd.read(max(self.BLOCK, need - avail))
            if not blk:
                break
            parts.append(blk)
            avail += len(blk)
        self.buf = b"".join(parts)
        self.pos = 0

    def read(self, n=-1):
        if n is None or n < 0:
            rest = self.buf[self.pos:] + self.fd.read()
            self.buf, self.pos = b"", 0
            return rest
        self._fill(n)
        out = self.buf[self.pos:self.pos + n]
        self.pos += len(out)
        return out

    def readline(self):
        while True:
            i = self.buf.find(b"\n", self.pos)
            if i >= 0:
                out = self.buf[self.pos:i + 1]
                self.pos = i + 1
                return out
            before = len(self.buf) - self.pos
            self._fill(before + 1)
            if len(self.buf) - self.pos == before:          # end of stream
                out = self.buf[self.pos:]
                self.pos = len(self.buf)
                return out

    def read_token(self):
        """Bytes up to (not including) the next space, consuming the space; b"" at end of stream."""
        while True:
            i = self.buf.find(b" ", self.pos)
            if i >= 0:
                out = self.buf[self.pos:i]
                self.pos = i + 1
                return out
            before = len(self.buf) - self.pos
            self._fill(before + 1)
            if len(self.buf) - self.pos == before:
                out = self.buf[self.pos:]
                self.pos = len(self.buf)
                return out

    def detach(self):
        unread = len(self.buf) - self.pos
        if unread:
            try:
                self.fd.seek(-unread, 1)
            except Exception:
                pass                                          # pipes: the read-ahead is simply dropped with the generator
        self.buf, self.pos = b"", 0

    def close(self):
        self.fd.close()


class _ArenaStream(_BufferedStream):
    """The read-ahead of the BLOCK readers: ONE recycled byte arena filled with ``readinto`` in 64 MiB reads.  The plain
    ``_BufferedStream`` builds a fresh ``bytes`` object per refill (``read`` + ``join``): at ark rates of GB/s the page faults
    of those ever-new buffers and the extra copy cost more than the parsing itself.  Valid bytes are ``arena[pos:end]``;
    ``addr`` is the arena's address for the native scanner / gatherer."""

    BLOCK = 1 << 26

    def __init__(self, fd):
        self.fd = fd
        self.buf = b""                                   # (unused: the arena replaces it)
        self.pos = self.end = 0
        self._alloc(self.BLOCK + (1 << 20))

    def _alloc(self, cap):
        self.arena = bytearray(cap)
        self._pin = (ctypes.c_char * cap).from_buffer(self.arena)          # also keeps the bytearray from being resized
        self.addr = ctypes.addressof(self._pin)
        self.view = memoryview(self.arena)

    def _fill(self, need):
        avail = self.end - self.pos
        if avail >= need:
            return
        if need > len(self.arena):                       # one record larger than the arena: grow, keep the unread bytes
            old = self.view[self.pos:self.end]
            keep = bytes(old)
            del old
            self.view.release()
            del self._pin
            self._alloc(max(need + (1 << 20), 2 * len(self.arena)))
            self.arena[:avail] = keep
            self.pos, self.end = 0, avail
        elif self.pos:
            self.arena[:avail] = self.arena[self.pos:self.end]              # unread tail to the front (usually < one record)
            self.pos, self.end = 0, avail
        readinto = getattr(self.fd, "readinto", None)
        while self.end - self.pos < need or self.end == avail:
            if readinto is not None:
                got = readinto(self.view[self.end:])
            else:
                blk = self.fd.read(len(self.arena) - self.end)
                got = len(blk) if blk else 0
                if got:
                    self.arena[self.end:self.end + got] = blk
            if not got:
                break
            self.end += got
            if self.end == len(self.arena):
                break

    def read(self, n=-1):
        if n is None or n < 0:
            rest = bytes(self.view[self.pos:self.end]) + self.fd.read()
            self.pos = self.end = 0
            return rest
        self._fill(n)
        out = bytes(self.view[self.pos:min(self.pos + n, self.end)])
        self.pos += len(out)
        return out

    def _until(self, sep, keep_sep):
        while True:
            i = self.arena.find(sep, self.pos, self.end)
            if i >= 0:
                out = bytes(self.view[self.pos:i + (1 if keep_sep else 0)])
                self.pos = i + 1
                return out
            before = self.end - self.pos
            self._fill(before + 1)
            if self.end - self.pos == before:            # end of stream
                out = bytes(self.view[self.pos:self.end])
                self.pos = self.end
                return out

    def readline(self):
        return self._until(b"\n", True)

    def read_token(self):
        return self._until(b" ", False)

    def detach(self):
        unread = self.end - self.pos
        if unread:
            try:
                self.fd.seek(-unread, 1)
            except Exception:
                pass                                          # pipes: the read-ahead is simply dropped with the generator
        self.pos = self.end = 0


def _read_exact(fd, n):
    """fd.read(n) that tolerates short reads from pipes; returns fewer bytes only at EOF."""
    buf = fd.read(n)
    if buf is None:
        buf = b""
    if len(buf) == n or not buf:
        return buf
    parts = [buf]
    got = len(buf)
    while got < n:
        more = fd.read(n - got)
        if not more:
            break
        parts.append(more)
        got += len(more)
    return b"".join(parts)


# ------------------------------------------------------------------------------------------------
# keys
# ------------------------------------------------------------------------------------------------
def read_key(fd):
    """Next utterance key of an ark stream, or None at end of stream."""
    if isinstance(fd, _BufferedStream):
        key = fd.read_token().decode().strip()
    else:
        chars = []
        while True:
            ch = fd.read(1)
            if not ch or ch == b" ":
                break
            chars.append(ch)
        key = b"".join(chars).decode().strip()
    if key == "":
        return None
    assert _KEY_OK.match(key) is not None, "malformed key %r" % key
    return key


# ------------------------------------------------------------------------------------------------
# float vectors
# ------------------------------------------------------------------------------------------------
def _read_vec_flt_binary(fd):
    tag = _read_exact(fd, 3).decode()
    if tag == "FV ":
        dt = np.dtype("<f4")
    elif tag == "DV ":
        dt = np.dtype("<f8")
    else:
        raise UnknownVectorHeader("The header contained '%s'" % tag)
    assert _read_exact(fd, 1) == b"\x04"
    (dim,) = struct.unpack("<i", _read_exact(fd, 4))
    return np.frombuffer(_read_exact(fd, dim * dt.itemsize), dtype=dt)


def read_vec_flt(file_or_fd):
    """One Kaldi float vector, binary ('FV '/'DV ') or text ('[ 1 2 3 ]')."""
    fd = open_or_fd(file_or_fd)
    try:
        flag = _read_exact(fd, 2)
        if flag == b"\x00B":
            return _read_vec_flt_binary(fd)
        toks = (flag + fd.readline()).decode().strip().split()
        toks = [t for t in toks if t not in ("[", "]")]
        return np.array(toks, dtype=float)
    finally:
        if fd is not file_or_fd:
            fd.close()


def read_vec_flt_ark(file_or_fd):
    """Generator of (key, vector) over an ark file / stream."""
    raw = open_or_fd(file_or_fd)
    fd = raw if isinstance(raw, _BufferedStream) else _BufferedStream(raw)
    try:
        key = read_key(fd)
        while key:
            yield key, read_vec_flt(fd)
            key = read_key(fd)
    finally:
        if raw is not file_or_fd:
            raw.close()
        elif fd is not raw:
            fd.detach()


def read_vec_flt_scp(file_or_fd):
    """Generator of (key, vector) following a Kaldi scp."""
    fd = open_or_fd(file_or_fd)
    try:
        for line in fd:
            if isinstance(line, bytes):
                line = line.decode()
            key, rxfile = line.strip("\n").split(" ", 1)
            yield key, read_vec_flt(rxfile.strip())
    finally:
        if fd is not file_or_fd:
            fd.close()


def _write_header(fd, key, tag):
    if key != "":
        fd.write((key + " ").encode())
        if hasattr(fd, "note_key"):
            fd.note_key(key)              # TableWriter: the scp offset points just past "key "
    fd.write(b"\x00B" + tag)


class TableWriter(object):
    """Write-only stream producing an ark file AND its scp index (``key path:offset`` per record),
    i.e. what the reference obtains by piping into Kaldi's ``copy-vector ark:- ark,scp:A,S``
    (local/tf/extract_xvectors.sh:74-88) -- without needing the Kaldi binary.  ``scp_ark_name`` is the
    ark path to print inside the scp (defaults to ``ark_path``)."""

    mode = "wb"

    def __init__(self, ark_path, scp_path, scp_ark_name=None):
        self._ark = open(ark_path, "wb")
        self._scp = open(scp_path, "wt")
        self._name = scp_ark_name or ark_path
        self._pos = 0

    def write(self, data):
        self._ark.write(data)
        self._pos += len(data)

    def note_key(self, key):
        self._scp.write("%s %s:%d\n" % (key, self._name, self._pos))

    def write_records(self, keys, key_bytes, bodies):
        """Many records at once: record i is ``key_bytes[i] + bodies[i]`` in the ark and its scp line points just past the
        key (the offset Kaldi prints), one ``write`` per file."""
        lines, pos = [], self._pos
        for k, kb, body in zip(keys, key_bytes, bodies):
            pos += len(kb)
            lines.append("%s %s:%d\n" % (k, self._name, pos))
            pos += len(body)
        self._ark.write(b"".join(x for pair in zip(key_bytes, bodies) for x in pair))
        self._scp.write("".join(lines))
        self._pos = pos

    def write_uniform_records(self, keys, key_len, blob):
        """``len(keys)`` records of equal size already serialised in ``blob``; every key takes ``key_len`` bytes (incl. the
        separating space)."""
        size = len(blob) // max(len(keys), 1)
        pos = self._pos + key_len
        lines = []
        for k in keys:
            lines.append("%s %s:%d\n" % (k, self._name, pos))
            pos += size
        self._ark.write(blob)
        self._scp.write("".join(lines))
        self._pos += len(blob)

    def native_targets(self):
        """(ark fd, scp fd, ark name in the scp, ark length so far) for xv_vec_records_write_fd, the Python-side buffers flushed;
        ``advance(new_length)`` afterwards."""
        self._ark.flush()
        self._scp.flush()
        return self._ark.fileno(), self._scp.fileno(), self._name, self._pos

    def advance(self, pos):
        self._pos = pos

    def close(self):
        self._ark.close()
        self._scp.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


class KeyRange(object):
    """Keys as they lie in some text (an scp table's, or keys joined for the purpose): ``klen[i]`` bytes at ``buf[off[i]]``.  A
    sequence of str -- decoded on first use, which a rank needs for its OWN shard only; the writer of a sharded job's table hands
    ``buf / off / klen`` of the other ranks' keys to the native record writer as they are (1 M keys never become Python objects)."""

    def __init__(self, buf, off, klen):
        self.buf, self.off, self.klen, self._list = buf, np.ascontiguousarray(off, np.int64), np.ascontiguousarray(klen, np.int32), None

    @classmethod
    def from_keys(cls, keys):
        enc = [k.encode() for k in keys]
        klen = np.fromiter((len(e) for e in enc), np.int32, len(enc))
        off = np.zeros(len(enc), np.int64)
        if len(enc) > 1:
            np.cumsum(klen[:-1].astype(np.int64) + 1, out=off[1:])
        out = cls(np.frombuffer(b"\n".join(enc), np.uint8), off, klen)
        out._list = list(keys)
        return out

    def tolist(self):
        if self._list is None:
            if len(self.off) == 0:
                self._list = []
            else:
                lo = int(self.off.min())
                span = self.buf[lo:int((self.off + self.klen).max())].tobytes()
                self._list = [span[o:o + l].decode() for o, l in zip((self.off - lo).tolist(), self.klen.tolist())]
        return self._list

    def __len__(self):
        return len(self.off)

    def __iter__(self):
        return iter(self.tolist())

    def __getitem__(self, i):
        if isinstance(i, (int, np.integer)):
            return self.tolist()[i]
        sub = KeyRange(self.buf, self.off[i], self.klen[i])            # slices, index arrays, boolean masks
        if self._list is not None and isinstance(i, slice):
            sub._list = self._list[i]
        return sub

    def __eq__(self, other):
        return self.tolist() == (other.tolist() if isinstance(other, KeyRange) else list(other))

    def __ne__(self, other):
        return not self == other

    __hash__ = None


class ScpText(object):
    """The text of an scp table and where its lines and keys lie in it (xv_scp_line_index; without the host library, or for a text
    that str.splitlines would cut differently, the same arrays from the Python split).  What a rank of a sharded job needs of a 1 M
    line table is its line COUNT (the cuts are by line number: extract_embedding._scp_shard), the lines of its own range and, on the
    rank that writes, the other ranges' keys as bytes."""

    def __init__(self, rxfilename):
        fid = open_or_fd(rxfilename, "rb")
        try:
            raw = fid.read()
        finally:
            fid.close()
        self.buf = np.frombuffer(raw, np.uint8)
        lib = _host_lib()
        n = -2
        if lib is not None and hasattr(lib, "xv_scp_line_index"):
            n = int(lib.xv_scp_line_index(self.buf.ctypes.data if len(raw) else None, len(raw), None, None, None, 0))
        if n >= 0:
            self.start, self.end, self.klen = np.empty(n, np.int64), np.empty(n, np.int64), np.empty(n, np.int32)
            if n:
                lib.xv_scp_line_index(self.buf.ctypes.data, len(raw), self.start.ctypes.data, self.end.ctypes.data, self.klen.ctypes.data, n)
            self.native = True
            return
        # the documented rule, in Python: str.splitlines, blank lines dropped, key = up to the first whitespace
        lines = [ln.strip() for ln in raw.decode().splitlines() if ln.strip()]
        enc = [ln.encode() for ln in lines]
        self.buf = np.frombuffer(b"\n".join(enc), np.uint8)
        size = np.fromiter((len(e) for e in enc), np.int64, len(enc))
        self.start = np.zeros(len(enc), np.int64)
        if len(enc) > 1:
            np.cumsum(size[:-1] + 1, out=self.start[1:])
        self.end = self.start + size
        self.klen = np.fromiter((len(ln.split(None, 1)[0].encode()) for ln in lines), np.int32, len(lines))
        self.native = False

    def __len__(self):
        return len(self.start)

    def lines(self, lo, hi):
        """Lines [lo, hi) as str, each with its newline (what MatScp / VecScp take as 'the table's lines, already read')."""
        if hi <= lo:
            return []
        text = self.buf[int(self.start[lo]):int(self.end[hi - 1])].tobytes().decode()
        return [ln + "\n" for ln in text.split("\n") if ln and not ln.isspace()]

    def keys(self, lo=0, hi=None):
        hi = len(self) if hi is None else hi
        return KeyRange(self.buf, self.start[lo:hi], self.klen[lo:hi])


def write_vec_flt(file_or_fd, v, key=""):
    """Write a binary Kaldi vector: ``key␠`` + ``\\0B`` + ``FV␠``|``DV␠`` + ``\\4`` + uint32 dim +
    little-endian payload."""
    fd = open_or_fd(file_or_fd, mode="wb")
    try:
        v = np.asarray(v)
        if v.dtype == np.float32:
            tag = b"FV "
        elif v.dtype == np.float64:
            tag = b"DV "
        else:
            raise UnsupportedDataType("'%s', please use 'float32' or 'float64'" % v.dtype)
        _write_header(fd, key, tag)
        fd.write(b"\x04" + struct.pack("<I", v.shape[0]))
        fd.write(np.ascontiguousarray(v).astype(v.dtype.newbyteorder("<"), copy=False).tobytes())
    finally:
        if fd is not file_or_fd:
            fd.close()


def _write_vec_records_native(fd, keys, mat, emitted):
    """The records of ``write_vec_flt_batch`` by xv_vec_records_write_fd (csrc/xv_host.cpp): straight from ``mat``'s rows (any row
    stride: the gathered ``[emitted? | x-vector]`` block is read in place) into the ark file and, for a TableWriter, its scp.  False when
    this sink / these arrays are not its case (the caller then serialises in Python: same bytes)."""
    lib = _host_lib()
    if lib is None or not hasattr(lib, "xv_vec_records_write_fd") or mat is None or mat.dtype != np.float32 or mat.shape[1] == 0 or \
            mat.strides[1] != 4 or mat.strides[0] % 4 or mat.strides[0] < 0 or not mat.dtype.isnative or np.little_endian is False:
        return False
    table = hasattr(fd, "native_targets")
    try:
        if table:
            ark_fd, scp_fd, name, pos = fd.native_targets()
        elif isinstance(fd, io.BufferedWriter) and isinstance(fd.raw, io.FileIO):
            fd.flush()
            ark_fd, scp_fd, name, pos = fd.fileno(), -1, "", 0
        else:
            return False
    except (OSError, ValueError, io.UnsupportedOperation):
        return False
    kr = keys if isinstance(keys, KeyRange) else KeyRange.from_keys(keys)
    mask = None if emitted is None else np.ascontiguousarray(emitted, np.uint8)
    new_pos = ctypes.c_int64(0)
    got = lib.xv_vec_records_write_fd(ark_fd, scp_fd, kr.buf.ctypes.data if len(kr.buf) else None, kr.off.ctypes.data, kr.klen.ctypes.data,
                                      len(kr), mat.ctypes.data, mat.shape[1], mat.strides[0] // 4,
                                      None if mask is None else mask.ctypes.data, name.encode(), pos, ctypes.byref(new_pos))
    if got < 0:
        raise IOError(-int(got), "write_vec_flt_batch: %s" % os.strerror(-int(got)))
    if table:
        fd.advance(int(new_pos.value))
    return True


def write_vec_flt_batch(file_or_fd, keys, vecs, emitted=None):
    """Write many float32 vectors as consecutive binary records (same bytes as write_vec_flt per key) with one ``write``
    call per batch.  ``vecs``: a float32 ``[n, D]`` array or a sequence of float32 vectors.  Equal-length vectors are
    serialised without per-record NumPy calls (and without any per-record Python when the keys have one length too).
    ``emitted`` (bool [n], optional): only these rows are written (a sharded job's gathered block carries a row per input
    utterance).  Into a file or a TableWriter the records are assembled by the host library from the rows as they lie."""
    if hasattr(file_or_fd, "write_vectors"):                 # an in-memory sink (extract_embedding.py, sharded mode)
        if emitted is not None:
            emitted = np.asarray(emitted, bool)
            keys, vecs = [k for k, ok in zip(keys, emitted.tolist()) if ok], np.asarray(vecs)[emitted]
        file_or_fd.write_vectors(keys, vecs)
        return
    fd = open_or_fd(file_or_fd, mode="wb")
    try:
        if len(keys) and isinstance(vecs, np.ndarray) and vecs.ndim == 2 and vecs.shape[0] == len(keys) and \
                os.environ.get("XVECTOR_NATIVE_WRITER", "1") != "0" and _write_vec_records_native(fd, keys, vecs, emitted):
            return
        if emitted is not None:
            emitted = np.asarray(emitted, bool)
            keys, vecs = [k for k, ok in zip(keys, emitted.tolist()) if ok], np.asarray(vecs)[emitted]
        elif isinstance(keys, KeyRange):
            keys = keys.tolist()
        n = len(keys)
        if n == 0:
            return
        mat = None
        if isinstance(vecs, np.ndarray) and vecs.ndim == 2:
            mat = vecs
        elif len({np.shape(v) for v in vecs}) == 1 and np.ndim(vecs[0]) == 1:
            mat = np.stack(vecs)
        if mat is not None and mat.dtype != np.float32:
            raise UnsupportedDataType("'%s', write_vec_flt_batch expects float32" % mat.dtype)
        if mat is None:                                      # ragged dimensions: one record at a time
            key_bytes, bodies = [], []
            for k, v in zip(keys, vecs):
                v = np.asarray(v)
                if v.dtype != np.float32:
                    raise UnsupportedDataType("'%s', write_vec_flt_batch expects float32" % v.dtype)
                key_bytes.append((k + " ").encode() if k != "" else b"")
                bodies.append(b"\x00BFV \x04" + struct.pack("<I", v.shape[0]) + np.ascontiguousarray(v).astype("<f4", copy=False).tobytes())
            blob = None
        else:
            assert mat.shape[0] == n
            head = b"\x00BFV \x04" + struct.pack("<I", mat.shape[1])
            payload = np.ascontiguousarray(mat).astype("<f4", copy=False)
            key_bytes = [(k + " ").encode() if k != "" else b"" for k in keys]
            klen = len(key_bytes[0])
            if all(len(kb) == klen for kb in key_bytes):
                # every record has the same size: assemble them as the rows of one uint8 matrix
                rec = np.empty((n, klen + len(head) + 4 * mat.shape[1]), dtype=np.uint8)
                if klen:
                    rec[:, :klen] = np.frombuffer(b"".join(key_bytes), dtype=np.uint8).reshape(n, klen)
                rec[:, klen:klen + len(head)] = np.frombuffer(head, dtype=np.uint8)
                rec[:, klen + len(head):] = payload.view(np.uint8).reshape(n, 4 * mat.shape[1])
                blob = rec.tobytes()
                bodies = None
            else:
                raw, step = payload.tobytes(), 4 * mat.shape[1]
                bodies = [head + raw[i * step:(i + 1) * step] for i in range(n)]
                blob = None
        if hasattr(fd, "write_records"):                     # TableWriter: ark + scp index
            if bodies is None:
                fd.write_uniform_records(keys, klen, blob)
            else:
                fd.write_records(keys, key_bytes, bodies)
        else:
            fd.write(blob if blob is not None else b"".join(x for pair in zip(key_bytes, bodies) for x in pair))
    finally:
        if fd is not file_or_fd:
            fd.close()


# ------------------------------------------------------------------------------------------------
# float matrices
# ------------------------------------------------------------------------------------------------
_U16_STEP = np.float32(1.52590218966964e-05)     # 1/65535, the constant Kaldi's compressed-matrix uses


def _read_compressed_mat(fd, fmt):
    """Kaldi CompressedMatrix.  'CM ' = per-column percentile headers + column-major uint8 with a
    3-segment piecewise-linear decode; 'CM2' = row-major uint16; 'CM3' = row-major uint8.
    All arithmetic in float32, as Kaldi (and the reference under NumPy>=2) evaluates it."""
    if fmt != "CM ":
        # Kaldi writes every token with a trailing space (WriteToken): "CM " is complete in three bytes, "CM2" / "CM3" are not
        if _read_exact(fd, 1) != b" ":
            raise UnknownMatrixHeader("The header contained '%s' without the separating space" % fmt)
    hdr = _read_exact(fd, 16)
    gmin, grange = np.frombuffer(hdr[:8], dtype="<f4")
    rows, cols = (int(v) for v in np.frombuffer(hdr[8:], dtype="<i4"))
    if rows < 0 or cols < 0:
        # (the reference would go on with read(-n), i.e. swallow the rest of the stream as this matrix's bytes)
        raise BadInputFormat("compressed matrix header with negative dimensions: %d x %d" % (rows, cols))
    if fmt == "CM ":
        pct = np.frombuffer(_read_exact(fd, cols * 8), dtype="<u2").reshape(cols, 4)
        q = (gmin + grange * _U16_STEP * pct.astype(np.float32)).astype(np.float32)   # [cols,4]
        p0, p25, p75, p100 = (q[:, i:i + 1] for i in range(4))
        u = np.frombuffer(_read_exact(fd, cols * rows), dtype=np.uint8).reshape(cols, rows)
        uf = u.astype(np.float32)
        lo = p0 + (p25 - p0) / np.float32(64.) * uf
        mid = p25 + (p75 - p25) / np.float32(128.) * (uf - np.float32(64.))
        hi = p75 + (p100 - p75) / np.float32(63.) * (uf - np.float32(192.))
        out = np.where(u <= 64, lo, np.where(u <= 192, mid, hi)).astype(np.float32)
        return out.T
    if fmt == "CM2":
        u = np.frombuffer(_read_exact(fd, rows * cols * 2), dtype="<u2").reshape(rows, cols)
        return (gmin + grange * _U16_STEP * u.astype(np.float32)).astype(np.float32)
    if fmt == "CM3":
        u = np.frombuffer(_read_exact(fd, rows * cols), dtype=np.uint8).reshape(rows, cols)
        return (gmin + grange * np.float32(1.0 / 255.0) * u.astype(np.float32)).astype(np.float32)
    raise UnknownMatrixHeader("The header contained '%s'" % fmt)


def _read_mat_binary(fd):
    tag = _read_exact(fd, 3).decode()
    if tag.startswith("CM"):
        return _read_compressed_mat(fd, tag)
    if tag == "FM ":
        dt = np.dtype("<f4")
    elif tag == "DM ":
        dt = np.dtype("<f8")
    else:
        raise UnknownMatrixHeader("The header contained '%s'" % tag)
    dims = _read_exact(fd, 10)
    _, rows, _, cols = struct.unpack("<bibi", dims)
    payload = _read_exact(fd, rows * cols * dt.itemsize)
    return np.frombuffer(payload, dtype=dt).reshape(rows, cols)


def _read_mat_ascii(fd):
    rows = []
    while True:
        line = fd.readline()
        if isinstance(line, bytes):
            line = line.decode()
        if len(line) == 0:
            raise BadInputFormat            # unexpected end of stream
        toks = line.split()
        if not toks:
            continue
        closing = toks[-1] == "]"
        if closing:
            toks = toks[:-1]
        rows.append(np.array(toks, dtype="float32"))
        if closing:
            return np.vstack(rows)


def read_mat(file_or_fd):
    """One Kaldi matrix, binary (FM/DM/CM*) or text."""
    fd = open_or_fd(file_or_fd)
    try:
        flag = _read_exact(fd, 2)
        if flag == b"\x00B":
            return _read_mat_binary(fd)
        assert flag == b" ["
        return _read_mat_ascii(fd)
    finally:
        if fd is not file_or_fd:
            fd.close()


_HOST_LIB = False          # False = not tried yet, None = unavailable


def _host_lib():
    """libxvector_host.so (csrc/xv_host.cpp): native scan of binary float-matrix records.  Optional: without it the
    generic per-record Python reader below is used."""
    global _HOST_LIB
    if _HOST_LIB is False:
        _HOST_LIB = None
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libxvector_host.so")
        if os.path.exists(path) and os.environ.get("XVECTOR_NO_HOST_LIB") != "1":
            try:
                lib = ctypes.CDLL(path)
                lib.xv_ark_scan_fm.restype = ctypes.c_int
                lib.xv_ark_scan_fm.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int] + \
                    [ctypes.c_void_p] * 5 + [ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_int)]
                if hasattr(lib, "xv_copy_bytes"):
                    lib.xv_copy_bytes.restype = None
                    lib.xv_copy_bytes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
                if hasattr(lib, "xv_ark_scan_fv"):
                    lib.xv_ark_scan_fv.restype = ctypes.c_int
                    lib.xv_ark_scan_fv.argtypes = lib.xv_ark_scan_fm.argtypes
                if hasattr(lib, "xv_ark_keys"):
                    lib.xv_ark_keys.restype = ctypes.c_int64
                    lib.xv_ark_keys.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                                ctypes.c_int64]
                if hasattr(lib, "xv_ark_gather_fm"):
                    lib.xv_ark_gather_fm.restype = ctypes.c_int64
                    lib.xv_ark_gather_fm.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                     ctypes.c_void_p]
                if hasattr(lib, "xv_ark_index_fd"):
                    lib.xv_ark_index_fd.restype = ctypes.c_int64
                    lib.xv_ark_index_fd.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
                                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                                    ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                                    ctypes.POINTER(ctypes.c_int)]
                if hasattr(lib, "xv_ark_decode_cm"):
                    lib.xv_ark_decode_cm.restype = ctypes.c_int
                    lib.xv_ark_decode_cm.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p,
                                                     ctypes.c_size_t] + [ctypes.c_void_p] * 5 + \
                        [ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_int), ctypes.c_int]
                if hasattr(lib, "xv_scp_line_index"):
                    lib.xv_scp_line_index.restype = ctypes.c_int64
                    lib.xv_scp_line_index.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                      ctypes.c_int64]
                if hasattr(lib, "xv_vec_records_write_fd"):
                    lib.xv_vec_records_write_fd.restype = ctypes.c_int64
                    lib.xv_vec_records_write_fd.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                            ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p,
                                                            ctypes.c_char_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
                _HOST_LIB = lib
            except OSError:
                _HOST_LIB = None
    return _HOST_LIB


_SCAN_MAX = 8192


def _decode_keys(lib, addr, buf, key_off, key_len):
    """Keys of the scanned records ``key_off`` / ``key_len`` (arrays) inside the block at ``addr`` (``buf`` = the same bytes as
    an indexable object): one native pass + one decode/split when the host library has ``xv_ark_keys``, per key otherwise or
    when a key is malformed (so that the assertion names it)."""
    n = len(key_off)
    if n and hasattr(lib, "xv_ark_keys"):
        out = np.empty(int(key_len.sum()) + n, np.uint8)
        ko, kl = np.ascontiguousarray(key_off, np.int64), np.ascontiguousarray(key_len, np.int32)
        w = lib.xv_ark_keys(addr, ko.ctypes.data, kl.ctypes.data, n, out.ctypes.data, len(out))
        if w >= 0:
            return out[:w].tobytes().decode().split("\n")
    keys = [bytes(buf[ko:ko + kl]).decode().strip() for ko, kl in zip(key_off.tolist(), key_len.tolist())]
    bad = [k for k in keys if _KEY_OK.match(k) is None]
    assert not bad, "malformed key %r" % bad[0]
    return keys


def _scan_fm_records(fd, lib):
    """Yield (key, matrix) for the run of binary 'FM ' records at the stream position, using the native scanner on
    whole buffered blocks; returns when the next record is something else (or at end of stream)."""
    key_off = np.empty(_SCAN_MAX, np.int64); key_len = np.empty(_SCAN_MAX, np.int32)
    data_off = np.empty(_SCAN_MAX, np.int64); rows = np.empty(_SCAN_MAX, np.int32); cols = np.empty(_SCAN_MAX, np.int32)
    nxt, stop = ctypes.c_size_t(0), ctypes.c_int(0)
    want = 1
    while True:
        fd._fill(want)
        buf, pos = fd.buf, fd.pos
        if len(buf) == pos:
            return                                              # end of stream
        n = lib.xv_ark_scan_fm(buf, pos, len(buf), _SCAN_MAX, key_off.ctypes.data, key_len.ctypes.data, data_off.ctypes.data,
                               rows.ctypes.data, cols.ctypes.data, ctypes.byref(nxt), ctypes.byref(stop))
        kos, kls, dos = key_off[:n].tolist(), key_len[:n].tolist(), data_off[:n].tolist()
        rs, cs = rows[:n].tolist(), cols[:n].tolist()
        for ko, kl, do, r, c in zip(kos, kls, dos, rs, cs):
            key = buf[ko:ko + kl].decode().strip()
            assert _KEY_OK.match(key) is not None, "malformed key %r" % key
            fd.pos = do + r * c * 4                             # consumed up to here if the caller stops now
            yield key, np.frombuffer(buf, dtype="<f4", count=r * c, offset=do).reshape(r, c)
        fd.pos = nxt.value
        if stop.value == 1:
            return                                              # a different record type follows
        if stop.value == 0:
            avail = len(buf) - fd.pos
            if n == 0:
                want = max(2 * avail, fd.BLOCK)                 # one record larger than what is buffered: grow
            else:
                want = avail + 1
            before = avail
            fd._fill(want)
            if len(fd.buf) - fd.pos == before:                  # nothing more to read
                return


def read_mat_ark_blocks(file_or_fd, alloc=None):
    """Generator of (keys, feats[sum T, F] float32, offsets[n+1]) over an ark stream: utterance i of a block is
    ``feats[offsets[i]:offsets[i+1]]``.  Runs of binary float matrices with one column count come out as ONE block per
    scanner pass (<= 8192 records or one 64 MiB arena) whose payloads are gathered by a single native, GIL-free call -- the
    per-utterance Python work of ``read_mat_ark`` (generator switch, frombuffer, copy) is what bounded the ark->ark rate.
    ``alloc(rows, cols)`` may supply the float32 storage of a block (callers that recycle their buffers; default: a fresh
    array).  Any other record (or a stream without the host library) is returned as a one-utterance block via the generic
    reader."""
    return _read_ark_blocks(file_or_fd, "xv_ark_scan_fm", lambda fd: np.ascontiguousarray(read_mat(fd), dtype=np.float32), alloc)


def read_vec_flt_ark_blocks(file_or_fd, alloc=None):
    """The same for float vectors (e.g. a VAD table): generator of (keys, values[sum dim] float32, offsets[n+1])."""
    for keys, vals, offsets in _read_ark_blocks(file_or_fd, "xv_ark_scan_fv",
                                                lambda fd: np.ascontiguousarray(read_vec_flt(fd), dtype=np.float32).reshape(-1, 1),
                                                alloc):
        yield keys, vals.reshape(-1), offsets


def _read_ark_blocks(file_or_fd, scan_name, read_one, alloc=None):
    raw = open_or_fd(file_or_fd)
    lib = _host_lib()
    if lib is not None and not (hasattr(lib, "xv_ark_gather_fm") and hasattr(lib, scan_name)):
        lib = None
    if isinstance(raw, _BufferedStream):
        fd = raw
    else:
        fd = _ArenaStream(raw) if lib is not None else _BufferedStream(raw)
    fast = lib is not None and isinstance(fd, _ArenaStream)
    if alloc is None:
        alloc = lambda r, c: np.empty((r, c), np.float32)
    try:
        if fast:
            scan = getattr(lib, scan_name)
            key_off = np.empty(_SCAN_MAX, np.int64); key_len = np.empty(_SCAN_MAX, np.int32)
            data_off = np.empty(_SCAN_MAX, np.int64); rows = np.empty(_SCAN_MAX, np.int32); cols = np.empty(_SCAN_MAX, np.int32)
            nxt, stop = ctypes.c_size_t(0), ctypes.c_int(0)
        while True:
            if fast:
                want = 1
                while True:
                    fd._fill(want)
                    if fd.end == fd.pos:
                        break
                    n = scan(fd.addr, fd.pos, fd.end, _SCAN_MAX, key_off.ctypes.data, key_len.ctypes.data,
                             data_off.ctypes.data, rows.ctypes.data, cols.ctypes.data, ctypes.byref(nxt),
                             ctypes.byref(stop))
                    arena = fd.arena
                    i0 = 0
                    while i0 < n:                               # split the pass where the column count changes
                        c = int(cols[i0])
                        same = np.flatnonzero(cols[i0:n] != c)
                        i1 = i0 + (int(same[0]) if len(same) else n - i0)
                        keys = _decode_keys(lib, fd.addr, arena, key_off[i0:i1], key_len[i0:i1])
                        offsets = np.zeros(i1 - i0 + 1, np.int64)
                        np.cumsum(rows[i0:i1], out=offsets[1:])
                        feats = alloc(int(offsets[-1]), c)
                        assert feats.dtype == np.float32 and feats.shape == (int(offsets[-1]), c) and feats.flags.c_contiguous
                        lib.xv_ark_gather_fm(fd.addr, data_off[i0:i1].ctypes.data, rows[i0:i1].ctypes.data, c, i1 - i0,
                                             feats.ctypes.data)
                        fd.pos = int(data_off[i1 - 1]) + int(rows[i1 - 1]) * c * 4
                        yield keys, feats, offsets
                        i0 = i1
                    fd.pos = nxt.value
                    if stop.value == 1:
                        break                                   # a different record type follows
                    if stop.value == 0:
                        avail = fd.end - fd.pos
                        want = max(2 * avail, 1 << 22) if n == 0 else avail + 1
                        fd._fill(want)
                        if fd.end - fd.pos == avail:            # nothing more to read
                            break
            key = read_key(fd)                                  # generic path: one record of any supported type
            if not key:
                break
            m = read_one(fd)
            yield [key], m, np.array([0, m.shape[0]], np.int64)
    finally:
        if raw is not file_or_fd:
            raw.close()
        elif fd is not raw:
            fd.detach()


# ------------------------------------------------------------------------------------------------
# in-place windows: the utterances stay where the stream was read to
# ------------------------------------------------------------------------------------------------
class ArkArena(object):
    """A long-lived read buffer of the in-place reader (``scan_mat_ark_windows``): the stream is read straight into it
    (``readinto``) and the matrices of the records found there are used where they lie.  Anonymous mmap with transparent huge
    pages where the kernel offers them: the first touch of a fresh arena is a page-fault storm otherwise (47 k faults per
    192 MB), and arenas are recycled process-wide (``arena_acquire`` / ``arena_release``) so that it is paid once."""

    def __init__(self, nbytes):
        import mmap
        self.buf = mmap.mmap(-1, int(nbytes))
        if hasattr(self.buf, "madvise") and hasattr(mmap, "MADV_HUGEPAGE"):
            try:
                self.buf.madvise(mmap.MADV_HUGEPAGE)
            except OSError:
                pass
        self._pin = (ctypes.c_char * len(self.buf)).from_buffer(self.buf)
        self.addr = ctypes.addressof(self._pin)
        self.view = memoryview(self.buf)

    def __len__(self):
        return len(self.buf)


_ARENA_FREE = {}          # size -> idle arenas of this process (at most _ARENA_KEEP per size)
_ARENA_KEEP = 6
_ARENA_LOCK = threading.Lock()


def arena_acquire(nbytes):
    with _ARENA_LOCK:
        free = _ARENA_FREE.get(int(nbytes))
        if free:
            return free.pop()
    return ArkArena(nbytes)


def arena_release(arena):
    with _ARENA_LOCK:
        free = _ARENA_FREE.setdefault(len(arena), [])
        if len(free) < _ARENA_KEEP and all(a is not arena for a in free):
            free.append(arena)


class ArkMats(object):
    """The float32 matrices of a window as a sequence that is NOT materialised: row counts (``lengths``) and row-0
    addresses (``addrs``) are arrays, element ``i`` becomes a NumPy view only when somebody indexes it.  ``holders`` are the
    objects (arenas / arrays) that own the memory: whoever keeps the window keeps them."""

    def __init__(self):
        self._addr, self._rows, self._cols, self.holders, self._count = [], [], [], [], [0]
        self.lengths = np.zeros(0, np.int64)
        self.addrs = np.zeros(0, np.uint64)

    def add(self, addr, rows, cols, holder):
        self._addr.append(np.asarray(addr, np.uint64)); self._rows.append(np.asarray(rows, np.int64))
        self._cols.append(int(cols)); self.holders.append(holder)
        self._count.append(self._count[-1] + len(self._rows[-1]))
        self.lengths = self._rows[0] if len(self._rows) == 1 else np.concatenate(self._rows)
        self.addrs = self._addr[0] if len(self._addr) == 1 else np.concatenate(self._addr)

    def uniform_cols(self):
        """The common column count, or None when the pieces disagree (or there is nothing)."""
        return self._cols[0] if self._cols and all(c == self._cols[0] for c in self._cols) else None

    def frames(self):
        return int(self.lengths.sum())

    def __len__(self):
        return self._count[-1]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        p = int(np.searchsorted(self._count, i, side="right")) - 1
        j = i - self._count[p]
        holder, r, c = self.holders[p], int(self._rows[p][j]), self._cols[p]
        if isinstance(holder, np.ndarray):
            return holder
        if isinstance(holder, list):
            return holder[j]
        return np.frombuffer(holder.buf, dtype="<f4", count=r * c, offset=int(self._addr[p][j]) - holder.addr).reshape(r, c)

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]


class ArkMapped(object):
    """Holder of the windows of an ark that is used where it already lies (``map_stream``): ``buf`` / ``addr`` as an ArkArena has
    them, so that ``ArkMats`` builds its views the same way; nothing to recycle."""

    def __init__(self, arr):
        self.buf = arr                                   # read-only uint8 array over the BytesIO buffer / the file mapping
        self.addr = int(arr.ctypes.data)

    def __len__(self):
        return int(self.buf.shape[0])


def map_stream(file_or_fd):
    """The unread rest of an IN-MEMORY stream (``io.BytesIO``) as one read-only uint8 array, without a copy (the array keeps the
    stream's bytes object alive; the stream position is moved to the end); None for everything else.  A regular file is NOT
    mapped: walking its page cache through an mmap makes the packer threads fault the pages in 4 KB at a time -- 188 k against
    203 k utt/s for the same ark read into the huge-page arenas (tmpfs, 50 k utterances)."""
    f = file_or_fd
    try:
        if type(f) is io.BytesIO:
            # getvalue() hands out the stream's own bytes object; getbuffer() would first COPY a BytesIO that still shares the
            # bytes it was built from -- 0.2 s for a 1.4 GB ark
            arr = np.frombuffer(f.getvalue(), np.uint8)[f.tell():]
            f.seek(0, 2)
            return arr if len(arr) else None
    except (ValueError, BufferError, io.UnsupportedOperation):
        pass
    return None


class MemStream(object):
    """A read-only binary stream over a uint8 array that is already in memory (the rest of a mapped ark), WITHOUT a copy of it:
    what ``io.BytesIO(arr.tobytes())`` would be, minus the 0.1-0.2 s per GB of the copy.  ``scan_mat_ark_windows`` moves its bytes
    into the arenas with the host library's copy, outside the interpreter lock, as it does for a BytesIO."""

    def __init__(self, arr):
        self.arr = arr
        self.pos = 0

    def tell(self):
        return self.pos

    def seek(self, pos, whence=0):
        self.pos = max(0, min(len(self.arr), int(pos) + (0 if whence == 0 else self.pos if whence == 1 else len(self.arr))))
        return self.pos

    def read(self, n=-1):
        end = len(self.arr) if n is None or n < 0 else min(len(self.arr), self.pos + int(n))
        out = self.arr[self.pos:end].tobytes()
        self.pos = end
        return out

    def readinto(self, b):
        view = memoryview(b).cast("B")
        n = min(len(view), len(self.arr) - self.pos)
        view[:n] = memoryview(self.arr[self.pos:self.pos + n])
        self.pos += n
        return n


def scan_mat_ark_mapped(arr, window_bytes, first_bytes=None, fallback=None):
    """``scan_mat_ark_windows`` over an ark that is already in memory (``map_stream``): the native scanner walks ``arr`` in
    windows of about ``window_bytes`` (the first one ``first_bytes``) and yields ``(keys, addr, rows, cols, holder)`` with the
    matrices where they lie in ``arr`` -- the reader thread copies NOTHING, a window is available as soon as its record headers
    are parsed.  A record the scanner does not take (another type, a truncated tail) ends the mapped walk: the rest goes to
    ``fallback(bytes-like)`` (a generator of the same items; default: the arena reader on a copy of the rest)."""
    lib = _host_lib()
    assert lib is not None and hasattr(lib, "xv_ark_scan_fm"), "scan_mat_ark_mapped needs libxvector_host.so"
    held = ArkMapped(arr)
    base = held.addr
    total = len(held)
    key_off = np.empty(_SCAN_MAX, np.int64); key_len = np.empty(_SCAN_MAX, np.int32)
    data_off = np.empty(_SCAN_MAX, np.int64); rows = np.empty(_SCAN_MAX, np.int32); cols = np.empty(_SCAN_MAX, np.int32)
    nxt, stop = ctypes.c_size_t(0), ctypes.c_int(0)
    pos, win = 0, int(first_bytes or window_bytes)
    while pos < total:
        end = min(total, pos + win)
        keys, a_parts, r_parts, c_now = [], [], [], None
        items = []

        def flush():
            if keys:
                items.append((list(keys), np.concatenate(a_parts) if len(a_parts) > 1 else a_parts[0],
                              np.concatenate(r_parts) if len(r_parts) > 1 else r_parts[0], c_now, held))
                del keys[:], a_parts[:], r_parts[:]
        start = pos
        while pos < end:
            n = lib.xv_ark_scan_fm(base, pos, end, _SCAN_MAX, key_off.ctypes.data, key_len.ctypes.data, data_off.ctypes.data,
                                   rows.ctypes.data, cols.ctypes.data, ctypes.byref(nxt), ctypes.byref(stop))
            i0 = 0
            while i0 < n:                                       # a change of the column count closes the item
                c = int(cols[i0])
                diff = np.flatnonzero(cols[i0:n] != c)
                i1 = i0 + (int(diff[0]) if len(diff) else n - i0)
                if c_now is not None and c != c_now:
                    flush()
                c_now = c
                keys.extend(_decode_keys(lib, base, arr, key_off[i0:i1], key_len[i0:i1]))
                a_parts.append(data_off[i0:i1].astype(np.uint64) + np.uint64(base))
                r_parts.append(rows[i0:i1].copy())
                i0 = i1
            pos = nxt.value
            if stop.value != 2:
                break                                           # 2 = scanner table full: scan on in the same window
        flush()
        for item in items:
            yield item
        if stop.value == 1 or (pos == start and end == total):
            # a record of another type, or a tail that is not a complete record: the generic reader reports / decodes it
            rest = arr[pos:]
            if fallback is None:
                def fallback(b):
                    pool = []

                    def take():
                        pool.append(ArkArena(max(len(b) + 64, 1 << 20)))
                        return pool[-1]
                    return scan_mat_ark_windows(MemStream(b), take)
            for item in fallback(rest):
                yield item
            return
        if pos == start:
            win *= 2                                            # one record larger than the window: look further
        else:
            win = int(window_bytes)


_CM_THREADS = max(1, int(os.environ.get("XVECTOR_CM_THREADS", "0")) or min(4, (os.cpu_count() or 2) // 2))    # decoder threads of xv_ark_decode_cm


def _cm_record_at(view, pos, end):
    """True when the record at ``view[pos:end]`` is a binary compressed matrix of the speech-feature kind ("<key> \\0BCM ")."""
    head = bytes(view[pos:min(end, pos + 4096)])
    sp = head.find(b" ")
    return sp >= 0 and (head[sp + 1:sp + 6] == b"\0BCM " or head[sp + 1:sp + 7] in (b"\0BCM2 ", b"\0BCM3 "))


def scan_mat_ark_windows(file_or_fd, take_arena, first_fill=None, release=None):
    """The in-place form of ``read_mat_ark_blocks``: generator of ``(keys, addr[n] uint64, rows[n] int32, cols, holder)``.
    The stream is read (``readinto``) into arenas that ``take_arena()`` hands out (``ArkArena``; the caller recycles them once it
    is done with the window), the native scanner locates the binary float-matrix records, and NOTHING is copied: ``addr[i]``
    is where row 0 of utterance i lies inside ``holder`` (the arena).  One item per arena and column count; the bytes of a
    record cut off by the arena's end are carried to the front of the next arena.  ``first_fill``: bytes to read into the first
    arena (then doubling): a consumer pipeline starts sooner on a short first window.  Compressed speech-feature matrices
    ("CM ") are decoded by the host library into further arenas (``holder`` = that arena).  Records of any other type are decoded
    by the generic reader and come out one at a time with ``holder`` = their own float32 array.  ``release(arena)``: called for
    an arena that was taken but became the holder of no item (it held nothing but the carried bytes of a record that went
    through the generic reader) -- the consumer never sees such an arena, so it cannot recycle it.  Needs the host library."""
    lib = _host_lib()
    assert lib is not None and hasattr(lib, "xv_ark_scan_fm"), "scan_mat_ark_windows needs libxvector_host.so"
    has_cm = hasattr(lib, "xv_ark_decode_cm")
    raw = open_or_fd(file_or_fd)
    readinto = getattr(raw, "readinto", None)
    # An in-memory stream is copied by the host library, outside the interpreter lock (BytesIO.readinto holds it for the whole
    # memcpy: the model load, the planner and the packer of the other threads would stand still meanwhile).
    # (getvalue() hands out the stream's own bytes object; getbuffer() would first COPY a BytesIO that still shares the bytes it
    # was built from -- 0.2 s for a 1.4 GB ark)
    mem = mem_addr = None
    if type(raw) is io.BytesIO and hasattr(lib, "xv_copy_bytes"):
        mem = raw.getvalue()
        mem_addr = np.frombuffer(mem, dtype=np.uint8).ctypes.data if len(mem) else None
    elif isinstance(raw, MemStream) and hasattr(lib, "xv_copy_bytes"):
        mem = raw.arr
        mem_addr = int(mem.ctypes.data) if len(mem) else None
    key_off = np.empty(_SCAN_MAX, np.int64); key_len = np.empty(_SCAN_MAX, np.int32)
    data_off = np.empty(_SCAN_MAX, np.int64); rows = np.empty(_SCAN_MAX, np.int32); cols = np.empty(_SCAN_MAX, np.int32)
    nxt, stop = ctypes.c_size_t(0), ctypes.c_int(0)
    carry, eof, unread, fill = b"", False, 0, first_fill
    arena, handed = None, [False]
    try:
        while carry or not eof:
            handed = [False]
            arena = take_arena()
            cap = len(arena)
            end = min(len(carry), cap)
            arena.buf[:end] = carry[:end]
            spill = carry[end:]                                  # (only when one record is larger than a whole arena)
            carry = b""
            limit = cap if not fill else min(cap, max(int(fill), end + 64))
            fill = None if not fill or limit == cap else 2 * limit
            while end < limit and not eof and not spill:
                if mem_addr is not None:
                    at = raw.tell()
                    got = min(limit - end, len(mem) - at)
                    if got > 0:
                        lib.xv_copy_bytes(arena.addr + end, mem_addr + at, got)
                        raw.seek(at + got)
                elif readinto is not None:
                    # in slices: a stream object may copy under the interpreter lock, and 72 MB in one call would stall
                    # every other thread of the pipeline for ~15 ms
                    got = readinto(arena.view[end:min(limit, end + (4 << 20))])
                else:
                    blk = raw.read(limit - end)
                    got = len(blk) if blk else 0
                    arena.buf[end:end + got] = blk or b""
                if not got:
                    eof = True
                else:
                    end += got
            unread = end
            pos = 0
            keys, a_parts, r_parts, c_now = [], [], [], None

            def flush(handed=handed):
                """The records collected so far as one item.  An arena is the holder of exactly ONE item (its consumer recycles
                it); the records of a further item of the same arena -- the column count changed inside it -- are copied out."""
                if not keys:
                    return None
                addr = np.concatenate(a_parts) if len(a_parts) > 1 else a_parts[0]
                nrow = np.concatenate(r_parts) if len(r_parts) > 1 else r_parts[0]
                holder = arena
                if handed[0]:
                    holder = [np.frombuffer(arena.buf, dtype="<f4", count=int(r) * c_now, offset=int(a) - arena.addr)
                              .reshape(int(r), c_now).copy() for a, r in zip(addr.tolist(), nrow.tolist())]
                    addr = np.array([m.__array_interface__["data"][0] for m in holder], np.uint64)
                handed[0] = True
                out = (list(keys), addr, nrow, c_now, holder)
                del keys[:], a_parts[:], r_parts[:]
                return out
            while pos < end:
                n = lib.xv_ark_scan_fm(arena.addr, pos, end, _SCAN_MAX, key_off.ctypes.data, key_len.ctypes.data,
                                       data_off.ctypes.data, rows.ctypes.data, cols.ctypes.data, ctypes.byref(nxt), ctypes.byref(stop))
                i0 = 0
                while i0 < n:                                   # a change of the column count closes the item
                    c = int(cols[i0])
                    diff = np.flatnonzero(cols[i0:n] != c)
                    i1 = i0 + (int(diff[0]) if len(diff) else n - i0)
                    if c_now is not None and c != c_now:
                        item = flush()
                        if item:
                            yield item
                    c_now = c
                    keys.extend(_decode_keys(lib, arena.addr, arena.buf, key_off[i0:i1], key_len[i0:i1]))
                    a_parts.append(data_off[i0:i1].astype(np.uint64) + np.uint64(arena.addr))
                    r_parts.append(rows[i0:i1].copy())
                    i0 = i1
                pos = nxt.value
                unread = end - pos
                if stop.value == 2:
                    continue                                    # scanner table full: scan on in the same arena
                if stop.value == 1 and has_cm and _cm_record_at(arena.view, pos, end):
                    # compressed speech-feature matrices (what make_mfcc.sh writes by default): decoded natively, on a few
                    # threads, into arenas of their own -- each the holder of one item; this arena keeps only the bytes
                    item = flush()
                    if item:
                        yield item
                    c_now = None
                    if handed[0]:
                        # this arena already went downstream as the holder of an item and may come back from take_arena() below
                        # while its tail is still needed: the tail goes to the front of the next arena instead
                        stop.value = 0
                        break
                    cm_from = pos
                    while True:
                        darena = take_arena()
                        n = lib.xv_ark_decode_cm(arena.addr, pos, end, _SCAN_MAX, darena.addr, len(darena), key_off.ctypes.data,
                                                 key_len.ctypes.data, data_off.ctypes.data, rows.ctypes.data, cols.ctypes.data,
                                                 ctypes.byref(nxt), ctypes.byref(stop), _CM_THREADS)
                        pos = nxt.value
                        unread = end - pos
                        if n > 0:
                            dkeys = _decode_keys(lib, arena.addr, arena.buf, key_off[:n], key_len[:n])
                            yield dkeys, data_off[:n].astype(np.uint64) + np.uint64(darena.addr), rows[:n].copy(), int(cols[0]), darena
                        elif release is not None:
                            release(darena)
                        if stop.value not in (2, 3, 4) or (n == 0 and stop.value == 3):
                            break                               # (a single matrix larger than an arena: the generic reader takes it)
                    if stop.value == 3:
                        stop.value = 1                          # ONE record through the generic reader below
                    elif stop.value == 1 and pos > cm_from:
                        continue                                # another record type follows: back to the float-matrix scanner
                    # (stop == 1 with nothing decoded: a "CM" record the decoder refuses -- negative dimensions, say.  Going back to
                    # the scanner would stop at the same byte again, for ever, an arena per turn: the generic reader takes the record
                    # and raises on it as it does for any malformed matrix)
                break
            item = flush()
            if item:
                yield item
            rest = bytes(arena.view[pos:end]) + spill
            if rest:
                if stop.value == 1 or (eof and not spill) or (pos == 0 and end == cap):
                    # a record of another type, a truncated tail, or a record larger than a whole arena: ONE record through
                    # the generic reader, whose read-ahead then becomes the carry of the next arena
                    bs = _BufferedStream(raw)
                    bs.buf, bs.pos = rest, 0
                    key = read_key(bs)
                    if key:
                        m = np.ascontiguousarray(read_mat(bs), dtype=np.float32)
                        carry = bs.buf[bs.pos:]
                        unread = len(carry)
                        yield [key], np.array([m.__array_interface__["data"][0]], np.uint64), np.array([m.shape[0]], np.int32), \
                            (m.shape[1] if m.ndim == 2 else 0), m
                    else:
                        carry, unread = b"", 0
                else:
                    carry = rest                                 # an incomplete record: goes to the front of the next arena
                    unread = len(carry)
            if not handed[0] and release is not None:
                release(arena)                                   # nobody downstream holds it: back to the pool from here
            arena = None
    finally:
        mem = None
        if arena is not None and not handed[0] and release is not None:
            release(arena)                                       # (an error while this arena was being read or scanned)
        if raw is not file_or_fd:
            raw.close()
        elif unread:
            try:
                raw.seek(-unread, 1)
            except Exception:
                pass


def is_regular_file(stream):
    """True for an open binary file object over a regular, seekable file (not a pipe, socket, BytesIO or decompressor)."""
    import stat
    try:
        return stat.S_ISREG(os.fstat(stream.fileno()).st_mode) and stream.seekable()
    except (AttributeError, OSError, ValueError, io.UnsupportedOperation):
        return False


def index_mat_ark_file(stream, with_keys=True):
    """Record index of the rest of an ark FILE of binary float matrices, without reading the matrices (``xv_ark_index_fd``: one
    small pread per record): ``(offsets int64[n+1], rows int32[n], cols int32[n], keys or None)`` -- record i occupies bytes
    ``[offsets[i], offsets[i+1])`` -- or None when the host library is missing or the file holds anything else (another record
    type, a truncated tail): callers then read the stream the ordinary way."""
    lib = _host_lib()
    if lib is None or not hasattr(lib, "xv_ark_index_fd") or not is_regular_file(stream):
        return None
    fd = stream.fileno()
    pos, end = int(stream.tell()), int(os.fstat(fd).st_size)
    cap = 1 << 16
    offs, rows_all, cols_all, key_parts = [], [], [], []
    nxt, used, stop = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int(0)
    off = np.empty(cap, np.int64); rows = np.empty(cap, np.int32); cols = np.empty(cap, np.int32)
    kbuf = np.empty(cap * 48, np.uint8) if with_keys else None
    while pos < end:
        n = int(lib.xv_ark_index_fd(fd, pos, end, cap, off.ctypes.data, rows.ctypes.data, cols.ctypes.data,
                                    kbuf.ctypes.data if with_keys else None, len(kbuf) if with_keys else 0, ctypes.byref(used),
                                    ctypes.byref(nxt), ctypes.byref(stop)))
        if stop.value == 1:
            return None
        if n == 0:
            if with_keys and stop.value == 2:
                kbuf = np.empty(len(kbuf) * 4, np.uint8)            # one key longer than the whole buffer
                continue
            return None
        offs.append(off[:n].copy()); rows_all.append(rows[:n].copy()); cols_all.append(cols[:n].copy())
        if with_keys:
            key_parts.append(kbuf[:used.value].tobytes())
        pos = int(nxt.value)
    offsets = np.concatenate(offs + [np.array([pos], np.int64)]) if offs else np.array([pos], np.int64)
    keys = None
    if with_keys:
        # record keys are arbitrary bytes up to the first space: one that is not UTF-8, or that does not match the table grammar, or
        # a count that does not line up with the offsets sends the caller to the documented fallback (None: the generic reader,
        # which raises where the reference's would) -- byte-range sharding must never pair a vector with the wrong key
        try:
            keys = [k.strip() for k in b"".join(key_parts).decode().split("\n")[:-1]]
        except UnicodeDecodeError:
            return None
        if len(keys) != len(offsets) - 1 or any(_KEY_OK.match(k) is None for k in keys):
            return None
    return offsets, (np.concatenate(rows_all) if rows_all else np.zeros(0, np.int32)), \
        (np.concatenate(cols_all) if cols_all else np.zeros(0, np.int32)), keys


class FileRange(object):
    """Bytes ``[start, end)`` of an open regular file as a read-only binary stream (``readinto`` / ``read`` / ``tell``) with its
    own position (pread: the underlying file object's position is not touched, several ranges of one file can be read at
    once).  ``FileRange.bytes_read`` counts what this process pulled through such ranges (tests: a rank reads its share only)."""
    bytes_read = 0

    def __init__(self, stream, start, end):
        self._fd, self._pos, self._end = stream.fileno(), int(start), int(end)
        self._owner = stream

    def readinto(self, b):
        view = memoryview(b).cast("B")
        want = min(len(view), self._end - self._pos)
        if want <= 0:
            return 0
        got = os.preadv(self._fd, [view[:want]], self._pos)
        self._pos += got
        FileRange.bytes_read += got
        return got

    def read(self, n=-1):
        want = self._end - self._pos if n is None or n < 0 else min(int(n), self._end - self._pos)
        if want <= 0:
            return b""
        data = os.pread(self._fd, want, self._pos)
        self._pos += len(data)
        FileRange.bytes_read += len(data)
        return data

    def tell(self):
        return self._pos

    def seek(self, offset, whence=0):
        base = {0: 0, 1: self._pos, 2: self._end}[whence]
        self._pos = max(0, min(self._end, base + int(offset)))
        return self._pos

    def seekable(self):
        return True

    def readable(self):
        return True

    def close(self):
        pass


def read_mat_ark(file_or_fd):
    """Generator of (key, matrix) over an ark file / stream (block-buffered: see _BufferedStream; runs of binary
    float matrices are located by the native scanner when libxvector_host.so is present)."""
    raw = open_or_fd(file_or_fd)
    fd = raw if isinstance(raw, _BufferedStream) else _BufferedStream(raw)
    lib = _host_lib()
    try:
        while True:
            if lib is not None:
                for item in _scan_fm_records(fd, lib):
                    yield item
            key = read_key(fd)                                  # generic path: one record of any supported type
            if not key:
                break
            yield key, read_mat(fd)
    finally:
        if raw is not file_or_fd:
            raw.close()
        elif fd is not raw:
            fd.detach()


def read_mat_scp(file_or_fd):
    """Generator of (key, matrix) following a Kaldi scp (``key path[:offset]`` per line)."""
    fd = open_or_fd(file_or_fd)
    try:
        for line in fd:
            if isinstance(line, bytes):
                line = line.decode()
            key, rxfile = line.strip("\n").split(" ", 1)
            yield key, read_mat(rxfile.strip())
    finally:
        if fd is not file_or_fd:
            fd.close()


_RX_OFFSET = re.compile(r"^(.*):(\d+)$")


class _ScpTable(object):
    """An scp as a table: iterate it for ``(key, matrix)`` pairs (= ``read_mat_scp``) or call ``blocks()`` for the
    ``(keys, feats, offsets)`` blocks of ``read_mat_ark_blocks``.  An scp written next to its ark (copy-feats,
    compute-mfcc-feats, TableWriter) lists the records of each ark file in file order, back to back: ``blocks()`` then reads
    such a run as ONE stream -- seek to the first record, let the native scanner/gatherer take whole passes -- instead of one
    open + seek + read per utterance, and checks every key against the scp.  Where the table departs from the ark order (a
    subset, a shuffled list, a piped rxfile) it falls back to entry-by-entry reads, so any scp gives the same result."""

    def __init__(self, file_or_fd):
        if isinstance(file_or_fd, (list, tuple)):               # the table's lines, already read (a shard of an scp)
            lines = file_or_fd
        else:
            fd = open_or_fd(file_or_fd)
            try:
                text = fd.read()                                  # one read, one decode, one split: a 50 k-line table in 25 ms
            finally:
                if fd is not file_or_fd:
                    fd.close()
            lines = (text.decode() if isinstance(text, bytes) else text).split("\n")
        # parsed on first use: a job builds its tables before it starts its reader thread, and the 20-30 ms per 50 k lines then run
        # on that thread, beside the model load, instead of in front of it
        self._lines, self._entries = lines, None

    @property
    def entries(self):
        entries = self._entries
        if entries is None:
            # key = up to the first whitespace, rxfile = the rest without surrounding whitespace (a line without one raises, as a
            # malformed table should -- on FIRST USE, which may be a reader thread, not at construction).  Two threads touching
            # the table first at the same time both parse the same lines: the list is published before the lines are dropped.
            lines = self._lines
            if lines is None:                        # the other thread finished between the two reads
                return self._entries
            entries = [(k, r.rstrip()) for k, r in (ln.split(None, 1) for ln in lines if ln and not ln.isspace())]
            self._entries = entries
            self._lines = None
        return entries

    def __len__(self):
        return len(self.entries)

    # subclasses provide _one(rxfile) -> float32 array of one record and _ark_blocks = the ark block reader of the record type
    _ark_blocks = None

    def __iter__(self):
        for key, rx in self.entries:
            yield key, self._one(rx)

    def blocks(self, alloc=None):
        ents, n, i = self.entries, len(self.entries), 0
        misses = 0
        while i < n:
            key, rx = ents[i]
            m = _RX_OFFSET.match(rx)
            if m is None or rx.endswith("|") or misses >= 2 or _host_lib() is None:
                mat = self._one(rx)                                                    # entry-by-entry
                yield [key], mat, np.array([0, mat.shape[0]], np.int64)
                i += 1
                continue
            path, start = m.group(1), int(m.group(2)) - len(key) - 1                  # the record starts at its key
            run = i
            while run < n and ents[run][1].startswith(path + ":"):
                run += 1
            got = 0
            with open(path, "rb") as f:
                f.seek(start)
                for bkeys, feats, off in type(self)._ark_blocks(f, alloc):
                    # entries of the table against the records as they lie; records the table leaves out are skipped
                    take, nxt_i = [], i
                    for j, k in enumerate(bkeys):
                        if nxt_i < run and k == ents[nxt_i][0]:
                            take.append(j)
                            nxt_i += 1
                    if take and take[-1] == len(take) - 1:                             # a prefix of the block: views
                        same = len(take)
                        yield bkeys[:same], feats[:int(off[same])], off[:same + 1]
                    elif take:
                        lens = np.array([int(off[j + 1] - off[j]) for j in take], np.int64)
                        sel = np.concatenate([feats[int(off[j]):int(off[j + 1])] for j in take]) if int(lens.sum()) else feats[:0]
                        yield [bkeys[j] for j in take], sel, np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
                    i = nxt_i
                    got += len(take)
                    if len(take) * 4 < len(bkeys) or i >= run:
                        if len(bkeys) >= 16 and i < run:
                            misses = 2                                                 # a sparse table: entry by entry from here on
                        break
            # a run that ended after a handful of records means the table does not follow the ark: stop re-seeking for it
            misses = max(misses, 2) if misses >= 2 else (misses + 1 if got < 4 and i < run else 0)
            if got == 0:                                                               # not even the first key matched
                mat = self._one(rx)
                yield [key], mat, np.array([0, mat.shape[0]], np.int64)
                i += 1


class MatScp(_ScpTable):
    """Feature table: ``(key, float32 [T, F])`` / blocks ``(keys, feats[sum T, F], offsets)`` / in-place windows."""
    _ark_blocks = staticmethod(read_mat_ark_blocks)

    def windows(self, take_arena, first_fill=None, release=None):
        """``scan_mat_ark_windows`` over the table: runs of entries that follow their ark are read in place, every key is
        checked against the table, records of the ark that the table does not list are skipped (a subset table costs the bytes
        of the records it leaves out, not a seek per gap); entries that do not follow their ark (shuffled lists, pipes) are read
        one by one.  ``release(arena)``
        takes back an arena none of whose records matched the table (the consumer never sees it).  Once two runs in a row ended
        after a handful of records the table is taken not to follow its arks and is read entry by entry from there on -- a
        table that alternates between matching and renamed keys would otherwise read a full arena per miss."""
        ents, n, i = self.entries, len(self.entries), 0
        misses, tripped = 0, False
        while i < n:
            key, rx = ents[i]
            m = _RX_OFFSET.match(rx)
            if m is None or rx.endswith("|") or tripped or _host_lib() is None:
                mat = self._one(rx)
                yield [key], np.array([mat.__array_interface__["data"][0]], np.uint64), np.array([mat.shape[0]], np.int32), \
                    mat.shape[1], mat
                i += 1
                continue
            path, start = m.group(1), int(m.group(2)) - len(key) - 1                  # the record starts at its key
            run = i
            while run < n and ents[run][1].startswith(path + ":"):
                run += 1
            got = 0
            with open(path, "rb") as f:
                f.seek(start)
                source = scan_mat_ark_windows(f, take_arena, first_fill if i == 0 else None, release)
                while True:
                    try:
                        bkeys, addr, rows, cols, holder = next(source)
                    except StopIteration:
                        break
                    except (AssertionError, UnicodeDecodeError, UnknownMatrixHeader, BadInputFormat, BadSampleSize, ValueError,
                            struct.error):
                        # ``start`` is a guess (the entry's offset minus the length of the TABLE's key): with a renamed key it
                        # points into the previous record and the scanner reads garbage -- a miss, not an error
                        break
                    # the table's entries of this ark, in order, against the records as they lie: records the table does not
                    # list (a subset table: utterances removed by a filter) are skipped, not a reason to stop
                    take, nxt_i = [], i
                    for j, k in enumerate(bkeys):
                        if nxt_i < run and k == ents[nxt_i][0]:
                            take.append(j)
                            nxt_i += 1
                    if len(take) == len(bkeys):
                        yield bkeys, addr, rows, cols, holder
                    elif take:
                        idx = np.asarray(take)
                        yield [bkeys[j] for j in take], addr[idx], rows[idx], cols, \
                            ([holder[j] for j in take] if isinstance(holder, list) else holder)
                    elif release is not None and isinstance(holder, ArkArena):
                        release(holder)                                                # nothing of it goes downstream
                    i = nxt_i
                    got += len(take)
                    if len(take) * 4 < len(bkeys) or i >= run:
                        # mostly records nobody asked for: a sparse table is read entry by entry from here on
                        tripped = tripped or (len(bkeys) >= 16 and i < run)
                        break
                source.close()
            misses = misses + 1 if got < 4 and i < run else 0
            tripped = tripped or misses >= 2
            if got == 0:                                                               # not even the first key matched
                mat = self._one(rx)
                yield [key], np.array([mat.__array_interface__["data"][0]], np.uint64), np.array([mat.shape[0]], np.int32), \
                    mat.shape[1], mat
                i += 1

    @staticmethod
    def _one(rx):
        return np.ascontiguousarray(read_mat(rx), dtype=np.float32)


class VecScp(_ScpTable):
    """Vector table (e.g. vad.scp): ``(key, float32 [dim])`` / blocks ``(keys, values[sum dim], offsets)``."""
    _ark_blocks = staticmethod(read_vec_flt_ark_blocks)

    @staticmethod
    def _one(rx):
        return np.ascontiguousarray(read_vec_flt(rx), dtype=np.float32)


def write_mat(file_or_fd, m, key=""):
    """Write a binary Kaldi matrix ('FM '/'DM ', row-major)."""
    fd = open_or_fd(file_or_fd, mode="wb")
    try:
        m = np.asarray(m)
        if m.dtype == np.float32:
            tag = b"FM "
        elif m.dtype == np.float64:
            tag = b"DM "
        else:
            raise UnsupportedDataType("'%s', please use 'float32' or 'float64'" % m.dtype)
        _write_header(fd, key, tag)
        fd.write(b"\x04" + struct.pack("<I", m.shape[0]) + b"\x04" + struct.pack("<I", m.shape[1]))
        fd.write(np.ascontiguousarray(m).astype(m.dtype.newbyteorder("<"), copy=False).tobytes())
    finally:
        if fd is not file_or_fd:
            fd.close()
