#!/usr/bin/env python
"""x-vector extraction worker -- MI355X twin of the reference's ``local/tf/extract_embedding.py``.

Same command line (extract_embedding.py:50-70 of the reference):

    extract_embedding.py --use-gpu {yes,no} --min-chunk-size N --chunk-size N \
        --feature-rspecifier RSPEC --vector-wspecifier WSPEC --model-dir DIR

and the same file protocol: ``ark,scp:A,S`` outputs are written to ``A.tmp.ark`` / ``S.tmp.scp`` and
renamed at the end with the scp text patched (reference lines 94-108, 134-148); the call returns early
when both outputs already exist (126-128); any exception prints a traceback and exits 1 (153-163).
A wspecifier that is a bare ``ark,scp:A,S`` (no ``| copy-vector`` pipe) is written natively by
``kaldi_io.TableWriter`` so no Kaldi binary is needed.  ``--use-gpu`` is accepted and ignored: the
forward pass always runs on the MI355X (see models.py).
"""
from __future__ import print_function

import argparse
import logging
import os
import re
import sys
import traceback

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

import kaldi_io  # noqa: E402
from models import Model, VectorCollector  # noqa: E402
from xvector_amd import jobclock  # noqa: E402

jobclock.mark("interpreter + imports")


def _import_torch_early():
    """``import torch`` (0.8 s, mostly dlopen) starts on a side thread the moment the worker is a main program, so that it runs
    next to argument parsing and the reading of the scp tables; ``Model.load_model``'s own import then finds it done."""
    import threading

    def run():
        try:
            import torch  # noqa: F401
        except Exception:          # the main thread's own import will report it
            pass
    threading.Thread(target=run, name="xv-import-torch", daemon=True).start()

logger = logging.getLogger('extract_embedding')
logger.setLevel(logging.INFO)
_handler = logging.StreamHandler()
_handler.setLevel(logging.INFO)
_handler.setFormatter(logging.Formatter("%(asctime)s [%(pathname)s:%(lineno)s - %(funcName)s - %(levelname)s ] %(message)s"))
logger.addHandler(_handler)


# (flag, dest, type, default, required, choices, help) -- the reference's command line, extract_embedding.py:50-70
_FLAGS = (
    ("--use-gpu", "use_gpu", str, "no", False, ("yes", "no"), "Accepted for compatibility; the extractor always runs on the GPU."),
    ("--min-chunk-size", "min_chunk_size", int, 100, False, None, "Minimum chunk-size allowed when extracting xvectors."),
    ("--chunk-size", "chunk_size", int, -1, False, None,
     "If set, extracts xvectors from specified chunk-size, and averages.  If not set, extracts an xvector from all "
     "available features."),
    ("--feature-rspecifier", "feature_rspecifier", str, None, True, None, "Kaldi rspecifier of the input features."),
    ("--vector-wspecifier", "vector_wspecifier", str, None, True, None,
     "Kaldi wspecifier for the x-vectors (ark, ark pipe, or ark,scp:A,S)."),
    ("--model-dir", "model_dir", str, None, True, None, "Model directory (model.meta + weights + done)."),
    # build-defined extension (defaults = reference behaviour): the Kaldi front-end of extract_xvectors.sh:68 on the GPU
    ("--cmn-window", "cmn_window", int, 0, False, None,
     "If > 0: sliding-window cepstral mean normalisation of this many frames (apply-cmvn-sliding --norm-vars=false) is "
     "applied on the GPU, so --feature-rspecifier can name raw features."),
    ("--cmn-center", "cmn_center", str, "yes", False, ("yes", "no"), "Centre the CMN window on the frame (--center=true)."),
    ("--vad-rspecifier", "vad_rspecifier", str, "", False, None,
     "If set: table of per-frame VAD decisions (same key order as the features); frames with decision 0 are dropped "
     "(select-voiced-frames) after the CMN."),
)


def get_args(argv=None):
    parser = argparse.ArgumentParser(description="Extract x-vectors from Kaldi features with the MI355X-native extractor.",
                                     formatter_class=argparse.ArgumentDefaultsHelpFormatter, conflict_handler='resolve')
    for flag, dest, typ, default, required, choices, text in _FLAGS:
        kw = dict(dest=dest, type=typ, help=text)
        if required:
            kw["required"] = True
        else:
            kw["default"] = default
        if choices:
            kw["choices"] = list(choices)
        parser.add_argument(flag, **kw)
    return process_args(parser.parse_args(argv))


def process_args(args):
    args.model_dir = args.model_dir.strip()
    if not args.model_dir or not os.path.exists(os.path.join(args.model_dir, 'model.meta')):
        raise Exception("This scripts expects the input model was exist in '{0}' directory.".format(args.model_dir))
    return args


_PAIRED = re.compile(r"^(?P<head>.*?)(?P<kind>ark,scp|scp,ark):(?P<first>[^,\s]+),(?P<second>[^,\s]+)$", re.S)


def process_wspecifier(wspecifier):
    """-> (temporary wspecifier, final ark, final scp); (wspecifier, None, None) when there is nothing to rename.
    Only the LAST whitespace-separated token may be the paired table (extract_embedding.py:94-108)."""
    tokens = wspecifier.split()
    m = _PAIRED.match(tokens[-1]) if tokens else None
    if m is None or m.group("head"):
        return wspecifier, None, None
    first, second = m.group("first"), m.group("second")
    ark, scp = (first, second) if m.group("kind") == "ark,scp" else (second, first)
    tmp = {"ark": ark + ".tmp.ark", "scp": scp + ".tmp.scp"}
    a, b = m.group("kind").split(",")
    prefix = "".join(t + " " for t in tokens[:-1])
    return "%s%s:%s,%s" % (prefix, m.group("kind"), tmp[a], tmp[b]), ark, scp


def _open_output(wspecifier, ark, scp):
    if ark is not None and not wspecifier.lstrip().startswith('|'):
        # the scp lines name the FINAL ark from the start: nothing to patch after the rename (the reference has to rewrite the
        # text because copy-vector only knows the temporary name, extract_embedding.py:139-148)
        return kaldi_io.TableWriter(ark + '.tmp.ark', scp + '.tmp.scp', scp_ark_name=ark)
    return kaldi_io.open_or_fd(wspecifier, 'wb')


def _native_table(wspecifier, ark):
    return ark is not None and not wspecifier.lstrip().startswith('|')


class _Discard(object):
    """Output sink of the non-root ranks (they write nothing: make_embedding gathers on rank 0)."""
    mode = "wb"

    def write(self, data):
        return len(data)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _Null(object):
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def _open_table(rspecifier, scp_reader, ark_reader):
    """(key, value) iterator for 'scp:FILE' (random-access table read sequentially) or, when ``ark_reader`` is given, for
    any ark rspecifier open_or_fd understands; None when the caller should open the rspecifier as an ark stream itself."""
    spec = rspecifier.strip()
    head = spec.split(':', 1)[0].replace(' ', '')
    if head.split(',')[0] == 'scp' and not spec.endswith('|'):
        return scp_reader(spec.split(':', 1)[1])
    if ark_reader is None:
        return None
    return ark_reader(kaldi_io.open_or_fd(spec))


def _scp_shard(spec, rank, world, vad_spec=None):
    """Lines [n*rank/world, n*(rank+1)/world) of the scp behind 'scp:FILE' as the table's lines: contiguous, so concatenating the
    ranks' outputs in rank order restores the input order (the role of utils/split_data.sh + split_scp.pl in
    extract_xvectors.sh:63-65).  With ``vad_spec`` (also an scp table) the second value is the VAD scp restricted to the same
    keys, in the same order.  Third value: the keys of ALL shards (every rank can tell which utterances its peers hold, so the
    final exchange carries no keys) -- as ``kaldi_io.KeyRange``s over the table's text: a rank decodes and splits the lines of
    its OWN range only (the line index of a 1 M-line table is one native pass, 20 ms; splitting every line on every rank was
    0.5 s of each rank's job), and the rank that writes hands the other ranges' keys to the native writer as bytes."""
    table = kaldi_io.ScpText(spec.split(':', 1)[1].strip())
    cuts = [len(table) * r // world for r in range(world + 1)]
    shard_keys = [table.keys(cuts[r], cuts[r + 1]) for r in range(world)]
    mine = table.lines(cuts[rank], cuts[rank + 1])
    if vad_spec is None:
        return mine, None, shard_keys
    vad = kaldi_io.ScpText(vad_spec.split(':', 1)[1].strip())
    lo, hi = cuts[rank], cuts[rank + 1]
    if len(vad) == len(table) and vad.keys(lo, hi) == shard_keys[rank]:
        return mine, vad.lines(lo, hi), shard_keys          # the usual case: vad.scp lists the same utterances in the same order
    lines, where = vad.lines(0, len(vad)), {}
    for k, ln in zip(vad.keys(), lines):
        where[k] = ln
    return mine, [where[k] for k in shard_keys[rank] if k in where], shard_keys


def _embedding_dim(model_dir):
    """Width of the x-vector this model directory produces (embed_layer-0: models.py:82-86), from its meta alone."""
    try:
        import json
        with open(os.path.join(model_dir, 'model.meta'), 'rb') as fid:
            return int(json.loads(fid.read().decode('utf-8'))["topology"]["embedding_sizes"][0])
    except Exception:          # a TensorFlow-written directory: every class of the reference has 512 (models.py:29)
        return 512


def _is_scp_table(rspecifier):
    spec = rspecifier.strip()
    return spec.split(':', 1)[0].replace(' ', '').split(',')[0] == 'scp' and not spec.endswith('|')


def eval_dnn(args):
    use_gpu = args.use_gpu == 'yes'
    wspecifier, ark, scp = process_wspecifier(args.vector_wspecifier)
    if ark is not None and scp is not None and os.path.exists(scp) and (os.path.exists(ark) or os.path.exists(ark + '.0')):
        logger.info('Both output ark and scp files exist. Return from this call.')
        return
    # under torchrun (one process per GPU) only rank 0 owns the output table.  An scp feature table is sharded by line
    # range -- every rank reads ONLY its utterances, extracts them as a single process would, and one RCCL gather at the
    # end brings the x-vectors to rank 0; any other rspecifier (ark stream, pipe) is read by every rank and sharded per window
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    root = rank == 0
    grouped = world > 1 or (os.environ.get("XV_FORCE_DIST") == "1" and "RANK" in os.environ)     # the latter: 1-rank group (tests)
    if grouped and _is_scp_table(args.feature_rspecifier) and (not args.vad_rspecifier or _is_scp_table(args.vad_rspecifier)) and \
            _native_table(wspecifier, ark) and os.environ.get("XVECTOR_SHARD_OUTPUT", "gather") == "files":
        return _extract_into_shard_files(args, use_gpu, ark, scp, rank, world)
    if grouped:
        # the process group is only needed for the ONE gather at the end: RCCL comes up on a side thread while this one loads
        # the model and extracts (nothing is written to the output before that gather, so the stdout redirection of
        # dist.init_process_group cannot hit an 'ark:-' stream)
        from xvector_amd import dist as xdist
        # (XVECTOR_GROUP_START=early: from the start of the job, next to the model load -- the round-3 behaviour, for A/B)
        xdist.init_process_group_async(after_mark=None if os.environ.get("XVECTOR_GROUP_START") == "early" else "first window launched")
    presharded = grouped and _is_scp_table(args.feature_rspecifier) and \
        (not args.vad_rspecifier or _is_scp_table(args.vad_rspecifier))
    model = Model()
    model.prepin_staging = True            # a worker job: the staging sets are pinned while the weights load (models.load_model)
    if presharded:
        feat_scp, vad_scp, shard_keys = _scp_shard(args.feature_rspecifier, rank, world, args.vad_rspecifier or None)
        # the job's one exchange is known now: [emitted? | x-vector] rows of every utterance, from host memory to rank 0's host memory
        xdist.set_gather_payload(sum(len(k) for k in shard_keys) * (1 + _embedding_dim(args.model_dir)) * 4)
        feats = kaldi_io.MatScp(feat_scp)
        vad = kaldi_io.VecScp(vad_scp) if vad_scp is not None else None
    else:
        vad = _open_table(args.vad_rspecifier, kaldi_io.VecScp, lambda stream: stream) if args.vad_rspecifier else None
        feats = _open_table(args.feature_rspecifier, kaldi_io.MatScp, None)
    collector = VectorCollector() if presharded else None
    jobclock.mark("tables opened")
    with (kaldi_io.open_or_fd(args.feature_rspecifier) if feats is None else _Null()) as input_fid:
        with (collector if presharded else _open_output(wspecifier, ark, scp) if root else _Discard()) as output_fid:
            model.make_embedding(input_fid if feats is None else feats, output_fid, args.model_dir, args.min_chunk_size,
                                 args.chunk_size, use_gpu, logger, vad_stream=vad, cmn_window=args.cmn_window,
                                 cmn_center=args.cmn_center == 'yes', distributed=not presharded)
    jobclock.mark("extraction (reader -> kernels -> last vector down)")
    if presharded:
        _gather_and_write(model, collector, shard_keys, wspecifier, ark, scp, rank, world)
    if grouped:
        # every rank stays until rank 0 has the vectors (a peer that tears the group down early would take the gather with it)
        from xvector_amd import dist as xdist
        xdist.finish_process_group()
    if not root:
        return
    if ark is not None:
        os.rename(ark + '.tmp.ark', ark)
    if scp is not None and _native_table(wspecifier, ark):
        os.rename(scp + '.tmp.scp', scp)                   # written by TableWriter with the final ark name in it
    elif scp is not None:
        with open(scp + '.tmp.scp', 'rt') as fid_in:
            text = fid_in.read().replace(ark + '.tmp.ark', ark)
        if text and text[-1] != '\n':
            text += '\n'
        with open(scp + '.tmp', 'wt') as fid_out:
            fid_out.write(text)
        os.rename(scp + '.tmp', scp)
        os.remove(scp + '.tmp.scp')
    jobclock.mark("rename")
    logger.info(jobclock.line())


def _extract_into_shard_files(args, use_gpu, ark, scp, rank, world):
    """XVECTOR_SHARD_OUTPUT=files -- the reference's own output protocol instead of the gather: there every job writes
    ``xvector.JOB.ark`` / ``.scp`` and the script concatenates the scp files (extract_xvectors.sh:83-95).  Here rank r extracts its
    line range of the scp into ``<ark>.r`` while it runs (no collector, no final burst of writes through one rank), and rank 0
    concatenates the ranks' scp parts -- whose lines name ``<ark>.r`` -- into ``<scp>`` in rank order = input order.  No process
    group exists in this mode: nothing waits for RCCL's bring-up (about a third of a short job's wall clock), N ranks write N
    files at once, and what is left of a job that dies late are the complete shards of the ranks that finished.  The ranks meet
    through the file system: a part appears under its final name (``<scp>.r.<job>.part``) only when it is complete.  ``<job>`` tells
    the parts of THIS job from what a job that died before its concatenation left behind (a rank 0 that reaches the wait loop first
    would otherwise take a stale part for rank r's result): XVECTOR_JOB_TOKEN (the package's launcher sets it), else torchrun's
    TORCHELASTIC_RUN_ID; ranks with neither write untokenised parts (``<scp>.r.part``), and rank 0 then accepts a part only if it was
    written after rank 0's own process was born -- a part an earlier, dead job left behind is older than that and is waited out (the
    job fails loudly on XVECTOR_SHARD_TIMEOUT if rank r never delivers) instead of being concatenated."""
    import glob
    import time
    # a token every rank PROVABLY shares: the launcher's (xvector_amd.launch exports XVECTOR_JOB_TOKEN), torchrun's run id; ranks
    # started any other way (per-rank shell wrappers, several nodes on one file system) share no parent pid, so without a token
    # the parts carry none -- and then nothing is swept: a part without a token cannot be told from a sibling's finished one.
    token = os.environ.get("XVECTOR_JOB_TOKEN") or os.environ.get("TORCHELASTIC_RUN_ID") or ""
    token = "".join(ch if ch.isalnum() or ch in "-_" else "_" for ch in token)
    part = ('.%s.part' % token) if token else '.part'
    job_started = time.time()
    if rank == 0 and token:
        started = time.time()
        for stale in glob.glob(glob.escape(scp) + '.*.part'):        # other jobs' leftovers (also ranks >= world of a larger run)
            try:
                if not stale.endswith(part) and os.path.getmtime(stale) < started - 1.0:
                    os.remove(stale)
            except OSError:
                pass
    feat_scp, vad_scp, _ = _scp_shard(args.feature_rspecifier, rank, world, args.vad_rspecifier or None)
    feats = kaldi_io.MatScp(feat_scp)
    vad = kaldi_io.VecScp(vad_scp) if vad_scp is not None else None
    jobclock.mark("tables opened")
    my_ark, my_scp = '%s.%d' % (ark, rank), '%s.%d%s' % (scp, rank, part)
    for stale in (my_ark, my_scp):
        if os.path.exists(stale):
            os.remove(stale)
    with kaldi_io.TableWriter(my_ark + '.tmp.ark', my_scp + '.tmp', scp_ark_name=my_ark) as out:
        worker = Model()
        worker.prepin_staging = True
        worker.make_embedding(feats, out, args.model_dir, args.min_chunk_size, args.chunk_size, use_gpu, logger, vad_stream=vad,
                               cmn_window=args.cmn_window, cmn_center=args.cmn_center == 'yes', distributed=False)
    os.rename(my_ark + '.tmp.ark', my_ark)
    os.rename(my_scp + '.tmp', my_scp)
    jobclock.mark("extraction + write of this rank's shard")
    if rank != 0:
        return
    deadline = time.time() + float(os.environ.get("XVECTOR_SHARD_TIMEOUT", "3600"))
    parts = ['%s.%d%s' % (scp, r, part) for r in range(world)]
    born = None if token else (jobclock.process_birth() or job_started)

    def delivered(p):
        try:
            return os.path.getmtime(p) >= born if born is not None else os.path.exists(p)
        except OSError:
            return False
    while not all(delivered(p) for p in parts):
        if time.time() > deadline:
            raise RuntimeError("sharded extraction: still waiting for %s" % ", ".join(p for p in parts if not delivered(p)))
        time.sleep(0.02)
    jobclock.mark("wait for the other ranks' shards")
    with open(scp + '.tmp', 'wb') as fid_out:
        for p in parts:
            with open(p, 'rb') as fid_in:
                fid_out.write(fid_in.read())
    os.rename(scp + '.tmp', scp)
    for p in parts:
        os.remove(p)
    jobclock.mark("scp concatenated")
    logger.info(jobclock.line())


def _gather_and_write(model, collector, shard_keys, wspecifier, ark, scp, rank, world):
    """The single exchange of the scp-sharded mode (models.gather_shard_vectors: ONE gather of ``[emitted? | x-vector]`` rows,
    no keys travel), then rank 0 writes the shards in rank order = input order."""
    from models import gather_shard_vectors
    shards = gather_shard_vectors(model.device_model, collector, shard_keys, rank, world)
    if rank == 0:
        with _open_output(wspecifier, ark, scp) as output_fid:
            for keys, vecs, emitted in shards:
                kaldi_io.write_vec_flt_batch(output_fid, keys, vecs, emitted)
        jobclock.mark("write")


def main(argv=None):
    if argv is None:
        _import_torch_early()
    args = get_args(argv)          # outside the try, as in the reference: --help / usage errors exit through argparse
    try:
        eval_dnn(args)
    except BaseException as e:
        if not isinstance(e, KeyboardInterrupt):
            traceback.print_exc()
        sys.exit(1)


if __name__ == "__main__":
    # RCCL's device code starts loading NOW, in front of `import torch` (xvector_amd/rccl_prewarm.py: only here, in the worker's own
    # process, which leaves through os._exit on every path below -- a process that opened librccl before torch must not run its exit
    # handlers).  The job's one exchange is then "a single RCCL gather" again without paying a second of bring-up for it.
    from xvector_amd import rccl_prewarm
    rccl_prewarm.start()
    code = 0
    try:
        main()
    except SystemExit as e:            # argparse's usage errors (2), main()'s sys.exit(1) after a traceback
        code = e.code if isinstance(e.code, int) else (0 if e.code is None else 1)
        if e.code is not None and not isinstance(e.code, int):
            print(e.code, file=sys.stderr)
    except BaseException:              # what the reference lets escape main() (process_args' exception): traceback, exit status 1
        traceback.print_exc()
        code = 1
    # a finished worker has nothing left to tear down in order: the outputs are closed and renamed.  Leaving through the
    # interpreter's shutdown (torch, the HIP runtime, RCCL's proxy threads) costs ~0.4 s of every job
    logging.shutdown()
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(code)
