#!/usr/bin/env bash
# x-vector extraction of one Kaldi data directory on the MI355X(s) of one node -- what local/tf/extract_xvectors.sh of the
# reference does with `nj` CPU jobs + Kaldi binaries, as ONE launch:
#   * raw features + VAD are read directly (sliding-window CMN and voiced-frame selection run on the GPU),
#   * with --ngpu N > 1 every rank reads and extracts its own line range of feats.scp / vad.scp and ONE RCCL gather brings
#     the vectors to rank 0, which writes xvector.ark/.scp in input order (no split_data.sh, no per-job arks to concatenate),
#   * speaker-level means as in stage 2 of the reference script.
# Usage: extract_xvectors_mi355x.sh [--ngpu N] [--chunk-size 10000] [--min-chunk-size 25] [--cmn-window 300] <nnet-dir> <data> <xvector-dir>
set -euo pipefail
ngpu=1; chunk_size=-1; min_chunk_size=25; cmn_window=300
while [[ $# -gt 3 ]]; do
  case "$1" in
    --ngpu) ngpu=$2; shift 2;;
    --chunk-size) chunk_size=$2; shift 2;;
    --min-chunk-size) min_chunk_size=$2; shift 2;;
    --cmn-window) cmn_window=$2; shift 2;;
    *) echo "$0: unknown option $1" >&2; exit 1;;
  esac
done
[[ $# -eq 3 ]] || { sed -n 2,10p "$0"; exit 1; }
srcdir=$1; data=$2; dir=$3
here=$(cd "$(dirname "$0")" && pwd)
for f in "$data/feats.scp" "$data/vad.scp" "$data/spk2utt"; do [[ -f $f ]] || { echo "$0: no such file $f" >&2; exit 1; }; done
# the nnet dir carries the chunk sizes the model was trained for (run_xvector.sh:77-79 of the reference)
[[ -f $srcdir/min_chunk_size ]] && min_chunk_size=$(cat "$srcdir/min_chunk_size")
[[ $chunk_size -le 0 && -f $srcdir/max_chunk_size ]] && chunk_size=$(cat "$srcdir/max_chunk_size")
model_dir=$srcdir; [[ -e $srcdir/model_final ]] && model_dir=$srcdir/model_final
mkdir -p "$dir/log"
export HSA_ENABLE_IPC_MODE_LEGACY=0
launcher=(python)
# one process per GPU through the package's own launcher (xvector_amd/launch.py: torch.distributed.run's environment contract
# without its elastic agent -- seconds of start-up per job); `python -m torch.distributed.run --nproc-per-node N ...` works as well
[[ $ngpu -gt 1 ]] && launcher=(env PYTHONPATH="$here/../..${PYTHONPATH:+:$PYTHONPATH}" python -m xvector_amd.launch --nproc "$ngpu")
"${launcher[@]}" "$here/extract_embedding.py" --use-gpu=yes --min-chunk-size="$min_chunk_size" --chunk-size="$chunk_size" \
    --feature-rspecifier="scp:$data/feats.scp" --vad-rspecifier="scp:$data/vad.scp" --cmn-window="$cmn_window" --cmn-center=yes \
    --vector-wspecifier="ark,scp:$dir/xvector.ark,$dir/xvector.scp" --model-dir="$model_dir" 2>&1 | tee "$dir/log/extract.log"
python "$here/speaker_mean.py" "$data/spk2utt" "$dir/xvector.scp" "$dir/spk_xvector.ark" "$dir/spk_xvector.scp" "$dir/num_utts.ark" \
    2>&1 | tee "$dir/log/speaker_mean.log"
