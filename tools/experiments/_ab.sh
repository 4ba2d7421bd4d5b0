set -e
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in flat mubuf; do
  cp tools/experiments/_pair8_$v.hip.txt x-vector-kaldi-tf_amd/csrc/xv_pair8.hip
  make -C x-vector-kaldi-tf_amd/csrc -j6 > /dev/null 2>&1
  echo "== $v"
  python tools/pair8_bench.py 2>&1 | grep "f16bf8, wave\|bf16x3,"
done
done
