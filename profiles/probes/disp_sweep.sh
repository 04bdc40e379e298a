set -e
cd $GRAFT_REPO_ROOT
SRC=$(ls pcgmix-*/csrc/pcgmix_saliency.hip)
for cfg in "128 8" "64 16" "64 32"; do
  set -- $cfg
  sed -i "s/^constexpr int kDispThreads = [0-9]*;/constexpr int kDispThreads = $1;/; s/^constexpr int kDispSplit = [0-9]*;/constexpr int kDispSplit = $2;/" $SRC
  make -C pcgmix-*/csrc -j8 > /dev/null 2>&1
  echo "== threads $1 split $2"
  python -m pytest tests/test_saliency_gpu.py -x -q -m gpu 2>&1 | tail -1
  python bench.py --kernels-only 2>&1 | grep salopt
done
