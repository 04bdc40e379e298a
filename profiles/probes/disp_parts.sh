# Timing-only experiment (results are wrong on purpose): the displacement kernel without its
# head/tail sums, and without its middle sum, to see which part the launch time belongs to.
set -e
cd $GRAFT_REPO_ROOT
SRC=$(ls pcgmix-*/csrc/pcgmix_saliency.hip)
cp $SRC /tmp/sal_orig.hip
echo "== full"; python bench.py --kernels-only 2>&1 | grep salopt
sed -i 's|    if (own_longer) {  // np.sum(s1\[:d\])|    if (own_longer \&\& T < 0) {  // np.sum(s1[:d])|' $SRC
grep -n "own_longer && T < 0" $SRC | head -2
make -C pcgmix-*/csrc -j8 > /dev/null 2>&1
echo "== mid only (no head/tail)"; python bench.py --kernels-only 2>&1 | grep salopt
cp /tmp/sal_orig.hip $SRC
sed -i 's|    float cur = pw_sum(mid, nS);|    float cur = mid.get(0);|' $SRC
make -C pcgmix-*/csrc -j8 > /dev/null 2>&1
echo "== head/tail only (no mid)"; python bench.py --kernels-only 2>&1 | grep salopt
cp /tmp/sal_orig.hip $SRC
make -C pcgmix-*/csrc -j8 > /dev/null 2>&1
