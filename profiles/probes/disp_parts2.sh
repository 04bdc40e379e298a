# Timing-only experiment: software-prefetched leaf loop, whole kernel / middle sum only / head+tail only.
set -e
cd $GRAFT_REPO_ROOT
SRC=$(ls pcgmix-*/csrc/pcgmix_saliency.hip)
cp $SRC /tmp/sal_orig.hip
python3 - "$SRC" <<'PY'
import sys
p=sys.argv[1]; s=open(p).read()
old='''  float r[8];
  elem.get8(start, r);
  int i = 8;
  for (; i < n - (n % 8); i += 8) {
    float v[8];
    elem.get8(start + i, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], v[j]);
  }'''
new='''  float r[8];
  elem.get8(start, r);
  int i = 8;
  const int nfull = n - (n % 8);
  if (i < nfull) {
    float cur[8];
    elem.get8(start + i, cur);
    for (i += 8; i < nfull; i += 8) {
      float nxt[8];
      elem.get8(start + i, nxt);
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], cur[j]);
#pragma unroll
      for (int j = 0; j < 8; ++j) cur[j] = nxt[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], cur[j]);
  }'''
assert old in s
open(p,'w').write(s.replace(old,new))
PY
cp $SRC /tmp/sal_pf.hip
make -C pcgmix-*/csrc -j8 > /dev/null 2>&1
echo "== prefetch, full"; python bench.py --kernels-only 2>&1 | grep salopt
sed -i 's|    if (own_longer) {  // np.sum(s1\[:d\])|    if (own_longer \&\& T < 0) {  // np.sum(s1[:d])|' $SRC
make -C pcgmix-*/csrc -j8 > /dev/null 2>&1
echo "== prefetch, mid only"; python bench.py --kernels-only 2>&1 | grep salopt
cp /tmp/sal_pf.hip $SRC
sed -i 's|    float cur = pw_sum(mid, nS);|    float cur = mid.get(0);|' $SRC
make -C pcgmix-*/csrc -j8 > /dev/null 2>&1
echo "== prefetch, head/tail only"; python bench.py --kernels-only 2>&1 | grep salopt
cp /tmp/sal_orig.hip $SRC
make -C pcgmix-*/csrc -j8 > /dev/null 2>&1
