# rocprofv3 kernel stats of the attention backward launches of one train step, per launch form:
#   one = one launch (BRANCHGAN_ATTN_BWD_ONE=1), split = dq / dkv / slots kernels (ONE=0), old = attn_bwd.hip's kernels (SPLIT=0)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=${1:-c2}
for v in one split old; do
  case $v in one) E="BRANCHGAN_ATTN_BWD_ONE=1";; split) E="BRANCHGAN_ATTN_BWD_ONE=0";; old) E="BRANCHGAN_ATTN_BWD_SPLIT=0";; esac
  env $E true
  export BRANCHGAN_ATTN_BWD_ONE=1 BRANCHGAN_ATTN_BWD_SPLIT=1
  [ $v = split ] && export BRANCHGAN_ATTN_BWD_ONE=0
  [ $v = old ] && export BRANCHGAN_ATTN_BWD_SPLIT=0
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_attn_$v -- python3 $R/benchmarks/attn_mix.py $WL bwd 5 > $R/gpurun_out/prof_attn_$v.log 2>&1 || exit 1
  f=$(ls -t $R/gpurun_out/prof_attn_$v/*/*kernel_stats.csv | head -n 1)
  echo "== $v"; grep attn $f | cut -d, -f1-4 | cut -c1-120
done
