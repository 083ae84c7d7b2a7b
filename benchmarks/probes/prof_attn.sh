cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 1 0; do
  BRANCHGAN_ATTN_BWD_SPLIT=$v timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_attn_split$v -- python3 $R/benchmarks/attn_mix.py c2 bwd 5 > $R/gpurun_out/prof_attn_split$v.log 2>&1 || exit 1
done
