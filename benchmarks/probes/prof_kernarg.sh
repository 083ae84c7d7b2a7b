# A/B of HIP_FORCE_DEV_KERNARG (kernel arguments in device memory instead of host memory) on the attention backward mix.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1; do
  HIP_FORCE_DEV_KERNARG=$v timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kernarg$v -- python3 $R/benchmarks/attn_mix.py c2 both 5 > $R/gpurun_out/prof_kernarg$v.log 2>&1 || exit 1
  f=$(ls -t $R/gpurun_out/prof_kernarg$v/*/*kernel_stats.csv | head -n 1)
  echo "== HIP_FORCE_DEV_KERNARG=$v"; grep attn $f | cut -d, -f1-4 | cut -c1-120
done
