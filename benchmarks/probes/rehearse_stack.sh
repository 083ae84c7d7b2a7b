R=$GRAFT_REPO_ROOT
export BRANCHGAN_REHEARSE_RCCL=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 BRANCHGAN_DP_SHARDED=1
for i in 1 2 3 4 5; do
  MASTER_PORT=$((29700 + i)) timeout -k 10 200 python -X faulthandler $R/bench.py --workload c1 --steps 4 --warmup 2 --no-cpu-baseline --no-fp32 > $R/gpurun_out/rs_$i.json 2> $R/gpurun_out/rs_$i.err
  rc=$?
  echo "run $i rc=$rc"
  if [ $rc -ne 0 ]; then grep -n "File \"\|Thread\|thread" $R/gpurun_out/rs_$i.err | grep -v "dist-packages/torch/_\|threading.py" | head -40 | cut -c1-180; break; fi
done
