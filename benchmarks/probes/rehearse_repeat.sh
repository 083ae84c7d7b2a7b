# the one-rank RCCL rehearsal of the multi-rank schedule, several times per form: does the capture race with the collectives' threads?
R=$GRAFT_REPO_ROOT
export BRANCHGAN_REHEARSE_RCCL=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1
for sh in 1; do
  for i in 1 2 3 4 5 6 7 8; do
    BRANCHGAN_DP_SHARDED=$sh MASTER_PORT=$((29600 + sh * 10 + i)) timeout -k 10 200 python $R/bench.py --workload c1 --steps 4 --warmup 2 --no-cpu-baseline --no-fp32 > $R/gpurun_out/rehearse_$sh$i.json 2> $R/gpurun_out/rehearse_$sh$i.err
    echo "sharded=$sh run $i rc=$? legacy-stream messages: $(grep -c 'legacy stream' $R/gpurun_out/rehearse_$sh$i.err)"
  done
done
