#!/bin/bash
# More than one RCCL rank on a box with ONE GPU: every rank claims to be on a host of its own (NCCL_HOSTID), so the
# library's "duplicate GPU" check does not apply and the ranks talk through its socket transport over the loopback
# interface.  Rates mean nothing (host-staged, one GPU time-shared); what this exercises is the code the shm test
# communicator cannot: ncclCommInitRank with n > 1, ncclCommSplit for the halo stream, the start-up self-test, grouped
# send / receive of halo rows, all-reduces and all-gathers of the real library.
#   tools/rccl_ranks_one_gpu.sh <nranks> <bench.py args...>       (logs: gpurun_out/rccl<n>_r<rank>.log)
n=${1:-2}; shift
mkdir -p gpurun_out
port=$((29600 + RANDOM % 300))
pids=()
for ((r = 0; r < n; r++)); do
  RANK=$r LOCAL_RANK=0 WORLD_SIZE=$n LOCAL_WORLD_SIZE=$n MASTER_ADDR=127.0.0.1 MASTER_PORT=$port \
  NCCL_HOSTID=cna_one_gpu_host_$r NCCL_SOCKET_IFNAME=lo NCCL_IB_DISABLE=1 \
  HSA_ENABLE_IPC_MODE_LEGACY=0 CNA_COMM_TIMEOUT=${CNA_COMM_TIMEOUT:-60} \
  timeout ${RANK_TIMEOUT:-300} python bench.py --gpus $n "$@" > gpurun_out/rccl${n}_r$r.log 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=$?; done
for ((r = 0; r < n; r++)); do echo "== rank $r"; tail -c 3000 gpurun_out/rccl${n}_r$r.log; done
exit $rc
