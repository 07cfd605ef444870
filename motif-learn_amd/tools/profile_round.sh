# Profiling passes of one round, run on the GPU box:  bash motif-learn_amd/tools/profile_round.sh r02
# (1) rocprofv3 --kernel-trace --stats of the default bench command (minus the side runs that launch the timed kernel again)
# (2) HBM traffic of the timed batch kernel: separate FETCH_SIZE / WRITE_SIZE passes (MI355X_MICROARCH.md "HBM")
# (3) SQ counters of the dense / maps / (64, 12) batch kernels, two counter sets in their own passes
# (4) kernel trace of the clustering consumers
# Everything lands under gpurun_out/<tag>_*; summaries are made afterwards with tools/pmc_summary.py and copied to profiles/.
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
set -e
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -o b -- python3 $R/bench.py --no-4096 --no-cpu-baseline --no-host-api > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/${TAG}_stats.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/${TAG}_pmc_$c -o p -- python3 $R/bench.py --only-timed-loop --steps 3 --warmup 1 > $R/gpurun_out/${TAG}_pmc_$c.json 2> $R/gpurun_out/${TAG}_pmc_$c.err
done
k=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_IFETCH SQ_WAIT_ANY"; do
  k=$((k+1))
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/${TAG}_sq$k -o p -- python3 $R/motif-learn_amd/tools/run_dense.py --reps 2 > $R/gpurun_out/${TAG}_sq$k.log 2>&1
done
# (4) the clustering consumers (zk_cluster.hip): kernel trace of tools/time_clustering.py
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_cluster -o c -- python3 $R/motif-learn_amd/tools/time_clustering.py > $R/gpurun_out/${TAG}_cluster.log 2>&1
find $R/gpurun_out/${TAG}_* -name "*.csv" | head -40
