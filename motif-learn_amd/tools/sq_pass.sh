# SQ counters of some kernels of tools/run_dense.py, two counter sets in their own passes:
#   bash motif-learn_amd/tools/sq_pass.sh TAG WHICH       (e.g.  r3_strip strip8)   -> gpurun_out/TAG_sq{1,2}, summary on stdout
TAG=$1; WHICH=${2:-strip8}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
k=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_IFETCH SQ_WAIT_ANY"; do
  k=$((k+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/${TAG}_sq$k -o p -- python3 $R/motif-learn_amd/tools/run_dense.py --reps 2 --which $WHICH > $R/gpurun_out/${TAG}_sq$k.log 2>&1 || exit 1
done
python3 $R/motif-learn_amd/tools/pmc_summary.py $R/gpurun_out/${TAG}_sq1 $R/gpurun_out/${TAG}_sq2 --match zk_
