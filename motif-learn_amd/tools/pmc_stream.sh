# PMC passes over the stream batch kernel (run on the GPU box: bash motif-learn_amd/tools/pmc_stream.sh); results land in gpurun_out/pmc_*
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INST_LEVEL_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_IFETCH"; do
  n=$(echo $set | md5sum | cut -c1-6)
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc_$n -o p -- python3 $R/motif-learn_amd/tools/sweep_batch.py --cases 48:8:f32 --paths 4 --rounds 3 --gb 4 > $R/gpurun_out/pmc_$n.log 2>&1 || exit 1
done
ls -R $R/gpurun_out/pmc_* | head -30
