#!/bin/bash
# PMC passes over one conv shape (tools/bench_one.py KIND): each counter group in its own rocprofv3 run.
KIND=${1:-fwd}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_VALU_MFMA_F32" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_WAVES_EQ_64"; do
  i=$((i+1))
  (cd $R && rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_$KIND/p$i -- python tools/bench_one.py $KIND 5 > /tmp/pmc_$KIND.p$i.log 2>&1) || tail -3 /tmp/pmc_$KIND.p$i.log
done
cd $R && python tools/pmc_summary.py /tmp/pmc_$KIND conv_
