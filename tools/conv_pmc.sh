#!/bin/bash
# counters of the forward 3x3 convolution (conv3x3_limb_kernel) on the EDSR trunk shape: tools/pmc.sh passes over tools/conv_time.py
export PMC_SCRIPT=conv_time.py PMC_ARGS=""
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
bash $R/tools/pmc.sh c1 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA && \
bash $R/tools/pmc.sh c2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS && \
bash $R/tools/pmc.sh c3 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS && \
bash $R/tools/pmc.sh c4 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY && \
python3 $R/tools/pmc_read.py --kernel conv3x3_limb c1 c2 c3 c4
