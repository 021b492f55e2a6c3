#!/bin/bash
export PMC_SCRIPT=conv_wgrad_time.py PMC_ARGS=""
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
bash $R/tools/pmc.sh w1 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA && \
bash $R/tools/pmc.sh w2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS && \
bash $R/tools/pmc.sh w3 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS && \
bash $R/tools/pmc.sh w4 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY && \
bash $R/tools/pmc.sh w5 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_WAVES && \
python3 $R/tools/pmc_read.py --kernel wgrad_limb w1 w2 w3 w4 w5
