#!/bin/bash
# counters of the training kernels (decode_rays_limb_kernel, render_pass_backward_gates_limb_kernel): tools/pmc.sh passes over the train bench
export PMC_SCRIPT=../bench.py PMC_ARGS="--workload train --steps 3 --warmup 1 --no-cpu-baseline"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
bash $R/tools/pmc.sh t1 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA && \
bash $R/tools/pmc.sh t2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS && \
bash $R/tools/pmc.sh t3 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS && \
bash $R/tools/pmc.sh t4 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR && \
for k in decode_rays_limb render_pass_backward_gates_limb; do python3 $R/tools/pmc_read.py --kernel $k t1 t2 t3 t4; done
