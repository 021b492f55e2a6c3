#!/bin/bash
# Where the issue slots of the fine render pass go (render_pass3_kernel<3>, 640 000 rays x 192 samples): rocprofv3 --pmc passes over
# tools/kbench_fine_pass.py (one counter group per pass, <= 4 counters), read out for the longest render_pass dispatch.
# SQ_* "cycle" counters are in quad-cycles per SIMD summed over the chip; SQ_VALU_MFMA_BUSY_CYCLES in cycles (32 per 32x32x16 bf16 MFMA).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
bash $R/tools/pmc.sh i1 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CYCLES && \
bash $R/tools/pmc.sh i2 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY && \
bash $R/tools/pmc.sh i3 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD && \
bash $R/tools/pmc.sh i4 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA && \
bash $R/tools/pmc.sh i5 SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16 && \
bash $R/tools/pmc.sh i6 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL && \
bash $R/tools/pmc.sh i7 TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum
python3 $R/tools/pmc_read.py --kernel render_pass3 i1 i2 i3 i4 i5 i6 i7
