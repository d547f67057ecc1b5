#!/bin/bash
# tools/corr_evidence.sh OUTDIR -- the measurements behind DESIGN.md section 4.2 (correlation kernel variants)
O=$1; mkdir -p $O
F='amdgpu.ids\|MIOpen'
./tools/micro/mfma4x4_rate > $O/corr_mfma_4x4x1_issue_rate.txt 2>&1
{ for v in "0 0" "30 0" "30 2"; do set -- $v; echo "== RFN_CORR_VARIANT=$1 RFN_CORR_MFMA_CFG=$2"; for t in corr_l1_fused corr_l1 corr_l2_fused; do RFN_CORR_VARIANT=$1 RFN_CORR_MFMA_CFG=$2 bash tools/corr_kt.sh $t; done; done; } > $O/corr_mfma_kernel_trace_ab.txt 2>&1
{ for c in 0 2; do for e in 1 32; do echo "== RFN_CORR_MFMA_CFG=$c, traced launch preceded by $((e-1)) back-to-back launches"; RFN_CORR_MFMA_CFG=$c RFN_CORR_TRACE_EVERY=$e timeout 200 python tools/corr_trace.py 2>&1 | grep -v "$F"; done; done; } > $O/corr_mfma_phase_trace.txt 2>&1
{ for c in 0 2; do echo "== RFN_CORR_VARIANT=30 RFN_CORR_MFMA_CFG=$c"; RFN_CORR_MFMA_CFG=$c RFN_CORR_VARIANT=30 bash tools/pmc_corr.sh corr_l1_fused; done; echo "== RFN_CORR_VARIANT=0"; RFN_CORR_VARIANT=0 bash tools/pmc_corr.sh corr_l1_fused; } > $O/corr_mfma_pmc.txt 2>&1
{ bash tools/corr_ablate.sh 0; RFN_CORR_MFMA_CFG=2 bash tools/corr_ablate.sh 30; } 2>&1 | grep -v "$F" > $O/corr_mfma_ablation.txt
