#!/bin/bash
# rocprofv3 counter passes (run on the GPU box through gpurun), ONE counter group per run as the guide prescribes, for every
# workload of profiles/tools/pmc_probe.py:   FETCH_SIZE | WRITE_SIZE | TCC hit / miss | (gn_step only) SQ instruction / cycle counters
# Outputs rocpd databases under gpurun_out/pmc/<workload>/; profiles/tools/pmc_report.py turns them into profiles/*.txt + traffic.json
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for W in ${PMC_WORKLOADS:-gn_step per_sample_sdf per_sample_sdf_tiled learned_covariances config4_xyh}; do
  OUT=$R/gpurun_out/pmc/$W
  mkdir -p "$OUT"
  P="python $R/profiles/tools/pmc_probe.py $W"
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT" -o fetch -- $P > "$OUT/fetch.log" 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT" -o write -- $P > "$OUT/write.log" 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d "$OUT" -o l2 -- $P > "$OUT/l2.log" 2>&1
  if [ "$W" = gn_step ] || [ "$W" = config4_xyh ] || [ "$W" = config4_perstate ]; then      # the SQ issue / wait counters: the headline kernel and (round 6) the d = 6 family
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d "$OUT" -o sq1 -- $P > "$OUT/sq1.log" 2>&1
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM -d "$OUT" -o sq2 -- $P > "$OUT/sq2.log" 2>&1
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VMEM -d "$OUT" -o sq3 -- $P > "$OUT/sq3.log" 2>&1
  fi
  ls "$OUT" | head -20
done
python $R/profiles/tools/pmc_report.py $R/gpurun_out/pmc $R/gpurun_out/pmc/traffic.json > $R/gpurun_out/pmc/report.txt 2>&1
find $R/gpurun_out/pmc -name '*.db' -delete      # the rocpd databases are tens of MB each; the report keeps what profiles/ needs
tail -5 $R/gpurun_out/pmc/report.txt
