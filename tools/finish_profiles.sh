#!/bin/bash
# After tools/collect_profiles.sh <tag> and tools/collect_ba_profiles.sh <tag>_ba (run on the GPU box, merged back into gpurun_out/):
# summarise the PMC passes, stamp profiles/pmc_traffic.json and copy the kernel statistics into profiles/ as round <rNN>.
#   usage: tools/finish_profiles.sh <tag> <rNN>      e.g. tools/finish_profiles.sh r03q r03
set -e
TAG=$1; R=$2; G=gpurun_out/$TAG
python3 tools/pmc_summary.py $G/pmc_fetch $G/pmc_write $G/pmc_tcc $G/pmc_insts $G/pmc_busy $G/pmc_mem > /tmp/pmc_all.csv
BIG=$(grep "k_fast_wave" /tmp/pmc_all.csv | sort -t, -k2 -n | tail -1 | cut -d, -f2)   # the all-levels launch of 64 frames
{
  echo "# rocprofv3 --pmc, one counter group per pass (tools/collect_profiles.sh $TAG 5), DVS_NO_OVERLAP=1 so every kernel is one launch per 64-frame batch (k_resize4: one launch per level; k_describe = orientation + descriptors; k_match_fp4<4> = the match of 64 frame pairs (FP4 matrix instruction, operands built in the kernel); k_fast_wave<48>: the all-levels launch, grid $BIG)"
  echo "# per-launch averages; FETCH_SIZE / WRITE_SIZE in KB as reported (gfx950: double FETCH_SIZE for HBM bytes, MI355X_MICROARCH.md); VALUBusy / SALUBusy / LDSBankConflict in percent"
  head -1 /tmp/pmc_all.csv
  grep "dvs::" /tmp/pmc_all.csv | grep -v "k_test_spin\|k_probe" | sed 's/^void //' | awk -F, -v big=$BIG '!/k_fast_wave/ || $2 == big'
} > profiles/${R}_pmc_summary.csv
python3 tools/make_pmc_traffic.py profiles/${R}_pmc_summary.csv
for k in overlap isolated; do cp $G/stats_$k/stats_${k}_kernel_stats.csv profiles/${R}_kernel_stats_$k.csv; done
B=gpurun_out/${TAG}_ba
[ -d $B ] && cp $B/ba_prof/ba_prof_kernel_stats.csv profiles/${R}_ba_eval_kernel_stats.csv && cp $B/ba_prof_batched/ba_prof_batched_kernel_stats.csv profiles/${R}_ba_eval_batched_kernel_stats.csv \
  && cp $B/ba_lm_prof/ba_lm_prof_kernel_stats.csv profiles/${R}_ba_lm_kernel_stats.csv
grep "k_fast_wave\|k_match_fp4" profiles/${R}_pmc_summary.csv | cut -c1-200
