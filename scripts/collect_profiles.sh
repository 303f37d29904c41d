#!/bin/bash
# usage (build container): bash scripts/collect_profiles.sh <tag of the bench/stats/pmc/timelines run> <tag of the small/models/features run>
# copies the judged summaries of scripts/artifacts.sh runs (gpurun_out/<tag>/) into profiles/r04_*
A=gpurun_out/${1:-r04b}
B=gpurun_out/${2:-r04a}
P=profiles
R=r04
cp $A/bench.json $P/${R}_bench.json
cp $A/kernel_summary.txt $P/${R}_kernel_summary.txt
cp $A/kernel_summary_fixed_ext.txt $P/${R}_kernel_summary_fixed_ext.txt
cp $A/kernel_stats.csv $P/${R}_kernel_stats.csv
cp $A/pmc_summary.txt $P/${R}_pmc_summary.txt
cp $A/pmc_traffic_fixed_ext.json $P/${R}_pmc_traffic_fixed_ext.json
cp $A/pmc_traffic_cycling.json $P/pmc_traffic.json
for t in 1M_fixed_ext 1M_cycling_torch 5M_dnloss_fixed 5M_dnloss_cycling 5M_dnloss_fixed_full_sort 1M_dnloss_cycling; do cp $A/timeline_$t.txt $P/${R}_timeline_$t.txt; done
for b in bench_10k_graphed_cycling bench_100k_graphed_cycling bench_1M_dnloss bench_1M_dnloss_fixed bench_5M_dnloss bench_5M_dnloss_fixed bench_5M_dnloss_fixed_full_sort bench_5M_dnloss_fixed_buckets \
         bench_5M_dnloss_fixed_buckets_sparse_rehearsal bench_100k bench_10k bench_10k_graphed bench_features; do [ -s $B/$b.json ] && cp $B/$b.json $P/${R}_$b.json; done
cp $B/timeline_1M_features_fixed.txt $P/${R}_timeline_1M_features_fixed.txt
cp $B/timeline_1M_features_fixed_dense.txt $P/${R}_timeline_1M_features_fixed_dense.txt
cp $B/bench_features_dense.json $P/${R}_bench_features_dense.json
# instruction-class tables of the compositing kernels' trip loops and the class-weighted issue budget (hipcc -S of the current source)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -fno-fast-math -S --cuda-device-only -o /tmp/blend_isa.s collab_splats_amd/csrc/blend.hip 2>/dev/null
python scripts/isa_table.py /tmp/blend_isa.s 'blend_bwd_kernelILi4ELi2ELb0ELb1ELi0E' --dump > $P/${R}_isa_blend_bwd.txt
python scripts/isa_table.py /tmp/blend_isa.s 'blend_fwd_kernelILi4ELi2ELi0ELb0E' --dump > $P/${R}_isa_blend_fwd.txt
python scripts/valu_budget.py /tmp/blend_isa.s $P/pmc_traffic.json > $P/${R}_valu_budget.json
ls $P | grep ${R} | wc -l
