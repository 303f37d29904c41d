#!/bin/bash
# usage (build container): bash scripts/collect_profiles.sh <tag of the bench/stats/pmc/timelines run> <tag of the small/models/features run>
# copies the judged summaries of scripts/artifacts.sh runs (gpurun_out/<tag>/) into profiles/r05_*
A=gpurun_out/${1:-r05b}
B=gpurun_out/${2:-r05a}
P=profiles
R=r05
cp $A/bench.json $P/${R}_bench.json
for f in bench_details bench_driver_protocol bench_driver_protocol_details; do [ -s $A/$f.json ] && cp $A/$f.json $P/${R}_$f.json; done
cp $A/kernel_summary.txt $P/${R}_kernel_summary.txt
cp $A/kernel_summary_fixed_ext.txt $P/${R}_kernel_summary_fixed_ext.txt
cp $A/kernel_stats.csv $P/${R}_kernel_stats.csv
cp $A/pmc_summary.txt $P/${R}_pmc_summary.txt
cp $A/pmc_traffic_fixed_ext.json $P/${R}_pmc_traffic_fixed_ext.json
cp $A/pmc_traffic_cycling.json $P/pmc_traffic.json
for t in 1M_fixed_ext 1M_cycling_torch 5M_dnloss_fixed 5M_dnloss_cycling 5M_dnloss_fixed_full_sort 1M_dnloss_cycling; do cp $A/timeline_$t.txt $P/${R}_timeline_$t.txt; done
for b in bench_10k_graphed_cycling bench_100k_graphed_cycling bench_1M_dnloss bench_1M_dnloss_fixed bench_5M_dnloss bench_5M_dnloss_fixed bench_5M_dnloss_fixed_full_sort bench_5M_dnloss_fixed_buckets \
         bench_5M_dnloss_fixed_buckets_sparse_rehearsal bench_5M_shared_rccl_world1 bench_100k bench_10k bench_10k_graphed bench_features; do [ -s $B/$b.json ] && cp $B/$b.json $P/${R}_$b.json; done
cp $B/timeline_1M_features_fixed.txt $P/${R}_timeline_1M_features_fixed.txt
# instruction-class tables of the compositing kernels' trip loops and the class-weighted issue budget (hipcc -S of the current source)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -fno-fast-math -fno-slp-vectorize -S --cuda-device-only -o /tmp/blend_isa.s collab_splats_amd/csrc/blend.hip 2>/dev/null
python scripts/hot_loop.py /tmp/blend_isa.s 'blend_bwd_kernelILi4ELi2ELb0ELb1ELi0E' row_half_mirror --dump > $P/${R}_isa_blend_bwd.txt
python scripts/hot_loop.py /tmp/blend_isa.s 'blend_fwd_kernelILi4ELi2ELi0ELb0E' --dump > $P/${R}_isa_blend_fwd.txt
ls $P | grep ${R} | wc -l
