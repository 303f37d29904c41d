#!/bin/bash
# usage (build container): bash scripts/collect_profiles.sh <gpurun_out tag> -- copy the judged summaries of a round3_artifacts.sh run into profiles/r03_*
T=gpurun_out/${1:-r03a}
P=profiles
cp $T/bench.json $P/r03_bench.json
cp $T/kernel_summary.txt $P/r03_kernel_summary.txt
cp $T/kernel_summary_fixed_ext.txt $P/r03_kernel_summary_fixed_ext.txt
cp $T/kernel_stats.csv $P/r03_kernel_stats.csv
cp $T/pmc/summary.json $P/r03_pmc_summary.json
cp $T/pmc_traffic.json $P/r03_pmc_traffic_fixed_ext.json
cp $T/pmc_traffic_cycling.json $P/pmc_traffic.json
for t in 1M_fixed_ext 1M_cycling_torch 5M_dnloss_fixed 100k_fixed_ext 10k_fixed_ext 1M_dnloss_cycling; do cp $T/timeline_$t.txt $P/r03_timeline_$t.txt; done
for b in bench_1M_dnloss bench_1M_dnloss_fixed bench_5M_dnloss bench_5M_dnloss_fixed bench_100k bench_10k bench_1M_dnloss_fixed_buckets bench_5M_dnloss_fixed_buckets bench_10k_graphed bench_10k_fixed_ext bench_100k_fixed_ext; do [ -s $T/$b.json ] && cp $T/$b.json $P/r03_$b.json; done
cp $T/features/features_bench.json $P/r03_features_bench.json
cp $T/features/features_kernel_summary.txt $P/r03_features_kernel_summary.txt
cp $T/features/features_pmc.json $P/r03_features_pmc.json
# instruction-class tables of the compositing kernels' trip loops (hipcc -S of the current source)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -fno-fast-math -S --cuda-device-only -o /tmp/blend_isa.s collab_splats_amd/csrc/blend.hip 2>/dev/null
python scripts/isa_table.py /tmp/blend_isa.s 'blend_bwd_kernelILi4ELi2ELb0ELb1ELi0E' --dump > $P/r03_isa_blend_bwd.txt
python scripts/isa_table.py /tmp/blend_isa.s 'blend_fwd_kernelILi4ELi2ELi0ELb0E' --dump > $P/r03_isa_blend_fwd.txt
ls $P | grep r03 | wc -l
