#!/bin/bash
# copies the results of tools/refresh_profiles.sh (gpurun_out/final/) into profiles/ under the round tag: bash tools/publish_profiles.sh r02
T=${1:-r02}; F=gpurun_out/final
cp $F/kernel_stats.csv profiles/${T}_bench_b256_kernel_stats.csv
cp $F/default.json profiles/${T}_bench_b256_default.json
cp $F/under_rocprof.json profiles/${T}_bench_b256_under_rocprof.json
python tools/pmc_traffic_summary.py $F/pmc_FETCH_SIZE $F/pmc_WRITE_SIZE $T > /dev/null
python tools/profile_summary.py ${T}_bench_b256 profiles/${T}_bench_b256_kernel_stats.csv profiles/${T}_bench_b256_under_rocprof.json profiles/${T}_bench_b256_default.json 'cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof -- python3 $REPO/bench.py --no-cpu-baseline'
python tools/pmc_mfma_summary.py $F/pmc_mfma $T > /dev/null
cp $F/vae_layers.md profiles/${T}_vae_layers.md
cp $F/sweep.md profiles/${T}_sweep.md
cp $F/sample_kernel_stats.csv profiles/${T}_sample_b1_kernel_stats.csv
(grep -v amdgpu $F/sample_under_rocprof.log; echo "-- un-profiled, 50 steps:"; grep -v amdgpu $F/sample.log) > profiles/${T}_sample_b1_under_rocprof.log
