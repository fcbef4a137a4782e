# PMC passes over the VAE conv path (run on the GPU box): bash tools/profile_vae_pmc.sh [B]; then
# python tools/pmc_vae_summary.py gpurun_out/vaepmc/FETCH_SIZE gpurun_out/vaepmc/WRITE_SIZE r03 B
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; B=${1:-2}
O=$R/gpurun_out/vaepmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$C -- python3 $R/tools/vae_bench.py $B f32 > $O/$C.log 2>&1 || exit 1
  find $O/$C -type f ! -name "*counter_collection.csv" ! -name "*kernel_trace.csv" -delete
done
echo vae pmc done
