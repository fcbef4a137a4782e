#!/bin/bash
# Runs on the GPU box (one MI355X): full GPU test suite, default bench line, rocprofv3 kernel-trace summary of the same
# command, VAE per-layer tables.  Everything lands under gpurun_out/final/; tools/profile_summary.py turns it into profiles/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
python -c "from kalle_audio_amd.build import source_stamp; print(source_stamp())" > $O/source_stamp.txt
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
timeout -k 10 600 python bench.py 2> $O/default.err | tail -n 1 > $O/default.json || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline 2> $O/prof.err | tail -n 1 > $O/under_rocprof.json || exit 1
cd $R
rm -f $O/vae_layers.md
for B in 1 2; do
  timeout -k 10 200 python tools/vae_bench.py $B f32 --layers --md $O/vae_layers.md > $O/vae_dec_$B.log 2>&1 || exit 1
done
for B in 1 2; do
  timeout -k 10 200 python tools/vae_bench.py $B f32 --layers --encode --md $O/vae_layers.md > $O/vae_enc_$B.log 2>&1 || exit 1
done
for B in 1 2 8; do timeout -k 10 200 python tools/vae_bench.py $B f32 > $O/vae_$B.log 2>&1 || exit 1; done
timeout -k 10 300 python tools/llasa_bench.py 16 1024 3 --infer > $O/llasa.log 2>&1 || exit 1
cp $(find $O/prof -name "*kernel_stats.csv" | head -n 1) $O/kernel_stats.csv
rm -rf $O/prof
echo done
# sweep of SURVEY 8(d)'s axes and the sampling path
timeout -k 10 900 python tools/bench_sweep.py $O/sweep.md > $O/sweep.log 2>&1 || exit 1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sprof -- python3 $R/tools/sample_bench.py 1 20 > $O/sample_under_rocprof.log 2> $O/sprof.err || exit 1
cp $(find $O/sprof -name "*kernel_stats.csv" | head -n 1) $O/sample_kernel_stats.csv
rm -rf $O/sprof
cd $R
timeout -k 10 300 python tools/sample_bench.py 1 50 > $O/sample.log 2>&1 || exit 1
echo extras done
# PMC passes (each its own run, counters only + kernel trace): HBM-side fetch / write bytes and MFMA-pipe busy cycles
if [ "$1" = "pmc" ]; then
  cd /tmp
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_$C -- python3 $R/tools/bench_one_step.py 256 1 > $O/pmc_$C.log 2>&1 || exit 1
    find $O/pmc_$C -type f ! -name "*counter_collection.csv" -delete
  done
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 $R/tools/bench_one_step.py 256 1 > $O/pmc_mfma.log 2>&1 || exit 1
  find $O/pmc_mfma -type f ! -name "*counter_collection.csv" -delete
  cd $R
  echo pmc done
fi
