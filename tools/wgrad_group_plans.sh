# grouped weight-gradient plans (whole tiles, slices of the rest) at the bench's token counts; run on the GPU box
cd ${GRAFT_REPO_ROOT:-.}
for T in 2016 8064 32256; do
  KALLE_GEMM_DEBUG=1 python tools/wgrad_group_bench.py $T 1 2>&1 | grep -v amdgpu.ids | sort | uniq -c
done
for P in 666,1 512,2 512,3 512,4 512,6 512,8 512,12 256,2 256,3 256,4; do KALLE_WGRAD_GROUP_PLAN=$P python tools/wgrad_group_bench.py 32256 1 2>&1 | grep tokens; done
