"""generates pkfma_bench.hip: does v_pk_fma_f32 (SGPR-pair weight, op_sel broadcast) lose cycles when its two 64-bit VGPR
operands (x pair, accumulator pair) sit in the same register banks (index mod 4)?"""
def body(acc_regs, x_regs):
    lines = []
    n = 0
    for j, xr in enumerate(x_regs):
        for i in range(8):
            a0, a1 = acc_regs[(2 * i) * 4 + j], acc_regs[(2 * i + 1) * 4 + j]
            s = 4 + 2 * i
            lines.append(f"v_pk_fma_f32 v[{a0}:{a0+1}], s[{s}:{s+1}], v[{xr}:{xr+1}], v[{a0}:{a0+1}] op_sel_hi:[0,1,1]")
            lines.append(f"v_pk_fma_f32 v[{a1}:{a1+1}], s[{s}:{s+1}], v[{xr}:{xr+1}], v[{a1}:{a1+1}] op_sel:[1,0,0] op_sel_hi:[1,1,1]")
    return lines

modes = {
    # 0: compiler-like: accumulators packed at consecutive even registers (banks alternate), x pairs likewise
    0: ([2 * k for k in range(64)], [128, 130, 132, 134]),
    # 1: conflict-free: accumulators at 4k (banks 0,1), x pairs at 4k+2 (banks 2,3)
    1: ([4 * k for k in range(64)], [2, 6, 10, 14]),
    # 2: always the same banks: accumulators at 4k+2... (needs 4k+2 < 256), x at 4k+2 too
    2: ([4 * k + 2 for k in range(60)] + [4 * k for k in range(4)], [242 - 0, 246, 250, 254]),
    # 3: plain v_fma_f32-free reference: x operand also an SGPR-free VGPR but accumulators 4k and x 4k+2, half the instructions
    #    independent accumulate chains only (same as 1) - sanity for the 4-cycle rate
}
src = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstdlib>']
for m, (acc, xr) in modes.items():
    if m == 2:
        acc = [4 * k + 2 for k in range(56)] + [4 * k for k in range(8)]
        xr = [226, 230, 234, 238]
    asm = body(acc, xr)
    text = "\\n".join(["s_mov_b32 s30, %0", "1:"] + asm + ["s_sub_u32 s30, s30, 1", "s_cmp_lg_u32 s30, 0", "s_cbranch_scc1 1b"])
    clob = ",".join([f'"v{r}"' for r in range(256)] + [f'"s{r}"' for r in range(4, 31)] + ['"scc"'])
    src.append(f'__global__ __launch_bounds__(512) void k{m}(int iters) {{ asm volatile("{text}" :: "s"(iters) : {clob}); }}')
src.append('''
int main(int argc, char** argv) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    void (*ks[3])(int) = {k0, k1, k2};
    const char* names[3] = {"packed-even (compiler-like)", "banks disjoint", "banks equal"};
    for (int w = 1; w <= 2; ++w)
    for (int m = 0; m < 3; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(ks[m], dim3(256), dim3(256 * w), 0, 0, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms * 1e-3 * clk * 1e3 / ((double)iters * 64 * w);
        printf("%d wave(s)/SIMD  %-30s %.3f ms  %.2f cycles per v_pk_fma_f32 (at %d MHz)  -> %.1f TFLOP/s chip\\n", w, names[m], ms, cyc,
               clk / 1000, 256.0 * 4 * w * iters * 64 * 256 / (ms * 1e-3) / 1e12);
    }
    return 0;
}''')
open("pkfma_bench.hip", "w").write("\n".join(src))
