// Micro-benchmark: issue rate of scalar vs packed f32 VALU ops at 4 waves/SIMD (1024-thread WGs, 1 WG per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
#define REP 64
template <int MODE> __global__ __launch_bounds__(1024) void k(float *out, int iters, float s) {
    float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 0.5f, a2 = a0 + 0.25f, a3 = a0 + 0.125f;
    float2v p0 = {a0, a1}, p1 = {a2, a3};
    float2v sv = {s, s}, sv2 = {s * 0.5f, s * 0.25f};
    float2v p2 = {a0 + 2.0f, a1 + 2.0f}, p3 = {a2 + 2.0f, a3 + 2.0f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (MODE == 0) {  // 4 scalar muls
                asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));
            } else if (MODE == 1) {  // 2 packed muls (same flops as mode 0)
                asm volatile("v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2" : "+v"(p0), "+v"(p1) : "v"(sv));
            } else if (MODE == 2) {  // 4 packed muls (2x flops of mode 0, same instruction count)
                asm volatile("v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2"
                             : "+v"(p0), "+v"(p1) : "v"(sv));
            } else if (MODE == 3) {  // 4 dependent scalar muls
                asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1"
                             : "+v"(a0) : "v"(s));
            } else if (MODE == 4) {  // 4 dependent packed muls
                asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1"
                             : "+v"(p0) : "v"(sv));
            } else if (MODE == 5) {  // 4 fma
                asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));
            } else if (MODE == 6) {  // 4 rsq
                asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            } else if (MODE == 7) {  // 4 packed adds
                asm volatile("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2"
                             : "+v"(p0), "+v"(p1) : "v"(sv));
            } else if (MODE == 9) {  // 4 packed fma, distinct accumulators, shared multiplicands
                asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %0, %0, %3, %2\n v_pk_fma_f32 %1, %1, %3, %2"
                             : "+v"(p0), "+v"(p1) : "v"(sv), "v"(sv2));
            } else if (MODE == 10) {  // 4 packed fma, three distinct register pairs per instruction
                asm volatile("v_pk_fma_f32 %0, %1, %4, %5\n v_pk_fma_f32 %1, %2, %5, %4\n v_pk_fma_f32 %2, %3, %4, %5\n v_pk_fma_f32 %3, %0, %5, %4"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(sv), "v"(sv2));
            } else if (MODE == 11) {  // 2 packed fma + 2 scalar fma interleaved
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_fma_f32 %2, %2, %6, %6\n v_pk_fma_f32 %1, %1, %5, %4\n v_fma_f32 %3, %3, %6, %6"
                             : "+v"(p0), "+v"(p1), "+v"(a0), "+v"(a1) : "v"(sv), "v"(sv2), "v"(s));
            } else if (MODE == 12) {  // 4 packed mul with op_sel broadcast of one half
                asm volatile("v_pk_mul_f32 %0, %0, %2 op_sel_hi:[1,0]\n v_pk_mul_f32 %1, %1, %2 op_sel_hi:[1,0]\n v_pk_mul_f32 %0, %0, %2 op_sel_hi:[1,0]\n v_pk_mul_f32 %1, %1, %2 op_sel_hi:[1,0]"
                             : "+v"(p0), "+v"(p1) : "v"(sv));
            } else if (MODE == 13) {  // 4 v_cndmask
                asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s) : "vcc");
            } else if (MODE == 8) {  // 4 v_pk_mov_b32
                asm volatile("v_pk_mov_b32 %0, %1, %0 op_sel:[0,1]\n v_pk_mov_b32 %1, %0, %1 op_sel:[1,0]\n v_pk_mov_b32 %0, %1, %0 op_sel:[0,1]\n v_pk_mov_b32 %1, %0, %1 op_sel:[1,0]"
                             : "+v"(p0), "+v"(p1));
            }
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE> void run(const char *name, float *d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    k<MODE><<<256, 1024>>>(d, 10, 1.0f);
    hipEventRecord(e0);
    k<MODE><<<256, 1024>>>(d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves x iters x REP x 4 instructions
    double instr = 4.0 * iters * REP * 4;
    printf("%-28s %8.3f ms  -> %.2f cycles/instr/SIMD at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / instr);
}
int main() {
    float *d; hipMalloc(&d, 256 * 1024 * 4);
    run<0>("4x v_mul_f32 indep", d);
    run<1>("2x v_pk_mul_f32 indep (x2 cnt)", d);
    run<2>("4x v_pk_mul_f32", d);
    run<3>("4x v_mul_f32 dependent", d);
    run<4>("4x v_pk_mul_f32 dependent", d);
    run<5>("4x v_fma_f32", d);
    run<6>("4x v_rsq_f32", d);
    run<7>("4x v_pk_add_f32", d);
    run<8>("4x v_pk_mov_b32", d);
    run<9>("4x v_pk_fma_f32 shared srcs", d);
    run<10>("4x v_pk_fma_f32 distinct srcs", d);
    run<11>("2x pk_fma + 2x fma mixed", d);
    run<12>("4x v_pk_mul_f32 op_sel bcast", d);
    run<13>("4x v_cndmask_b32", d);
    return 0;
}
