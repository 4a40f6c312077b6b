// Development micro-benchmark: interval between DEPENDENT kernel launches on one stream (the floor under the streaming
// back-end's ~140 launches per frame), for an empty kernel and for a copy kernel of the size of one fs_k_iterate launch.
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench/launch_floor.hip -o variants/launch_floor && variants/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty() {}
__global__ void k_copy(const float4 *a, float4 *b, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}
struct Desc { const float4 *a; float4 *b; int n; int pad[29]; };  // an episode descriptor reached through a launch list
__global__ void k_copy_indirect(const Desc *descs, const int *ids, int flip) {
    const int e = ids[blockIdx.y];
    if (e < 0) return;
    const Desc &D = descs[e];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < D.n) { if (flip) D.b[i] = D.a[i]; else ((float4 *)D.a)[i] = D.b[i]; }
}
int main() {
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int N = 2000;
    float4 *a, *b; hipMalloc(&a, 16 << 20); hipMalloc(&b, 16 << 20); hipMemset(a, 0, 16 << 20);
    for (int cfg = 0; cfg < 5; ++cfg) {
        const int wgs[5] = {1, 64, 875, 2700, 16}, thr[5] = {64, 256, 256, 256, 1024};
        for (int kind = 0; kind < 2; ++kind) {
            for (int w = 0; w < 100; ++w) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
            hipStreamSynchronize(st);
            hipEventRecord(e0, st);
            for (int i = 0; i < N; ++i) {
                if (kind == 0) hipLaunchKernelGGL(k_empty, dim3(wgs[cfg]), dim3(thr[cfg]), 0, st);
                else hipLaunchKernelGGL(k_copy, dim3(wgs[cfg]), dim3(thr[cfg]), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, wgs[cfg] * thr[cfg]);
            }
            hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%5d workgroups x %4d threads, %s: %.2f us per dependent launch\n", wgs[cfg], thr[cfg], kind ? "copy " : "empty", ms * 1e3 / N);
        }
    }
    {   // the same copy with the streaming kernels' two scalar indirections (launch list -> descriptor -> arrays)
        Desc h = {a, b, 875 * 256, {0}}; Desc *d; int *ids; int zero = 0;
        hipMalloc(&d, sizeof(Desc)); hipMalloc(&ids, 4);
        hipMemcpy(d, &h, sizeof(Desc), hipMemcpyHostToDevice); hipMemcpy(ids, &zero, 4, hipMemcpyHostToDevice);
        for (int w = 0; w < 100; ++w) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_copy_indirect, dim3(875, 1), dim3(256), 0, st, d, ids, i & 1);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  875 workgroups x  256 threads, copy through launch list + descriptor: %.2f us per dependent launch\n", ms * 1e3 / N);
    }
    return 0;
}
