// Micro-benchmark: cost and correctness of a barrier + data exchange between workgroups of ONE XCD (workgroup ids
// congruent mod 8) versus device-wide agent-scope synchronisation, on MI355X (8 XCDs, one L2 each).
//   mode 0: agent-scope release / acquire fences around an agent-scope atomic counter (what a device-wide barrier needs)
//   mode 1: same-XCD only: stores drained with s_waitcnt, arrival + polling with L2 atomics (RMW), vector L1 invalidated
//           with `buffer_inv sc1`, NO L2 write-back
// Each round every participating workgroup writes a 4 KiB slice of a shared buffer, synchronises, and sums the slices of
// all its partners; the sums are checked on the host.  Spins give up after a bounded number of polls (error flag).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define SLICE 1024  // floats per workgroup
#define MAX_POLLS (1u << 18)

template <int MODE> __device__ __forceinline__ bool group_barrier(unsigned *counter, unsigned n, unsigned &phase, int *err) {
    if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else __builtin_amdgcn_s_waitcnt(0);  // all stores of this wave have reached L2 (vector L1 is write-through)
    __syncthreads();
    ++phase;
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned target = phase * n;
        if (MODE == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            unsigned polls = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target)
                if (++polls > MAX_POLLS) { ok = false; break; }
        } else if (MODE == 1) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            unsigned polls = 0;  // RMW polls execute in L2: never served from a stale L1 line
            while (__hip_atomic_fetch_add(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target)
                if (++polls > MAX_POLLS) { ok = false; break; }
        } else {  // MODE 2: counter through agent-scope relaxed atomics, data path as in mode 1
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned polls = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++polls > MAX_POLLS) { ok = false; break; }
            }
        }
        if (!ok) { atomicExch(err, 1); printf("timeout: block %d phase %u counter %u target %u\n", blockIdx.x, phase, __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), phase * n); }
    }
    __syncthreads();
    if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    else asm volatile("buffer_inv sc1" ::: "memory");
    return ok;
}

// groups: workgroups g, g+8, g+16, ... (k_per_group of them) form group g (same XCD if placement is b % 8)
template <int MODE> __global__ __launch_bounds__(256) void k(float *buf, unsigned *counters, float *sums, int *err, int *xcc,
                                                          int k_per_group, int rounds) {
    const int g = blockIdx.x % 8, m = blockIdx.x / 8;  // group, member
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[blockIdx.x] = (int)(id & 0xf);
    }
    float *gbuf = buf + (size_t)g * k_per_group * SLICE;
    unsigned phase = 0;
    float acc = 0.0f;
    for (int r = 0; r < rounds; ++r) {
        for (int i = threadIdx.x; i < SLICE; i += blockDim.x) gbuf[m * SLICE + i] = (float)(r * 31 + m * 7 + (i & 3));
        if (!group_barrier<MODE>(counters + g, k_per_group, phase, err)) return;
        for (int p = 0; p < k_per_group; ++p)
            for (int i = threadIdx.x; i < SLICE; i += blockDim.x) acc += gbuf[p * SLICE + i];
        if (!group_barrier<MODE>(counters + g, k_per_group, phase, err)) return;  // before the slices are overwritten
    }
    // block reduce
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) sums[blockIdx.x] = red[0];
}

template <int MODE> void run(const char *name, int k_per_group, int rounds) {
    const int blocks = 8 * k_per_group;
    float *buf, *sums; unsigned *counters; int *err, *xcc;
    hipMalloc(&buf, sizeof(float) * blocks * SLICE); hipMalloc(&sums, sizeof(float) * blocks);
    hipMalloc(&counters, sizeof(unsigned) * 8); hipMalloc(&err, sizeof(int)); hipMalloc(&xcc, sizeof(int) * blocks);
    hipMemset(counters, 0, sizeof(unsigned) * 8); hipMemset(err, 0, sizeof(int));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    void *args[] = {&buf, &counters, &sums, &err, &xcc, &k_per_group, &rounds};
    hipEventRecord(e0);
    hipError_t rc = hipLaunchCooperativeKernel((const void *)k<MODE>, dim3(blocks), dim3(256), args, 0, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> h(blocks); std::vector<int> hx(blocks); int herr = 0;
    hipMemcpy(h.data(), sums, sizeof(float) * blocks, hipMemcpyDeviceToHost);
    hipMemcpy(hx.data(), xcc, sizeof(int) * blocks, hipMemcpyDeviceToHost);
    hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost);
    double want = 0.0;  // float accumulation order differs per thread; compare loosely but tightly enough to see stale data
    for (int r = 0; r < rounds; ++r) for (int p = 0; p < k_per_group; ++p) for (int i = 0; i < SLICE; ++i) want += r * 31 + p * 7 + (i & 3);
    int bad = 0, split = 0;
    for (int b = 0; b < blocks; ++b) { if (fabs(h[b] - want) > 1e-4 * want) ++bad; if (hx[b] != hx[b % 8]) ++split; }
    printf("%-34s groups of %2d: launch %s, %7.3f ms for %d rounds -> %6.2f us per barrier; wrong sums %d, timeouts %d, "
           "groups spanning XCDs %d (xcc of blocks 0..7: %d %d %d %d %d %d %d %d)\n", name, k_per_group,
           rc == hipSuccess ? "ok" : hipGetErrorString(rc), ms, rounds, ms * 1e3 / (2.0 * rounds), bad, herr, split,
           hx[0], hx[1], hx[2], hx[3], hx[4], hx[5], hx[6], hx[7]);
    hipFree(buf); hipFree(sums); hipFree(counters); hipFree(err); hipFree(xcc);
}

int main() {
    for (int kpg : {2, 4, 16}) {
        run<0>("agent-scope fences + atomics", kpg, 200);
        run<1>("same-XCD: L2 atomics + buffer_inv", kpg, 200);
        run<2>("same-XCD data, agent atomics", kpg, 200);
    }
    return 0;
}
