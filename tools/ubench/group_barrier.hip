// Cost of a per-step hand-off among the 16 workgroups of a row block (what a persistent encoder-GRU recurrence would need instead of one
// launch per step): every workgroup publishes 256 floats with agent-scope (sc1, write-through) atomic stores, bumps its group's counter,
// waits for the other 15, then reads their 16 x 256 floats with agent-scope atomic loads.  No device-scope fences (an L2 write-back /
// invalidate per step is what made the fused attention combine slow).  Bounded spin: gives up (flag) instead of hanging.
// build: hipcc --offload-arch=gfx950 -O3 group_barrier.hip -o group_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void k(float* buf, int* cnt, int* fail, float* out, int iters, int gsize) {
    const int wg = blockIdx.x, tid = threadIdx.x;
    const int group = wg / gsize, first = group * gsize;
    float acc = 0.f;
    __shared__ int ok;
    for (int t = 0; t < iters; ++t) {
        float* b = buf + (size_t)(t & 1) * gridDim.x * 256;
        __hip_atomic_store(b + wg * 256 + tid, (float)(t + wg) + acc * 1e-9f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);                 // vmcnt(0): this thread's store has been acknowledged
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(cnt + group, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(cnt + group, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gsize * (t + 1)) {
                if (++spins > 2000000) { *fail = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            ok = 1;
        }
        __syncthreads();
        if (*(volatile int*)fail) return;
#pragma unroll 4
        for (int j = 0; j < gsize; ++j) acc += __hip_atomic_load(b + (first + j) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    out[wg * 256 + tid] = acc;
}

int main() {
    const int grid = 256, iters = 2000;
    float *buf, *out; int *cnt, *fail;
    (void)hipMalloc(&buf, sizeof(float) * 2 * grid * 256); (void)hipMalloc(&out, sizeof(float) * grid * 256);
    (void)hipMalloc(&cnt, sizeof(int) * grid); (void)hipMalloc(&fail, sizeof(int));
    for (int gsize : {16, 16, 4, 1}) {
        (void)hipMemset(cnt, 0, sizeof(int) * grid); (void)hipMemset(fail, 0, sizeof(int));
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, buf, cnt, fail, out, iters, gsize);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        int f; (void)hipMemcpy(&f, fail, sizeof(int), hipMemcpyDeviceToHost);
        float o; (void)hipMemcpy(&o, out, sizeof(float), hipMemcpyDeviceToHost);
        // expected out[0] for wg 0: sum over t, j of (t + first + j)
        printf("group of %2d workgroups: %.2f us per step (%d steps)%s  check %.0f\n", gsize, ms * 1e3 / iters, iters, f ? "  SPIN TIMEOUT" : "", o);
    }
    return 0;
}
