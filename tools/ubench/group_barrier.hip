// Cost of a per-step hand-off among the 16 workgroups of a row block (what a persistent encoder-GRU recurrence would need instead of one
// launch per step): every workgroup publishes 256 floats, bumps its group's counter, waits for the other 15, then reads their 16 x 256
// floats.  Two variants:
//   agent : agent-scope (sc1, through memory) atomic stores / loads / counter -- correct wherever the workgroups run;
//   xcd   : the group's workgroups are chosen from ONE XCD (ids congruent mod 8: the dispatcher deals ids round-robin to the 8 XCDs; each
//           workgroup reports its XCC_ID so that the assumption is checked) and use workgroup-scope (sc0: bypass the CU's vector L1, hit
//           the XCD's shared L2) accesses -- coherent through that L2 without a trip to memory.
// No device-scope fences (an L2 write-back / invalidate per step is what made the fused attention combine slow).  Bounded spins.
// build: hipcc --offload-arch=gfx950 -O3 group_barrier.hip -o group_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int SCOPE>
__global__ __launch_bounds__(256) void k(float* buf, int* cnt, int* fail, float* out, int* xcc, int iters, int gsize, int same_xcd, int zero) {
    const int wg = blockIdx.x, tid = threadIdx.x;
    // same_xcd: group g = (xcd = g % 8, slot = g / 8), member j = workgroup ((j * (G/8) + slot) * 8 + xcd), G groups = gridDim.x / gsize
    const int ngroups = gridDim.x / gsize;
    int group, member_stride, first;
    if (same_xcd) { const int xcd = wg % 8, q = wg / 8, slots = ngroups / 8; group = (q % slots) * 8 + xcd; first = (q % slots) * 8 + xcd; member_stride = slots * 8; }
    else { group = wg / gsize; first = group * gsize; member_stride = 1; }
    if (tid == 0) xcc[wg] = __builtin_amdgcn_s_getreg((4 << 11) | (0 << 6) | 20) & 0xf;     // hwreg(HW_REG_XCC_ID = 20), bits [3:0]
    float acc = 0.f;
    for (int t = 0; t < iters; ++t) {
        float* b = buf + (size_t)(t & 1) * gridDim.x * 256;
        __hip_atomic_store(b + wg * 256 + tid, (float)(t + wg) + acc * 1e-9f, __ATOMIC_RELAXED, SCOPE);
        __builtin_amdgcn_s_waitcnt(0);                 // vmcnt(0): this thread's store has been acknowledged
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(cnt + group, 1, __ATOMIC_RELAXED, SCOPE);
            int spins = 0;
            while (__hip_atomic_load(cnt + group, __ATOMIC_RELAXED, SCOPE) < gsize * (t + 1)) {
                if (++spins > 4000000) { *fail = 1; break; }
            }
        }
        __syncthreads();
        if (*(volatile int*)fail) return;
#pragma unroll 4
        for (int j = 0; j < gsize; ++j) {
            float* p = b + (first + j * member_stride) * 256 + tid;
            // one XCD: a workgroup-scope LOAD may hit this CU's vector L1 (stale: measured wrong sums in every workgroup); an atomic
            // read-modify-write executes in the XCD's L2 and returns what the other CUs' write-through stores left there (`zero` is a
            // run-time 0: the compiler turns an RMW with a constant identity operand back into a load)
            if (SCOPE == __HIP_MEMORY_SCOPE_WORKGROUP) acc += __uint_as_float(__hip_atomic_fetch_or(reinterpret_cast<unsigned*>(p), (unsigned)zero, __ATOMIC_RELAXED, SCOPE));
            else acc += __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE);
        }
    }
    out[wg * 256 + tid] = acc;
}

int main() {
    const int grid = 256, iters = 2000, gsize = 16;
    float *buf, *out; int *cnt, *fail, *xcc;
    (void)hipMalloc(&buf, sizeof(float) * 2 * grid * 256); (void)hipMalloc(&out, sizeof(float) * grid * 256);
    (void)hipMalloc(&cnt, sizeof(int) * grid); (void)hipMalloc(&fail, sizeof(int)); (void)hipMalloc(&xcc, sizeof(int) * grid);
    for (int variant = 0; variant < 4; ++variant) {
        const int same_xcd = variant >= 2;
        (void)hipMemset(cnt, 0, sizeof(int) * grid); (void)hipMemset(fail, 0, sizeof(int));
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        if (same_xcd) hipLaunchKernelGGL(k<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(grid), dim3(256), 0, 0, buf, cnt, fail, out, xcc, iters, gsize, 1, 0);
        else hipLaunchKernelGGL(k<__HIP_MEMORY_SCOPE_AGENT>, dim3(grid), dim3(256), 0, 0, buf, cnt, fail, out, xcc, iters, gsize, 0, 0);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        int f; (void)hipMemcpy(&f, fail, sizeof(int), hipMemcpyDeviceToHost);
        static float o[256 * 256]; (void)hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
        static int x[256]; (void)hipMemcpy(x, xcc, sizeof(x), hipMemcpyDeviceToHost);
        // expected: sum over t, j of (t + member id)
        int bad = 0, xcd_ok = 1;
        for (int wg = 0; wg < grid; ++wg) {
            double want = 0; int ngroups = grid / gsize;
            int first, stride; if (same_xcd) { int xcd = wg % 8, q = wg / 8, slots = ngroups / 8; first = (q % slots) * 8 + xcd; stride = slots * 8; } else { first = wg / gsize * gsize; stride = 1; }
            for (int j = 0; j < gsize; ++j) { want += (double)iters * (first + j * stride) + (double)iters * (iters - 1) / 2; if (same_xcd && x[first + j * stride] != x[wg]) xcd_ok = 0; }
            if (fabs(o[wg * 256] - want) > 1e-3 * want) ++bad;
        }
        printf("%s: %.2f us per step (%d steps)%s  wrong sums in %d of %d workgroups%s   XCC ids of workgroups 0..15: ", same_xcd ? "one XCD, L2-coherent (sc0)" : "any XCD, through memory (sc1)",
               ms * 1e3 / iters, iters, f ? "  SPIN TIMEOUT" : "", bad, grid, same_xcd ? (xcd_ok ? ", groups on one XCD: yes" : ", groups on one XCD: NO") : "");
        for (int i = 0; i < 16; ++i) printf("%d", x[i]);
        printf("\n");
    }
    return 0;
}
