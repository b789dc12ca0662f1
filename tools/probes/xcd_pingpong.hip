// Flag ping-pong between two workgroups of one kernel, same XCD (block ids 0 and 8) versus different XCDs (0 and 1),
// agent-scope relaxed atomics: how long does one hand-over take when both ends share an L2, and when they do not?
// Decides whether a persistent single-XCD training-step kernel (grid barriers through one L2) can beat 8 launches.
//   hipcc --offload-arch=gfx950 -O3 xcd_pingpong.hip -o xcd_pingpong
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void pingpong(unsigned* flags, int partner_block, int iters, long long* cycles, unsigned* xcc_ids) {
    // flags[0]: written by block 0, flags[32]: written by the partner (separate cache lines)
    if (threadIdx.x != 0) return;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (blockIdx.x == 0) xcc_ids[0] = xcc & 0xf;
    if ((int)blockIdx.x == partner_block) xcc_ids[1] = xcc & 0xf;
    if (blockIdx.x == 0) {
        long long t0 = wall_clock64();
        for (int i = 1; i <= iters; ++i) {
            __hip_atomic_store(&flags[0], (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(&flags[32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)i) {}
        }
        cycles[0] = wall_clock64() - t0;
    } else if ((int)blockIdx.x == partner_block) {
        for (int i = 1; i <= iters; ++i) {
            while (__hip_atomic_load(&flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)i) {}
            __hip_atomic_store(&flags[32], (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main() {
    unsigned *flags, *xcc;
    long long* cyc;
    hipMalloc(&flags, 4096); hipMalloc(&cyc, 64); hipMalloc(&xcc, 64);
    const int iters = 2000;
    for (int partner : {8, 1, 2, 4, 16, 9}) {
        hipMemset(flags, 0, 4096);
        hipLaunchKernelGGL(pingpong, dim3(32), dim3(64), 0, 0, flags, partner, iters, cyc, xcc);
        hipDeviceSynchronize();
        long long c; unsigned ids[2];
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        hipMemcpy(ids, xcc, 8, hipMemcpyDeviceToHost);
        printf("block 0 (XCC %u) <-> block %2d (XCC %u): %.3f us per round trip (two hand-overs)\n", ids[0], partner, ids[1],
               c / 100.0 / iters);
    }
    return 0;
}
