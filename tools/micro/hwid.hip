// Which SIMD does each wave of a 512-thread workgroup land on?  (HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh[12] se[15:13])
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/hwid.hip -o gpurun_out/hwid ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void k(unsigned* out, int regs) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main() {
    const int nb = 512;
    unsigned* d; hipMalloc(&d, nb * 8 * 4);
    hipLaunchKernelGGL(k, dim3(nb), dim3(512), 65536, 0, d, 0);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 8);
    hipMemcpy(h.data(), d, nb * 8 * 4, hipMemcpyDeviceToHost);
    int pair_same = 0, adj_same = 0;
    for (int b = 0; b < nb; ++b) {
        if (b < 6) { printf("wg %d simd:", b); for (int w = 0; w < 8; ++w) printf(" %u", (h[b * 8 + w] >> 4) & 3); printf("  cu %u\n", (h[b * 8] >> 8) & 15); }
        for (int w = 0; w < 4; ++w) pair_same += (((h[b * 8 + w] >> 4) & 3) == ((h[b * 8 + w + 4] >> 4) & 3));
        for (int w = 0; w < 8; w += 2) adj_same += (((h[b * 8 + w] >> 4) & 3) == ((h[b * 8 + w + 1] >> 4) & 3));
    }
    printf("waves (w, w+4) on the same SIMD: %d of %d pairs; waves (2k, 2k+1): %d of %d\n", pair_same, nb * 4, adj_same, nb * 4);
    return 0;
}
