// Victim kernel of tools/micro/xcc_replay_probe.py: one work-group per CU that owns `lds_bytes` of LDS and a full register file's worth of
// waves for about `cycles` shader clocks — the footprint of this repository's sweep / weight-gradient kernels without any of their code.
// It writes a pattern into its LDS, spins, and checks the pattern at the end (a context save / restore that loses LDS shows as err != 0).
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o dbg/libspin_lds.so tools/micro/spin_lds.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(512) void spin_lds_kernel(unsigned* err, long long cycles, int lds_words) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < lds_words; i += blockDim.x) lds[i] = 0x9e3779b9u * (unsigned)(i + 1) + blockIdx.x;
    __syncthreads();
    const long long t0 = clock64();
    float acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = (float)(threadIdx.x + i);
    while (clock64() - t0 < cycles) {
#pragma unroll
        for (int i = 0; i < 64; ++i) acc[i] = acc[i] * 1.0000001f + 0.5f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) s += acc[i];
    __syncthreads();
    unsigned bad = 0;
    for (int i = threadIdx.x; i < lds_words; i += blockDim.x) bad += (lds[i] != 0x9e3779b9u * (unsigned)(i + 1) + blockIdx.x);
    if (bad) atomicAdd(err, bad);
    if (s == 12345.678f) atomicAdd(err + 1, 1u);          // keeps `acc` alive
}

// The same victim with a dynamically indexed private array: it needs SCRATCH memory (private_segment_fixed_size > 0), like the spilling
// sweep kernels and wgrad_small_p24 of this repository — and unlike anything the pure-torch probes launch.
__global__ __launch_bounds__(512) void spin_scratch_kernel(unsigned* err, long long cycles, int lds_words, int stride) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < lds_words; i += blockDim.x) lds[i] = 0x9e3779b9u * (unsigned)(i + 1) + blockIdx.x;
    __syncthreads();
    volatile float priv[48];
    for (int i = 0; i < 48; ++i) priv[i] = (float)(threadIdx.x * 48 + i);
    const long long t0 = clock64();
    int j = threadIdx.x % 48;
    float s = 0.f;
    while (clock64() - t0 < cycles) {
        s += priv[j];
        priv[j] = priv[j] + 0.f;
        j = (j + stride) % 48;
    }
    unsigned bad = 0;
    for (int i = 0; i < 48; ++i) bad += (priv[i] != (float)(threadIdx.x * 48 + i));
    __syncthreads();
    for (int i = threadIdx.x; i < lds_words; i += blockDim.x) bad += (lds[i] != 0x9e3779b9u * (unsigned)(i + 1) + blockIdx.x);
    if (bad) atomicAdd(err, bad);
    if (s == 12345.678f) atomicAdd(err + 1, 1u);
}

// Growing scratch needs, like a training step's first kernels (reverse sweep 20 B, adjoint forward sweep 60 B, thin-layer gradients 64 B, f32
// sweeps 232 B per lane): every kernel that needs more scratch than its queue has makes the runtime re-size the queue's scratch.
template <int NF>
__global__ __launch_bounds__(512) void scratch_n_kernel(unsigned* err, long long cycles, int stride) {
    volatile float priv[NF];
    for (int i = 0; i < NF; ++i) priv[i] = (float)(threadIdx.x * NF + i);
    const long long t0 = clock64();
    int j = threadIdx.x % NF;
    float s = 0.f;
    while (clock64() - t0 < cycles) { s += priv[j]; priv[j] = priv[j] + 0.f; j = (j + stride) % NF; }
    unsigned bad = 0;
    for (int i = 0; i < NF; ++i) bad += (priv[i] != (float)(threadIdx.x * NF + i));
    if (bad) atomicAdd(err, bad);
    if (s == 12345.678f) atomicAdd(err + 1, 1u);
}
extern "C" int scratch_n_launch(unsigned* err, int which, int blocks, long long cycles, void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    switch (which) {
        case 0: hipLaunchKernelGGL(scratch_n_kernel<6>, dim3(blocks), dim3(512), 0, st, err, cycles, 5); break;
        case 1: hipLaunchKernelGGL(scratch_n_kernel<16>, dim3(blocks), dim3(512), 0, st, err, cycles, 5); break;
        case 2: hipLaunchKernelGGL(scratch_n_kernel<24>, dim3(blocks), dim3(512), 0, st, err, cycles, 5); break;
        default: hipLaunchKernelGGL(scratch_n_kernel<60>, dim3(blocks), dim3(512), 0, st, err, cycles, 7); break;
    }
    return (int)hipGetLastError();
}

extern "C" int spin_scratch_launch(unsigned* err, int blocks, long long cycles, int lds_bytes, void* stream) {
    static int cur = -1;
    if (lds_bytes != cur) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spin_scratch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return (int)e;
        cur = lds_bytes;
    }
    hipLaunchKernelGGL(spin_scratch_kernel, dim3(blocks), dim3(512), lds_bytes, reinterpret_cast<hipStream_t>(stream), err, cycles, lds_bytes / 4, 7);
    return (int)hipGetLastError();
}

extern "C" int spin_lds_launch(unsigned* err, int blocks, long long cycles, int lds_bytes, void* stream) {
    static int cur = -1;
    if (lds_bytes != cur) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spin_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return (int)e;
        cur = lds_bytes;
    }
    hipLaunchKernelGGL(spin_lds_kernel, dim3(blocks), dim3(512), lds_bytes, reinterpret_cast<hipStream_t>(stream), err, cycles, lds_bytes / 4);
    return (int)hipGetLastError();
}
