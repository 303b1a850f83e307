// Accuracy of the hardware v_sin_f32 / v_cos_f32 (input in TURNS) on a reduced argument, against double sin/cos of the
// exact fp32 input x (radians), for the argument magnitudes the SIREN sees (|w0 z| up to ~60).   hipcc -O3 ... && ./a.out
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>
__global__ void k(const float* x, float* s, float* c, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    // x / (2 pi) in two pieces: 1/(2 pi) = hi + lo
    const float hi = 0.15915494309189535f, lo = (float)(0.15915494309189533576888 - (double)0.15915494309189535f);
    const float kk = rintf(v * hi);
    float r = fmaf(v, hi, -kk);
    r = fmaf(v, lo, r);
    s[i] = __builtin_amdgcn_sinf(r);
    c[i] = __builtin_amdgcn_cosf(r);
}
int main() {
    const int n = 1 << 24;
    std::vector<float> x(n), s(n), c(n);
    unsigned long long st = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; double u = (st >> 11) * (1.0 / 9007199254740992.0);
        double range = (i & 3) == 0 ? 1.0 : ((i & 3) == 1 ? 8.0 : 60.0); x[i] = (float)((u * 2 - 1) * range); }
    float *dx, *ds, *dc;
    hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, ds, dc, n);
    hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
    double es[3] = {0, 0, 0}, ec[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const int b = i & 3; const int g = b == 0 ? 0 : (b == 1 ? 1 : 2);
        es[g] = fmax(es[g], fabs((double)s[i] - sin((double)x[i])));
        ec[g] = fmax(ec[g], fabs((double)c[i] - cos((double)x[i])));
    }
    printf("max abs err  |x|<1: sin %.3e cos %.3e   |x|<8: sin %.3e cos %.3e   |x|<60: sin %.3e cos %.3e\n", es[0], ec[0], es[1], ec[1], es[2], ec[2]);
    return 0;
}
