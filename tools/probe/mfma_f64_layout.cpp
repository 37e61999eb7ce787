// Probe: fragment layout of v_mfma_f64_16x16x4_f64 on gfx950 (tools only).
// hipcc --offload-arch=gfx950 -O2 mfma_f64_layout.cpp -o mfma_probe && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* D) {   // A[16][4], B[4][16], D raw [64][4]
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + (l / 16)];
    const double b = B[(l / 16) * 16 + (l % 16)];
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}
int main() {
    double hA[64], hB[64], hD[256], ref[16][16];
    for (int i = 0; i < 64; ++i) { hA[i] = sin(i * 0.37) + 0.1 * i; hB[i] = cos(i * 0.91) - 0.05 * i; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j]; ref[i][j] = s; }
    double *dA, *dB, *dD;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
    hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost);
    // hypothesis 1: D[i = 4*(l/16)+r][j = l%16]; hypothesis 2: D[i = (l/16) + 4*r][j = l%16]
    double e1 = 0, e2 = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        e1 = fmax(e1, fabs(hD[l * 4 + r] - ref[4 * (l / 16) + r][l % 16]));
        e2 = fmax(e2, fabs(hD[l * 4 + r] - ref[(l / 16) + 4 * r][l % 16]));
    }
    printf("hyp1 (i=4*(l/16)+r) maxerr %.3e ; hyp2 (i=l/16+4r) maxerr %.3e\n", e1, e2);
    return 0;
}
