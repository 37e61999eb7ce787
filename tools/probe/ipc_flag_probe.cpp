// Probe: one-sided put + flag between two PROCESSES through HIP IPC, on whatever GPUs are visible.
//   ipc_flag_probe A <dir> [dev]   owner: allocates buffer+flag, exports handles, waits on the flag with a kernel
//   ipc_flag_probe B <dir> [dev]   peer : opens the handles, kernel-stores into the buffer, then raises the flag
// Build: hipcc --offload-arch=gfx950 -O2 ipc_flag_probe.cpp -o ipc_flag_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unistd.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(2); } } while (0)

__global__ void k_put(double* dst, int n, double v) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = v + i;
}
__global__ void k_flag(unsigned* flag, unsigned v) {
    __atomic_store_n(flag, v, __ATOMIC_RELEASE);      // system scope by default for __atomic builtins
}
__global__ void k_wait(unsigned* flag, unsigned target, unsigned* err, long long budget) {
    const long long t0 = wall_clock64();
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) < target) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > budget) { *err = 1; return; }
    }
}
__global__ void k_sum(const double* p, int n, double* out) {
    double s = 0; for (int i = 0; i < n; ++i) s += p[i]; *out = s;
}

int main(int argc, char** argv) {
    const char role = argv[1][0];
    const std::string dir = argv[2];
    const int dev = argc > 3 ? atoi(argv[3]) : 0;
    const int unc = argc > 4 ? atoi(argv[4]) : 1;
    CK(hipSetDevice(dev));
    const int n = 1 << 16;
    hipStream_t s; CK(hipStreamCreate(&s));
    if (role == 'A') {
        double* buf; unsigned* flag; unsigned* err; double* out;
        if (unc) {
            CK(hipExtMallocWithFlags((void**)&buf, n * 8, hipDeviceMallocUncached));
            CK(hipExtMallocWithFlags((void**)&flag, 256, hipDeviceMallocUncached));
        } else {
            CK(hipMalloc((void**)&buf, n * 8)); CK(hipMalloc((void**)&flag, 256));
        }
        CK(hipMalloc((void**)&err, 4)); CK(hipMalloc((void**)&out, 8));
        CK(hipMemset(buf, 0, n * 8)); CK(hipMemset(flag, 0, 256)); CK(hipMemset(err, 0, 4));
        CK(hipDeviceSynchronize());
        hipIpcMemHandle_t h[2];
        CK(hipIpcGetMemHandle(&h[0], buf)); CK(hipIpcGetMemHandle(&h[1], flag));
        FILE* f = fopen((dir + "/handles.tmp").c_str(), "wb"); fwrite(h, sizeof h, 1, f); fclose(f);
        rename((dir + "/handles.tmp").c_str(), (dir + "/handles.bin").c_str());
        int wclk = 0; CK(hipDeviceGetAttribute(&wclk, hipDeviceAttributeWallClockRate, dev));   // kHz
        for (unsigned round = 1; round <= 3; ++round) {
            hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, s, flag, round, err, (long long)wclk * 20000LL);  // 20 s
            hipLaunchKernelGGL(k_sum, dim3(1), dim3(1), 0, s, buf, 1024, out);
            CK(hipStreamSynchronize(s));
            unsigned e; double o; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&o, out, 8, hipMemcpyDeviceToHost));
            const double expect = 1024.0 * round * 1000.0 + 1023.0 * 1024.0 / 2.0;
            printf("A round %u: timeout=%u sum=%.1f expect=%.1f %s\n", round, e, o, expect, (e == 0 && o == expect) ? "OK" : "BAD");
            fflush(stdout);
        }
        // hipStreamWaitValue32 variant
        int can = 0; (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev);
        printf("A canUseStreamWaitValue=%d\n", can); fflush(stdout);
        if (can) {
            hipError_t e = hipStreamWaitValue32(s, flag, 4, hipStreamWaitValueGte, 0xffffffffu);
            printf("A hipStreamWaitValue32 enqueue: %s\n", hipGetErrorString(e)); fflush(stdout);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_sum, dim3(1), dim3(1), 0, s, buf, 1024, out);
                CK(hipStreamSynchronize(s));
                double o; CK(hipMemcpy(&o, out, 8, hipMemcpyDeviceToHost));
                printf("A waitvalue round: sum=%.1f expect=%.1f\n", o, 1024.0 * 4000.0 + 1023.0 * 512.0);
            }
        }
        FILE* g = fopen((dir + "/done").c_str(), "w"); fclose(g);
    } else {
        std::string p = dir + "/handles.bin";
        for (int i = 0; i < 600 && access(p.c_str(), R_OK) != 0; ++i) usleep(100000);
        hipIpcMemHandle_t h[2];
        FILE* f = fopen(p.c_str(), "rb"); if (!f) { printf("B: no handles\n"); return 3; }
        fread(h, sizeof h, 1, f); fclose(f);
        double* buf; unsigned* flag;
        CK(hipIpcOpenMemHandle((void**)&buf, h[0], hipIpcMemLazyEnablePeerAccess));
        CK(hipIpcOpenMemHandle((void**)&flag, h[1], hipIpcMemLazyEnablePeerAccess));
        for (unsigned round = 1; round <= 4; ++round) {
            usleep(300000);
            hipLaunchKernelGGL(k_put, dim3(n / 256), dim3(256), 0, s, buf, n, round * 1000.0);
            hipLaunchKernelGGL(k_flag, dim3(1), dim3(1), 0, s, flag, round);
            CK(hipStreamSynchronize(s));
            printf("B round %u put done\n", round); fflush(stdout);
        }
        std::string d = dir + "/done";
        for (int i = 0; i < 300 && access(d.c_str(), R_OK) != 0; ++i) usleep(100000);
        CK(hipIpcCloseMemHandle(buf)); CK(hipIpcCloseMemHandle(flag));
    }
    printf("%c exit\n", role);
    return 0;
}
