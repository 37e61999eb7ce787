// Probe (round 4): the fixed cost of ONE short call - 40 dependent launches between two host synchronisations, the shape of
// bench.py's blocks of 20 iterations - by how the call is enqueued and how its completion reaches the host.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/call_floor.hip -o build_ab/call_floor && build_ab/call_floor
// Every kernel spins SPIN_US on the 100 MHz wall clock in 512 workgroups x 512 threads (the footprint of the stepping kernels);
// floor = median wall time of a call - 40 x (time per launch of a 4000-launch chain).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#include <vector>
#include <algorithm>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Big { void* p[40]; double d[8]; int i[16]; };   // ~450 bytes of kernel arguments, like StretchArgs

__global__ __launch_bounds__(512) void k_spin(const Big a, long long ticks) {
    extern __shared__ char smem[];
    asm volatile("v_mov_b32 v100, 0" ::: "v100");
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    if (a.i[0] == 12345) smem[threadIdx.x] = 1;
}
// the call's last launch: one thread stores the call's sequence number into host memory (system scope, write-through)
__global__ void k_done(unsigned* flag, unsigned seq) { __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
// the same store from the tail of a spinning kernel's workgroup 0 - NOT a completion of the grid, only of that workgroup
// (timing probe for "what would a flag from the last launch itself save")
__global__ __launch_bounds__(512) void k_spin_flag(const Big a, long long ticks, unsigned* flag, unsigned seq) {
    extern __shared__ char smem[];
    asm volatile("v_mov_b32 v100, 0" ::: "v100");
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    if (a.i[0] == 12345) smem[threadIdx.x] = 1;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

using clk = std::chrono::high_resolution_clock;
static double us_since(clk::time_point t0) { return std::chrono::duration<double, std::micro>(clk::now() - t0).count(); }
static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char** argv) {
    const int NL = argc > 1 ? atoi(argv[1]) : 40;
    const long long spin_us = argc > 2 ? atoll(argv[2]) : 8;
    const long long ticks = spin_us * 100;
    const int WGS = 512, NT = 512, LDS = 40000, REPS = 200;
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t s;
    CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));
    Big a{};
    unsigned* flag;
    CK(hipHostMalloc((void**)&flag, 64, hipHostMallocMapped | hipHostMallocCoherent));
    *flag = 0;
    unsigned* dflag;
    CK(hipHostGetDevicePointer((void**)&dflag, flag, 0));
    volatile unsigned* vflag = flag;
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    auto launch = [&] { hipLaunchKernelGGL(k_spin, dim3(WGS), dim3(NT), LDS, s, a, ticks); };

    // per-launch time inside a long chain
    for (int i = 0; i < 200; ++i) launch();
    CK(hipStreamSynchronize(s));
    auto t0 = clk::now();
    for (int i = 0; i < 4000; ++i) launch();
    CK(hipStreamSynchronize(s));
    const double per = us_since(t0) / 4000;
    printf("chain: %.2f us per launch (spin %lld us) -> %d launches = %.1f us\n", per, spin_us, NL, per * NL);

    hipGraph_t g; hipGraphExec_t ge, gef;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < NL; ++i) launch();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    // graph with a trailing flag store (sequence number read from a device word the host bumps? no: a fixed node, seq = 1; the host
    // resets the flag to 0 before every replay)
    hipGraph_t g2;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < NL; ++i) launch();
    hipLaunchKernelGGL(k_done, dim3(1), dim3(1), 0, s, dflag, 1u);
    CK(hipStreamEndCapture(s, &g2));
    CK(hipGraphInstantiate(&gef, g2, nullptr, nullptr, 0));

    unsigned seq = 1;
    struct Variant { const char* name; int id; };
    const Variant vs[] = {
        {"eager, hipStreamSynchronize", 0},
        {"eager, hipStreamSynchronize + hipDeviceSynchronize", 1},
        {"eager, k_done -> host flag, host spins", 2},
        {"eager, k_done -> host flag, host spins, then hipDeviceSynchronize", 3},
        {"eager, hipStreamQuery spin", 4},
        {"eager, hipEventRecord + hipEventQuery spin", 5},
        {"eager, last launch's workgroup tail -> host flag (not a grid completion)", 6},
        {"graph replay, hipStreamSynchronize", 7},
        {"graph replay + flag node, host spins", 8},
        {"graph replay + flag node, host spins, then hipDeviceSynchronize", 9},
        {"1 launch, hipStreamSynchronize", 10},
        {"1 launch + k_done, host spins", 11},
        {"eager, hipDeviceSynchronize only", 12},
    };
    for (const Variant& v : vs) {
        std::vector<double> ts, tenq;
        for (int r = 0; r < REPS + 20; ++r) {
            CK(hipDeviceSynchronize());
            *flag = 0;
            const unsigned my = (v.id == 8 || v.id == 9) ? 1u : ++seq;
            auto t0 = clk::now();
            int nl = NL;
            switch (v.id) {
            case 7: CK(hipGraphLaunch(ge, s)); break;
            case 8: case 9: CK(hipGraphLaunch(gef, s)); break;
            case 10: launch(); nl = 1; break;
            case 11: launch(); hipLaunchKernelGGL(k_done, dim3(1), dim3(1), 0, s, dflag, my); nl = 1; break;
            case 6:
                for (int i = 0; i < NL - 1; ++i) launch();
                hipLaunchKernelGGL(k_spin_flag, dim3(WGS), dim3(NT), LDS, s, a, ticks, dflag, my);
                break;
            default:
                for (int i = 0; i < NL; ++i) launch();
                if (v.id == 2 || v.id == 3) hipLaunchKernelGGL(k_done, dim3(1), dim3(1), 0, s, dflag, my);
                if (v.id == 5) CK(hipEventRecord(ev, s));
            }
            const double enq = us_since(t0);
            switch (v.id) {
            case 0: case 7: case 10: CK(hipStreamSynchronize(s)); break;
            case 1: CK(hipStreamSynchronize(s)); CK(hipDeviceSynchronize()); break;
            case 2: case 6: case 8: case 11: while (*vflag != my) {} break;
            case 3: case 9: while (*vflag != my) {} CK(hipDeviceSynchronize()); break;
            case 4: while (hipStreamQuery(s) == hipErrorNotReady) {} break;
            case 5: while (hipEventQuery(ev) == hipErrorNotReady) {} break;
            case 12: CK(hipDeviceSynchronize()); break;
            }
            const double t = us_since(t0);
            if (r >= 20) { ts.push_back(t - per * nl); tenq.push_back(enq); }
        }
        printf("  %-74s floor %6.1f us (min %6.1f)   enqueue %6.1f us\n", v.name, median(ts), *std::min_element(ts.begin(), ts.end()), median(tenq));
    }
    return 0;
}
