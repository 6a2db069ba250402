// stream_pairs.hip -- which pairs of a process's HIP streams run side by side on gfx950?  
//
// The HIP runtime maps streams onto a pool of hardware queues (GPU_MAX_HW_QUEUES; a new stream takes the least-shared
// queue) and the queues onto the chip's dispatch pipes.  A receiver bank's two chains are only as parallel as their two
// streams are, and the same bank measured 0.98 ms per step or 1.25 ... 1.38 by what the process had created before.
// This program creates N streams and, for every pair (i, j), launches kernel A on stream i (twice the chip's resident
// workgroups, each sleeping ~16 us), kernel B on stream j right behind it, and prints A's time in units of A alone:
//   ~1.0   one after the other (the two streams share a hardware queue)
//   ~1.45  B's workgroups come in only as A's first round retires (A's whole grid is placed first)
//   ~1.85  B's workgroups are placed beside A's from the start
//   ~3.0   B is served first although launched second (seen for streams four apart in creation order)
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/stream_pairs.hip -o tools/micro/stream_pairs ; run: stream_pairs [N = 10] [pre = 0]
// (pre: streams created and left idle BEFORE the N measured ones, the way a host process has its own)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(64) void probe_kernel(long long ticks)
{
    const long long t0 = wall_clock64();                 // 100 MHz
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 10, pre = argc > 2 ? atoi(argv[2]) : 0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    std::vector<hipStream_t> idle(pre), s(N);
    for (auto &x : idle) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    for (auto &x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned) prop.multiProcessorCount * 32 * 2;
    auto timed = [&](int i, int j, float *ms) -> int {
        CK(hipEventRecord(e0, s[i]));
        hipLaunchKernelGGL(probe_kernel, dim3(grid), dim3(64), 0, s[i], 1600LL);
        CK(hipEventRecord(e1, s[i]));
        if (j >= 0) hipLaunchKernelGGL(probe_kernel, dim3(grid), dim3(64), 0, s[j], 1600LL);
        CK(hipStreamSynchronize(s[i]));
        if (j >= 0) CK(hipStreamSynchronize(s[j]));
        CK(hipEventElapsedTime(ms, e0, e1));
        return 0;
    };
    float t1 = 0, t;
    for (int k = 0; k < 3; k++) { if (timed(0, -1, &t)) return 1; if (k == 0 || t < t1) t1 = t; }
    printf("%s, %d CUs, GPU_MAX_HW_QUEUES=%s: %d streams (after %d idle ones), one kernel alone %.1f us\n      ", prop.gcnArchName,
           prop.multiProcessorCount, getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(default)", N, pre, t1 * 1e3);
    for (int j = 0; j < N; j++) printf("  B=%-2d", j);
    printf("\n");
    for (int i = 0; i < N; i++) {
        printf("A=%-2d  ", i);
        for (int j = 0; j < N; j++) {
            if (i == j) { printf("   -  "); continue; }
            if (timed(i, j, &t)) return 1;
            printf(" %5.2f", t / t1);
        }
        printf("\n");
    }
    return 0;
}
