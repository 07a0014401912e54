// Dev tool: cost of a dependent chain of short kernels, stream launches vs one captured hipGraph.
//   hipcc --offload-arch=gfx950 -O3 tools/launch_gap_probe.hip -o tools/launch_gap_probe.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void tiny(float* p, int spin) {
    float v = p[threadIdx.x];
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
    p[threadIdx.x] = v;
}
int main() {
    float* d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int N = 2720;
    for (int grid : {1, 256, 1024}) for (int spin : {0, 2000}) {
        for (int rep = 0; rep < 2; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, d, spin);
            auto t1 = std::chrono::steady_clock::now();
            CK(hipStreamSynchronize(s));
            auto t2 = std::chrono::steady_clock::now();
            if (rep) printf("grid %4d spin %4d stream: enqueue %.2f us/launch, total %.2f us/launch\n", grid, spin,
                            std::chrono::duration<double, std::micro>(t1 - t0).count() / N, std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, d, spin);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            CK(hipGraphLaunch(ge, s));
            auto t1 = std::chrono::steady_clock::now();
            CK(hipStreamSynchronize(s));
            auto t2 = std::chrono::steady_clock::now();
            if (rep) printf("grid %4d spin %4d graph : enqueue %.2f us/node,   total %.2f us/node\n", grid, spin,
                            std::chrono::duration<double, std::micro>(t1 - t0).count() / N, std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
