// Developer tool: what does a dependent kernel boundary cost on this GPU, and what would a grid-wide barrier inside ONE kernel cost
// instead?  The small-batch AR loop is a chain of ~34 short dependent launches per 25-frame step: this measures the floor
// of that chain (launch + drain per kernel) against the same chain as phases of one co-resident kernel separated by a device-scope
// barrier.  Every phase moves one float per thread from a DIFFERENT workgroup's slot of the previous phase (other XCDs included), so
// a barrier that does not make writes visible across XCDs fails the check at the end.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/launch_bench.hip -o tools/launch_bench.bin && timeout 60 tools/launch_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

constexpr int kThreads = 256;

__device__ __forceinline__ void phase_body(const float* src, float* dst, int wg, int nwg, int tid) {
    const int from = (wg + 37) % nwg;  // another workgroup (another XCD for most)
    dst[wg * kThreads + tid] = src[from * kThreads + tid] + 1.0f;
}

__global__ __launch_bounds__(kThreads) void step_kernel(const float* src, float* dst) { phase_body(src, dst, blockIdx.x, gridDim.x, threadIdx.x); }

// sense-free counting barrier: phase p waits until the counter reaches (p + 1) * nwg.  Release: the workgroup's stores are made
// visible at agent scope before its arrival; acquire: the caches are invalidated after the wait.
__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

__global__ __launch_bounds__(kThreads) void mega_kernel(float* a, float* b, unsigned* counter, int phases) {
    const int wg = blockIdx.x, nwg = gridDim.x, tid = threadIdx.x;
    for (int p = 0; p < phases; ++p) {
        const float* src = (p & 1) ? b : a;
        float* dst = (p & 1) ? a : b;
        // (plain loads of data another XCD wrote: must not come from a stale L2 line -> the acquire fence above)
        phase_body(src, dst, wg, nwg, tid);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // every thread: its own stores written back before the workgroup arrives
        grid_barrier(counter, (unsigned)(p + 1) * nwg);
    }
}

int main(int argc, char** argv) {
    const int phases = argc > 1 ? atoi(argv[1]) : 400;
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int nwg : {8, 32, 64, 256}) {
        float *a, *b;
        unsigned* counter;
        const size_t n = (size_t)nwg * kThreads;
        CHECK(hipMalloc(&a, n * 4));
        CHECK(hipMalloc(&b, n * 4));
        CHECK(hipMalloc(&counter, 256));
        std::vector<float> h(n);
        auto run = [&](bool mega) {
            CHECK(hipMemsetAsync(a, 0, n * 4, s));
            CHECK(hipMemsetAsync(b, 0, n * 4, s));
            CHECK(hipMemsetAsync(counter, 0, 256, s));
            CHECK(hipEventRecord(e0, s));
            if (mega) {
                hipLaunchKernelGGL(mega_kernel, dim3(nwg), dim3(kThreads), 0, s, a, b, counter, phases);
            } else {
                for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(step_kernel, dim3(nwg), dim3(kThreads), 0, s, (p & 1) ? b : a, (p & 1) ? a : b);
            }
            CHECK(hipGetLastError());
            CHECK(hipEventRecord(e1, s));
            CHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(h.data(), (phases & 1) ? b : a, n * 4, hipMemcpyDeviceToHost));
            bool ok = true;
            for (size_t i = 0; i < n; ++i) ok = ok && h[i] == (float)phases;
            return std::make_pair(ms, ok);
        };
        run(false);
        run(true);
        const auto k = run(false);
        const auto m = run(true);
        printf("%3d workgroups x %d phases: dependent launches %.2f us each (%s), grid barrier in one kernel %.2f us each (%s)\n", nwg, phases,
               k.first * 1e3 / phases, k.second ? "ok" : "WRONG", m.first * 1e3 / phases, m.second ? "ok" : "WRONG");
        CHECK(hipFree(a));
        CHECK(hipFree(b));
        CHECK(hipFree(counter));
    }
    return 0;
}
