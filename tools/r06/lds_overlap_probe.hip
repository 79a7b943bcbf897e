// lds_overlap_probe.hip -- do the LDS allocations of workgroups of DIFFERENT kernels that share a CU ever overlap?
// Every workgroup fills its static LDS block with a pattern derived from (kernel tag, workgroup id, round), holds it for a few
// microseconds, and checks it.  Kernels with the LDS footprints of the engine's kernels (70 660 B "wide" predictor cell, 46 592 B
// log-mel front-end, 100 356 B encoder cell, 71 684 B, 38 916 B) run on separate streams at the same time.
// build: hipcc --offload-arch=gfx950 -O2 -o lds_overlap_probe lds_overlap_probe.hip ; run: ./lds_overlap_probe [ms per kernel]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int BYTES, int THREADS>
__global__ __launch_bounds__(THREADS) void k_hold(unsigned tag, int rounds, int hold_ticks, unsigned long long* bad, unsigned* first_bad) {
    __shared__ unsigned buf[BYTES / 4];
    constexpr int N = BYTES / 4;
    const unsigned wg = blockIdx.x;
    unsigned long long my_bad = 0;
    for (int r = 0; r < rounds; ++r) {
        const unsigned seed = tag * 0x9e3779b9u + wg * 0x85ebca6bu + (unsigned)r * 0xc2b2ae35u;
        for (int i = threadIdx.x; i < N; i += THREADS) buf[i] = seed ^ (unsigned)i * 2654435761u;
        __syncthreads();
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)hold_ticks) { __builtin_amdgcn_s_sleep(8); }
        __syncthreads();
        for (int i = threadIdx.x; i < N; i += THREADS) {
            const unsigned v = buf[i];
            if (v != (seed ^ (unsigned)i * 2654435761u)) {
                ++my_bad;
                if (atomicAdd(&first_bad[0], 1u) < 8u) { const unsigned k = atomicAdd(&first_bad[1], 1u); if (k < 8) { first_bad[2 + 4 * k] = tag; first_bad[3 + 4 * k] = wg; first_bad[4 + 4 * k] = (unsigned)i; first_bad[5 + 4 * k] = v; } }
            }
        }
        __syncthreads();
    }
    if (my_bad) atomicAdd(bad, my_bad);
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    unsigned long long* bad; unsigned* fb;
    CHECK(hipMalloc(&bad, 8 * sizeof(unsigned long long))); CHECK(hipMemset(bad, 0, 8 * sizeof(unsigned long long)));
    CHECK(hipMalloc(&fb, 64 * sizeof(unsigned))); CHECK(hipMemset(fb, 0, 64 * sizeof(unsigned)));
    hipStream_t s[5];
    for (auto& x : s) CHECK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    const int hold = 300;         // 100 MHz wall clock: 3 us
    struct Combo { const char* name; bool on[5]; };
    const Combo combos[] = {
        {"wide(70660) + fe_mel(46592)", {1, 1, 0, 0, 0}},
        {"wide + fe_mel + enc(100356)", {1, 1, 1, 0, 0}},
        {"wide8(71684) + fe_mel", {0, 1, 0, 1, 0}},
        {"narrow(38916) + fe_mel + enc", {0, 1, 1, 0, 1}},
        {"all five", {1, 1, 1, 1, 1}},
    };
    for (const Combo& cb : combos) {
        CHECK(hipMemset(bad, 0, 8 * sizeof(unsigned long long))); CHECK(hipMemset(fb, 0, 64 * sizeof(unsigned)));
        for (int rep = 0; rep < reps; ++rep) {
            if (cb.on[0]) hipLaunchKernelGGL((k_hold<70660 / 4 * 4, 256>), dim3(1536), dim3(256), 0, s[0], 1u, 4, hold, bad + 0, fb);
            if (cb.on[1]) hipLaunchKernelGGL((k_hold<46592, 320>), dim3(512), dim3(320), 0, s[1], 2u, 4, hold / 2, bad + 1, fb);
            if (cb.on[2]) hipLaunchKernelGGL((k_hold<100356, 512>), dim3(256), dim3(512), 0, s[2], 3u, 4, hold, bad + 2, fb);
            if (cb.on[3]) hipLaunchKernelGGL((k_hold<71684, 512>), dim3(1024), dim3(512), 0, s[3], 4u, 4, hold, bad + 3, fb);
            if (cb.on[4]) hipLaunchKernelGGL((k_hold<38916, 512>), dim3(1024), dim3(512), 0, s[4], 5u, 4, hold, bad + 4, fb);
        }
        CHECK(hipDeviceSynchronize());
        unsigned long long h[8]; unsigned hf[64];
        CHECK(hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hf, fb, sizeof(hf), hipMemcpyDeviceToHost));
        printf("%-34s corrupted words per kernel: wide %llu  fe_mel %llu  enc %llu  wide8 %llu  narrow %llu\n", cb.name, h[0], h[1], h[2], h[3], h[4]);
        for (unsigned k = 0; k < hf[1] && k < 8; ++k) printf("    e.g. kernel %u workgroup %u word %u holds %08x\n", hf[2 + 4 * k], hf[3 + 4 * k], hf[4 + 4 * k], hf[5 + 4 * k]);
    }
    return 0;
}
