// host_call_cost.hip -- host time of the runtime calls a host push makes (this ROCm, this box): hipMemcpyAsync of one chunk batch from
// pinned memory on a copy stream, hipEventRecord, hipStreamWaitEvent, a small kernel launch.  build: hipcc --offload-arch=gfx950 -O2 -o host_call_cost host_call_cost.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k_small(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t bytes = 64 * 1280 * 4;
    float *h = nullptr, *d = nullptr, *pg = (float*)malloc(bytes);
    hipHostMalloc((void**)&h, bytes); hipMalloc((void**)&d, bytes * 64); memset(pg, 1, bytes);
    hipStream_t sc, sm; hipStreamCreateWithFlags(&sc, hipStreamNonBlocking); hipStreamCreateWithFlags(&sm, hipStreamNonBlocking);
    std::vector<hipEvent_t> ev(64); for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    double t_mc = 0, t_cp = 0, t_er = 0, t_sw = 0, t_kl = 0, t_er2 = 0; const int N = 2000;
    for (int i = 0; i < N + 100; ++i) {
        const bool m = i >= 100;
        double a = now(); memcpy(h, pg, bytes); double b = now();
        hipMemcpyAsync(d + (size_t)(i % 64) * (bytes / 4), h, bytes, hipMemcpyHostToDevice, sc); double c = now();
        hipEventRecord(ev[i % 64], sc); double e = now();
        hipStreamWaitEvent(sm, ev[i % 64], 0); double f = now();
        hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, sm, d); double g = now();
        hipEventRecord(ev[(i + 32) % 64], sm); double hh = now();
        if (m) { t_mc += b - a; t_cp += c - b; t_er += e - c; t_sw += f - e; t_kl += g - f; t_er2 += hh - g; }
        if ((i & 15) == 15) hipDeviceSynchronize();
    }
    printf("per call, host microseconds (mean of %d): memcpy 328 KB pageable->pinned %.1f, hipMemcpyAsync H2D %.1f, hipEventRecord (copy stream) %.1f, "
           "hipStreamWaitEvent %.1f, kernel launch %.1f, hipEventRecord (main stream) %.1f\n", N, t_mc / N, t_cp / N, t_er / N, t_sw / N, t_kl / N, t_er2 / N);
    return 0;
}
