// fe_mfma_explore.hip -- which part of the log-mel kernel is hit in R?  The REAL neighbour (the 64 x 64 bf16 vocabulary GEMM with the
// shallow operand ring, see fe_mfma_repro.hip) beside SYNTHETIC, self-checking victims with the log-mel kernel's shape (5 waves, its
// LDS footprint): every value a victim reads from LDS is known analytically, every read is checked, and every read is done TWICE
// (second read behind a compiler barrier) so that a wrong first read can be told from wrong LDS contents.
//   victim 0: the FFT passes' access pattern -- 8 x float2 per lane written at stride 64 (ds_write2st64_b64), read back transposed
//             as 8 contiguous float2 (ds_read2_b64 / ds_read_b128)
//   victim 1: the same with 4-byte accesses only (ds_write_b32 / ds_read_b32)
//   victim 2: victim 0 in one-wave workgroups (64 threads)
// build (from the repository root): hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/r06/fe_mfma_explore tools/r06/fe_mfma_explore.hip
#include "../../libreasr_amd/csrc/lasr_kernels.hip.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace lasr;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <class T> static T* up(const std::vector<T>& v) { T* d = nullptr; if (hipMalloc((void**)&d, v.size() * sizeof(T)) != hipSuccess) return nullptr; (void)hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice); return d; }
static unsigned rng_state = 12345u;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffff) / 65536.0f - 0.5f; }

// counters: [0] checks, [1] first read wrong, [2] second read wrong, [3] first wrong AND second right, [4..] samples
template <int KIND, int NWV>
__global__ __launch_bounds__(64 * NWV) void k_victim(unsigned long long* cnt, int rounds, float salt) {
    __shared__ float2 sz[NWV][512 + 8];
    __shared__ float filler[NWV == 5 ? (46592 - 5 * 520 * 8) / 4 : 16];
    const int w = threadIdx.x >> 6, j = threadIdx.x & 63;
    if (threadIdx.x < 16) filler[threadIdx.x] = salt;
    float2* z = sz[w];
    unsigned long long n1 = 0, n2 = 0, n12 = 0, nc = 0;
    for (int r = 0; r < rounds; ++r) {
        const float base = salt + (float)((r * 31 + blockIdx.x * 7 + blockIdx.y) & 1023) * 1024.f;
        if constexpr (KIND == 1) {
            float* zf = (float*)z;
#pragma unroll
            for (int k0 = 0; k0 < 8; ++k0) { zf[2 * (k0 * 64 + j)] = base + (float)(k0 * 64 + j); zf[2 * (k0 * 64 + j) + 1] = -(base + (float)(k0 * 64 + j)); }
        } else {
#pragma unroll
            for (int k0 = 0; k0 < 8; ++k0) { const float v = base + (float)(k0 * 64 + j); z[k0 * 64 + j] = float2{v, -v}; }
        }
        wave_sync_lds();
        const int k0 = j >> 3, b = j & 7;
        float2 q1[8], q2[8];
        if constexpr (KIND == 1) {
            const float* zf = (const float*)z;
#pragma unroll
            for (int aa = 0; aa < 8; ++aa) { q1[aa].x = zf[2 * (k0 * 64 + 8 * aa + b)]; q1[aa].y = zf[2 * (k0 * 64 + 8 * aa + b) + 1]; }
        } else {
#pragma unroll
            for (int aa = 0; aa < 8; ++aa) q1[aa] = z[k0 * 64 + 8 * aa + b];
        }
        wave_sync_lds();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int aa = 0; aa < 8; ++aa) { const volatile double* zp = (const volatile double*)&z[k0 * 64 + 8 * aa + b]; const double d = *zp; q2[aa] = __builtin_bit_cast(float2, d); }
        wave_sync_lds();
#pragma unroll
        for (int aa = 0; aa < 8; ++aa) {
            const float want = base + (float)(k0 * 64 + 8 * aa + b);
            const bool b1 = q1[aa].x != want || q1[aa].y != -want, b2 = q2[aa].x != want || q2[aa].y != -want;
            n1 += b1; n2 += b2; n12 += (b1 && !b2); ++nc;
        }
    }
    if (n1 | n2) { atomicAdd(&cnt[1], n1); atomicAdd(&cnt[2], n2); atomicAdd(&cnt[3], n12); }
    if (threadIdx.x == 0) atomicAdd(&cnt[0], nc * 64 * NWV);
    if (filler[1] == 12345.f) cnt[7] = 1;
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 500, rounds = argc > 2 ? atoi(argv[2]) : 12, per = argc > 3 ? atoi(argv[3]) : 4;
    const int M = 128, K = 1536, V = 2048, ROWS = 1024, KC = K / 32;
    std::vector<unsigned short> a((size_t)KC * (ROWS / 16) * 64 * 8), w((size_t)(V / 64) * 4 * KC * 64 * 8);
    for (auto& x : a) x = (unsigned short)(0x3c00 + (int)(frand() * 512.f) + (frand() > 0 ? 0x8000 : 0));
    for (auto& x : w) x = (unsigned short)(0x3c00 + (int)(frand() * 512.f) + (frand() > 0 ? 0x8000 : 0));
    GemmArgs g{};
    g.A[0] = up(a); g.a_mt_total[0] = ROWS / 16; g.a_mt_off[0] = 0; g.KC[0] = KC; g.W[0] = up(w); g.M = ROWS; g.prio = 1;
    EpiLinearT<4>::Args e{};
    float* logits = nullptr; CHECK(hipMalloc((void**)&logits, sizeof(float) * ROWS * V));
    std::vector<float> bias(V, 0.25f);
    e.bias = up(bias); e.out = logits; e.ldo = V; e.n_rows = ROWS; e.t_idx = nullptr; e.T_row = nullptr; e.M = M; e.W = 8;
    hipStream_t sa, sb;
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    unsigned long long* cnt = nullptr; CHECK(hipMalloc((void**)&cnt, 64));
    for (int kind = 0; kind < 3; ++kind)
        for (int with = 1; with >= 0; --with) {
            CHECK(hipMemset(cnt, 0, 64));
            for (int i = 0; i < launches; ++i) {
                if (with) for (int q = 0; q < per; ++q) hipLaunchKernelGGL((k_gemm<OpsBF16, EpiLinearT<4>, 4, 4, false, 2>), dim3(V / 64, ROWS / 64), dim3(256), 0, sb, g, e);
                if (kind == 0) hipLaunchKernelGGL((k_victim<0, 5>), dim3(4, M), dim3(320), 0, sa, cnt, rounds, 1.0f);
                if (kind == 1) hipLaunchKernelGGL((k_victim<1, 5>), dim3(4, M), dim3(320), 0, sa, cnt, rounds, 1.0f);
                if (kind == 2) hipLaunchKernelGGL((k_victim<0, 1>), dim3(20, M), dim3(64), 0, sa, cnt, rounds, 1.0f);
            }
            CHECK(hipDeviceSynchronize());
            unsigned long long h[8]; CHECK(hipMemcpy(h, cnt, 64, hipMemcpyDeviceToHost));
            printf("victim %d %s the GEMM: %llu checked reads, first read wrong %llu, second read wrong %llu, first wrong but second right %llu\n",
                   kind, with ? "beside" : "without", h[0], h[1], h[2], h[3]);
        }
    return 0;
}
