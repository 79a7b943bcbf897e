// vgpr_pack_probe.hip -- is a small wave safe beside two large waves on a SIMD whose register file is (nearly) fully allocated?
// Aggressor: 256-thread workgroups (one wave per SIMD), 66 560 B of LDS (two workgroups per CU), a bf16 MFMA loop on accumulators
// and a register allocation fixed by an empty asm clobber (ALLOC registers).  Victim: 64-thread workgroups that keep NV registers
// with known contents through a loop of LDS transposes and VALU work and check them; VALLOC registers.
// Both run at the same time on two streams; the victim reports every mismatch.
// build: hipcc --offload-arch=gfx950 -O2 -o vgpr_pack_probe vgpr_pack_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int TAG>
__global__ __launch_bounds__(256) void k_aggr(const f32x4* __restrict__ src, float* out, int iters) {
    __shared__ float lds[66560 / 4];
    if constexpr (TAG == 184) asm volatile("" ::: "v100", "a83");
    if constexpr (TAG == 232) asm volatile("" ::: "v131", "a99");
    if constexpr (TAG == 240) asm volatile("" ::: "v131", "a107");
    if constexpr (TAG == 248) asm volatile("" ::: "v131", "a115");
    if constexpr (TAG == 256) asm volatile("" ::: "v131", "a123");
    if constexpr (TAG == 1232) asm volatile("" ::: "v231");            // arch VGPRs only
    if constexpr (TAG == 1240) asm volatile("" ::: "v239");
    if constexpr (TAG == 1248) asm volatile("" ::: "v247");
    if constexpr (TAG == 1168) asm volatile("" ::: "v167");
    if constexpr (TAG == 1160) asm volatile("" ::: "v159");
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 a = src[threadIdx.x], b = src[256 + threadIdx.x];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
        a = src[(it * 64 + threadIdx.x) & 1023];
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    lds[threadIdx.x] = s;
    __syncthreads();
    if (lds[(threadIdx.x + 1) & 255] == 12345.f) out[0] = s;
}

template <int VTAG>
__global__ __launch_bounds__(64) void k_victim(unsigned long long* bad, int rounds) {
    // 40+ live registers filled by wide LDS reads (ds_read_b128 / ds_read_b64), as in the FFT passes of the log-mel kernel
    __shared__ float4 z4[64 * 10];
    __shared__ float2 z2[64 * 4];
    const int j = threadIdx.x;
    unsigned long long nb = 0;
    for (int r = 0; r < rounds; ++r) {
        const float base = (float)(r * 8192);
#pragma unroll
        for (int k = 0; k < 10; ++k) { const float v = base + (float)(4 * (k * 64 + j)); z4[k * 64 + j] = float4{v, v + 1.f, v + 2.f, v + 3.f}; }
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float v = base + 4096.f + (float)(2 * (k * 64 + j)); z2[k * 64 + j] = float2{v, v + 1.f}; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float4 q[10]; float2 p[4];
        const int jj = (j * 7 + r) & 63;          // another lane's slots
#pragma unroll
        for (int k = 0; k < 10; ++k) q[k] = z4[k * 64 + jj];
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] = z2[k * 64 + jj];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const float v = base + (float)(4 * (k * 64 + jj));
            if (q[k].x != v || q[k].y != v + 1.f || q[k].z != v + 2.f || q[k].w != v + 3.f) ++nb;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float v = base + 4096.f + (float)(2 * (k * 64 + jj));
            if (p[k].x != v || p[k].y != v + 1.f) ++nb;
        }
    }
    if (nb) atomicAdd(bad, nb);
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int TAG, int VTAG>
int run(const f32x4* src, float* out, unsigned long long* bad, hipStream_t sa, hipStream_t sv, int reps) {
    CHECK(hipMemset(bad, 0, 8));
    for (int rep = 0; rep < reps; ++rep) {
        hipLaunchKernelGGL((k_aggr<TAG>), dim3(512), dim3(256), 0, sa, src, out, 20000);
        for (int q = 0; q < 4; ++q) hipLaunchKernelGGL((k_victim<VTAG>), dim3(2048), dim3(64), 0, sv, bad, 400);
    }
    CHECK(hipDeviceSynchronize());
    unsigned long long h = 0;
    CHECK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
    printf("aggressor alloc tag %4d, victim alloc %2d: %llu mismatches\n", TAG, VTAG, h);
    return 0;
}
int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    f32x4* src; float* out; unsigned long long* bad;
    CHECK(hipMalloc(&src, 1024 * 16 * 2)); CHECK(hipMemset(src, 0x3c, 1024 * 16 * 2)); CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&bad, 8));
    hipStream_t sa, sv;
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    run<184, 16>(src, out, bad, sa, sv, reps); run<184, 48>(src, out, bad, sa, sv, reps);
    run<232, 16>(src, out, bad, sa, sv, reps); run<232, 32>(src, out, bad, sa, sv, reps); run<232, 48>(src, out, bad, sa, sv, reps);
    run<240, 16>(src, out, bad, sa, sv, reps); run<240, 32>(src, out, bad, sa, sv, reps); run<240, 48>(src, out, bad, sa, sv, reps);
    run<248, 16>(src, out, bad, sa, sv, reps); run<248, 32>(src, out, bad, sa, sv, reps);
    run<256, 16>(src, out, bad, sa, sv, reps); run<256, 48>(src, out, bad, sa, sv, reps);
    run<1232, 48>(src, out, bad, sa, sv, reps); run<1240, 32>(src, out, bad, sa, sv, reps); run<1248, 16>(src, out, bad, sa, sv, reps);
    run<1168, 16>(src, out, bad, sa, sv, reps); run<1160, 32>(src, out, bad, sa, sv, reps);
    return 0;
}
