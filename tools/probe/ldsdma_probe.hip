// Probe: per-CU fill rate of the encoder cell's operand stream (fragment-major A shared by the workgroups of an m-group,
// weight panel private to a workgroup) with (a) a register ring of global_load_dwordx4 and (b) an LDS-DMA ring
// (global_load_lds_dwordx4 + ds_read_b128), at several ring depths, with and without the cell's MFMA work.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ldsdma_probe ldsdma_probe.hip && ./ldsdma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int N> __device__ __forceinline__ void wait_vm() {
    // s_waitcnt vmcnt(N) only (expcnt / lgkmcnt left at their maxima); gfx9 encoding: vmcnt[3:0] | [15:14]
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

// MODE 0: register ring, MODE 1: LDS-DMA ring.  BF: bf16 MFMA (1 per fragment pair) instead of f32 (4).  MFMA: do the math.
template <int MODE, int D, int NW, int NCH, bool BF, bool MFMA>
__global__ __launch_bounds__(NW * 64) void k_probe(const f32x4* __restrict__ A, const f32x4* __restrict__ W, int mt_total, float* out) {
    extern __shared__ f32x4 ring[];      // MODE 1: [NW][D][4][64]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int jb = blockIdx.x, mg = blockIdx.y;
    constexpr int KC = NCH * NW;
    const f32x4* ap[2];
    const f32x4* wp[2];
    for (int mt = 0; mt < 2; ++mt) ap[mt] = A + (size_t)(mg * 2 + mt) * 64 + lane;
    for (int s = 0; s < 2; ++s) wp[s] = W + (size_t)(jb * 2 + s) * KC * 64 + lane;
    const size_t a_step = (size_t)mt_total * 64;
    f32x4 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mma = [&](const f32x4& a0, const f32x4& a1, const f32x4& b0, const f32x4& b1) {
        if constexpr (MFMA) {
            const f32x4 av[2] = {a0, a1}, bv[2] = {b0, b1};
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    if constexpr (BF) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[mt]), __builtin_bit_cast(bf16x8, bv[nt]), acc[mt][nt], 0, 0, 0);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][e], bv[nt][e], acc[mt][nt], 0, 0, 0);
                    }
                }
        } else {
            acc[0][0] += a0 * b0; acc[1][1] += a1 * b1;
        }
    };
    if constexpr (MODE == 0) {
        f32x4 f[D][4];
        auto load = [&](int slot, int i) {
            const int c = w + i * NW;
            f[slot][0] = ap[0][(size_t)c * a_step]; f[slot][1] = ap[1][(size_t)c * a_step];
            f[slot][2] = wp[0][(size_t)c * 64]; f[slot][3] = wp[1][(size_t)c * 64];
        };
#pragma unroll
        for (int d = 0; d < D - 1; ++d) if (d < NCH) load(d, d);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (i + D - 1 < NCH) load((i + D - 1) % D, i + D - 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(f[i % D][0], f[i % D][1], f[i % D][2], f[i % D][3]);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        f32x4* my = ring + (size_t)w * D * 4 * 64;
        auto issue = [&](int slot, int i) {
            const int c = w + i * NW;
            f32x4* dst = my + (size_t)slot * 4 * 64;
            __builtin_amdgcn_global_load_lds((const void*)(ap[0] + (size_t)c * a_step), (__attribute__((address_space(3))) void*)(dst + 0 * 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void*)(ap[1] + (size_t)c * a_step), (__attribute__((address_space(3))) void*)(dst + 1 * 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void*)(wp[0] + (size_t)c * 64), (__attribute__((address_space(3))) void*)(dst + 2 * 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void*)(wp[1] + (size_t)c * 64), (__attribute__((address_space(3))) void*)(dst + 3 * 64), 16, 0, 0);
        };
#pragma unroll
        for (int d = 0; d < D - 1; ++d) if (d < NCH) issue(d, d);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (i + D - 1 < NCH) { issue((i + D - 1) % D, i + D - 1); wait_vm<4 * (D - 1)>(); }
            else {
                // tail: loads of chunks i+1 .. NCH-1 may stay in flight
                constexpr int dummy = 0; (void)dummy;
                const int rem = NCH - 1 - i;            // compile-time after unrolling
                if (rem >= 7) wait_vm<28>(); else if (rem == 6) wait_vm<24>(); else if (rem == 5) wait_vm<20>(); else if (rem == 4) wait_vm<16>();
                else if (rem == 3) wait_vm<12>(); else if (rem == 2) wait_vm<8>(); else if (rem == 1) wait_vm<4>(); else wait_vm<0>();
            }
            __builtin_amdgcn_sched_barrier(0);
            const f32x4* src = my + (size_t)(i % D) * 4 * 64 + lane;
            const f32x4 a0 = src[0], a1 = src[64], b0 = src[128], b1 = src[192];
            mma(a0, a1, b0, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 123.456f) out[0] = s;      // keep the work alive
}

template <int MODE, int D, int NW, int NCH, bool BF, bool MFMA>
void run(const char* tag, const f32x4* A, const f32x4* W, float* out, hipStream_t st) {
    const dim3 grid(128, 2), block(NW * 64);
    const size_t lds = MODE == 1 ? (size_t)NW * D * 4 * 64 * 16 : 0;
    if (lds > 160 * 1024) return;
    auto k = k_probe<MODE, D, NW, NCH, BF, MFMA>;
    if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, grid, block, lds, st, A, W, 4, out);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    const int iters = 200;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, grid, block, lds, st, A, W, 4, out);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / iters;
    const double bytes_cu = (double)NCH * NW * 4 * 1024;
    printf("%-8s mode %d D %2d NW %d %s %s: %6.2f us  %5.1f GB/s per CU  (lds %zu KB)\n", tag, MODE, D, NW, BF ? "bf16" : "f32 ", MFMA ? "mfma" : "load", us,
           bytes_cu / us / 1e3, lds / 1024);
    CK(hipGetLastError());
}

int main() {
    // f32 cell: K = 2048 = 128 chunks of 16;  bf16: K = 2048 = 64 chunks of 32.  A: [KC][mt_total = 4][64][16 B]; W: [256 tiles][KC][64][16 B]
    const size_t KCmax = 128;
    f32x4 *A, *W; float* out;
    CK(hipMalloc(&A, KCmax * 4 * 64 * 16)); CK(hipMalloc(&W, (size_t)256 * KCmax * 64 * 16)); CK(hipMalloc(&out, 64));
    std::vector<float> h(KCmax * 4 * 64 * 4);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hw((size_t)256 * KCmax * 64 * 4);
    for (auto& v : hw) v = (float)rand() / RAND_MAX - 0.5f;
    CK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    // f32, 4 waves (NCH = 32 chunks per wave)
#define F32(MODE, D, MF) run<MODE, D, 4, 32, false, MF>("f32/4w", A, W, out, st);
    F32(0, 2, false) F32(0, 3, false) F32(0, 4, false) F32(0, 6, false) F32(0, 8, false)
    F32(1, 2, false) F32(1, 3, false) F32(1, 4, false) F32(1, 6, false) F32(1, 8, false)
    F32(0, 3, true) F32(0, 4, true)
    F32(1, 3, true) F32(1, 4, true) F32(1, 6, true) F32(1, 8, true)
    // f32, 8 waves (16 chunks per wave)
#define F32W8(MODE, D, MF) run<MODE, D, 8, 16, false, MF>("f32/8w", A, W, out, st);
    F32W8(0, 3, false) F32W8(1, 3, false) F32W8(1, 4, false) F32W8(0, 3, true) F32W8(1, 3, true) F32W8(1, 4, true)
    // bf16, 8 waves (64 chunks of 32 k: 8 per wave) and 4 waves (16 per wave)
#define BF8(MODE, D, MF) run<MODE, D, 8, 8, true, MF>("bf16/8w", A, W, out, st);
    BF8(0, 3, false) BF8(0, 4, false) BF8(0, 6, false) BF8(1, 3, false) BF8(1, 4, false) BF8(1, 5, false)
    BF8(0, 3, true) BF8(1, 3, true) BF8(1, 4, true) BF8(1, 5, true)
#define BF4(MODE, D, MF) run<MODE, D, 4, 16, true, MF>("bf16/4w", A, W, out, st);
    BF4(0, 3, true) BF4(0, 6, true) BF4(1, 4, true) BF4(1, 6, true) BF4(1, 8, true)
    return 0;
}
