// Probe (round 5): the RECURRENT half of one encoder LSTM layer for T timesteps of 64 streams,
//   gates[64, 4H] = h_{t-1}[64, H] W_hh^T + gx_t   ->  LSTM cell  ->  h_t          (H = 1024, exact-f32 MFMA or bf16 operands)
// (A) launch per timestep: every launch streams its W_hh slice from global memory (what the engine's K = H cell does, in this
//     probe's tiling: 4 units x 4 gates x 64 rows per workgroup, 256 workgroups, K split over 4 waves);
// (B) weight-stationary + step-persistent: ONE launch, the workgroup's W_hh slice (16 columns x H: 64 KB f32 / 32 KB bf16) is
//     loaded into LDS once and stays there for all T timesteps; h_t is published write-through (sc1 stores), a grid barrier
//     (one monotonic counter per XCD-slot + a top counter, relaxed polls, one agent-scope acquire per workgroup) separates the
//     timesteps, the next step's A operand is read back from L2;
// (C) the barrier alone (B without the math).
// Both forms compute the same thing in the same summation order: (B) is checked against (A) bit for bit over all T steps
// (a stale h fragment anywhere shows as a mismatch).  Every spin is bounded (give-up code in `fail`).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o persist_cell_probe persist_cell_probe.hip && ./persist_cell_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) unsigned gu32;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int H = 1024, M = 64, MT = 4, NWG = 256, NWV = 4;     // 256 workgroups x 4 units = H
constexpr int NXCD = 8;

__device__ __forceinline__ unsigned short f2bf(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float sigmoid_(float x) { return 1.0f / (1.0f + expf(-x)); }

template <bool BF> struct Ops;
template <> struct Ops<false> {
    static constexpr int KCH = 16, KC = H / 16;
    __device__ static __forceinline__ void mma(f32x4& acc, const f32x4& a, const f32x4& b) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
    }
};
template <> struct Ops<true> {
    static constexpr int KCH = 32, KC = H / 32;
    __device__ static __forceinline__ void mma(f32x4& acc, const f32x4& a, const f32x4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
};

struct Args {
    const f32x4* W;          // [NWG][KC][64] fragments: lane = g*16 + col (col = gate*4 + unit), k = KCH*c + EPL*g + e
    f32x4* h[2];             // fragment-major [KC][MT][64] 16-byte pieces (f32: 4 values, bf16: 8), ping-pong
    float* hf;               // [T][M][H] f32 copy of every h_t (the check)
    const float* gx;         // [T][4H][M]   x-side pre-activations + bias (random), column-major like the engine's gx
    float* c0;               // [H][M] initial cell state
    unsigned* bar;           // [NXCD * 32 + 32]: per-slot arrival counters (128 B apart), top counter, fail word
    int T, t0;               // timesteps of this launch, index of the first
    int math;                // 0: barrier only
};

// one monotonic arrival counter per slot (workgroup % 8: the XCD under the observed dispatch rule; any placement is correct) and
// a top counter the slot's last arriver bumps; everybody polls the top counter.  Payload was stored write-through before.
__device__ __forceinline__ bool grid_barrier(unsigned* bar, unsigned epoch, int wg) {
    __shared__ int ok_s;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave drains its sc1 stores
    __syncthreads();
    if (threadIdx.x == 0) {
        gu32* slot = (gu32*)(bar + (wg % NXCD) * 32);
        gu32* top = (gu32*)(bar + NXCD * 32);
        const unsigned per = NWG / NXCD;
        const unsigned a = __hip_atomic_fetch_add(slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a + 1 == per * epoch) __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        unsigned spins = 0;
        while (__hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < NXCD * epoch) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { ok = 0; __hip_atomic_store((gu32*)(bar + NXCD * 32 + 16), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");       // ONE acquire per workgroup: drops this CU's stale lines
        ok_s = ok;
    }
    __syncthreads();
    return ok_s != 0;
}

// PERSIST: W slice in LDS for the whole launch, T steps with barriers; else one step, W from global
template <bool BF, bool PERSIST>
__global__ __launch_bounds__(256) void k_cell(const Args a) {
    using O = Ops<BF>;
    constexpr int KC = O::KC, NCH = KC / NWV;
    extern __shared__ f32x4 lds[];                               // PERSIST: [KC][64] W fragments, then red
    f32x4* Wl = lds;
    float* red = (float*)(lds + (PERSIST ? KC * 64 : 0));        // [NWV][64 rows][17]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wg = blockIdx.x;
    const f32x4* Wg = a.W + (size_t)wg * KC * 64;
    if constexpr (PERSIST) {
        for (int i = tid; i < KC * 64; i += 256) Wl[i] = Wg[i];
        __syncthreads();
    }
    // epilogue item of this thread: row = tid / 4, unit uu = tid % 4 (the 4 units of a row sit in 4 adjacent lanes)
    const int row = tid >> 2, uu = tid & 3, u = wg * 4 + uu;
    float c_state = a.c0[(size_t)u * M + row];
    for (int s = 0; s < a.T; ++s) {
        const int t = a.t0 + s;
        const f32x4* hin = a.h[t & 1];
        f32x4* hout = a.h[(t & 1) ^ 1];
        float x[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) x[g] = a.gx[((size_t)t * 4 * H + (size_t)g * H + u) * M + row];
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.math) {
#pragma unroll 4
            for (int i = 0; i < NCH; ++i) {
                const int c = w + i * NWV;
                f32x4 af[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) af[mt] = hin[((size_t)c * MT + mt) * 64 + lane];
                const f32x4 bfrag = PERSIST ? Wl[c * 64 + lane] : Wg[(size_t)c * 64 + lane];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) O::mma(acc[mt], af[mt], bfrag);
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(w * 64 + mt * 16 + 4 * (lane >> 4) + r) * 17 + (lane & 15)] = acc[mt][r];
        __syncthreads();
        float gsum[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v = 0.f;
#pragma unroll
            for (int ww = 0; ww < NWV; ++ww) v += red[(ww * 64 + row) * 17 + g * 4 + uu];
            gsum[g] = v + x[g];
        }
        const float c2 = sigmoid_(gsum[1]) * c_state + sigmoid_(gsum[0]) * tanhf(gsum[2]);
        const float h2 = sigmoid_(gsum[3]) * tanhf(c2);
        c_state = c2;
        a.hf[((size_t)t * M + row) * H + u] = h2;
        // the row's 4 units -> one 16-byte piece of the fragment-major h (f32) / half of one (bf16: 8 units per piece,
        // written as 8-byte halves), stored write-through (sc1): visible to the other XCDs without a release fence
        const float h1 = __shfl_down(h2, 1), h2b = __shfl_down(h2, 2), h3 = __shfl_down(h2, 3);
        if (uu == 0) {
            const int k0 = wg * 4;                               // first unit of this workgroup = K index of the next step
            if constexpr (!BF) {
                const int c = k0 >> 4, g = (k0 >> 2) & 3;
                typedef __attribute__((address_space(1))) unsigned long long gu64;
                gu64* dst = (gu64*)(hout + ((size_t)c * MT + (row >> 4)) * 64 + g * 16 + (row & 15));
                const unsigned long long lo = (unsigned long long)__float_as_uint(h2) | ((unsigned long long)__float_as_uint(h1) << 32);
                const unsigned long long hi = (unsigned long long)__float_as_uint(h2b) | ((unsigned long long)__float_as_uint(h3) << 32);
                __hip_atomic_store(dst, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // global_store_dwordx2 ... sc1
                __hip_atomic_store(dst + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                const int c = k0 >> 5, g = (k0 >> 3) & 3, half = (k0 >> 2) & 1;
                unsigned long long* dst = (unsigned long long*)(hout + ((size_t)c * MT + (row >> 4)) * 64 + g * 16 + (row & 15)) + half;
                const unsigned long long v = (unsigned long long)f2bf(h2) | ((unsigned long long)f2bf(h1) << 16) |
                                             ((unsigned long long)f2bf(h2b) << 32) | ((unsigned long long)f2bf(h3) << 48);
                __hip_atomic_store((__attribute__((address_space(1))) unsigned long long*)dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if constexpr (PERSIST) {
            if (s + 1 < a.T && !grid_barrier(a.bar, (unsigned)(s + 1), wg)) return;
        }
        __syncthreads();
    }
    a.c0[(size_t)u * M + row] = c_state;
}

static unsigned short h_bf(float x) { unsigned u; memcpy(&u, &x, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }

template <bool BF>
static void run(int T, int reps) {
    using O = Ops<BF>;
    constexpr int KC = O::KC;
    const size_t frag_units = (size_t)KC * 64;                   // 16-byte units per workgroup of W, per m-tile row block of h
    std::vector<f32x4> Wh((size_t)NWG * frag_units), h0((size_t)KC * MT * 64);
    srand(7);
    auto rnd = [] { return (float)rand() / RAND_MAX - 0.5f; };
    const int EPL = BF ? 8 : 4;
    auto fill = [&](std::vector<f32x4>& v, float scale) {
        for (auto& q : v) {
            if (!BF) q = f32x4{rnd() * scale, rnd() * scale, rnd() * scale, rnd() * scale};
            else { unsigned short e[8]; for (int i = 0; i < 8; ++i) e[i] = h_bf(rnd() * scale); memcpy(&q, e, 16); }
        }
    };
    (void)EPL;
    fill(Wh, 0.06f); fill(h0, 1.0f);
    std::vector<float> gx((size_t)T * 4 * H * M), c0((size_t)H * M);
    for (auto& v : gx) v = rnd();
    for (auto& v : c0) v = rnd();
    Args a{};
    f32x4 *dW, *dh[2]; float *dhf[2], *dgx, *dc; unsigned* dbar;
    CK(hipMalloc(&dW, Wh.size() * 16)); CK(hipMemcpy(dW, Wh.data(), Wh.size() * 16, hipMemcpyHostToDevice));
    for (int p = 0; p < 2; ++p) CK(hipMalloc(&dh[p], h0.size() * 16));
    for (int p = 0; p < 2; ++p) CK(hipMalloc(&dhf[p], sizeof(float) * (size_t)T * M * H));
    CK(hipMalloc(&dgx, gx.size() * 4)); CK(hipMemcpy(dgx, gx.data(), gx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dc, c0.size() * 4));
    CK(hipMalloc(&dbar, 4 * (NXCD * 32 + 32)));
    a.W = dW; a.h[0] = dh[0]; a.h[1] = dh[1]; a.gx = dgx; a.c0 = dc; a.bar = dbar; a.math = 1;
    const size_t lds_p = (size_t)KC * 64 * 16 + (size_t)NWV * 64 * 17 * 4, lds_l = (size_t)NWV * 64 * 17 * 4;
    CK(hipFuncSetAttribute((const void*)k_cell<BF, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto reset = [&] {
        CK(hipMemcpyAsync(dh[0], h0.data(), h0.size() * 16, hipMemcpyHostToDevice, st));
        CK(hipMemsetAsync(dh[1], 0, h0.size() * 16, st));
        CK(hipMemcpyAsync(dc, c0.data(), c0.size() * 4, hipMemcpyHostToDevice, st));
        CK(hipMemsetAsync(dbar, 0, 4 * (NXCD * 32 + 32), st));
    };
    auto timed = [&](auto&& body) {
        float best = 1e30f, sum = 0.f;
        for (int r = 0; r < reps + 1; ++r) {
            reset();
            CK(hipEventRecord(e0, st));
            body();
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r) { best = ms < best ? ms : best; sum += ms; }
        }
        return std::pair<float, float>(1e3f * best / T, 1e3f * sum / reps / T);
    };
    // (A) launch per step
    a.hf = dhf[0];
    auto ra = timed([&] { for (int t = 0; t < T; ++t) { Args b = a; b.T = 1; b.t0 = t; hipLaunchKernelGGL((k_cell<BF, false>), dim3(NWG), dim3(256), lds_l, st, b); } });
    // (B) persistent
    a.hf = dhf[1];
    auto rb = timed([&] { Args b = a; b.T = T; b.t0 = 0; hipLaunchKernelGGL((k_cell<BF, true>), dim3(NWG), dim3(256), lds_p, st, b); });
    unsigned barw[NXCD * 32 + 32];
    CK(hipMemcpy(barw, dbar, sizeof(barw), hipMemcpyDeviceToHost));
    std::vector<float> ha((size_t)T * M * H), hb((size_t)T * M * H);
    CK(hipMemcpy(ha.data(), dhf[0], ha.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), dhf[1], hb.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; int first_t = -1; double maxd = 0.0, asum = 0.0;
    for (size_t i = 0; i < ha.size(); ++i) {
        if (memcmp(&ha[i], &hb[i], 4)) { ++bad; if (first_t < 0) first_t = (int)(i / ((size_t)M * H)); }
        maxd = std::fmax(maxd, std::fabs((double)ha[i] - hb[i])); asum += std::fabs(ha[i]);
    }
    // (C) barrier only
    a.math = 0;
    auto rc = timed([&] { Args b = a; b.T = T; b.t0 = 0; hipLaunchKernelGGL((k_cell<BF, true>), dim3(NWG), dim3(256), lds_p, st, b); });
    // (B2) persistent launches of 2 steps (the streaming model step: W slice reloaded into LDS per launch, one barrier)
    a.math = 1; a.hf = dhf[1];
    auto rb2 = timed([&] { for (int t = 0; t < T; t += 2) { Args b = a; b.T = 2; b.t0 = t; CK(hipMemsetAsync(dbar, 0, 4 * (NXCD * 32 + 32), st));
                                                                hipLaunchKernelGGL((k_cell<BF, true>), dim3(NWG), dim3(256), lds_p, st, b); } });
    const double wbytes = (double)H * 4 * H * (BF ? 2 : 4);
    printf("{\"operands\": \"%s\", \"T\": %d, \"launch_per_step_us\": {\"min\": %.2f, \"avg\": %.2f}, \"persistent_us_per_step\": {\"min\": %.2f, \"avg\": %.2f}, "
           "\"barrier_only_us_per_step\": {\"min\": %.2f, \"avg\": %.2f}, \"persistent_2step_launches_us_per_step\": {\"min\": %.2f, \"avg\": %.2f}, "
           "\"bitwise_mismatches\": %zu, \"first_bad_step\": %d, \"max_abs_diff\": %.3g, \"mean_abs_h\": %.3g, \"barrier_fail_word\": %u, "
           "\"W_hh_MB\": %.1f, \"lds_bytes_per_workgroup\": %zu}\n",
           BF ? "bf16" : "f32", T, ra.first, ra.second, rb.first, rb.second, rc.first, rc.second, rb2.first, rb2.second, bad, first_t, maxd,
           asum / ha.size(), barw[NXCD * 32 + 16], wbytes / 1e6, lds_p);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 64, reps = argc > 2 ? atoi(argv[2]) : 5;
    run<false>(T, reps);
    run<true>(T, reps);
    return 0;
}
