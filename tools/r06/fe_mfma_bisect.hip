// fe_mfma_bisect.hip -- bisecting the victim of R: a COPY of k_fe_mel's compute part with switches (VAR bits), beside the real
// neighbour (operand ring depth 2).  Derived from fe_mfma_repro.hip -- stand-alone reproducer of round 6's R, built from the engine's two REAL kernels (no engine, no Python):
//   victim     k_fe_mel<10>   streaming log-mel front-end (radix-8 FFT in LDS, 5 waves, 46 592 B of LDS) over a fixed PCM ring,
//                             launched back to back on stream A, a per-row checksum of its output behind every launch;
//   neighbour  k_gemm<OpsBF16, EpiLinearT<4>, 4, 4, false, D>   the 64 x 64 vocabulary GEMM of the beam search (1024 x 2048 x 1536,
//                             bf16 MFMA, 4 waves, 66 560 B of LDS) in a loop on stream B.
// The two share no byte of memory.  Every victim launch must reproduce the first one (which runs alone) bit for bit.
//   ./fe_mfma_repro [launches = 1000] [lds_pad_bytes = 0] [neighbour launches per victim launch = 4]
// lds_pad_bytes = unused dynamic LDS of the victim: 51712 (98 304 B in total) keeps its workgroups off every CU that holds a
// workgroup of the neighbour (98 304 + 66 560 > 160 KB) -- the engine's fix (lasr_ctx::fe_lds_pad).
// build (from the repository root): hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/r06/fe_mfma_repro tools/r06/fe_mfma_repro.hip
#include "../../libreasr_amd/csrc/lasr_kernels.hip.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace lasr;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <class T> static T* up(const std::vector<T>& v) { T* d = nullptr; if (hipMalloc((void**)&d, v.size() * sizeof(T)) != hipSuccess) return nullptr; (void)hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice); return d; }
static unsigned rng_state = 12345u;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffff) / 65536.0f - 0.5f; }


// a copy of fft1024_power with switches (FV bits): 1 unit twiddles instead of the LDS tables, 2 no block barrier inside (the caller has
// one behind the table staging), 4 stop behind pass 1, 8 stop behind pass 2, 16 stop behind pass 3 (P[k] = |z[k]|^2 then)
template <int FV>
__device__ __forceinline__ void fft_var(cf (&v)[8], float2* z, float* P, const float2* s_tw512, const float2* s_tw1024, int j) {
    auto tw5 = [&](int i) -> float2 { if constexpr (FV & 1) return float2{1.f, 0.f}; else return s_tw512[i]; };
    auto finish = [&]() {
        wave_sync_lds();
        if constexpr (FV & 32) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { const int k = j + 64 * i; const float2 q = z[k]; P[k] = q.x * q.x + q.y * q.y; }
            if (j == 0) P[512] = 0.f;
        } else {
            for (int k = j; k <= 512; k += 64) { const float2 q = z[k & 511]; P[k] = q.x * q.x + q.y * q.y; }
        }
        wave_sync_lds();
    };
    if constexpr (!(FV & 64)) dft8(v);
    if constexpr (!(FV & 2)) __syncthreads();
#pragma unroll
    for (int k0 = 0; k0 < 8; ++k0) { const float2 tw = tw5(j * k0); const cf u = cmul(v[k0], cf{tw.x, tw.y}); z[k0 * 64 + j] = float2{u.x, u.y}; }
    if constexpr (FV & 4) { finish(); return; }
    wave_sync_lds();
    {
        const int k0 = j >> 3, b = j & 7;
#pragma unroll
        for (int aa = 0; aa < 8; ++aa) { const float2 q = z[k0 * 64 + 8 * aa + b]; v[aa] = cf{q.x, q.y}; }
        dft8(v);
        wave_sync_lds();
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) { const float2 tw = tw5(8 * b * k1); const cf u = cmul(v[k1], cf{tw.x, tw.y}); z[k0 * 64 + k1 * 8 + b] = float2{u.x, u.y}; }
    }
    if constexpr (FV & 8) { finish(); return; }
    wave_sync_lds();
    {
        const int k0 = j >> 3, k1 = j & 7;
#pragma unroll
        for (int b = 0; b < 8; ++b) { const float2 q = z[k0 * 64 + k1 * 8 + b]; v[b] = cf{q.x, q.y}; }
        dft8(v);
        wave_sync_lds();
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) z[k0 + 8 * k1 + 64 * k2] = float2{v[k2].x, v[k2].y};
    }
    if constexpr (FV & 16) { finish(); return; }
    wave_sync_lds();
    for (int k = j; k <= 512; k += 64) {
        const float2 zk = z[k & 511];
        const float2 zn = z[(512 - k) & 511];
        const cf e = cf{0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y)};
        const cf d = cf{0.5f * (zk.x - zn.x), 0.5f * (zk.y + zn.y)};
        float2 tw; if constexpr (FV & 1) tw = float2{1.f, 0.f}; else tw = s_tw1024[k];
        const cf o = cmul(cf{tw.x, tw.y}, mul_mi(d));
        const cf X = cadd(e, o);
        P[k] = X.x * X.x + X.y * X.y;
    }
    wave_sync_lds();
}

// VAR bits: 1 samples from a formula (no global loads in the waves), 2 a block barrier between table staging and everything else,
// 4 no mel stage (the frame's power spectrum summed per lane instead), 8 block barriers instead of the wave-local LDS ordering
// between the FFT passes (done by running the FFT's LDS traffic of each wave one after the other), 16 one wave per workgroup does the work (the others idle)
template <int VAR, int FV = -1>
__global__ __launch_bounds__(320) void k_fe_var(const FeMelArgs a) {
    constexpr int NSTACK = 10, NWV = 5;
    __shared__ float2 sz[NWV][512 + 8];
    __shared__ float sp[NWV][520];
    __shared__ float s_fbw[1536];
    __shared__ int s_fbs[128], s_fbo[129];
    __shared__ float2 s_tw512[512], s_tw1024[513];
    const int w = threadIdx.x >> 6, j = threadIdx.x & 63;
    if constexpr (VAR & 128) asm volatile("" ::: "v127");           // 128: the victim declares 128 registers
    if constexpr (VAR & 256) asm volatile("" ::: "v199");           // 256: ... 200
    const int tp = blockIdx.x >> 1, half = blockIdx.x & 1, row = blockIdx.y;
    const int pos = a.tp_pk[row] >> 4;
    const int NR = a.ring_chunks;
    const int age = (a.age_pk[row] >> (4 * tp)) & 15;
    const MelTables tab{s_fbw, s_fbs, s_fbo, s_tw512, s_tw1024};
    if constexpr (!(VAR & 32)) stage_mel_tables(tab, a.tw512, a.tw1024, a.fb_w, a.fb_start, a.fb_off, a.fb_nnz, a.n_mels);      // 32: no table staging
    if constexpr (VAR & 2) __syncthreads();
    const int f = half * NWV + w;
    const int head = (pos - age - a.n_window + 2 * NR) % NR;
    const float* ring = a.pcm + (size_t)row * NR * a.chunk;
    const int N = a.n_window * a.chunk;
    const int base = (a.frame0 + f) * a.hop - 512;
    auto sample = [&](int n) -> float {
        if (n < a.win_off || n >= a.win_off + a.win_len) return 0.f;
        if constexpr (VAR & 1) return (float)((n * 37 + row * 11 + f * 5) & 255) * (1.0f / 256.0f) - 0.5f;
        const float wv = a.window[n];
        int q = base + n;
        if (q < 0) q = -q;
        if (q >= N) q = 2 * (N - 1) - q;
        int ck = 0, wi = q;
        while (wi >= a.chunk) { wi -= a.chunk; ++ck; }
        int slot = head + ck;
        if (slot >= NR) slot -= NR;
        return ring[(size_t)slot * a.chunk + wi] * wv;
    };
    cf v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) { const int n = j + 64 * m; v[m] = cf{sample(2 * n), sample(2 * n + 1)}; }
    if constexpr (FV < 0) fft1024_power(v, sz[w], sp[w], s_tw512, s_tw1024, j); else fft_var<FV>(v, sz[w], sp[w], s_tw512, s_tw1024, j);
    float* out = a.pend + ((size_t)row * a.pend_frames + (size_t)tp * NSTACK + f) * a.n_mels;
    if constexpr (VAR & 4) {
        float acc0 = 0.f, acc1 = 0.f;
        for (int k = j; k < 512; k += 128) { acc0 += sp[w][k]; acc1 += sp[w][k + 64]; }
        out[j] = acc0; out[64 + j] = acc1;
    } else {
        mel_log(sp[w], tab, a.n_mels, j, out);
    }
}
template <int VAR, int FV = -1>
static int run(int launches, int lds_pad, int per) {
    constexpr int D = 2;
    const int M = 128, NM = 128, NFFT = 1024, HOP = 160, WIN = 400, CHUNK = 1280, NW_ = 3, NB = 2, NR = NW_ + NB - 1;
    // ---- the victim's constants: Hann window, twiddles, a triangular filterbank (any valid sparse filterbank will do)
    std::vector<float> win(NFFT, 0.f);
    for (int i = 0; i < WIN; ++i) win[(NFFT - WIN) / 2 + i] = 0.5f - 0.5f * cosf(2.0f * 3.14159265358979f * i / WIN);
    std::vector<float2> tw512(512), tw1024(513);
    for (int i = 0; i < 512; ++i) tw512[i] = float2{cosf(-2.f * 3.14159265358979f * i / 512), sinf(-2.f * 3.14159265358979f * i / 512)};
    for (int i = 0; i <= 512; ++i) tw1024[i] = float2{cosf(-2.f * 3.14159265358979f * i / 1024), sinf(-2.f * 3.14159265358979f * i / 1024)};
    std::vector<int> fb_start(NM), fb_off(NM + 1, 0); std::vector<float> fb_w;
    for (int m = 0; m < NM; ++m) {
        const int lo = 1 + 3 * m, width = 3 + m / 12;       // 3 .. 13 bins, ascending
        fb_start[m] = lo; fb_off[m] = (int)fb_w.size();
        for (int k = 0; k < width && lo + k <= 512; ++k) fb_w.push_back(1.0f - fabsf((k + 0.5f) / width * 2.f - 1.f));
    }
    fb_off[NM] = (int)fb_w.size();
    if (fb_w.size() > 1536) { printf("filterbank too large\n"); return 1; }
    std::vector<float> ring((size_t)M * NR * CHUNK);
    for (auto& x : ring) x = frand();
    FeMelArgs m{};
    m.window = up(win); m.tw512 = up(tw512); m.tw1024 = up(tw1024); m.fb_start = up(fb_start); m.fb_off = up(fb_off); m.fb_w = up(fb_w);
    m.n_mels = NM; m.hop = HOP; m.fb_nnz = (int)fb_w.size(); m.win_off = (NFFT - WIN) / 2; m.win_len = WIN;
    m.pcm = up(ring); m.chunk = CHUNK; m.n_window = NW_; m.ring_chunks = NR; m.frame0 = 9;          // (the reference geometry: T // 3 + 1)
    float* pend = nullptr; int* trow = nullptr; int* rpos = nullptr;
    CHECK(hipMalloc((void**)&pend, sizeof(float) * M * NB * 10 * NM)); CHECK(hipMalloc((void**)&trow, 4 * M)); CHECK(hipMalloc((void**)&rpos, 4 * M));
    CHECK(hipMemset(pend, 0, sizeof(float) * M * NB * 10 * NM));
    m.pend = pend; m.pend_frames = NB * 10; m.trow_out = trow; m.ring_pos = rpos;
    for (int r = 0; r < 512; ++r) { m.idx[r] = -1; m.tp_pk[r] = 0; m.age_pk[r] = 0; }
    for (int r = 0; r < M; ++r) { m.tp_pk[r] = (unsigned char)(((r % NR) << 4) | NB); m.age_pk[r] = (unsigned short)((NB - 1) | (0 << 4)); }
    // ---- the neighbour's operands: random bf16 bit patterns with small exponents
    const int K = 1536, V = 2048, ROWS = 1024, KC = K / 32;
    std::vector<unsigned short> a((size_t)KC * (ROWS / 16) * 64 * 8), w((size_t)(V / 64) * 4 * KC * 64 * 8);
    for (auto& x : a) x = (unsigned short)(0x3c00 + (int)(frand() * 512.f) + (frand() > 0 ? 0x8000 : 0));
    for (auto& x : w) x = (unsigned short)(0x3c00 + (int)(frand() * 512.f) + (frand() > 0 ? 0x8000 : 0));
    GemmArgs g{};
    g.A[0] = up(a); g.a_mt_total[0] = ROWS / 16; g.a_mt_off[0] = 0; g.KC[0] = KC; g.W[0] = up(w); g.M = ROWS; g.prio = getenv("PRIO") ? atoi(getenv("PRIO")) : 1;
    EpiLinearT<4>::Args e{};
    float* logits = nullptr; CHECK(hipMalloc((void**)&logits, sizeof(float) * ROWS * V));
    std::vector<float> bias(V, 0.25f);
    e.bias = up(bias); e.out = logits; e.ldo = V; e.n_rows = ROWS; e.t_idx = nullptr; e.T_row = nullptr; e.M = M; e.W = 8;
    hipStream_t sa, sb;
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    CHECK(hipFuncSetAttribute((const void*)k_fe_var<VAR, FV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 46592));
    unsigned* log = nullptr; CHECK(hipMalloc((void**)&log, sizeof(unsigned) * (size_t)(launches + 1) * M));
    RowSumArgs ra{}; ra.s[0] = RowSumSrc{pend, 2, 0, 0, NB * 10 * NM};
    for (int i = 0; i <= launches; ++i) {
        if (i > 0) for (int q = 0; q < per; ++q) hipLaunchKernelGGL((k_gemm<OpsBF16, EpiLinearT<4>, 4, 4, false, D>), dim3(V / 64, ROWS / 64), dim3(256), 0, sb, g, e);
        hipLaunchKernelGGL((k_fe_var<VAR, FV>), dim3(2 * NB, M), dim3(320), lds_pad, sa, m);
        hipLaunchKernelGGL(k_dbg_rowsum, dim3(M, 1), dim3(256), 0, sa, ra, M, 1, log + (size_t)i * M);
        if (i == 0) CHECK(hipStreamSynchronize(sa));
    }
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned> h((size_t)(launches + 1) * M);
    CHECK(hipMemcpy(h.data(), log, sizeof(unsigned) * h.size(), hipMemcpyDeviceToHost));
    int bad_l = 0, bad_r = 0; unsigned any = 0;
    for (int r = 0; r < M; ++r) any |= h[r];
    for (int i = 1; i <= launches; ++i) { int nb = 0; for (int r = 0; r < M; ++r) nb += h[(size_t)i * M + r] != h[r]; bad_r += nb; bad_l += nb != 0; }
    printf("victim variant %2d fft %2d, victim LDS pad %6d B, %d neighbour launches per victim launch: %d of %d launches differ from the first (%d rows)%s\n",
           VAR, FV, lds_pad, per, bad_l, launches, bad_r, any ? "" : "  [reference output is all zero?]");
    return 0;
}
int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 1000, pad = argc > 2 ? atoi(argv[2]) : 0, per = argc > 3 ? atoi(argv[3]) : 4;
    printf("neighbour wave priority %s\n", getenv("PRIO") ? getenv("PRIO") : "1");
    if (run<37, 7>(launches, pad, per)) return 1;              // the smallest failing victim
    if (run<37 + 128, 7>(launches, pad, per)) return 1;        // ... declaring 128 registers
    if (run<37 + 256, 7>(launches, pad, per)) return 1;        // ... 200 registers
    if (run<0>(launches, pad, per)) return 1;                  // the whole kernel's compute part
    return 0;
}
