// lasr_gemm.hip.h -- the skinny-GEMM core (MFMA) and its fused epilogues.
// Included from lasr_kernels.hip.h inside namespace lasr.
//
//   out[rows, N] = A[rows, K] * W[N, K]^T      rows = stream slots, W streamed from HBM
//
// A workgroup owns MT 16-row m-tiles x NT 16-column n-tiles and the whole K range, which it splits
// over its NW waves (wave w takes K chunks w, w+NW, ...).  Operands go global -> VGPR directly in
// MFMA-fragment order (one 16-byte load per lane = a perfectly coalesced 1 KiB wave load) through a
// register ring; the NW partial tiles are reduced through LDS and a fused epilogue finishes the
// cell / projection for the rows x units the workgroup owns.
//
// Operand types (Ops): every fragment is 16 bytes per lane in both.
//   OpsF32 : v_mfma_f32_16x16x4_f32, chunk = 16 k.  frag[lane = g*16+i][e] = X[16*tile+i][16c + 4g + e]
//            (four MFMAs per fragment pair; exact f32 fmaf chains, only the k order differs from a
//            sequential dot product)
//   OpsBF16: v_mfma_f32_16x16x32_bf16, chunk = 32 k.  frag[lane = g*16+i][e] = X[16*tile+i][32c + 8g + e]
//            (one MFMA per fragment pair; bf16 operands, f32 accumulate)
// Activations [rows, K] are stored [K/chunk][rows/16][64 lanes][16 B]; weights tile-major
// [n_tile][K/chunk][64 lanes][16 B], so every operand address below is in 16-byte units and the K
// loop is identical for both types.
//
// Tilings of the recurrent cells (U = hidden units per workgroup):
//   U = 8  ("C", encoder): MT = 2, NT = 2: 8 units x 4 gates x 32 rows; grid (H/8, M/32) = 256
//           workgroups at 64 rows.  2 A + 2 W fragment loads per chunk: the least L2->CU traffic a
//           1024-output per-CU tile allows (32 x 32).
//   U = 4  ("A", predictor / joint): MT = 4, NT = 1: 4 units x all gates x up to 64 rows; grid
//           (H/4, M/64).  Every weight byte is fetched exactly once and all 256 CUs pull on the
//           weight stream even when a single m-tile is active (decode).
// COMPACT epilogues (predictor path) gather the rows whose flag is set into dense m-tiles inside
// the kernel (ballot prefix scan -> LDS row map), so MFMA work follows the number of emitting
// streams while the weight stream stays full width.
//
// The K loop is a static schedule whenever KC == NCH * NW for a known NCH: both phases (x part,
// h part) form ONE fully unrolled chunk stream, so hipcc can count every load and emit partial
// s_waitcnt vmcnt(N) (a dynamic loop is drained to vmcnt(0) at each loop head), and
// sched_barrier(0) keeps the scheduler from sinking the loads next to their first use.

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short f32_to_bf16(float x) {   // round to nearest even
    unsigned u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short v) { return __uint_as_float((unsigned)v << 16); }

struct OpsF32 {
    static constexpr int KCH = 16, EPL = 4, BF = 0;
    typedef float elem;
    __device__ static __forceinline__ void mma(f32x4& acc, const f32x4& a, const f32x4& b) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
    }
    // element index of (row r, column k) in a fragment-major activation buffer
    __device__ static __forceinline__ size_t aoff(int r, int k, int mt_total) {
        return ((size_t)((k >> 4) * mt_total + (r >> 4)) * 64 + (((k >> 2) & 3) * 16 + (r & 15))) * 4 + (k & 3);
    }
    __device__ static __forceinline__ float ld(const void* p, size_t i) { return ((const float*)p)[i]; }
    __device__ static __forceinline__ void st(void* p, size_t i, float v) { ((float*)p)[i] = v; }
};
struct OpsBF16 {
    static constexpr int KCH = 32, EPL = 8, BF = 1;
    typedef unsigned short elem;
    __device__ static __forceinline__ void mma(f32x4& acc, const f32x4& a, const f32x4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
    __device__ static __forceinline__ size_t aoff(int r, int k, int mt_total) {
        return ((size_t)((k >> 5) * mt_total + (r >> 4)) * 64 + (((k >> 3) & 3) * 16 + (r & 15))) * 8 + (k & 7);
    }
    __device__ static __forceinline__ float ld(const void* p, size_t i) { return bf16_to_f32(((const unsigned short*)p)[i]); }
    __device__ static __forceinline__ void st(void* p, size_t i, float v) { ((unsigned short*)p)[i] = f32_to_bf16(v); }
};
// The same bf16 operands through the OLDER matrix instruction: two v_mfma_f32_16x16x16_bf16 per fragment pair (elements 0-3 and 4-7 of
// both fragments: the k pairing of A and B is the same, only the summation is split in two).  Used by the WIDE decode tilings (64 x 64
// outputs per workgroup: 16 MFMAs per 8 fragment loads): beside a workgroup that streams operands into v_mfma_f32_16x16x32_bf16 at
// that density, waves of OTHER kernels on the same CU computed wrong results (round 6, R: DESIGN 5a); with this instruction in the
// same kernel: never (tools/r06/fe_mfma_repro.hip: 0 of 1 000 against 37 / 619).  These kernels wait for L2, not for the matrix pipe.
typedef short s16x4 __attribute__((ext_vector_type(4)));
struct OpsBF16k16 : OpsBF16 {
    __device__ static __forceinline__ void mma(f32x4& acc, const f32x4& a, const f32x4& b) {
        const s16x8 a8 = __builtin_bit_cast(s16x8, a), b8 = __builtin_bit_cast(s16x8, b);
        const s16x4 a0 = {a8[0], a8[1], a8[2], a8[3]}, a1 = {a8[4], a8[5], a8[6], a8[7]};
        const s16x4 b0 = {b8[0], b8[1], b8[2], b8[3]}, b1 = {b8[4], b8[5], b8[6], b8[7]};
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1, b1, acc, 0, 0, 0);
    }
};
template <class Ops> struct WideOps { typedef Ops type; };
template <> struct WideOps<OpsBF16> { typedef OpsBF16k16 type; };

// runtime-typed access for the small kernels (is_bf16 flag)
__device__ __forceinline__ size_t act_off(int bf, int r, int k, int mt_total) {
    return bf ? OpsBF16::aoff(r, k, mt_total) : OpsF32::aoff(r, k, mt_total);
}
__device__ __forceinline__ float act_ld(int bf, const void* p, size_t i) { return bf ? OpsBF16::ld(p, i) : OpsF32::ld(p, i); }
__device__ __forceinline__ void act_st(int bf, void* p, size_t i, float v) {
    if (bf) OpsBF16::st(p, i, v); else OpsF32::st(p, i, v);
}

constexpr int PROF_W = 1024;     // per-launch slots of the in-job kernel timer (workgroup id modulo PROF_W)
struct GemmArgs {
    const void* A[2];       // phase operand: fragment-major (row-major when AROW), element type of Ops
    int a_mt_total[2];      // m-tiles in A's fragment layout (lda in elements when AROW)
    int a_mt_off[2];        // m-tile index of row 0 inside A
    int KC[2];              // K chunks (of Ops::KCH) per phase; 0 = phase absent
    const void* W[2];       // packed weights: [n_group][slot][KC][fragment]
    int a_rows;             // AROW only: loads of rows >= a_rows are clamped (0 = no clamp)
    const int* compact;     // COMPACT epilogues: per-row flag
    int M;                  // rows scanned for compaction
    const int* parent;      // beam search (AROW, W > 1): phase-1 rows are read from the row's parent hypothesis
    int beam_w;             //   slot: row r -> (r / W) * W + parent[r]; nullptr / 0: identity (greedy)
    unsigned long long* dbg; // optional per-workgroup phase timestamps [blocks][16] (LASR_DBG_TIMING)
    unsigned long long* prof; // optional: per-workgroup entry (prof[wg % PROF_W]) and exit (prof_x[wg % PROF_W]) wall_clock64(), plain stores --
    unsigned long long* prof_x; //   max exit - min entry over the workgroups = the kernel's own duration (atomics on one word cost the job 7 %)
    int prio;                // wave priority for the whole kernel (s_setprio 0..3): 1 for the decode-stream GEMMs, 0 otherwise
    int skip_idle;           // COMPACT: an m-group without a compacted row returns at once (its rows' carry is done by k_beam_carry)
};

// beam search: physical row of the parent hypothesis of row r (W slots per stream, contiguous)
__device__ __forceinline__ int beam_prow(const int* parent, int W, int r) { return (r / W) * W + parent[r]; }

template <int MASK>
struct PopCount {
    static constexpr int value = (MASK & 1) + PopCount<(MASK >> 1)>::value;
};
template <>
struct PopCount<0> {
    static constexpr int value = 0;
};

template <int MT, int NS>
struct Frag {
    f32x4 a[MT];                       // 16-byte containers (f32x4 or 8 x bf16)
    f32x4 b[NS > 0 ? NS : 1];
};

// weight-fragment addressing of one phase.  DEAD >= 0: columns [DEAD, DEAD+4) of the 16-column tile
// carry no weights in this phase; the fragment stores only the 12 live columns (768 B) and the dead
// lanes feed zeros to the MFMA.
template <int DEAD>
struct WLane {
    int off;      // in 16-byte units
    bool live;
    static constexpr int FRU = DEAD >= 0 ? 48 : 64;    // 16-byte units per fragment
    __device__ __forceinline__ explicit WLane(int lane) {
        if constexpr (DEAD >= 0) {
            const int col = lane & 15, gq = lane >> 4;
            live = !(col >= DEAD && col < DEAD + 4);
            off = gq * 12 + (col < DEAD ? col : col - 4);
        } else {
            live = true;
            off = lane;
        }
    }
};

// Dynamic fallback (any KC): per-phase loop, D-deep ring with clamped (always valid, countable) loads.
template <class Ops, int TILES, int DEAD, int MT, int MTP, int NT, int NW, int D>
__device__ __forceinline__ void gemm_phase(f32x4 (&acc)[MT][NT], const f32x4* const (&aptr)[MT], size_t a_step,
                                           const f32x4* __restrict__ Wp, int KC, int jb, int w, int wu, int lane) {
    constexpr int NS = PopCount<TILES>::value;
    if constexpr (NS == 0) {
        return;
    } else {
        if (KC <= 0) return;
        const WLane<DEAD> wl(lane);
        const f32x4* wb = Wp + (size_t)jb * NS * KC * WLane<DEAD>::FRU + wl.off;
        const int n = (KC - wu + NW - 1) / NW;         // chunks of this wave (<= 0: none); SGPR: real branches
        auto load = [&](Frag<MTP, NS>& f, int i) {
            int c = w + i * NW;
            c = c < KC ? c : KC - 1;                   // clamp: the load stays valid and countable
#pragma unroll
            for (int mt = 0; mt < MTP; ++mt) f.a[mt] = aptr[mt][(size_t)c * a_step];
#pragma unroll
            for (int s = 0; s < NS; ++s)
                f.b[s] = wl.live ? wb[((size_t)s * KC + c) * WLane<DEAD>::FRU] : f32x4{0.f, 0.f, 0.f, 0.f};
        };
        auto compute = [&](const Frag<MTP, NS>& f) {
#pragma unroll
            for (int mt = 0; mt < MTP; ++mt) {
                int s = 0;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    if ((TILES >> nt) & 1) { Ops::mma(acc[mt][nt], f.a[mt], f.b[s]); ++s; }
            }
        };
        Frag<MTP, NS> f[D];
#pragma unroll
        for (int d = 0; d < D - 1; ++d) load(f[d], d);
        for (int i = 0; i < n; i += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                load(f[(d + D - 1) % D], i + d + D - 1);
                if (d == 0 || i + d < n) compute(f[d]);
            }
        }
    }
}

// Static K schedule: NCH0 + NCH1 chunks per wave (KC == NCH * NW exactly), fully unrolled.
template <class Ops, class Epi, int MT, int MTP, int NT, int NW, int D, int NCH0, int NCH1, bool AROW>
__device__ __forceinline__ void gemm_static(f32x4 (&acc)[MT][NT], const f32x4* const (&ap0)[MT], size_t a_step0,
                                            const f32x4* const (&ap1)[MT], size_t a_step1, const GemmArgs& g, int jb,
                                            int w, int lane, const f32x4* wpre) {
    constexpr int NS0 = PopCount<Epi::PH0_TILES>::value, NS1 = PopCount<Epi::PH1_TILES>::value;
    constexpr int NSM = NS0 > NS1 ? NS0 : NS1;
    constexpr int N0 = NS0 > 0 ? NCH0 : 0, N1 = NS1 > 0 ? NCH1 : 0, NTOT = N0 + N1;
    const WLane<Epi::PH0_DEAD> wl0(lane);
    const WLane<Epi::PH1_DEAD> wl1(lane);
    constexpr int FR0 = WLane<Epi::PH0_DEAD>::FRU, FR1 = WLane<Epi::PH1_DEAD>::FRU;
    const int KC0 = NCH0 * NW, KC1 = NCH1 * NW;
    const f32x4* wb0 = NS0 > 0 ? (const f32x4*)g.W[0] + (size_t)jb * NS0 * KC0 * FR0 + wl0.off : nullptr;
    const f32x4* wb1 = NS1 > 0 ? (const f32x4*)g.W[1] + (size_t)jb * NS1 * KC1 * FR1 + wl1.off : nullptr;
    Frag<MTP, NSM> f[D];
    auto load = [&](Frag<MTP, NSM>& fr, int i) {       // i is a compile-time constant after unrolling
        // row-major A: a wave takes chunk PAIRS (two 64-B halves of the same 128-B lines, second hits L1)
        auto chunk = [&](int q) { return AROW ? ((q >> 1) * NW + w) * 2 + (q & 1) : w + q * NW; };
        if (i < N0) {
            const int c = chunk(i);
#pragma unroll
            for (int mt = 0; mt < MTP; ++mt) fr.a[mt] = ap0[mt][(size_t)c * a_step0];
#pragma unroll
            for (int s = 0; s < NS0; ++s)
                fr.b[s] = (i == 0 && wpre) ? wpre[s]          // chunk 0 was issued at kernel entry
                          : wl0.live ? wb0[((size_t)s * KC0 + c) * FR0] : f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            const int c = chunk(i - N0);
#pragma unroll
            for (int mt = 0; mt < MTP; ++mt) fr.a[mt] = ap1[mt][(size_t)c * a_step1];
#pragma unroll
            for (int s = 0; s < NS1; ++s)
                fr.b[s] = (i == 0 && wpre) ? wpre[s]
                          : wl1.live ? wb1[((size_t)s * KC1 + c) * FR1] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto compute = [&](const Frag<MTP, NSM>& fr, int i) {
        const int tiles = i < N0 ? Epi::PH0_TILES : Epi::PH1_TILES;
#pragma unroll
        for (int mt = 0; mt < MTP; ++mt) {
            int s = 0;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                if ((tiles >> nt) & 1) { Ops::mma(acc[mt][nt], fr.a[mt], fr.b[s]); ++s; }
        }
    };
    // sched_barrier(0): hipcc's scheduler otherwise sinks the loads next to their first use
#pragma unroll
    for (int d = 0; d < D - 1; ++d)
        if (d < NTOT) load(f[d], d);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NTOT; ++i) {
        if (i + D - 1 < NTOT) load(f[(i + D - 1) % D], i + D - 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(f[i % D], i);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NW, int ROWS, int LD>
struct RedView {
    const float* p;
    __device__ __forceinline__ float sum(int row, int col) const {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += p[(w * ROWS + row) * LD + col];
        return s;
    }
};

// One workgroup = (n-group jb = blockIdx.x, m-group mg = blockIdx.y of MT m-tiles).
// D > 0: ring depth; D < 0: latency-bound kernel, depth from the register budget.
// gemm_body: the workgroup's work with its LDS handed in (k_gemm: one problem per launch; k_gemm_multi: several independent
// problems of the same kind in one launch, blockIdx.z selects the problem).
template <class Ops, class Epi, int MT, int NW, bool AROW, int D>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, const typename Epi::Args& ea, const int jb, const int mg,
                                          float* red, int* row_map, int& n_act_s) {
    constexpr int NT = Epi::NT, ROWS = MT * 16, LD = NT * 16 + 1;
    // Loop bounds derived from the wave index must be wave-uniform FOR THE COMPILER: a condition it
    // believes divergent is lowered to EXEC masking, and MFMA ignores EXEC (a masked-off v_mfma still
    // accumulates) -- seen as double-counted K chunks in the bf16 path.  Addresses keep the VGPR copy
    // (hipcc schedules the fully unrolled K stream better with vector address math).
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wu = __builtin_amdgcn_readfirstlane(w);       // the same value, as an SGPR: control flow only
    if (g.prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (g.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (g.prio == 1) __builtin_amdgcn_s_setprio(1);
    unsigned long long* dbg = g.dbg ? g.dbg + ((size_t)mg * gridDim.x + jb) * 16 : nullptr;
    if (dbg && tid == 0) { dbg[0] = __builtin_amdgcn_s_memtime(); dbg[5] = wall_clock64(); }
    const int prof_wg = (int)(((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) % PROF_W);
    if (g.prof && tid == 0) g.prof[prof_wg] = (unsigned long long)wall_clock64();

    // (0) this wave's first weight fragment does not depend on flags / compaction: put it in flight
    //     before anything else (first-touch latency of the weight stream is the long pole)
    constexpr int NS0k = PopCount<Epi::PH0_TILES>::value, NS1k = PopCount<Epi::PH1_TILES>::value;
    constexpr int NSMk = NS0k > NS1k ? NS0k : NS1k;
    f32x4 wpre[NSMk];
    {
        constexpr int PHF = NS0k > 0 ? 0 : 1;                                  // first phase with tiles
        constexpr int NSF = NS0k > 0 ? NS0k : NS1k;
        constexpr int DEADF = NS0k > 0 ? Epi::PH0_DEAD : Epi::PH1_DEAD;
        const WLane<DEADF> wl(lane);
        const int KCF = g.KC[PHF];
        int c0 = AROW ? 2 * w : w;
        c0 = c0 < KCF ? c0 : KCF - 1;
        const f32x4* wb = (const f32x4*)g.W[PHF] + (size_t)jb * NSF * KCF * WLane<DEADF>::FRU + wl.off;
#pragma unroll
        for (int s = 0; s < NSMk; ++s)
            wpre[s] = (s < NSF && wl.live) ? wb[((size_t)s * KCF + c0) * WLane<DEADF>::FRU] : f32x4{0.f, 0.f, 0.f, 0.f};
    }

    int n_act = g.M;
    if constexpr (Epi::COMPACT) {
        if (w == 0) {   // ballot prefix scan over the row flags (M <= 1024).  The flags of all (up to 16) 64-row chunks are loaded
            // up front: one round trip instead of one per chunk (1024 hypothesis rows: 6 us of every workgroup's life before)
            int cnt = 0;
            const int nch = (g.M + 63) >> 6;
            int fl[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = i * 64 + lane;
                fl[i] = (i < nch && r < g.M) ? g.compact[r] : 0;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i < nch) {
                    const bool f = fl[i] != 0;
                    const unsigned long long m = __ballot(f);
                    if (f) row_map[cnt + __popcll(m & ((1ull << lane) - 1ull))] = i * 64 + lane;
                    cnt += __popcll(m);
                }
            }
            if (lane == 0) n_act_s = cnt;
        }
        __syncthreads();
        n_act = __builtin_amdgcn_readfirstlane(n_act_s);
        if (g.skip_idle && mg * ROWS >= n_act) return;       // uniform over the workgroup
    }

    bool tile_on[MT];
    const f32x4* ap0[MT];
    const f32x4* ap1[MT];
    size_t a_step0, a_step1;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int vr = mg * ROWS + mt * 16 + (lane & 15);   // virtual row of this lane's A slot
        int orow;
        if constexpr (Epi::COMPACT) {
            tile_on[mt] = (mg * ROWS + mt * 16) < n_act;
            orow = vr < n_act ? row_map[vr] : (n_act > 0 ? row_map[n_act - 1] : 0);
        } else {
            tile_on[mt] = Epi::tile_active(ea, mg * MT + mt, lane);
            orow = vr;
        }
        if constexpr (AROW) {   // row-major: lane (i, g) reads 16 B of row i at element KCH*c + EPL*g
            int rr = orow;
            if (g.a_rows > 0 && rr >= g.a_rows) rr = g.a_rows - 1;
            const int rp = g.parent ? beam_prow(g.parent, g.beam_w, rr) : rr;     // recurrent state lives in the parent's slot
            ap0[mt] = (const f32x4*)g.A[0] + (size_t)rr * (g.a_mt_total[0] / Ops::EPL) + (lane >> 4);
            ap1[mt] = g.A[1] ? (const f32x4*)g.A[1] + (size_t)rp * (g.a_mt_total[1] / Ops::EPL) + (lane >> 4) : nullptr;
        } else {
            const size_t in_tile = (size_t)(lane >> 4) * 16 + (orow & 15);
            ap0[mt] = (const f32x4*)g.A[0] + (size_t)(g.a_mt_off[0] + (orow >> 4)) * 64 + in_tile;
            ap1[mt] = (const f32x4*)g.A[1] + (size_t)(g.a_mt_off[1] + (orow >> 4)) * 64 + in_tile;
        }
    }
    if constexpr (AROW) {
        a_step0 = 4; a_step1 = 4;                        // KCH / EPL 16-byte units per chunk
    } else {
        a_step0 = (size_t)g.a_mt_total[0] * 64; a_step1 = (size_t)g.a_mt_total[1] * 64;
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (dbg && tid == 0) dbg[1] = __builtin_amdgcn_s_memtime();
    // (1) operands of this thread's epilogue item (biases, old state, table rows): issued now so
    //     their latency overlaps the K loop instead of trailing it
    const typename Epi::Pre pre = Epi::template prefetch<MT>(ea, tid, jb, mg, n_act, row_map);

    // number of leading m-tiles to process: 1 + index of the highest active tile (rows of inactive
    // tiles inside the prefix are computed and then masked by the epilogue)
    int P = 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        if (tile_on[mt]) P = mt + 1;
    auto run_phases = [&](auto mtp_tag) {
        constexpr int MTP = decltype(mtp_tag)::value;
        constexpr bool P0 = NS0k > 0, P1 = NS1k > 0;
        constexpr int FPC = MTP + NSMk;                                   // fragments per chunk
        // latency-bound kernels (D < 0): ring depth by fragments per chunk.  f32 operands want a SHALLOW ring
        // (whole-job +7 % with 3 instead of 9: the memory path degrades with more requests in flight),
        // bf16 operands (half the bytes per request) keep the deep one (+2 %)
        constexpr int DD = D > 0 ? D
                           : Ops::BF ? (FPC <= 2 ? 9 : FPC <= 3 ? 7 : FPC <= 5 ? 4 : 3)
                                     : (FPC <= 5 && FPC > 3 ? 4 : 3);
        const int k0 = P0 ? g.KC[0] : 0, k1 = P1 ? g.KC[1] : 0;
#define LASR_TRY(N0, N1)                                                                                              \
    if ((!P0 || k0 == (N0) * NW) && (!P1 || k1 == (N1) * NW)) {                                                       \
        gemm_static<Ops, Epi, MT, MTP, NT, NW, DD, N0, N1, AROW>(acc, ap0, a_step0, ap1, a_step1, g, jb, w, lane, wpre); \
        return;                                                                                                        \
    }
        // static schedules for the shapes of the shipped / benchmarked models (chunks per wave and phase)
        if constexpr (Ops::BF) {
            LASR_TRY(4, 4)       // K = 1024 / 1024
            LASR_TRY(5, 4)       // K = 1280 / 1024  (encoder layer 0)
            LASR_TRY(6, 6)       // K = 1536 / 1536
            LASR_TRY(5, 6)       // K = 1280 / 1536
            LASR_TRY(12, 12)     // K = 1536 / 1536 on 4 waves (the wide decode tilings of configs[4])
            LASR_TRY(8, 8)       // K = 1024 / 1024 on 4 waves
        } else {
            LASR_TRY(4, 4)       // (16-wave K split)
            LASR_TRY(5, 4)
            LASR_TRY(16, 16)     // (4-wave K split)
            LASR_TRY(20, 16)
            LASR_TRY(8, 8)
            LASR_TRY(10, 8)
            LASR_TRY(12, 12)
            LASR_TRY(10, 12)
        }
#undef LASR_TRY
        gemm_phase<Ops, Epi::PH0_TILES, Epi::PH0_DEAD, MT, MTP, NT, NW, (DD > 4 ? 4 : DD)>(acc, ap0, a_step0, (const f32x4*)g.W[0], g.KC[0], jb, w, wu, lane);
        gemm_phase<Ops, Epi::PH1_TILES, Epi::PH1_DEAD, MT, MTP, NT, NW, (DD > 4 ? 4 : DD)>(acc, ap1, a_step1, (const f32x4*)g.W[1], g.KC[1], jb, w, wu, lane);
    };
    if constexpr (MT == 1) {
        if (P == 1) run_phases(std::integral_constant<int, 1>{});   // wave-uniform: an idle workgroup streams nothing
    } else {
        if (P == MT) run_phases(std::integral_constant<int, MT>{});
        else if (P == 1) run_phases(std::integral_constant<int, 1>{});
        else if (P == 2) run_phases(std::integral_constant<int, 2>{});
        else if (P == 3) run_phases(std::integral_constant<int, (MT > 3 ? 3 : MT)>{});
    }

    if (dbg && tid == 0) dbg[2] = __builtin_amdgcn_s_memtime();
    if (dbg && lane == 0 && wu < 8) dbg[8 + wu] = __builtin_amdgcn_s_memtime();   // per-wave end of the K loop
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[(w * ROWS + mt * 16 + 4 * (lane >> 4) + r) * LD + nt * 16 + (lane & 15)] = acc[mt][nt][r];
    __syncthreads();
    if (dbg && tid == 0) dbg[3] = __builtin_amdgcn_s_memtime();
    Epi::template run<MT>(ea, RedView<NW, ROWS, LD>{red}, tid, jb, mg, n_act, row_map, pre, NW * 64);
    if (dbg && tid == 0) { dbg[4] = __builtin_amdgcn_s_memtime(); dbg[6] = wall_clock64(); }
    if (g.prof && tid == 0) g.prof_x[prof_wg] = (unsigned long long)wall_clock64();
}

template <class Ops, class Epi, int MT, int NW, bool AROW, int D>
__global__ __launch_bounds__(NW * 64) void k_gemm(const GemmArgs g, const typename Epi::Args ea) {
    constexpr int NT = Epi::NT, ROWS = MT * 16, LD = NT * 16 + 1;
    __shared__ float red[NW * ROWS * LD];
    __shared__ int row_map[Epi::COMPACT ? 1024 : 1];
    __shared__ int n_act_s;
    gemm_body<Ops, Epi, MT, NW, AROW, D>(g, ea, blockIdx.x, blockIdx.y, red, row_map, n_act_s);
}

// Up to NPMAX independent problems of one kind in ONE launch (grid.z = problems): the encoder's layer wavefront -- cell (l, t)
// and cell (l + 1, t - 1) depend on nothing of each other, so the cells of an anti-diagonal share a launch and its fixed
// costs (launch gap, prologue, LDS reduction, epilogue overlap the other cells' K loops).
constexpr int NPMAX = 8;
template <class Epi>
struct MultiArgs {
    GemmArgs g[NPMAX];
    typename Epi::Args ea[NPMAX];
};
template <class Ops, class Epi, int MT, int NW, bool AROW, int D>
__global__ __launch_bounds__(NW * 64) void k_gemm_multi(const MultiArgs<Epi> m) {
    constexpr int NT = Epi::NT, ROWS = MT * 16, LD = NT * 16 + 1;
    __shared__ float red[NW * ROWS * LD];
    __shared__ int row_map[Epi::COMPACT ? 1024 : 1];
    __shared__ int n_act_s;
    const int p = blockIdx.z;
    gemm_body<Ops, Epi, MT, NW, AROW, D>(m.g[p], m.ea[p], blockIdx.x, blockIdx.y, red, row_map, n_act_s);
}

// Two INDEPENDENT problems of different kinds in one launch (1-D grid: the first na workgroups are problem A's, in its own
// (n-group fastest) order, the rest problem B's): the LM step and the predictor / joint chain of a decode iteration both start
// from the token the selection kernel has just written and meet again at the next selection -- layer l of the one and stage l of
// the other share a launch, so the two chains overlap without a second stream (whose hardware queue the runtime picks at graph
// replay: see DESIGN.md, LM shallow fusion).  A problem built for fewer waves than the launch has leaves the surplus waves at
// entry (s_barrier counts the surviving waves of a workgroup).
template <class Ops, class EpiA, int MTa, int NWa, bool AROWa, int Da, class EpiB, int MTb, int NWb, bool AROWb, int Db>
__global__ __launch_bounds__((NWa > NWb ? NWa : NWb) * 64) void k_gemm2(const GemmArgs ga, const typename EpiA::Args ea, const int nga,
                                                                          const int na, const GemmArgs gb, const typename EpiB::Args eb,
                                                                          const int ngb) {
    constexpr int SA = NWa * (MTa * 16) * (EpiA::NT * 16 + 1), SB = NWb * (MTb * 16) * (EpiB::NT * 16 + 1);
    __shared__ float red[SA > SB ? SA : SB];
    __shared__ int row_map[(EpiA::COMPACT || EpiB::COMPACT) ? 1024 : 1];
    __shared__ int n_act_s;
    const int id = blockIdx.x;
    if (id < na) {
        if (NWa < NWb && (int)threadIdx.x >= NWa * 64) return;
        gemm_body<Ops, EpiA, MTa, NWa, AROWa, Da>(ga, ea, id % nga, id / nga, red, row_map, n_act_s);
    } else {
        if (NWb < NWa && (int)threadIdx.x >= NWb * 64) return;
        const int j = id - na;
        gemm_body<Ops, EpiB, MTb, NWb, AROWb, Db>(gb, eb, j % ngb, j / ngb, red, row_map, n_act_s);
    }
}

// ------------------------------------------------------------------------------------------------
// Epilogues.  prefetch / run are force-inlined: left to itself hipcc emitted EpiNBRC<f32, non-table>::run out of line
// (kernel arguments copied to scratch, a call, flat loads through the argument pointer: +2 % whole job when inlined).  A tile column of (gate g, unit uu) is g*U + uu.  Buffers that are A operands of another
// GEMM (h, BN(h), the joint activation) hold Ops::elem; everything else is f32.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool any16(bool flag, int lane) {
    return (__ballot(flag && lane < 16) != 0ull);   // wave-uniform OR over lanes 0..15
}

// ---- LSTM cell (torch gate order i,f,g,o; custom_rnn.py:172, haste/lstm.py:34-68) + BN(eval) fold.
// h ping-pongs between two buffers (every workgroup reads all of h while others write their units);
// rows that do not advance are carried to the other buffer by the workgroup that owns the units.
// ENC: fragment-major state; row r advances at step t iff t < T_row[r].
// PRED (COMPACT): row-major state [M][H]; the rows that emitted advance; phase X is the per-token
// table tab[token][4H] when TABLE.
// Each workgroup tile has exactly ROWS*U = 256 (row, unit) items: one per thread.
template <class Ops, bool PRED, bool TABLE, int U>
struct EpiLSTM {
    static constexpr int NT = U / 4;                       // 4 gates x U units = NT 16-column tiles
    static constexpr int PH0_TILES = TABLE ? 0 : ((1 << NT) - 1);
    static constexpr int PH1_TILES = (1 << NT) - 1;
    static constexpr int PH0_DEAD = -1, PH1_DEAD = -1;
    static constexpr bool COMPACT = PRED;
    struct Args {
        const float* bias;     // [4H] = b_ih + b_hh (folded into tab when TABLE)
        const float* tab;      // [V][4H]
        const int* token;      // [M]
        const int* flag;       // ENC: T_row[M];  PRED: emit[M]
        int t;                 // ENC: time step
        unsigned long long tile_mask;   // ENC: bit mt set iff m-tile mt has a row with t < T_row (host-computed:
                               //      no global load sits in front of the first weight load)
        float* c;              // [H][M] cell state (f32), in place (beam: the OTHER parity, see c_in)
        const void* h_in;      // current parity
        void* h_out;           // other parity
        void* y;               // BN(h'); ENC fragment-major (may be nullptr), PRED row-major
        int y_mt_total, y_mt_off;
        const float* bn_s;
        const float* bn_t;
        int H, M, MT;
        // beam search (PRED, W > 1): every hypothesis slot may be re-parented each round, so c and y
        // ping-pong like h, old state is read from the parent's slot and rows that are not extended
        // carry (h, c, y) of their parent into their own slot of the other parity
        const int* parent;
        int W;
        const float* c_in;
        const void* y_in;
        int no_carry;          // beam: the rows that are not extended have been carried by k_beam_carry
    };
    struct Pre {
        int r;                 // row this thread finishes (-1: none)
        bool carry;            // PRED: this thread carries h of original row vr (did not emit)
        float carry_h, carry_c, carry_y;
        bool act;
        float x[4];            // bias or table values per gate
        float c_old, h_old, s, t;
    };
    __device__ static bool tile_active(const Args& a, int mt, int lane) { return (a.tile_mask >> mt) & 1ull; }
    __device__ static size_t hidx(const Args& a, int r, int u) { return PRED ? (size_t)r * a.H + u : Ops::aoff(r, u, a.MT); }
    template <int MTB>
    __device__ static __forceinline__ Pre prefetch(const Args& a, int tid, int jb, int mg, int n_act, const int* row_map) {
        constexpr int ROWS = MTB * 16;
        static_assert(ROWS * U <= 256, "at most one (row, unit) item per thread");
        Pre p;
        p.r = -1; p.carry = false; p.act = false;
        if (tid >= ROWS * U) return p;
        const int row = tid % ROWS, uu = tid / ROWS, vr = mg * ROWS + row, u = jb * U + uu, H = a.H;
        const bool beam = PRED && a.W > 1;
        if (PRED) {
            if (!a.no_carry && vr < a.M && !a.flag[vr]) {
                const int pr = beam ? beam_prow(a.parent, a.W, vr) : vr;
                p.carry = true; p.carry_h = Ops::ld(a.h_in, hidx(a, pr, u));
                if (beam) { p.carry_c = a.c_in[(size_t)u * a.M + pr]; p.carry_y = Ops::ld(a.y_in, hidx(a, pr, u)); }
            }
            if (vr >= n_act) return p;
            p.r = row_map[vr];
            p.act = true;
        } else {
            p.r = vr;
            p.act = a.t < a.flag[vr];
            if (!p.act) { p.h_old = Ops::ld(a.h_in, hidx(a, vr, u)); return p; }
        }
        if (TABLE) {
            const float* tb = a.tab + (size_t)a.token[p.r] * 4 * H + u;
            p.x[0] = tb[0]; p.x[1] = tb[H]; p.x[2] = tb[2 * H]; p.x[3] = tb[3 * H];
        } else {
            p.x[0] = a.bias[u]; p.x[1] = a.bias[H + u]; p.x[2] = a.bias[2 * H + u]; p.x[3] = a.bias[3 * H + u];
        }
        p.c_old = beam ? a.c_in[(size_t)u * a.M + beam_prow(a.parent, a.W, p.r)] : a.c[(size_t)u * a.M + p.r];
        p.s = a.bn_s[u]; p.t = a.bn_t[u];
        return p;
    }
    template <int MTB, class Red>
    __device__ static __forceinline__ void run(const Args& a, const Red red, int tid, int jb, int mg, int n_act, const int* row_map,
                               const Pre& p, int) {
        constexpr int ROWS = MTB * 16;
        if (tid >= ROWS * U) return;
        const int row = tid % ROWS, uu = tid / ROWS, vr = mg * ROWS + row, u = jb * U + uu;
        if (PRED && p.carry) {
            Ops::st(a.h_out, hidx(a, vr, u), p.carry_h);
            if (a.W > 1) { a.c[(size_t)u * a.M + vr] = p.carry_c; Ops::st(a.y, hidx(a, vr, u), p.carry_y); }
        }
        if (p.r < 0) return;
        const size_t ho = hidx(a, p.r, u);
        if (!p.act) {
            Ops::st(a.h_out, ho, p.h_old);
            return;
        }
        const float gi = red.sum(row, 0 * U + uu) + p.x[0], gf = red.sum(row, 1 * U + uu) + p.x[1];
        const float gg = red.sum(row, 2 * U + uu) + p.x[2], go = red.sum(row, 3 * U + uu) + p.x[3];
        const float c2 = sigmoid_(gf) * p.c_old + sigmoid_(gi) * tanhf(gg);
        const float h2 = sigmoid_(go) * tanhf(c2);
        a.c[(size_t)u * a.M + p.r] = c2;
        Ops::st(a.h_out, ho, h2);
        if (PRED) Ops::st(a.y, ho, h2 * p.s + p.t);
        else if (a.y) Ops::st(a.y, Ops::aoff(p.r + 16 * a.y_mt_off, u, a.y_mt_total), h2 * p.s + p.t);
    }
};

// ---- encoder LSTM cell, tiling "D": U = 12 units x 4 gates = 48 columns (3 n-tiles) x 64 rows (4 m-tiles) per workgroup, 8 waves.
// For hidden sizes that are multiples of 12 and >= 128 rows (configs[4]: 1536 units, 128 streams): H / 12 x M / 64 = 256
// workgroups, exactly one per CU, 7 fragment loads per 12 MFMAs (tiling C: 4 per 4, three 32 x 32 workgroups per CU): the
// operand bytes a CU pulls through L2 per launch drop from 1.18 MB to 0.69 MB.  64 x 12 = 768 (row, unit) items per workgroup:
// an item loop, operands loaded in the epilogue.  Same arithmetic per item as EpiLSTM<enc>.
template <class Ops, int U_>
struct EpiLSTMe {
    static constexpr int U = U_, NT = U_ / 4;
    static constexpr int PH0_TILES = (1 << NT) - 1, PH1_TILES = (1 << NT) - 1;
    static constexpr int PH0_DEAD = -1, PH1_DEAD = -1;
    static constexpr bool COMPACT = false;
    using Args = typename EpiLSTM<Ops, false, false, 8>::Args;
    struct Pre {};
    __device__ static bool tile_active(const Args& a, int mt, int lane) { return (a.tile_mask >> mt) & 1ull; }
    template <int MTB>
    __device__ static __forceinline__ Pre prefetch(const Args&, int, int, int, int, const int*) { return Pre{}; }
    template <int MTB, class Red>
    __device__ static __forceinline__ void run(const Args& a, const Red red, int tid, int jb, int mg, int, const int*, const Pre&, int nthr) {
        constexpr int ROWS = MTB * 16;
        const int H = a.H;
        for (int item = tid; item < ROWS * U; item += nthr) {
            const int row = item % ROWS, uu = item / ROWS, vr = mg * ROWS + row, u = jb * U + uu;
            const size_t ho = Ops::aoff(vr, u, a.MT);
            if (!(a.t < a.flag[vr])) {                   // the row does not advance: carried to the other parity
                Ops::st(a.h_out, ho, Ops::ld(a.h_in, ho));
                continue;
            }
            const float x[4] = {a.bias[u], a.bias[H + u], a.bias[2 * H + u], a.bias[3 * H + u]};
            const float gi = red.sum(row, 0 * U + uu) + x[0], gf = red.sum(row, 1 * U + uu) + x[1];
            const float gg = red.sum(row, 2 * U + uu) + x[2], go = red.sum(row, 3 * U + uu) + x[3];
            const float c2 = sigmoid_(gf) * a.c[(size_t)u * a.M + vr] + sigmoid_(gi) * tanhf(gg);
            const float h2 = sigmoid_(go) * tanhf(c2);
            a.c[(size_t)u * a.M + vr] = c2;
            Ops::st(a.h_out, ho, h2);
            if (a.y) Ops::st(a.y, Ops::aoff(vr + 16 * a.y_mt_off, u, a.y_mt_total), h2 * a.bn_s[u] + a.bn_t[u]);
        }
    }
};

// ---- NBRC / GRU-v1 cell (haste/nbrc.py:30-64; layout z,r,g), predictor only (COMPACT, row-major):
//   z = s(Wx_z + Rh_z), r = s(Wx_r + Rh_r), g = tanh(Wx_g + r * Rh_g), h' = z h + (1 - z) g.
// Pseudo-gates {z, r, gx, gh} in one 16-column tile (4 units): the x phase feeds z,r,gx (gh columns
// dead), the h phase z,r,gh (gx dead).  TABLE: Wx (+ input bias) comes from tab[token][3H] and the x
// phase is absent.
template <class Ops, bool TABLE>
struct EpiNBRC {
    static constexpr int U = 4;
    static constexpr int NT = 1;
    static constexpr int PH0_TILES = TABLE ? 0 : 1;
    static constexpr int PH1_TILES = 1;
    static constexpr int PH0_DEAD = 12;   // gh columns carry no x weights
    static constexpr int PH1_DEAD = 8;    // gx columns carry no h weights
    static constexpr bool COMPACT = true;
    struct Args {
        const float* bias;     // [3H] input bias (folded into tab when TABLE)
        const float* rbias;    // [3H] recurrent bias
        const float* tab;      // [V][3H]
        const int* token;
        const int* emit;
        const void* h_in;      // [M][H] current parity
        void* h_out;           // other parity
        void* y;               // BN(h') [M][H]
        const float* bn_s;
        const float* bn_t;
        int H, M;
        const int* parent;     // beam search (W > 1): see EpiLSTM::Args
        int W;
        const void* y_in;
        int no_carry;
    };
    struct Pre {
        int r;
        bool carry;
        float carry_h, carry_y, h, xz, xr, xg, rz, rr, rg, s, t;
    };
    template <int MTB>
    __device__ static __forceinline__ Pre prefetch(const Args& a, int tid, int jb, int mg, int n_act, const int* row_map) {
        constexpr int ROWS = MTB * 16;
        static_assert(ROWS * U == 256, "one (row, unit) item per thread");
        Pre p;
        p.r = -1; p.carry = false;
        if (tid >= 256) return p;
        const int row = tid % ROWS, uu = tid / ROWS, vr = mg * ROWS + row, u = jb * U + uu, H = a.H;
        const bool beam = a.W > 1;
        if (!a.no_carry && vr < a.M && !a.emit[vr]) {
            const int pr = beam ? beam_prow(a.parent, a.W, vr) : vr;
            p.carry = true; p.carry_h = Ops::ld(a.h_in, (size_t)pr * H + u);
            if (beam) p.carry_y = Ops::ld(a.y_in, (size_t)pr * H + u);
        }
        if (vr >= n_act) return p;
        p.r = row_map[vr];
        p.h = Ops::ld(a.h_in, (size_t)(beam ? beam_prow(a.parent, a.W, p.r) : p.r) * H + u);
        if (TABLE) {
            const float* tb = a.tab + (size_t)a.token[p.r] * 3 * H + u;
            p.xz = tb[0]; p.xr = tb[H]; p.xg = tb[2 * H];
        } else {
            p.xz = a.bias[u]; p.xr = a.bias[H + u]; p.xg = a.bias[2 * H + u];
        }
        p.rz = a.rbias[u]; p.rr = a.rbias[H + u]; p.rg = a.rbias[2 * H + u];
        p.s = a.bn_s[u]; p.t = a.bn_t[u];
        return p;
    }
    template <int MTB, class Red>
    __device__ static __forceinline__ void run(const Args& a, const Red red, int tid, int jb, int mg, int n_act, const int* row_map,
                               const Pre& p, int) {
        constexpr int ROWS = MTB * 16;
        if (tid >= 256) return;
        const int row = tid % ROWS, uu = tid / ROWS, vr = mg * ROWS + row, u = jb * U + uu, H = a.H;
        if (p.carry) {
            Ops::st(a.h_out, (size_t)vr * H + u, p.carry_h);
            if (a.W > 1) Ops::st(a.y, (size_t)vr * H + u, p.carry_y);
        }
        if (p.r < 0) return;
        const size_t ho = (size_t)p.r * H + u;
        const float vz = red.sum(row, 0 * U + uu), vr_ = red.sum(row, 1 * U + uu), vgh = red.sum(row, 3 * U + uu);
        const float xg = TABLE ? p.xg : red.sum(row, 2 * U + uu) + p.xg;
        const float z = sigmoid_(vz + p.xz + p.rz);
        const float rr = sigmoid_(vr_ + p.xr + p.rr);
        const float gc = tanhf(xg + rr * (vgh + p.rg));
        const float h2 = z * p.h + (1.0f - z) * gc;
        Ops::st(a.h_out, ho, h2);
        Ops::st(a.y, ho, h2 * p.s + p.t);
    }
};

// ---- wide predictor tilings (many decoder rows: beam search, > 256 streams): 16 units x all gates x 64 rows per workgroup
// (NT = 4 of the 4-unit tiles of tiling A: same packed weights).  With hundreds of rows the A operand, not the weights, is what
// moves: a 4-unit workgroup reads its 64 rows' whole [x, h] for 4 units' worth of outputs, a 16-unit one for 16 (1/4 of the
// L2 -> CU traffic: configs[4] at beam 8 has 1024 rows x 3072 x 2 B per row group and 384 n-groups).  Column of (pseudo-gate
// pg, unit uu of 16) in the 64-column workgroup tile: (uu / 4) * 16 + pg * 4 + uu % 4.  Item loop instead of one item per
// thread, operands loaded in the epilogue (at these sizes the K loop is tens of microseconds).  Same arithmetic per item as
// EpiLSTM<PRED> / EpiNBRC.
template <class Ops, bool TABLE, int NTW = 4>
struct EpiLSTMw {
    static constexpr int U = 4 * NTW, NT = NTW;
    static constexpr int PH0_TILES = TABLE ? 0 : (1 << NTW) - 1, PH1_TILES = (1 << NTW) - 1;
    static constexpr int PH0_DEAD = -1, PH1_DEAD = -1;
    static constexpr bool COMPACT = true;
    using Args = typename EpiLSTM<Ops, true, TABLE, 4>::Args;
    struct Pre {};
    template <int MTB>
    __device__ static __forceinline__ Pre prefetch(const Args&, int, int, int, int, const int*) { return Pre{}; }
    template <int MTB, class Red>
    __device__ static __forceinline__ void run(const Args& a, const Red red, int tid, int jb, int mg, int n_act, const int* row_map,
                                               const Pre&, int nthr) {
        constexpr int ROWS = MTB * 16;
        const int H = a.H;
        const bool beam = a.W > 1;
        for (int item = tid; item < ROWS * U; item += nthr) {
            const int row = item % ROWS, uu = item / ROWS, vr = mg * ROWS + row, u = jb * U + uu;
            if (!a.no_carry && vr < a.M && !a.flag[vr]) {          // a row that did not emit: carried to the other parity
                const int pr = beam ? beam_prow(a.parent, a.W, vr) : vr;
                Ops::st(a.h_out, (size_t)vr * H + u, Ops::ld(a.h_in, (size_t)pr * H + u));
                if (beam) {
                    a.c[(size_t)u * a.M + vr] = a.c_in[(size_t)u * a.M + pr];
                    Ops::st(a.y, (size_t)vr * H + u, Ops::ld(a.y_in, (size_t)pr * H + u));
                }
            }
            if (vr >= n_act) continue;
            const int r = row_map[vr];
            float x[4];
            if (TABLE) {
                const float* tb = a.tab + (size_t)a.token[r] * 4 * H + u;
                x[0] = tb[0]; x[1] = tb[H]; x[2] = tb[2 * H]; x[3] = tb[3 * H];
            } else {
                x[0] = a.bias[u]; x[1] = a.bias[H + u]; x[2] = a.bias[2 * H + u]; x[3] = a.bias[3 * H + u];
            }
            const float c_old = beam ? a.c_in[(size_t)u * a.M + beam_prow(a.parent, a.W, r)] : a.c[(size_t)u * a.M + r];
            const int cb = (uu >> 2) * 16 + (uu & 3);
            const float gi = red.sum(row, cb) + x[0], gf = red.sum(row, cb + 4) + x[1];
            const float gg = red.sum(row, cb + 8) + x[2], go = red.sum(row, cb + 12) + x[3];
            const float c2 = sigmoid_(gf) * c_old + sigmoid_(gi) * tanhf(gg);
            const float h2 = sigmoid_(go) * tanhf(c2);
            a.c[(size_t)u * a.M + r] = c2;
            Ops::st(a.h_out, (size_t)r * H + u, h2);
            Ops::st(a.y, (size_t)r * H + u, h2 * a.bn_s[u] + a.bn_t[u]);
        }
    }
};

template <class Ops, bool TABLE, int NTW = 4>
struct EpiNBRCw {
    static constexpr int U = 4 * NTW, NT = NTW;
    static constexpr int PH0_TILES = TABLE ? 0 : (1 << NTW) - 1, PH1_TILES = (1 << NTW) - 1;
    static constexpr int PH0_DEAD = 12, PH1_DEAD = 8;
    static constexpr bool COMPACT = true;
    using Args = typename EpiNBRC<Ops, TABLE>::Args;
    struct Pre {};
    template <int MTB>
    __device__ static __forceinline__ Pre prefetch(const Args&, int, int, int, int, const int*) { return Pre{}; }
    template <int MTB, class Red>
    __device__ static __forceinline__ void run(const Args& a, const Red red, int tid, int jb, int mg, int n_act, const int* row_map,
                                               const Pre&, int nthr) {
        constexpr int ROWS = MTB * 16;
        const int H = a.H;
        const bool beam = a.W > 1;
        for (int item = tid; item < ROWS * U; item += nthr) {
            const int row = item % ROWS, uu = item / ROWS, vr = mg * ROWS + row, u = jb * U + uu;
            if (!a.no_carry && vr < a.M && !a.emit[vr]) {
                const int pr = beam ? beam_prow(a.parent, a.W, vr) : vr;
                Ops::st(a.h_out, (size_t)vr * H + u, Ops::ld(a.h_in, (size_t)pr * H + u));
                if (beam) Ops::st(a.y, (size_t)vr * H + u, Ops::ld(a.y_in, (size_t)pr * H + u));
            }
            if (vr >= n_act) continue;
            const int r = row_map[vr];
            const float h = Ops::ld(a.h_in, (size_t)(beam ? beam_prow(a.parent, a.W, r) : r) * H + u);
            float xz, xr, xg;
            if (TABLE) {
                const float* tb = a.tab + (size_t)a.token[r] * 3 * H + u;
                xz = tb[0]; xr = tb[H]; xg = tb[2 * H];
            } else {
                xz = a.bias[u]; xr = a.bias[H + u]; xg = a.bias[2 * H + u];
            }
            const int cb = (uu >> 2) * 16 + (uu & 3);
            const float vz = red.sum(row, cb), vr_ = red.sum(row, cb + 4), vgh = red.sum(row, cb + 12);
            if (!TABLE) xg = red.sum(row, cb + 8) + xg;
            const float z = sigmoid_(vz + xz + a.rbias[u]);
            const float rr = sigmoid_(vr_ + xr + a.rbias[H + u]);
            const float gc = tanhf(xg + rr * (vgh + a.rbias[2 * H + u]));
            const float h2 = z * h + (1.0f - z) * gc;
            Ops::st(a.h_out, (size_t)r * H + u, h2);
            Ops::st(a.y, (size_t)r * H + u, h2 * a.bn_s[u] + a.bn_t[u]);
        }
    }
};

// ---- plain linear: out[row][col] = acc + bias[col], row-major f32.  NTW n-tiles (16 * NTW columns) per workgroup: 1 for the
// skinny problems (every weight byte fetched once per m-group), 4 for hundreds of rows (beam search: 1024 hypothesis rows x
// 1536 -- with 16-column workgroups the ACTIVATIONS are what moves, 128 n-groups x 3 MB = 400 MB through L2 per launch).
template <int NTW>
struct EpiLinearT {
    static constexpr int NT = NTW;
    static constexpr int PH0_TILES = (1 << NTW) - 1, PH1_TILES = 0, PH0_DEAD = -1, PH1_DEAD = -1;
    static constexpr bool COMPACT = false;
    struct Args {
        const float* bias;    // may be nullptr
        float* out;
        int ldo;
        int n_rows;           // rows >= n_rows are not written
        const int* t_idx;     // optional row gate: row active iff t_idx[r % M] < T_row[r % M]
        const int* T_row;
        int M;
        const int* ring_base; // optional: GEMM row (t*M + r) is written to row ((ring_base[r] + t) % ring)*M + r
        int ring;
        int W;                // beam search: row r belongs to stream r / W (t_idx, T_row are per stream)
        const float* row_scale;   // optional (int8-served LM): out = acc * (row_scale[r] * w_scale) + bias
        float w_scale;
    };
    __device__ static bool row_on(const Args& a, int r) {
        if (r >= a.n_rows) return false;
        if (!a.t_idx) return true;
        if (a.W > 1) return a.t_idx[r / a.W] < a.T_row[r / a.W];
        const int q = r % a.M;                 // lookahead: rows [k*M, (k+1)*M) evaluate frame t + k
        return a.t_idx[q] + r / a.M < a.T_row[q];
    }
    __device__ static bool tile_active(const Args& a, int mt, int lane) {
        return any16(lane < 16 && row_on(a, mt * 16 + (lane & 15)), lane);
    }
    struct Pre {};
    template <int MTB>
    __device__ static __forceinline__ Pre prefetch(const Args&, int, int, int, int, const int*) { return Pre{}; }
    template <int MTB, class Red>
    __device__ static __forceinline__ void run(const Args& a, const Red red, int tid, int jb, int mg, int, const int*, const Pre&,
                               int nthr) {
        constexpr int ROWS = MTB * 16, CW = 16 * NTW;
        for (int it = tid; it < ROWS * CW; it += nthr) {
            const int col = it % CW, row = it / CW;        // consecutive threads -> consecutive columns
            const int r = mg * ROWS + row;
            if (!row_on(a, r)) continue;
            const int n = jb * CW + col;
            size_t orow = (size_t)r;
            if (a.ring_base) {
                const int t = r / a.M, q = r - t * a.M;
                orow = (size_t)((a.ring_base[q] + t) % a.ring) * a.M + q;
            }
            float v = red.sum(row, col);
            if (a.row_scale) v = __fmul_rn(v, __fmul_rn(a.row_scale[r], a.w_scale));     // dequantise, then the float bias (no FMA)
            a.out[orow * a.ldo + n] = __fadd_rn(v, a.bias ? a.bias[n] : 0.f);
        }
    }
};

using EpiLinear = EpiLinearT<1>;

// ---- predictor half of the joint + fused joint activation (COMPACT over emitting rows):
//   pp[r] = h_pred[r] W1p^T + b1                 for rows that emitted
//   ja[r] = tanh(pe[t_idx[r]][r] + pp[r])        for every row still decoding (fragment-major:
//                                                it is the A operand of the logits GEMM)
// Joint.forward 'concat' (models.py:132-140): Linear(cat(pred, enc)) == W1p pred + W1e enc + b1.
// Workgroup (jb, mg) refreshes ja for its compacted (emitting) rows and for the NON-emitting rows
// of the original row range [mg*ROWS, (mg+1)*ROWS).
template <class Ops, int NTW = 1>
struct EpiPPJ {
    static constexpr int NT = NTW;
    static constexpr int PH0_TILES = (1 << NTW) - 1, PH1_TILES = 0, PH0_DEAD = -1, PH1_DEAD = -1;
    static constexpr bool COMPACT = true;
    struct Args {
        const float* b1;
        float* pp;            // [M][J] f32 (beam: the other parity, see pp_in)
        const float* pe;      // [ring][M_enc][J] f32
        const int* t_idx;     // per stream
        const int* T_row;
        const int* emit;
        void* ja;             // fragment-major [J/KCH][MT][64][16 B]
        int J, M, MT;
        int ring;             // pe holds frame t of row r at slot t % ring (ring >= frames of a step)
        // beam search (W > 1): rows are hypothesis slots, W per stream; a slot that was not extended
        // takes pp of its parent slot (current parity) into its own slot of the other parity
        const int* parent;
        int W, M_enc;
        const float* pp_in;
        int la;               // greedy lookahead: ja is also produced for frames t+1 .. t+la-1 (rows k*M + r)
        int no_carry;         // beam: pp / ja of the rows that are not extended are done by k_beam_carry
    };
    struct Pre {};
    template <int MTB>
    __device__ static __forceinline__ Pre prefetch(const Args&, int, int, int, int, const int*) { return Pre{}; }
    template <int MTB, class Red>
    __device__ static __forceinline__ void run(const Args& a, const Red red, int tid, int jb, int mg, int n_act, const int* row_map,
                               const Pre&, int nthr) {
        constexpr int ROWS = MTB * 16, CW = 16 * NTW;
        const bool beam = a.W > 1;
        for (int it = tid; it < ROWS * CW; it += nthr) {
            const int col = it % CW, row = it / CW;
            const int j = jb * CW + col;
            const int vr = mg * ROWS + row;
            if (vr < n_act) {                           // an emitting row compacted into this group
                const int r = row_map[vr];
                const float p = red.sum(row, col) + a.b1[j];
                a.pp[(size_t)r * a.J + j] = p;
                const int q = beam ? r / a.W : r;       // stream of the row
                const int t = a.t_idx[q], Tq = a.T_row[q];
                if (beam) {
                    if (t < Tq)
                        Ops::st(a.ja, Ops::aoff(r, j, a.MT), tanhf(a.pe[((size_t)(t % a.ring) * a.M_enc + q) * a.J + j] + p));
                } else {
                    for (int k = 0; k < a.la && t + k < Tq; ++k)
                        Ops::st(a.ja, Ops::aoff(k * a.M + r, j, a.MT),
                                tanhf(a.pe[((size_t)((t + k) % a.ring) * a.M + r) * a.J + j] + p));
                }
            }
            const int r = vr;                           // original row of this range, if it did not emit
            if (!a.no_carry && r < a.M && !a.emit[r]) {
                const int q = beam ? r / a.W : r;
                const int t = a.t_idx[q];
                float p;
                if (beam) {
                    p = a.pp_in[(size_t)beam_prow(a.parent, a.W, r) * a.J + j];
                    a.pp[(size_t)r * a.J + j] = p;
                    if (t < a.T_row[q])
                        Ops::st(a.ja, Ops::aoff(r, j, a.MT), tanhf(a.pe[((size_t)(t % a.ring) * a.M_enc + q) * a.J + j] + p));
                } else {
                    const int Tq = a.T_row[q];
                    if (t >= Tq) continue;
                    p = a.pp[(size_t)r * a.J + j];
                    for (int k = 0; k < a.la && t + k < Tq; ++k)
                        Ops::st(a.ja, Ops::aoff(k * a.M + r, j, a.MT),
                                tanhf(a.pe[((size_t)((t + k) % a.ring) * a.M + r) * a.J + j] + p));
                }
            }
        }
    }
};
