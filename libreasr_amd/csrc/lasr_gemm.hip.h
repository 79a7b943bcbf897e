// lasr_gemm.hip.h -- the skinny-GEMM core (v_mfma_f32_16x16x4_f32) and its fused epilogues.
// Included from lasr_kernels.hip.h inside namespace lasr (after frag_off / sigmoid_).
//
//   out[rows, N] = A[rows, K] * W[N, K]^T      rows = stream slots, W streamed from HBM
//
// A workgroup owns MT 16-row m-tiles x NT 16-column n-tiles and the whole K range, which it splits
// over its NW waves (wave w takes K chunks w, w+NW, ...).  Operands go global -> VGPR directly in
// MFMA-fragment order (1 KiB perfectly coalesced wave loads, see lasr_kernels.hip.h) through a
// 3-deep register ring whose loads are UNCONDITIONAL (chunk indices are clamped, never branched
// on) so that the compiler can count them and emit partial s_waitcnt vmcnt(N) instead of
// draining the queue before every MFMA block.  The NW partial tiles are reduced through LDS and a
// fused epilogue finishes the cell / projection for the rows x units the workgroup owns.
//
// Two tilings of the same math are used:
//   U = 16 ("B"): MT = 1, NT = gates : 16 hidden units x all gates x 16 rows per workgroup;
//                 grid (H/16, M/16).  Weights are re-read by the M/16 row groups (via L2).
//   U = 4  ("A"): MT = 4, NT = 1     : 4 hidden units x all gates x 64 rows per workgroup;
//                 grid (H/4, M/64).  Every weight byte is fetched exactly once chip-wide and all
//                 256 CUs pull on the weight stream even when a single m-tile is active (decode).
// COMPACT epilogues (predictor path) gather the rows whose flag is set into dense m-tiles inside
// the kernel (ballot prefix scan -> LDS row map), so MFMA work follows the number of emitting
// streams while the weight stream stays full width.

struct GemmArgs {
    const float* A[2];      // phase operand: fragment-major (row-major when AROW)
    const float* A_alt[2];  // optional second buffer: row r reads A_alt when a_sel[r] != 0
    const int* a_sel[2];    // per-row buffer selector (predictor h ping-pong), may be nullptr
    int a_mt_total[2];      // m-tiles in A's fragment layout (lda when AROW)
    int a_mt_off[2];        // m-tile index of row 0 inside A
    int KC[2];              // K chunks (of 16) per phase; 0 = phase absent
    const float* W[2];      // packed weights: [n_group][slot][KC][fragment]
    int a_rows;             // AROW only: loads of rows >= a_rows are clamped (0 = no clamp)
    const int* compact;     // COMPACT epilogues: per-row flag
    int M;                  // rows scanned for compaction
};

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
    f32x4 a[MT];
    f32x4 b[NS > 0 ? NS : 1];
};

// DEAD >= 0: columns [DEAD, DEAD+4) of the 16-column tile carry no weights in this phase; the
// fragment stores only the 12 live columns (768 B) and the dead lanes feed zeros to the MFMA.
template <int TILES, int DEAD, int MT, int NT, int NW>
__device__ __forceinline__ void gemm_phase(f32x4 (&acc)[MT][NT], const float* const (&aptr)[MT], size_t a_step,
                                           const bool (&tile_on)[MT], const float* __restrict__ Wp, int KC, int jb,
                                           int w, int lane) {
    constexpr int NS = PopCount<TILES>::value;
    if constexpr (NS == 0) {
        return;
    } else {
        if (KC <= 0) return;
        constexpr int FR = DEAD >= 0 ? 192 : 256;
        int loff = lane * 4;
        bool live = true;
        if constexpr (DEAD >= 0) {
            const int col = lane & 15, gq = lane >> 4;
            live = !(col >= DEAD && col < DEAD + 4);
            const int ai = col < DEAD ? col : col - 4;
            loff = (gq * 12 + ai) * 4;
        }
        const float* wb = Wp + (size_t)jb * NS * KC * FR + loff;
        const int n = (KC - w + NW - 1) / NW;          // chunks of this wave (<= 0: none)
        auto load = [&](Frag<MT, NS>& f, int i) {
            int c = w + i * NW;
            c = c < KC ? c : KC - 1;                   // clamp: the load stays valid and countable
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                if (MT == 1 || tile_on[mt]) f.a[mt] = *reinterpret_cast<const f32x4*>(aptr[mt] + (size_t)c * a_step);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (DEAD < 0 || live)
                    f.b[s] = *reinterpret_cast<const f32x4*>(wb + ((size_t)s * KC + c) * FR);
                else
                    f.b[s] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        auto compute = [&](const Frag<MT, NS>& f) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if (MT > 1 && !tile_on[mt]) continue;
                    int s = 0;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if ((TILES >> nt) & 1) {
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[mt][e], f.b[s][e], acc[mt][nt], 0, 0, 0);
                            ++s;
                        }
                    }
                }
            }
        };
        Frag<MT, NS> f0, f1, f2;
        load(f0, 0);
        load(f1, 1);
        for (int i = 0; i < n; i += 3) {
            load(f2, i + 2);
            compute(f0);
            load(f0, i + 3);
            if (i + 1 < n) compute(f1);
            load(f1, i + 4);
            if (i + 2 < n) compute(f2);
        }
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
template <class Epi, int MT, int NW, bool AROW>
__global__ __launch_bounds__(NW * 64) void k_gemm(const GemmArgs g, const typename Epi::Args ea) {
    constexpr int NT = Epi::NT, ROWS = MT * 16, LD = NT * 16 + 1;
    __shared__ float red[NW * ROWS * LD];
    __shared__ int row_map[Epi::COMPACT ? 1024 : 1];
    __shared__ int n_act_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int jb = blockIdx.x, mg = blockIdx.y;

    int n_act = g.M;
    if constexpr (Epi::COMPACT) {
        if (w == 0) {   // ballot prefix scan over the row flags (M <= 1024)
            int cnt = 0;
            for (int base = 0; base < g.M; base += 64) {
                const int r = base + lane;
                const bool f = r < g.M && g.compact[r] != 0;
                const unsigned long long m = __ballot(f);
                if (f) row_map[cnt + __popcll(m & ((1ull << lane) - 1ull))] = r;
                cnt += __popcll(m);
            }
            if (lane == 0) n_act_s = cnt;
        }
        __syncthreads();
        n_act = n_act_s;
        if (mg * ROWS >= n_act && !Epi::RUN_ALWAYS) return;
    }

    bool tile_on[MT];
    const float* ap0[MT];
    const float* ap1[MT];
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
        if constexpr (AROW) {
            int rr = orow;
            if (g.a_rows > 0 && rr >= g.a_rows) rr = g.a_rows - 1;
            ap0[mt] = g.A[0] + (size_t)rr * g.a_mt_total[0] + (lane >> 4) * 4;
            ap1[mt] = g.A[1] ? g.A[1] + (size_t)rr * g.a_mt_total[1] + (lane >> 4) * 4 : nullptr;
        } else {
            const size_t in_tile = ((size_t)(lane >> 4) * 16 + (orow & 15)) * 4;
            const float* b0 = (g.a_sel[0] && g.a_sel[0][orow]) ? g.A_alt[0] : g.A[0];
            const float* b1 = (g.a_sel[1] && g.a_sel[1][orow]) ? g.A_alt[1] : g.A[1];
            ap0[mt] = b0 + (size_t)(g.a_mt_off[0] + (orow >> 4)) * 256 + in_tile;
            ap1[mt] = b1 + (size_t)(g.a_mt_off[1] + (orow >> 4)) * 256 + in_tile;
        }
    }
    if constexpr (AROW) {
        a_step0 = 16; a_step1 = 16;
    } else {
        a_step0 = (size_t)g.a_mt_total[0] * 256; a_step1 = (size_t)g.a_mt_total[1] * 256;
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    bool any_on = false;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) any_on = any_on || tile_on[mt];
    if (any_on) {   // wave-uniform: no weight is streamed for a workgroup without active rows
        gemm_phase<Epi::PH0_TILES, Epi::PH0_DEAD, MT, NT, NW>(acc, ap0, a_step0, tile_on, g.W[0], g.KC[0], jb, w, lane);
        gemm_phase<Epi::PH1_TILES, Epi::PH1_DEAD, MT, NT, NW>(acc, ap1, a_step1, tile_on, g.W[1], g.KC[1], jb, w, lane);
    }

#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[(w * ROWS + mt * 16 + 4 * (lane >> 4) + r) * LD + nt * 16 + (lane & 15)] = acc[mt][nt][r];
    __syncthreads();
    Epi::template run<MT>(ea, RedView<NW, ROWS, LD>{red}, tid, jb, mg, n_act, row_map);
}

// ------------------------------------------------------------------------------------------------
// Epilogues.  run<MT>() walks the (row, unit) items of the workgroup tile with a 256-thread stride;
// a tile column of (gate g, unit uu) is g*U + uu.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool any16(bool flag, int lane) {
    return (__ballot(flag && lane < 16) != 0ull);   // wave-uniform OR over lanes 0..15
}
__device__ __forceinline__ size_t hfrag(int r, int u, int MT_all) {
    return ((size_t)((u >> 4) * MT_all + (r >> 4)) * 64 + (((u >> 2) & 3) * 16 + (r & 15))) * 4 + (u & 3);
}

// ---- LSTM cell (torch gate order i,f,g,o; custom_rnn.py:172, haste/lstm.py:34-68) + BN(eval) fold.
// ENC: row r is active at step t iff t < T_row[r]; inactive rows carry h to the other parity buffer.
// PRED (COMPACT): only emitting rows are touched; h ping-pong is per row (hsel); phase X is the
// per-token table tab[token][4H] when TABLE.
template <bool PRED, bool TABLE, int U>
struct EpiLSTM {
    static constexpr int NT = U == 16 ? 4 : 1;
    static constexpr int PH0_TILES = TABLE ? 0 : (U == 16 ? 0xF : 1);
    static constexpr int PH1_TILES = U == 16 ? 0xF : 1;
    static constexpr int PH0_DEAD = -1, PH1_DEAD = -1;
    static constexpr bool COMPACT = PRED;
    static constexpr bool RUN_ALWAYS = false;
    struct Args {
        const float* bias;     // [4H] = b_ih + b_hh (folded into tab when TABLE)
        const float* tab;      // [V][4H]
        const int* token;      // [M]
        const int* flag;       // ENC: T_row[M]
        int t;                 // ENC: time step
        float* c;              // [H][M] cell state, in place
        const float* h_in;     // ENC: fragment-major current parity
        float* h_out;          // ENC: other parity
        float* hbuf[2];        // PRED: both buffers; row r reads hbuf[hsel[r]], writes hbuf[hsel[r]^1]
        const int* hsel;       // PRED
        float* y;              // BN(h') fragment-major; may be nullptr
        int y_mt_total, y_mt_off;
        const float* bn_s;
        const float* bn_t;
        int H, M, MT;
    };
    __device__ static bool tile_active(const Args& a, int mt, int lane) {
        return any16(lane < 16 && a.t < a.flag[mt * 16 + (lane & 15)], lane);
    }
    template <int MTB, class Red>
    __device__ static void run(const Args& a, const Red red, int tid, int jb, int mg, int n_act, const int* row_map) {
        constexpr int ROWS = MTB * 16;
        const int H = a.H;
        for (int it = tid; it < ROWS * U; it += 256) {
            const int row = it % ROWS, uu = it / ROWS;
            const int vr = mg * ROWS + row;
            const int u = jb * U + uu;
            int r = vr;
            float* hout = a.h_out;
            if (PRED) {
                if (vr >= n_act) continue;
                r = row_map[vr];
                hout = a.hbuf[a.hsel[r] ^ 1];
            }
            const size_t ho = hfrag(r, u, a.MT);
            if (!PRED && !(a.t < a.flag[r])) {
                hout[ho] = a.h_in[ho];
                continue;
            }
            float gi = red.sum(row, 0 * U + uu), gf = red.sum(row, 1 * U + uu);
            float gg = red.sum(row, 2 * U + uu), go = red.sum(row, 3 * U + uu);
            if (TABLE) {
                const float* tb = a.tab + (size_t)a.token[r] * 4 * H + u;
                gi += tb[0]; gf += tb[H]; gg += tb[2 * H]; go += tb[3 * H];
            } else {
                gi += a.bias[u]; gf += a.bias[H + u]; gg += a.bias[2 * H + u]; go += a.bias[3 * H + u];
            }
            const size_t co = (size_t)u * a.M + r;
            const float c2 = sigmoid_(gf) * a.c[co] + sigmoid_(gi) * tanhf(gg);
            const float h2 = sigmoid_(go) * tanhf(c2);
            a.c[co] = c2;
            hout[ho] = h2;
            if (a.y) a.y[hfrag(r + 16 * a.y_mt_off, u, a.y_mt_total)] = h2 * a.bn_s[u] + a.bn_t[u];
        }
    }
};

// ---- NBRC / GRU-v1 cell (haste/nbrc.py:30-64; layout z,r,g), predictor only (COMPACT):
//   z = s(Wx_z + Rh_z), r = s(Wx_r + Rh_r), g = tanh(Wx_g + r * Rh_g), h' = z h + (1 - z) g.
// Pseudo-gates {z, r, gx, gh}: the x phase feeds z,r,gx, the h phase z,r,gh.  TABLE: Wx (+ input
// bias) comes from tab[token][3H] and the x phase is absent.
template <bool TABLE, int U>
struct EpiNBRC {
    static constexpr int NT = U == 16 ? 4 : 1;
    static constexpr int PH0_TILES = TABLE ? 0 : (U == 16 ? 0x7 : 1);
    static constexpr int PH1_TILES = U == 16 ? 0xB : 1;
    static constexpr int PH0_DEAD = U == 16 ? -1 : 12;   // U=4: gh columns carry no x weights
    static constexpr int PH1_DEAD = U == 16 ? -1 : 8;    //      gx columns carry no h weights
    static constexpr bool COMPACT = true;
    static constexpr bool RUN_ALWAYS = false;
    struct Args {
        const float* bias;     // [3H] input bias (folded into tab when TABLE)
        const float* rbias;    // [3H] recurrent bias
        const float* tab;      // [V][3H]
        const int* token;
        float* hbuf[2];
        const int* hsel;
        float* y;              // BN(h') fragment-major [H/16][MT][64][4]
        const float* bn_s;
        const float* bn_t;
        int H, MT;
    };
    template <int MTB, class Red>
    __device__ static void run(const Args& a, const Red red, int tid, int jb, int mg, int n_act, const int* row_map) {
        constexpr int ROWS = MTB * 16;
        const int H = a.H;
        for (int it = tid; it < ROWS * U; it += 256) {
            const int row = it % ROWS, uu = it / ROWS;
            const int vr = mg * ROWS + row;
            if (vr >= n_act) continue;
            const int r = row_map[vr], u = jb * U + uu;
            const int sel = a.hsel[r];
            const size_t ho = hfrag(r, u, a.MT);
            const float h = a.hbuf[sel][ho];
            const float vz = red.sum(row, 0 * U + uu), vr_ = red.sum(row, 1 * U + uu);
            const float vgh = red.sum(row, 3 * U + uu);
            float xz, xr, xg;
            if (TABLE) {
                const float* tb = a.tab + (size_t)a.token[r] * 3 * H + u;
                xz = tb[0]; xr = tb[H]; xg = tb[2 * H];
            } else {
                xz = a.bias[u]; xr = a.bias[H + u]; xg = red.sum(row, 2 * U + uu) + a.bias[2 * H + u];
            }
            const float z = sigmoid_(vz + xz + a.rbias[u]);
            const float rr = sigmoid_(vr_ + xr + a.rbias[H + u]);
            const float gc = tanhf(xg + rr * (vgh + a.rbias[2 * H + u]));
            const float h2 = z * h + (1.0f - z) * gc;
            a.hbuf[sel ^ 1][ho] = h2;
            a.y[ho] = h2 * a.bn_s[u] + a.bn_t[u];
        }
    }
};

// ---- plain linear: out[row][col] = acc + bias[col], row-major.
struct EpiLinear {
    static constexpr int NT = 1;
    static constexpr int PH0_TILES = 1, PH1_TILES = 0, PH0_DEAD = -1, PH1_DEAD = -1;
    static constexpr bool COMPACT = false;
    static constexpr bool RUN_ALWAYS = false;
    struct Args {
        const float* bias;    // may be nullptr
        float* out;
        int ldo;
        int n_rows;           // rows >= n_rows are not written
        const int* t_idx;     // optional row gate: row active iff t_idx[r % M] < T_row[r % M]
        const int* T_row;
        int M;
    };
    __device__ static bool row_on(const Args& a, int r) {
        if (r >= a.n_rows) return false;
        if (!a.t_idx) return true;
        const int q = r % a.M;
        return a.t_idx[q] < a.T_row[q];
    }
    __device__ static bool tile_active(const Args& a, int mt, int lane) {
        return any16(lane < 16 && row_on(a, mt * 16 + (lane & 15)), lane);
    }
    template <int MTB, class Red>
    __device__ static void run(const Args& a, const Red red, int tid, int jb, int mg, int, const int*) {
        constexpr int ROWS = MTB * 16;
        for (int it = tid; it < ROWS * 16; it += 256) {
            const int col = it & 15, row = it >> 4;        // consecutive threads -> consecutive columns
            const int r = mg * ROWS + row;
            if (!row_on(a, r)) continue;
            const int n = jb * 16 + col;
            a.out[(size_t)r * a.ldo + n] = red.sum(row, col) + (a.bias ? a.bias[n] : 0.f);
        }
    }
};

// ---- predictor half of the joint + fused joint activation (COMPACT over emitting rows):
//   pp[r] = h_pred[r] W1p^T + b1                 for rows that emitted
//   ja[r] = tanh(pe[t_idx[r]][r] + pp[r])        for every row still decoding (fragment-major:
//                                                it is the A operand of the logits GEMM)
// Joint.forward 'concat' (models.py:132-140): Linear(cat(pred, enc)) == W1p pred + W1e enc + b1.
// Workgroup (jb, mg) refreshes ja for its compacted (emitting) rows and for the NON-emitting rows
// of the original row range [mg*ROWS, (mg+1)*ROWS); workgroup (0,0) also flips the predictor
// ping-pong selector of the rows that just advanced.
struct EpiPPJ {
    static constexpr int NT = 1;
    static constexpr int PH0_TILES = 1, PH1_TILES = 0, PH0_DEAD = -1, PH1_DEAD = -1;
    static constexpr bool COMPACT = true;
    static constexpr bool RUN_ALWAYS = true;
    struct Args {
        const float* b1;
        float* pp;            // [M][J]
        const float* pe;      // [T][M][J]
        const int* t_idx;
        const int* T_row;
        const int* emit;
        int* hsel;
        float* ja;            // fragment-major [J/16][MT][64][4]
        int J, M, MT;
    };
    template <int MTB, class Red>
    __device__ static void run(const Args& a, const Red red, int tid, int jb, int mg, int n_act, const int* row_map) {
        constexpr int ROWS = MTB * 16;
        for (int it = tid; it < ROWS * 16; it += 256) {
            const int col = it & 15, row = it >> 4;
            const int j = jb * 16 + col;
            const int vr = mg * ROWS + row;
            if (vr < n_act) {                           // an emitting row compacted into this group
                const int r = row_map[vr];
                const float p = red.sum(row, col) + a.b1[j];
                a.pp[(size_t)r * a.J + j] = p;
                const int t = a.t_idx[r];
                if (t < a.T_row[r]) a.ja[hfrag(r, j, a.MT)] = tanhf(a.pe[((size_t)t * a.M + r) * a.J + j] + p);
            }
            const int r = vr;                           // original row of this range, if it did not emit
            if (r < a.M && !a.emit[r]) {
                const int t = a.t_idx[r];
                if (t < a.T_row[r])
                    a.ja[hfrag(r, j, a.MT)] = tanhf(a.pe[((size_t)t * a.M + r) * a.J + j] + a.pp[(size_t)r * a.J + j]);
            }
        }
        if (jb == 0 && mg == 0)
            for (int q = tid; q < n_act; q += 256) a.hsel[row_map[q]] ^= 1;
    }
};
