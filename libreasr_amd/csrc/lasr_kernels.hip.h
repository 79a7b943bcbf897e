// lasr_kernels.hip.h -- gfx950 (MI355X / CDNA4) device code of the streaming RNN-T path.
//
// Data layout ("fragment-major", the layout every hot matmul operand lives in)
// ---------------------------------------------------------------------------
// All dense contractions here are skinny:  out[rows, N] = A[rows, K] * W[N, K]^T  with
// rows = stream slots (<= a few hundred) and the weights dominating the bytes.  They run on the
// exact-f32 matrix instruction v_mfma_f32_16x16x4_f32 (64 lanes; lane l feeds A[i=l&15][k=l>>4]
// and B[k=l>>4][j=l&15]; C/D: row = 4*(l>>4)+reg, col = l&15).
//
// A 16x16 (rows x k) block of an operand is stored as one contiguous 1 KiB "fragment":
//     frag[lane = g*16 + i][e]  =  X[16*tile + i][16*c + 4*g + e],   g,e in 0..3, i in 0..15
// so that one 16-byte load per lane (a perfectly coalesced 1 KiB wave load) delivers the A (or B)
// operands of FOUR consecutive MFMAs (MFMA e consumes element e; the k index it contracts is
// 16c + 4g + e for both operands, so the products line up; only the summation order differs
// from a k-sequential dot product).
//   activations [rows, K]:  off(r,k) = ((c*mt_total + r/16)*64 + g*16 + r%16)*4 + e
//   weights: tile-major, [n_tile][c][lane][e]  (a tile's K-panel is one contiguous stream)
//
// The GEMM kernel (k_gemm, lasr_gemm.hip.h) gives each workgroup MT m-tiles x NT n-tiles, splits K
// over its NW waves (register prefetch ring, no LDS staging), reduces the NW partial tiles through
// LDS and runs a fused epilogue (LSTM / NBRC cell math + BatchNorm(eval) fold, joint projections,
// tanh).  With dtype = bf16 the same layout holds with 32-k chunks of 8 x bf16 per lane
// (v_mfma_f32_16x16x32_bf16); "bf" flags below select the element type of activation buffers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace lasr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// precise activations (no fast-math: token-for-token parity after hundreds of recurrent steps)
__device__ __forceinline__ float sigmoid_(float x) { return 1.0f / (1.0f + expf(-x)); }

#include "lasr_gemm.hip.h"

// ------------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------------
// joint activation for all rows (start of a step): ja = tanh(pe[t_idx] + pp).
// W > 1 (beam search): rows are hypothesis slots, W per stream; t_idx / T_row / pe are per stream (M_enc rows)
inline __global__ void k_ja(const float* __restrict__ pe, const float* __restrict__ pp, const int* __restrict__ t_idx,
                     const int* __restrict__ T_row, void* __restrict__ ja, int J, int M, int MT, int ring, int bf,
                     int W, int M_enc, int la) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * J) return;
    const int r = idx / J, j = idx - r * J;
    const int q = W > 1 ? r / W : r;
    const int t = t_idx ? t_idx[q] : 0;
    const float p = pp[(size_t)r * J + j];
    for (int k = 0; k < la; ++k) {           // greedy lookahead: frames t .. t+la-1 (rows k*M + r); la = 1 otherwise
        if (T_row && t + k >= T_row[q]) return;
        act_st(bf, ja, act_off(bf, k * M + r, j, MT), tanhf(pe[((size_t)((t + k) % ring) * M_enc + q) * J + j] + p));
    }
}

// continuous greedy loop, admission of newly encoded steps (<= 512 rows): the rows' new "frames available" counts come BY
// VALUE (the host knows every row's cumulative frame count), are stored for the kernels behind this one and used here for
// the joint activation of the rows that were idle -- one launch instead of a counter-advance kernel per admitted step + k_ja
struct AvailV { int v[512]; };
inline __global__ void k_ja_admit(const float* __restrict__ pe, const float* __restrict__ pp, const int* __restrict__ t_idx,
                           const AvailV av, int* __restrict__ avail_out, void* __restrict__ ja, int J, int M, int MT, int ring,
                           int bf, int la) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * J) return;
    const int r = idx / J, j = idx - r * J;
    const int Tav = av.v[r];
    if (j == 0) avail_out[r] = Tav;
    const int t = t_idx[r];
    const float p = pp[(size_t)r * J + j];
    for (int k = 0; k < la; ++k) {
        if (t + k >= Tav) return;
        act_st(bf, ja, act_off(bf, k * M + r, j, MT), tanhf(pe[((size_t)((t + k) % ring) * M + r) * J + j] + p));
    }
}

// fragment-major -> row-major [rows][K] f32
inline __global__ void k_from_frag(const void* __restrict__ src, int mt_total, int mt_off, float* __restrict__ dst, int ldd,
                            int rows, int K, int bf) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * K) return;
    const int r = idx / K, k = idx - r * K;
    dst[(size_t)r * ldd + k] = act_ld(bf, src, act_off(bf, r + 16 * mt_off, k, mt_total));
}
// flat f32 <-> element-typed copies (op-level entry points in bf16 mode)
inline __global__ void k_to_elem(const float* __restrict__ src, void* __restrict__ dst, size_t n, int bf) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) act_st(bf, dst, i, src[i]);
}
inline __global__ void k_from_elem(const void* __restrict__ src, float* __restrict__ dst, size_t n, int bf) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = act_ld(bf, src, i);
}
// [H][M] -> [rows][H]
inline __global__ void k_c_to_rows(const float* __restrict__ c, int M, float* __restrict__ dst, int rows, int H) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * H) return;
    const int r = idx / H, u = idx - r * H;
    dst[idx] = c[(size_t)u * M + r];
}

// LASR_DBG_ENCLOG: out[e][r] = sum over k of the bit pattern of element (r, k) of source e (mod 2^32: exact, order-free).
// kind 0: fragment-major activations (element type by bf), kind 1: unit-major f32 state c[k][M], kind 2: row-major f32 [M][K].
// grid (M, entries), 256 threads.
struct RowSumSrc { const void* p; int kind, mt_total, mt_off, K; };
struct RowSumArgs { RowSumSrc s[32]; };
inline __global__ void k_dbg_rowsum(const RowSumArgs a, int M, int bf, unsigned* __restrict__ out) {
    __shared__ unsigned acc;
    const RowSumSrc s = a.s[blockIdx.y];
    const int r = blockIdx.x;
    if (threadIdx.x == 0) acc = 0u;
    __syncthreads();
    unsigned v = 0u;
    for (int k = threadIdx.x; k < s.K; k += blockDim.x) {
        if (s.kind == 1) v += __float_as_uint(((const float*)s.p)[(size_t)k * M + r]);
        else if (s.kind == 2) v += __float_as_uint(((const float*)s.p)[(size_t)r * s.K + k]);
        else if (bf) v += ((const unsigned short*)s.p)[OpsBF16::aoff(r + 16 * s.mt_off, k, s.mt_total)];
        else v += __float_as_uint(((const float*)s.p)[OpsF32::aoff(r + 16 * s.mt_off, k, s.mt_total)]);
    }
    atomicAdd(&acc, v);
    __syncthreads();
    if (threadIdx.x == 0) out[(size_t)blockIdx.y * M + r] = acc;
}

// encoder output of the last layer: fragment-major rows (t*M + b) -> out[b][t][H]
inline __global__ void k_enc_out(const void* __restrict__ y, int mt_total, int M, float* __restrict__ out, int B, int T, int H, int bf) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * T * H) return;
    const int u = (int)(idx % H);
    const int t = (int)((idx / H) % T);
    const int b = (int)(idx / ((size_t)H * T));
    out[idx] = act_ld(bf, y, act_off(bf, t * M + b, u, mt_total));
}

// deterministic pseudo-random fill in [-1, 1) (micro-benchmark operands)
inline __global__ void k_fill_rand(void* __restrict__ p, size_t n, unsigned seed, int bf) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)i * 2654435761u ^ (seed * 40503u + 0x9e3779b9u);
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    act_st(bf, p, i, (float)(x >> 8) * (2.0f / 16777216.0f) - 1.0f);
}

// (re)initialise recurrent state of flagged rows from the learned initial states
// (custom_rnn.py:152-158; Transducer.transcribe_stream reset(), models.py:480-500).
struct ResetArgs {
    const int* what;          // [M] bit 1: encoder, bit 2: predictor
    int mask;                 // bits of `what` this launch honours
    int M, MT, H, Le, Lp, pred_lstm, bos, bf;
    int W, Md;                // beam search: predictor rows are hypothesis slots (W per stream, Md = M * W rows);
    double* score;            //   a reset stream restarts with slot 0 = the BOS hypothesis (score 0), others dead
    int* alive;
    int* inB;
    int* parent;
    void* enc_h[16];          // current-parity fragment buffers (element-typed)
    float* enc_c[16];
    const float* enc_h0[16];  // [H]
    const float* enc_c0[16];
    void* pred_h[8];          // current-parity buffers, row-major [M][H] (element-typed)
    float* pred_c[8];
    const float* pred_h0[8];
    const float* pred_c0[8];
    int* token;
    int* emit;
    // greedy decode: the predictor state AFTER its BOS step and the joint's predictor half for it (models.py:489) are constants of
    // the model -- captured once at lasr_create (k_bos_capture) and stored here instead of running the predictor on BOS per reset
    const float* bos_h[8];    // [H] per layer (null: the caller runs the BOS pass)
    const float* bos_c[8];
    const float* bos_pp;      // [J]
    float* pp;                // [M][J]
    int J;
};
struct BosArgs {
    const void* pred_h[8]; const float* pred_c[8]; const float* pp;
    float* bos_h[8]; float* bos_c[8]; float* bos_pp;
    int H, J, Lp, M, lstm, bf;
};
// row 0 of the predictor state / pp (just refreshed by a BOS pass) -> the constants of ResetArgs
inline __global__ void k_bos_capture(const BosArgs a) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u < a.H)
        for (int l = 0; l < a.Lp; ++l) {
            a.bos_h[l][u] = act_ld(a.bf, a.pred_h[l], (size_t)u);
            if (a.lstm) a.bos_c[l][u] = a.pred_c[l][(size_t)u * a.M];
        }
    if (u < a.J) a.bos_pp[u] = a.pp[u];
}
inline __global__ void k_reset_rows(const ResetArgs a) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.M * a.H) return;
    const int r = idx / a.H, u = idx - r * a.H;
    const int wh = a.what[r] & a.mask;
    const int W = a.W > 1 ? a.W : 1;
    const int rp = r * W;                                  // predictor row (slot 0 of the stream)
    const bool bos = a.bos_pp != nullptr && W == 1;
    if (u == 0 && (a.mask & 2)) {
        a.emit[rp] = ((wh & 2) && !bos) ? 1 : 0;
        if (wh & 2) a.token[rp] = a.bos;
        if (W > 1)
            for (int b = 0; b < W; ++b) {
                a.parent[rp + b] = b;
                if (b) a.emit[rp + b] = 0;
                if (wh & 2) { a.score[rp + b] = b ? -INFINITY : 0.0; a.alive[rp + b] = b ? 0 : 1; a.inB[rp + b] = 0; }
            }
    }
    const size_t ho = act_off(a.bf, r, u, a.MT);
    if (wh & 1)
        for (int l = 0; l < a.Le; ++l) {
            act_st(a.bf, a.enc_h[l], ho, a.enc_h0[l][u]);
            a.enc_c[l][(size_t)u * a.M + r] = a.enc_c0[l][u];
        }
    if (wh & 2) {
        for (int l = 0; l < a.Lp; ++l) {
            act_st(a.bf, a.pred_h[l], (size_t)rp * a.H + u, bos ? a.bos_h[l][u] : a.pred_h0[l][u]);
            if (a.pred_lstm) a.pred_c[l][(size_t)u * (W > 1 ? a.Md : a.M) + rp] = bos ? a.bos_c[l][u] : a.pred_c0[l][u];
        }
        if (bos)
            for (int j = u; j < a.J; j += a.H) a.pp[(size_t)rp * a.J + j] = a.bos_pp[j];
    }
}

// per-step decode state
struct DecState {
    int* t_idx;        // [M] current encoder frame of the row inside this step
    int* iters;        // [M] joint evaluations done on the current frame
    int* token;        // [M] last emitted token (predictor input)
    int* emit;         // [M] 1 iff the row emitted a token in the last select
    int* step_ntok;    // [M] tokens emitted in this step
    int* step_tok;     // [M][tok_cap]
    int tok_cap;
    double* logp_sum;  // [M] sum of log p of every decision (models.py:420-422)
    int* sum_iters;    // [M] evaluations in this step
    int* n_ones;       // [M] frames finished after exactly one evaluation (alignment_score)
    int* unfinished;   // [n_iter_slots] rows still decoding after iteration i
    // continuous mode (decode loop running across chunk boundaries, lasr_step_submit/_wait):
    //   t_idx = global frame cursor, T_row = frames available, step_ntok = tokens emitted so far,
    //   step_tok = token ring of tok_cap entries per row
    int cont;
    int* host_cur;     // [M] pinned host memory: the row's frame cursor, stored by the last iteration of a group (the host
                       //     derives "step j of row r is decoded" for EVERY step in flight from it: no per-step target upload)
    int* host_ntot;    // [M] pinned (diagnostics, may be nullptr): tokens emitted so far, stored with host_cur
    int* ntok_end;     // [M][end_slots] tokens emitted when row r finished its step j (slot j % end_slots)
    int step_T;        // frames per model step (n_buffer)
    int end_slots;
    // continuous mode, last iteration of a group: the last workgroup to finish publishes the "rows still
    // behind" count straight into pinned host memory (no copy kernel between the group and the host)
    int* done_blocks;  // [64] ring, like unfinished
    int* host_flag;    // nullptr: nothing to publish in this launch
    int* iter_ctr;     // continuous mode: device-side iteration counter (the flag-ring slot is iter_ctr & 63), so
                       // a group of iterations is launch-invariant and can be replayed as a hipGraph
    // LM shallow fusion (LMFuser.fuse, lm.py:59-79): standardised LM log-probs of the row's last token
    const float* lmz;  // [M][V] (nullptr: no LM attached)
    const int* lm_valid; // [M] the LM has advanced at least once since the last LM reset
    float lm_alpha, lm_theta, lm_min;
    unsigned long long* dbg;   // LASR_DBG_TIMING: phase stamps of k_select's workgroup 0 (wall clock, 10 ns ticks)
};

// block-wide sum over 256 threads (4 waves); every thread gets the result
__device__ __forceinline__ float block_sum_256(float x, float* sh4, int lane, int w) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    __syncthreads();                                 // sh4 may still be read from a previous use
    if (lane == 0) sh4[w] = x;
    __syncthreads();
    return sh4[0] + sh4[1] + sh4[2] + sh4[3];
}

// lasr_overlap_probe (and round 3's stream-sensitivity probes): one wave holds its stream for `ticks` of the 100 MHz wall clock without
// touching memory -- the marginal cost of a microsecond on either stream of the pipelined protocol
inline __global__ void k_delay(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// experiment (lasr_bench_neighbour): a synthetic neighbour for the two-stream job, one wave per workgroup, running for `ticks` of the
// 100 MHz wall clock on a stream of its own.  k_nb_mfma issues f32 MFMAs back to back from registers (no memory traffic: it takes
// matrix-pipe cycles of the SIMD it sits on and nothing else); k_nb_load streams a large buffer with 8 x 1 KB loads in flight per
// wave (no MFMA: it takes L2 / fabric / HBM bandwidth).  Each writes its iteration count to done[blockIdx.x].
inline __global__ __launch_bounds__(64) void k_nb_mfma(unsigned long long ticks, unsigned long long* __restrict__ done, float* __restrict__ sink) {
    f32x4 acc0{0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    const float a = 1e-3f * (float)threadIdx.x, b = 1.0f - 1e-4f * (float)threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    unsigned long long n = 0;
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {          // 64 MFMAs (16x16x4 f32: 2048 flop each) per check of the clock
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, acc3, 0, 0, 0);
        }
        n += 64;
    }
    if (threadIdx.x == 0) done[blockIdx.x] = n;
    if (acc0[0] + acc1[1] + acc2[2] + acc3[3] == 12345.678f) sink[0] = acc0[0];      // (keeps the MFMAs alive)
}
// region_vec > 0: every wave walks its own region of that many 16-byte vectors again and again with ordinary loads (96 KB per wave:
// larger than the CU's L1, and 32 CUs x 96 KB stay inside an XCD's 4 MB L2 -- the L2 -> CU path and nothing behind it); region_vec
// == 0: all waves stream the whole buffer (temporal != 0: ordinary loads -- a 128 MB buffer then lives in the Infinity Cache;
// temporal == 0: non-temporal loads of a 512 MB buffer -- HBM).
inline __global__ __launch_bounds__(64) void k_nb_load(unsigned long long ticks, const f32x4* __restrict__ buf, size_t n_vec, size_t region_vec,
                                                int temporal, unsigned long long* __restrict__ done, float* __restrict__ sink) {
    const size_t span = region_vec ? region_vec : n_vec;
    const size_t base = region_vec ? (size_t)blockIdx.x * region_vec : 0;
    const size_t stride = region_vec ? 64 * 8 : (size_t)gridDim.x * 64 * 8;
    size_t idx = region_vec ? threadIdx.x : ((size_t)blockIdx.x * 8) * 64 + threadIdx.x;
    f32x4 s{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = wall_clock64();
    unsigned long long n = 0;
    while (wall_clock64() - t0 < ticks) {
        f32x4 v[8];
        if (temporal) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = buf[base + (idx + (size_t)i * 64) % span];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_nontemporal_load(buf + base + (idx + (size_t)i * 64) % span);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
        idx = (idx + stride) % span;
        n += 8;                                   // 8 KB per wave and iteration
    }
    if (threadIdx.x == 0) done[blockIdx.x] = n;
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) sink[0] = s[0];
}

// ---- wave-wide reductions on DPP row operations + v_readlane instead of ds_bpermute butterflies (a __shfl_xor step is an LDS
// crossbar round trip of ~90 cycles per dword; k_beam_select spent 17 of its 20 us in such steps).  Lane pairing: quad_perm for
// xor 1 / xor 2, row_half_mirror / row_mirror for the 8- and 16-lane steps (after the quad steps all lanes of a quad hold the same
// value, so mirroring pairs the same operands as xor 4 / xor 8), then the four row results combined as (r0 + r1) + (r2 + r3):
// the same association as the xor butterfly -- sums are bit-identical to it.  Every lane returns the wave's result.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float readlane_f32(float x, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane)); }
__device__ __forceinline__ float wave_sum_f32(float x) {
    x += dpp_f32<0xB1>(x); x += dpp_f32<0x4E>(x); x += dpp_f32<0x141>(x); x += dpp_f32<0x140>(x);
    return (readlane_f32(x, 0) + readlane_f32(x, 16)) + (readlane_f32(x, 32) + readlane_f32(x, 48));
}
__device__ __forceinline__ float wave_max_f32(float x) {
    x = fmaxf(x, dpp_f32<0xB1>(x)); x = fmaxf(x, dpp_f32<0x4E>(x)); x = fmaxf(x, dpp_f32<0x141>(x)); x = fmaxf(x, dpp_f32<0x140>(x));
    return fmaxf(fmaxf(readlane_f32(x, 0), readlane_f32(x, 16)), fmaxf(readlane_f32(x, 32), readlane_f32(x, 48)));
}
// argmax under the total order (value descending, ord ascending)
template <int CTRL>
__device__ __forceinline__ void dpp_argmax_step(double& best, int& ord) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(best), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(best), CTRL, 0xf, 0xf, false);
    const int oo = __builtin_amdgcn_update_dpp(0, ord, CTRL, 0xf, 0xf, false);
    const double ob = __hiloint2double(hi, lo);
    if (ob > best || (ob == best && oo < ord)) { best = ob; ord = oo; }
}
__device__ __forceinline__ void wave_argmax_f64(double& best, int& ord) {
    dpp_argmax_step<0xB1>(best, ord); dpp_argmax_step<0x4E>(best, ord); dpp_argmax_step<0x141>(best, ord); dpp_argmax_step<0x140>(best, ord);
    const int lo = __double2loint(best), hi = __double2hiint(best);
    double b = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    int o = __builtin_amdgcn_readlane(ord, 0);
#pragma unroll
    for (int r = 16; r < 64; r += 16) {
        const double ob = __hiloint2double(__builtin_amdgcn_readlane(hi, r), __builtin_amdgcn_readlane(lo, r));
        const int oo = __builtin_amdgcn_readlane(ord, r);
        if (ob > b || (ob == b && oo < o)) { b = ob; o = oo; }
    }
    best = b; ord = o;
}

// argmax of (float value, int index): larger value first, smaller index among equals
template <int CTRL>
__device__ __forceinline__ void dpp_argmax32_step(float& best, int& arg) {
    const float ob = dpp_f32<CTRL>(best);
    const int oa = __builtin_amdgcn_update_dpp(0, arg, CTRL, 0xf, 0xf, false);
    if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
}
__device__ __forceinline__ void wave_argmax_f32(float& best, int& arg) {
    dpp_argmax32_step<0xB1>(best, arg); dpp_argmax32_step<0x4E>(best, arg); dpp_argmax32_step<0x141>(best, arg); dpp_argmax32_step<0x140>(best, arg);
    float b = readlane_f32(best, 0);
    int a = __builtin_amdgcn_readlane(arg, 0);
#pragma unroll
    for (int r = 16; r < 64; r += 16) {
        const float ob = readlane_f32(best, r);
        const int oa = __builtin_amdgcn_readlane(arg, r);
        if (ob > b || (ob == b && oa < a)) { b = ob; a = oa; }
    }
    best = b; arg = a;
}

// Synchronous protocol, end of a group of decode iterations: the step's results so far (token counts + tokens) go to the
// pinned result block by zero-copy stores and the "rows still decoding" word is published LAST, by a system-scope release
// store of the last workgroup to arrive -- the host spins on that word.  (Two hipMemcpyAsync calls "payload, then flag" are
// NOT such a protocol: HIP orders the copies on the stream, not their visibility to a host that has not synchronised; a
// fresh flag over a stale payload was observed at a rate of 3e-3 per stream, tests/soak.py.)
inline __global__ __launch_bounds__(256) void k_publish(const int* __restrict__ src, int* __restrict__ dst_host, int n,
                                                 const int* __restrict__ flag_src, int* __restrict__ flag_host, int* __restrict__ arrivals) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst_host[i] = src[i];
    __threadfence_system();                          // this thread's stores are visible system-wide ...
    __syncthreads();                                 // ... and so are the whole workgroup's
    if (threadIdx.x == 0) {
        if (atomicAdd(arrivals, 1) == (int)gridDim.x - 1) {
            atomicExch(arrivals, 0);                 // the next launch is stream-ordered behind this one
            __hip_atomic_store(flag_host, *flag_src, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// frames of newly encoded steps become visible to the decode loop (continuous mode)
inline __global__ void k_advance(int* __restrict__ counter, const int* __restrict__ add, int M) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) counter[i] += add[i];
}

inline __global__ void k_step_begin(DecState s, int M, int n_iter_slots, int reset_metrics) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) {
        s.t_idx[i] = 0;
        s.iters[i] = 0;
        s.emit[i] = 0;
        s.step_ntok[i] = 0;
        if (reset_metrics) {
            s.logp_sum[i] = 0.0;
            s.sum_iters[i] = 0;
            s.n_ones[i] = 0;
        }
    }
    if (i < n_iter_slots) s.unfinished[i] = 0;
}

// log-softmax + argmax over the vocabulary and the greedy state machine of
// Transducer.decode_greedy / transcribe_stream (models.py:405-443, 530-571), one workgroup per row.
// PLAIN: only (argmax, log p) are produced (op-level joint entry point).
//
// Lookahead (la > 1): the iteration evaluated frames t, t+1, .. t+la-1 of the row against the SAME
// predictor state (logits rows r, M + r, 2M + r, ..).  As long as the decision is blank the predictor
// state does not change, so the next frame's evaluation is exactly what the reference would compute
// next: the row consumes the whole run of blanks and, if it comes, the first token after it, in one
// iteration.  The decisions, their order and every metric are those of the one-frame-per-iteration
// loop; only the number of launches per frame changes.  (With an LM attached la = 1.)
// LAT: compile-time bound of `la` (1, 2 or 4).  The logits of ALL lookahead frames are loaded up front (their addresses depend
// on nothing this kernel loads) and their statistics are reduced together: one load round trip and two block reductions per
// launch whatever la is; the decisions are then replayed in order from the per-frame (argmax, log p) by every thread (no
// barrier in the state machine).  Same decisions, same order, same arithmetic per frame as the one-frame loop.
// KEEP: logits kept in registers per thread and frame (8 covers V <= 2048, 16 V <= 4096; the rest of a larger row is re-read).  A
// slot past V holds -inf and contributes an exact zero, so the results do not depend on KEEP.
template <bool PLAIN, int LAT, int KEEP>
__global__ __launch_bounds__(256) void k_select(const float* __restrict__ logits, int V, int blank, int max_iters,
                                                const int* __restrict__ T_row, DecState s, int iter_slot_in,
                                                float* __restrict__ out_logp, int* __restrict__ out_arg, int la, int M) {
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool dbgt = !PLAIN && s.dbg && r == 0 && tid == 0;
    const unsigned long long t_entry = dbgt ? wall_clock64() : 0ull;
    float zv[LAT][KEEP];
#pragma unroll
    for (int k = 0; k < LAT; ++k) {
        const float* z = logits + (size_t)((k < la ? k : 0) * M + r) * V;
#pragma unroll
        for (int q = 0; q < KEEP; ++q) {
            const int j = tid + 256 * q;
            zv[k][q] = j < V ? z[j] : -INFINITY;
        }
    }
    const int iter_no = (!PLAIN && s.cont) ? (int)(*(const unsigned*)s.iter_ctr & 0x3fffffffu) : iter_slot_in;   // same value in every workgroup of the launch
    const int iter_slot = (!PLAIN && s.cont) ? (iter_no & 63) : iter_slot_in;
    if (!PLAIN && s.cont && r == 0 && tid == 0) {                                           // recycle the flag rings
        s.unfinished[(iter_slot + 32) & 63] = 0;
        s.done_blocks[(iter_slot + 32) & 63] = 0;
    }
    auto publish = [&]() {           // thread 0 of every workgroup, after its last store of this launch
        if (PLAIN || !s.cont) return;
        if (s.host_flag) __threadfence_system();     // this row's tokens / marks (pinned memory) before the count
        if (atomicAdd(&s.done_blocks[iter_slot], 1) == (int)gridDim.x - 1) {      // last workgroup of the launch
            if (s.host_flag) {
                const int v = atomicAdd(&s.unfinished[iter_slot], 0);
                __hip_atomic_store(s.host_flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            *s.iter_ctr = iter_no + 1;
        }
    };
    // state of the row (in flight together with the logits)
    int t = 0, Tr = 1, it_cur = 0, n0 = 0, si0 = 0, no0 = 0;
    double lp0 = 0.0;
    if (!PLAIN) {
        t = s.t_idx[r]; Tr = T_row[r];
        if (t >= Tr) {
            if (tid == 0) {
                s.emit[r] = 0;
                if (s.cont && s.host_flag) {                      // progress made in earlier iterations of this group
                    s.host_cur[r] = t;
                    if (s.host_ntot) s.host_ntot[r] = s.step_ntok[r];
                }
                publish();
            }
            return;
        }
        it_cur = s.iters[r];                                      // every thread replays the state machine
        if (tid == 0) { n0 = s.step_ntok[r]; si0 = s.sum_iters[r]; no0 = s.n_ones[r]; lp0 = s.logp_sum[r]; }
    }
    if (dbgt) { s.dbg[0] = t_entry; s.dbg[1] = wall_clock64(); }  // state loaded (stamps of a launch in which row 0 decodes)
    const int nla = PLAIN ? 1 : min(min(la, LAT), Tr - t);       // frames this launch may decide (uniform over the workgroup)
    __shared__ float sv[LAT][4];
    __shared__ int si[LAT][4];
    __shared__ float ss[LAT][4];
    float best[LAT], logp[LAT], sums[LAT];
    int arg[LAT];
    // ---- argmax of every frame (ascending j per thread: first max wins), all frames through ONE barrier
#pragma unroll
    for (int k = 0; k < LAT; ++k) {
        best[k] = -INFINITY; arg[k] = 0x7fffffff; logp[k] = 0.f;
        if (k >= nla) continue;
        const float* z = logits + (size_t)(k * M + r) * V;
#pragma unroll
        for (int q = 0; q < KEEP; ++q)
            if (zv[k][q] > best[k]) { best[k] = zv[k][q]; arg[k] = tid + 256 * q; }
        for (int j = tid + 256 * KEEP; j < V; j += 256) {
            const float x = z[j];
            if (x > best[k]) { best[k] = x; arg[k] = j; }
        }
        wave_argmax_f32(best[k], arg[k]);
        if (lane == 0) { sv[k][w] = best[k]; si[k][w] = arg[k]; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < LAT; ++k) {
        if (k >= nla) continue;
        best[k] = sv[k][0]; arg[k] = si[k][0];
#pragma unroll
        for (int q = 1; q < 4; ++q)
            if (sv[k][q] > best[k] || (sv[k][q] == best[k] && si[k][q] < arg[k])) { best[k] = sv[k][q]; arg[k] = si[k][q]; }
        const float* z = logits + (size_t)(k * M + r) * V;
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < KEEP; ++q) sum += expf(zv[k][q] - best[k]);      // exp(-inf) = 0 for the padding
        for (int j = tid + 256 * KEEP; j < V; j += 256) sum += expf(z[j] - best[k]);
        sum = wave_sum_f32(sum);                     // same association as the xor butterfly: bit-identical
        if (lane == 0) ss[k][w] = sum;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < LAT; ++k) {
        if (k >= nla) continue;
        const float sum = ss[k][0] + ss[k][1] + ss[k][2] + ss[k][3];
        sums[k] = sum;
        logp[k] = -logf(sum);                // log_softmax at the argmax = z_max - logsumexp
    }
    if (PLAIN) {
        if (tid == 0) { out_logp[r] = logp[0]; out_arg[r] = arg[0]; }
        return;
    }
    if (dbgt) s.dbg[2] = wall_clock64();                          // statistics of all frames done
    // The frame that emits in this launch, if any: the first non-blank decision among the frames the row may decide (blank frames in
    // front of it are consumed with the predictor AND the LM state unchanged, so the re-pick below sees exactly the state the
    // one-frame loop would have at that decision: lookahead and LM fusion compose).
    int ke = -1;
#pragma unroll
    for (int k = 0; k < LAT; ++k)
        if (ke < 0 && k < nla && arg[k] != blank) ke = k;
    int tok_e = 0;
#pragma unroll
    for (int k = 0; k < LAT; ++k)
        if (k == ke) tok_e = arg[k];
    if (s.lmz && ke >= 0 && s.lm_valid[r]) {
        // LMFuser.fuse (lm.py:59-79), only for a non-blank decision (models.py:427-431; uniform over
        // the workgroup): standardise the joint log-softmax (utils.py:162-164: subtract the mean,
        // divide by the unbiased std + 1e-5), entry 0 = MIN_VAL, re-pick argmax(alpha*lm + theta*joint).
        // V <= 4096 (checked at lasr_attach_lm): every value of the row is in zv[].
        __shared__ float sh4[4];
        float zl[KEEP];                              // the emitting frame's logits
        float bestk = 0.f, sumk = 1.f;
#pragma unroll
        for (int k = 0; k < LAT; ++k)
            if (k == ke) {
                bestk = best[k]; sumk = sums[k];
#pragma unroll
                for (int q = 0; q < KEEP; ++q) zl[q] = zv[k][q];
            }
        const float lse = logf(sumk);
        float part = 0.f;
#pragma unroll
        for (int q = 0; q < KEEP; ++q) {
            const int j = tid + 256 * q;
            zl[q] = j < V ? (zl[q] - bestk) - lse : 0.f;        // log-softmax; padding contributes nothing
            part += zl[q];
        }
        const float mean = block_sum_256(part, sh4, lane, w) / (float)V;
        part = 0.f;
#pragma unroll
        for (int q = 0; q < KEEP; ++q) {
            const int j = tid + 256 * q;
            zl[q] = j < V ? zl[q] - mean : 0.f;                // t.add_(-t.mean())
            part += zl[q];
        }
        const float mean2 = block_sum_256(part, sh4, lane, w) / (float)V;   // torch.std subtracts its own mean again
        part = 0.f;
#pragma unroll
        for (int q = 0; q < KEEP; ++q) {
            const int j = tid + 256 * q;
            const float d = j < V ? zl[q] - mean2 : 0.f;
            part += d * d;
        }
        const float sd = sqrtf(block_sum_256(part, sh4, lane, w) / (float)(V - 1));
        const float den = sd + 1e-5f;
        const float* lz = s.lmz + (size_t)r * V;
        float fb = -INFINITY;
        int fa = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < KEEP; ++q) {
            const int j = tid + 256 * q;
            if (j >= V) continue;
            const float jo = j == 0 ? s.lm_min : zl[q] / den;
            const float f = __fadd_rn(__fmul_rn(s.lm_alpha, lz[j]), __fmul_rn(s.lm_theta, jo));   // no FMA contraction
            if (f > fb) { fb = f; fa = j; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(fb, o);
            const int oa = __shfl_xor(fa, o);
            if (ob > fb || (ob == fb && oa < fa)) { fb = ob; fa = oa; }
        }
        __syncthreads();
        if (lane == 0) { sv[0][w] = fb; si[0][w] = fa; }
        __syncthreads();
        fb = sv[0][0]; fa = si[0][0];
#pragma unroll
        for (int q = 1; q < 4; ++q)
            if (sv[0][q] > fb || (sv[0][q] == fb && si[0][q] < fa)) { fb = sv[0][q]; fa = si[0][q]; }
        tok_e = fa;                          // the emitted token; log p stays the unfused one (models.py:422)
    }
    // ---- the decisions, in frame order (models.py:405-443, 530-571): a row consumes its run of blank frames and, if it
    // comes, the first token after it.  Every thread tracks t and the per-frame evaluation count; thread 0 owns the rest.
    int emitted = 0;
#pragma unroll
    for (int k = 0; k < LAT; ++k) {
        if (k >= nla || emitted) continue;
        const bool nonblank = arg[k] != blank;      // decided by the joint alone (models.py:424-431)
        int it = it_cur + 1;
        if (tid == 0) { lp0 += (double)logp[k]; si0 += 1; }
        bool frame_done = true;
        if (nonblank) {
            emitted = 1;
            if (tid == 0) {
                const int tok = tok_e;               // (k == ke: the first non-blank frame, re-picked above when an LM is attached)
                if (s.cont) s.step_tok[(size_t)r * s.tok_cap + (n0 % s.tok_cap)] = tok;
                else if (n0 < s.tok_cap) s.step_tok[(size_t)r * s.tok_cap + n0] = tok;
                n0 += 1;
                s.token[r] = tok;
            }
            frame_done = it >= max_iters;           // per-frame symbol cap
        }
        if (frame_done) {
            if (tid == 0 && it == 1) no0 += 1;
            it = 0;
            t += 1;
            if (tid == 0 && s.cont && t % s.step_T == 0)      // the row just finished one of its model steps
                s.ntok_end[(size_t)r * s.end_slots + ((t / s.step_T - 1) % s.end_slots)] = n0;
        }
        it_cur = it;
    }
    if (tid != 0) return;
    if (dbgt) s.dbg[3] = wall_clock64();                          // decisions replayed
    s.emit[r] = emitted;
    s.logp_sum[r] = lp0;
    s.sum_iters[r] = si0;
    s.n_ones[r] = no0;
    s.step_ntok[r] = n0;
    s.t_idx[r] = t;
    s.iters[r] = it_cur;
    if (s.cont && s.host_flag) {
        s.host_cur[r] = t;
        if (s.host_ntot) s.host_ntot[r] = n0;
    }
    if (t < Tr) atomicAdd(&s.unfinished[iter_slot], 1);     // continuous mode: rows that still have encoded frames to decode
    if (dbgt) s.dbg[4] = wall_clock64();                          // state stored
    publish();
    if (dbgt) s.dbg[5] = wall_clock64();
}
// launch with the compile-time lookahead bound that covers `la`
template <bool PLAIN>
inline void launch_select(hipStream_t st, int rows, const float* logits, int V, int blank, int max_iters, const int* T_row,
                          const DecState& s, int iter_slot, float* out_logp, int* out_arg, int la, int M) {
    if (V <= 2048 && !s.lmz) {      // (the LM re-pick keeps the whole row in registers: 16 slots)
        if (la <= 1) hipLaunchKernelGGL((k_select<PLAIN, 1, 8>), dim3(rows), dim3(256), 0, st, logits, V, blank, max_iters, T_row, s, iter_slot, out_logp, out_arg, 1, M);
        else if (la == 2) hipLaunchKernelGGL((k_select<PLAIN, 2, 8>), dim3(rows), dim3(256), 0, st, logits, V, blank, max_iters, T_row, s, iter_slot, out_logp, out_arg, la, M);
        else hipLaunchKernelGGL((k_select<PLAIN, 4, 8>), dim3(rows), dim3(256), 0, st, logits, V, blank, max_iters, T_row, s, iter_slot, out_logp, out_arg, la, M);
        return;
    }
    if (la <= 1) hipLaunchKernelGGL((k_select<PLAIN, 1, 16>), dim3(rows), dim3(256), 0, st, logits, V, blank, max_iters, T_row, s, iter_slot, out_logp, out_arg, 1, M);
    else if (la == 2) hipLaunchKernelGGL((k_select<PLAIN, 2, 16>), dim3(rows), dim3(256), 0, st, logits, V, blank, max_iters, T_row, s, iter_slot, out_logp, out_arg, la, M);
    else hipLaunchKernelGGL((k_select<PLAIN, 4, 16>), dim3(rows), dim3(256), 0, st, logits, V, blank, max_iters, T_row, s, iter_slot, out_logp, out_arg, la, M);
}

// LMFuser.advance (lm.py:49-53) for the rows that just emitted a token: log_softmax of the LM's output
// layer, standardise (utils.py:162-164), entry 0 = MIN_VAL.  One workgroup per row, V <= 4096.
// Beam search (W > 1): rows are hypothesis slots; a slot that was not extended takes the LM output of its PARENT slot from
// the other parity (lmz_in / valid_in), like the predictor state.
// KEEP: register slots per thread for the row (8 covers V <= 2048, 16 V <= 4096); a slot past V holds -inf / 0 and contributes an
// exact zero to every sum, so the results do not depend on KEEP (LASR_KEEP16=1: 16 slots whatever V, round 3's kernel)
template <int KEEP>
__global__ __launch_bounds__(256) void k_lm_post(const float* __restrict__ raw, const int* __restrict__ emit,
                                                 float* __restrict__ lmz, int* __restrict__ lm_valid, int V, float min_val,
                                                 const int* __restrict__ parent, int W, const float* __restrict__ lmz_in,
                                                 const int* __restrict__ valid_in) {
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (!emit[r]) {
        if (W > 1) {
            const int pr = (r / W) * W + parent[r];
            const float* src = lmz_in + (size_t)pr * V;
            float* dst = lmz + (size_t)r * V;
            for (int j = tid; j < V; j += 256) dst[j] = src[j];
            if (tid == 0) lm_valid[r] = valid_in[pr];
        }
        return;
    }
    __shared__ float sh4[4];
    const float* z = raw + (size_t)r * V;
    float zv[KEEP];
    float best = -INFINITY;
#pragma unroll
    for (int q = 0; q < KEEP; ++q) {
        const int j = tid + 256 * q;
        zv[q] = j < V ? z[j] : -INFINITY;
        best = fmaxf(best, zv[q]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o));
    if (lane == 0) sh4[w] = best;
    __syncthreads();
    best = fmaxf(fmaxf(sh4[0], sh4[1]), fmaxf(sh4[2], sh4[3]));
    float part = 0.f;
#pragma unroll
    for (int q = 0; q < KEEP; ++q) part += expf(zv[q] - best);
    const float lse = logf(block_sum_256(part, sh4, lane, w));
    part = 0.f;
#pragma unroll
    for (int q = 0; q < KEEP; ++q) {
        const int j = tid + 256 * q;
        zv[q] = j < V ? (zv[q] - best) - lse : 0.f;
        part += zv[q];
    }
    const float mean = block_sum_256(part, sh4, lane, w) / (float)V;
    part = 0.f;
#pragma unroll
    for (int q = 0; q < KEEP; ++q) {
        const int j = tid + 256 * q;
        zv[q] = j < V ? zv[q] - mean : 0.f;
        part += zv[q];
    }
    const float mean2 = block_sum_256(part, sh4, lane, w) / (float)V;
    part = 0.f;
#pragma unroll
    for (int q = 0; q < KEEP; ++q) {
        const int j = tid + 256 * q;
        const float d = j < V ? zv[q] - mean2 : 0.f;
        part += d * d;
    }
    const float den = sqrtf(block_sum_256(part, sh4, lane, w) / (float)(V - 1)) + 1e-5f;
    float* o = lmz + (size_t)r * V;
#pragma unroll
    for (int q = 0; q < KEEP; ++q) {
        const int j = tid + 256 * q;
        if (j < V) o[j] = j == 0 ? min_val : zv[q] / den;
    }
    if (tid == 0) lm_valid[r] = 1;
}

// LMFuser.reset (lm.py:81-83) for rows flagged with bit 4: no logits, zero LSTM state
struct LmResetArgs {
    const int* what;
    int M, H, L, bf;
    int W, Md;           // beam search: W hypothesis slots per stream (rows r W .. r W + W - 1 of Md)
    void* h[8];          // current parity, row-major [Md][H]
    float* c[8];         // [H][Md]
    int* lm_valid;
    // int8-served LM: the quantised image of every layer's h kept beside it (see k_lm_cell_q): zeros with scale 0.1 for h = 0
    unsigned short* qh[8]; float* sxh[8]; int Kp;
};
inline __global__ void k_lm_reset(const LmResetArgs a) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.M * a.H) return;
    const int r = idx / a.H, u = idx - r * a.H;
    if (!(a.what[r] & 4)) return;
    for (int b = 0; b < a.W; ++b) {
        const int rp = r * a.W + b;
        if (u == 0) a.lm_valid[rp] = 0;
        for (int l = 0; l < a.L; ++l) {
            act_st(a.bf, a.h[l], (size_t)rp * a.H + u, 0.f);
            a.c[l][(size_t)u * a.Md + rp] = 0.f;
            if (a.qh[l]) { a.qh[l][(size_t)rp * a.Kp + u] = 0; if (u == 0) a.sxh[l][rp] = 0.1f; }
        }
    }
}

// ---- LM served as the reference serves it (load_lm, lm.py:97: quantize_dynamic({LSTM, Linear}, qint8)): every gate matmul
// and the output layer are dynamically quantised int8 GEMVs.  Per row (= per call of the batch-1 reference):
//   (scale, zp) = ChooseQuantizationParams(min(x, 0), max(x, 0), 0, 127)      7-bit activations (reduce_range)
//   q = clamp(rint(x * (1 / scale)) + zp, 0, 127);   acc = sum_k (q_k - zp) * qw[n][k]   (int32);   y = acc * (scale * sw) + b
// (q - zp) and the int8 weights are small integers: stored as bf16 they are exact, and so is their f32-accumulated MFMA
// dot product below K = 1032, i.e. the integer arithmetic of fbgemm is reproduced bit for bit by the bf16 GEMM core.
// One workgroup per row; dst row stride ldd >= K (columns [K, ldd) are zero padding up to the 32-wide MFMA chunk).
// (scale, zero point) of a row from its range [mn, mx] (which contains 0): every thread of the 256-thread block passes its partial
// range; sqp[0] = scale, sqp[1] = zero point afterwards (block-wide barriers inside)
__device__ __forceinline__ void lm_qparams(float mn, float mx, float* smn, float* smx, float* sqp, int tid) {
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
    if (lane == 0) { smn[w] = mn; smx[w] = mx; }
    __syncthreads();
    if (tid == 0) {
        mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
        mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
        double scale = ((double)mx - (double)mn) / 127.0;
        if ((float)scale == 0.f || isinf(1.0f / (float)scale)) scale = 0.1;
        const double zmin = 0.0 - (double)mn / scale, zmax = 127.0 - (double)mx / scale;
        const double emin = fabs((double)mn / scale), emax = 127.0 + fabs((double)mx / scale);
        const double izp = emin < emax ? zmin : zmax;
        const float zp = izp < 0.0 ? 0.f : izp > 127.0 ? 127.f : (float)rint(izp);
        sqp[0] = (float)scale; sqp[1] = zp;
    }
    __syncthreads();
}
inline __global__ __launch_bounds__(256) void k_lm_quant(const float* __restrict__ src, int lds, int K, unsigned short* __restrict__ dst,
                                                  int ldd, float* __restrict__ scale_out) {
    const int r = blockIdx.x, tid = threadIdx.x;
    const float* x = src + (size_t)r * lds;
    float mn = 0.f, mx = 0.f;                                  // the range always contains 0
    for (int k = tid; k < K; k += 256) { const float v = x[k]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
    __shared__ float smn[4], smx[4], sqp[2];
    lm_qparams(mn, mx, smn, smx, sqp, tid);
    if (tid == 0) scale_out[r] = sqp[0];
    const float inv = 1.0f / sqp[0], zp = sqp[1];
    unsigned short* d = dst + (size_t)r * ldd;
    for (int k = tid; k < ldd; k += 256) {
        float q = 0.f;
        if (k < K) q = fminf(fmaxf(rintf(__fmul_rn(x[k], inv)) + zp, 0.f), 127.f) - zp;     // an integer in [-127, 127]
        d[k] = f32_to_bf16(q);
    }
}

// LSTM cell of the int8-served LM for the rows that emitted: gates = (layer 0: tab[token] | x-side GEMV) + h-side GEMV (each
// already dequantised + its bias, torch gate order i, f, g, o); fp32 state, h row-major [M][H], c [H][M].
// The new h is also QUANTISED here (qh / sxh: the image every dynamically quantised matmul that takes this h as its input would
// compute for itself -- the x side of the layer above in this step, the h side of this layer in the next one; the parameters
// depend on the vector alone, so one image serves both and equals what k_lm_quant makes of the stored h bit for bit): the LM
// step needs no quantisation launch (21 -> 13 launches per decode iteration).  H <= 1024.
inline __global__ __launch_bounds__(256) void k_lm_cell_q(const float* __restrict__ gx, const float* __restrict__ tab, const int* __restrict__ token,
                                                   const float* __restrict__ gh, const int* __restrict__ emit, float* __restrict__ h,
                                                   float* __restrict__ c, int H, int M, unsigned short* __restrict__ qh,
                                                   float* __restrict__ sxh, int Kp) {
    const int r = blockIdx.x, tid = threadIdx.x;
    if (!emit[r]) return;
    const float* a = tab ? tab + (size_t)token[r] * 4 * H : gx + (size_t)r * 4 * H;
    const float* b = gh + (size_t)r * 4 * H;
    float hv[4];
    float mn = 0.f, mx = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int u = tid + 256 * q;
        hv[q] = 0.f;
        if (u >= H) continue;
        const float gi = a[u] + b[u], gf = a[H + u] + b[H + u], gg = a[2 * H + u] + b[2 * H + u], go = a[3 * H + u] + b[3 * H + u];
        const float c2 = sigmoid_(gf) * c[(size_t)u * M + r] + sigmoid_(gi) * tanhf(gg);
        c[(size_t)u * M + r] = c2;
        hv[q] = sigmoid_(go) * tanhf(c2);
        h[(size_t)r * H + u] = hv[q];
        mn = fminf(mn, hv[q]); mx = fmaxf(mx, hv[q]);
    }
    __shared__ float smn[4], smx[4], sqp[2];
    lm_qparams(mn, mx, smn, smx, sqp, tid);
    if (tid == 0) sxh[r] = sqp[0];
    const float inv = 1.0f / sqp[0], zp = sqp[1];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int u = tid + 256 * q;
        if (u >= Kp) continue;
        float v = 0.f;
        if (u < H) v = fminf(fmaxf(rintf(__fmul_rn(hv[q], inv)) + zp, 0.f), 127.f) - zp;
        qh[(size_t)r * Kp + u] = f32_to_bf16(v);
    }
}

// ------------------------------------------------------------------------------------------------
// Beam search (the reference has none -- the spec is oracle/rnnt_oracle.py:_beam_frame).
// Rows are hypothesis slots, W per stream (row = stream * W + slot).  One selection round per decode
// iteration and stream: candidates = slots already done with the frame (B, carried unchanged) +
// (slot in A) x (token v) with score + log p(v); the W best survive in order (score desc, then
// source slot asc, carry first, v asc) and become slots 0..W-1: blank extensions join B, non-blank
// ones are extended (emit = 1: the predictor kernels step them from their PARENT's state) and are
// evaluated again unless the per-frame cap is reached.  When every slot is in B the stream moves to
// the next frame.  W = 1 is exactly the greedy machine of k_select.
// ------------------------------------------------------------------------------------------------
struct BeamState {
    int W, V, blank, max_iters, Md;
    int* t_idx;        // [M] per stream
    int* iters;        // [M] rounds done on the current frame
    const int* T_row;  // [M]
    double* score;     // [Md] sum of log p of every decision (blank included), f64 sum of f32 terms
    int* alive;        // [Md]
    int* inB;          // [Md]
    int* token;        // [Md] token of the extension (predictor input)
    int* emit;         // [Md] 1 iff the slot was extended by a non-blank token in this round
    int* parent;       // [Md] slot (0..W-1) whose state this slot continues
    int* trellis;      // [n_iter_slots][Md]  (parent << 16) | (token + 1 if extended else 0); -1 stream idle, -2 dead slot
    int* unfinished;   // [n_iter_slots]
    unsigned long long* dbg;   // LASR_DBG_TIMING: phase timestamps of workgroup 0 (wall_clock64, 10 ns ticks)
    // continuous mode (lasr_step_submit / lasr_step_wait with beam > 1): the selection loop keeps running across chunk boundaries,
    // every stream on its own frame cursor (rounds per model step = those the stream needs, not the maximum over the batch).
    //   t_idx = global frame cursor, T_row = frames available; the round's records go to slot (round % tring) of the pinned
    //   trellis ring, with a per-stream "frame finished in this round" word beside them; when a stream finishes a model step
    //   (cursor % step_T == 0) its slot scores / alive flags are stored for the host under the step's index.
    int cont;
    int tring;           // rounds in the trellis ring
    int* frame_done;     // [tring][M] pinned: 1 iff the stream finished its frame in that round
    int* iter_ctr;       // device round counter (launch-invariant groups: hipGraph replay)
    int* done_blocks;    // [64]
    int* host_flag;      // last round of a group: "streams with frames left" goes here (pinned), else nullptr
    int* host_cur;       // [M] pinned: frame cursors, stored with the flag
    int step_T, end_slots;
    double* end_score;   // [M][end_slots][W] pinned
    int* end_alive;      // [M][end_slots] pinned (bit j = slot j alive)
    // LM shallow fusion inside the beam (spec: oracle _beam_frame with an LM): a hypothesis offers its blank extension and its
    // BEST non-blank extension only; the emitted token of a selected non-blank extension is re-picked by k_beam_fuse, which then
    // also owns the host publication of the round (lm_on: k_beam_select does not publish to the host)
    int lm_on;
    int* done2;          // [64] second arrival counter (k_beam_fuse)
};

// k_beam_select with ONE WAVE PER HYPOTHESIS ROW (V <= 2048, the 512-thread form's arithmetic): wave b keeps row b's logits in
// registers (32 per lane), so the row's statistics and its ordered top-W by log p are wave-local -- DPP reductions on floats, no
// block barrier -- and only the W x W row winners meet in LDS, where wave 0 picks the ordered top-W in doubles as before.  The
// 512-thread kernel spread every row over all waves: its per-wave top-W ran W passes of an f64 argmax over W rows' heads in
// EVERY wave (12 of 25 us at W = 8) behind six block barriers.  Same results bit for bit: a lane holds the tokens of the
// "virtual threads" lane + 64 vw (vw = 0..7) of the old layout, the exp-sums are taken per virtual wave with the same butterfly
// and added in the same order, and the selection itself is a total order (score descending, ordinal ascending) -- any correct
// algorithm returns the same list.  (A row's candidates in the global top-W are its best by log p: the score offset is per row.)
template <int WT>
__global__ __launch_bounds__(64 * WT) void k_beam_select_rw(const float* __restrict__ logits, BeamState s, int iter_slot) {
    constexpr int VW = 8, KEEP = 4, NTV = 512;      // virtual waves / slots per virtual thread / virtual threads (the old layout)
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int W = s.W, V = s.V, r0 = q * W;
    const int b = w;                                  // this wave's row
    float zl[VW][KEEP];
    {
        const float* z = logits + (size_t)(r0 + (b < W ? b : 0)) * V;
#pragma unroll
        for (int vw = 0; vw < VW; ++vw)
#pragma unroll
            for (int k = 0; k < KEEP; ++k) {
                const int v = lane + 64 * vw + NTV * k;
                zl[vw][k] = v < V ? z[v] : -INFINITY;
            }
    }
    const int iter_no = s.cont ? (int)(*(const unsigned*)s.iter_ctr & 0x3fffffffu) : iter_slot;     // same value in every workgroup
    const int uslot = s.cont ? (iter_no & 63) : iter_slot;                                          // slot of the flag rings
    int* tre = s.trellis + (size_t)(s.cont ? iter_no % s.tring : iter_slot) * s.Md + r0;
    if (s.cont && q == 0 && tid == 0) {                                                             // recycle the flag rings
        s.unfinished[(uslot + 32) & 63] = 0;
        s.done_blocks[(uslot + 32) & 63] = 0;
    }
    auto publish = [&]() {           // thread 0 of every workgroup, after its last store of this launch (continuous mode)
        if (!s.cont) return;
        if (s.host_flag && !s.lm_on) __threadfence_system();     // this stream's records (pinned memory) before the count
        if (atomicAdd(&s.done_blocks[uslot], 1) == (int)gridDim.x - 1) {      // last workgroup of the launch
            if (s.host_flag && !s.lm_on) {
                const int v = atomicAdd(&s.unfinished[uslot], 0);
                __hip_atomic_store(s.host_flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            *s.iter_ctr = iter_no + 1;
        }
    };
    const bool dbg = s.dbg && q == 0 && tid == 0;
    const unsigned long long t_entry = dbg ? wall_clock64() : 0ull;
    const int t = s.t_idx[q], Tr = s.T_row[q];
    if (dbg && t < Tr) s.dbg[0] = t_entry;
    if (t >= Tr) {                                   // stream has nothing to decode: identity round
        if (tid < W) { s.emit[r0 + tid] = 0; s.parent[r0 + tid] = tid; tre[tid] = -1; }
        if (tid == 0 && s.cont) {
            s.frame_done[(size_t)(iter_no % s.tring) * gridDim.x + q] = 0;
            if (s.host_flag) s.host_cur[q] = t;
            publish();
        }
        return;
    }
    __shared__ double cand_sc[WT][WT];
    __shared__ int cand_ord[WT][WT];
    // (the table starts empty: a wave that looks at it while others are still writing -- the pruning below -- sees "no candidate")
    if (tid < WT * WT) { cand_sc[tid / WT][tid % WT] = -INFINITY; cand_ord[tid / WT][tid % WT] = 0x7fffffff; }
    __syncthreads();
    // ---- this wave's row: state, statistics, ordered top-W by log p
    const bool in = b < W;
    const double scb = in ? s.score[r0 + b] : -INFINITY;
    const int alb = in ? s.alive[r0 + b] : 0, ibb = in ? s.inB[r0 + b] : 0;
    const bool inA = alb && !ibb;                    // wave-uniform
    if (dbg) s.dbg[1] = wall_clock64();
    if (!inA) {
#pragma unroll
        for (int vw = 0; vw < VW; ++vw)
#pragma unroll
            for (int k = 0; k < KEEP; ++k) zl[vw][k] = -INFINITY;
    }
    float m = -INFINITY;
#pragma unroll
    for (int vw = 0; vw < VW; ++vw)
#pragma unroll
        for (int k = 0; k < KEEP; ++k) m = fmaxf(m, zl[vw][k]);
    m = wave_max_f32(m);
    float lg = 0.f;
    if (inA) {
        float x = 0.f;
#pragma unroll
        for (int vw = 0; vw < VW; ++vw) {             // the old layout's per-wave partial sums, in its order
            float part = 0.f;
#pragma unroll
            for (int k = 0; k < KEEP; ++k) part += expf(zl[vw][k] - m);       // exp(-inf) = 0 for the padding
            x += wave_sum_f32(part);
        }
        lg = logf(x);
    }
    int nb_arg = -1;
    if (s.lm_on && inA) {                            // argmax of z over the non-blank tokens (first maximum)
        float x = -INFINITY;
        int a = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < KEEP; ++k)
#pragma unroll
            for (int vw = 0; vw < VW; ++vw) {         // ascending v per lane: the first maximum wins
                const int v = lane + 64 * vw + NTV * k;
                if (v != s.blank && v < V && zl[vw][k] > x) { x = zl[vw][k]; a = v; }
            }
        wave_argmax_f32(x, a);
        nb_arg = a;
    }
    if (dbg) s.dbg[2] = wall_clock64();
    if (!alb) {
        // no candidate
    } else if (ibb) {                                // carried unchanged: ONE candidate, score sc (+ 0), ordinal b (V + 1)
        if (lane == 0) { cand_sc[b][0] = scb + 0.0; cand_ord[b][0] = b * (V + 1); }
    } else {
#pragma unroll
        for (int vw = 0; vw < VW; ++vw)
#pragma unroll
            for (int k = 0; k < KEEP; ++k) {
                zl[vw][k] = (zl[vw][k] - m) - lg;      // log p (padding stays -inf)
                if (s.lm_on) {                        // LM fusion: only the blank and the row's best non-blank token are candidates
                    const int v = lane + 64 * vw + NTV * k;
                    if (v != s.blank && v != nb_arg) zl[vw][k] = -INFINITY;
                }
            }
        // Two levels per lane: the best of each group k (its 8 tokens lane + 64 vw + 512 k, ascending v) is kept beside the values,
        // so a pass compares 4 group heads instead of 32 values, and only the winner's group (wave-uniform index) is rescanned.
        // Ascending v inside a group and ascending k across groups: among equal log p the lower token comes first, as before.
        float gm[KEEP];
        int ga[KEEP];
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            float x = -INFINITY;
            int a = 0x7fffffff;
#pragma unroll
            for (int vw = 0; vw < VW; ++vw)
                if (zl[vw][k] > x) { x = zl[vw][k]; a = lane + 64 * vw + NTV * k; }
            gm[k] = x; ga[k] = a;
        }
        for (int j = 0; j < W; ++j) {
            float best = -INFINITY;
            int arg = 0x7fffffff;
#pragma unroll
            for (int k = 0; k < KEEP; ++k)
                if (gm[k] > best) { best = gm[k]; arg = ga[k]; }
            wave_argmax_f32(best, arg);
            if (!(best > -INFINITY)) break;           // the row's candidates are exhausted (wave-uniform)
            const double myv = scb + (double)best;
            if (lane == 0) { cand_sc[b][j] = myv; cand_ord[b][j] = b * (V + 1) + 1 + arg; }
            {
                // Pruning: once W candidates of ANY row are strictly better than this one, neither it nor anything this row could
                // still offer is in the global top-W -- stop.  The other waves' entries are read while they write them: an entry is
                // either a published candidate or still "none", and candidates only ever get added, so a stale view can only make
                // the wave stop later, never too early.  (The final merge runs behind a barrier and sees everything.)
                // (the margin makes the count immune to a torn 64-bit read, should the hardware ever split one: a half-written entry
                //  is a NaN or the new value with its low word zero, 2^-20 relative off)
                const double other = lane < WT * WT ? ((const volatile double*)&cand_sc[0][0])[lane] : -INFINITY;
                if (__popcll(__ballot(other - myv > 1e-3 + 2e-6 * fabs(myv))) >= W) break;
            }
            const int kk = arg / NTV, vv = (arg - kk * NTV) >> 6;       // wave-uniform (the winner is)
            const bool mine = (arg & 63) == lane;                       // the winner's lane takes it out of its group
#pragma unroll
            for (int k = 0; k < KEEP; ++k) {
                if (k != kk) continue;
                float x = -INFINITY;
                int a = 0x7fffffff;
#pragma unroll
                for (int vw = 0; vw < VW; ++vw) {
                    if (mine && vw == vv) zl[vw][k] = -INFINITY;
                    if (zl[vw][k] > x) { x = zl[vw][k]; a = lane + 64 * vw + NTV * k; }
                }
                gm[k] = x; ga[k] = a;
            }
        }
    }
    __syncthreads();
    if (w != 0) return;
    if (dbg) s.dbg[9] = wall_clock64();
    // ---- the ordered top-W of the WT x WT candidate table, by wave 0 (round 6; one thread before: W passes over W list heads + W
    // slots of bookkeeping = 3.1 us of 7.6 at W = 4, 7.6 of 13.3 at W = 8).  Lane l holds candidate (row l / WT, position l % WT); its
    // rank = the number of candidates that come before it in the total order (score descending, ordinal ascending: ordinals are
    // unique, so ranks are distinct) -- the same ordered list the W-way merge of the sorted rows produced, whatever the rows' order.
    double my_sc = -INFINITY;
    int my_ord = 0x7fffffff;
    if (lane < WT * WT) { my_sc = cand_sc[lane / WT][lane % WT]; my_ord = cand_ord[lane / WT][lane % WT]; }
    const bool valid = my_sc > -INFINITY;
    int rank = 0;
    {
        const int my_hi = __double2hiint(my_sc), my_lo = __double2loint(my_sc);
#pragma unroll
        for (int c2 = 0; c2 < WT * WT; ++c2) {                  // candidate c2 broadcast from its lane (v_readlane: scalar operands, no LDS round trip)
            const double osc = __hiloint2double(__builtin_amdgcn_readlane(my_hi, c2), __builtin_amdgcn_readlane(my_lo, c2));
            const int oord = __builtin_amdgcn_readlane(my_ord, c2);
            rank += (osc > -INFINITY && (osc > my_sc || (osc == my_sc && oord < my_ord))) ? 1 : 0;
        }
    }
    // slot j (lane j < W) takes the candidate of rank j
    double sel = -INFINITY;
    int sord = 0x7fffffff;
    for (int jj = 0; jj < W; ++jj) {
        const unsigned long long m = __ballot(valid && rank == jj);
        if (!m) break;                                        // candidates exhausted: the remaining slots are dead (wave-uniform)
        const int src = __builtin_ctzll(m);
        const double v = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(my_sc), src), __builtin_amdgcn_readlane(__double2loint(my_sc), src));
        const int o = __builtin_amdgcn_readlane(my_ord, src);
        if (lane == jj) { sel = v; sord = o; }
    }
    if (dbg) s.dbg[3] = wall_clock64();
    const int round = s.iters[q] + 1;
    const bool slot = lane < W, live = slot && sel > -INFINITY;
    int inb = 1;
    if (slot) {
        const int r = r0 + lane;
        if (!live) {
            s.alive[r] = 0; s.emit[r] = 0; s.parent[r] = lane; s.score[r] = -INFINITY; tre[lane] = -2;
        } else {
            const int pb = sord / (V + 1), k = sord - pb * (V + 1);
            s.alive[r] = 1; s.parent[r] = pb; s.score[r] = sel;
            int em = 0;
            if (k > 0 && k - 1 != s.blank) {
                em = 1;
                s.token[r] = k - 1;
                inb = round >= s.max_iters ? 1 : 0;
            }
            s.emit[r] = em;
            tre[lane] = (pb << 16) | (em ? k : 0);
        }
    }
    const bool all_b = __ballot(slot && !inb) == 0ull;
    const int am = (int)(__ballot(live) & ((1ull << W) - 1ull));
    int tn = t, rn = round;
    if (all_b) { tn = t + 1; rn = 0; }
    if (slot) s.inB[r0 + lane] = all_b ? 0 : (live ? inb : 0);
    if (s.cont && all_b && tn % s.step_T == 0 && slot)      // the stream just finished one of its model steps: scores for the host
        s.end_score[((size_t)q * s.end_slots + ((tn / s.step_T - 1) % s.end_slots)) * W + lane] = sel;
    if (lane != 0) return;
    s.t_idx[q] = tn; s.iters[q] = rn;
    if (s.cont) {
        s.frame_done[(size_t)(iter_no % s.tring) * gridDim.x + q] = all_b ? 1 : 0;
        if (all_b && tn % s.step_T == 0) s.end_alive[(size_t)q * s.end_slots + ((tn / s.step_T - 1) % s.end_slots)] = am;
        if (s.host_flag) s.host_cur[q] = tn;
    }
    if (tn < Tr) atomicAdd(&s.unfinished[uslot], 1);
    if (dbg) s.dbg[4] = wall_clock64();
    publish();
}

// continuous beam loop, admission of newly encoded steps (<= 512 streams): frames-available counts by value (as k_ja_admit) and
// the joint activation of every hypothesis slot of the streams that have a frame to decode (rows = stream * W + slot)
inline __global__ void k_ja_admit_beam(const float* __restrict__ pe, const float* __restrict__ pp, const int* __restrict__ t_idx,
                                const AvailV av, int* __restrict__ avail_out, void* __restrict__ ja, int J, int Md, int W, int M_enc,
                                int MT, int ring, int bf) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Md * J) return;
    const int r = idx / J, j = idx - r * J, q = r / W;
    const int Tav = av.v[q];
    if (j == 0 && r == q * W) avail_out[q] = Tav;
    const int t = t_idx[q];
    if (t >= Tav) return;
    act_st(bf, ja, act_off(bf, r, j, MT), tanhf(pe[((size_t)(t % ring) * M_enc + q) * J + j] + pp[(size_t)r * J + j]));
}

// Beam search, one selection round: every hypothesis slot that was NOT extended takes the state of its parent slot (current
// parity) into its own slot of the other parity -- predictor h / c / BN(h) of every layer, the predictor half of the joint -- and,
// if its stream still has a frame to decode, its joint activation.  This used to ride in the epilogues of the predictor / joint
// GEMMs, 16 units (32 bytes) of a row per workgroup: 1536 workgroups of scattered 2-byte copies, 7 us each whether or not they
// had a row to compute (profiles/r04/r04_pred_timeline_cfg5.txt).  Here a row is one workgroup and the copies are whole rows.
// blockIdx.y == 0: row-major parts (h, y, pp, ja), one workgroup per slot; blockIdx.y == 1: the unit-major cell state c
// ([H][Md]: threads run over the slots, blockIdx.x over 16-unit slices).
struct BeamCarryArgs {
    const int* emit; const int* parent; int W, Md, H, J, Lp, bf, lstm;
    const void* h_in[8]; void* h_out[8];
    const void* y_in[8]; void* y_out[8];
    const float* c_in[8]; float* c_out[8];
    const float* pp_in; float* pp_out;
    const float* pe; const int* t_idx; const int* T_row; void* ja; int MTj, ring, M_enc;
};
// the work of carry block (bx, by) -- by 0: slot bx's row-major parts, by 1: block bx of the cell state; any block size >= 256
__device__ __forceinline__ void beam_carry_body(const BeamCarryArgs& a, const int bx, const int by) {
    if (by == 1) {
        // cell state: block = (16-unit slice, 256-slot block); a thread moves its slot's 16 units of every layer (the slots of a
        // stream sit next to each other: the gathered parent values come from the same 32-byte neighbourhood)
        const int nrb = (a.Md + 255) / 256;
        const int u0 = (bx / nrb) * 16, r = (bx % nrb) * 256 + threadIdx.x;
        if (!a.lstm || threadIdx.x >= 256 || u0 >= a.H || r >= a.Md || a.emit[r]) return;
        const int pr = (r / a.W) * a.W + a.parent[r];
        for (int l = 0; l < a.Lp; ++l) {
            float v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = a.c_in[l][(size_t)(u0 + q) * a.Md + pr];
#pragma unroll
            for (int q = 0; q < 16; ++q) a.c_out[l][(size_t)(u0 + q) * a.Md + r] = v[q];
        }
        return;
    }
    const int r = bx;
    if (r >= a.Md || a.emit[r]) return;
    const int pr = (r / a.W) * a.W + a.parent[r], H = a.H, J = a.J;
    const int q = r / a.W;
    const int t = a.t_idx[q];
    const bool live = t < a.T_row[q];
    const float* pes = a.pe + ((size_t)(t % a.ring) * a.M_enc + q) * J;
    if (a.bf) {
        // everything in 16-byte pieces: H % 32 == 0 and J % 32 == 0 with bf16 operands; 8 consecutive joint columns of a row are
        // one 16-byte piece of the fragment-major activation (k & 7 contiguous)
        for (int l = 0; l < a.Lp; ++l) {
            const uint4* hs = (const uint4*)((const unsigned short*)a.h_in[l] + (size_t)pr * H);
            const uint4* ys = (const uint4*)((const unsigned short*)a.y_in[l] + (size_t)pr * H);
            uint4* hd = (uint4*)((unsigned short*)a.h_out[l] + (size_t)r * H);
            uint4* yd = (uint4*)((unsigned short*)a.y_out[l] + (size_t)r * H);
            for (int i = threadIdx.x; i < H / 8; i += blockDim.x) { const uint4 hv = hs[i], yv = ys[i]; hd[i] = hv; yd[i] = yv; }
        }
        const float4* ps = (const float4*)(a.pp_in + (size_t)pr * J);
        float4* pd = (float4*)(a.pp_out + (size_t)r * J);
        const float4* es = (const float4*)pes;
        for (int i = threadIdx.x; i < J / 8; i += blockDim.x) {
            const float4 p0 = ps[2 * i], p1 = ps[2 * i + 1];
            pd[2 * i] = p0; pd[2 * i + 1] = p1;
            if (live) {
                const float4 e0 = es[2 * i], e1 = es[2 * i + 1];
                uint4 o;
                o.x = (unsigned)f32_to_bf16(tanhf(e0.x + p0.x)) | ((unsigned)f32_to_bf16(tanhf(e0.y + p0.y)) << 16);
                o.y = (unsigned)f32_to_bf16(tanhf(e0.z + p0.z)) | ((unsigned)f32_to_bf16(tanhf(e0.w + p0.w)) << 16);
                o.z = (unsigned)f32_to_bf16(tanhf(e1.x + p1.x)) | ((unsigned)f32_to_bf16(tanhf(e1.y + p1.y)) << 16);
                o.w = (unsigned)f32_to_bf16(tanhf(e1.z + p1.z)) | ((unsigned)f32_to_bf16(tanhf(e1.w + p1.w)) << 16);
                *(uint4*)((unsigned short*)a.ja + OpsBF16::aoff(r, 8 * i, a.MTj)) = o;
            }
        }
        return;
    }
    for (int l = 0; l < a.Lp; ++l) {
        const float4* hs = (const float4*)((const float*)a.h_in[l] + (size_t)pr * H);
        const float4* ys = (const float4*)((const float*)a.y_in[l] + (size_t)pr * H);
        float4* hd = (float4*)((float*)a.h_out[l] + (size_t)r * H);
        float4* yd = (float4*)((float*)a.y_out[l] + (size_t)r * H);
        for (int i = threadIdx.x; i < H / 4; i += blockDim.x) { const float4 hv = hs[i], yv = ys[i]; hd[i] = hv; yd[i] = yv; }
    }
    const float4* ps = (const float4*)(a.pp_in + (size_t)pr * J);
    float4* pd = (float4*)(a.pp_out + (size_t)r * J);
    const float4* es = (const float4*)pes;
    for (int i = threadIdx.x; i < J / 4; i += blockDim.x) {      // 4 consecutive joint columns = one 16-byte piece of the f32 fragment
        const float4 p0 = ps[i];
        pd[i] = p0;
        if (live) {
            const float4 e0 = es[i];
            *(float4*)((float*)a.ja + OpsF32::aoff(r, 4 * i, a.MTj)) = float4{tanhf(e0.x + p0.x), tanhf(e0.y + p0.y), tanhf(e0.z + p0.z), tanhf(e0.w + p0.w)};
        }
    }
}
inline __global__ __launch_bounds__(256) void k_beam_carry(const BeamCarryArgs a) { beam_carry_body(a, blockIdx.x, blockIdx.y); }
// The carry as extra workgroups of a GEMM launch of the same round (the joint-half GEMM, the last of the predictor chain): rows
// blockIdx.y >= m_groups of the grid are carry blocks -- Md slot blocks, then the cell-state blocks.  The carry depends on the
// selection kernel only and touches no row the chain's GEMMs touch, so it needs no launch (and no launch boundary) of its own:
// its copies run beside the (mostly idle-skipping) GEMM workgroups.
template <class Ops, class Epi, int MT, int NW, bool AROW, int D>
__global__ __launch_bounds__(NW * 64) void k_gemm_carry(const GemmArgs g, const typename Epi::Args ea, const BeamCarryArgs ca, const int m_groups) {
    constexpr int NT = Epi::NT, ROWS = MT * 16, LD = NT * 16 + 1;
    __shared__ float red[NW * ROWS * LD];
    __shared__ int row_map[Epi::COMPACT ? 1024 : 1];
    __shared__ int n_act_s;
    if ((int)blockIdx.y >= m_groups) {
        const int bid = ((int)blockIdx.y - m_groups) * (int)gridDim.x + (int)blockIdx.x;
        if (bid < ca.Md) beam_carry_body(ca, bid, 0);
        else beam_carry_body(ca, bid - ca.Md, 1);
        return;
    }
    gemm_body<Ops, Epi, MT, NW, AROW, D>(g, ea, blockIdx.x, blockIdx.y, red, row_map, n_act_s);
}

// start of a beam decode step: per-stream cursors and the iteration flags
inline __global__ void k_beam_begin(BeamState s, int M, int n_iter_slots) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) { s.t_idx[i] = 0; s.iters[i] = 0; }
    if (i < s.Md) { s.emit[i] = 0; s.parent[i] = i % s.W; s.inB[i] = 0; }
    if (i < n_iter_slots) s.unfinished[i] = 0;
}

// LM shallow fusion inside the beam (spec: oracle _beam_frame with an LM attached): the token of every slot that was extended
// by its parent's best non-blank token in this round is re-picked as LMFuser.fuse does (lm.py:59-79) from the PARENT's joint
// log-softmax and LM output; the round's record is patched accordingly.  One workgroup per hypothesis slot.  In continuous mode
// this launch, not k_beam_select, publishes the round to the host (the records are final only now).
// KEEP: as in k_lm_post.
template <int KEEP>
__global__ __launch_bounds__(256) void k_beam_fuse(const float* __restrict__ logits, BeamState s, int iter_slot,
                                                   const float* __restrict__ lmz, const int* __restrict__ lm_valid, float alpha,
                                                   float theta, float lm_min) {
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int W = s.W, V = s.V, q = r / W;
    const int iter_no = s.cont ? (int)((*(const unsigned*)s.iter_ctr - 1u) & 0x3fffffffu) : iter_slot;   // k_beam_select has moved on by one
    const int uslot = s.cont ? (iter_no & 63) : iter_slot;
    if (s.cont && r == 0 && tid == 0) s.done2[(uslot + 32) & 63] = 0;
    const int par = s.parent[r], prow = q * W + par;
    if (s.emit[r] && lm_valid[prow]) {               // uniform over the workgroup
        __shared__ float sh4[4];
        __shared__ float sv[4];
        __shared__ int si[4];
        const float* z = logits + (size_t)prow * V;
        float zv[KEEP];
        float best = -INFINITY;
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            const int j = tid + 256 * k;
            zv[k] = j < V ? z[j] : -INFINITY;
            best = fmaxf(best, zv[k]);
        }
        best = wave_max_f32(best);
        if (lane == 0) sh4[w] = best;
        __syncthreads();
        best = fmaxf(fmaxf(sh4[0], sh4[1]), fmaxf(sh4[2], sh4[3]));
        float part = 0.f;
#pragma unroll
        for (int k = 0; k < KEEP; ++k) part += expf(zv[k] - best);
        part = wave_sum_f32(part);
        __syncthreads();
        if (lane == 0) sh4[w] = part;
        __syncthreads();
        const float lse = logf(sh4[0] + sh4[1] + sh4[2] + sh4[3]);     // (same sums as k_select's: the joint log-softmax the greedy fuser sees)
        part = 0.f;
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            const int j = tid + 256 * k;
            zv[k] = j < V ? (zv[k] - best) - lse : 0.f;
            part += zv[k];
        }
        const float mean = block_sum_256(part, sh4, lane, w) / (float)V;
        part = 0.f;
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            const int j = tid + 256 * k;
            zv[k] = j < V ? zv[k] - mean : 0.f;                      // t.add_(-t.mean())
            part += zv[k];
        }
        const float mean2 = block_sum_256(part, sh4, lane, w) / (float)V;   // torch.std subtracts its own mean again
        part = 0.f;
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            const int j = tid + 256 * k;
            const float d = j < V ? zv[k] - mean2 : 0.f;
            part += d * d;
        }
        const float den = sqrtf(block_sum_256(part, sh4, lane, w) / (float)(V - 1)) + 1e-5f;
        const float* lz = lmz + (size_t)prow * V;
        float fb = -INFINITY;
        int fa = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            const int j = tid + 256 * k;
            if (j >= V) continue;
            const float jo = j == 0 ? lm_min : zv[k] / den;
            const float f = __fadd_rn(__fmul_rn(alpha, lz[j]), __fmul_rn(theta, jo));   // no FMA contraction
            if (f > fb) { fb = f; fa = j; }
        }
        wave_argmax_f32(fb, fa);
        __syncthreads();
        if (lane == 0) { sv[w] = fb; si[w] = fa; }
        __syncthreads();
        fb = sv[0]; fa = si[0];
#pragma unroll
        for (int k = 1; k < 4; ++k)
            if (sv[k] > fb || (sv[k] == fb && si[k] < fa)) { fb = sv[k]; fa = si[k]; }
        if (tid == 0) {
            s.token[r] = fa;
            s.trellis[(size_t)(s.cont ? iter_no % s.tring : iter_slot) * s.Md + r] = (par << 16) | (fa + 1);
        }
    }
    if (tid == 0 && s.cont) {
        if (s.host_flag) __threadfence_system();
        if (atomicAdd(&s.done2[uslot], 1) == (int)gridDim.x - 1 && s.host_flag) {
            const int v = atomicAdd(&s.unfinished[uslot], 0);
            __hip_atomic_store(s.host_flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// front-end: log-mel (TransformTime, transforms.py:306-323) -- framing, Hann window, 1024-point
// real FFT (as a 512-point complex FFT, radix 8x8x8 in registers + LDS transposes), power,
// sparse HTK mel filterbank, log(x + 1e-6).  One wave per frame, 4 frames per workgroup.
// ------------------------------------------------------------------------------------------------
struct cf { float x, y; };
__device__ __forceinline__ cf cadd(cf a, cf b) { return cf{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cf csub(cf a, cf b) { return cf{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cf cmul(cf a, cf b) { return cf{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cf mul_mi(cf a) { return cf{a.y, -a.x}; }   // * (-i)

// forward 8-point DFT, natural order in and out: X[k] = sum_n x[n] e^{-2 pi i n k / 8}
__device__ __forceinline__ void dft8(cf (&v)[8]) {
    const float h = 0.70710678118654752440f;
    cf a0 = cadd(v[0], v[4]), a1 = csub(v[0], v[4]);
    cf a2 = cadd(v[2], v[6]), a3 = mul_mi(csub(v[2], v[6]));
    cf a4 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
    cf a6 = cadd(v[3], v[7]), a7 = mul_mi(csub(v[3], v[7]));
    cf b0 = cadd(a0, a2), b2 = csub(a0, a2);       // even part: DFT4 of (x0,x2,x4,x6)
    cf b1 = cadd(a1, a3), b3 = csub(a1, a3);
    cf c0 = cadd(a4, a6), c2 = csub(a4, a6);       // odd part: DFT4 of (x1,x3,x5,x7)
    cf c1 = cadd(a5, a7), c3 = csub(a5, a7);
    // twiddles W8^k on the odd part: W8^1 = (1-i)/sqrt2, W8^2 = -i, W8^3 = (-1-i)/sqrt2
    cf t1 = cf{(c1.x + c1.y) * h, (c1.y - c1.x) * h};
    cf t2 = mul_mi(c2);
    cf t3 = cf{(-c3.x + c3.y) * h, (-c3.y - c3.x) * h};
    v[0] = cadd(b0, c0); v[4] = csub(b0, c0);
    v[1] = cadd(b1, t1); v[5] = csub(b1, t1);
    v[2] = cadd(b2, t2); v[6] = csub(b2, t2);
    v[3] = cadd(b3, t3); v[7] = csub(b3, t3);
}

struct MelArgs {
    const float* window;     // [n_fft] Hann(win) zero-padded, centred
    const float2* tw512;     // [512]  e^{-2 pi i m / 512}
    const float2* tw1024;    // [513]  e^{-2 pi i k / 1024}
    const int* fb_start;     // [n_mels] first bin of filter m
    const int* fb_off;       // [n_mels + 1] offsets into fb_w
    const float* fb_w;       // nonzero filter weights, bin-ascending per filter
    int n_mels, hop;
    // source addressing
    const float* pcm;        // offline: [B][N];  stream: ring windows [M][n_window][chunk]
    long long N;             // samples per row (offline: signal length; stream: n_window*chunk)
    int stream;              // 0 offline (reflect padded), 1 stream window
    const int* ring_head;    // stream: [M] index of the oldest chunk in the ring
    int chunk, n_window;
    const int* row_sel;      // stream: [M] destination frame slot base (>=0) or -1 = skip row
    int frame0;              // stream: first frame index of the window to compute (a = T/3 + 1)
    int frames_per_row;      // frames computed per row by this launch
    float* out;              // [rows][out_frames][n_mels]
    int out_frames;          // row stride in frames
    // offline, ragged batch (optional): per-row length, source offset into pcm, frame count
    const long long* row_N;
    const long long* row_src_off;
    const int* row_frames;
    int win_off, win_len;    // non-zero span of the window inside the n_fft frame
    int fb_nnz;              // total non-zero filter weights (<= 1536: staged in LDS)
    // streaming, <= 512 rows: the step's per-row command travels BY VALUE with this launch instead of through a
    // host->device copy of a command block (a copy costs 4-6 us plus a bubble on either side, twice per model step):
    // sel_v replaces row_sel, and workgroup (0, row) stores the row's frame count of this model step where the
    // kernels behind this one on the stream (stack + LayerNorm, the cells, the joint GEMM) read it
    int by_value;
    int* trow_out;           // [M] (nullptr: this chunk does not run the model)
    short sel_v[512];
    unsigned char trow_v[512];
    // stream: the PCM ring holds ring_chunks >= n_window chunks per row; the window of a frame ends `age` chunks
    // before the newest one (by_value: age_v[row], else 0)
    int ring_chunks;
    unsigned char age_v[512];
    // whole-window mode (stream == 0, by_value): launch row i reads pcm + i * N (reflect padded, like offline) and writes
    // frames frame0 .. frame0 + frames_per_row - 1 to row dst_row_v[i] of `out`, frame slot sel_v[i] + k (lasr_step_window)
    short dst_row_v[512];
};

// staging of the FFT twiddles and the sparse filterbank into LDS (every thread of the block takes part)
struct MelTables {
    float* fbw; int* fbs; int* fbo; float2* tw512; float2* tw1024;
};
__device__ __forceinline__ void stage_mel_tables(const MelTables& t, const float2* tw512, const float2* tw1024, const float* fb_w,
                                                 const int* fb_start, const int* fb_off, int fb_nnz, int n_mels) {
    const int nt = blockDim.x;
    for (int q = threadIdx.x; q < 512; q += nt) t.tw512[q] = tw512[q];
    for (int q = threadIdx.x; q <= 512; q += nt) t.tw1024[q] = tw1024[q];
    for (int q = threadIdx.x; q < fb_nnz; q += nt) t.fbw[q] = fb_w[q];
    for (int q = threadIdx.x; q < n_mels; q += nt) t.fbs[q] = fb_start[q];
    for (int q = threadIdx.x; q <= n_mels; q += nt) t.fbo[q] = fb_off[q];
}

// 1024-point real FFT of one frame as a 512-point complex FFT (radix 8 x 8 x 8 in registers, LDS transposes in the wave's
// scratch z[512 + 8]) + untangle + power spectrum P[0..512].  v[m] = (x[2n], x[2n+1]) for n = j + 64 m on entry.
// Contains block-wide barriers: EVERY wave of the workgroup must call it (the first barrier also publishes the
// staged tables).
// ordering of one wave's own LDS traffic: writes by some lanes, then reads of those addresses by other lanes of the SAME wave.
// The LDS unit serves a wave's instructions in issue order; what is needed is that the compiler keeps the order and that
// the writes have been issued (lgkmcnt) before the dependent reads are
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0), vmcnt / expcnt untouched
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void fft1024_power(cf (&v)[8], float2* z, float* P, const float2* s_tw512, const float2* s_tw1024, int j) {
    dft8(v);
    __syncthreads();                                     // staged twiddles / filterbank visible
#pragma unroll
    for (int k0 = 0; k0 < 8; ++k0) {
        const float2 tw = s_tw512[j * k0];
        const cf u = cmul(v[k0], cf{tw.x, tw.y});
        z[k0 * 64 + j] = float2{u.x, u.y};
    }
    wave_sync_lds();
    // ---- pass 2: thread (k0 = j>>3, b = j&7): DFT over a of u_k0[8a + b], twiddle W64^{b k1}
    {
        const int k0 = j >> 3, b = j & 7;
#pragma unroll
        for (int aa = 0; aa < 8; ++aa) {
            const float2 q = z[k0 * 64 + 8 * aa + b];
            v[aa] = cf{q.x, q.y};
        }
        dft8(v);
        wave_sync_lds();
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) {
            const float2 tw = s_tw512[8 * b * k1];
            const cf u = cmul(v[k1], cf{tw.x, tw.y});
            z[k0 * 64 + k1 * 8 + b] = float2{u.x, u.y};
        }
    }
    wave_sync_lds();
    // ---- pass 3: thread (k0, k1): DFT over b -> Z[k0 + 8 k1 + 64 k2]
    {
        const int k0 = j >> 3, k1 = j & 7;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const float2 q = z[k0 * 64 + k1 * 8 + b];
            v[b] = cf{q.x, q.y};
        }
        dft8(v);
        wave_sync_lds();
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) z[k0 + 8 * k1 + 64 * k2] = float2{v[k2].x, v[k2].y};
    }
    wave_sync_lds();
    // ---- real-FFT untangle + power: X[k] = (Z[k] + conj Z[512-k])/2 + W1024^k (Z[k] - conj Z[512-k])/(2i)
    for (int k = j; k <= 512; k += 64) {
        const float2 zk = z[k & 511];
        const float2 zn = z[(512 - k) & 511];
        const cf e = cf{0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y)};
        const cf d = cf{0.5f * (zk.x - zn.x), 0.5f * (zk.y + zn.y)};   // (Z[k] - conj Zn)/2
        const float2 tw = s_tw1024[k];
        const cf o = cmul(cf{tw.x, tw.y}, mul_mi(d));                  // W^k * d / i
        const cf X = cadd(e, o);
        P[k] = X.x * X.x + X.y * X.y;
    }
    wave_sync_lds();
}
// sparse HTK mel of the power spectrum + log(x + 1e-6): lane j takes filters j, j + 64, ..
__device__ __forceinline__ void mel_log(const float* P, const MelTables& t, int n_mels, int j, float* out) {
    // HTK filters widen with the mel index (3 bins at the bottom, ~35 at the top): with n_mels == 128 lane j takes filters
    // j and 127 - j, so every lane walks about the same number of taps (lane 63 used to walk the two widest ones)
    const bool paired = n_mels == 128;
    for (int i = 0, m = j; m < n_mels; ++i, m += 64) {
        const int mm = (paired && i == 1) ? 127 - j : m;
        const int s0 = t.fbs[mm], o0 = t.fbo[mm], cnt = t.fbo[mm + 1] - o0;
        // taps are summed in bin order (one fma chain, the order of the per-chunk kernels); the LDS reads of four taps are issued
        // together so that the chain waits for one LDS round trip per four taps instead of one per tap
        float acc = 0.f;
        int q = 0;
        for (; q + 4 <= cnt; q += 4) {
            const float p0 = P[s0 + q], p1 = P[s0 + q + 1], p2 = P[s0 + q + 2], p3 = P[s0 + q + 3];
            const float w0 = t.fbw[o0 + q], w1 = t.fbw[o0 + q + 1], w2 = t.fbw[o0 + q + 2], w3 = t.fbw[o0 + q + 3];
            acc += p0 * w0; acc += p1 * w1; acc += p2 * w2; acc += p3 * w3;
        }
        for (; q < cnt; ++q) acc += P[s0 + q] * t.fbw[o0 + q];
        out[mm] = logf(acc + 1e-6f);
    }
}

inline __global__ __launch_bounds__(256) void k_logmel(const MelArgs a) {
    __shared__ float2 sz[4][512 + 8];
    __shared__ float sp[4][520];
    __shared__ float s_fbw[1536];
    __shared__ int s_fbs[128], s_fbo[129];
    __shared__ float2 s_tw512[512], s_tw1024[513];
    const int w = threadIdx.x >> 6, j = threadIdx.x & 63;
    // twiddles + filterbank -> LDS once per workgroup (coalesced, all in flight together): the FFT
    // passes and the mel loop then have no global loads (each was a dependent ~1 us L2 round trip)
    const MelTables tab{s_fbw, s_fbs, s_fbo, s_tw512, s_tw1024};
    stage_mel_tables(tab, a.tw512, a.tw1024, a.fb_w, a.fb_start, a.fb_off, a.fb_nnz, a.n_mels);
    int fidx = blockIdx.x * 4 + w;                       // frame within the row
    const int row = blockIdx.y;
    // a wave without a frame recomputes the last valid one and skips the store: every wave of
    // the workgroup must reach every __syncthreads()
    int n_frames = a.frames_per_row;
    long long N = a.N;
    if (a.row_frames) {
        n_frames = a.row_frames[row];
        if (n_frames <= 0) return;                       // uniform over the workgroup
        N = a.row_N[row];
    }
    const bool valid = fidx < n_frames;
    if (!valid) fidx = n_frames - 1;
    int out_frame = fidx;
    int t = fidx;
    const float* src = a.pcm;
    int head = 0;
    const int NR = a.ring_chunks > 0 ? a.ring_chunks : a.n_window;
    if (a.stream) {
        if (a.by_value && a.trow_out && blockIdx.x == 0 && threadIdx.x == 0) a.trow_out[row] = a.trow_v[row];
        const int sel = a.by_value ? (int)a.sel_v[row] : a.row_sel[row];
        if (sel < 0) return;                             // uniform over the workgroup (row = blockIdx.y)
        out_frame = sel + fidx;
        t = a.frame0 + fidx;
        // ring_head = next write slot = oldest chunk; the window ends `age` chunks before the newest
        head = (a.ring_head[row] - (a.by_value ? (int)a.age_v[row] : 0) - a.n_window + 2 * NR) % NR;
        src = a.pcm + (size_t)row * NR * a.chunk;
    } else if (a.row_frames) {
        src = a.pcm + a.row_src_off[row];
    } else {
        src = a.pcm + (size_t)row * a.N;
        if (a.by_value) { out_frame = (int)a.sel_v[row] + fidx; t = a.frame0 + fidx; }
    }
    const int out_row = (!a.stream && a.by_value) ? (int)a.dst_row_v[row] : row;
    const long long base = (long long)t * a.hop - 512;
    auto sample = [&](int n) -> float {                  // windowed sample n of the 1024-frame
        // the window is zero outside [win_off, win_off + win_len): decided from the index, so the
        // window and the PCM loads are independent (no load -> branch -> load chain)
        if (n < a.win_off || n >= a.win_off + a.win_len) return 0.f;
        const float wv = a.window[n];
        long long q = base + n;
        if (q < 0) q = -q;                               // reflect (torch.stft center=True, pad_mode="reflect")
        if (q >= N) q = 2 * (N - 1) - q;
        float x;
        if (a.stream) {
            const int qi = (int)q;                       // < n_window * chunk
            const int ck = qi / a.chunk, wi = qi - ck * a.chunk;
            int slot = head + ck;
            if (slot >= NR) slot -= NR;
            x = src[(size_t)slot * a.chunk + wi];
        } else {
            x = src[q];
        }
        return x * wv;
    };
    // ---- pass 1: z[n] = x[2n] + i x[2n+1]; thread j takes n = j + 64 m
    cf v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int n = j + 64 * m;
        v[m] = cf{sample(2 * n), sample(2 * n + 1)};
    }
    fft1024_power(v, sz[w], sp[w], s_tw512, s_tw1024, j);
    // ---- sparse HTK mel + log
    if (!valid) return;
    mel_log(sp[w], tab, a.n_mels, j, a.out + ((size_t)out_row * a.out_frames + out_frame) * a.n_mels);
}

// Streaming front-end of one model step, log-mel half (<= 512 rows): workgroup (2 t' + half, row) computes five of the ten
// log-mel frames of stacked frame t' (one wave per frame) into the row's pending-frame buffer; k_stack_ln then stacks +
// LayerNorms from there.  n_buffer x 2 x rows workgroups of 5 waves: 256 for the reference shape at 64 streams, one per CU
// (round 2's one-launch form had 128 workgroups of 10 waves: half the CUs idle).  The ring append of the NEWEST client chunk is part of this
// launch (lasr_push_submit): rows with idx >= 0 take the chunk from the caller's buffer `src` -- every wave that needs samples of
// the slot being written reads them from `src`, workgroup (0, row) copies the chunk into the ring and publishes the new ring
// position -- so a model step costs one ring-append launch less.  Same arithmetic per frame as k_logmel.
struct FeMelArgs {
    const float* window; const float2* tw512; const float2* tw1024;
    const int* fb_start; const int* fb_off; const float* fb_w;
    int n_mels, hop, fb_nnz, win_off, win_len;
    float* pcm;              // [M][ring_chunks][chunk]
    int* ring_pos;           // [M] next write slot (device copy, kept in step for k_push_pcm)
    int chunk, n_window, ring_chunks, frame0;
    float* pend;             // [M][n_buffer * n_stack][n_mels]
    int pend_frames;
    int* trow_out;           // [M]: the row's frame count of this step, for the kernels behind this one
    int* enc_frames;         // pipelined protocol (else nullptr): [M] frames encoded so far; this launch snapshots it into
    int* enc_base;           //   enc_base (the ring base of this step's joint GEMM) and advances it by the step's frames
    const float* src;        // fused ring append (else nullptr): [n][chunk], row r takes chunk idx[r]
    const float* src2;       // deferred append of the PREVIOUS chunk of the same rows (else nullptr): the chunk of the last lasr_push_submit
                             // that completed no model step was not appended by a launch of its own -- this launch appends both (src2 first)
    short idx[512];          // -1: the row is not pushed by this launch
    unsigned char tp_pk[512];    // frames of this step (low nibble) | ring position BEFORE this launch's append (high nibble)
    unsigned short age_pk[512];  // 4 bits per t': chunks pushed since the window of stacked frame t' was current; 15: already in pend
};
template <int NSTACK>
__global__ __launch_bounds__(32 * NSTACK) void k_fe_mel(const FeMelArgs a) {
    constexpr int NWV = NSTACK / 2;                      // waves per workgroup = frames per half
    __shared__ float2 sz[NWV][512 + 8];
    __shared__ float sp[NWV][520];
    __shared__ float s_fbw[1536];
    __shared__ int s_fbs[128], s_fbo[129];
    __shared__ float2 s_tw512[512], s_tw1024[513];
    const int w = threadIdx.x >> 6, j = threadIdx.x & 63;
    const int tp = blockIdx.x >> 1, half = blockIdx.x & 1, row = blockIdx.y;
    const int Tr = a.tp_pk[row] & 15, pos = a.tp_pk[row] >> 4;
    const int si = a.src ? (int)a.idx[row] : -1;
    const int NR = a.ring_chunks;
    const bool two = si >= 0 && a.src2 != nullptr;       // this row's previous chunk is still waiting in src2
    const int pos_new = two ? (pos + 1) % NR : pos;      // ring slot of the newest chunk
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) {
            a.trow_out[row] = Tr;
            if (a.enc_frames) { const int e = a.enc_frames[row]; a.enc_base[row] = e; a.enc_frames[row] = e + Tr; }
            if (si >= 0) a.ring_pos[row] = (pos_new + 1) % NR;
        }
        if (si >= 0) {                                   // the ring append of this row's newest chunk (and of the deferred one)
            for (int q = two ? 0 : 1; q < 2; ++q) {
                float* d = a.pcm + ((size_t)row * NR + (q ? pos_new : pos)) * a.chunk;
                const float* s = (q ? a.src : a.src2) + (size_t)si * a.chunk;
                if ((a.chunk & 3) == 0 && ((((uintptr_t)s) | ((uintptr_t)d)) & 15) == 0) {
                    for (int i = threadIdx.x; i < a.chunk / 4; i += blockDim.x) ((float4*)d)[i] = ((const float4*)s)[i];
                } else {
                    for (int i = threadIdx.x; i < a.chunk; i += blockDim.x) d[i] = s[i];
                }
            }
        }
    }
    if (tp >= Tr) return;                                // uniform over the workgroup
    const int age = (a.age_pk[row] >> (4 * tp)) & 15;
    if (age == 15) return;                               // uniform: these frames were computed early, they are in pend
    const MelTables tab{s_fbw, s_fbs, s_fbo, s_tw512, s_tw1024};
    stage_mel_tables(tab, a.tw512, a.tw1024, a.fb_w, a.fb_start, a.fb_off, a.fb_nnz, a.n_mels);
    const int f = half * NWV + w;                        // log-mel frame of the stacked frame
    const int pos_after = si >= 0 ? (pos_new + 1) % NR : pos;
    const int head = (pos_after - age - a.n_window + 2 * NR) % NR;
    const float* ring = a.pcm + (size_t)row * NR * a.chunk;
    const float* fresh = si >= 0 ? a.src + (size_t)si * a.chunk : nullptr;
    const float* fresh2 = two ? a.src2 + (size_t)si * a.chunk : nullptr;
    const int N = a.n_window * a.chunk;
    const int base = (a.frame0 + f) * a.hop - 512;
    auto sample = [&](int n) -> float {
        if (n < a.win_off || n >= a.win_off + a.win_len) return 0.f;
        const float wv = a.window[n];
        int q = base + n;
        if (q < 0) q = -q;                               // reflect (torch.stft center=True, pad_mode="reflect")
        if (q >= N) q = 2 * (N - 1) - q;
        int ck = 0, wi = q;                              // q < n_window * chunk: a compare chain instead of an integer division
        while (wi >= a.chunk) { wi -= a.chunk; ++ck; }
        int slot = head + ck;
        if (slot >= NR) slot -= NR;
        const float x = (fresh && slot == pos_new) ? fresh[wi] : (fresh2 && slot == pos) ? fresh2[wi] : ring[(size_t)slot * a.chunk + wi];
        return x * wv;
    };
    cf v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int n = j + 64 * m;
        v[m] = cf{sample(2 * n), sample(2 * n + 1)};
    }
    fft1024_power(v, sz[w], sp[w], s_tw512, s_tw1024, j);
    mel_log(sp[w], tab, a.n_mels, j, a.pend + ((size_t)row * a.pend_frames + (size_t)tp * NSTACK + f) * a.n_mels);
}

// Resample (transforms.py:135-144: torchaudio 0.6.0 transforms.Resample = kaldi LinearResample, un-vendored):
// polyphase windowed-sinc, one filter per output phase (U = sr_out / gcd phases, `taps` taps):
//   out[n] = sum_j w[n % U][j] * x[first[n % U] + (n / U) * in_unit + j]     (x = 0 outside [0, N_in))
// HBM-bound: every input sample is read ~taps * sr_out / sr_in times through L1/L2, written once.
inline __global__ void k_resample(const float* __restrict__ x, long long N_in, const int* __restrict__ first,
                           const float* __restrict__ w, int U, int taps, int in_unit, float* __restrict__ out,
                           long long N_out) {
    const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N_out) return;
    const float* xr = x + (size_t)blockIdx.y * N_in;
    const int ph = (int)(n % U);
    const long long start = first[ph] + (n / U) * in_unit;
    const float* wp = w + (size_t)ph * taps;
    float acc = 0.f;
    for (int j = 0; j < taps; ++j) {
        const long long i = start + j;
        if (i >= 0 && i < N_in) acc += wp[j] * xr[i];
    }
    out[(size_t)blockIdx.y * N_out + n] = acc;
}

// StackDownsample (transforms.py:436-441): feats[row][t'][m*n_stack + k] = logmel[row][f0 + stride*t' + k][m]
inline __global__ void k_stack(const float* __restrict__ logmel, int T_frames, int n_mels, int n_stack, int stride,
                        float* __restrict__ feats, int Tp, int F) {
    const int tp = blockIdx.x, row = blockIdx.y;
    const float* lm = logmel + (size_t)row * T_frames * n_mels;
    float* o = feats + ((size_t)row * Tp + tp) * F;
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        const int m = f / n_stack, k = f - m * n_stack;
        o[f] = lm[(size_t)(stride * tp + k) * n_mels + m];
    }
}

// Stack (optional) + LayerNorm(feature_sz) (models.py:84,107) -> fragment-major x0[t'].
// One wave per (row, t').  mode 0: src = log-mel [rows][src_frames][n_mels] (stack on the fly with
// frame base `stride * t'` for offline or `n_stack * t'` for the stream buffer);
// mode 1: src = stacked features [..][F] row-major, row r frame t' at src[(row_off[r] + t') * F].
struct StackLnArgs {
    const float* src;
    int mode;
    int src_frames;          // mode 0: frames per row in src
    int frame_step;          // mode 0: frame advance per t' (stride offline, n_stack in stream buffer)
    const long long* row_off;// mode 1: [M] first stacked-frame index of row r (in frames)
    const int* T_row;        // [M] frames of row r in this step
    const float* ln_w;
    const float* ln_b;
    void* x0;                // fragment-major [F/chunk][Tcap*MT][64][16 B], element-typed
    int F, n_mels, n_stack, M, MT, mt_total, bf;
    float* feats_out;        // optional row-major copy of the un-normalised stacked features [M][Tmax][F]
    int Tmax;
};
// VPL: max values per lane (F <= 64 * VPL); NSTACK > 0: compile-time n_stack (F == 64 * VPL exactly,
// constant divisors, fully unrolled independent loads), 0: runtime shape
template <int VPL, int NSTACK = 0>
__global__ __launch_bounds__(256) void k_stack_ln(const StackLnArgs a) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tp = blockIdx.x * 4 + w, row = blockIdx.y;
    if (tp >= a.T_row[row]) return;
    float x[VPL];
    float sum = 0.f;
    const int n_stack = NSTACK > 0 ? NSTACK : a.n_stack;
#pragma unroll
    for (int q = 0; q < VPL; ++q) {
        const int f = lane + 64 * q;
        float val = 0.f;
        if (NSTACK == 0 && f >= a.F) {
        } else if (a.mode == 0) {
            const int m = f / n_stack, k = f - m * n_stack;
            val = a.src[((size_t)row * a.src_frames + (size_t)a.frame_step * tp + k) * a.n_mels + m];
        } else {
            val = a.src[(size_t)(a.row_off[row] + tp) * a.F + f];
        }
        x[q] = val;
        sum += val;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mu = sum / (float)a.F;
    float var = 0.f;
#pragma unroll
    for (int q = 0; q < VPL; ++q) {
        const float d = (lane + 64 * q < a.F) ? x[q] - mu : 0.f;
        var += d * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o);
    const float rstd = 1.0f / sqrtf(var / (float)a.F + 1e-5f);
#pragma unroll
    for (int q = 0; q < VPL; ++q) {
        const int f = lane + 64 * q;
        if (f >= a.F) continue;
        if (a.feats_out) a.feats_out[((size_t)row * a.Tmax + tp) * a.F + f] = x[q];
        const float y = (x[q] - mu) * rstd * a.ln_w[f] + a.ln_b[f];
        // element (t', row, f): m-tile index tp*MT + row/16 inside a layout of mt_total m-tiles
        act_st(a.bf, a.x0, act_off(a.bf, tp * a.MT * 16 + row, f, a.mt_total), y);
    }
}

// Stack + LayerNorm of the streaming front-end for one m-tile (16 rows) and one stacked frame per workgroup: wave w takes row
// 16 mt + w.  The 10 log-mel frames of a row are read with coalesced 256-byte loads and transposed through LDS into the stacked
// order (feat[m * 10 + k] = mel[k][m]); the statistics are summed in k_stack_ln's order (lane j: f = j + 64 q, then the xor tree),
// so the result is bit-identical to it; the tile's output is then written as whole 1 KiB fragments (k_stack_ln scatters 4-byte
// stores over them: 12 us in the job for 128 (row, frame) pairs, most of it store traffic and four dependent round trips).
struct LnTileArgs {
    const float* pend;       // [M][pend_frames][128]
    int pend_frames;
    const int* T_row;        // [M]
    const float* ln_w; const float* ln_b;
    void* x0;
    int MT, mt_total, bf;
};
inline __global__ __launch_bounds__(1024) void k_ln_tile(const LnTileArgs a) {
    constexpr int F = 1280, NS = 10, NM = 128, LD = F + 4;
    extern __shared__ float xs[];                    // [16][LD]
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int mt = blockIdx.x, tp = blockIdx.y, r = 16 * mt + w;
    const bool act = tp < a.T_row[r];
    float* xr = xs + w * LD;
    float lw[20], lb[20];
#pragma unroll
    for (int q = 0; q < 20; ++q) { lw[q] = a.ln_w[lane + 64 * q]; lb[q] = a.ln_b[lane + 64 * q]; }
    if (act) {
        const float* p = a.pend + ((size_t)r * a.pend_frames + (size_t)tp * NS) * NM;
        float v0[NS], v1[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) { v0[k] = p[k * NM + lane]; v1[k] = p[k * NM + lane + 64]; }
#pragma unroll
        for (int k = 0; k < NS; ++k) { xr[lane * NS + k] = v0[k]; xr[(lane + 64) * NS + k] = v1[k]; }
        wave_sync_lds();
        float x[20];
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < 20; ++q) { x[q] = xr[lane + 64 * q]; sum += x[q]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float mu = sum / (float)F;
        float var = 0.f;
#pragma unroll
        for (int q = 0; q < 20; ++q) { const float d = x[q] - mu; var += d * d; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o);
        const float rstd = 1.0f / sqrtf(var / (float)F + 1e-5f);
        wave_sync_lds();
#pragma unroll
        for (int q = 0; q < 20; ++q) xr[lane + 64 * q] = (x[q] - mu) * rstd * lw[q] + lb[q];
    }
    __syncthreads();
    // the tile's fragments: chunk c of K, lane = g * 16 + i holds row i, k = KCH c + EPL g + e
    // (gridDim.z > 1: every z-slice recomputes the tile's statistics -- 80 KB of L2 reads -- and stores its share of the K chunks:
    //  the store phase, the longer half of this kernel, then runs on gridDim.z times as many CUs)
    const size_t frag0 = (size_t)tp * a.MT + mt;
    if (!a.bf) {
        const int c_lo = (F / 16) * blockIdx.z / gridDim.z, c_hi = (F / 16) * (blockIdx.z + 1) / gridDim.z;
        for (int idx = c_lo * 64 + threadIdx.x; idx < c_hi * 64; idx += 1024) {
            const int c = idx >> 6, l = idx & 63, g = l >> 4, i = l & 15;
            if (tp >= a.T_row[16 * mt + i]) continue;
            const float* src = xs + i * LD + 16 * c + 4 * g;
            ((float4*)a.x0)[((size_t)c * a.mt_total + frag0) * 64 + l] = float4{src[0], src[1], src[2], src[3]};
        }
    } else {
        const int c_lo = (F / 32) * blockIdx.z / gridDim.z, c_hi = (F / 32) * (blockIdx.z + 1) / gridDim.z;
        for (int idx = c_lo * 64 + threadIdx.x; idx < c_hi * 64; idx += 1024) {
            const int c = idx >> 6, l = idx & 63, g = l >> 4, i = l & 15;
            if (tp >= a.T_row[16 * mt + i]) continue;
            const float* src = xs + i * LD + 32 * c + 8 * g;
            uint4 o;
            o.x = (unsigned)f32_to_bf16(src[0]) | ((unsigned)f32_to_bf16(src[1]) << 16);
            o.y = (unsigned)f32_to_bf16(src[2]) | ((unsigned)f32_to_bf16(src[3]) << 16);
            o.z = (unsigned)f32_to_bf16(src[4]) | ((unsigned)f32_to_bf16(src[5]) << 16);
            o.w = (unsigned)f32_to_bf16(src[6]) | ((unsigned)f32_to_bf16(src[7]) << 16);
            ((uint4*)a.x0)[((size_t)c * a.mt_total + frag0) * 64 + l] = o;
        }
    }
}

// streaming: append one client chunk per flagged row to its ring window
struct PushIdx { short idx[512]; };   // staging row of slot r (-1: slot not pushed), passed by value
inline __global__ void k_push_pcm(const float* __restrict__ src, const int* __restrict__ src_idx, const PushIdx pidx,
                           float* __restrict__ win, int* __restrict__ ring_pos, int chunk, int n_window) {
    const int row = blockIdx.x;
    const int si = src_idx ? src_idx[row] : (int)pidx.idx[row];
    if (si < 0) return;
    const int pos = ring_pos[row];
    float* d = win + ((size_t)row * n_window + pos) * chunk;
    const float* s = src + (size_t)si * chunk;
    if ((chunk & 3) == 0 && ((((uintptr_t)s) | ((uintptr_t)d)) & 15) == 0) {     // 16-byte requests (the source may sit across PCIe)
        const float4* s4 = (const float4*)s;
        float4* d4 = (float4*)d;
        for (int i = threadIdx.x; i < chunk / 4; i += blockDim.x) d4[i] = s4[i];
    } else {
        for (int i = threadIdx.x; i < chunk; i += blockDim.x) d[i] = s[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) ring_pos[row] = (pos + 1) % n_window;
}

}  // namespace lasr
