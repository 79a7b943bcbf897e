/* lasr_debug.h -- measurement, debug and experiment entry points of liblasr_hip.so.
 *
 * NOT part of the drop-in boundary (include/lasr.h holds the calls that replace reference interfaces, each mapped to its
 * file:line in INTEGRATION.md).  Nothing here has a counterpart in the reference; bench.py, the tools/ scripts and a few GPU
 * tests use these to time kernels inside the job, read resident state back for comparison with the oracle, and run the
 * experiments recorded under profiles/.  tests/test_abi.py checks that the library exports exactly lasr.h + lasr_debug.h. */
#ifndef LASR_DEBUG_H
#define LASR_DEBUG_H

#include "lasr.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Debug (LASR_DBG_TIMING=1 at create): per-workgroup phase timestamps (s_memtime at entry / setup /
 * K-loop end / reduce / exit, wall clock at entry / exit) of the last launch of each GEMM kind
 * (0 encoder cell, 1-2 predictor layers, 3 PPJ, 4 logits): out[5][4096][16] (slots 8..15: per-wave end of the K loop). */
int lasr_debug_timing(lasr_ctx* c, unsigned long long* out);

/* Debug / test read-out of the resident per-slot state (what the reference keeps in Python closures: the servicer's window,
 * Buffer.saved, the encoder / predictor state of Transducer.transcribe_stream, models.py:466-500).  Synchronises both engine
 * streams, then copies one [rows][cols] row-major float32 matrix to `out` (host, `cap` floats; *rows / *cols optional):
 *   what 0: LayerNorm'ed stacked features of frame `index` of the last step   [M][feat]
 *        1 / 2: encoder h / c of layer `index`                                [M][hidden]
 *        3: encoder half of the joint (synchronous protocol), frame `index`   [M][joint]
 *        4: predictor half of the joint                                        [M * beam][joint]
 *        5: predictor h of layer `index`                                       [M * beam][hidden]
 *        6: the PCM ring                                                       [M][(n_window + n_buffer - 1) * chunk]
 *        7: pending log-mel frames (Buffer)                                    [M][n_buffer * n_stack * n_mels]
 *        8: last encoder layer's BatchNorm'ed output, frame `index`            [M][hidden]
 *       10: (LASR_DBG_ENCLOG + LASR_DBG_PENDLOG=1) the pending log-mel frames as they were behind logged step `index`
 *        9: integers as floats [8][M]: device ring position, host mirror, chunks pushed, frames pending, frames of the
 *           last step, frame cursor, last token, emitted flag
 * M = max_streams rounded up to 64.  LASR_EFULL if cap is too small (*rows / *cols are set). */
int lasr_debug_read(lasr_ctx* c, int what, int index, float* out, size_t cap, int* rows, int* cols);

/* Race detector for the two-stream protocols (LASR_DBG_ENCLOG=N at create): behind the encoder cells of each of the first N model
 * steps one extra launch on the main stream stores, per row, exact checksums (sum of the element bit patterns mod 2^32) of the
 * step's LayerNorm'ed input frames, every layer's c and h, and the last layer's output frames: out[step][32][M] with entries
 * 0..T-1 input frames, T..T+L-1 c, T+L..T+2L-1 h, T+2L..2T+2L-1 output frames, then the pending log-mel frames and the PCM ring
 * (T = n_buffer, L = enc_layers).  The encoder does
 * not depend on the decode stream, so the log of a pipelined run must equal the log of a synchronous run of the same input word
 * for word (tools/r06/enc_racelog.py).  *steps = steps logged; the log restarts after the call. */
int lasr_debug_enclog(lasr_ctx* c, unsigned* out, size_t cap, int* steps);

/* Interference probe (needs an idle engine whose PCM ring holds audio): `iters` back-to-back launches of the streaming log-mel
 * kernel on the main stream over the SAME ring contents, a per-row checksum of the output behind each, while the decode stream
 * runs `per_iter` launches per iteration of aggressor 0 nothing, 1 the vocabulary GEMM over all hypothesis rows, 2 a predictor pass,
 * 3 the joint half.  lds_pad = unused dynamic LDS of the log-mel launches in bytes (-1: what the engine uses, see
 * lasr_ctx::fe_lds_pad; 0: none -- the round-6 interference shows up beside aggressor 1 with bf16 operands and >= 512 hypothesis
 * rows).  The first launch runs alone and is the reference: *bad_launches / *bad_rows = launches / (launch, row) pairs whose
 * output differs from it.  With the engine's setting both must be 0: the two streams share no data.  The log-mel launches write to
 * scratch buffers (the slots' pending frames stay); aggressors 2 and 3 overwrite the slots' predictor / joint state: reset the slots
 * before streaming through them again. */
int lasr_debug_fe_race(lasr_ctx* c, int iters, int aggressor, int per_iter, int lds_pad, int* bad_launches, int* bad_rows);

/* Engine configuration as resolved at lasr_create (defaults + LASR_* environment switches): *value = the integer behind `key`.
 * Keys: "enc_wave", "enc_u12", "main_graph", "pump_G", "la_stream", "la_offline", "cell_nw", "use_graphs", "M", "fe_lds_pad", "roctx" (1: LASR_ROCTX found the marker library).  LASR_EINVAL for an unknown key.  bench.py records these beside every line. */
int lasr_debug_config(lasr_ctx* c, const char* key, int* value);

/* Roofline micro-benchmark of the dominant kernel (one encoder LSTM-cell launch: all rows active,
 * layer `layer`), `iters` back-to-back launches timed with HIP events on the ctx stream.
 * Returns average microseconds per launch in *us. */
int lasr_bench_cell(lasr_ctx* c, int layer, int iters, double* us);

/* Diagnostic of the pipelined protocol's premise: one wave per engine stream (the ctx stream, the decode stream) holds its stream
 * for delay_us microseconds; *ratio = wall time / delay_us -- about 1 when the two streams run concurrently, about 2 when the
 * runtime has put both on one hardware queue.  Needs an idle engine (no submitted step). */
int lasr_overlap_probe(lasr_ctx* c, int delay_us, double* ratio);
/* Experiment hook (no reference counterpart): a synthetic neighbour beside the job, on a stream and hardware queue of its own.
 * kind 1: n_wg one-wave workgroups issue f32 MFMAs from registers for `ms` milliseconds (matrix-pipe cycles, no memory traffic);
 * kind 2: they stream a 512 MB buffer with non-temporal loads (HBM bandwidth, no MFMA), kind 3: each wave re-reads its own 96 KB (hits in L2:
 * the L2 -> CU path), kind 4: they stream a 128 MB buffer with ordinary loads (Infinity Cache); the call returns at once.  kind 0: wait for the neighbour, *rate = what it
 * achieved (TFLOP/s for kind 1, GB/s for the others).  bench.py --neighbour: what a neighbour that takes only ONE resource costs the
 * two-stream job says which resource the job is short of. */
int lasr_bench_neighbour(lasr_ctx* c, int kind, int n_wg, int ms, double* rate);

/* In-job timing of the dominant kernel (bench.py `roofline`): while on, the encoder-cell sequence of every model
 * step (enc_layers x frames back-to-back launches of the fused LSTM-cell GEMM) is bracketed by one HIP-event pair
 * on the stream the cells run on.  lasr_cell_prof_read drains the pairs: microseconds and cell launches
 * accumulated since profiling was switched on (average launch duration = us_total / launches, next to whatever
 * else shares the GPU -- the number a rocprofv3 kernel trace of the same run shows). */
/* on: 0 off; 1 HIP-event pairs + in-kernel clocks; 2 in-kernel clocks only (lasr_cell_prof_kernel) -- an event record
 * between two kernels costs the stream a bubble of a few microseconds, twice per model step in mode 1. */
int lasr_cell_prof(lasr_ctx* c, int on);
/* Stream timeline of the pipelined protocol (diagnostics): while on, timestamped marks are recorded on the main
 * stream (tag 1 push, 3 first cell, 4 cells done, 5 model step enqueued) and on the decode stream (10 group reached,
 * 11 + 100 G [+ 1000: steps admitted] admission done, 12 group done); at most 8192 marks.  lasr_trace_read
 * synchronises the device and returns the marks in record order with their time in microseconds since lasr_trace(c, 1). */
int lasr_trace(lasr_ctx* c, int on);
int lasr_trace_read(lasr_ctx* c, double* us, int* tags, int cap, int* n);
int lasr_cell_prof_read(lasr_ctx* c, double* us_total, long long* launches);
/* ... and the kernels' own durations over the same period (at most 32768 launches): per cell launch, max exit - min
 * entry of the device's constant wall clock over the kernel's workgroups -- what a kernel trace reports as the kernel's
 * duration (no launch gaps, no event overhead).  *cells (may be NULL) = LSTM cells those launches computed: the encoder
 * pass runs as a layer wavefront, a launch holds the independent cells (l, t) of one anti-diagonal.  Synchronises the device. */
int lasr_cell_prof_kernel(lasr_ctx* c, double* us_total, long long* launches, long long* cells);

/* Capacity of the native serving front (lasr_front_*) with NATIVE per-stream producers: one thread per stream pushes its
 * n_chunks chunks (chunks_per_push per call) and reads its results to the end.  pcm [n_streams][n_chunks * chunk] float32 host;
 * tokens [n_streams][cap] / n_tok [n_streams] receive every stream's token ids; *seconds = wall time from the common start to
 * the last stream's end-of-stream; stats4 (optional) = lasr_front_stats.  Creates and destroys its own front on `c`. */
int lasr_bench_front(lasr_ctx* c, int depth, int reset_steps, int n_streams, const float* pcm, int n_chunks, int chunks_per_push,
                     int32_t* tokens, int cap, int* n_tok, double* seconds, long long* stats4);

#ifdef __cplusplus
}
#endif
#endif /* LASR_DEBUG_H */
