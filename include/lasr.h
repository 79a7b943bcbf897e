/*
 * lasr.h -- C ABI of liblasr_hip.so: the MI355X (gfx950) streaming RNN-Transducer inference path.
 *
 * The reference (iceychris/LibreASR) has NO FFI / plugin / operator interface: its hot path is a
 * Python call surface over torch CPU ops.  This header is therefore the boundary a maintainer
 * binds with ctypes (INTEGRATION.md shows the stub); each entry point names the reference
 * interface it replaces (paths relative to the reference tree).
 *
 * Conventions
 *   - every function returns LASR_OK (0) or a negative LASR_E* code; nothing throws or aborts;
 *     lasr_last_error() gives a human-readable message for the last failure on that ctx.
 *   - the caller owns every buffer it passes; the ctx owns all device memory it allocates.
 *   - a ctx is single-caller (one scheduler thread per GPU); all work is enqueued on the
 *     hipStream_t given at creation (pass torch.cuda.current_stream().cuda_stream, or NULL).
 *   - "row" == stream slot: slot s occupies batch row s of every device buffer, so there is no
 *     state gather/scatter; rows that do not take part in a call are masked.
 *   - pointers documented "host or device" are classified with hipPointerGetAttributes.
 */
#ifndef LASR_H
#define LASR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LASR_OK 0
#define LASR_EINVAL (-1)   /* bad argument / unsupported shape          */
#define LASR_ENOMEM (-2)   /* device or host allocation failed          */
#define LASR_EHIP (-3)     /* a HIP runtime call failed                 */
#define LASR_ESTATE (-4)   /* call not valid in the slot's/ctx's state  */
#define LASR_EFULL (-5)    /* no free stream slot / output buffer small */

typedef struct lasr_ctx lasr_ctx;

/* Model + front-end description.  Mirrors config/testing.yaml:202-229 (model) and :133-143,
 * :350-374 (front-end), and Transducer.__init__ (libreasr/lib/models.py:190-234). */
typedef struct {
    int32_t feat;        /* feature_sz = n_mels * n_stack (1280)                       */
    int32_t hidden;      /* hidden_sz == out_sz (encoder.linear / predictor.linear = id) */
    int32_t enc_layers;  /* encoder LSTM layers                                          */
    int32_t pred_layers; /* predictor layers                                             */
    int32_t pred_cell;   /* 0 = NBRC/GRU cell (haste/nbrc.py), 1 = LSTM                  */
    int32_t embed;       /* embed_sz; if embed != hidden a Linear(embed,hidden) follows  */
    int32_t joint;       /* joint_sz                                                     */
    int32_t vocab;       /* vocab_sz                                                     */
    int32_t blank;       /* 0   (models.py:203)                                          */
    int32_t bos;         /* 2   (models.py:227)                                          */
    int32_t n_fft;       /* 1024 */
    int32_t win;         /* 400  */
    int32_t hop;         /* 160  */
    int32_t n_mels;      /* 128  */
    int32_t n_stack;     /* 10   */
    int32_t stride;      /* 8  (StackDownsample.downsample)                              */
    int32_t n_buffer;    /* 2  (Buffer.n_buffer, transforms.py:455-471)                  */
    int32_t n_window;    /* 3  (BUFFER_N_FRAMES, api-server.py:26)                       */
    int32_t chunk;       /* client chunk length in samples (1280 = 80 ms)                */
    int32_t sample_rate; /* 16000 */
    int32_t dtype;       /* 0 = f32; 1 = bf16 MFMA operands (weights + GEMM-input        */
                         /* activations), f32 accumulate / cell state / logits; needs    */
                         /* feat, hidden, joint % 32 == 0                                */
    int32_t max_streams; /* number of stream slots (= batch rows)                        */
    int32_t max_iters_offline; /* 3  (decode_greedy default, models.py:369)             */
    int32_t max_iters_stream;  /* 10 (transcribe_stream default, models.py:458)         */
    int32_t beam;        /* 1 = greedy (the only decode the reference has); 2..8 = beam  */
                         /* search width W (spec: oracle _beam_frame):     */
                         /* W hypothesis slots per stream, streams x W <= 1024 rows,     */
                         /* vocab <= 2048 (a hypothesis row's logits live in one wave).  */
                         /* lasr_fetch then returns the WHOLE current best hypothesis    */
                         /* after every model step (it may change retroactively),        */
                         /* neg_logp = -its score, align = 0.  Both protocols (the        */
                         /* pipelined one: <= 512 stream slots); an attached LM is fused  */
                         /* inside the beam (see lasr_attach_lm).                         */
} lasr_model_desc;

/* Fills `d` with the reference defaults listed above (4x1024 encoder, 2xNBRC predictor). */
void lasr_default_desc(lasr_model_desc* d);

/* Number of float32 values lasr_create expects in `weights` for `d` (0 if d is invalid).
 * The blob is the reference state_dict flattened in this order, every tensor in its reference
 * layout:
 *   encoder.input_norm.{weight,bias}
 *   per encoder layer i: rnn_stack.hs.i [2,H] (h0,c0); bns.i.{weight,bias,running_mean,running_var};
 *                        rnns.i.{weight_ih_l0 [4H,I], weight_hh_l0 [4H,H], bias_ih_l0, bias_hh_l0}
 *   predictor.embed.weight [V,E]; if E != H: predictor.ffn.{weight [H,E], bias}
 *   per predictor layer i: hs.i [S,H] (S=1 NBRC, 2 LSTM); bns.i.* (as above);
 *        NBRC: rnns.i.{kernel [I,3H], recurrent_kernel [H,3H], bias [3H], recurrent_bias [3H]}
 *        LSTM: rnns.i.{weight_ih_l0, weight_hh_l0, bias_ih_l0, bias_hh_l0}
 *   joint.joint.0.{weight [J,2H], bias}; joint.joint.2.{weight [V,J], bias}                  */
size_t lasr_weight_count(const lasr_model_desc* d);

/* Replaces: load_stuff / Transducer.from_config + load_asr_model (libreasr/lib/inference.py:18-51,
 * models.py:236-259, model_utils.py:61-95).  Packs the weights into MFMA-fragment order, folds
 * BatchNorm(eval) into scale/shift, builds the predictor input tables, uploads everything. */
int lasr_create(int device, const lasr_model_desc* d, const float* weights, size_t n_weights,
                void* hip_stream, lasr_ctx** out);
void lasr_destroy(lasr_ctx* c);
const char* lasr_last_error(const lasr_ctx* c);

/* ---- stream slots: per-stream state lives on the device (replaces the Python closure state of
 * Transducer.transcribe_stream, models.py:466-500, the Buffer transform's `saved` list,
 * transforms.py:461-471, and the servicer's 3-chunk `frames` list, api-server.py:85-102). */
int lasr_stream_open(lasr_ctx* c, int* slot);
/* what: 1 = encoder state, 2 = predictor (re-run on BOS), 4 = LM state (reset_lm, models.py:491-492; no-op without an attached LM), 8 = front-end
 * window + frame buffer; OR-able.  reset() of models.py:494-497 == 1|2|4. */
#define LASR_RESET_IF_DECODED 16   /* OR into `what` (with bits 1 | 2 | 4 only): accept a slot whose submitted steps are uncollected  */
                                   /* but already DECODED for this slot (see lasr_peek_slot); without it a slot with a submitted,    */
                                   /* uncollected step is always refused (LASR_ESTATE)                                                */
int lasr_stream_reset(lasr_ctx* c, int slot, int what);
/* lasr_stream_reset of n distinct slots with the same `what` in one call (one command block, one set of launches): every slot is
 * checked before anything changes -- an error leaves all of them as they were.  A scheduler's tick that applies the servicer's
 * reset rule (api-server.py:131-134) to several streams at once. */
int lasr_stream_reset_many(lasr_ctx* c, const int* slots, int n, int what);
int lasr_stream_close(lasr_ctx* c, int slot);

/* ---- streaming hot path, batched over n slots ------------------------------------------------
 * lasr_push_pcm: one client chunk of `chunk` float32 samples per listed slot (replaces
 * tensorize + the window cat of api-server.py:88-102).  pcm: [n, chunk] host or device.
 *   - HOST memory (pageable or pinned): copied into the engine's pinned staging ring BEFORE the call returns (helper
 *     threads: LASR_PUSH_THREADS, default 2); the caller's buffer is free on return.
 *   - DEVICE memory: read by a kernel in stream order on the ctx stream, like hipMemcpyAsync(D2D) would.
 *   - lasr_push_pcm_ex with LASR_PUSH_PINNED_NOCOPY (opt-in): pcm must be PINNED host memory (hipHostMalloc /
 *     hipHostRegister / a torch pin_memory() tensor); nothing is copied, a kernel reads the buffer over PCIe AFTER the call
 *     has returned -- also after lasr_step_submit / lasr_push_submit have returned.  *ticket identifies the push: the buffer
 *     must stay untouched until lasr_push_consumed(ctx, ticket) returns 1 (0 = not yet; an event query, no blocking).
 *     Every host push gets a ticket (ticket may be NULL); device pushes report -1.
 *   - lasr_push_submit with LASR_PUSH_DEVICE_STABLE (opt-in, DEVICE memory): the caller promises that the buffer stays
 *     unchanged until the model step this chunk belongs to has been collected (lasr_step_wait) or lasr_sync has returned.  The
 *     chunk of a call that completes no model step is then not appended to the PCM ring by a launch of its own: the next
 *     lasr_push_submit of the same slots appends both chunks in its front-end launch (one launch less per model step on the
 *     stream that binds the job).  Host pushes get this without a flag -- their source is the engine's own staging entry.
 *     Results are the same either way; any other call that touches the ring appends the waiting chunk first.
 * lasr_step_stream: for every listed slot whose window is full, computes the log-mel frames of
 * the window's middle (TransformTime + StreamPostprocess + StackDownsample, transforms.py:
 * 306-342,436-441), buffers them (Buffer), and for slots whose buffer reached n_buffer runs
 * encoder + greedy decode with carried state (models.py:506-575).  Blocks until the tokens of
 * this step are on the host.  n_ran (optional) = number of slots the model ran for. */
#define LASR_PUSH_PINNED_NOCOPY 1
#define LASR_PUSH_DEVICE_STABLE 2
int lasr_push_pcm_ex(lasr_ctx* c, const int* slots, int n, const float* pcm, int flags, long long* ticket);
int lasr_push_consumed(lasr_ctx* c, long long ticket);
int lasr_push_pcm(lasr_ctx* c, const int* slots, int n, const float* pcm);
int lasr_step_stream(lasr_ctx* c, const int* slots, int n, int* n_ran);

/* Generic streaming clients (any chunk length, any sample rate; the browser client sends 44.1 / 48 kHz,
 * apps/web/src/App.js:68): one call per servicer iteration with the window the servicer has concatenated from the
 * last 3 client frames (api-server.py:83-115).  pcm = [n][N] float32, host or device, one window per listed slot, at
 * `sr` Hz.  Does what x_tfm_stream does per call: Resample of the whole window to 16 kHz (transforms.py:141-144;
 * skipped when sr is 16000), log-mel of the window (reflect padding at both ends), frames T//3 + 1 .. + n_stack
 * (transforms.py:335-342), stack, Buffer(n_buffer); when a slot's buffer is full the model runs (synchronous protocol,
 * like lasr_step_stream; n_ran = slots that ran it; tokens through lasr_fetch).  LASR_EINVAL if the window is too short
 * for n_stack frames after the cut.  A slot uses EITHER this form OR lasr_push_pcm + lasr_step_* (LASR_ESTATE when a slot
 * that still has frames pending from the other form is passed).  At most 512 stream slots. */
int lasr_step_window(lasr_ctx* c, const int* slots, int n, const float* pcm, int64_t N, int sr, int* n_ran);

/* Pipelined form (throughput mode): lasr_step_submit enqueues this chunk's front-end + encoder on the
 * ctx stream and returns; lasr_step_wait keeps ONE greedy decode loop running on a second HIP stream and
 * blocks until the tokens of the OLDEST submitted model step are on the host (n_ran = its slot count,
 * 0 if no model step was pending); rows that finish a step early continue with the frames of the later,
 * already encoded steps.  Issue submit(k+1) ... before wait(k): the encoders of the next chunks then
 * overlap the latency-bound decode loop.  Up to lasr_max_inflight() model steps may be submitted and not
 * yet collected: 25 with the reference front-end (n_buffer 2, max_iters_stream 10; 15 until round 5); fewer when n_buffer *
 * max_iters_stream is large (the limit keeps the per-row rings of encoder frames (64) and tokens (512) from
 * wrapping).  A deep pipeline is what absorbs bursty streams: a row that emits many tokens on a few frames falls
 * behind while the others run ahead on the frames of later steps.  At the limit lasr_step_submit returns LASR_ESTATE and changes nothing (the pushed chunk stays
 * pushed: call lasr_step_wait, then submit again).
 * Every other state-changing call returns LASR_ESTATE while a submitted step is uncollected.
 * Results are identical to lasr_step_stream (same kernels, same order per stream).  With beam > 1 the selection loop runs
 * across chunk boundaries too (every stream on its own frame cursor); lasr_fetch after lasr_step_wait hands out the best
 * hypothesis as of that model step. */
int lasr_step_submit(lasr_ctx* c, const int* slots, int n);
/* lasr_push_pcm_ex + lasr_step_submit for the same slot list in ONE call (same results): when the chunk completes a model
 * step the front-end launch takes the newest chunk from `pcm` and appends it to the PCM ring itself -- one launch less on
 * the critical stream per model step.  LASR_ESTATE (in-flight limit reached, unsupported shape for the pipelined protocol) and
 * LASR_EINVAL mean NOTHING was pushed: call lasr_step_wait and repeat the call.  LASR_EHIP (the runtime failed underneath) leaves
 * the chunk pushed. */
int lasr_push_submit(lasr_ctx* c, const int* slots, int n, const float* pcm, int flags, long long* ticket);
/* lasr_push_submit with the chunk of slots[i] at rows[i] (HOST memory, pageable or pinned, `chunk` floats each; free on return):
 * what a server holds when every connection's frame sits in its own receive buffer (api-server.py:88-91 tensorizes one message
 * per stream; a scheduler that batches them would otherwise gather the rows into one array first).  The gather is the copy into
 * the engine's staging ring that a host push makes anyway.  Same error contract as lasr_push_submit. */
int lasr_push_submit_rows(lasr_ctx* c, const int* slots, int n, const float* const* rows, long long* ticket);
int lasr_step_wait(lasr_ctx* c, int* n_ran);
/* Non-consuming look at ONE slot's submitted, uncollected model steps (greedy decode): *n_inflight of them, of which the
 * oldest *n_decoded are finished for this slot; counts[k] (k < *n_decoded, at most cap_steps) = tokens of step k, concatenated
 * in tokens[cap].  lasr_step_wait / lasr_fetch still hand the same tokens out later.  A slot whose submitted steps are all
 * decoded may be reset (lasr_stream_reset with bits 1 | 2 | 4 and LASR_RESET_IF_DECODED) before they are collected: this is how a scheduler applies the
 * servicer's reset rule (api-server.py:44-50, 131-134: judge the step, maybe reset, only then run the stream's next step)
 * without waiting for `steps in flight` collections.  LASR_EFULL: buffers too small; LASR_ESTATE with beam > 1. */
int lasr_peek_slot(lasr_ctx* c, int slot, int32_t* tokens, int cap, int32_t* counts, int cap_steps, int* n_decoded,
                   int* n_inflight);
/* The same for n slots in one call: the skip[i] oldest steps of slots[i] are left out of row i of tokens [n][cap] /
 * counts [n][cap_steps] (skip may be NULL); n_decoded[i] counts every decoded step of the slot, the skipped ones included. */
int lasr_peek_many(lasr_ctx* c, const int* slots, int n, const int* skip, int32_t* tokens, int cap, int32_t* counts,
                   int cap_steps, int* n_decoded, int* n_inflight);
int lasr_step_pending(lasr_ctx* c);   /* submitted model steps not yet collected (0..lasr_max_inflight()) */
int lasr_max_inflight(const lasr_ctx* c);

/* ---- offline path (Transcribe RPC, api-server.py:64-80 -> Transducer.transcribe,
 * models.py:365-455): whole utterances, fresh state, max_iters_offline.
 * pcm: concatenated samples of the n utterances (host or device), n_samples[i] each (>= 2400).
 * Results are fetched with lasr_fetch on the same slots. */
int lasr_transcribe_pcm(lasr_ctx* c, const int* slots, int n, const float* pcm,
                        const int64_t* n_samples);
/* Same, starting from stacked features [sum(n_frames), feat] row-major (x_tfm output). */
int lasr_transcribe_feats(lasr_ctx* c, const int* slots, int n, const float* feats,
                          const int32_t* n_frames);

/* Streaming on feature chunks instead of PCM (the reference's own split: x_tfm_stream output ->
 * Transducer.transcribe_stream, models.py:506-575): feats [n, T, feat] host or device, carried
 * state, max_iters_stream.  Blocks until the tokens are on the host. */
int lasr_step_feats(lasr_ctx* c, const int* slots, int n, const float* feats, int T);

/* New tokens of `slot` since the last fetch (int32 ids incl. nothing for blanks); with beam > 1
 * the complete best hypothesis as of the last model step (empty if no step ran since the last fetch).
 * neg_logp / align (optional): offline metrics of the last lasr_transcribe_* call
 * (-sum log p of every decision, models.py:420-422,455; alignment_score, models.py:445-453). */
int lasr_fetch(lasr_ctx* c, int slot, int32_t* tokens, int cap, int* n_new, double* neg_logp,
               double* align);

/* Batched form: tokens [n, cap] (row i = new tokens of slots[i]), n_new [n].  One call per step
 * instead of one per stream. */
int lasr_fetch_many(lasr_ctx* c, const int* slots, int n, int32_t* tokens, int cap, int* n_new);

/* ---- op-level entry points (parity tests and roofline micro-benchmarks).  Device pointers
 * unless noted; all enqueue on the ctx stream and return without synchronising. -------------- */
/* log-mel of whole signals: pcm [B, N] -> logmel [B, T, n_mels], T = 1 + N / hop. */
int lasr_logmel(lasr_ctx* c, const float* pcm, int B, int64_t N, float* logmel);
/* stacked features [B, T', feat] from logmel [B, T, n_mels] (StackDownsample). */
int lasr_stack(lasr_ctx* c, const float* logmel, int B, int T, float* feats, int* Tp);
/* Encoder.forward with fresh (learned) initial state: feats [B, T', feat] -> out [B, T', hidden]
 * (B <= max_streams).  h_out/c_out optional [enc_layers, B, hidden]. */
int lasr_encoder(lasr_ctx* c, const float* feats, int B, int Tp, float* out, float* h_out,
                 float* c_out);
/* Predictor on a token sequence per row from the learned initial state: tok [B, U] (host int32)
 * -> out [B, hidden] after the last token. */
int lasr_predictor(lasr_ctx* c, const int32_t* tok, int B, int U, float* out);
/* Joint + log-softmax: h_pred, h_enc [B, hidden] -> logits [B, vocab] (pre-softmax),
 * logp_max [B], argmax [B] (device). */
int lasr_joint(lasr_ctx* c, const float* h_pred, const float* h_enc, int B, float* logits,
               float* logp_max, int32_t* argmax);

/* ---- timing / introspection ------------------------------------------------------------------ */
typedef struct {
    double frontend_ms, encoder_ms, decode_ms; /* HIP-event time of the last step's stages     */
    int32_t decode_iters;                      /* joint evaluations rounds in the last step    */
    int32_t frames;                            /* encoder frames (T) of the last step          */
    double cell_ms;                            /* sum over the LSTM-cell launches of last step */
    int32_t cell_launches;
} lasr_step_stats;
int lasr_get_stats(lasr_ctx* c, lasr_step_stats* s);
/* enable per-stage HIP-event timing (costs a few us per step) */
int lasr_set_profiling(lasr_ctx* c, int on);
/* hipStreamSynchronize on the ctx stream */
int lasr_sync(lasr_ctx* c);

/* ---- Resampling for non-16 kHz clients.  Replaces Resample.encodes, transforms.py:135-144
 * (torchaudio 0.6.0 transforms.Resample == compliance.kaldi.resample_waveform, lowpass_filter_width 6: a
 * dependency that is NOT in the reference tree -- restated from its published algorithm, parity unpinned).
 * pcm / out: device [B][N_in] / [B][*N_out] float32; out == NULL only returns *N_out.  The reference
 * resamples every transform call separately (the whole utterance, or each 3-chunk window), and so do the
 * mirrors in libreasr_amd/lib/transforms.py; the fused streaming entry points take 16 kHz PCM. */
int lasr_resample(lasr_ctx* c, const float* pcm, int B, int64_t N_in, int sr_in, float* out, int64_t* N_out);

/* ---- LM shallow fusion.  Replaces LMFuser / LM of lm.py:20-83 as wired into both greedy
 * loops (models.py:401,431,440 / :478,558,569; attached to the model at config.py:143-157): after a
 * non-blank decision the token is re-picked as argmax(alpha * standardize(LM log-probs) + theta *
 * standardize(joint log-softmax)) with entry 0 forced to min_val on both sides; the LM is stepped on every
 * emitted token.  The blank / non-blank decision and the accumulated log p are those of the joint alone.
 * fp32 (or the ctx's bf16 operand mode); the int8-served form of the reference (lm.py:97) is lasr_attach_lm_int8 (greedy only).
 * With beam > 1 (no reference: spec = oracle _beam_frame with an LM) every hypothesis slot carries its own LM state; a hypothesis
 * offers its blank extension and its BEST non-blank extension per round, and the token a selected non-blank extension emits is the
 * fuser's re-pick -- beam = 1 is then exactly the reference's greedy loop with fusion.  lasr_stream_reset bit 4 resets the LM state
 * (reset_lm, models.py:491-492); lasr_transcribe_* start every utterance with a fresh LM state.
 * Weight blob (float32): embed.weight [V,E]; per layer l: rnn.weight_ih_l{l} [4H,I], rnn.weight_hh_l{l}
 * [4H,H], rnn.bias_ih_l{l} [4H], rnn.bias_hh_l{l} [4H] (torch gate order i,f,g,o); linear.weight [V,H]
 * (equal to embed.weight when tied, lm.py:27-29); linear.bias [V]. */
typedef struct {
    int32_t vocab, embed, hidden, layers;
    float alpha, theta, min_val;      /* lm.py:13-15: 0.1, 1.0, -10.0 */
} lasr_lm_desc;
size_t lasr_lm_weight_count(const lasr_lm_desc* d);
int lasr_attach_lm(lasr_ctx* c, const lasr_lm_desc* d, const float* weights, size_t n_weights);
/* The same LM as the reference SERVES it: load_lm (lm.py:86-100) applies maybe_quantize (utils.py:197-210) =
 * torch.quantization.quantize_dynamic({nn.LSTM, nn.Linear}, qint8).  Same weight blob; the engine quantises the weights
 * (per tensor, symmetric int8) and, per row and per matmul at run time, the activations (7 bits), accumulates in exact
 * integer arithmetic and dequantises -- the numerics of the installed torch's x86 / fbgemm engine (un-vendored upstream;
 * restated in oracle/rnnt_oracle.py and pinned there to the reference's own quantised LM).  embed, hidden <= 1024. */
int lasr_attach_lm_int8(lasr_ctx* c, const lasr_lm_desc* d, const float* weights, size_t n_weights);

/* ---- Native serving front.  Replaces the per-RPC worker threads of the reference servicer
 * (api-server.py:82-115: the per-stream frame list and window; :139: ThreadPoolExecutor(max_workers=4), one batch-1 model call per
 * stream and chunk) and the Python tick loop of libreasr_amd.server.Scheduler for the per-stream-producer form: every stream owns
 * a single-producer ring of client chunks in host memory; ONE native thread owns the engine, batches one chunk of every stream
 * that has one waiting into lasr_push_submit_rows (ring addresses, no gather), keeps up to `depth` model steps in flight
 * (<= lasr_max_inflight) and delivers every collected step's tokens to the streams' result queues.  reset_steps > 0 applies the
 * servicer's reset rule (api-server.py:44-50,131-134: after reset_steps model steps since the last reset -- 25 = 4000 ms / 160 ms
 * for the reference geometry -- the first step whose text is empty resets encoder, predictor and LM) between two model steps of
 * the stream, as the reference does; 0 = no rule.  "Text is empty" = the step emitted no token, or only tokens named by
 * lasr_front_set_empty_tokens (pieces the servicer's tokenizer decodes to "": the reference tests `y_one != ""`, :124).  Greedy decode, 16 kHz chunks of lasr_model_desc.chunk samples (the fused
 * streaming path); other client rates / frame lengths keep going through lasr_step_window.  While a front exists it is the only
 * caller of the engine's streaming entry points; lasr_front_pause hands the engine to the caller (unary Transcribe).
 * Thread-safety: one producer (push / eof) and one consumer (next) per stream, any number of streams, any threads. */
typedef struct lasr_front lasr_front;
int lasr_front_create(lasr_ctx* c, int depth, int reset_steps, lasr_front** out);
/* Shutdown in two steps.  lasr_front_stop: the front thread stops submitting; every producer blocked on a full ring and every
 * consumer blocked in lasr_front_next returns LASR_ESTATE (results already queued are still handed out); the handle stays valid,
 * so the application can join those threads.  lasr_front_destroy: collects what is in flight, closes the front's streams, joins
 * its threads, frees the handle -- no other thread may be inside a lasr_front_* call any more; the front must go before
 * lasr_destroy of its context.  (destroy without stop is fine when nothing can be blocked.) */
int lasr_front_stop(lasr_front* f);
void lasr_front_destroy(lasr_front* f);
/* Stream id = engine slot | generation << 16: an id that outlives its stream (a reader thread still holding it after close, while
 * the slot already serves the next client) names nothing -- push / eof / next / close on it return LASR_ESTATE and touch no state.
 * LASR_EFULL when no slot is free. */
int lasr_front_open(lasr_front* f, int* stream);
/* n_chunks client chunks ([n_chunks][chunk] float32, host memory) appended to the stream; copied before the call returns; blocks
 * while the stream's ring (64 chunks) is full -- until there is room, or the stream is closed / the front stopped (LASR_ESTATE). */
int lasr_front_push(lasr_front* f, int stream, const float* pcm, int n_chunks);
int lasr_front_eof(lasr_front* f, int stream);          /* no more chunks: an end-of-stream result follows the last step's */
/* Next result of the stream, in model-step order.  *flags: 1 = a model step (n_tokens new ids, possibly none), 2 = the reset rule
 * fired after this step, 4 = end of stream.  Blocks up to timeout_ms (< 0: until there is one); returns 1 on time-out,
 * LASR_EFULL if cap is too small (nothing consumed). */
int lasr_front_next(lasr_front* f, int stream, int32_t* tokens, int cap, int* n_tokens, int* flags, int timeout_ms);
/* Takes no more input (a producer blocked in lasr_front_push returns LASR_ESTATE, and close waits until it has), waits for the
 * stream's steps in flight, then frees the slot; a consumer blocked in lasr_front_next returns LASR_ESTATE. */
int lasr_front_close(lasr_front* f, int stream);
/* The reset rule's emptiness test on TEXT: ids[0..n) decode to the empty string in the servicer's tokenizer (a lone word-boundary
 * piece); a step whose tokens are all among them counts as empty, like a step without tokens.  Default (never called / n = 0):
 * only "no token".  Call before streams are opened. */
int lasr_front_set_empty_tokens(lasr_front* f, const int32_t* ids, int n);
/* Collect every step in flight and keep the front thread out of the engine until lasr_front_resume (same thread; does not nest). */
int lasr_front_pause(lasr_front* f);
int lasr_front_resume(lasr_front* f);
/* Counters since create: lasr_push_submit_rows calls, model steps submitted, rows of those steps, resets by the rule. */
int lasr_front_stats(lasr_front* f, long long* ticks, long long* steps, long long* rows, long long* resets);
const char* lasr_front_error(const lasr_front* f);      /* text of the engine error that stopped the front, if any */

/* Bench / debug / trace entry points (lasr_bench_*, lasr_debug_*, lasr_trace*, lasr_cell_prof*, lasr_overlap_probe) are not part
 * of the drop-in surface: they are declared in include/lasr_debug.h. */

#ifdef __cplusplus
}
#endif
#endif /* LASR_H */
