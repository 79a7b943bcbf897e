#!/usr/bin/env python
"""
bench.py -- audio-sec/sec of the streaming RNN-T hot path on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1]): 64 concurrent 16 kHz streams per GPU, 4x1024 uni-LSTM encoder,
2xNBRC predictor (the reference's shipped predictor cell), J=1024, V=2048, greedy decode, fp32,
reference-faithful streaming call pattern: 80 ms client chunks, 3-chunk sliding window, 2-frame
Buffer => the model runs every second chunk on 2 stacked frames (api-server.py:83-115,
transforms.py:326-342,455-471, models.py:457-577).

A "step" = one pass of the hot path over one batch of synthetic input: one 5.12 s SEGMENT (64 chunks of 80 ms,
--chunks-per-step) of every stream of the rank = 327.68 audio-seconds at 64 streams; each chunk is pushed for all
streams (lasr_push_submit / lasr_step_wait) and its tokens are fetched to the host.  A segment must be longer than
the software pipeline: 18 model steps (36 chunks) are in flight, and the timed region starts and ends on an idle
GPU -- with rounds 3-5's 16-chunk segment (8 model steps) the fill and drain of the pipeline were 4 % of the
driver's 20-step region (54.4 k against 56.5 k sustained); at 64 chunks they are 1 %.  `value` does not otherwise
depend on the segment length.  Synthetic PCM is resident in HBM before the timed region.
Streams are independent: rank r owns streams [64 r, 64 r + 64), there is no data-path collective
("scaling": "weak"); torch.distributed (RCCL) is used only for the barrier and the max-over-ranks.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...                      # spawns N ranks itself (torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

STREAMS_PER_GPU = 64
CHUNK = 1280                 # 80 ms at 16 kHz
SR = 16000
PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E ~8 TB/s
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0
PRIME_CHUNKS = 24             # untimed chunks before the W warm-up steps: window fill + hipGraph instantiation
CHUNKS_PER_STEP = 64          # one step = a 5.12 s segment of every stream (longer than the 18-deep pipeline: see the docstring)
PCM_PERIOD = 320              # distinct synthetic chunks per stream (25.6 s); longer runs cycle through them
METRIC = "audio-sec/sec/GPU (16 kHz streaming RNN-T) + p50 per-chunk latency"


def dist_env():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    return rank, world, local


def dist_init(world, use_cuda):
    """One process per GPU; backend "nccl" is RCCL on ROCm, gloo for the CPU self-test."""
    # LASR_BENCH_FORCE_DIST=1: a world of ONE rank still goes through the process group (on a 1-GPU box this is the only way to
    # run the job's RCCL calls -- init, barrier, all_reduce, all_gather on device tensors -- on real hardware)
    if world == 1 and not os.environ.get("LASR_BENCH_FORCE_DIST"):
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    # LASR_BENCH_BACKEND=gloo: dry run of the multi-rank path where RCCL cannot be used (several ranks on ONE GPU)
    # (no device_id= here: binding the device at init makes the process group initialise RCCL eagerly, and the job then runs at
    #  37 k instead of 50 k audio-s/s on the MI355X -- measured at world 1, profiles/r03/r03_experiments.txt M; torch.cuda.set_device
    #  has already put this rank on its GPU, the lazily created communicator costs 1-2 %)
    backend = os.environ.get("LASR_BENCH_BACKEND", "nccl") if use_cuda else "gloo"
    if use_cuda and backend == "nccl" and os.environ.get("LASR_BENCH_EAGER_RCCL"):      # A/B: the eager initialisation (see above)
        import torch
        dist.init_process_group(backend=backend, device_id=torch.device("cuda", torch.cuda.current_device()))
    else:
        dist.init_process_group(backend=backend)
    return dist


_HOST_PG = [None]


def host_barrier(dist):
    """The barriers that bracket the timed region run on a gloo group beside the RCCL one: a barrier on the default (RCCL)
    group would create the communicator -- channels, streams, proxy threads -- right in front of t0, and an eagerly created
    communicator has been seen to put both engine streams on one hardware queue (profiles/r03/r03_experiments.txt M).  RCCL is
    still what carries the job's collectives (the max / sum all_reduce and the all_gather of the per-rank figures), after t1."""
    if dist is None:
        return
    if _HOST_PG[0] is None:
        _HOST_PG[0] = dist.new_group(backend="gloo") if dist.get_backend() != "gloo" else dist.group.WORLD
    dist.barrier(group=_HOST_PG[0])


def shard_streams(total_streams, world, rank):
    """Contiguous block partition: stream s -> rank s // (total/world)."""
    per = total_streams // world
    return list(range(rank * per, (rank + 1) * per))


def aggregate(dist, elapsed_local, units_local, device):
    """max-over-ranks wall time, sum-over-ranks units.  Returns (elapsed_max, units_total)."""
    if dist is None:
        return elapsed_local, units_local
    import torch
    t = torch.tensor([elapsed_local], dtype=torch.float64, device=device)
    u = torch.tensor([units_local], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU.
    Refuses (rc 2) when fewer than N devices are visible instead of silently measuring fewer."""
    if not os.environ.get("LASR_BENCH_SAME_GPU") and "--selftest-dist" not in argv:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            print(f"bench.py: --gpus {n} but only {have} GPU(s) visible", file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def cell_flops(cfg, layer, rows):
    H = cfg["hidden"]
    I = cfg["feat"] if layer == 0 else H
    return 2.0 * rows * 4 * H * (I + H)


def timeline_breakdown(marks):
    """Per-model-step budget of the two streams from lasr_trace marks (tools/stream_timeline.py, condensed).  Main stream: 1 push,
    3 first cell, 4 cells done, 5 model step enqueued; decode stream: 10 group reached, 11 + 100 G (+ 1000: steps admitted)
    admission done, 12 group done.  Microseconds; None when the region is too short."""
    m = [(t, us) for t, us in marks if us >= 0 and t != 20]
    steps, cur = [], {}
    for t, us in m:
        if t >= 10:
            continue
        if t == 1:
            cur.setdefault("push", []).append(us)
        elif t == 3:
            cur["c0"] = us
        elif t == 4:
            cur["c1"] = us
        elif t == 5:
            cur["end"] = us
            if "c0" in cur and "c1" in cur and len(cur.get("push", [])) >= 2:
                steps.append(cur)
            cur = {}
    steps = steps[len(steps) // 3:]
    if len(steps) < 8:
        return None
    period = float(np.mean(np.diff([q["end"] for q in steps])))
    out = {"main_period_us": round(period, 1),
           "frontend_us": round(float(np.mean([q["c0"] - q["push"][-2] for q in steps])), 1),
           "encoder_cells_us": round(float(np.mean([q["c1"] - q["c0"] for q in steps])), 1),
           "joint_half_us": round(float(np.mean([q["end"] - q["c1"] for q in steps])), 1),
           "main_idle_us": round(float(np.mean([b["push"][-2] - a["end"] for a, b in zip(steps[:-1], steps[1:])])), 1)}
    groups, g = [], {}
    for t, us in m:
        if t < 10:
            continue
        if t == 10:
            g = {"reach": us}
        elif t % 100 == 11:
            g["adm"], g["G"] = us, (t % 1000) // 100
        elif t == 12:
            g["end"] = us
            if "adm" in g and "reach" in g:
                groups.append(g)
            g = {}
    groups = [q for q in groups if q["reach"] >= steps[0]["push"][0]]
    if len(groups) >= 4:
        span = groups[-1]["end"] - groups[0]["reach"]
        busy = sum(q["end"] - q["adm"] for q in groups)
        it = sum(q["G"] for q in groups)
        out.update({"decode_us_per_iteration": round(busy / max(1, it), 1), "decode_iterations_per_step": round(it / (span / period), 2),
                    "decode_stream_busy_frac": round(busy / span, 3),
                    "decode_waiting_for_an_encoder_frac": round(sum(q["adm"] - q["reach"] for q in groups) / span, 3)})
    return out


def decode_weight_bytes(cfg, esz):
    """Bytes of predictor / joint weights ONE greedy decode iteration streams (whatever the number of rows that emitted):
    predictor layer 0 recurrent half (the input half is a per-token table), layers >= 1 both halves, the predictor half of
    the joint (W1p) and the vocabulary projection (W2)."""
    H, J, V = cfg["hidden"], cfg["joint"], cfg["vocab"]
    G = 4 if str(cfg["pred_cell"]).upper() == "LSTM" else 3
    b = G * H * H                                   # layer 0: R
    b += (cfg["pred_layers"] - 1) * 2 * G * H * H   # layers >= 1: W and R
    b += J * H + V * J
    return float(b * esz)


def flop_per_frame(cfg, n_tok):
    """algorithmic FLOP per stacked frame (80 ms of one stream), elementwise work excluded."""
    F, H, J, V, E = cfg["feat"], cfg["hidden"], cfg["joint"], cfg["vocab"], cfg["embed"]
    G = 4 if cfg["pred_cell"] == "LSTM" else 3
    enc = sum(2.0 * 4 * H * ((F if l == 0 else H) + H) for l in range(cfg["enc_layers"]))
    return enc + 2.0 * H * J + (1.0 + n_tok) * 2.0 * J * V + n_tok * (2.0 * H * J + 2.0 * E * H + cfg["pred_layers"] * 2.0 * G * H * 2 * H)


def workload_name(args, cfg, B):
    base = {"cfg2": 1, "cfg5": 4}.get(args.model)
    if args.model == "cfg2" and (args.dtype == "bf16" or args.beam > 1):
        base = 2
    tag = f"configs[{base}]" if base is not None else f"model '{args.model}' (not a BASELINE config)"
    exact = (args.model == "cfg2" and args.dtype == "f32" and args.beam == 1 and B == 64) or \
            (args.model == "cfg2" and args.dtype == "bf16" and args.beam == 4 and B == 64) or \
            (args.model == "cfg5" and args.dtype == "bf16" and args.beam == 8 and B == 128)
    if not exact and base is not None:
        tag += " variant"
    lm = getattr(args, "lm", "none")
    if lm != "none":
        tag += f" + LM shallow fusion (4x768, {lm})"
    return (f"{tag}: {B} concurrent 16 kHz streams/GPU, {cfg['enc_layers']}x{cfg['hidden']} uni-LSTM encoder, "
            f"{cfg['pred_layers']}x{cfg['pred_cell']} predictor, J={cfg['joint']}, V={cfg['vocab']}, "
            f"{'greedy' if args.beam == 1 else 'beam width ' + str(args.beam)}, "
            f"{'fp32' if args.dtype == 'f32' else 'bf16 operands / f32 accumulate'}, 80 ms chunks, "
            "3-chunk window, 2-frame buffer (model every 160 ms)")


def cpu_reference_path(cfg, sd, n_streams, n_chunks, threads=2, rows=None, gpu_steps=None):
    """The reference's CPU execution path (torch-CPU operators, batch 1 per stream, set_num_threads(2) as
    inference.py:21) on the host cores, on a bounded sample of the same synthetic workload: the FIRST n_chunks chunks of the
    job's own first streams (`rows`), so that its tokens can be laid beside what the GPU run fetched for those streams
    (`gpu_steps[r]` = the token lists of stream r's model steps from chunk 0): `tokens_equal_gpu`."""
    from libreasr_amd import synth
    from oracle import torch_cpu as TC       # baseline leg only; never on the product path
    if rows is None:
        rows = [synth.synth_pcm(1, n_chunks * CHUNK, seed=1234 + s)[0] for s in range(n_streams)]
    rows = [np.asarray(r[:n_chunks * CHUNK], dtype=np.float32) for r in rows[:n_streams]]
    n_streams = len(rows)
    TC.time_stream_path(sd, cfg, rows[:1], min(n_chunks, 12), threads=threads)          # warm-up (thread pools, mkldnn)
    dt, toks = TC.time_stream_path(sd, cfg, rows, n_chunks, threads=threads)
    check = {}
    if gpu_steps:
        n_model = max(0, (n_chunks - 2) // 2)          # window full at chunk 3, then every second chunk: model steps within n_chunks
        rows_chk = [r for r in range(min(n_streams, len(gpu_steps))) if len(gpu_steps[r]) >= n_model]
        bad = [r for r in rows_chk if [t for st in gpu_steps[r][:n_model] for t in st] != [int(t) for t in toks[r]]]
        check = {"tokens_equal_gpu": (not bad) if rows_chk else None, "rows_compared": len(rows_chk),
                 "tokens_compared": int(sum(len(toks[r]) for r in rows_chk)), "rows_differing": bad,
                 "what": "the streams' first chunks through the reference's torch-CPU path against the token lists the GPU run "
                         "fetched for the same streams and model steps"}
    return {"value": round(n_streams * n_chunks * CHUNK / SR / dt, 2), "unit": "audio-sec/sec", "cores": int(threads), **check,
            "kind": "port",
            "path": "the reference's torch-CPU execution path restated on the installed torch (oracle/torch_cpu.py: "
                    "nn.LayerNorm, nn.LSTM + BatchNorm1d per layer, NBRC as torch matmuls, Linear/tanh/Linear, log_softmax, "
                    "torch.stft front-end), batch 1 per stream, streams one after the other, torch.set_num_threads(2) "
                    "as libreasr/lib/inference.py:21; tokens pinned to the reference's goldens in tests/test_oracle.py",
            "sample": f"{n_streams} streams x {n_chunks} chunks of 80 ms ({n_streams * n_chunks * CHUNK / SR:.0f} audio-s)",
            "seconds": round(dt, 2), "tokens": int(sum(len(t) for t in toks)), "host_cores_available": os.cpu_count(),
            "reference_speed_pin": _reference_speed_pin()}


def _reference_speed_pin():
    """The port's SPEED against the reference's own transcribe_stream, measured where the reference exists (authoring container,
    oracle/time_reference.py -> profiles/r06/reference_cpu_path.json): read from the committed file, not measured here."""
    try:
        with open(os.path.join(ROOT, "profiles", "r06", "reference_cpu_path.json")) as f:
            d = json.load(f)
        return {"port_over_reference": d["port_over_reference_mean"], "runs": d["runs"], "file": "profiles/r06/reference_cpu_path.json",
                "note": "measured in the authoring container (8 cores), where /root/reference can be imported: the reference's own "
                        "Transducer.transcribe_stream + stream Pipeline against this port, same streams / torch / threads, tokens identical"}
    except Exception:
        return None


def cpu_best_effort(cfg, sd, n_streams, n_chunks):
    """SURVEY 8d (ii) "best-effort CPU": every host core, the encoder batched over all the streams of the GPU batch, the
    greedy loop per stream, torch.backends.mkldnn on and off.  Still a PORT of the reference's operators (oracle/torch_cpu.py)."""
    import torch
    from libreasr_amd import synth
    from oracle import torch_cpu as TC       # baseline leg only; never on the product path
    rows = [synth.synth_pcm(1, n_chunks * CHUNK, seed=1234 + s)[0] for s in range(n_streams)]
    # torch's intra-op pool over ALL cores of a 256-core host is slower than over a few (every small op of the greedy loop pays
    # the fork/join): the thread count is chosen by a short probe among 8 / 16 / 32 (<= the cores the box has)
    avail = os.cpu_count() or 1
    cand = sorted({min(avail, t) for t in (8, 16, 32)})
    probe = {}
    for t in cand:
        TC.time_stream_path_batched(sd, cfg, rows[:16], 6, threads=t)
        probe[t] = TC.time_stream_path_batched(sd, cfg, rows[:16], 10, threads=t)[0]
    cores = min(probe, key=probe.get)
    res = {"thread_probe_s": {str(t): round(v, 3) for t, v in probe.items()}}
    for mk in (True, False):
        with torch.backends.mkldnn.flags(enabled=mk):
            TC.time_stream_path_batched(sd, cfg, rows[:8], min(n_chunks, 8), threads=cores)      # warm-up
            dt, toks = TC.time_stream_path_batched(sd, cfg, rows, n_chunks, threads=cores)
        res["mkldnn_on" if mk else "mkldnn_off"] = {"value": round(n_streams * n_chunks * CHUNK / SR / dt, 1), "seconds": round(dt, 2),
                                                    "tokens": int(sum(len(t) for t in toks))}
    best = max((v for k, v in res.items() if k.startswith("mkldnn")), key=lambda v: v["value"])
    return {"value": best["value"], "unit": "audio-sec/sec", "cores": int(cores), "kind": "port", **res,
            "host_cores_available": avail,
            "sample": f"{n_streams} streams x {n_chunks} chunks of 80 ms, encoder batched over the {n_streams} streams "
                      f"(nn.LSTM, batch {n_streams}), front-end batched, greedy loop per stream; torch.set_num_threads({cores}) "
                      f"(fastest of {cand} in a short probe)"}


def cpu_numpy_port(cfg, sd, n_streams, n_chunks):
    """The numpy oracle (the parity checker) timed the same way: second CPU figure."""
    from libreasr_amd import synth
    from oracle import rnnt_oracle as O      # baseline leg only; never on the product path
    rows = [synth.synth_pcm(1, n_chunks * CHUNK, seed=1234 + s)[0] for s in range(n_streams)]
    m = O.OracleTransducer(sd, cfg)
    fes = [O.StreamFrontend() for _ in rows]
    decs = [m.stream_decoder() for _ in rows]
    t0 = time.perf_counter()
    for k in range(n_chunks):
        for b, row in enumerate(rows):
            o = fes[b].push(row[k * CHUNK:(k + 1) * CHUNK])
            if o is not None:
                decs[b].step(o)
    dt = time.perf_counter() - t0
    try:
        import threadpoolctl
        cores = max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    return {"value": round(n_streams * n_chunks * CHUNK / SR / dt, 2), "unit": "audio-sec/sec", "cores": int(cores),
            "kind": "port", "sample": f"{n_streams} streams x {n_chunks} chunks, numpy oracle, batch 1 per stream",
            "seconds": round(dt, 2)}


def pin_to_gpu_numa_node(local):
    """Multi-rank runs: this process (and every thread the library creates later: pump, push helpers) onto the CPUs of the NUMA node
    the rank's GPU hangs off (sysfs: /sys/bus/pci/devices/<bdf>/numa_node -> /sys/devices/system/node/nodeN/cpulist).  Returns
    (node, n_cpus) or (None, 0) when the topology does not say (single-socket hosts report -1)."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(local)
        bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            node = int(f.read().strip())
        if node < 0:
            return None, 0
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            spec = f.read().strip()
        cpus = set()
        for part in spec.split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return node, 0
        os.sched_setaffinity(0, cpus)
        return node, len(cpus)
    except Exception:
        return None, 0


def emulation_parity(eng, args, slots, pcm_host, n_rows, n_chunks):
    """bench.py --emu-parity: the first n_rows streams of the rank, reset and replayed from chunk 0 through the SYNCHRONOUS protocol,
    against the numpy oracle's emulation of the same operand type and search (the checker: oracle/parity.py).  Greedy: token lists;
    beam: the best hypothesis after every model step and the final score."""
    from oracle import parity as PR          # checker leg only; never on the product path
    rows = list(range(min(n_rows, len(slots))))
    sl = [slots[r] for r in rows]
    for s_ in sl:
        eng.reset(s_, 15)
    hist = {r: [] for r in rows}
    toks = {r: [] for r in rows}
    score = {r: 0.0 for r in rows}
    for k in range(n_chunks):
        eng.push(sl, np.ascontiguousarray(pcm_host[rows, k * CHUNK:(k + 1) * CHUNK]))
        if not eng.step(sl):
            continue
        for r, s_ in zip(rows, sl):
            t, nl, _ = eng.fetch(s_, cap=8192 if args.beam > 1 else 256)
            if args.beam > 1:
                hist[r].append(t if t else (hist[r][-1] if hist[r] else []))
                score[r] = -nl
            else:
                toks[r].append(t)
    eps = 0.03
    operand = "bf16" if args.dtype == "bf16" else "f32"
    if args.beam > 1:
        res = PR.beam_rows_vs_emulation(args.model, args.beam, pcm_host, rows, n_chunks, hist, score, eps, operand=operand)
    else:
        res = PR.greedy_rows_vs_emulation(args.model, pcm_host, rows, n_chunks, toks, eps, operand=operand)
    return {"rows": len(rows), "chunks": n_chunks, "exact": res["exact"], "tie": res["tie"], "near": res["near"], "failed": len(res["failures"]),
            "tie_margins": res["margins"] + res["near_margins"], "eps": eps, "eps_wide": res["eps_wide"],
            "against": f"numpy oracle, operand={operand}" + (f", _beam_frame spec, width {args.beam}" if args.beam > 1 else ", greedy")
                       + " (bf16 operands / beam search have no reference implementation: parity unpinned, SURVEY 8a D4 / 8c)",
            "criterion": "exact = identical at every model step; tie / near = the emulation had a decision with a margin below eps / "
                         "eps_wide at or before the first step that differs; failed = a difference no such decision explains",
            "full_size": "all rows x 48 chunks and 4 rows x 128 chunks: tests/test_gpu_round2.py, counts in profiles/r06/parity_counts.json"}


OTHER_CONFIGS = [
    ("configs[2]", ["--model", "cfg2", "--dtype", "bf16", "--beam", "4", "--streams", "64", "--steps", "4", "--warmup", "1"]),
    ("configs[4] per-GPU shape", ["--model", "cfg5", "--dtype", "bf16", "--beam", "8", "--streams", "128", "--steps", "4", "--warmup", "1"]),
]


def reproducibility_check(cfg, sd, B, dtype, beam, depth, device, n_chunks=24, runs=2):
    """The encoder side of a pipelined run (front-end, LayerNorm, cells) reads nothing the decode stream writes, so the exact per-row
    checksums the library stores behind every model step (LASR_DBG_ENCLOG, lasr_debug_enclog) must equal those of a synchronous run
    of the same input word for word -- and with beam search the best hypotheses and scores must be the same in every run.  Round 6
    found configs[4]'s shape failing this; the check rides in the line so that the driver sees it hold.  A fresh
    engine of this leg's shape, `runs` pipelined runs of `n_chunks` chunks against one synchronous run."""
    from libreasr_amd import synth
    from libreasr_amd.engine import Engine
    old = os.environ.get("LASR_DBG_ENCLOG")
    os.environ["LASR_DBG_ENCLOG"] = "64"
    try:
        eng = Engine(sd, cfg, max_streams=B, device=device, dtype=dtype, beam=beam)
    finally:
        if old is None:
            os.environ.pop("LASR_DBG_ENCLOG", None)
        else:
            os.environ["LASR_DBG_ENCLOG"] = old
    try:
        slots = [eng.open() for _ in range(B)]
        pcm = np.stack([synth.synth_pcm(1, n_chunks * CHUNK, seed=4321 + s)[0] for s in range(B)])
        cap = 8192 if beam > 1 else 64

        def one(mode):
            for sl in slots:
                eng.reset(sl, 15)
            eng.debug_enclog()
            res = []
            for k in range(n_chunks):
                x = pcm[:, k * CHUNK:(k + 1) * CHUNK]
                if mode == "sync":
                    eng.push(slots, x)
                    if eng.step(slots):
                        res.append(eng.fetch_many(slots, cap))
                    continue
                eng.push_submit(slots, x)
                while eng.pending() >= depth:
                    if eng.wait():
                        res.append(eng.fetch_many(slots, cap))
            while eng.pending():
                if eng.wait():
                    res.append(eng.fetch_many(slots, cap))
            return eng.debug_enclog(), repr(res)
        ref_log, _ = one("sync")
        logs_differ, results_differ, first = 0, 0, None
        for r in range(runs):
            log, res = one("pipe")
            if r == 0:
                first = res
            if log.shape != ref_log.shape or (log[:, :, :B] != ref_log[:, :, :B]).any():
                logs_differ += 1
            if res != first:
                results_differ += 1
        return {"model_steps_logged": int(ref_log.shape[0]), "pipelined_runs": runs, "streams": B,
                "runs_whose_encoder_side_differs_from_the_synchronous_run": logs_differ,
                "runs_whose_results_differ_from_the_first_pipelined_run": results_differ,
                "fe_lds_pad": eng.config("fe_lds_pad"),
                "what": "exact per-row checksums of x0 / c / h of every layer / output frames / pending log-mel frames / PCM ring behind every "
                        "model step (lasr_debug_enclog), pipelined against synchronous; fetched results run to run (DESIGN 5a, tests/test_gpu_race.py)"}
    finally:
        eng.close()


def other_config_legs():
    """The BASELINE configs the headline line does not run (bf16 + beam search), each as a CHILD run of this script on the same GPU
    behind the headline's own legs (the parent is idle meanwhile): same measurement code, same line; the fields a reader needs are
    copied into `other_configs`.  ~20-40 s each (import, synthetic weights, engine, 8-10 steps, the checker)."""
    legs = []
    for tag, argv in OTHER_CONFIGS:
        t0 = time.perf_counter()
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1"] + argv + ["--no-cpu-baseline", "--no-extras", "--sustained-s", "0",
                                                                                  "--other-configs", "0", "--emu-parity", "4:24"]
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=env)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if not lines:
                legs.append({"config": tag, "error": f"rc {r.returncode}: no line", "stderr_tail": r.stderr[-300:]})
                continue
            c = json.loads(lines[-1])
            rf = c.get("roofline", {})
            legs.append({"config": tag, "workload": c["config"]["workload"], "value": c["value"], "unit": c["unit"], "dtype": c["dtype"],
                         "steps": c["steps"], "warmup": c["warmup"], "ms_per_step": c["ms_per_step"],
                         "pipeline": c["config"].get("pipeline"),
                         "p50_model_chunk_ms": c["latency_ms"]["p50_model_chunk"], "p95_model_chunk_ms": c["latency_ms"]["p95_model_chunk"],
                         "selection_rounds_per_model_step": c.get("iterations_per_model_step"),
                         "roofline": {k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launch_us", "launches_timed",
                                                              "launch_us_rocprof", "launch_us_rocprof_file", "frac_rocprof", "traffic_file",
                                                              "launch_us_isolated", "weight_bytes_per_launch")},
                         "parity": c.get("parity"), "reproducibility": c.get("reproducibility"), "rc": r.returncode, "leg_seconds": round(time.perf_counter() - t0, 1),
                         "command": "python bench.py " + " ".join(cmd[2:])})
        except Exception as e:
            legs.append({"config": tag, "error": str(e)[:300]})
    return legs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--chunks-per-step", type=int, default=CHUNKS_PER_STEP,
                    help="80 ms chunks per stream in one step (default 64 = a 5.12 s segment)")
    ap.add_argument("--model", default="cfg2")
    ap.add_argument("--streams", type=int, default=STREAMS_PER_GPU)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32 = BASELINE configs[1] (the headline metric); bf16 = configs[2] arithmetic "
                         "(bf16 MFMA operands, f32 accumulate/state)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the offline and PCIe-inclusive legs")
    ap.add_argument("--host-pcm", action="store_true",
                    help="hand lasr_push_pcm HOST buffers every chunk in the MAIN timed region (a PCIe-inclusive leg is "
                         "reported beside the headline anyway)")
    ap.add_argument("--cpu-streams", type=int, default=12)
    ap.add_argument("--cpu-chunks", type=int, default=150)
    ap.add_argument("--cpu-be-chunks", type=int, default=60,
                    help="chunks per stream of the best-effort CPU leg (all host cores, encoder batched over the streams)")
    ap.add_argument("--beam", type=int, default=1, help="beam width (1 = greedy, the headline config)")
    ap.add_argument("--depth", type=int, default=None,
                    help="pipelined mode: model steps in flight before the oldest is collected (1..25); default 18 greedy (the deepest whose "
                         "p95 push->tokens stays under 5 ms over a 5 s run), beam 6")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="synchronous lasr_step_stream per chunk instead of the submit/wait software pipeline")
    ap.add_argument("--prof-steps", type=int, default=2,
                    help="steps of the extra PROFILED region behind the timed one (in-kernel clocks of the cell launches; the cells then "
                         "run as plain launches, not as the main-stream graph); used unless --cell-prof-in-timed is 1 or 2")
    ap.add_argument("--cell-prof-in-timed", type=int, default=0,
                    help="timers of the dominant kernel inside the timed region: 0 = none (default since round 5: the HIP-event pair of "
                         "mode 3 showed as two 5.9 us gaps per model step on the main stream in the kernel trace and cost the job 2 %%: "
                         "profiles/r05/r05_experiments.txt A; both timers then come from --prof-steps further steps of the same job); "
                         "3 = one HIP-event pair per model step around the cell sequence on the cells' stream; 1 = events + "
                         "per-workgroup clock stores in the cell kernels (plain launches), 2 = clocks only")
    ap.add_argument("--sustained-s", type=float, default=5.0,
                    help="N = 1 only: seconds of the same job run once more behind everything else (`sustained`: value, p50, cell us "
                         "under sustained clocks); 0 = off")
    ap.add_argument("--check-rows", type=int, default=64,
                    help="self-check: rows replayed through the synchronous protocol after the timed region (0 = off)")
    ap.add_argument("--split-push", action="store_true",
                    help="lasr_push_pcm + lasr_step_submit as two calls instead of lasr_push_submit (A/B)")
    ap.add_argument("--device-stable", type=int, default=1,
                    help="1: device pushes carry LASR_PUSH_DEVICE_STABLE (the resident PCM is never rewritten), so the chunk that "
                         "completes no model step is appended by the next call's front-end launch; 0: plain append launch (A/B)")
    ap.add_argument("--lm", choices=["none", "fp32", "int8"], default="none",
                    help="extra line: LM shallow fusion in the greedy loop (the reference's served configuration, config/testing.yaml: "
                         "lm.enable) with a synthetic 4 x 768 LM: fp32 / bf16 operands like the model, or int8-served as load_lm does")
    ap.add_argument("--neighbour", default=None, metavar="KIND:WGS:MS",
                    help="experiment: a synthetic neighbour beside the timed region (lasr_bench_neighbour) -- KIND mfma | load (HBM) | l2 | mall, WGS one-wave "
                         "workgroups, MS milliseconds from the start of the timed region (shorter than the region: its closing synchronisation would wait for the rest); the line then carries what the neighbour achieved")
    ap.add_argument("--trace", default=None, help="diagnostics: dump the two-stream mark timeline (lasr_trace) of the timed region to this file")
    ap.add_argument("--other-configs", type=int, default=1,
                    help="1 (default): when this run is the headline workload at N = 1 with its extras, short legs of BASELINE configs[2] "
                         "(cfg2 bf16 beam 4, 64 streams) and configs[4]'s per-GPU shape (cfg5 bf16 beam 8, 128 streams) run behind it, each "
                         "as a child run of this script, and go into the line as `other_configs` (value, p50, roofline, parity); 0 = off")
    ap.add_argument("--emu-parity", default=None, metavar="ROWS:CHUNKS",
                    help="checker leg (bf16 / beam have no reference): the first ROWS streams replayed from chunk 0 for CHUNKS chunks "
                         "through the synchronous protocol against the numpy oracle's emulation (oracle/parity.py): `parity` in the line")
    ap.add_argument("--selftest-dist", action="store_true",
                    help="CPU-only: exercise sharding + aggregation over gloo (no GPU work)")
    args = ap.parse_args()

    if args.depth is None:
        # Steps in flight trade latency for throughput: a deeper pipeline lets bursty rows fall behind while the others run ahead,
        # so a decode iteration (which streams the same weights for 3 rows as for 64) serves more rows -- greedy, f32 / bf16 (round 5,
        # profiles/r05/r05_experiments.txt I): 12 -> 54.0 / 93.8 k at p50 2.2 / 1.25 ms, 15 -> 55.7 / 97.5 k, 20 -> 56.6 / 100.5 k at
        # 3.5 / 1.95 ms, 25 -> 56.9 / 100.8 k at 4.35 / 2.4 ms.  Default: the deepest whose p50 AND p95 push->tokens stay under the
        # north star's 5 ms IN THE SUSTAINED LEG (round 6, VERDICT r5 item 5: 20 sat on the line -- p95 4.66 ms in the timed region,
        # 5.12 ms over 4 s; 18: 4.29 / 4.57 ms at 56.46 against 56.66 k sustained, -0.35 %; 16: 4.05 ms, -1.2 %;
        # profiles/r06/depth_sweep.txt).  Beam: a model step of selection rounds is long (configs[4]: 3 -> 20.3 k at 2.8 ms,
        # 5 -> 25.4-26.0 k at 3.7 ms, 6 -> 27.2-27.7 k at 4.2 ms p50 / 6.9 ms p95: round 6 G); 6 keeps the p50 under 5 ms.
        args.depth = 18 if args.beam == 1 else 6
    rank, world, local = dist_env()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    if os.environ.get("LASR_BENCH_SAME_GPU"):      # dry run: every rank on GPU 0
        local = 0
    if args.selftest_dist:
        import torch
        dist = dist_init(world, use_cuda=False)
        mine = shard_streams(args.streams * world, world, rank)
        el, units = aggregate(dist, 1.0 + 0.5 * rank, float(len(mine)), torch.device("cpu"))
        if rank == 0:
            print(json.dumps({"selftest": True, "world": world, "elapsed_max": el, "units_total": units,
                              "first_stream_rank0": mine[0], "n_local": len(mine)}))
        if dist is not None:
            dist.destroy_process_group()
        return

    # The contract is ONE JSON line on stdout.  RCCL / HIP runtime libraries print their warnings to stdout (RCCL's "NCCL WARN"
    # lines do, also at process-group teardown): everything this process writes to fd 1 goes to stderr from here on, and rank 0
    # writes the line to the real stdout as the last thing it does.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    from libreasr_amd import synth
    from libreasr_amd.engine import Engine
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    if not os.environ.get("LASR_BENCH_SAME_GPU") and torch.cuda.device_count() <= local:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # several ranks on one host: each onto the CPUs next to its GPU, before the library creates its threads (LASR_BENCH_NUMA_PIN=0: off)
    numa_node, numa_cpus = (pin_to_gpu_numa_node(local) if world > 1 and os.environ.get("LASR_BENCH_NUMA_PIN", "1") != "0" else (None, 0))
    dist = dist_init(world, use_cuda=True)

    cfg = synth.model_cfg(args.model)
    sd = synth.synth_state_dict(cfg, seed=0)
    B = args.streams
    eng = Engine(sd, cfg, max_streams=B, device=local, dtype=args.dtype, beam=args.beam)
    if args.lm != "none":
        eng.attach_lm(synth.synth_lm_state_dict("lm768"), int8=args.lm == "int8")
    eng_cfg = {}
    for key in ("enc_wave", "enc_u12", "main_graph", "pump_G", "la_stream", "cell_nw", "push_lazy", "pump_nap_pct", "fe_lds_pad"):
        try:
            eng_cfg[key] = eng.config(key)
        except Exception:
            pass
    my_streams = shard_streams(B * world, world, rank)
    CPS = max(1, args.chunks_per_step)
    K, W = args.steps * CPS, args.warmup * CPS            # in chunks from here on
    P = max(0, PRIME_CHUNKS - W)
    extras = rank == 0 and not args.no_extras and args.beam == 1 and not args.no_pipeline
    n_chunks = min(PCM_PERIOD, P + W + K + (2 * min(K, 640) + 192 if extras else 0) + max(0, min(args.steps, args.prof_steps)) * CPS + 4)
    # synthetic PCM for this rank's streams (seeded per global stream id), resident in HBM,
    # laid out [chunk][stream][1280] so that one step reads one contiguous block
    pcm_host = np.stack([synth.synth_pcm(1, n_chunks * CHUNK, seed=1234 + s)[0] for s in my_streams])
    pcm_dev = torch.as_tensor(pcm_host.reshape(B, n_chunks, CHUNK).transpose(1, 0, 2).copy()).to(device)
    pcm_host_chunks = np.ascontiguousarray(pcm_host.reshape(B, n_chunks, CHUNK).transpose(1, 0, 2))
    pcm_pinned_chunks = torch.from_numpy(pcm_host_chunks).pin_memory()      # PCIe-inclusive legs: pinned client buffers
    slots = [eng.open() for _ in range(B)]
    assert slots == list(range(B))

    pipelined = not args.no_pipeline            # beam > 1: the selection loop runs across chunk boundaries as well (round 3)
    FETCH_CAP = 64 if args.beam == 1 else 8192      # beam: every fetch hands out the whole current best hypothesis
    push_t = {}                                   # chunk index -> host time of its push (latency bookkeeping)
    order = []                                    # model chunks submitted and not yet collected
    NCHK = min(B, max(0, args.check_rows)) if args.beam == 1 else 0
    rec_steps = [[] for _ in range(NCHK)]         # self-check: per checked row, the token list of every model step so far
    recording = [True]
    host_us = {"push": 0.0, "submit": 0.0, "wait": 0.0, "fetch": 0.0, "n_model_steps": 0}   # host time per call kind (timed region)

    def record(tok_lists):
        if recording[0]:
            for r in range(NCHK):
                rec_steps[r].append(tok_lists[r])

    def one_step(k, lat_out=None, host=False):
        """One 80 ms chunk for every stream.  Synchronous mode: push + step + fetch.  Pipelined mode:
        push + submit (front-end and encoder of chunk k go to the GPU), then collect the tokens of
        the oldest model step once `depth` are in flight; its decode loop runs on a second HIP stream."""
        t_push = time.perf_counter()
        src = pcm_dev if not host else (pcm_pinned_chunks if host in ("pinned", "pinned_nocopy") else pcm_host_chunks)
        nocopy = host == "pinned_nocopy"
        if pipelined and not args.split_push:
            # push + submit in one call: the front-end launch of a model step appends the newest chunk itself
            before = eng.pending()
            # (the resident PCM is never rewritten: LASR_PUSH_DEVICE_STABLE holds trivially)
            eng.push_submit(slots, src[k % n_chunks], pinned_nocopy=nocopy, device_stable=(not host) and bool(args.device_stable))
            host_us["push"] += time.perf_counter() - t_push
            ntok, done = 0, 0
            if eng.pending() > before:
                order.append(k)
                push_t[k] = t_push
                host_us["n_model_steps"] += 1
            if eng.pending() >= args.depth:
                done, ntok = collect(lat_out)
            return done, ntok
        eng.push(slots, src[k % n_chunks], pinned_nocopy=nocopy)
        t1 = time.perf_counter()
        host_us["push"] += t1 - t_push
        ntok, done = 0, 0
        if not pipelined:
            if eng.step(slots):
                toks = eng.fetch_many(slots, cap=FETCH_CAP)
                record(toks)
                ntok = sum(len(t) for t in toks)
                done = 1
                if lat_out is not None:
                    lat_out.append(time.perf_counter() - t_push)
            return done, ntok
        before = eng.pending()
        eng.submit(slots)
        host_us["submit"] += time.perf_counter() - t1
        if eng.pending() > before:
            order.append(k)
            push_t[k] = t_push
            host_us["n_model_steps"] += 1
        if eng.pending() >= args.depth:
            done, ntok = collect(lat_out)
        return done, ntok

    def collect(lat_out):
        t0 = time.perf_counter()
        if not eng.wait():
            return 0, 0
        t1 = time.perf_counter()
        toks = eng.fetch_many(slots, cap=FETCH_CAP)
        host_us["wait"] += t1 - t0
        host_us["fetch"] += time.perf_counter() - t1
        record(toks)
        ntok = sum(len(t) for t in toks)
        kk = order.pop(0)
        if lat_out is not None:
            lat_out.append(time.perf_counter() - push_t.pop(kk))
        return 1, ntok

    def drain(lat_out=None):
        ntok = 0
        while pipelined and eng.pending():
            ntok += collect(lat_out)[1]
        return ntok

    region_cpu = [0.0]

    def timed_region(k0, n, lat_out, host=False, stats=None, barrier=True, before=None):
        """barrier + sync | n steps, every token on the host | sync + barrier.  Returns (elapsed, tokens).
        before: called behind the opening synchronisation (the --neighbour experiment starts its kernel there: started earlier, the
        synchronisation would wait for it and the neighbour would run alone)."""
        torch.cuda.synchronize(device)
        if dist is not None and barrier:
            host_barrier(dist)
        if before is not None:
            before()
        tokens = 0
        cpu0 = time.process_time()                 # CPU seconds of ALL threads of this process (API thread, pump, push helpers)
        t0 = time.perf_counter()
        for k in range(k0, k0 + n):
            ran, ntok = one_step(k, lat_out, host)
            tokens += ntok
            if ran and stats is not None:
                stats(eng.stats())
        tokens += drain(lat_out)
        torch.cuda.synchronize(device)
        region_cpu[0] = time.process_time() - cpu0
        if dist is not None and barrier:
            host_barrier(dist)
        return time.perf_counter() - t0, tokens

    # CPython's cyclic GC would stop this (single) host thread for tens of ms in the middle of the timed
    # region (a full collection walks every container alive in the process, deterministically at the same
    # chunk): collect now, park the survivors in the permanent generation, keep the collector off while timing
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
    for k in range(P + W):                        # P priming chunks (engine start-up), then the W warm-up steps
        one_step(k)
    drain()
    if not pipelined:
        eng.set_profiling(True)
    lat_model, enc_ms, dec_ms, fe_ms, iters = [], [], [], [], []

    def on_stats(st):
        iters.append(st["decode_iters"])
        if not pipelined:
            enc_ms.append(st["encoder_ms"]); dec_ms.append(st["decode_ms"]); fe_ms.append(st["frontend_ms"])

    # The dominant kernel is timed INSIDE the timed region with one HIP-event pair per model step around the cell sequence, on the
    # cells' stream (`launch_us_events`; with bf16 operands the cell sequence is one hipGraph replay per model step).  The kernels' own
    # durations (`launch_us`: max exit - min entry of the device wall clock over a launch's workgroups, one plain store per
    # workgroup at entry and exit) need a per-launch slot pointer, i.e. plain launches: they come from a short region of the same
    # job right behind the timed one.  (Round 2's timers used two atomics on one word per workgroup: they cost the job 10 % and
    # inflated the cell's own duration by 2 us; profiles/r03/r03_experiments.txt A.)
    if args.trace:
        eng.trace(True)
    elif args.cell_prof_in_timed:
        eng.cell_prof(args.cell_prof_in_timed)
    for k in host_us:
        host_us[k] = 0.0 if k != "n_model_steps" else 0
    if dist is not None:
        host_barrier(dist)                        # (creates the gloo group outside the timed region)
    overlap_before = eng.overlap_probe(10000) if pipelined else float("nan")      # 2 x 10 ms: do the engine's streams overlap?
    nb, nb_start = None, None
    if args.neighbour:
        kind, wgs, ms = args.neighbour.split(":")
        nb = {"kind": kind, "workgroups": int(wgs), "ms": int(ms)}
        eng.bench_neighbour(1, 1, 1); eng.bench_neighbour(0)         # (stream, buffers, code: set up outside the timed region)
        nb_start = lambda: eng.bench_neighbour({"mfma": 1, "load": 2, "l2": 3, "mall": 4}[kind], int(wgs), int(ms))
    elapsed, tokens = timed_region(P + W, K, lat_model, host=args.host_pcm, stats=on_stats, before=nb_start)
    cpu_timed = region_cpu[0]
    if nb is not None:
        nb["timed_region_ms"] = 1e3 * elapsed
        nb["achieved"] = eng.bench_neighbour(0)
        nb["unit"] = "TFLOP/s (f32 MFMA)" if nb["kind"] == "mfma" else "GB/s"
    host_timed = dict(host_us)
    recording[0] = False                          # the self-check compares everything up to the end of the timed region
    if args.trace:
        with open(args.trace, "w") as f:
            json.dump({"marks": eng.trace_read(), "elapsed_us": 1e6 * elapsed, "chunks": K}, f)
        eng.trace(False)
    k_next = P + W + K
    prof_value = None
    cell_us_total, cell_launches = (0.0, 0)
    cell_kernel_us_total, cell_kernel_launches, cell_kernel_cells = (0.0, 0, 0)
    Kp = 0
    if not args.trace:
        if args.cell_prof_in_timed in (1, 3):
            cell_us_total, cell_launches = eng.cell_prof_read()          # HIP-event pairs of the timed region
        if args.cell_prof_in_timed in (1, 2):
            Kp = K
            cell_kernel_us_total, cell_kernel_launches, cell_kernel_cells = eng.cell_prof_kernel()
        else:
            Kp = max(0, min(args.steps, args.prof_steps)) * CPS
            if Kp:                                # the kernels' own durations: a short region of the same job with the in-kernel clocks on
                eng.cell_prof(1 if args.cell_prof_in_timed == 0 else 2)
                prof_elapsed, _ = timed_region(k_next, Kp, None, host=args.host_pcm, barrier=False)
                k_next += Kp
                prof_value = Kp * B * CHUNK / SR / prof_elapsed
                if args.cell_prof_in_timed == 0:
                    cell_us_total, cell_launches = eng.cell_prof_read()
                cell_kernel_us_total, cell_kernel_launches, cell_kernel_cells = eng.cell_prof_kernel()
    eng.cell_prof(False)
    eng.set_profiling(False)

    audio_local = K * B * CHUNK / SR
    elapsed_max, audio_total = aggregate(dist, elapsed, audio_local, device)
    # per-rank figures on rank 0 (a straggler is visible in the driver's N-GPU line): value, elapsed, host time per model step
    nms = max(1, host_timed["n_model_steps"])
    overlap_after = eng.overlap_probe(10000) if pipelined else float("nan")       # ... and after the job's RCCL collectives
    mine = [float(rank), audio_local / elapsed, elapsed, 1e6 * host_timed["push"] / nms, 1e6 * host_timed["submit"] / nms,
            1e6 * host_timed["wait"] / nms, 1e6 * host_timed["fetch"] / nms, overlap_before, overlap_after, cpu_timed / max(1e-9, elapsed),
            float(-1 if numa_node is None else numa_node), float(numa_cpus), float(eng_cfg.get("pump_nap_pct", 0)),
            float(np.median(lat_model)) * 1e3 if lat_model else float("nan")]
    per_rank = [mine]
    if dist is not None:
        t = torch.tensor(mine, dtype=torch.float64, device=device)
        bufs = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(bufs, t)
        per_rank = [b.tolist() for b in bufs]

    rc_final = 0
    line = ""
    if rank == 0:
        L, H = cfg["enc_layers"], cfg["hidden"]
        bf = args.dtype == "bf16"
        n_tok = tokens / max(1, K * B) if args.beam == 1 else 0.0
        # dominant kernel: the fused LSTM-cell GEMM (k_gemm<EpiLSTM>).  In-job average launch duration from the HIP
        # events of the timed region; algorithmic work per launch = mean over the layers (layer 0 has K = feat + H)
        flops_mean = float(np.mean([cell_flops(cfg, l, B) for l in range(L)]))
        wbytes_mean = float(np.mean([(2.0 if bf else 4.0) * 4 * H * ((cfg["feat"] if l == 0 else H) + H) for l in range(L)]))
        cell_us_ev = cell_us_total / cell_launches if cell_launches else float("nan")       # HIP events: includes the launch gaps
        # the kernel's own duration (max exit - min entry of the device wall clock over its workgroups): what a kernel trace
        # (rocprofv3 --kernel-trace --stats) reports as the average duration of this kernel
        cell_us = cell_kernel_us_total / cell_kernel_launches if cell_kernel_launches else cell_us_ev
        # the encoder pass is a layer wavefront: a launch holds the independent cells of one anti-diagonal (1.6 on average
        # for 4 layers x 2 frames); algorithmic work per launch = cells per launch x the mean cell
        cells_per_launch = cell_kernel_cells / cell_kernel_launches if cell_kernel_launches else 1.0
        flops_mean *= cells_per_launch
        wbytes_mean *= cells_per_launch
        # `frac` comes from THIS run's own measurement (launch_us: in-kernel clocks over the launches of the profiled steps).  The
        # committed kernel trace of the same command (rocprofv3 --kernel-trace --stats under profiles/) is reported BESIDE it
        # (launch_us_rocprof, frac_rocprof) and never substituted: it averages warm-up, the offline / PCIe legs and the tracer's
        # own overhead
        wkey = f"{args.model}_{args.dtype}_{B}_beam{args.beam}"
        rocprof_us, rocprof_src, traffic, pmc_src = None, None, None, None
        try:
            with open(os.path.join(ROOT, "profiles", "cell_rocprof.json")) as f:
                rp = json.load(f).get(wkey)
            if rp:
                rocprof_us, rocprof_src = float(rp["avg_us"]), rp.get("file")
        except Exception:
            pass
        try:                                # HBM bytes per launch from the committed PMC passes
            with open(os.path.join(ROOT, "profiles", "cell_pmc.json")) as f:
                pm = json.load(f).get(wkey)
            if pm:
                traffic, pmc_src = int(pm["hbm_bytes_per_launch"] * cells_per_launch), pm.get("file")      # (the passes measured a one-cell launch)
        except Exception:
            pass
        dur_us = cell_us
        achieved = flops_mean / (dur_us * 1e-6) / 1e12
        job_tflops = audio_total / elapsed_max * 12.5 * flop_per_frame(cfg, n_tok) / 1e12
        out = {
            "metric": METRIC,
            "value": round(audio_total / elapsed_max, 1),
            "unit": "audio-sec/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed_max / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic" + (" (PCM handed over in host memory every chunk: PCIe-inclusive)" if args.host_pcm else ""),
            "config": {"workload": workload_name(args, cfg, B),
                       "step": f"one {CPS * 80} ms segment ({CPS} chunks of 80 ms) of each of the {B} streams = "
                               f"{CPS * B * CHUNK / SR:.2f} audio-s per GPU and step",
                       "chunks_per_step": CPS, "ms_per_chunk": round(1e3 * elapsed_max / K, 4),
                       "streams_per_gpu": B, "chunk_ms": 80, "parallelism": f"dp{world} (independent streams, no collective)",
                       "pipeline": (f"submit/wait, {args.depth} model steps in flight: encoder of later chunks on the main stream, "
                                    "one continuous greedy loop on a second stream") if pipelined else "synchronous",
                       "decode_groups": ("launched by the library's native pump thread (LASR_PUMP=0: by the API calls)"
                                         if os.environ.get("LASR_PUMP", "1") != "0" else "launched by the API calls (LASR_PUMP=0)") if pipelined else None,
                       "priming_chunks": P,
                       "engine": eng_cfg},        # lasr_debug_config: defaults + LASR_* switches as resolved at lasr_create
            "per_gpu_value": round(audio_total / elapsed_max / world, 1),
            **({"neighbour": nb, "data_note": "EXPERIMENT: a synthetic neighbour ran beside the timed region -- not a benchmark line"} if nb else {}),
            "per_rank": [{"rank": int(v[0]), "value": round(v[1], 1), "elapsed_s": round(v[2], 5),
                          "host_us_per_model_step": {"push": round(v[3], 1), "submit": round(v[4], 1), "wait_incl_spin": round(v[5], 1),
                                                     "fetch": round(v[6], 1), "busy": round(v[3] + v[4] + v[6], 1)},
                          # two 10 ms delay kernels, one per engine stream, wall time / 10 ms: ~1 = concurrent, ~2 = one hardware queue
                          "overlap_probe": {"before_timed": round(v[7], 3), "after_collectives": round(v[8], 3)},
                          "streams_overlap": bool(v[7] < 1.5) if v[7] == v[7] else None,
                          # CPU seconds of the whole process (API thread + the library's pump and push-helper threads) per second of
                          # the timed region: how many host cores this rank keeps busy (spinning included)
                          "host_cores_busy": round(v[9], 2),
                          # multi-rank runs: the NUMA node of the rank's GPU the process was pinned to (null: single rank / unknown
                          # topology), the CPUs it may run on, the pump thread's nap share (LASR_PUMP_NAP_PCT; 75 by default there)
                          "numa_node": (int(v[10]) if v[10] >= 0 else None), "cpus_pinned": int(v[11]), "pump_nap_pct": int(v[12]),
                          "p50_model_chunk_ms": round(v[13], 4) if v[13] == v[13] else None} for v in per_rank],
            "latency_ms": {"definition": "host time from lasr_push_pcm of a model chunk to its tokens on the host"
                                         + (" (pipelined: includes the queueing behind the steps in flight)" if pipelined else ""),
                           "p50_model_chunk": round(1e3 * float(np.median(lat_model)), 4) if lat_model else None,
                           "p95_model_chunk": round(1e3 * float(np.percentile(lat_model, 95)), 4) if lat_model else None,
                           "p50_per_40ms_equiv": round(0.5e3 * float(np.median(lat_model)), 4) if lat_model else None},
            "stage_ms_per_model_step": {"frontend": round(float(np.mean(fe_ms)), 4) if fe_ms else None,
                                        "encoder": round(float(np.mean(enc_ms)), 4) if enc_ms else None,
                                        "decode": round(float(np.mean(dec_ms)), 4) if dec_ms else None,
                                        "decode_iters": round(float(np.mean(iters)), 2) if iters else None},
            "tokens_per_frame": round(n_tok, 4) if args.beam == 1 else None,
            # the decode stream in bytes: every iteration streams the same predictor / joint weights through the
            # fabric beside the encoder cells, whatever the number of rows that emitted
            "iterations_per_model_step": round(float(np.mean(iters)), 3) if iters else None,
            "decode_fabric_MB_per_model_step": (round(float(np.mean(iters)) * decode_weight_bytes(cfg, 2.0 if bf else 4.0) / 1e6, 1)
                                                if iters and args.beam == 1 else None),
            "roofline": {"bound": "mfma", "kernel": (f"k_gemm<EpiLSTM> (encoder LSTM cell, {B} rows, mean over the {L} layers)" if cells_per_launch <= 1.0 else
                                                    f"k_gemm_multi<EpiLSTM> (encoder LSTM cells of one layer-wavefront diagonal, {B} rows; "
                                                    f"{cells_per_launch:.2f} cells per launch on average, mean cell over the {L} layers)"),
                         "cells_per_launch": round(cells_per_launch, 3),
                         "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                         "launch_us": round(cell_us, 3), "launches_timed": int(cell_kernel_launches or cell_launches),
                         "launch_us_rocprof": rocprof_us, "launch_us_rocprof_file": rocprof_src, "traffic_file": pmc_src,
                         "frac_basis": "launch_us: in-kernel clocks of this run (the committed trace's figure is frac_rocprof, beside it)",
                         "frac_rocprof": (round(flops_mean / (rocprof_us * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4) if rocprof_us and not bf else None),
                         "measured_in": ("launch_us and launch_us_events: the timed region itself" if args.cell_prof_in_timed in (1, 2) else
                                         f"launch_us_events: the timed region (one HIP-event pair per model step around the cell graph); "
                                         f"launch_us: {Kp // CPS} further steps of the same job right behind it, cells as plain launches "
                                         "with in-kernel clocks" if args.cell_prof_in_timed == 3 else
                                         f"both timers: {Kp // CPS} further steps of the same job right behind the timed region (`value_profiled` "
                                         "= its rate): an event record in front of and behind the cell sequence costs the main stream ~6 us each "
                                         "(profiles/r05/r05_experiments.txt A), so the timed region itself carries no timer"),
                         "value_profiled": round(prof_value, 1) if prof_value else None,
                         "timing": "in-job (next to the decode stream), every cell launch of the timed region: kernel duration = max exit - "
                                   "min entry of the device wall clock over the launch's workgroups (the quantity rocprofv3 --kernel-trace "
                                   "reports); launch_us_events = HIP-event pairs on the cells' stream around each model step's cell sequence "
                                   "/ cells (adds the launch gaps and the events' own cost)",
                         "launch_us_events": round(cell_us_ev, 3),
                         "flops_per_launch": flops_mean, "weight_bytes_per_launch": wbytes_mean,
                         "whole_job": {"tflops": round(job_tflops, 2),
                                       "frac": round(job_tflops / (PEAK_BF16_MFMA_TFLOPS if bf else PEAK_F32_MFMA_TFLOPS), 4),
                                       "note": "algorithmic FLOP of SURVEY 8d per frame (measured tokens/frame) x frames/s / MFMA peak"}},
        }
        if bf:
            # 64 rows x 2 flop / 2 B = 64 flop/B is far below the bf16 ridge (2500 TFLOP/s / 8 TB/s = 312):
            # the bf16 cell is bound by streaming its weights, so it is priced against HBM bandwidth
            gbs = wbytes_mean / (dur_us * 1e-6) / 1e9
            out["roofline"].update({"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                    "frac": round(gbs / PEAK_HBM_GBS, 4),
                                    "frac_rocprof": round(wbytes_mean / (rocprof_us * 1e-6) / 1e9 / PEAK_HBM_GBS, 4) if rocprof_us else None,
                                    "note": "algorithmic bytes = packed bf16 weights of one cell (W_ih + W_hh)"})
        try:                                                               # the kernel alone on the GPU (micro-benchmark)
            out["roofline"]["launch_us_isolated"] = round(eng.bench_cell(layer=1, iters=300), 3)
        except Exception as e:
            out["roofline"]["launch_us_isolated"] = None
        if extras:
            # PCIe-inclusive leg (SURVEY 8d "from first PCM byte available on host"): the same K steps again with every
            # chunk handed to lasr_push_pcm as a host array; reported beside the headline, never as `value`
            try:
                # (one-time costs of the host paths stay out of the timed legs: the engine's pinned + device staging rings have 64
                #  entries each (2 x 21 MB), first touched by the first 64 host pushes; the helper threads start with the first push.
                #  Round 5 warmed 16 chunks per mode, so the leg that ran FIRST -- the no-copy one -- paid the rest inside its timed
                #  region: its 39 k against 52 k was that, not the mode (tools/r06/nocopy_probe.py: 49-53 k no-copy against 47-51 k
                #  copied, either order).  Now: 96 untimed chunks per mode, the copied mode timed first.)
                timed_region(k_next, 96, None, host=True, barrier=False)
                timed_region(k_next + 96, 96, None, host="pinned_nocopy", barrier=False)
                k_next += 192
                Kh = min(K, 640)
                lat3 = []
                dt3, _ = timed_region(k_next, Kh, lat3, host=True, barrier=False)
                lat2 = []
                dt2, _ = timed_region(k_next + Kh, Kh, lat2, host="pinned_nocopy", barrier=False)
                k_next += 2 * Kh
                out["pcie_inclusive"] = {"value": round(Kh * B * CHUNK / SR / dt3, 1), "unit": "audio-sec/sec", "chunks": Kh,
                                         "p50_model_chunk_ms": round(1e3 * float(np.median(lat3)), 4) if lat3 else None,
                                         "note": "same steps, every chunk handed over as a PAGEABLE host array (what a server's receive path has): "
                                                 "copied into the engine's pinned staging ring before the call returns (helper threads), DMA'd "
                                                 "from there into a device staging entry on a copy-only stream: 328 KB per chunk",
                                         "pinned_nocopy": {"value": round(Kh * B * CHUNK / SR / dt2, 1),
                                                           "p50_model_chunk_ms": round(1e3 * float(np.median(lat2)), 4) if lat2 else None,
                                                           "note": "opt-in LASR_PUSH_PINNED_NOCOPY: the DMA reads the caller's pinned buffer, "
                                                                   "no host copy (buffer lifetime: lasr_push_consumed)"}}
            except Exception as e:
                out["pcie_inclusive"] = {"error": str(e)[:200]}
            # secondary figure: the offline path (Transcribe RPC) on whole 20.65 s utterances (the demo's length)
            try:
                n_off = 330400
                off_pcm = [torch.as_tensor(synth.synth_pcm(1, n_off, seed=5000 + s)[0]).to(device) for s in range(min(B, 64))]
                eng.transcribe_pcm(slots[:len(off_pcm)], off_pcm)              # warm-up (buffer growth)
                torch.cuda.synchronize(device)
                t_off = time.perf_counter()
                eng.transcribe_pcm(slots[:len(off_pcm)], off_pcm)
                n_off_tok = sum(len(t) for t in eng.fetch_many(slots[:len(off_pcm)], cap=2048))
                dt_off = time.perf_counter() - t_off
                out["offline"] = {"audio_sec_per_sec": round(len(off_pcm) * n_off / SR / dt_off, 1), "utterances": len(off_pcm),
                                  "seconds_each": round(n_off / SR, 2), "wall_ms": round(1e3 * dt_off, 2), "tokens": n_off_tok,
                                  "note": "lasr_transcribe_pcm: fresh state, max_iters 3, synchronous decode loop"}
            except Exception as e:                                            # never let the extra figure break the contract line
                out["offline"] = {"error": str(e)[:200]}
        if args.sustained_s > 0 and world == 1 and pipelined and not args.trace:
            # the same job for >= sustained_s seconds (the headline's timed region is tens of milliseconds: nothing in it shows the
            # clocks the chip sustains under this load): value, latency and the cell kernel's own duration (in-kernel clocks of the
            # first 4096 cell launches of the leg)
            try:
                n_s = int(np.ceil(args.sustained_s / max(1e-6, elapsed / args.steps)))
                eng.cell_prof(2)
                lat_s = []
                dt_s, _ = timed_region(k_next, n_s * CPS, lat_s, host=args.host_pcm, barrier=False)
                k_next += n_s * CPS
                cus, cl, _cc = eng.cell_prof_kernel()
                eng.cell_prof(False)
                out["sustained"] = {"seconds": round(dt_s, 3), "steps": n_s, "value": round(n_s * CPS * B * CHUNK / SR / dt_s, 1),
                                    "p50_model_chunk_ms": round(1e3 * float(np.median(lat_s)), 4) if lat_s else None,
                                    "p95_model_chunk_ms": round(1e3 * float(np.percentile(lat_s, 95)), 4) if lat_s else None,
                                    "cell_launch_us": round(cus / cl, 3) if cl else None, "cell_launches_timed": int(cl),
                                    "note": "the headline job again, back to back for this long, same process and engine"}
            except Exception as e:
                out["sustained"] = {"error": str(e)[:200]}
        if world == 1 and pipelined and not args.trace and extras:
            # where a model step's time goes on each stream: 40 further steps of the same job with the library's trace marks on
            # (event records on both streams: they cost a few microseconds each, which is why the timed region carries none)
            try:
                eng.trace(True)
                try:
                    timed_region(k_next, 640, None, host=args.host_pcm, barrier=False)
                    k_next += 640
                    tl = timeline_breakdown(eng.trace_read())
                finally:
                    eng.trace(False)
                if tl:
                    out["stream_timeline_us_per_model_step"] = {**tl, "measured_in": "the last two thirds of 640 further chunks of the same job with lasr_trace marks on both streams (event records: the marks slow the job by a few percent)"}
                    out["stage_ms_per_model_step"].update({"frontend": round(tl["frontend_us"] / 1e3, 4), "encoder": round(tl["encoder_cells_us"] / 1e3, 4),
                                                           "decode": round(tl.get("decode_us_per_iteration", 0.0) * tl.get("decode_iterations_per_step", 0.0) / 1e3, 4) or None,
                                                           "note": "pipelined protocol: from the traced steps (stream_timeline_us_per_model_step); the two stages overlap on two streams"})
            except Exception as e:
                out["stream_timeline_us_per_model_step"] = {"error": str(e)[:200]}
        if NCHK:
            # self-check: the first NCHK streams again, from their first chunk, on freshly reset slots through the SYNCHRONOUS
            # protocol (push -> step -> fetch per chunk); per model step the tokens must equal what the run above fetched
            try:
                rows = slots[:NCHK]
                for s_ in rows:
                    eng.reset(s_, 15)
                sync_steps = [[] for _ in rows]
                for k in range(P + W + K):
                    eng.push(rows, pcm_dev[k % n_chunks][:NCHK])
                    if eng.step(rows):
                        toks = eng.fetch_many(rows, cap=FETCH_CAP)
                        for r in range(NCHK):
                            sync_steps[r].append(toks[r])
                n_tok_chk = sum(len(t) for r in range(NCHK) for t in sync_steps[r])
                bad = [(r, j) for r in range(NCHK) for j in range(max(len(sync_steps[r]), len(rec_steps[r])))
                       if j >= len(sync_steps[r]) or j >= len(rec_steps[r]) or sync_steps[r][j] != rec_steps[r][j]]
                out["tokens_checked"] = int(n_tok_chk)
                out["tokens_equal"] = not bad
                out["self_check"] = {"rows": NCHK, "model_steps_per_row": len(sync_steps[0]), "chunks": P + W + K,
                                     "what": "streams 0..rows-1 replayed from chunk 0 through lasr_step_stream on reset slots; per model step "
                                             "the token lists must equal those the " + ("pipelined " if pipelined else "") + "run fetched "
                                             "(priming + warm-up + timed region)",
                                     "first_mismatch": [int(bad[0][0]), int(bad[0][1])] if bad else None}
            except Exception as e:
                out["tokens_checked"] = 0
                out["tokens_equal"] = False
                out["self_check"] = {"error": str(e)[:300]}
        if args.emu_parity and world == 1:
            # checker leg: engine (sync protocol) against the oracle's emulation of the same operand type / search
            try:
                out["parity"] = emulation_parity(eng, args, slots, pcm_host, *[int(v) for v in args.emu_parity.split(":")])
            except Exception as e:
                out["parity"] = {"error": str(e)[:300]}
        if world == 1 and pipelined and (args.emu_parity or extras):
            try:
                out["reproducibility"] = reproducibility_check(cfg, sd, B, args.dtype, args.beam, args.depth, local)
            except Exception as e:
                out["reproducibility"] = {"error": str(e)[:300]}
        if (args.other_configs and world == 1 and extras and args.model == "cfg2" and args.dtype == "f32" and args.beam == 1
                and B == STREAMS_PER_GPU and args.lm == "none" and not args.trace and not args.neighbour):
            out["other_configs"] = other_config_legs()
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (rank 0 would stall the other ranks' teardown)
            gc.enable()
            try:
                can_cmp = args.beam == 1 and args.dtype == "f32" and (P + W + K) >= args.cpu_chunks and args.cpu_chunks <= n_chunks
                out["cpu_baseline"] = cpu_reference_path(cfg, sd, args.cpu_streams, args.cpu_chunks, rows=list(pcm_host) if can_cmp else None,
                                                         gpu_steps=rec_steps if can_cmp else None)
            except Exception as e:
                out["cpu_baseline"] = {"value": None, "error": str(e)[:300]}
            try:
                out["cpu_baseline"]["numpy_port"] = cpu_numpy_port(cfg, sd, 4, 60)
            except Exception as e:
                out["cpu_baseline"]["numpy_port"] = {"error": str(e)[:200]}
            try:
                out["cpu_baseline"]["best_effort"] = cpu_best_effort(cfg, sd, B, args.cpu_be_chunks)
            except Exception as e:
                out["cpu_baseline"]["best_effort"] = {"error": str(e)[:200]}
        if dist is not None:
            out["dist"] = {"backend": dist.get_backend(), "world": world,
                           "rccl_init": "eager" if os.environ.get("LASR_BENCH_EAGER_RCCL") else "lazy",
                           "collectives": "all_reduce(MAX, SUM) of f64 and all_gather of the per-rank figures on " + dist.get_backend()
                                          + " (device tensors), after the timed region; the barriers around the timed region on a gloo "
                                            "group beside it (host_barrier)"}
        line = json.dumps(out)
        if out.get("tokens_equal") is False:
            print("bench.py: SELF-CHECK FAILED: " + json.dumps(out.get("self_check")), file=sys.stderr, flush=True)
            rc_final = 3
    eng.close()
    if dist is not None:
        host_barrier(dist)
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        os.write(real_stdout, (line + "\n").encode())
    if rc_final:
        raise SystemExit(rc_final)


if __name__ == "__main__":
    main()
