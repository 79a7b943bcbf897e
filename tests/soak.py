#!/usr/bin/env python3
"""Soak harness for intermittent parity failures.

Replays the driver's situation -- the GPU tests that run before
tests/test_gpu_parity.py::test_streaming_matches_reference[tiny-3.0-3] in the SAME process (shared engine
cache) -- and then loops streaming scenarios against the reference goldens / the numpy oracle, logging every
mismatch as (iteration, stream, model step, token index).  On the first mismatching model step of an iteration
the resident state of the engine (PCM ring, pending log-mel frames, LayerNorm'ed features, encoder state and
output, joint halves, predictor state: lasr_debug_read) is compared with the oracle's values for that step, so
the stage that went wrong is named instead of guessed.

  python tests/soak.py --preamble parity --iters 1000            # the driver's failing case
  python tests/soak.py --scenario pipe --iters 300               # 4 streams out of phase, pipelined protocol
  LASR_SPIN=0 LASR_POISON=1 python tests/soak.py ...             # bisect switches (read by liblasr_hip.so)

Output: one JSON line per mismatch on stdout and a summary line at the end (also written to --out)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def log(**kw):
    print(json.dumps(kw), flush=True)


def run_preamble(kind):
    """The tests the driver runs before the one that failed, in this process."""
    import pytest
    if kind == "none":
        return None
    if kind == "full":          # every GPU test file that sorts before test_gpu_parity.py, then its op-level / offline tests
        args = ["-q", "-m", "gpu", "-p", "no:cacheprovider",
                os.path.join(ROOT, "tests", "test_gpu_api.py"), os.path.join(ROOT, "tests", "test_gpu_beam.py"),
                os.path.join(ROOT, "tests", "test_gpu_bf16.py"), os.path.join(ROOT, "tests", "test_gpu_lm.py")]
        rc = pytest.main(args)
        log(event="preamble_files", rc=int(rc))
    keep = "test_logmel or test_stack_layout or test_encoder or test_predictor_and_joint or test_offline_transcribe"
    rc = pytest.main(["-q", "-m", "gpu", "-p", "no:cacheprovider", "-k", keep, os.path.join(ROOT, "tests", "test_gpu_parity.py")])
    log(event="preamble_parity", rc=int(rc))
    return rc


class OracleTrace:
    """The oracle's per-model-step values of one stream (computed lazily, cached)."""

    def __init__(self, m, chunks):
        from oracle import rnnt_oracle as O
        self.O, self.m, self.chunks = O, m, chunks
        self.steps = None

    def build(self):
        if self.steps is not None:
            return self.steps
        O, m = self.O, self.m
        fe, dec = O.StreamFrontend(), m.stream_decoder()
        steps = []
        for k, ch in enumerate(self.chunks):
            o = fe.push(ch)
            if o is None:
                continue
            st_before = dec.enc_state
            enc, st_after = m.encoder(o[None], st_before)
            x_ln = O.layer_norm(o, m.ln_w, m.ln_b)
            toks = dec.step(o)
            steps.append(dict(chunk=k, feats=o, x_ln=x_ln, enc=enc[0], state=st_after, tokens=list(toks),
                              pe=(enc[0] @ m.j0_w[:, m.H:].T).astype(np.float32),
                              window=np.concatenate(self.chunks[max(0, k - 3):k + 1])))
        self.steps = steps
        return steps


def maxerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max())


def dump_state(eng, slot, tr, j):
    """Engine state after model step j of `slot` against the oracle's."""
    st = tr.build()[j]
    d = eng.desc
    out = {}
    ints = eng.debug_read("ints")
    out["ints"] = {k: int(ints[i, slot]) for i, k in enumerate(
        ["ring_pos", "h_ring_pos", "n_chunks", "n_pend", "T_row", "t_idx", "token", "emit"])}
    # PCM ring: the newest chunk sits at ring_pos - 1
    ring = eng.debug_read("ring")[slot].reshape(-1, d.chunk)
    pos = out["ints"]["ring_pos"]
    nr = ring.shape[0]
    have = np.concatenate([ring[(pos - nr + i) % nr] for i in range(nr)])
    want = st["window"][-nr * d.chunk:]
    out["ring"] = maxerr(have[-len(want):], want)
    # pending log-mel frames of the step: feats[t'][m * n_stack + k] = mel[k][m]
    pend = eng.debug_read("pend")[slot].reshape(d.n_buffer, d.n_stack, d.n_mels)
    mel = st["feats"].reshape(d.n_buffer, d.n_mels, d.n_stack).transpose(0, 2, 1)
    out["pend"] = [maxerr(pend[t], mel[t]) for t in range(d.n_buffer)]
    out["x0"] = [maxerr(eng.debug_read("x0", t)[slot], st["x_ln"][t]) for t in range(d.n_buffer)]
    out["enc_out"] = [maxerr(eng.debug_read("enc_out", t)[slot], st["enc"][t]) for t in range(d.n_buffer)]
    out["pe"] = [maxerr(eng.debug_read("pe", t)[slot], st["pe"][t]) for t in range(d.n_buffer)]
    out["enc_h"] = [maxerr(eng.debug_read("enc_h", l)[slot], st["state"][l][0][0]) for l in range(d.enc_layers)]
    out["enc_c"] = [maxerr(eng.debug_read("enc_c", l)[slot], st["state"][l][1][0]) for l in range(d.enc_layers)]
    return out


HOST_PCM = [False]          # --host-pcm: chunks are handed over as numpy arrays (pageable host memory: staging ring + DMA path)


def _buf(T, x):
    return np.ascontiguousarray(x) if HOST_PCM[0] else T.dev(x)


def scenario_sync(T, name, n_sec, n_streams, iters, dump, golden_dir, stop_after):
    """Body of test_streaming_matches_reference, `iters` times."""
    from libreasr_amd import synth
    eng, m, cfg = T.engine(name)
    g = np.load(os.path.join(golden_dir, f"model_{name}.npz"))
    pcm = synth.synth_pcm(n_streams, int(16000 * n_sec), seed=1234)
    chunks = [synth.stream_chunks(p, 1280, lead=1, tail=10) for p in pcm]
    want = [list(map(int, g[f"st_tokens_{s}"])) for s in range(n_streams)]
    traces = [OracleTrace(m, chunks[s]) for s in range(n_streams)]
    want_steps = None
    if dump:
        want_steps = [[st["tokens"] for st in tr.build()] for tr in traces]
        for s in range(n_streams):
            assert sum(want_steps[s], []) == want[s], "oracle != golden"
    bad = 0
    for it in range(iters):
        slots = [eng.open() for _ in range(n_streams)]
        try:
            got = [[] for _ in slots]
            step_no = 0
            dumped = False
            for k in range(len(chunks[0])):
                eng.push(slots, _buf(T, np.stack([c[k] for c in chunks])))
                ran = eng.step(slots)
                for s, slot in enumerate(slots):
                    t, _, _ = eng.fetch(slot)
                    got[s] += t
                    if ran and dump and not dumped and t != want_steps[s][step_no]:
                        dumped = True
                        log(event="step_mismatch", iter=it, stream=s, step=step_no, got=t, want=want_steps[s][step_no],
                            state=dump_state(eng, slot, traces[s], step_no))
                if ran:
                    step_no += 1
            for s in range(n_streams):
                if got[s] != want[s]:
                    bad += 1
                    first = next((i for i, (a, b) in enumerate(zip(got[s], want[s])) if a != b), min(len(got[s]), len(want[s])))
                    tail_equal = got[s][first + 1:] == want[s][first + 1:]
                    log(event="mismatch", scenario="sync", iter=it, stream=s, index=first,
                        got=got[s][first] if first < len(got[s]) else None, want=want[s][first] if first < len(want[s]) else None,
                        len_got=len(got[s]), len_want=len(want[s]), rest_equal=tail_equal)
        finally:
            for slot in slots:
                eng.close_slot(slot)
        if stop_after and bad >= stop_after:
            return bad, it + 1
    return bad, iters


def scenario_pipe(T, name, n_sec, n_streams, iters, golden_dir, depth, stop_after, fused):
    """Streams started out of phase on the pipelined protocol (push + submit / push_submit, `depth` steps in flight) against
    the reference goldens per stream."""
    from libreasr_amd import synth
    eng, m, cfg = T.engine(name)
    g = np.load(os.path.join(golden_dir, f"model_{name}.npz"))
    n_gold = sum(1 for k in g.files if k.startswith("st_tokens_"))
    pcm = synth.synth_pcm(n_gold, int(16000 * n_sec), seed=1234)
    chunks = [synth.stream_chunks(pcm[s % n_gold], 1280, lead=1, tail=10) for s in range(n_streams)]
    want = [list(map(int, g[f"st_tokens_{s % n_gold}"])) for s in range(n_streams)]
    n_chunks = len(chunks[0])
    bad = 0
    for it in range(iters):
        slots = [eng.open() for _ in range(n_streams)]
        start = [(s * 1 + it) % 4 for s in range(n_streams)]        # stream s starts `start[s]` chunks late
        got = [[] for _ in slots]
        try:
            def collect():
                if eng.wait():
                    for s, t in enumerate(eng.fetch_many(slots, 64)):
                        got[s] += t
            for k in range(n_chunks + max(start)):
                act = [s for s in range(n_streams) if 0 <= k - start[s] < n_chunks]
                if not act:
                    continue
                sl = [slots[s] for s in act]
                buf = _buf(T, np.stack([chunks[s][k - start[s]] for s in act]))
                if fused:
                    eng.push_submit(sl, buf)
                else:
                    eng.push(sl, buf)
                    eng.submit(sl)
                while eng.pending() >= depth:
                    collect()
            while eng.pending():
                collect()
            for s in range(n_streams):
                if got[s] != want[s]:
                    bad += 1
                    first = next((i for i, (a, b) in enumerate(zip(got[s], want[s])) if a != b), min(len(got[s]), len(want[s])))
                    log(event="mismatch", scenario="pipe", iter=it, stream=s, index=first, start=start[s],
                        got=got[s][first] if first < len(got[s]) else None, want=want[s][first] if first < len(want[s]) else None,
                        len_got=len(got[s]), len_want=len(want[s]), rest_equal=got[s][first + 1:] == want[s][first + 1:])
        finally:
            for slot in slots:
                eng.close_slot(slot)
        if stop_after and bad >= stop_after:
            return bad, it + 1
    return bad, iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preamble", choices=["none", "parity", "full"], default="parity")
    ap.add_argument("--scenario", choices=["sync", "pipe", "both"], default="sync")
    ap.add_argument("--model", default="tiny")
    ap.add_argument("--n-sec", type=float, default=3.0)
    ap.add_argument("--streams", type=int, default=3)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--no-dump", action="store_true")
    ap.add_argument("--host-pcm", action="store_true", help="push numpy arrays (pageable host memory) instead of device tensors")
    ap.add_argument("--stop-after", type=int, default=0, help="stop after this many mismatching streams (0: run all iterations)")
    ap.add_argument("--out", default="")
    a = ap.parse_args()

    HOST_PCM[0] = a.host_pcm
    import test_gpu_parity as T          # the driver's module: shared engine cache, same helper functions
    golden_dir = os.path.join(ROOT, "tests", "golden")
    t0 = time.time()
    run_preamble(a.preamble)
    summary = dict(event="summary", preamble=a.preamble, model=a.model, host_pcm=a.host_pcm, streams=a.streams, env={k: v for k, v in os.environ.items() if k.startswith(("LASR_", "AMD_", "HSA_", "GPU_", "HIP_"))})
    if a.scenario in ("sync", "both"):
        bad, n = scenario_sync(T, a.model, a.n_sec, a.streams, a.iters, not a.no_dump, golden_dir, a.stop_after)
        summary.update(sync_bad_streams=bad, sync_iters=n)
    if a.scenario in ("pipe", "both"):
        for fused in (False, True):
            bad, n = scenario_pipe(T, a.model, a.n_sec, max(4, a.streams), a.iters, golden_dir, a.depth, a.stop_after, fused)
            summary.update({f"pipe_bad_streams_{'fused' if fused else 'split'}": bad, f"pipe_iters_{'fused' if fused else 'split'}": n})
    summary["seconds"] = round(time.time() - t0, 1)
    log(**summary)
    if a.out:
        with open(a.out, "a") as f:
            f.write(json.dumps(summary) + "\n")
    bad_total = sum(v for k, v in summary.items() if "bad" in k)
    sys.exit(1 if bad_total else 0)


if __name__ == "__main__":
    main()
