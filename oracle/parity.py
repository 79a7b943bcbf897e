"""
TEST INFRASTRUCTURE ONLY (checker; never on the product path).  bf16-operand / beam parity against the numpy oracle's
emulation, row-parallel.

bf16 operands and beam search have no reference implementation (*parity unpinned*); the contract is the
oracle's operand="bf16" emulation and its `_beam_frame` spec (oracle/rnnt_oracle.py), pinned through "W = 1 == greedy" and
"f32 oracle == reference goldens".  Criterion (tests/test_gpu_round2.py): EXACT agreement with the emulation up to the first
decision whose margin IN THE EMULATION is below eps -- a rounding tie (the f32 accumulation order of the MFMA K split
differs from numpy's, so bf16 roundings of carried state can differ in the last ulp); a disagreement at a larger margin is
a failure.

The emulation is a per-row numpy loop (0.05-0.2 s per model step and row): rows are independent, so they are spread over a
pool of SPAWNED worker processes (a fork of a process that holds a HIP context is not safe); each worker rebuilds the seeded
synthetic weights once.  Used by tests/ and by bench.py's checker legs (`other_configs[*].parity`)."""
import os

import numpy as np

_MODELS = {}


def _model(name, operand, seed=0):
    from libreasr_amd import synth
    from oracle import rnnt_oracle as O
    key = (name, operand, seed)
    if key not in _MODELS:
        cfg = synth.model_cfg(name)
        _MODELS[key] = O.OracleTransducer(synth.synth_state_dict(cfg, seed=seed), cfg, operand=operand)
    return _MODELS[key]


def _beam_row(job):
    """-> (best hypothesis after every model step, per-step smallest margin, final best score)"""
    name, operand, W, pcm_row, n_chunks = job
    from oracle import rnnt_oracle as O
    fe, dec = O.StreamFrontend(), O.StreamBeamDecoder(_model(name, operand), W)
    hist = []
    for k in range(n_chunks):
        o = fe.push(pcm_row[k * 1280:(k + 1) * 1280])
        if o is not None:
            hist.append([int(t) for t in dec.step(o)[0]])
    return hist, [float(m) for m in dec.step_margin], float(dec.best()[1])


def _greedy_row(job):
    """-> (tokens, per-model-step token lists, per-model-step smallest top-1 minus top-2 logit of the step's joint evaluations)"""
    name, operand, pcm_row, n_chunks = job
    from oracle import rnnt_oracle as O
    fe, dec = O.StreamFrontend(), _model(name, operand).stream_decoder()
    steps, step_margin = [], []
    for k in range(n_chunks):
        o = fe.push(pcm_row[k * 1280:(k + 1) * 1280])
        if o is not None:
            n0, d0 = len(dec.y), len(dec.decisions)
            dec.step(o)
            steps.append([int(t) for t in dec.y[n0:]])
            ms = [float(m) for _, m in dec.decisions[d0:]]
            step_margin.append(min(ms) if ms else float("inf"))
    return [int(t) for t in dec.y], steps, step_margin


def _run(fn, jobs, workers=None):
    if workers is None:
        workers = max(1, min(16, (os.cpu_count() or 2) // 2, len(jobs)))
    if workers <= 1 or len(jobs) <= 2:
        return [fn(j) for j in jobs]
    import multiprocessing as mp
    env_keep = os.environ.get("OMP_NUM_THREADS")
    os.environ["OMP_NUM_THREADS"] = "2"             # (numpy matvecs: the rows are the parallelism)
    try:
        with mp.get_context("spawn").Pool(workers) as pool:
            return pool.map(fn, jobs, chunksize=1)
    finally:
        if env_keep is None:
            os.environ.pop("OMP_NUM_THREADS", None)
        else:
            os.environ["OMP_NUM_THREADS"] = env_keep


def _classify(rows, per_row, eps, eps_wide):
    """per_row: (row, first differing model step or None, smallest margin of the emulation's decisions up to and including it).
    exact: identical at every model step.  tie: the emulation had a decision with margin < eps by then (bf16: f32 accumulation
    order differs between the MFMA K split and numpy, so roundings of carried bf16 state can differ in the last ulp; once a
    decision flips -- or a token moves by a frame, which token lists per model step may not show at once -- the paths part for
    good).  near: eps <= margin < eps_wide (rounds 2-4's bound).  failed: no such decision: a difference nothing explains."""
    out = dict(checked=len(rows), exact=0, tie=0, near=0, margins=[], near_margins=[], failures=[], eps=eps, eps_wide=eps_wide)
    for row, bad, mg in per_row:
        if bad is None:
            out["exact"] += 1
        elif mg < eps:
            out["tie"] += 1; out["margins"].append(float(mg))
        elif mg < eps_wide:
            out["near"] += 1; out["near_margins"].append(float(mg))
        else:
            out["failures"].append((int(row), int(bad), float(mg)))
    out["margins"].sort(); out["near_margins"].sort()
    return out


def beam_rows_vs_emulation(name, W, pcm, rows, n_chunks, hist, score, eps, operand="bf16", workers=None, eps_wide=0.08):
    """hist[i] = the engine's best hypothesis after every model step of row i, score[i] its final score (rows listed in `rows`;
    pcm [B, >= n_chunks * 1280])."""
    res = _run(_beam_row, [(name, operand, W, np.asarray(pcm[i], np.float32), n_chunks) for i in rows], workers)
    per_row, score_bad = [], []
    for i, (ref_hist, step_margin, best) in zip(rows, res):
        got = hist[i]
        if len(ref_hist) != len(got):
            per_row.append((i, 0, float("inf")))
            continue
        bad = next((j for j in range(len(ref_hist)) if ref_hist[j] != [int(t) for t in got[j]]), None)
        if bad is None and abs(score[i] - best) >= 0.02 * max(1.0, abs(score[i])):
            score_bad.append((int(i), float(score[i]), float(best)))
        per_row.append((i, bad, min(step_margin[:bad + 1]) if bad is not None else 0.0))
    out = _classify(rows, per_row, eps, eps_wide)
    out["chunks"] = int(n_chunks)
    out["score_mismatch_on_identical_hypotheses"] = score_bad
    return out


def greedy_rows_vs_emulation(name, pcm, rows, n_chunks, got_steps, eps, operand="bf16", workers=None, eps_wide=0.08):
    """got_steps[i] = the engine's token lists per model step of row i over the first n_chunks chunks."""
    res = _run(_greedy_row, [(name, operand, np.asarray(pcm[i], np.float32), n_chunks) for i in rows], workers)
    per_row = []
    for i, (ref, steps, step_margin) in zip(rows, res):
        g = [[int(t) for t in st] for st in got_steps[i]]
        if len(g) != len(steps):
            per_row.append((i, 0, float("inf")))
            continue
        bad = next((j for j in range(len(steps)) if steps[j] != g[j]), None)
        per_row.append((i, bad, min(step_margin[:bad + 1]) if bad is not None else 0.0))
    out = _classify(rows, per_row, eps, eps_wide)
    out["chunks"] = int(n_chunks)
    out["tokens"] = int(sum(len(r[0]) for r in res))
    return out


def edit_distance(a, b):
    """Levenshtein distance of two token lists (token error rate = distance / len(reference))."""
    a, b = list(a), list(b)
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i] + [0] * len(b)
        for j, y in enumerate(b, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y))
        prev = cur
    return prev[-1]


def record(test, **kw):
    """Per-test parity counts -> gpurun_out/parity_counts.json (merged back by gpurun, committed under profiles/r0N/): the ratios
    must be on record, not only printed."""
    import json
    os.makedirs("gpurun_out", exist_ok=True)
    path = "gpurun_out/parity_counts.json"
    try:
        with open(path) as f:
            d = json.load(f)
    except Exception:
        d = {}
    d[test] = kw
    with open(path, "w") as f:
        json.dump(d, f, indent=1, sort_keys=True)
