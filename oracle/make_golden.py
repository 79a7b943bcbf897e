"""
TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz by running the REFERENCE's own code
(/root/reference, imported read-only through oracle/ref_fixture.py) on seeded synthetic weights
and PCM (libreasr_amd/synth.py).  Run in the authoring container:

    python -m oracle.make_golden

The fixtures travel to the GPU box; /root/reference does not.  Everything stored is an OUTPUT of
the reference (features, encoder/predictor activations, joint logits, token ids); inputs are
regenerated from seeds on both sides.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_fixture as rf          # noqa: E402
from libreasr_amd import synth                # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def t(x):
    return torch.as_tensor(np.asarray(x))


def golden_frontend():
    x_tfm, s_tfm, AT = rf.ref_transforms()
    pcm = synth.synth_pcm(2, 16000 + 937, seed=7)            # ragged length on purpose
    out = {}
    import libreasr.lib.transforms as T
    tt = [f for f in x_tfm.fs if isinstance(f, T.TransformTime)][0]
    for s in range(2):
        spec = tt(AT(t(pcm[s][None]), 16000))[0].numpy()     # [T,128]
        out[f"logmel_{s}"] = spec.astype(np.float32)
        feats = x_tfm(AT(t(pcm[s][None]), 16000))[0, :, :, 0].numpy()
        out[f"feats_{s}"] = feats.astype(np.float32)
    # silence -> log(1e-6)
    out["logmel_zero"] = tt(AT(torch.zeros(1, 3840), 16000))[0].numpy().astype(np.float32)
    # streaming front-end: api-server.py:83-115 window + x_tfm_stream, 80 ms chunks
    chunks = synth.stream_chunks(pcm[0], 1280, lead=1, tail=2)
    frames, outs, pattern = [], [], []
    for c in chunks:
        frames.append(t(c[None]))
        if len(frames) != 3:
            continue
        aud = torch.cat(frames, dim=1)
        del frames[0]
        o = s_tfm(AT(aud, 16000))
        pattern.append(0 if o is None else 1)
        if o is not None:
            outs.append(o[:, :, 0].numpy())
    out["stream_pattern"] = np.array(pattern, dtype=np.int32)
    out["stream_feats"] = np.stack(outs).astype(np.float32)   # [n_calls, 2, 1280]
    np.savez_compressed(os.path.join(OUT, "frontend.npz"), **out)
    print("frontend:", {k: v.shape for k, v in out.items()})


def golden_model(name, n_sec, n_streams, store_acts, lm_name=None, lm_int8=False, tag="", n_samples=None):
    """tag: suffix of the file name (the *_long sets: SURVEY 8d's workload length, hundreds of recurrent steps pinned to the
    reference directly instead of through GPU == oracle on long inputs and oracle == reference on short ones)."""
    cfg = synth.model_cfg(name)
    sd = synth.synth_state_dict(cfg, seed=0)
    m = rf.ref_transducer(cfg, sd)
    if lm_name:      # config.py:143-157 attaches the LM to the model; LMFuser (lm.py:43-83) uses it in both decoders
        make = rf.ref_lm_int8 if lm_int8 else rf.ref_lm          # int8: what load_lm serves (lm.py:97)
        m.lm = make(synth.lm_cfg(lm_name), synth.synth_lm_state_dict(lm_name))
    x_tfm, s_tfm, AT = rf.ref_transforms()
    pcm = synth.synth_pcm(n_streams, int(n_samples) if n_samples else int(16000 * n_sec), seed=1234)
    out = {"n_samples": np.int64(len(pcm[0])), "n_streams": np.int32(n_streams)} if tag else {}
    with torch.no_grad():
        for s in range(n_streams):
            feats = x_tfm(AT(t(pcm[s][None]), 16000))[0]                 # [T',1280,1]
            # offline greedy (models.py:369-455)
            txt, neg_logp, metrics, extra = m.decode_greedy(feats)
            toks = [int(v) for v in txt.split()] if txt else []
            if lm_name:  # what the LM changed (for the record): same weights without the LM
                m_lm, m.lm = m.lm, None
                t0 = m.decode_greedy(feats)[0]
                m.lm = m_lm
                out[f"off_tokens_nolm_{s}"] = np.array([int(v) for v in t0.split()] if t0 else [], dtype=np.int32)
            out[f"off_tokens_{s}"] = np.array(toks, dtype=np.int32)
            out[f"off_neglogp_{s}"] = np.float64(neg_logp)
            out[f"off_align_{s}"] = np.float64(metrics["alignment_score"])
            out[f"off_iters_{s}"] = np.array(extra["iters"], dtype=np.int32)
            lp = torch.stack([o.reshape(-1) for o in extra["outs"]]).numpy()   # log-softmax rows
            out[f"off_logp_first_{s}"] = lp[:6].astype(np.float32)
            if store_acts and s == 0:
                enc, st = m.encoder(feats[None], return_state=True)
                out["enc_out_0"] = enc[0].numpy().astype(np.float32)
                out["enc_h_0"] = np.stack([a[0][0, 0].numpy() for a in st]).astype(np.float32)
                out["enc_c_0"] = np.stack([a[1][0, 0].numpy() for a in st]).astype(np.float32)
                hp, ps = m.predictor(torch.LongTensor([[m.bos]]))
                hp2, ps2 = m.predictor(torch.LongTensor([[5]]), state=ps)
                out["pred_bos"] = hp[0, 0].numpy().astype(np.float32)
                out["pred_bos_5"] = hp2[0, 0].numpy().astype(np.float32)
                j = m.joint(hp2[None], enc[0, 3][None, None, None])
                out["joint_logits"] = j.reshape(-1).numpy().astype(np.float32)
            # streaming (api-server.py:83-135 + models.py:457-577), 80 ms chunks, api-client.py:32-47
            s_tfm.fs[-1].saved.clear()
            chunks = synth.stream_chunks(pcm[s], 1280, lead=1, tail=10)

            def gen():
                frames = []
                for c in chunks:
                    frames.append(t(c[None]))
                    if len(frames) != 3:
                        continue
                    aud = torch.cat(frames, dim=1)
                    del frames[0]
                    yield s_tfm(AT(aud, 16000))

            per_chunk, y_all = [], []
            for y, y_one, reset_fn in m.transcribe_stream(gen(), m.lang.denumericalize):
                per_chunk.append(len(y) - len(y_all))
                y_all = [int(v) for v in y]       # with an LM, fuse() hands back 1-element tensors (lm.py:73)
            out[f"st_tokens_{s}"] = np.array(y_all, dtype=np.int32)
            out[f"st_counts_{s}"] = np.array(per_chunk, dtype=np.int32)
            nb = sum(1 for v in extra["iters"] for _ in range(1))
            print(f"  {name} s{s}: T'={feats.shape[0]} offline tokens={len(toks)} "
                  f"evals={int(np.sum(extra['iters']))} stream tokens={len(y_all)} calls={len(per_chunk)}")
    if lm_name and lm_int8:      # a few raw LM outputs of the quantised reference LM, to pin the int8 emulation itself
        with torch.no_grad():
            lp, st = m.lm(torch.LongTensor([[5]]))
            lp2, _ = m.lm(torch.LongTensor([[7]]), st)
        out["lm_logp_tok5"] = lp.reshape(-1).numpy().astype(np.float32)
        out["lm_logp_tok5_7"] = lp2.reshape(-1).numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, f"model_{name}{'__' + lm_name if lm_name else ''}{'_int8' if lm_int8 else ''}{tag}.npz"), **out)


SERVICER_STREAMS, servicer_pcm = synth.SERVICER_STREAMS, synth.servicer_pcm


def golden_servicer(name="tiny"):
    """Message sequences of the reference's OWN ASRServicer.TranscribeStream (api-server.py:82-135: 3-frame window, char diff,
    "same diff twice" bail-out, 4 s reset rule) driven with api-client.py:32-47 chunking (one zero frame, 80 ms frames, ten zero
    frames), one stream at a time (the reference's stream Pipeline holds ONE Buffer for all streams, transforms.py:455-471)."""
    cfg = synth.model_cfg(name)
    sd = synth.synth_state_dict(cfg, seed=0)
    m = rf.ref_transducer(cfg, sd)
    x_tfm, s_tfm, AT = rf.ref_transforms()
    sv, mod, resets = rf.ref_servicer(m, s_tfm, x_tfm)
    out = {"n": np.int32(len(SERVICER_STREAMS))}
    for i, (seed, spec) in enumerate(SERVICER_STREAMS):
        pcm = servicer_pcm(seed, spec)
        s_tfm.fs[-1].saved.clear()
        del resets[:]
        reqs = [mod.ap.Audio(data=c.tobytes(), sr=16000) for c in synth.stream_chunks(pcm, 1280, lead=1, tail=10)]
        with torch.no_grad():
            msgs = [t_.data for t_ in sv.TranscribeStream(iter(reqs), None)]
            text = sv.Transcribe(mod.ap.Audio(data=pcm.tobytes(), sr=16000), None).data
        out[f"msgs_{i}"] = np.array(msgs if msgs else [""], dtype=np.str_)
        out[f"n_msgs_{i}"] = np.int32(len(msgs))
        out[f"resets_{i}"] = np.array(resets, dtype=np.int32)      # value of `steps` at each reset (>= 25 = 4 s / 160 ms)
        out[f"unary_{i}"] = np.str_(text)
        print(f"  servicer {name} stream {i}: {len(pcm) / 16000:.2f} s, {len(msgs)} messages, resets at steps {list(resets)}")
    # the same servicer code on a client that sends 100 ms frames (1600 samples): the per-window path of the engine
    # (lasr_step_window) and the servers' generic-client form are checked against this, resets included
    i = 4
    pcm = servicer_pcm(*SERVICER_STREAMS[i])
    s_tfm.fs[-1].saved.clear()
    del resets[:]
    reqs = [mod.ap.Audio(data=c.tobytes(), sr=16000) for c in synth.stream_chunks(pcm, 1600, lead=1, tail=8)]
    with torch.no_grad():
        msgs = [t_.data for t_ in sv.TranscribeStream(iter(reqs), None)]
    out["msgs100_4"] = np.array(msgs if msgs else [""], dtype=np.str_)
    out["n_msgs100_4"] = np.int32(len(msgs))
    out["resets100_4"] = np.array(resets, dtype=np.int32)
    print(f"  servicer {name} stream {i} in 100 ms frames: {len(msgs)} messages, resets at steps {list(resets)}")
    np.savez_compressed(os.path.join(OUT, f"servicer_{name}.npz"), **out)


def golden_flac():
    """Known-answer artefact in the tree: demo FLAC STREAMINFO MD5 of the decoded PCM."""
    p = os.path.join(rf.REF_ROOT, "demo", "3729-6852-0035.flac")
    if not os.path.exists(p):
        return
    from libreasr_amd import flac
    pcm, sr, md5_ok = flac.decode(p)
    feats = None
    x_tfm, _, AT = rf.ref_transforms()
    feats = x_tfm(AT(t(pcm[None]), sr))[0, :, :, 0].numpy()
    np.savez_compressed(os.path.join(OUT, "demo_flac.npz"), n_samples=np.int64(len(pcm)), sr=np.int32(sr),
                        md5_ok=np.bool_(md5_ok), feats_first=feats[:4].astype(np.float32),
                        feats_last=feats[-2:].astype(np.float32), n_frames=np.int32(feats.shape[0]),
                        pcm_head=pcm[:4096].astype(np.float32), pcm_sum=np.float64(pcm.astype(np.float64).sum()))
    print("flac:", len(pcm), sr, md5_ok, feats.shape)


def golden_speech(names=("cfg2", "ref6")):
    """configs[0]'s utterance -- the reference's demo FLAC, real speech, 330 400 samples = 20.65 s -- through the reference's OWN
    pipeline and decode on the synthetic weights: the PCM itself (int16: exactly what the FLAC holds, the
    reference's test data; STREAMINFO MD5 verified), its log-mel (TransformTime), offline + streaming tokens (api-client.py:32-59
    chunking) and the message sequence of the reference's own servicer for the same frames.  Every GPU input before round 6 was
    chirp + noise from synth.synth_pcm: low-energy / silent stretches of real audio never reached k_logmel."""
    p = os.path.join(rf.REF_ROOT, "demo", "3729-6852-0035.flac")
    from libreasr_amd import flac
    pcm, sr, md5_ok = flac.decode(p)
    assert md5_ok and sr == 16000
    i16 = np.round(pcm * 32768.0).astype(np.int16)
    assert np.array_equal(i16.astype(np.float32) / 32768.0, pcm.astype(np.float32)), "the FLAC holds 16-bit PCM"
    x_tfm, s_tfm, AT = rf.ref_transforms()
    import libreasr.lib.transforms as T
    tt = [f for f in x_tfm.fs if isinstance(f, T.TransformTime)][0]
    out = {"pcm_i16": i16, "sr": np.int32(sr), "names": np.array(names, dtype=np.str_)}
    out["logmel"] = tt(AT(t(pcm[None]), sr))[0].numpy().astype(np.float32)            # [2066, 128]
    feats_t = x_tfm(AT(t(pcm[None]), sr))[0]                                          # [258, 1280, 1]
    feats = feats_t[:, :, 0].numpy()
    out["feats_first"] = feats[:4].astype(np.float32); out["feats_last"] = feats[-2:].astype(np.float32)
    out["n_frames"] = np.int32(feats.shape[0])
    chunks = synth.stream_chunks(pcm, 1280, lead=1, tail=10)
    for name in names:
        cfg = synth.model_cfg(name)
        sd = synth.synth_state_dict(cfg, seed=0)
        m = rf.ref_transducer(cfg, sd)
        with torch.no_grad():
            txt, neg_logp, metrics, extra = m.decode_greedy(feats_t)
            toks = [int(v) for v in txt.split()] if txt else []
            out[f"{name}_off_tokens"] = np.array(toks, dtype=np.int32)
            out[f"{name}_off_neglogp"] = np.float64(neg_logp)
            out[f"{name}_off_align"] = np.float64(metrics["alignment_score"])
            s_tfm.fs[-1].saved.clear()

            def gen():
                frames = []
                for c in chunks:
                    frames.append(t(c[None]))
                    if len(frames) != 3:
                        continue
                    aud = torch.cat(frames, dim=1)
                    del frames[0]
                    yield s_tfm(AT(aud, 16000))

            per_chunk, y_all = [], []
            for y, y_one, reset_fn in m.transcribe_stream(gen(), m.lang.denumericalize):
                per_chunk.append(len(y) - len(y_all))
                y_all = [int(v) for v in y]
            out[f"{name}_st_tokens"] = np.array(y_all, dtype=np.int32)
            out[f"{name}_st_counts"] = np.array(per_chunk, dtype=np.int32)
            # the reference's own servicer on the same frames (api-server.py:82-135) and its unary Transcribe (:64-80)
            sv, mod, resets = rf.ref_servicer(m, s_tfm, x_tfm)
            s_tfm.fs[-1].saved.clear()
            reqs = [mod.ap.Audio(data=c.tobytes(), sr=16000) for c in chunks]
            msgs = [t_.data for t_ in sv.TranscribeStream(iter(reqs), None)]
            text = sv.Transcribe(mod.ap.Audio(data=pcm.astype(np.float32).tobytes(), sr=16000), None).data
            out[f"{name}_msgs"] = np.array(msgs if msgs else [""], dtype=np.str_)
            out[f"{name}_n_msgs"] = np.int32(len(msgs))
            out[f"{name}_resets"] = np.array(resets, dtype=np.int32)
            out[f"{name}_unary"] = np.str_(text)
        print(f"  speech {name}: offline {len(toks)} tokens, stream {len(y_all)} tokens in {len(per_chunk)} calls, "
              f"{len(msgs)} messages, resets {list(resets)}")
    np.savez_compressed(os.path.join(OUT, "speech_demo.npz"), **out)
    print("speech:", len(pcm), out["logmel"].shape, feats.shape)


if __name__ == "__main__":
    assert rf.available(), "/root/reference is required to generate goldens"
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["frontend", "tiny", "tiny_lstm", "cfg2", "cfg2_lstm", "ref6", "cfg5", "flac", "lm", "lm_int8", "long", "servicer", "speech"]
    if "frontend" in which:
        golden_frontend()
    if "tiny" in which:
        golden_model("tiny", 3.0, 3, True)
    if "tiny_lstm" in which:
        golden_model("tiny_lstm", 3.0, 2, True)
    if "cfg2" in which:
        golden_model("cfg2", 4.0, 2, True)
    if "cfg2_lstm" in which:
        golden_model("cfg2_lstm", 2.0, 1, True)
    if "ref6" in which:
        golden_model("ref6", 2.0, 1, False)       # the reference's shipped shape (config/testing.yaml:202-229)
    if "cfg5" in which:
        golden_model("cfg5", 2.0, 1, False)       # BASELINE configs[4] model shape, fp32
    if "lm" in which:                             # LM shallow fusion, fp32 LM
        golden_model("tiny_soft", 3.0, 3, False, lm_name="tiny_lm")
        golden_model("tiny_lstm", 3.0, 2, False, lm_name="tiny_lm_untied")
        golden_model("cfg2", 3.0, 1, False, lm_name="lm768")
    if "lm_int8" in which:                        # the LM as the reference serves it: int8 dynamic quantisation (lm.py:97)
        golden_model("tiny_soft", 3.0, 3, False, lm_name="tiny_lm", lm_int8=True)
        golden_model("tiny_lstm", 3.0, 2, False, lm_name="tiny_lm_untied", lm_int8=True)
        golden_model("cfg2", 3.0, 1, False, lm_name="lm768", lm_int8=True)
    if "long" in which:                           # SURVEY 8d's offline unit: 330 400 samples = 20.65 s, T' = 258
        golden_model("cfg2", 0, 3, False, tag="_long", n_samples=330400)
        golden_model("ref6", 10.0, 1, False, tag="_long")
        golden_model("cfg5", 10.0, 1, False, tag="_long")
    if "servicer" in which:
        golden_servicer("tiny")
    if "flac" in which:
        try:
            golden_flac()
        except ImportError as e:
            print("flac golden skipped:", e)
    if "speech" in which:
        golden_speech()
