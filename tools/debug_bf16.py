import sys, numpy as np, torch
sys.path.insert(0, ".")
from libreasr_amd import synth
from libreasr_amd.engine import Engine
from oracle import rnnt_oracle as O
name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
cfg = synth.model_cfg(name); sd0 = synth.synth_state_dict(cfg, seed=0)
pcm = synth.synth_pcm(2, 8000, seed=1234)
feats = np.stack([O.features_offline(p) for p in pcm])
def run(tag, sd, dt):
    eng = Engine(sd, cfg, max_streams=16, dtype=dt)
    m = O.OracleTransducer(sd, cfg, operand=dt)
    f = feats[:, :1]
    out, h, c = eng.encoder(torch.as_tensor(f).cuda(), return_state=True)
    ref, st = m.encoder(f)
    print(tag, dt, "out", np.abs(out.cpu().numpy() - ref).max(),
          "h", [float(np.abs(h[l].cpu().numpy() - st[l][0]).max()) for l in range(len(st))],
          "c", [float(np.abs(c[l].cpu().numpy() - st[l][1]).max()) for l in range(len(st))])
    rng = np.random.default_rng(0); H = cfg["hidden"]
    a = rng.standard_normal((5, H)).astype(np.float32); b = rng.standard_normal((5, H)).astype(np.float32)
    logits, lp, am = eng.joint(torch.as_tensor(a).cuda(), torch.as_tensor(b).cuda())
    print(tag, dt, "joint", np.abs(logits.cpu().numpy() - m.joint_logp(a, b)[1]).max())
    toks = np.array([[2, 5, 7]], dtype=np.int32)
    hp = eng.predictor(toks).cpu().numpy(); stp = None
    for t in toks[0]:
        x, stp = m.predictor([t], stp)
    print(tag, dt, "pred", np.abs(hp[0] - x[0]).max())
    eng.close()
run("base", sd0, "bf16")
sd = dict(sd0); k = "encoder.rnn_stack.rnns.0.weight_hh_l0"; sd[k] = np.zeros_like(np.asarray(sd[k]))
run("no_hh0", sd, "bf16")
sd = dict(sd0); k = "encoder.rnn_stack.rnns.0.weight_ih_l0"; sd[k] = np.zeros_like(np.asarray(sd[k]))
run("no_ih0", sd, "bf16")
