"""Mirror of libreasr/lib/models.py `Transducer` (inference surface only).

transcribe(x)            -> (text, {"alignment_score": float})                    (models.py:365-367)
transcribe_stream(stream, denumericalizer, max_iters=10, ...) -> generator of
                            (y_all: list[int], y_chunk_text: str, reset_fn)        (models.py:457-577)
Encoder / predictor / joint / greedy loop run in liblasr_hip.so; one stream slot per call."""
import torch


class Transducer:
    def __init__(self, engine, lang):
        self.engine, self.lang = engine, lang
        self.blank, self.bos = engine.desc.blank, engine.desc.bos
        self.lm = None

    def eval(self):
        return self

    def transcribe(self, x, **kwargs):
        """x: [T', 1280, 1] (or [T', 1280]) features, as produced by x_tfm(aud)[0]."""
        tokens, neg_logp, metrics = self.decode_greedy(x, **kwargs)
        return self.lang.denumericalize(tokens), metrics

    def decode(self, x, **kwargs):
        tokens, neg_logp, _ = self.decode_greedy(x, **kwargs)
        return self.lang.denumericalize(tokens), neg_logp

    def decode_greedy(self, x, max_iters=None, **kwargs):
        if max_iters is not None and max_iters != self.engine.desc.max_iters_offline:
            raise ValueError("max_iters is fixed at engine creation (max_iters_offline)")
        x = torch.as_tensor(x)
        feats = x.reshape(x.shape[0], -1)
        slot = self.engine.open()
        try:
            self.engine.transcribe_feats([slot], [feats])
            tokens, neg_logp, align = self.engine.fetch(slot)
        finally:
            self.engine.close_slot(slot)
        return tokens, neg_logp, {"alignment_score": align}

    def transcribe_stream(self, stream, denumericalizer, max_iters=10, alpha=0.3, theta=1.0):
        if max_iters != self.engine.desc.max_iters_stream:
            raise ValueError("max_iters is fixed at engine creation (max_iters_stream)")
        eng = self.engine
        slot = eng.open()                      # open == reset(): learned initial states, predictor on BOS

        def reset():                           # models.py:494-497
            eng.reset(slot, 1 | 2 | 4)

        y = []
        try:
            for chunk in stream:
                if chunk is None:              # models.py:509
                    continue
                chunk = torch.as_tensor(chunk)
                feats = chunk.reshape(1, chunk.shape[0], -1)
                eng.step_feats([slot], feats)
                y_seq, _, _ = eng.fetch(slot)
                if eng.beam > 1:               # beam search hands out the whole best hypothesis (it may change)
                    n_same = 0
                    while n_same < min(len(y), len(y_seq)) and y[n_same] == y_seq[n_same]:
                        n_same += 1
                    y, y_seq = list(y_seq), y_seq[n_same:]
                else:
                    y = y + y_seq
                yield y, denumericalizer(y_seq), reset
        finally:
            eng.close_slot(slot)
