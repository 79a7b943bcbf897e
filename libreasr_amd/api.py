"""`LibreASR` facade named by the north star: load / transcribe / stream on top of the fused
device path (PCM in, token ids / text out; features never leave the GPU).

    asr = LibreASR.load("en", synthetic="cfg2")
    text = asr.transcribe(pcm)                       # 1-D float32 16 kHz, or path to a .flac / .wav (any rate: resampled on the GPU)
    for text_so_far in asr.stream(chunks): ...       # 80 ms float32 chunks (bytes / arrays / tensors)
"""
import numpy as np
import torch

from .lib.inference import load_stuff
from .lib.utils import tensorize


class LibreASR:
    def __init__(self, conf, lang, model, x_tfm, x_tfm_stream):
        self.conf, self.lang, self.model = conf, lang, model
        self.x_tfm, self.x_tfm_stream = x_tfm, x_tfm_stream
        self.engine = model.engine

    @classmethod
    def load(cls, lang="en", **kw):
        return cls(*load_stuff(lang, **kw))

    @staticmethod
    def _pcm(x):
        if isinstance(x, (bytes, bytearray)):
            return tensorize(bytes(x))[0].numpy()
        if isinstance(x, str):
            from . import flac
            pcm, sr, _ = flac.decode(x)
            if sr != 16000:
                raise NotImplementedError("only 16 kHz input is in scope")
            return pcm
        if isinstance(x, torch.Tensor):
            return x.reshape(-1)
        return np.asarray(x, dtype=np.float32).reshape(-1)

    def _utterance(self, x):
        """One utterance at the model rate.  Files (.flac, .wav; first channel, as ChannelCut transforms.py:128-132) at another rate
        go through the engine's resampler, the call the servicer makes for a unary request (Resample.encodes, transforms.py:135-144)."""
        if not isinstance(x, str):
            return self._pcm(x)
        if x.lower().endswith((".wav", ".wave")):
            from . import wav
            pcm, sr, _ = wav.decode(x)
        else:
            from . import flac
            pcm, sr, _ = flac.decode(x)
        if sr != self.engine.desc.sample_rate:
            pcm = self.engine.resample(torch.as_tensor(np.ascontiguousarray(pcm)[None]).to(self.engine.device), sr)[0]
        return pcm

    def transcribe(self, audio, return_ids=False):
        """Whole utterance(s): fresh state, greedy, max_iters_offline (Transcribe RPC, api-server.py:64-80)."""
        batch = audio if isinstance(audio, (list, tuple)) else [audio]
        slots = [self.engine.open() for _ in batch]
        try:
            self.engine.transcribe_pcm(slots, [self._utterance(a) for a in batch])
            ids = [self.engine.fetch(s)[0] for s in slots]
        finally:
            for s in slots:
                self.engine.close_slot(s)
        out = ids if return_ids else [self.lang.denumericalize(i) for i in ids]
        return out if isinstance(audio, (list, tuple)) else out[0]

    def stream(self, chunks, return_ids=False):
        """One stream of client chunks (TranscribeStream RPC, api-server.py:82-134): yields the
        hypothesis so far after every model call."""
        eng = self.engine
        slot = eng.open()
        y = []
        try:
            for ch in chunks:
                pcm = self._pcm(ch)
                pcm = pcm if isinstance(pcm, torch.Tensor) else np.asarray(pcm, np.float32)
                n = eng.desc.chunk
                if pcm.shape[0] < n:              # api-client.py:40-41 pads the last slice with zeros
                    pad = np.zeros(n, np.float32)
                    pad[: pcm.shape[0]] = np.asarray(pcm)
                    pcm = pad
                eng.push([slot], pcm[None] if not isinstance(pcm, torch.Tensor) else pcm[None])
                if eng.step([slot]):
                    got = eng.fetch(slot)[0]
                    y = got if eng.beam > 1 else y + got      # beam: the whole best hypothesis
                    yield list(y) if return_ids else self.lang.denumericalize(y)
        finally:
            eng.close_slot(slot)
