"""Mirror of the inference transform Pipelines (libreasr/lib/transforms.py; config/testing.yaml:339-374).

x_tfm(AudioTensor [1,N])        -> tensor [1, T', 1280, 1]            (api-server.py:74-75)
x_tfm_stream(AudioTensor [1,3n]) -> tensor [n_buffer*T', 1280, 1] or None (stateful Buffer, api-server.py:114)

The arithmetic (framing, window, rFFT, mel, log, stacking) runs in the gfx950 kernels behind
lasr_logmel / lasr_stack; these classes only reproduce the call surface and the slicing / buffering
semantics (StreamPostprocess transforms.py:335-342, Buffer transforms.py:455-471)."""
import torch


class AudioTensor(torch.Tensor):
    """Tensor carrying `.sr` (fastai2_audio AudioTensor stand-in)."""

    @staticmethod
    def __new__(cls, x, sr=16000):
        t = torch.as_tensor(x).as_subclass(cls)
        t.sr = sr
        return t

    def __init__(self, x, sr=16000):
        self.sr = sr


class _Base:
    def __init__(self, engine, channels=1, target_sr=16000):
        self.engine, self.channels, self.target_sr = engine, channels, target_sr

    def _prep(self, aud):
        sr = getattr(aud, "sr", self.target_sr)
        x = torch.as_tensor(aud).as_subclass(torch.Tensor)
        if x.dim() == 1:
            x = x[None]
        x = x[: self.channels]            # ChannelCut (transforms.py:122-132)
        x = x.to(self.engine.device, torch.float32).contiguous()
        if sr != self.target_sr:          # Resample (transforms.py:135-144), per call like the reference; equal
            x = self.engine.resample(x, sr)   # rates are passed through untouched
        return x


class OfflinePipeline(_Base):
    def __call__(self, aud):
        x = self._prep(aud)
        lm = self.engine.logmel(x)                       # TransformTime
        st = self.engine.stack(lm)                       # StackDownsample
        return st.unsqueeze(-1)                          # FixDimensions -> [1, T', 1280, 1]


class StreamPipeline(_Base):
    def __init__(self, engine, n_stack=10, n_buffer=2, **kw):
        super().__init__(engine, **kw)
        self.n_stack, self.n_buffer = n_stack, n_buffer
        self.saved = []

    def __call__(self, aud):
        x = self._prep(aud)
        lm = self.engine.logmel(x)                       # [1, T, 128]
        a = lm.shape[1] // 3 + 1                         # StreamPostprocess (transforms.py:338-341)
        lm = lm[:, a:][:, : self.n_stack].contiguous()
        st = self.engine.stack(lm).unsqueeze(-1)         # [1, T', 1280, 1]
        self.saved.append(st)
        if len(self.saved) == self.n_buffer:             # Buffer (transforms.py:463-471)
            cat = torch.cat(self.saved, dim=1)
            self.saved.clear()
            return cat[0]
        return None
