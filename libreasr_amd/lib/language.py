"""Mirror of libreasr/lib/language.py (inference subset): token ids -> text.

The reference decodes with a YouTokenToMe BPE model (language.py:115-155).  The package is not
installed here and no tokenizer model ships with the reference tree (it comes with the model
release, docs/docs.md:139): `TokenizedLanguage` uses the real package when it is importable and
otherwise the restated model reader in `yttm.py` (parity unpinned, see there); without a model
file parity is defined on token ids and `IdLanguage` is the stand-in."""


class IdLanguage:
    """denumericalize(ids) -> space-separated ids, ignoring blank (ignore_ids=[0], language.py:139)."""

    def __init__(self, ignore_ids=(0,)):
        self.ignore_ids = set(ignore_ids)

    def denumericalize(self, ids):
        return " ".join(str(int(i)) for i in ids if int(i) not in self.ignore_ids)

    def __len__(self):
        return 0


class TokenizedLanguage:
    """language.py:115-155: numericalize / denumericalize / get_idx / get_token / len on a YTTM BPE model."""
    SOS, EOS = "<s>", "</s>"                         # language.py:20-21 (stripped before encoding)

    def __init__(self, model_file="tmp/tokenizer.yttm-model", ignore_ids=(0,)):
        try:
            import youtokentome as yttm          # the reference's dependency, when present
        except ImportError:
            from . import yttm                   # restated model reader / decoder
        self._yttm = yttm
        self.mf = model_file
        self.tokenizer = yttm.BPE(model=model_file)
        self.ignore_ids = list(ignore_ids)

    def numericalize(self, text, sos=False, dropout=0):
        text = text.lower().strip().replace(self.SOS, "").replace(self.EOS, "")
        return self.tokenizer.encode([text], output_type=self._yttm.OutputType.ID, dropout_prob=dropout)[0]

    def denumericalize(self, nummed, strip_zeros=True):
        if not isinstance(nummed, (list, tuple)):
            nummed = [nummed]
        return self.tokenizer.decode([[int(i) for i in nummed]], ignore_ids=self.ignore_ids)[0]

    def get_idx(self, tok):
        return self.numericalize(tok)[0]

    def get_token(self, num, strip_zeros=False):
        return self.denumericalize(num)[0]

    def __len__(self):
        return self.tokenizer.vocab_size()


def get_language(model_file=None):
    if model_file:
        return TokenizedLanguage(model_file)
    return IdLanguage()
