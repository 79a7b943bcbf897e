"""Mirror of libreasr/lib/language.py (inference subset): token ids -> text.

The reference decodes with a YouTokenToMe BPE model (language.py:115-155).  Neither the package
nor a tokenizer model file is available here (they ship with the model release, docs/docs.md:139),
so parity is defined on token ids; `IdLanguage` is the stand-in and `TokenizedLanguage` is used
when youtokentome and a model file exist."""


class IdLanguage:
    """denumericalize(ids) -> space-separated ids, ignoring blank (ignore_ids=[0], language.py:139)."""

    def __init__(self, ignore_ids=(0,)):
        self.ignore_ids = set(ignore_ids)

    def denumericalize(self, ids):
        return " ".join(str(int(i)) for i in ids if int(i) not in self.ignore_ids)

    def __len__(self):
        return 0


class TokenizedLanguage:
    def __init__(self, model_file, ignore_ids=(0,)):
        import youtokentome as yttm      # not installed in this image; raises ImportError
        self.bpe = yttm.BPE(model=model_file)
        self.ignore_ids = list(ignore_ids)

    def denumericalize(self, ids):
        return self.bpe.decode([int(i) for i in ids], ignore_ids=self.ignore_ids)[0]


def get_language(model_file=None):
    if model_file:
        try:
            return TokenizedLanguage(model_file)
        except ImportError:
            pass
    return IdLanguage()
