"""Reader / decoder (and dropout-free encoder) for YouTokenToMe BPE model files (`*.yttm-model`).

The reference turns token ids into text with `youtokentome.BPE(model=...).decode([ids], ignore_ids=[0])`
(libreasr/lib/language.py:115-142; model shipped as `tokenizer.yttm-model` in the release archive,
libreasr/lib/model_utils.py:70-95).  youtokentome is a third-party dependency that is neither vendored
in the reference tree nor installed here (requirements.inference.txt:20, unpinned), so this module
restates its published model format and decode rule -- **parity unpinned** (no golden from the real
package can be produced in this container; the tests use a hand-made model whose expected strings
follow from the rules below).

Model file (text, what `BPEState::dump` writes):
    <n_chars> <n_rules>
    <unicode code point> <id>          x n_chars
    <x> <y> <z>                        x n_rules      (merge of token ids x, y -> new id z, in priority order)
    <unk_id> <pad_id> <bos_id> <eos_id>
Token -> code points: a char token is its code point; a merged token z is recipe[x] + recipe[y].
U+2581 ("▁") marks a word start.  decode(): special ids print as <UNK>/<PAD>/<BOS>/<EOS>, a token that
starts with ▁ prints as " " + rest, ids in `ignore_ids` are skipped, and a single leading space of the
sentence is dropped (`BaseEncoder::decode` / `id_to_subword`).
"""
from enum import Enum

SPACE_CP = 0x2581
UNK_TOKEN, PAD_TOKEN, BOS_TOKEN, EOS_TOKEN = "<UNK>", "<PAD>", "<BOS>", "<EOS>"


class OutputType(Enum):
    ID = 1
    SUBWORD = 2


class BPE:
    def __init__(self, model, n_threads=-1):
        with open(model, "r", encoding="utf-8") as f:
            tok = f.read().split()
        if len(tok) < 2:
            raise ValueError(f"{model}: not a YouTokenToMe model file")
        it = iter(tok)
        n_chars, n_rules = int(next(it)), int(next(it))
        self.char2id, self.id2char = {}, {}
        for _ in range(n_chars):
            cp, i = int(next(it)), int(next(it))
            self.char2id[cp] = i
            self.id2char[i] = cp
        self.rules = [(int(next(it)), int(next(it)), int(next(it))) for _ in range(n_rules)]
        self.unk_id, self.pad_id, self.bos_id, self.eos_id = (int(next(it)) for _ in range(4))
        self.recipe = {i: [cp] for i, cp in self.id2char.items()}
        for x, y, z in self.rules:
            if x not in self.recipe or y not in self.recipe:
                raise ValueError(f"{model}: rule ({x}, {y}) -> {z} refers to an undefined token")
            self.recipe[z] = self.recipe[x] + self.recipe[y]
        self.rank = {(x, y): (r, z) for r, (x, y, z) in enumerate(self.rules)}
        self._special = {}
        for i, name in ((self.unk_id, UNK_TOKEN), (self.pad_id, PAD_TOKEN), (self.bos_id, BOS_TOKEN), (self.eos_id, EOS_TOKEN)):
            if i >= 0:
                self._special[i] = name
        self._size = max(list(self.recipe) + list(self._special)) + 1

    # ---- vocabulary
    def vocab_size(self):
        return self._size

    def id_to_subword(self, i, replace_space=False):
        i = int(i)
        if not 0 <= i < self._size:
            raise ValueError(f"id {i} is outside of the vocabulary [0, {self._size})")
        if i in self._special:
            return self._special[i]
        cps = self.recipe[i]
        if replace_space and cps and cps[0] == SPACE_CP:
            return " " + "".join(map(chr, cps[1:]))
        return "".join(map(chr, cps))

    def vocab(self):
        return [self.id_to_subword(i) if (i in self.recipe or i in self._special) else "" for i in range(self._size)]

    def subword_to_id(self, subword):
        for i, name in self._special.items():
            if subword == name:
                return i
        cps = [ord(c) for c in subword]
        for i, r in self.recipe.items():
            if r == cps:
                return i
        return self.unk_id

    # ---- ids -> text (what the inference path uses)
    def decode(self, ids, ignore_ids=None):
        if len(ids) and not isinstance(ids[0], (list, tuple)):
            ids = [ids]
        skip = set(int(i) for i in (ignore_ids or ()))
        out = []
        for sent in ids:
            s, first = "", True
            for i in sent:
                i = int(i)
                if i in skip:
                    continue
                s += self.id_to_subword(i, replace_space=True)
                if first and s and s[0] == " ":
                    s = s[1:]
                first = False
            out.append(s)
        return out

    # ---- text -> ids (dropout-free BPE: merge the adjacent pair of lowest rule rank until none applies)
    def _encode_word(self, cps):
        toks = [self.char2id.get(cp, self.unk_id) for cp in cps]
        while len(toks) > 1:
            best = None
            for k in range(len(toks) - 1):
                r = self.rank.get((toks[k], toks[k + 1]))
                if r is not None and (best is None or r[0] < best[0]):
                    best = (r[0], k, r[1])
            if best is None:
                break
            toks[best[1]:best[1] + 2] = [best[2]]
        return toks

    def encode(self, sentences, output_type=OutputType.ID, bos=False, eos=False, reverse=False, dropout_prob=0):
        if dropout_prob:
            raise NotImplementedError("BPE-dropout is a training-time feature")
        single = isinstance(sentences, str)
        res = []
        for s in ([sentences] if single else sentences):
            ids = [self.bos_id] if bos else []
            for word in s.split():
                ids += self._encode_word([SPACE_CP] + [ord(c) for c in word])
            if eos:
                ids.append(self.eos_id)
            if reverse:
                ids = ids[::-1]
            res.append(ids if output_type == OutputType.ID else [self.id_to_subword(i) for i in ids])
        return res[0] if single else res
