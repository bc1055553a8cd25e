"""CLIP byte-pair tokenisation of the label prompts (ref src/models.py:155-166: `_processor(text=[to_encode], ...)`).

The reference lets HF's `AutoProcessor` (a `CLIPTokenizer` under `OwlViTProcessor`) turn its three prompts per label into
`input_ids`.  Neither box can download the CLIP vocabulary, so it is not shipped: a maintainer who has the HF cache passes the
two files every CLIP checkpoint carries (`vocab.json`, `merges.txt`) to `load_model(..., vocab=, merges=)` and gets the reference's
query-bank initialisation from the unchanged `main.py:42` call.  The algorithm is the published one (openai/CLIP `simple_tokenizer.py`,
HF `tokenization_clip.py`): NFC + whitespace collapse + lower-casing, the CLIP split pattern, GPT-2's byte -> printable-unicode map,
greedy lowest-rank pair merging with `</w>` on a word's last symbol, `<|startoftext|>` / `<|endoftext|>` around the ids, and
`OwlViTProcessor`'s padding (`padding="max_length"`, 16 positions, pad id 0).  tests/test_tokenizer.py checks it id for id against
`transformers.CLIPTokenizer` built from the same (synthetic) vocabulary files.
"""
import json
import unicodedata
from functools import lru_cache

import numpy as np
import regex

_PAT = regex.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", regex.IGNORECASE)


@lru_cache()
def bytes_to_unicode():
    """GPT-2's reversible byte -> unicode-character map (printable stand-ins for the 68 bytes that are whitespace / control)."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


class ClipBPE:
    def __init__(self, vocab, merges, max_length: int = 16, pad_id: int = 0):
        """vocab: path to vocab.json (token -> id) or the dict itself; merges: path to merges.txt (first line = version header) or a list of pairs."""
        if isinstance(vocab, (str, bytes)) or hasattr(vocab, "__fspath__"):
            with open(vocab, encoding="utf-8") as f:
                vocab = json.load(f)
        if isinstance(merges, (str, bytes)) or hasattr(merges, "__fspath__"):
            with open(merges, encoding="utf-8") as f:
                lines = f.read().strip().split("\n")
            if lines and lines[0].startswith("#"):
                lines = lines[1:]
            merges = [tuple(ln.split()) for ln in lines if ln.strip()]
        self.encoder = dict(vocab)
        self.ranks = {tuple(m): i for i, m in enumerate(merges)}
        self.byte_encoder = bytes_to_unicode()
        self.max_length, self.pad_id = max_length, pad_id
        for tok in ("<|startoftext|>", "<|endoftext|>"):
            if tok not in self.encoder:
                raise ValueError(f"ClipBPE: vocabulary lacks {tok}")
        self.bos, self.eos = self.encoder["<|startoftext|>"], self.encoder["<|endoftext|>"]
        self.unk = self.encoder.get("<|endoftext|>")
        self._cache = {"<|startoftext|>": ("<|startoftext|>",), "<|endoftext|>": ("<|endoftext|>",)}      # a literal special token in a prompt keeps its id

    def _bpe(self, token: str):
        if token in self._cache:
            return self._cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        while len(word) > 1:
            pairs = set(zip(word[:-1], word[1:]))
            best = min(pairs, key=lambda p: self.ranks.get(p, float("inf")))
            if best not in self.ranks:
                break
            a, b = best
            out, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and word[i] == a and word[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(word[i])
                    i += 1
            word = tuple(out)
        self._cache[token] = word
        return word

    def encode(self, text: str):
        """ids of one prompt WITHOUT the start / end tokens."""
        text = unicodedata.normalize("NFC", text)
        text = regex.sub(r"\s+", " ", text).strip().lower()
        ids = []
        for tok in regex.findall(_PAT, text):
            tok = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder.get(piece, self.unk) for piece in self._bpe(tok))
        return ids

    def __call__(self, prompts) -> np.ndarray:
        """[N, max_length] int64: <|startoftext|> ids <|endoftext|>, padded with `pad_id` (what `OwlViTProcessor(text=...)` returns as `input_ids`);
        a prompt that does not fit is truncated with the end token kept last, as HF does."""
        out = np.full((len(prompts), self.max_length), self.pad_id, np.int64)
        for n, ptxt in enumerate(prompts):
            ids = [self.bos] + self.encode(ptxt)
            ids = ids[: self.max_length - 1] + [self.eos]
            out[n, : len(ids)] = ids
        return out


def label_prompts(labelmap):
    """ref src/models.py:155-159: three prompts per label, class-major."""
    to_encode = []
    for label in labelmap.values():
        to_encode.append(label)
        to_encode.append("a photo of " + label)
        to_encode.append("a " + label + " in an environment")
    return to_encode
