"""Byte-level BPE tokenizer with the algorithm of openai-CLIP's `clip.simple_tokenizer.SimpleTokenizer`
(the tokenizer behind the `clip.tokenize` calls of the reference, e.g. models/clip_encoders.py:60,
utils/clip_pseudolabels.py:25): lower-cased, whitespace-cleaned text -> regex pre-tokens -> bytes mapped to
printable unicode -> merges ranked by the vocabulary file -> ids; `<|startoftext|>` / `<|endoftext|>` are the
last two ids.  The merges file (`bpe_simple_vocab_16e6.txt.gz`, 262 145 lines in the upstream package) is NOT
available offline: point $CLIP_BPE_VOCAB at it (or drop it next to this file) and `clip.tokenize` switches from
the documented word-hash stand-in to real BPE.  Upstream also runs `ftfy.fix_text`, which is not installed
here; text that needs mojibake repair will tokenize differently.
tests/test_tokenizer.py cross-checks this implementation against transformers.CLIPTokenizer on a synthetic
merges table.

`encode` runs natively when libgrip_amd.so is built (csrc/bpe.cpp behind grip_bpe_* of the C ABI: pre-tokenisation of
ASCII text and the merge loop in C++, results cached per word; non-ASCII text is pre-tokenised by the Unicode-aware
pattern here and merged natively per pre-token).  `encode_python` is the literal Python form of the same algorithm; the
test suite holds the two (and the independent HF implementation) equal.
"""
import gzip
import html
import os
from functools import lru_cache

import regex as re

_PAT = r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+"""


@lru_cache()
def bytes_to_unicode():
    """Reversible map of the 256 byte values to printable unicode characters (GPT-2 / CLIP convention)."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    cs = bs[:]
    n = 0
    for b in range(2 ** 8):
        if b not in bs:
            bs.append(b)
            cs.append(2 ** 8 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


def get_pairs(word):
    return set(zip(word[:-1], word[1:]))


def basic_clean(text):
    return html.unescape(html.unescape(text)).strip()


def whitespace_clean(text):
    return re.sub(r"\s+", " ", text).strip()


def default_bpe_path():
    p = os.environ.get("CLIP_BPE_VOCAB")
    if p:
        return p
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "bpe_simple_vocab_16e6.txt.gz")


def read_merges(bpe_path, n_merges=49152 - 256 - 2):
    opener = gzip.open if bpe_path.endswith(".gz") else open
    with opener(bpe_path, "rb") as f:
        lines = f.read().decode("utf-8").split("\n")
    return [tuple(m.split()) for m in lines[1: 1 + n_merges] if len(m.split()) == 2]


class SimpleTokenizer:
    def __init__(self, bpe_path=None, merges=None):
        self.byte_encoder = bytes_to_unicode()
        self.byte_decoder = {v: k for k, v in self.byte_encoder.items()}
        if merges is None:
            merges = read_merges(bpe_path or default_bpe_path())
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab]
        for m in merges:
            vocab.append("".join(m))
        vocab.extend(["<|startoftext|>", "<|endoftext|>"])
        self.encoder = dict(zip(vocab, range(len(vocab))))
        self.decoder = {v: k for k, v in self.encoder.items()}
        self.bpe_ranks = dict(zip(merges, range(len(merges))))
        self.cache = {"<|startoftext|>": "<|startoftext|>", "<|endoftext|>": "<|endoftext|>"}
        self.pat = re.compile(_PAT, re.IGNORECASE)
        self.sot, self.eot = self.encoder["<|startoftext|>"], self.encoder["<|endoftext|>"]
        self._native = self._make_native(merges)

    def _make_native(self, merges):
        """grip_bpe handle over the same merges table, or None when the library is not built (the tokenizer is host code: it
        must keep working for vocabulary inspection on a machine without the extension)."""
        try:
            import ctypes
            from .. import native
            lib = native.lib()
            blob = "\n".join(f"{a} {b}" for a, b in merges).encode("utf-8")
            h = ctypes.c_void_p()
            native.check(lib.grip_bpe_create(blob, len(blob), ctypes.byref(h)))
            sot, eot, vs = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
            native.check(lib.grip_bpe_special_ids(h, ctypes.byref(sot), ctypes.byref(eot), ctypes.byref(vs)))
            assert (sot.value, eot.value, vs.value) == (self.sot, self.eot, len(self.encoder))
            self._lib, self._check, self._buf = lib, native.check, (ctypes.c_int32 * 4096)()
            return h
        except Exception:
            return None

    def __del__(self):
        try:
            if getattr(self, "_native", None):
                self._lib.grip_bpe_destroy(self._native)
        except Exception:
            pass

    def bpe(self, token):
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        pairs = get_pairs(word)
        if not pairs:
            return token + "</w>"
        while True:
            bigram = min(pairs, key=lambda pair: self.bpe_ranks.get(pair, float("inf")))
            if bigram not in self.bpe_ranks:
                break
            first, second = bigram
            new_word, i = [], 0
            while i < len(word):
                try:
                    j = word.index(first, i)
                    new_word.extend(word[i:j])
                    i = j
                except ValueError:
                    new_word.extend(word[i:])
                    break
                if word[i] == first and i < len(word) - 1 and word[i + 1] == second:
                    new_word.append(first + second)
                    i += 2
                else:
                    new_word.append(word[i])
                    i += 1
            word = tuple(new_word)
            if len(word) == 1:
                break
            pairs = get_pairs(word)
        out = " ".join(word)
        self.cache[token] = out
        return out

    def encode(self, text):
        if self._native is None:
            return self.encode_python(text)
        import ctypes
        text = whitespace_clean(basic_clean(text)).lower()
        n = ctypes.c_int()
        if text.isascii():
            raw = text.encode("ascii")
            if 2 * len(raw) + 8 > len(self._buf):
                self._buf = (ctypes.c_int32 * (2 * len(raw) + 8))()
            self._check(self._lib.grip_bpe_encode_ascii(self._native, raw, len(raw), self._buf, len(self._buf), ctypes.byref(n)))
            return list(self._buf[: n.value])
        ids = []
        for token in re.findall(self.pat, text):
            if token in ("<|startoftext|>", "<|endoftext|>"):
                ids.append(self.encoder[token])
                continue
            raw = token.encode("utf-8")
            if len(raw) + 8 > len(self._buf):
                self._buf = (ctypes.c_int32 * (len(raw) + 8))()
            self._check(self._lib.grip_bpe_encode_word(self._native, raw, len(raw), self._buf, len(self._buf), ctypes.byref(n)))
            ids.extend(self._buf[: n.value])
        return ids

    def encode_python(self, text):
        ids = []
        text = whitespace_clean(basic_clean(text)).lower()
        for token in re.findall(self.pat, text):
            token = "".join(self.byte_encoder[b] for b in token.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe(token).split(" "))
        return ids

    def decode(self, tokens):
        text = "".join(self.decoder[t] for t in tokens)
        return bytearray(self.byte_decoder[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")
