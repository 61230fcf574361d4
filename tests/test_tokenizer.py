"""CPU: the BPE tokenizer (grip_amd.clip.simple_tokenizer, algorithm of openai-CLIP's SimpleTokenizer) against an
independent implementation (transformers.CLIPTokenizer) on a synthetic merges table -- the real
bpe_simple_vocab_16e6.txt.gz is not available offline -- and the clip.tokenize switch-over via $CLIP_BPE_VOCAB."""
import gzip
import importlib
import os

import pytest

transformers = pytest.importorskip("transformers")

CORPUS = ("a photo of a forest . annual crop land , herbaceous vegetation ; highway or road ! industrial buildings "
          "pasture land permanent crop residential buildings river sea lake x x x x airplane 737-800 it's the dog's texture "
          "banded blotchy braided bubbly bumpy chequered cobwebbed cracked crosshatched crystalline dotted fibrous").split()


def _train_merges(words, n):
    """Tiny BPE trainer (most frequent pair first) just to get a plausible ranked merges table."""
    import collections
    from grip_amd.clip.simple_tokenizer import bytes_to_unicode
    b2u = bytes_to_unicode()
    vocab = collections.Counter(tuple([b2u[b] for b in w.encode()][:-1] + [b2u[w.encode()[-1]] + "</w>"]) for w in words)
    merges = []
    for _ in range(n):
        pairs = collections.Counter()
        for w, c in vocab.items():
            for p in zip(w[:-1], w[1:]):
                pairs[p] += c
        if not pairs:
            break
        best = max(sorted(pairs), key=lambda p: pairs[p])
        merges.append(best)
        new = collections.Counter()
        for w, c in vocab.items():
            out, i = [], 0
            while i < len(w):
                if i < len(w) - 1 and (w[i], w[i + 1]) == best:
                    out.append(w[i] + w[i + 1]); i += 2
                else:
                    out.append(w[i]); i += 1
            new[tuple(out)] += c
        vocab = new
    return merges


@pytest.fixture(scope="module")
def merges():
    import grip_amd  # noqa: F401
    return _train_merges(CORPUS, 120)


def test_bpe_matches_transformers_clip_tokenizer(merges):
    import grip_amd  # noqa: F401
    from grip_amd.clip.simple_tokenizer import SimpleTokenizer
    tk = SimpleTokenizer(merges=merges)
    hf = transformers.CLIPTokenizer(vocab=dict(tk.encoder), merges=[tuple(m) for m in merges])
    texts = ["a photo of a forest", "X X X X annual crop land", "a photo of a {}sea lake", "It's the dog's  texture!!",
             "737-800, an airplane", "crosshatched   cobwebbed\tcracked", "unseenword zzz qqq", ""]
    assert tk._native is not None, "libgrip_amd.so is built in this tree: the native BPE must be the one behind encode()"
    for t in texts:
        want = hf(t, add_special_tokens=False)["input_ids"]
        assert tk.encode(t) == want, (t, tk.encode(t), want)               # native (csrc/bpe.cpp)
        assert tk.encode_python(t) == want, (t, tk.encode_python(t), want)   # literal Python form
    ids = tk.encode("a photo of a river")
    assert tk.decode(ids).strip() == "a photo of a river"


def test_tokenize_switches_to_bpe_when_a_vocab_file_is_supplied(merges, tmp_path, monkeypatch):
    import grip_amd  # noqa: F401
    from grip_amd.clip import clip as gclip
    from grip_amd.clip.simple_tokenizer import SimpleTokenizer
    path = tmp_path / "bpe_test_vocab.txt.gz"
    with gzip.open(path, "wb") as f:
        f.write(("\"bpe_simple_vocab_16e6.txt#version: 0.2\n" + "\n".join(" ".join(m) for m in merges) + "\n").encode())
    monkeypatch.setenv("CLIP_BPE_VOCAB", str(path))
    monkeypatch.setattr(gclip, "_TOKENIZER", None)
    tk = SimpleTokenizer(merges=merges)
    t = gclip.tokenize(["a photo of a forest", "X X river"])
    assert t.shape == (2, 77)
    row = t[0].tolist()
    n = 2 + len(tk.encode("a photo of a forest"))
    assert row[0] == tk.sot and row[n - 1] == tk.eot and row[1: n - 1] == tk.encode("a photo of a forest") and not any(row[n:])
    assert int(t[0].argmax()) == n - 1                     # EOT is the largest id: the EOT gather of the encoders still works
    with pytest.raises(RuntimeError):
        gclip.tokenize(["forest " * 100])
    monkeypatch.setattr(gclip, "_TOKENIZER", None)          # leave the module in its default (stand-in) state
    monkeypatch.delenv("CLIP_BPE_VOCAB")
    assert gclip.tokenize(["x"])[0, 1].item() == 343


def test_native_bpe_equals_python_bpe_on_edge_cases(merges):
    """csrc/bpe.cpp against the literal Python algorithm: contractions and their priority over punctuation runs, digits one
    at a time, special tokens inside text, html entities, tabs / newlines, non-ASCII text (Unicode-aware pre-tokenisation on
    the host + native merges per pre-token), very long words, empty input, repeated calls (the per-word cache)."""
    import random

    import grip_amd  # noqa: F401
    from grip_amd.clip.simple_tokenizer import SimpleTokenizer
    tk = SimpleTokenizer(merges=merges)
    assert tk._native is not None
    texts = ["", " ", "it's we're they'll i'd you've i'm don't", "!!'s ''t 'x '", "a1b22c333 737-800", "<|startoftext|>a photo<|endoftext|> of",
             "forest&amp;river &lt;tag&gt;", "tab\tnew\nline\r\n  end", "caf\u00e9 na\u00efve \u00fcber stra\u00dfe", "\u6f22\u5b57 kanji \u0440\u0435\u043a\u0430 river",
             "x" * 300, "A PHOTO OF A FOREST", "semi;colon:colon,comma.dot", "forest forest forest river forest"]
    rnd = random.Random(3)
    alphabet = "abcdefghijklmnopqrstuvwxyz  0123456789'-.,!?"
    texts += ["".join(rnd.choice(alphabet) for _ in range(rnd.randint(1, 60))) for _ in range(200)]
    for t in texts + texts[:20]:
        assert tk.encode(t) == tk.encode_python(t), repr(t)
