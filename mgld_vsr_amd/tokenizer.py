"""CLIP byte-pair-encoding tokenizer (the `open_clip.tokenize` call of FrozenOpenCLIPEmbedder.forward,
ldm/modules/encoders/modules.py:174).

open_clip (an un-vendored dependency of the reference: `import open_clip`, modules.py:12) ships the algorithm as
`open_clip.tokenizer.SimpleTokenizer` over the merge table `bpe_simple_vocab_16e6.txt.gz`.  The ALGORITHM is restated here from the
published CLIP tokenizer (lower-cased, whitespace-collapsed text; GPT-2 byte <-> unicode table; the `<|startoftext|>` ... pattern;
greedy lowest-rank pair merging with the `</w>` end-of-word marker; [SOT, tokens..., EOT] padded with zeros to 77, truncated with EOT
kept last).  The MERGE TABLE is data, not code, and is not in this repository: `find_vocab()` looks for it in `$MGLD_BPE_VOCAB`, in an
installed `open_clip` / `clip` package directory (located without importing the package), or next to this file.  The empty prompt —
the only one the inference scripts use — needs no table.
"""
import gzip
import html
import importlib.util
import os
from functools import lru_cache

import torch

VOCAB_FILE = "bpe_simple_vocab_16e6.txt.gz"


@lru_cache()
def bytes_to_unicode():
    """the reversible byte -> printable-unicode table of the GPT-2 / CLIP tokenizers"""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


def get_pairs(word):
    return set(zip(word[:-1], word[1:]))


def basic_clean(text):
    try:
        import ftfy       # open_clip fixes mojibake first when ftfy is present; plain ASCII prompts are unaffected
        text = ftfy.fix_text(text)
    except ImportError:
        pass
    return html.unescape(html.unescape(text)).strip()


def whitespace_clean(text):
    return " ".join(text.split()).strip()


def find_vocab():
    cands = [os.environ.get("MGLD_BPE_VOCAB")]
    for pkg in ("open_clip", "clip"):
        try:
            spec = importlib.util.find_spec(pkg)
        except (ImportError, ValueError):
            spec = None
        if spec is not None and spec.submodule_search_locations:
            cands += [os.path.join(d, VOCAB_FILE) for d in spec.submodule_search_locations]
    cands.append(os.path.join(os.path.dirname(os.path.abspath(__file__)), VOCAB_FILE))
    for c in cands:
        if c and os.path.exists(c):
            return c
    return None


class SimpleTokenizer:
    def __init__(self, bpe_path=None, merges=None, vocab_size=49408):
        """bpe_path: the gzip'ed merge table (first line is a header); merges: alternatively the merge list itself [(a, b), ...]
        (tests).  The real table is cut to vocab_size - 256 - 2 merges, as open_clip does (49152 - 256 - 2 + 1 lines)."""
        self.byte_encoder = bytes_to_unicode()
        self.byte_decoder = {v: k for k, v in self.byte_encoder.items()}
        if merges is None:
            with gzip.open(bpe_path) as fh:
                lines = fh.read().decode("utf-8").split("\n")
            lines = lines[1:vocab_size - 256 - 2 - 256 + 1]
            merges = [tuple(m.split()) for m in lines]
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab]
        vocab += ["".join(m) for m in merges]
        vocab += ["<start_of_text>", "<end_of_text>"]
        self.encoder = dict(zip(vocab, range(len(vocab))))
        self.decoder = {v: k for k, v in self.encoder.items()}
        self.bpe_ranks = dict(zip(merges, range(len(merges))))
        self.cache = {"<start_of_text>": "<start_of_text>", "<end_of_text>": "<end_of_text>"}
        import regex
        self.pat = regex.compile(r"""<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""", regex.IGNORECASE)
        self.sot, self.eot = self.encoder["<start_of_text>"], self.encoder["<end_of_text>"]

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
                except ValueError:
                    new_word.extend(word[i:])
                    break
                new_word.extend(word[i:j])
                i = j
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
        word = " ".join(word)
        self.cache[token] = word
        return word

    def encode(self, text):
        out = []
        text = whitespace_clean(basic_clean(text)).lower()
        for token in self.pat.findall(text):
            token = "".join(self.byte_encoder[b] for b in token.encode("utf-8"))
            out.extend(self.encoder[t] for t in self.bpe(token).split(" "))
        return out

    def decode(self, tokens):
        text = "".join(self.decoder[t] for t in tokens)
        return bytearray([self.byte_decoder[c] for c in text]).decode("utf-8", errors="replace").replace("</w>", " ")

    def tokenize(self, texts, context_length=77):
        """open_clip.tokenize: LongTensor [n, context_length] = [SOT, bpe tokens..., EOT, 0...]; too-long prompts are cut and end in EOT"""
        if isinstance(texts, str):
            texts = [texts]
        res = torch.zeros(len(texts), context_length, dtype=torch.long)
        for i, t in enumerate(texts):
            toks = [self.sot] + self.encode(t) + [self.eot]
            if len(toks) > context_length:
                toks = toks[:context_length]
                toks[-1] = self.eot
            res[i, :len(toks)] = torch.tensor(toks)
        return res
