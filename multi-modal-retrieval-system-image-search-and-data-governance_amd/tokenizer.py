"""Host-side tokenizers for the two text towers.

* ``ClipBPETokenizer`` -- the byte-level BPE ``clip.tokenize`` uses (reference code/search_image.py:334,
  code/test_clip.py:9, code/utils.py:87): lower-case, collapse whitespace, split with the CLIP pattern, map bytes to
  printable unicode, merge by rank with a ``</w>`` end-of-word marker, wrap in <|startoftext|> ... <|endoftext|>,
  zero-pad to 77, RuntimeError when too long.  The merge table (``bpe_simple_vocab_16e6.txt.gz`` of the ``clip``
  package, or an HF ``merges.txt``) is NOT reachable offline, so a path must be supplied; without one the package
  accepts token ids only.
* ``BertWordPieceTokenizer`` -- ``BertTokenizer(...)(texts, padding=True)['input_ids']`` as the CN pipeline calls it
  (reference code/test_taiyi.py:11,13; CLIP-Chinese/lab_chinese.py:81-93): BERT basic tokenisation (CJK characters
  isolated, lower-casing + accent stripping, punctuation split) then greedy longest-match WordPiece, [CLS] ... [SEP],
  padded with [PAD] to the longest text of the batch.  Needs the model's ``vocab.txt``.

Integer work on the host, like the reference's own tokenizers; pinned in tests/test_tokenizer.py against the
``transformers`` tokenizers (the library the reference calls) on synthetic vocabularies.
"""
import gzip
import html
import unicodedata
from functools import lru_cache
from typing import Dict, List, Sequence, Union

import torch

try:  # \\p{L} / \\p{N} classes
    import regex as re
except ImportError:  # pragma: no cover
    re = None


@lru_cache()
def bytes_to_unicode() -> Dict[int, str]:
    """Reversible byte -> printable unicode map of the GPT-2 / CLIP BPE."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


def _pairs(word):
    return {(a, b) for a, b in zip(word[:-1], word[1:])}


class ClipBPETokenizer:
    PATTERN = r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+"""

    def __init__(self, merges_path: str = None, merges: Sequence[str] = None, max_merges: int = 49152 - 256 - 2):
        if re is None:
            raise RuntimeError("the `regex` module is required for the CLIP pattern")
        if merges is None:
            if merges_path is None:
                raise RuntimeError("BPE merge table not available offline: pass merges_path (bpe_simple_vocab_16e6.txt.gz "
                                   "or merges.txt) or give token ids to tokenize()")
            opener = gzip.open if merges_path.endswith(".gz") else open
            with opener(merges_path, "rt", encoding="utf-8") as f:
                lines = f.read().split("\n")
            merges = [l for l in lines[1:] if l.strip()]          # first line is a header in both formats
        merges = [tuple(m.split()) for m in list(merges)[:max_merges]]
        self.byte_encoder = bytes_to_unicode()
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab]
        vocab += ["".join(m) for m in merges]
        vocab += ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = {t: i for i, t in enumerate(vocab)}
        self.bpe_ranks = {m: i for i, m in enumerate(merges)}
        self.cache = {"<|startoftext|>": "<|startoftext|>", "<|endoftext|>": "<|endoftext|>"}
        self.pat = re.compile(self.PATTERN, re.IGNORECASE)
        self.sot, self.eot = self.encoder["<|startoftext|>"], self.encoder["<|endoftext|>"]

    def bpe(self, token: str) -> str:
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        pairs = _pairs(word)
        if not pairs:
            return token + "</w>"
        while True:
            bigram = min(pairs, key=lambda p: self.bpe_ranks.get(p, float("inf")))
            if bigram not in self.bpe_ranks:
                break
            first, second = bigram
            new, i = [], 0
            while i < len(word):
                try:
                    j = word.index(first, i)
                except ValueError:
                    new.extend(word[i:])
                    break
                new.extend(word[i:j])
                i = j
                if word[i] == first and i < len(word) - 1 and word[i + 1] == second:
                    new.append(first + second)
                    i += 2
                else:
                    new.append(word[i])
                    i += 1
            word = tuple(new)
            if len(word) == 1:
                break
            pairs = _pairs(word)
        out = " ".join(word)
        self.cache[token] = out
        return out

    @staticmethod
    def clean(text: str) -> str:
        try:
            import ftfy                                        # the clip package fixes mojibake first; absent offline
            text = ftfy.fix_text(text)
        except ImportError:
            pass
        text = html.unescape(html.unescape(text)).strip()
        return re.sub(r"\s+", " ", text).strip().lower()

    def encode(self, text: str) -> List[int]:
        ids: List[int] = []
        for tok in self.pat.findall(self.clean(text)):
            tok = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe(tok).split(" "))
        return ids

    def __call__(self, texts: Union[str, Sequence[str]], context_length: int = 77, truncate: bool = False) -> torch.Tensor:
        if isinstance(texts, str):
            texts = [texts]
        out = torch.zeros(len(texts), context_length, dtype=torch.int32)
        for i, t in enumerate(texts):
            ids = [self.sot] + self.encode(t) + [self.eot]
            if len(ids) > context_length:
                if not truncate:
                    raise RuntimeError(f"Input {texts[i]} is too long for context length {context_length}")
                ids = ids[:context_length]
                ids[-1] = self.eot
            out[i, :len(ids)] = torch.tensor(ids, dtype=torch.int32)
        return out


# ------------------------------------------------------------------ BERT WordPiece
def _is_cjk(cp: int) -> bool:
    return ((0x4E00 <= cp <= 0x9FFF) or (0x3400 <= cp <= 0x4DBF) or (0x20000 <= cp <= 0x2A6DF) or (0x2A700 <= cp <= 0x2B73F)
            or (0x2B740 <= cp <= 0x2B81F) or (0x2B820 <= cp <= 0x2CEAF) or (0xF900 <= cp <= 0xFAFF) or (0x2F800 <= cp <= 0x2FA1F))


def _is_punct(ch: str) -> bool:
    cp = ord(ch)
    if (33 <= cp <= 47) or (58 <= cp <= 64) or (91 <= cp <= 96) or (123 <= cp <= 126):
        return True
    return unicodedata.category(ch).startswith("P")


def _is_ws(ch: str) -> bool:
    return ch in " \t\n\r" or unicodedata.category(ch) == "Zs"


def _is_control(ch: str) -> bool:
    if ch in "\t\n\r":
        return False
    return unicodedata.category(ch).startswith("C")


class BertWordPieceTokenizer:
    def __init__(self, vocab_path: str = None, vocab: Sequence[str] = None, do_lower_case: bool = True,
                 max_chars_per_word: int = 100):
        if vocab is None:
            if vocab_path is None:
                raise RuntimeError("BERT vocabulary (vocab.txt) not available offline: pass vocab_path or give token ids")
            with open(vocab_path, encoding="utf-8") as f:
                vocab = [l.rstrip("\n") for l in f]
        self.vocab = {t: i for i, t in enumerate(vocab) if t != "" or i == 0}
        self.lower = do_lower_case
        self.max_chars = max_chars_per_word
        self.unk, self.cls, self.sep, self.pad = (self.vocab[t] for t in ("[UNK]", "[CLS]", "[SEP]", "[PAD]"))

    def _basic(self, text: str) -> List[str]:
        out = []
        for ch in text:                                          # clean + isolate CJK characters
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_control(ch):
                continue
            if _is_ws(ch):
                out.append(" ")
            elif _is_cjk(cp):
                out.extend((" ", ch, " "))
            else:
                out.append(ch)
        words = []
        for w in "".join(out).strip().split():
            if self.lower:
                w = w.lower()
                w = "".join(c for c in unicodedata.normalize("NFD", w) if unicodedata.category(c) != "Mn")
            cur = ""
            for ch in w:                                         # split on punctuation
                if _is_punct(ch):
                    if cur:
                        words.append(cur)
                    words.append(ch)
                    cur = ""
                else:
                    cur += ch
            if cur:
                words.append(cur)
        return words

    def _wordpiece(self, word: str) -> List[int]:
        if len(word) > self.max_chars:
            return [self.unk]
        ids, start = [], 0
        while start < len(word):
            end, cur = len(word), None
            while start < end:
                sub = ("##" if start > 0 else "") + word[start:end]
                if sub in self.vocab:
                    cur = self.vocab[sub]
                    break
                end -= 1
            if cur is None:
                return [self.unk]
            ids.append(cur)
            start = end
        return ids

    def encode(self, text: str) -> List[int]:
        ids = [self.cls]
        for w in self._basic(text):
            ids.extend(self._wordpiece(w))
        return ids + [self.sep]

    def __call__(self, texts: Union[str, Sequence[str]], max_length: int = 512) -> torch.Tensor:
        """-> int32 [N, longest] padded with [PAD] (``padding=True``)."""
        if isinstance(texts, str):
            texts = [texts]
        rows = [self.encode(t)[:max_length] for t in texts]
        T = max(len(r) for r in rows)
        out = torch.full((len(rows), T), self.pad, dtype=torch.int32)
        for i, r in enumerate(rows):
            out[i, :len(r)] = torch.tensor(r, dtype=torch.int32)
        return out
