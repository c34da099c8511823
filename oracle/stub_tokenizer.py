"""TEST INFRASTRUCTURE ONLY — a deterministic stand-in for the HF tokenizer `UniversalPrompting` is built around
(reference training/prompting_utils.py:18-37 uses: add_special_tokens, add_tokens, convert_tokens_to_ids, bos/eos/pad
ids, len(), and __call__(texts, truncation=...)['input_ids']).  No tokenizer files are available offline; this one maps
whitespace-separated words to ids by a fixed hash, like GPT-2's it adds neither <bos> nor <eos> on its own, and the
literal word "<bos>" maps to the bos id so that the "text already starts with bos" branch (:50-51) can be exercised."""


class StubTokenizer:
    def __init__(self, vocab=300, bos=256, eos=256):
        self.base_vocab = vocab
        self.bos_token_id, self.eos_token_id = bos, eos
        self.pad_token_id = None
        self.added = {}

    def __len__(self):
        return self.base_vocab + len(self.added)

    def add_special_tokens(self, d):
        for key, tok in d.items():
            if tok not in self.added:
                self.added[tok] = len(self)
            if key == "pad_token":
                self.pad_token_id = self.added[tok]
        return len(d)

    def add_tokens(self, toks):
        for t in toks:
            if t not in self.added:
                self.added[t] = len(self)
        return len(toks)

    def _one(self, tok):
        if tok in self.added:
            return self.added[tok]
        if tok == "<bos>":
            return self.bos_token_id
        h = 0
        for ch in tok:
            h = (h * 131 + ord(ch)) % 1000003
        return h % (self.base_vocab - 1)  # never the bos/eos id by accident when bos = vocab - 1 .. keep it simple

    def convert_tokens_to_ids(self, toks):
        if isinstance(toks, str):
            return self._one(toks)
        return [self._one(t) for t in toks]

    def __call__(self, texts, truncation=False, **kw):
        if isinstance(texts, str):
            texts = [texts]
        return {"input_ids": [[self._one(w) for w in t.split()] for t in texts]}
