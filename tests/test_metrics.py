"""The self-critical reward without the caller supplying it (SURVEY next-row N2): native CIDEr-D pinned against the
reference's own implementation (fixture G14, tests/golden/make_golden.py), the PTB-style tokenizer on hand-checked PTB
conventions (the Java tokenizer itself cannot run in the build image: that part is unpinned), and cider_reward_fn wiring."""
import json
import os

import numpy as np
import torch

from tests.helpers import GOLDEN


def _g14():
    g = json.load(open(os.path.join(GOLDEN, "cider_g14.json")))
    fix = lambda d: {int(k): v for k, v in d.items()}
    return g, fix(g["train"]), fix(g["gts"]), fix(g["res"])


def test_cider_matches_reference_implementation():
    from grit_amd.datasets.caption.metrics import Cider
    g, train, gts, res = _g14()
    mean, scores = Cider(train).compute_score(gts, res)
    np.testing.assert_allclose(scores, g["scores"], rtol=1e-12, atol=1e-12)
    assert abs(mean - g["mean"]) < 1e-12
    mean2, scores2 = Cider().compute_score(gts, res)  # no training corpus: statistics of the call itself
    np.testing.assert_allclose(scores2, g["scores_nocorpus"], rtol=1e-12, atol=1e-12)
    assert abs(mean2 - g["mean_nocorpus"]) < 1e-12
    # pre-cooked references (what a training loop would keep per image) give the same numbers
    c = Cider(train)
    cooked = {k: c.cook(v) for k, v in gts.items()}
    np.testing.assert_allclose(c.compute_score(cooked, res)[1], g["scores"], rtol=1e-12, atol=1e-12)


def test_ptb_style_tokenizer_conventions():
    from grit_amd.datasets.caption.metrics import PTBTokenizer
    tok = lambda s: PTBTokenizer.tokenize([s])[0][0]
    assert tok("A man's dog can't sit, on the (red) bench.") == "a man 's dog ca n't sit on the red bench"
    assert tok('Two "big" dogs -- they\'re running... fast!') == "two big dogs they 're running fast"
    assert tok("U.S. flag at 3.5 feet; it's a well-known sign") == "u.s. flag at 3.5 feet it 's a well-known sign"
    assert tok("I cannot see: we've gone") == "i can not see we 've gone"
    # the reference's call forms: list of strings, list of lists, dict
    assert PTBTokenizer.tokenize([["a cat.", "A dog!"], ["x"]]) == {0: ["a cat", "a dog"], 1: ["x"]}
    assert PTBTokenizer.tokenize({7: ["Hello, world"]}) == {7: ["hello world"]}


def test_ptb_tokenizer_against_the_hand_checked_rule_table():
    """tests/golden/ptb_rules_table.json: 43 rows of published PTB / CoreNLP-3.4.1 conventions (clitics, quotes, brackets,
    hyphens, numbers with separators, slash escaping, the punctuation the CIDEr pipeline strips), each with the rule it
    illustrates.  A table checked by hand -- not output of the Java tokenizer, which cannot run in the build image."""
    from grit_amd.datasets.caption.metrics import PTBTokenizer
    from grit_amd.datasets.caption.metrics.tokenizer import ptb_tokens
    table = json.load(open(os.path.join(GOLDEN, "ptb_rules_table.json")))
    assert len(table["rows"]) >= 40
    for row in table["rows"]:
        assert ptb_tokens(row["text"]) == row["tokens"], row["rule"]
        assert PTBTokenizer.tokenize([row["text"]])[0][0] == row["after_punctuation_filter"], row["rule"]


def test_cider_reward_fn_default_tokenizer():
    """engine.caption_engine.cider_reward_fn with the native tokenizer: tokens -> decode -> tokenise -> CIDEr -> [B, beam]."""
    from grit_amd.datasets.caption.metrics import Cider, PTBTokenizer
    from grit_amd.engine.caption_engine import cider_reward_fn

    class Field(object):  # the slice of TextField.decode the reward needs
        itos = ['<unk>', '<pad>', '<bos>', '<eos>', 'a', 'dog', 'cat', 'sits', 'runs']

        def decode(self, rows):
            out = []
            for r in rows.tolist():
                words = []
                for t in r:
                    if t == 3:
                        break
                    words.append(self.itos[t])
                out.append(' '.join(words))
            return out

    train = PTBTokenizer.tokenize({0: ["a dog sits.", "A dog runs"], 1: ["a cat sits", "a cat runs!"]})
    fn = cider_reward_fn(Cider(train), Field())
    tokens = torch.tensor([[[4, 5, 7, 3], [4, 6, 7, 3]], [[4, 6, 8, 3], [4, 5, 3, 3]]])
    batch = {'captions': [["A dog sits.", "a dog runs"], ["a cat runs", "A cat sits"]]}
    reward = fn(tokens, batch)
    assert reward.shape == (2, 2) and reward.dtype == torch.float32
    assert reward[0, 0] > reward[0, 1] and reward[1, 0] > reward[1, 1]  # the matching caption scores higher
