"""Caption tokenisation for the CIDEr reward / metrics without the JVM.

The reference pipes the captions through Stanford CoreNLP 3.4.1's `edu.stanford.nlp.process.PTBTokenizer -preserveLines
-lowerCase` (a Java program, datasets/caption/metrics/tokenizer.py:26-52) and then drops a fixed list of punctuation tokens.
`PTBTokenizer.tokenize` keeps that call form and the output structure ({id: [tokenised caption, ...]}); the tokenisation
itself is a native restatement of the PTB rules that matter for image captions (the third-party algorithm is not part of the
reference tree and Java is not available in the build image, so this restatement is UNPINNED against the Java program; it is
pinned against a hand-checked TABLE of the published Penn-Treebank / CoreNLP conventions, tests/golden/ptb_rules_table.json):

  * lower-casing; one line per caption ('\\n' inside a caption becomes a space);
  * double quotes -> `` / '' , brackets -> -LRB- -RRB- -LSB- -RSB- -LCB- -RCB-, "..." kept as one token, "--" as one token;
  * , ; : @ # $ % & ? ! and sentence-final periods are split off; a period inside a token (u.s., 3.5) stays, and so do a comma
    or colon between digits (1,000 / 10:30) and an ampersand between letters (at&t); '/' and '*' are escaped (black\\/white);
  * a single quote that opens a word becomes ` (and is then dropped with the other quote tokens);
  * clitics are split the PTB way: can't -> ca n't, dog's -> dog 's, i'm -> i 'm, they're -> they 're, we've, he'll, she'd;
    cannot -> can not, gonna / wanna / gotta -> gon na / wan na / got ta;
  * hyphenated words stay one token (CoreNLP 3.4.1 default for PTB3 escaping);
  * then every token of the reference's punctuation list is removed.
"""
import re

PUNCTUATIONS = ["''", "'", "``", "`", "-LRB-", "-RRB-", "-LCB-", "-RCB-",
                ".", "?", "!", ",", ":", "-", "--", "...", ";"]  # reference tokenizer.py:21-22

_BRACKETS = {'(': '-LRB-', ')': '-RRB-', '[': '-LSB-', ']': '-RSB-', '{': '-LCB-', '}': '-RCB-'}
_CONTRACTIONS = [(re.compile(r"\b(can)(not)\b"), r"\1 \2"), (re.compile(r"\b(gon)(na)\b"), r"\1 \2"),
                 (re.compile(r"\b(wan)(na)\b"), r"\1 \2"), (re.compile(r"\b(got)(ta)\b"), r"\1 \2"),
                 (re.compile(r"\b(d)('ye)\b"), r"\1 \2"), (re.compile(r"\b(gim)(me)\b"), r"\1 \2"),
                 (re.compile(r"\b(lem)(me)\b"), r"\1 \2")]


def ptb_tokens(sentence):
    """One caption -> list of PTB-style tokens (lower-cased), before the punctuation filter."""
    s = sentence.lower().replace('\n', ' ')
    s = re.sub(r'^"', r'`` ', s)                                   # opening quotes
    s = re.sub(r'(``)', r' \1 ', s)
    s = re.sub(r'([ (\[{<])"', r'\1 `` ', s)
    s = re.sub(r"(^|[ (\[{<])'(?!(?:s|m|d|ll|re|ve|em|til|tis|twas|cause|n)\b)(?=[a-z0-9])", r'\1 ` ', s)  # opening single quote
    s = re.sub(r'\.\.\.', ' ... ', s)
    s = re.sub(r'(?<![0-9]),|,(?![0-9])', ' , ', s)                # 1,000 stays one token
    s = re.sub(r'(?<![0-9]):|:(?![0-9])', ' : ', s)                # 10:30 stays one token
    s = re.sub(r'(?<![a-z0-9])&|&(?![a-z0-9])', ' & ', s)          # at&t stays one token
    s = re.sub(r'([;@#$%])', r' \1 ', s)
    s = re.sub(r'([/*])', r'\\\1', s)                              # PTB3 escaping of / and * inside a token: black\/white
    s = re.sub(r'([^.])(\.)([\])}>"\']*)\s*$', r'\1 \2\3 ', s)     # sentence-final period only
    s = re.sub(r'([?!])', r' \1 ', s)
    s = re.sub(r"([^'])' ", r"\1 ' ", s)
    s = re.sub(r'([\]\[(){}<>])', lambda m: ' ' + _BRACKETS.get(m.group(1), m.group(1)) + ' ', s)
    s = re.sub(r'--', ' -- ', s)
    s = ' ' + s + ' '
    s = re.sub(r'"', " '' ", s)                                    # closing quotes
    s = re.sub(r"(\S)('')", r"\1 \2 ", s)
    s = re.sub(r"([^' ])('[sm]|'d|') ", r"\1 \2 ", s)              # 's 'm 'd and a trailing apostrophe
    s = re.sub(r"([^' ])('ll|'re|'ve|n't) ", r"\1 \2 ", s)
    for pattern, repl in _CONTRACTIONS:
        s = pattern.sub(repl, s)
    return s.split()


class PTBTokenizer(object):
    """Call-compatible stand-in for the reference's wrapper of the Stanford tokenizer."""

    punctuations = PUNCTUATIONS

    @classmethod
    def tokenize(cls, corpus):
        if isinstance(corpus, (list, tuple)):
            if isinstance(corpus[0], (list, tuple)):
                corpus = {i: c for i, c in enumerate(corpus)}
            else:
                corpus = {i: [c] for i, c in enumerate(corpus)}
        drop = set(cls.punctuations)
        out = {}
        for k, caps in corpus.items():
            out[k] = [' '.join(w for w in ptb_tokens(c) if w not in drop) for c in caps]
        return out
