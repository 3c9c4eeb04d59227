"""Text <-> label-id codec of the acoustic model (host side, integer work).

Drop-in for the static helpers of the reference's DataProcessor
(/root/reference/util/dataprocessor.py:73-95 clean_label, :121-176 get_str_labels,
:179-205 get_labels_str, :98-118 one-hot) and the ENGLISH_CHAR_MAP constant
(/root/reference/models/SpeechRecognizer.py:21-36).  80 tokens: apostrophe
contractions, double letters, a-z, A-Z (word starts), "'" and '_' (EOS == CTC blank).
"""
import logging
import string

import numpy as np

_CONTRACTIONS = ["'d", "'ll", "'m", "'nt", "'s", "s'", "'t", "'ve"]
_DOUBLES = [2 * c for c in "bcdefgiklmnoprstuz"]
ENGLISH_CHAR_MAP = (_CONTRACTIONS + _DOUBLES + list(string.ascii_lowercase)
                    + list(string.ascii_uppercase) + ["'", "_"])

_PUNCT_DROPPED = ".,?!:"
_PUNCT_TO_SPACE = "-_"


def clean_label(text):
    """Lower-case, strip, drop . , ? ! :, turn - and _ into spaces, collapse ONE level of
    double spaces (the reference runs a single replace, so triples leave a double)."""
    out = text.strip().lower()
    out = out.translate({ord(c): None for c in _PUNCT_DROPPED})
    out = out.translate({ord(c): " " for c in _PUNCT_TO_SPACE})
    return out.replace("  ", " ")


def _camel(text):
    """'the brown fox' -> 'TheBrownFox' (spaces removed, word starts upper-cased)."""
    pieces = []
    start = True
    for ch in text:
        if ch == " ":
            start = True
            continue
        pieces.append(ch.upper() if start else ch)
        start = False
    return "".join(pieces)


def get_str_labels(char_map, text, add_eos=True):
    """Greedy longest-match tokeniser: 3-char then 2-char tokens are matched on the
    lower-cased text, single chars case-sensitively; an unknown character stops the
    encoding with a warning (as the reference does); EOS = len(char_map)-1 appended."""
    index = {tok: i for i, tok in reversed(list(enumerate(char_map)))}   # first occurrence wins
    s = _camel(text)
    ids = []
    pos = 0
    while pos < len(s):
        for width in (3, 2, 1):
            if pos + width > len(s):
                continue
            tok = s[pos:pos + width]
            key = tok.lower() if width > 1 else tok
            if key in index:
                ids.append(index[key])
                pos += width
                break
        else:
            logging.warning("Unable to process label : %s", s)
            break
    if add_eos:
        ids.append(len(char_map) - 1)
    return ids


def get_labels_str(char_map, label):
    """Ids -> text: out-of-range ids are skipped, the FIRST EOS token is removed, a space
    goes in front of every capitalised token except the first, everything lower-cased."""
    toks = [char_map[int(i)] for i in label if 0 <= int(i) < len(char_map)]
    eos = char_map[-1]
    if eos in toks:
        del toks[toks.index(eos)]
    words = []
    for n, tok in enumerate(toks):
        if n and tok.isupper():
            words.append(" ")
        words.append(tok.lower())
    return "".join(words)


def get_str_to_one_hot_encoded(char_map, text, add_eos=True):
    ids = get_str_labels(char_map, text, add_eos=add_eos)
    eye = np.eye(len(char_map))
    return [eye[i].copy() for i in ids]
