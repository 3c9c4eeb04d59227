"""ORACLE (test infrastructure only) -- label codec restatement.

CPU restatement of the reference's text <-> label-id codec.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this
package; the product path never does.

Follows:
  * char map ............ /root/reference/models/SpeechRecognizer.py:21-36
  * clean_label ......... /root/reference/util/dataprocessor.py:73-95
  * get_str_labels ...... /root/reference/util/dataprocessor.py:121-176
  * get_labels_str ...... /root/reference/util/dataprocessor.py:179-205

Pinned by the reference's own known answers (util/test_dataProcessor.py:132-149,
models/test_LanguageModel.py:72-73) and by fixtures generated from the imported
reference codec (tests/golden/labels.json, tools/make_golden.py).
"""

# 80 tokens; index 79 ('_') is both the end-of-sentence token and the CTC blank.
CHAR_MAP = (
    ["'d", "'ll", "'m", "'nt", "'s", "s'", "'t", "'ve"]
    + [c + c for c in "bcdefgiklmnoprstuz"]
    + [chr(ord("a") + i) for i in range(26)]
    + [chr(ord("A") + i) for i in range(26)]
    + ["'", "_"]
)
assert len(CHAR_MAP) == 80


def clean_label(text):
    text = text.strip().lower()
    for ch in ".,?!:":
        text = text.replace(ch, "")
    text = text.replace("-", " ").replace("_", " ")
    return text.replace("  ", " ")


def str_to_labels(char_map, text, add_eos=True):
    """CamelCase the words, then greedy longest match (3, 2, 1 chars)."""
    camel = []
    upper_next = True
    for ch in text:
        if ch == " ":
            upper_next = True
        elif upper_next:
            camel.append(ch.upper())
            upper_next = False
        else:
            camel.append(ch)
    s = "".join(camel)
    out = []
    i = 0
    n = len(s)
    while i < n:
        hit = False
        for width in (3, 2):
            if n - i >= width:
                tok = s[i:i + width].lower()
                if tok in char_map:
                    out.append(char_map.index(tok))
                    i += width
                    hit = True
                    break
        if hit:
            continue
        tok = s[i:i + 1]
        if tok in char_map:
            out.append(char_map.index(tok))
            i += 1
            continue
        break  # the reference logs a warning and stops encoding here
    if add_eos:
        out.append(len(char_map) - 1)
    return out


def labels_to_str(char_map, label):
    toks = [char_map[k] for k in label if 0 <= k < len(char_map)]
    if char_map[-1] in toks:
        toks.remove(char_map[-1])  # first EOS only
    out = []
    for i, tok in enumerate(toks):
        if i != 0 and tok.isupper():
            out.append(" ")
        out.append(tok.lower())
    return "".join(out)
