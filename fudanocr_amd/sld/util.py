"""Label codec of the stroke-level-decomposition recognizer (reference stroke-level-decomposition/util.py:14-17,
90-116): alphabet '<12345$' ('<' start, '$' end), teacher-forcing input shifted right by one.

The reference reads its character -> stroke-list table from ./data/decompose-stroke-3755.txt at import; here the
table is an argument (`load_stroke_table(path)` reads the same file format: `<char> | <digits>` per line), so nothing
of the reference's data has to ship with the package."""
import torch

alphabet_stroke = "<12345$"
alp2num_stroke = {c: i for i, c in enumerate(alphabet_stroke)}


def get_alphabet(mode):
    if mode != "stroke":
        raise NotImplementedError("only the stroke mode (BASELINE configs[4]) is built")
    return alphabet_stroke


def load_stroke_table(path):
    table = {}
    for line in open(path, "r", encoding="utf-8"):
        parts = line.split()
        if len(parts) >= 3:
            table[parts[0]] = parts[2].strip()
    return table


def converter(mode, label, table=None, device="cuda", strokes=False):
    """-> (length [B], text_input [B, Lmax], text_all [sum L], character_level_label); tensors on `device`.
    strokes=True: `label` already holds stroke strings (synthetic batches); otherwise the first character of every
    label is looked up in `table` like util.py:96."""
    if mode != "stroke":
        raise NotImplementedError("only the stroke mode is built")
    character_level_label = label
    seqs = [(s if strokes else table[s[0]]) + "$" for s in label]
    batch, lens = len(seqs), [len(s) for s in seqs]
    max_length = max(lens)
    text_input = torch.zeros(batch, max_length, dtype=torch.long)
    for i, s in enumerate(seqs):
        for j in range(len(s) - 1):
            text_input[i][j + 1] = alp2num_stroke[s[j]]
    text_all = torch.tensor([alp2num_stroke[c] for s in seqs for c in s], dtype=torch.long)
    length = torch.tensor(lens, dtype=torch.long).to(device)
    length._focr_host = lens                   # host copy: the ragged gather needs no device -> host sync
    # rows of the [B * Lmax, classes] logits that belong to real label positions (model/transformer.py forward): built here,
    # with the rest of the encoding, so that the step itself contains no host -> device copy (engine replay)
    length._focr_idx = torch.tensor([i * max_length + j for i, n in enumerate(lens) for j in range(n)],
                                    dtype=torch.long).to(device)
    return length, text_input.to(device), text_all.to(device), character_level_label


def tensor2str(mode, tensor):
    return "".join(alphabet_stroke[int(i)] for i in tensor)
