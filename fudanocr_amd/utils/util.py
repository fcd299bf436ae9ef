"""String helpers of the harness (reference utils/util.py:12-24)."""
import string


def str_filt(str_, voc_type):
    alpha = {"digit": string.digits, "lower": string.digits + string.ascii_lowercase,
             "upper": string.digits + string.ascii_letters,
             "all": string.digits + string.ascii_letters + string.punctuation}[voc_type]
    if voc_type == "lower":
        str_ = str_.lower()
    return "".join(ch for ch in str_ if ch in alpha)


class AttrDict(dict):
    """Minimal stand-in for EasyDict (absent in this image): nested attribute access for the YAML."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = AttrDict(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    __setattr__ = dict.__setitem__
