from .crnn import CRNN, BidirectionalLSTM  # noqa: F401
