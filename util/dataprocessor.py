"""Drop-in for the label codec half of the reference's util/dataprocessor.py (:73-205).
The corpus walkers (:263-328) are host-side I/O outside the hot-path scope."""
from rnn_speech_amd import labels as _labels


class DataProcessor(object):
    clean_label = staticmethod(_labels.clean_label)
    get_str_labels = staticmethod(_labels.get_str_labels)
    get_labels_str = staticmethod(_labels.get_labels_str)
    get_str_to_one_hot_encoded = staticmethod(_labels.get_str_to_one_hot_encoded)
