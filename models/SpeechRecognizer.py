"""Drop-in for the reference's models/SpeechRecognizer.py: the per-language character map (:21-56)
and the train / test split of the discovered corpora (load_acoustic_dataset, :58-99)."""
import logging
import math
import os
import random

from rnn_speech_amd.labels import ENGLISH_CHAR_MAP, clean_label

_CHAR_MAPS = {"english": ENGLISH_CHAR_MAP}


class SpeechRecognizer(object):
    """Holds the label alphabet of one language; `num_labels` includes the EOS / CTC-blank token."""

    def __init__(self, language="english"):
        try:
            self.char_map = _CHAR_MAPS[language]
        except KeyError:
            raise ValueError("Invalid parameter 'language' for method '__init__'")
        self.num_labels = len(self.char_map)

    get_char_map = lambda self: self.char_map                  # noqa: E731
    get_char_map_length = lambda self: self.num_labels         # noqa: E731

    @staticmethod
    def load_acoustic_dataset(training_dataset_dirs, test_dataset_dirs=None, training_filelist_cache=None,
                              ordered=False, train_frac=None):
        """-> (train_set, test_set), lists of [audio_file, label, audio_length].  Each entry of the two
        `*_dirs` arguments (comma separated) is a corpus directory in one of the reference's four layouts
        or -- an addition of this build -- a `path<TAB>transcript` manifest file.  Training items are
        sorted by duration (`ordered`) or shuffled; the test set is the test directories, else the tail
        `1 - train_frac` of the training list, else empty."""
        train_set = _load(training_dataset_dirs, training_filelist_cache)
        if ordered:
            train_set.sort(key=lambda item: item[2])
        else:
            random.shuffle(train_set)
        if test_dataset_dirs is not None:
            test_set = _load(test_dataset_dirs, None)
        elif train_frac is not None:
            keep = max(1, int(math.floor(train_frac * len(train_set))))
            train_set, test_set = train_set[:keep], train_set[keep:]
        else:
            test_set = []
        logging.info("Using %d files in train set", len(train_set))
        logging.info("Using %d size of test set", len(test_set))
        return train_set, test_set


def _load(spec, cache):
    from rnn_speech_amd.corpus import DataProcessor, audio_duration
    entries = [e for e in spec.replace(" ", "").split(",") if e]
    dirs = [e for e in entries if not os.path.isfile(e)]
    items = []
    for manifest in (e for e in entries if os.path.isfile(e)):
        with open(manifest) as fh:
            for line in fh:
                if "\t" in line:
                    audio, text = line.rstrip("\n").split("\t", 1)
                    items.append([audio, clean_label(text), audio_duration(audio) if os.path.exists(audio) else 0.0])
    if dirs:
        items += DataProcessor(",".join(dirs), file_cache=cache).get_dataset()
    return items
