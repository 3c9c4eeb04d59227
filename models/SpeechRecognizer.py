"""Drop-in for the reference's models/SpeechRecognizer.py (:21-56): the per-language character map.
Corpus discovery (load_acoustic_dataset, :58-99) is out of the hot-path scope;
datasets reach AcousticModel.build_dataset as item lists (see stt.load_manifest)."""
from rnn_speech_amd.labels import ENGLISH_CHAR_MAP

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
