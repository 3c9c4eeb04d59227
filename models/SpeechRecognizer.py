"""Drop-in for the reference's models/SpeechRecognizer.py (:21-56): the character map and
its wrapper.  Corpus discovery (load_acoustic_dataset, :58-99) is out of the hot-path scope
(SURVEY.md 8f-3); datasets are handed to AcousticModel.build_dataset as item lists."""
from rnn_speech_amd.labels import ENGLISH_CHAR_MAP


class SpeechRecognizer(object):
    def __init__(self, language="english"):
        if language != "english":
            raise ValueError("Invalid parameter 'language' for method '__init__'")
        self.char_map = ENGLISH_CHAR_MAP
        self.num_labels = len(self.char_map)

    def get_char_map(self):
        return self.char_map

    def get_char_map_length(self):
        return len(self.char_map)
