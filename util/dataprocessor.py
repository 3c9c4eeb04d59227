"""Drop-in for the reference's util/dataprocessor.py: corpus discovery + file-list cache (:21-71,
:208-337) and the label codec (:73-205), implemented in rnn_speech_amd.corpus / rnn_speech_amd.labels."""
from rnn_speech_amd.corpus import (DataProcessor, DEFAULT_MIN_TEXT_LENGTH, DEFAULT_MIN_AUDIO_LENGTH,  # noqa: F401
                                   audio_duration, corpus_type, find_files)
