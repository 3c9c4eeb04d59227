"""Drop-in for the reference's models/AcousticModel.py: same import path and class name,
implemented by rnn_speech_amd (HIP kernels on MI355X behind include/amdspeech.h)."""
from rnn_speech_amd.acoustic_model import AcousticModel, Session, OutOfRangeError, bucketed_order  # noqa: F401
