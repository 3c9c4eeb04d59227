"""Drop-in for the reference's util/hyperparams.py."""
from rnn_speech_amd.hyperparams import HyperParameterHandler, read_config_file  # noqa: F401
