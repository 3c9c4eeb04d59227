"""Drop-in for the reference's util/audioprocessor.py: AudioProcessor on the HIP front end."""
from rnn_speech_amd.audioprocessor import AudioProcessor, FRAME_SIZE, FRAME_STRIDE  # noqa: F401
