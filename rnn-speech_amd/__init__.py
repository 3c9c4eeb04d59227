"""rnn-speech_amd -- MI355X-native acoustic-model training path (MFCC/fbank ->
stacked LSTM -> CTC) behind the reference's models.AcousticModel /
util.audioprocessor.AudioProcessor surface.  Import it as `rnn_speech_amd`
(see ../rnn_speech_amd/__init__.py); the compute is csrc/*.hip behind the C ABI
declared in ../include/amdspeech.h."""
