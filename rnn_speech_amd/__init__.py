"""Import shim: the package directory is `rnn-speech_amd/` (not an importable
identifier), so `import rnn_speech_amd` resolves its submodules there."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "rnn-speech_amd")
__path__.insert(0, _real)
__version__ = "0.1.0"
