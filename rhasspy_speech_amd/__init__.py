"""MI355X-native transcribe hot path of rhasspy-speech (drop-in for the Kaldi subprocess pipeline).

Exports mirror rhasspy_speech/__init__.py:1-6 for the classes on the path."""
from .tools import KaldiTools
from .transcribe_wav import KaldiNnet3WavTranscriber

__all__ = ["KaldiNnet3WavTranscriber", "KaldiTools"]
