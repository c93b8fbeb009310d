"""speech_amd -- MI355X (gfx950) implementation of the Kaldi-compatible feature extractors of erikedwards4/speech
behind that project's C ABI.  See DESIGN.md."""
from .api import (SpeechError, kaldi_spectrogram, kaldi_fbank, kaldi_mfcc, kaldi_plp, Plan, Batch, make_opts, frame_sizes)  # noqa: F401
