// fast_mel_xwide.hip -- the 2048-point ("xwide": 44.1 / 48 kHz at 25 ms) instantiations of the fbank / mfcc / plp kernel of
// fast_mel.hip, compiled as a translation unit of their own so that the two build side by side.
#define SPEECH_MEL_XWIDE_TU
#include "fast_mel.hip"
